"""Time the causal video DiT at the reference's Self-Forcing / CausVid shape (fastgen/configs/experiments/WanT2V/config_sf.py:21,43:
Wan2.1-T2V-1.3B, latents [16, 21, 60, 104] = 480p, chunk of 3 frames, t_list of 4 steps): one network call of the student loop on
chunk k (4680 query tokens over (3 k + 3) * 1560 cached + own keys), and the whole 21-frame 4-step loop (7 chunks x (4 + 1) calls).
Random-init weights of the 1.3B architecture, synthetic latents and text embeddings.
    python scripts/wan_bench.py [--layers=30] [--loop]"""
import sys
import time

import torch

from fastgen_amd.methods.distribution_matching.causvid import CausVidModel
from fastgen_amd.networks.Wan.network_causal import CausalWan

LAYERS = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--layers=")), 30)
net = CausalWan(num_layers=LAYERS).cuda().eval()
D, Fd, H, fs, Lt = 1536, 8960, 12, 1560, 512
x = torch.randn(1, 16, 21, 60, 104, device="cuda")
text = torch.randn(1, Lt, 4096, device="cuda")
t = torch.tensor([0.7], dtype=torch.float64, device="cuda")
with torch.inference_mode():
    for k in range(7):  # fill the cache chunk by chunk, timing the call on each chunk
        xs = x[:, :, 3 * k: 3 * k + 3]
        net(xs, t, condition=text, cur_start_frame=3 * k, store_kv=True, is_ar=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            net(xs, t, condition=text, cur_start_frame=3 * k, store_kv=True, is_ar=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        L, Lkv = 3 * fs, (3 * k + 3) * fs
        gf = LAYERS * (2 * L * D * (4 * D + 2 * Fd + 2 * D) + 4 * L * Lkv * D + 4 * L * Lt * D) / 1e9
        print(f"chunk {k}: {dt * 1e3:8.2f} ms / call  ({gf:7.1f} GFLOP: {gf / dt / 1e3:6.1f} TFLOP/s)", flush=True)
    if "--loop" in sys.argv:
        noise = torch.randn(1, 16, 21, 60, 104, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        CausVidModel.generator_fn(net, noise, student_sample_steps=4, t_list=[0.999, 0.937, 0.833, 0.624, 0.0], condition=text)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"21-frame 4-step CausVid loop (35 network calls): {dt:.3f} s = {21 / dt:.2f} latent frames / s")
