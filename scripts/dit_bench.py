"""Time the DiT-XL/2 forward (fastgen/configs/net.py:124-127: hidden 1152, 28 blocks, 16 heads x 72, 256 tokens) on one MI355X.
Algorithmic work per image and forward: 28 x (qkv 2.04 + attention 0.30 + proj 0.68 + MLP 5.44) GFLOP + embeddings = 237 GFLOP.
    python scripts/dit_bench.py [--mode=bf16x3|bf16] [batch ...]"""
import sys
import time

import torch

from fastgen_amd.networks.DiT.network import DiT

MODE = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--mode=")), "bf16x3")
D, depth, T, Hd = 1152, 28, 256, 4608
GF = depth * (2 * T * D * 3 * D + 4 * T * T * D + 2 * T * D * D + 4 * T * D * Hd + 2 * D * 6 * D) / 1e9
net = DiT(compute_dtype=MODE).cuda().eval()
with torch.no_grad():
    g = torch.Generator().manual_seed(0)
    for n, p in net.named_parameters():  # O(1) signal in every branch (the reference zero-initialises the adaLN layers)
        p.copy_((torch.randn(p.shape, generator=g) * (0.1 if p.dim() == 1 else p.shape[-1] ** -0.5)).cuda())
print(f"compute mode {MODE}; {GF:.1f} GFLOP / image / forward")
for B in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [64, 256]:
    x = torch.randn(B, 4, 32, 32, device="cuda")
    t = torch.rand(B, dtype=torch.float64, device="cuda") * 0.99
    c = torch.randint(0, 1000, (B,), device="cuda")
    with torch.inference_mode():
        for _ in range(2):
            net(x, t, condition=c)
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), 4
        for _ in range(n):
            net(x, t, condition=c)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:4d}: {dt * 1e3:8.2f} ms / forward  {B / dt:8.1f} img/s  {B * GF / dt / 1e3:7.1f} algorithmic TFLOP/s")
