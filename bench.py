"""Headline benchmark: images/sec of 4-step DMD2-style student sampling with the EDM CIFAR-10 U-Net
(BASELINE.json configs[1]: batch 512 per MI355X, synthetic latents, random-init weights of that architecture).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: FastGenModel.generator_fn(net, noise[512,3,32,32], 4 steps,
'sde') on each rank = one hipGraph replay of latents -> 4 x (U-Net forward, re-noise).  Inputs are resident in HBM
before the timed region.  Ranks are independent replicas (sampling shards by image, no collective on the data path);
the timed region is bracketed by barrier + device synchronize on both sides and the MAX over ranks is reported.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     — the dominant kernel (fused GN+SiLU+conv3x3 at 32x32) timed live with HIP events on its launch stream
  cpu_baseline — the CPU oracle (torch fp32 restatement of the reference path, oracle/edm_ref.py) on this host's cores
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE_FWD = 42.383  # BASELINE.md section 2 (conv 42.034 + attention 0.340 + linear 0.009), 2*MAC
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md:41-43; bf16x3 issues three bf16 MFMAs per algorithmic product
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3}
DOMINANT_KERNEL = {
    "bf16x3": "conv3_x3ws_kernel<RES_NONE,32x32> (GroupNorm+SiLU -> hi/lo split -> 3x3 conv as 3 bf16 16x16x32 MFMAs per product -> "
              "bias/temb/residual/scale + GN statistics)",
    "bf16": "conv3_ws_kernel<RES_NONE,32x32> (GroupNorm+SiLU -> 3x3 conv -> bias/temb/residual/scale + GN statistics)",
    "fp32": "conv_fused_kernel<float,3,PRO_GN_SILU,RES_NONE,32x32>",
}
# how the JSON line names the arithmetic: bf16x3 is the fp32-grade mode (fp32 tensors and accumulation, every conv product from
# hi/lo-split operands on the bf16 matrix pipe, held to the exact-fp32 parity tolerance in tests/test_gpu_parity.py)
DTYPE_LABEL = {"bf16x3": "fp32 (tensors, accumulate) with split-bf16 x3 MFMA conv products", "fp32": "fp32", "bf16": "bf16"}
TRAFFIC_FILE = {"bf16x3": "r03_x3_pmc_dominant_kernel.json", "bf16": "r01_ws_pmc_dominant_kernel.json", "fp32": "none"}
EDM_CIFAR10 = dict(img_resolution=32, img_channels=3, label_dim=10, sigma_shift=0.0, sigma_data=0.5,
                   model_type="SongUNet", augment_dim=9, model_channels=128, channel_mult=[2, 2, 2],
                   channel_mult_noise=1, embedding_type="positional", encoder_type="standard",
                   decoder_type="standard", resample_filter=[1, 1], dropout=0.0, label_dropout=0, r_timestep=False,
                   drop_precond=None)


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_dist(world: int, backend: str):
    """One process per GPU; rendezvous from MASTER_ADDR/PORT (torch.distributed.run sets them)."""
    import torch.distributed as dist

    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, init_method="env://")
    return dist


def timed_region(step_fn, steps: int, warmup: int, world: int, sync_fn, device=None) -> float:
    """W untimed warm-up steps, then EXACTLY `steps` steps between barrier+sync fences; returns the max over ranks
    of the elapsed seconds.  Shared by the GPU bench and the gloo CPU test (tests/test_dist_gloo.py)."""
    import torch.distributed as dist

    for _ in range(warmup):
        step_fn()
    sync_fn()
    if world > 1:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    if world > 1:
        dist.barrier()
    sync_fn()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def host_cores():
    """(threads to use, description) - the CPU threads this process may really use: min(affinity mask, cgroup cpu.max quota, 16 = the
    per-GPU host share of the benchmark boxes).  Asking torch for more threads than the quota allows makes the CPU leg crawl."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    n = max(1, min(aff, quota if quota is not None else aff, 16))
    return n, f"min(affinity mask {aff}, cgroup cpu.max {quota if quota is not None else 'max'}, per-GPU host share 16) of {os.cpu_count()} logical CPUs"


def cpu_baseline(budget_s: float = 20.0):
    """The oracle (kind 'port': the repo's torch-CPU restatement, pinned to reference fixtures) on this host:
    EDM CIFAR-10, fp32, batch 16, 4-step 'sde' with injected eps — BASELINE.md section 4."""
    from oracle import edm_ref as R

    cores, cores_how = host_cores()
    torch.set_num_threads(cores)
    cfg = R.CIFAR10
    sd = R.random_state_dict(cfg, seed=1234)
    B = 16
    noise = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    cond = torch.nn.functional.one_hot(torch.arange(B) % 10, 10).float()
    eps = [torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(s)) for s in (1, 2, 3)]
    R.generator_fn(sd, cfg, noise, cond, 4, sample_type="sde", eps_list=eps)  # warm-up
    times, t0 = [], time.perf_counter()
    while len(times) < 5 or (time.perf_counter() - t0 < budget_s and len(times) < 9):  # at least 5 batches: the MEDIAN batch is reported
        t1 = time.perf_counter()
        R.generator_fn(sd, cfg, noise, cond, 4, sample_type="sde", eps_list=eps)
        times.append(time.perf_counter() - t1)
        if time.perf_counter() - t0 > 3 * budget_s:
            break
    dt = time.perf_counter() - t0
    med = sorted(times)[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "img/s", "cores": cores, "kind": "port", "cores_granted": cores_how,
            "sample": f"median of {len(times)} batches of 16 images (fastest {B / min(times):.1f}, slowest {B / max(times):.1f} img/s), "
                      f"4-step sde, fp32 torch-CPU oracle ({dt:.1f} s)"}


def transformer_rows(dev):
    """Secondary measurements (not the headline metric): wall time of one forward, max-synchronised, after warm-up."""
    import time

    from fastgen_amd.networks.DiT.network import DiT
    from fastgen_amd.networks.Wan.network_causal import CausalWan

    out = {}

    def timeit(fn, n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n

    # DiT-XL/2 (fastgen/configs/net.py:124-127): hidden 1152, 28 blocks, 16 heads x 72, 256 tokens; 237 GFLOP / image / forward
    D, depth, T, Hd, B = 1152, 28, 256, 4608, 256
    gf = depth * (2 * T * D * 3 * D + 4 * T * T * D + 2 * T * D * D + 4 * T * D * Hd + 2 * D * 6 * D) / 1e9
    net = DiT(compute_dtype="bf16").to(dev).eval()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for _, p in net.named_parameters():  # O(1) signal in every branch (the reference zero-initialises the adaLN layers)
            p.copy_((torch.randn(p.shape, generator=g) * (0.1 if p.dim() == 1 else p.shape[-1] ** -0.5)).to(dev))
    x = torch.randn(B, 4, 32, 32, device=dev)
    t = torch.rand(B, dtype=torch.float64, device=dev) * 0.99
    c = torch.randint(0, 1000, (B,), device=dev)
    with torch.inference_mode():
        for _ in range(3):
            net(x, t, condition=c)
        dt = timeit(lambda: net(x, t, condition=c), 5)
    out["dit_xl2_forward"] = {"batch": B, "dtype": "bf16", "ms": round(dt * 1e3, 2), "img_per_s": round(B / dt, 1),
                              "algorithmic_tflops": round(B * gf / dt / 1e3, 1), "frac_of_2.5PF": round(B * gf / dt / 1e3 / 2500.0, 4),
                              "parity": "restated-timm (oracle pinned to the reference's DiT class on restated timm Attention / Mlp / PatchEmbed)"}
    # the 4-step student loop on it as ONE fg_dit_sampler_run call (one hipGraph replay per batch)
    from fastgen_amd.methods.model import FastGenModel

    onehot = torch.nn.functional.one_hot(c, 1000).float()
    loop = lambda: FastGenModel.generator_fn(net, x, condition=onehot, student_sample_steps=4, student_sample_type="sde", seed=1)
    for _ in range(2):
        loop()
    dtl = timeit(loop, 3)
    out["dit_xl2_4step_student_loop"] = {"batch": B, "dtype": "bf16", "ms": round(dtl * 1e3, 2), "img_per_s": round(B / dtl, 1),
                                         "frac_of_2.5PF": round(4 * B * gf / dtl / 1e3 / 2500.0, 4), "graph": True, "parity": "restated-timm"}
    del net
    torch.cuda.empty_cache()
    # causal video DiT 1.3B at 480p (fastgen/configs/experiments/WanT2V/config_sf.py: latents [16, 21, 60, 104], chunks of 3 frames):
    # one network call of the student loop on chunk k = 4 680 query tokens over (k + 1) * 4 680 cached + own keys
    Dw, Fd, fs, Lt, layers = 1536, 8960, 1560, 512, 30
    net = CausalWan(num_layers=layers).to(dev).eval()
    xv = torch.randn(1, 16, 21, 60, 104, device=dev)
    text = torch.randn(1, Lt, 4096, device=dev)
    tv = torch.tensor([0.7], dtype=torch.float64, device=dev)
    calls = {}
    with torch.inference_mode():
        for k in range(7):
            xs = xv[:, :, 3 * k: 3 * k + 3]
            call = lambda: net(xs, tv, condition=text, cur_start_frame=3 * k, store_kv=True, is_ar=True)
            call()
            if k in (0, 6):
                dtk = timeit(call, 3)
                L, Lkv = 3 * fs, (3 * k + 3) * fs
                gfk = layers * (2 * L * Dw * (4 * Dw + 2 * Fd + 2 * Dw) + 4 * L * Lkv * Dw + 4 * L * Lt * Dw) / 1e9
                calls[f"chunk{k}"] = {"ms": round(dtk * 1e3, 2), "algorithmic_tflops": round(gfk / dtk / 1e3, 1)}
    out["causal_video_dit_1p3b_480p_call"] = dict(dtype="bf16", batch=1, parity="unpinned (diffusers' arithmetic restated: oracle/wan_ref.py)", **calls)
    # the whole 21-frame 4-step CausVid student loop (7 chunks x (4 denoising calls + 1 cache-fill call)) as ONE fg_wan_sampler_run call,
    # one hipGraph per chunk: latent frames per second
    from fastgen_amd.methods.distribution_matching.causvid import CausVidModel

    vloop = lambda: CausVidModel.generator_fn(net, xv, student_sample_steps=4, condition=text, student_sample_type="sde", seed=1)
    vloop()
    dtv = timeit(vloop, 2)
    out["causal_video_dit_1p3b_480p_4step_loop"] = {"dtype": "bf16", "batch": 1, "latent_frames": 21, "network_calls": 35, "s": round(dtv, 3),
                                                    "latent_frames_per_s": round(21 / dtv, 2), "graph": "one per chunk", "parity": "unpinned"}
    del net
    torch.cuda.empty_cache()
    return out


def result_line(value, world, steps, warmup, dt, dtype, batch, sample_steps, graph, roofline, cpu):
    """The ONE JSON line of the driver contract (fields and meanings: the task brief, 'Measurement')."""
    return {
        "metric": "images/sec at 4-step distilled (DMD2) sampling", "value": round(value, 2), "unit": "img/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL.get(dtype, dtype), "data": "synthetic",
        "config": {"workload": f"EDM CIFAR-10 32x32 SongUNet (55.7M params) DMD2 {sample_steps}-step 'sde' sampling, "
                               f"batch={batch} per GPU", "global_batch": batch * world, "sample_steps": sample_steps,
                   "parallelism": f"replicas x{world} (no data-path collective)", "graph": graph},
        "roofline": roofline, "cpu_baseline": cpu,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="images per GPU per step")
    ap.add_argument("--sample-steps", type=int, default=4)
    ap.add_argument("--dtype", choices=["bf16x3", "fp32", "bf16"], default="bf16x3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bf16 line and the transformer-row timings reported beside the headline")
    args = ap.parse_args()

    rank, local_rank, world = dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with python -m torch.distributed.run "
                             f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the fastgen_amd path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = init_dist(world, "nccl")

    from fastgen_amd import _lib
    from fastgen_amd.methods.model import FastGenModel
    from fastgen_amd.networks.EDM.network import EDMPrecond

    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)  # seed + rank, utils/basic_utils.py:140-143
    noise = torch.randn(B, 3, 32, 32, device=dev, generator=gen)
    cond = torch.nn.functional.one_hot(torch.arange(B, device=dev) % 10, 10).float()

    def measure(dtype: str):
        """(img/s, seconds of the timed region, roofline block) of one compute mode: W warm-up + exactly K timed steps."""
        # random-init weights of the named architecture; seeded N(0, 1/fan_in) so activations stay O(1) (the reference's
        # default init scales the residual branches by 1e-5, which makes the data trivial)
        net = EDMPrecond(compute_dtype=dtype, **EDM_CIFAR10).randomize_parameters_(seed=1234).to(dev).eval()
        step_seed = [0]

        def step():
            step_seed[0] += 1
            return FastGenModel.generator_fn(net, noise, condition=cond, student_sample_steps=args.sample_steps,
                                             student_sample_type="sde", seed=step_seed[0] * world + rank,
                                             use_graph=not args.no_graph)

        dt = timed_region(step, args.steps, args.warmup, world, lambda: torch.cuda.synchronize(dev), dev)
        value = B * args.steps * world / dt
        out = step()
        assert torch.isfinite(out).all(), "non-finite samples"
        # ---- roofline of the dominant kernel, measured live with HIP events on its launch stream -------------------
        roof = None
        if rank == 0:
            with torch.inference_mode():
                _, h = net._engine(dev)
            L = _lib.lib()
            _lib.check(L.fg_edm_profile_begin(h))
            for _ in range(2):
                step()
            n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            _lib.check(L.fg_edm_profile_end(h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)))
            ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
            peak = PEAK_TFLOPS[dtype]
            # HBM bytes per launch of that kernel come from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see the file);
            # they cannot be collected from inside this process, so the committed profile of the same command is quoted
            traffic, traffic_source = None, None
            tfile = os.path.join(ROOT, "profiles", TRAFFIC_FILE[dtype])
            if B == 512 and os.path.exists(tfile):
                traffic = int(json.load(open(tfile))["hbm_bytes_per_launch"])
                traffic_source = f"profiles/{TRAFFIC_FILE[dtype]} (rocprofv3 --pmc passes of this command, 2 x FETCH_SIZE + WRITE_SIZE per launch; not re-collected in this run)"
            whole = value / world * GFLOP_PER_IMAGE_FWD * args.sample_steps / 1e3
            roof = {"bound": "mfma", "kernel": DOMINANT_KERNEL[dtype],
                    "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_source, "launches": n.value, "avg_launch_us": round(ms.value * 1e3 / max(n.value, 1), 2),
                    "avg_launch_gflop": round(fl.value / max(n.value, 1) / 1e9, 2),
                    "whole_step": {"achieved": round(whole, 2), "frac": round(whole / peak, 4),
                                   "note": "all kernels of the 4-step graph, 42.383 GFLOP/image/forward"}}
            if dtype == "bf16x3":
                roof["peak_note"] = ("algorithmic FLOP/s against 2.5 PFLOP/s / 3: each product is three bf16 MFMAs "
                                     "(hi/lo-split operands, fp32 accumulate)")
                roof["mfma_issue_frac_of_2.5PF"] = round(3 * ach / 2500.0, 4)
        del net
        torch.cuda.empty_cache()
        return value, dt, roof

    value, dt, roof = measure(args.dtype)
    # the narrower bf16-storage mode (what the reference computes under precision_amp=bfloat16), reported beside the headline
    secondary = None
    if args.dtype != "bf16" and not args.no_secondary:
        v2, dt2, roof2 = measure("bf16")
        secondary = {"bf16": {"value": round(v2, 2), "unit": "img/s", "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                              "note": "bf16 MFMA operands AND bf16 activation storage: narrower than the reference's fp32 "
                                      "config, not the headline", "roofline": roof2}}
    # the rows next to the headline path, one GPU only: DiT-XL/2 forward and the causal video DiT's network call,
    # both at the precision their reference configs run (bf16), random-init weights of the named architectures
    if world == 1 and not args.no_secondary:
        try:
            secondary = dict(secondary or {})
            secondary["transformer_rows"] = transformer_rows(dev)
        except Exception as e:  # (never lose the headline line to a secondary measurement)
            secondary["transformer_rows"] = {"error": f"{type(e).__name__}: {e}"}
    if world > 1:
        dist.barrier()

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        line = result_line(value, world, args.steps, args.warmup, dt, args.dtype, B, args.sample_steps, not args.no_graph, roof, cpu)
        if secondary:
            line["secondary"] = secondary
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
