"""Time `fg_op_gemm_bf16` (fastgen_amd/csrc/gemm.hip) on the DiT-XL/2 block shapes (hidden 1152, MLP 4608) and the 1.3B video DiT's
(hidden 1536, MLP 8960), random bf16 operands, against the 2.5 PFLOP/s dense bf16 roof.
    python scripts/gemm_bench.py [tokens] [--order=N ...]
--order is fg_op_gemm_bf16's tile_order: low bits = tile order (1 XCD-aware), + 16 forces the register-staged gemm_bf16_kernel, + 32 the
8-wave ping-pong gemm_bf16_pp_kernel, + 256 the narrow-tile gemm_bf16_pp2_kernel (default: 17 and 33, i.e. register-staged, then ping-pong).
FA_LIB=<path> loads another build of the library (A / B runs)."""
import ctypes
import sys

import torch

from fastgen_amd import _lib

M = next((int(a) for a in sys.argv[1:] if not a.startswith("--")), 65536)
ORDERS = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--order=")] or [16 + 1, 32 + 1]
import os
if os.environ.get("FA_LIB"):  # an experimental build of the library (A / B runs)
    _lib.LIB_PATH = os.path.abspath(os.environ["FA_LIB"])
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_w = torch.randn(8192, 8192, device="cuda").bfloat16()
for _ in range(50):  # bring the chip to its loaded clock / power state before the first timed shape
    _w @ _w
torch.cuda.synchronize()
del _w
for name, n, k, act in [("qkv-like 1152->3456", 3456, 1152, 0), ("proj 1152->1152", 1152, 1152, 0), ("fc1 1152->4608 gelu", 4608, 1152, 1),
                        ("fc2 4608->1152", 1152, 4608, 0), ("wan 1536->1536", 1536, 1536, 0), ("wan ffn 1536->8960 gelu", 8960, 1536, 1),
                        ("wan ffn 8960->1536", 1536, 8960, 0)]:
    a = torch.randn(M, k, device="cuda").bfloat16()
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda")
    out = torch.empty(M, n, dtype=torch.bfloat16, device="cuda")
    for order in ORDERS:
        run = lambda: _lib.check(L.fg_op_gemm_bf16(p(a), p(w), p(bias), p(out), M, n, k, act, None, 0, 1, None, order, st))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        tf = 2.0 * M * n * k / us / 1e6
        print(f"{name:26s} M={M} order={order}: {us:9.1f} us  {tf:7.1f} TFLOP/s  {tf / 25:5.1f} % of 2.5 PF", flush=True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    af, wf = a, w
    for _ in range(2):
        torch.matmul(af, wf.t())
    t0.record()
    for _ in range(5):
        torch.matmul(af, wf.t())
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 200
    print(f"{'  (torch.matmul = hipBLASLt)':26s} {'':20s} {us:9.1f} us  {2.0 * M * n * k / us / 1e6:7.1f} TFLOP/s", flush=True)
