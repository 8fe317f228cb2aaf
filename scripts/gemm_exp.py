"""Timing experiments on the ping-pong GEMM (act bit 4: no stores, bit 8: no epilogue)."""
import ctypes, sys
import torch
from fastgen_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libfastgen_amd.so", "libfastgen_amd_timing.so")  # `make -C fastgen_amd/csrc timing` (act bits 4 / 8 / 16 exist only there)
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_w = torch.randn(8192, 8192, device="cuda").bfloat16()
for _ in range(50):  # bring the chip to its loaded clock / power state before the first timed shape
    _w @ _w
torch.cuda.synchronize()
del _w
M = 65536
for name, n, k in [("qkv-like", 3456, 1152), ("n=3584 (14 full tiles)", 3584, 1152), ("k=2304", 3584, 2304), ("k=4608 n=1024", 1024, 4608), ("k=576", 3584, 576)]:
    a = torch.randn(M, k, device="cuda").bfloat16()
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda")
    out = torch.empty(M, n, dtype=torch.bfloat16, device="cuda")
    for act in (0, 4, 8):
        for order in (32 + 1,):
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
            print(f"{name:24s} act={act} order={order}: {us:8.1f} us {2.0 * M * n * k / us / 1e6:7.1f} TF/s", flush=True)
