"""Cycle stamps of workgroup 0 of the ping-pong GEMM (fg_op_gemm_bf16 tile_order bit 128: printed by the library per tile and wave group)."""
import ctypes, sys
import torch
from fastgen_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libfastgen_amd.so", "libfastgen_amd_timing.so")  # `make -C fastgen_amd/csrc timing` (act bits 4 / 8 / 16 exist only there)
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_w = torch.randn(8192, 8192, device="cuda").bfloat16()
for _ in range(30):
    _w @ _w
torch.cuda.synchronize()
M = 65536
for name, n, k, act in [("qkv-like", 3456, 1152, 0), ("fc1 gelu", 4608, 1152, 1), ("fc2", 1152, 4608, 0)]:
    a = torch.randn(M, k, device="cuda").bfloat16()
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda")
    out = torch.empty(M, n, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        _lib.check(L.fg_op_gemm_bf16(p(a), p(w), p(bias), p(out), M, n, k, act, None, 0, 1, None, 33, st))
    torch.cuda.synchronize()
    print(name, file=sys.stderr, flush=True)
    _lib.check(L.fg_op_gemm_bf16(p(a), p(w), p(bias), p(out), M, n, k, act, None, 0, 1, None, 33 + 128, st))
    torch.cuda.synchronize()
