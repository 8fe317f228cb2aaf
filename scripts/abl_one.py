"""Run one ablation configuration of the conv kernels repeatedly (for rocprofv3 counter passes).  Debug hook only.
usage: python scripts/abl_one.py <dbg> [cin] [resid] [iters]"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(root, "fastgen_amd", "libfastgen_amd.so"))
L.fg_debug_conv_bench.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
dbg = int(sys.argv[1]); cin = int(sys.argv[2]) if len(sys.argv) > 2 else 256
resid = int(sys.argv[3]) if len(sys.argv) > 3 else 0; iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ms = ctypes.c_float()
assert L.fg_debug_conv_bench(1, 512, cin, 32, 3, resid, dbg, iters, ctypes.byref(ms)) == 0
print(f"dbg={dbg} cin={cin} resid={resid}: {ms.value*1e3:.1f} us")
