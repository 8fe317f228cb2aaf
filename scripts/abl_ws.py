"""Ablation of the wave-specialised conv (conv_ws.hip) on the dominant shape.  Debug hook only; not product code."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(root, "fastgen_amd", "libfastgen_amd.so"))
L.fg_debug_conv_bench.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
def run(dtype, B, cin, res, ks, resid, dbg, iters=10):
    ms = ctypes.c_float()
    assert L.fg_debug_conv_bench(dtype, B, cin, res, ks, resid, dbg, iters, ctypes.byref(ms)) == 0
    return ms.value
names = {-1: "dispatch path", 0: "old kernel (full)", 64: "ws full", 65: "ws no staging", 66: "ws no retire", 67: "ws no staging, no retire",
         71: "ws consumers + barriers only", 79: "ws MFMA + A reads only", 80: "ws producers only (no MFMA)",
         88: "ws producers only, no refill", 72: "ws no weight refill", 96: "ws retire: no store", 128: "ws retire: no LDS read", 160: "ws retire: VALU only"}
for cin in (256, 512):
    gf = 2.0 * 512 * 32 * 32 * 256 * 9 * cin / 1e9
    for resid in (1, 0):
        print(f"Cin={cin} resid={resid} ({gf:.0f} GFLOP)")
        for dbg in (-1, 0, 64, 65, 66, 67, 72, 80, 96, 128, 160):
            ms = run(1, 512, cin, 32, 3, resid, dbg)
            print(f"  {names[dbg]:32s} {ms*1e3:8.1f} us  {gf/ms:7.1f} TFLOP/s", flush=True)
