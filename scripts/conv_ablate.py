"""Compile-time ablation of the dominant fused conv (bf16 3x3 at 32x32): each variant is the production kernel with one
feature compiled out (mask: 1 staging, 2 weight refill, 4 epilogue).  Debug hook fg_debug_conv_bench; not product code."""
import ctypes, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(root, "fastgen_amd", "libfastgen_amd.so"))
L.fg_debug_conv_bench.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
def run(dtype, B, cin, res, ks, resid, dbg, iters=10):
    ms = ctypes.c_float()
    assert L.fg_debug_conv_bench(dtype, B, cin, res, ks, resid, dbg, iters, ctypes.byref(ms)) == 0
    return ms.value
names = {-1: "production (dispatch path)", 0: "full", 1: "no staging", 2: "no weight refill", 3: "no staging, no refill", 4: "no epilogue",
         5: "no staging, no epilogue", 6: "no refill, no epilogue", 7: "core: LDS reads + MFMA only"}
for cin in (256, 512):
    gf = 2.0 * 512 * 32 * 32 * 256 * 9 * cin / 1e9
    for resid in (1, 0):
        print(f"Cin={cin} resid={resid} ({gf:.0f} GFLOP)")
        for dbg in (-1, 0, 1, 2, 3, 4, 5, 6, 7):
            ms = run(1, 512, cin, 32, 3, resid, dbg)
            print(f"  {names[dbg]:32s} {ms*1e3:8.1f} us  {gf/ms:7.1f} TFLOP/s", flush=True)
