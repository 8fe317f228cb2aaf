"""Ablation timing of the fused conv kernel (debug hook fg_debug_conv_bench; not part of the product path)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastgen_amd", "libfastgen_amd.so"))
L.fg_debug_conv_bench.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
L.fg_last_error.restype = ctypes.c_char_p
def run(dtype, B, cin, res, ks, resid, dbg, iters=10):
    ms = ctypes.c_float()
    rc = L.fg_debug_conv_bench(dtype, B, cin, res, ks, resid, dbg, iters, ctypes.byref(ms))
    assert rc == 0, L.fg_last_error()
    return ms.value
names = {0: "full", 1: "no-staging", 2: "no-B-refill", 4: "no-epilogue", 8: "no-MFMA/LDS-read", 3: "no-stage,no-B", 7: "no-stage,no-B,no-epi",
         15: "nothing", 12: "no-epi,no-MFMA", 6: "no-B,no-epi", 5: "no-stage,no-epi", 14: "only-staging", 13: "only-B-refill", 11: "only-epilogue"}
for dtype, B, cin, res, resid in ((1, 512, 256, 32, 1), (1, 512, 256, 32, 0), (1, 512, 256, 16, 1), (0, 512, 256, 32, 1)):
    gf = 2.0 * B * res * res * 256 * 9 * cin / 1e9
    print(f"--- dtype={'bf16' if dtype else 'fp32'} B={B} Cin={cin} {res}x{res} resid={resid}  ({gf:.0f} GFLOP)")
    for dbg in (0, 1, 2, 4, 8, 3, 6, 5, 7, 12, 14, 13, 11, 15):
        ms = run(dtype, B, cin, res, 3, resid, dbg)
        print(f"  dbg={dbg:2d} {names[dbg]:24s} {ms*1e3:9.1f} us   {gf/ms:8.1f} TFLOP/s-equivalent", flush=True)
