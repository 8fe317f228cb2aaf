import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(root, "fastgen_amd", "libfastgen_amd.so"))
L.fg_debug_conv_bench.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
def run(dtype, B, cin, res, ks, resid, dbg, iters=10):
    ms = ctypes.c_float()
    assert L.fg_debug_conv_bench(dtype, B, cin, res, ks, resid, dbg, iters, ctypes.byref(ms)) == 0
    return ms.value
for res in (32, 16):
  gf = 2.0 * 512 * res * res * 256 * 9 * 256 / 1e9
  for resid in (1, 0):
    print("res", res, "resid", resid)
    for name, dbg in [("PRODUCTION", -1), ("dbg-build full", 0), ("no-stage", 1), ("no-B", 2), ("no-epi", 4), ("no-stage,no-B", 3), ("no-stage,no-epi", 5), ("no-B,no-epi", 6), ("core only", 7)]:
        ms = run(1, 512, 256, res, 3, resid, dbg)
        print(f"  {name:28s} {ms*1e3:8.1f} us  {gf/ms:7.1f} TF", flush=True)
