import ctypes, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(root, "fastgen_amd", "libfastgen_amd.so"))
L.fg_debug_conv_bench.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
def run(dtype, B, cin, res, ks, resid, dbg, iters=10):
    ms = ctypes.c_float()
    assert L.fg_debug_conv_bench(dtype, B, cin, res, ks, resid, dbg, iters, ctypes.byref(ms)) == 0
    return ms.value
gf = 2.0 * 512 * 32 * 32 * 256 * 9 * 256 / 1e9
for resid in (1, 0):
    for name, dbg in [("PRODUCTION", -1), ("dbg full", 0)] + [(f"stagger {d}K", d << 8) for d in (4, 8, 16, 24, 32, 48)]:
        ms = run(1, 512, 256, 32, 3, resid, dbg)
        print(f"resid={resid} {name:16s} {ms*1e3:8.1f} us  {gf/ms:7.1f} TF", flush=True)
