import ctypes, os
root = "/root/repo"
L = ctypes.CDLL(os.path.join(root, "fastgen_amd", "libfastgen_amd.so"))
L.fg_debug_conv_bench.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
def run(dtype, B, cin, res, ks, resid, dbg, iters=10):
    ms = ctypes.c_float()
    assert L.fg_debug_conv_bench(dtype, B, cin, res, ks, resid, dbg, iters, ctypes.byref(ms)) == 0
    return ms.value
gf = 2.0 * 512 * 32 * 32 * 256 * 9 * 256 / 1e9
for dbg in (1, 5, 7, 16+1, 16+5, 16+7, 16+4):
    ms = run(1, 512, 256, 32, 3, 0, dbg)
    print(f"dbg={dbg:2d} {ms*1e3:8.1f} us {gf/ms:7.1f} TF", flush=True)
