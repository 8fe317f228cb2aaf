"""Timing of the 3x3 conv at 8x8 / 16x16 / 32x32 through the production dispatch (debug hook)."""
import ctypes, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(root, "fastgen_amd", "libfastgen_amd.so"))
L.fg_debug_conv_bench.argtypes = [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
for rep in range(2):
    for res in (8, 16, 32):
        for resid in (0, 1):
            ms = ctypes.c_float()
            assert L.fg_debug_conv_bench(1, 512, 256, res, 3, resid, -1, 20, ctypes.byref(ms)) == 0
            gf = 2.0 * 512 * res * res * 256 * 9 * 256 / 1e9
            print(f"res={res:2d} resid={resid}: {ms.value*1e3:7.1f} us  {gf/ms.value:7.1f} TFLOP/s", flush=True)
