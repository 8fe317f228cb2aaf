"""Cost of the sample-writer step after generator_fn (scripts/fid/compute_fid_from_ckpts.py:199) at the bench batch:
torch's elementwise chain + fp32/uint8 device-to-host copy versus fg_op_images_to_u8 + uint8 copy.  Run on an MI355X."""
import time

import torch

from fastgen_amd.utils.images import images_to_uint8

B = 512
x = torch.randn(B, 3, 32, 32, device="cuda") * 0.6


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print(f"B={B}: torch chain (mul, add, clip, to uint8, permute) + D2H : {timed(lambda: (x * 127.5 + 128).clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1).cpu()):8.1f} us")
print(f"B={B}: fg_op_images_to_u8 + D2H                               : {timed(lambda: images_to_uint8(x).cpu()):8.1f} us")
print(f"B={B}: fp32 samples D2H only (what a host-side conversion needs): {timed(lambda: x.cpu()):8.1f} us")
