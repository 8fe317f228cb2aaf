"""Time fg_op_conv_wgrad (and, when present, the other training-step ops) on the U-Net's layer shapes.  Run on an MI355X."""
import sys

import torch

from fastgen_amd import _lib

L = _lib.lib()
B = int(next((a for a in sys.argv[1:] if a.isdigit()), 64))
X3 = "--x3" in sys.argv  # fp32 tensors, split-bf16 products (the reference's training precision): fg_op_conv_wgrad_f32
for res, cin, cout, ks in ((32, 256, 256, 3), (32, 512, 256, 3), (32, 384, 256, 3), (16, 256, 256, 3), (16, 512, 256, 3),
                           (8, 256, 256, 3), (8, 512, 256, 3), (32, 512, 256, 1), (16, 256, 768, 1)):
    a = torch.randn(B, res, res, cin, device="cuda").to(torch.float32 if X3 else torch.bfloat16)
    d = torch.randn(B, res, res, cout, device="cuda").to(torch.float32 if X3 else torch.bfloat16)
    dw = torch.zeros(cout, cin, ks, ks, device="cuda")
    nbytes = L.fg_op_conv_wgrad_workspace_bytes(B, res, cin, cout, ks)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check((L.fg_op_conv_wgrad_f32 if X3 else L.fg_op_conv_wgrad)(a.data_ptr(), d.data_ptr(), dw.data_ptr(), B, res, cin, cout, ks, 0, ws.data_ptr(), nbytes, s))

    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * B * res * res * cin * cout * ks * ks
    print(f"wgrad{' x3' if X3 else ''} B={B} {res}x{res} {cin}->{cout} k{ks}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  (workspace {nbytes / 2**20:.0f} MiB)")
