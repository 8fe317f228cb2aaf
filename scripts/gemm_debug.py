"""Error map of fg_op_gemm_bf16 against torch on small shapes (debug aid)."""
import ctypes, sys
import torch
from fastgen_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (m, n, k, order) in [(256, 256, 128, 32), (512, 512, 128, 32), (256, 256, 256, 32), (512, 256, 1152, 32), (1000, 1152, 1152, 32)]:
    g = torch.Generator().manual_seed(1)
    a = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().cuda()
    bias = torch.randn(n, generator=g).cuda()
    out = torch.full((m, n), 777.0, dtype=torch.bfloat16, device="cuda")
    _lib.check(L.fg_op_gemm_bf16(p(a), p(w), p(bias), p(out), m, n, k, 0, None, 0, 1, None, order, st))
    torch.cuda.synchronize()
    want = a.float() @ w.float().t() + bias
    err = (out.float() - want).abs()
    bad = err > 0.05
    print(f"m={m} n={n} k={k}: max err {err.max().item():.3f}, bad {bad.float().mean().item():.4f}, untouched {(out.float() == 777.0).float().mean().item():.4f}")
    if bad.any():
        # 16x16-block map of the first 256x256 tile
        blk = bad[:256, :256].float().reshape(16, 16, 16, 16).mean(dim=(1, 3))
        for r in range(16):
            print("  " + " ".join(f"{v:.1f}" for v in blk[r].tolist()))
        r, c = torch.nonzero(bad)[0].tolist()
        print("  first bad", r, c, out[r, c].item(), want[r, c].item())
        # is the value some other element of the same row?
        row = want[r]
        j = (row - out[r, c].float()).abs().argmin().item()
        print("  closest in row: col", j, row[j].item())
    if bad.any():
        sub = bad[:32, :64].int()
        for r in range(0, 32, 4):
            print("  row", r, "".join(str(v) for v in sub[r].tolist()))
