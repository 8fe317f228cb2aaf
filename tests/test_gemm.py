"""Token GEMM of the transformer blocks (fastgen_amd/csrc/gemm.hip, `fg_op_gemm_bf16`): the `nn.Linear` calls of the reference's
DiTBlock (fastgen/networks/DiT/network.py:168-198) as they run under bf16 autocast - bf16 operands, fp32 accumulation - with the fused
epilogues (bias, tanh-GELU, adaLN gate x value + residual).  Checked against torch fp32 matmul on the SAME bf16 operands: the only
differences are the accumulation order and the final bf16 rounding, so the tolerance is one bf16 ulp of the result (2^-8 relative
to the row scale) for every shape, tile order, ragged token count and epilogue."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(a, w, bias=None, act=0, gate=None, gate_rows=1, resid=None, order=1):
    from fastgen_amd import _lib

    m, k = a.shape
    n = w.shape[0]
    out = torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(_lib.lib().fg_op_gemm_bf16(p(a), p(w), p(bias), p(out), m, n, k, act, p(gate), gate.shape[1] if gate is not None else 0,
                                          gate_rows, p(resid), order, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out


def _ref(a, w, bias=None, act=0, gate=None, gate_rows=1, resid=None):
    v = a.float() @ w.float().t()
    if bias is not None:
        v = v + bias
    if act == 1:
        v = torch.nn.functional.gelu(v, approximate="tanh")
    if gate is not None:
        v = v * gate.repeat_interleave(gate_rows, dim=0)[: v.shape[0]]
    if resid is not None:
        v = v + resid.float()
    return v


_SHAPES = [(256, 256, 64), (512, 192, 128), (1000, 1152, 1152), (4680, 1536, 1536), (77, 3456, 1152), (2048, 4608, 1152), (2048, 1152, 4608),
           (300, 64, 256), (513, 384, 384)]
# every shape in the launcher's own choice of kernel and tile order (1) and forced onto each kernel (16 + 1 register-staged, 32 + 1 LDS-DMA
# ping-pong); the linear tile orders (0, 32 + 0) on the shapes with the most tiles
_CASES = [(s, o) for s in _SHAPES for o in (1, 16 + 1, 32 + 1)] + [(s, o) for s in _SHAPES[2:7:2] for o in (0, 32 + 0)]


@pytest.mark.parametrize("shape,order", _CASES)
def test_plain_gemm_matches_fp32_matmul_of_the_same_operands(shape, order):
    m, n, k = shape
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    a = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().cuda()
    bias = torch.randn(n, generator=g).cuda()
    got = _run(a, w, bias, order=order).float()
    want = _ref(a, w, bias)
    err = (got - want).abs().max().item()
    assert err <= 2 ** -8 * want.abs().max().item() + 1e-6, (m, n, k, order, err)


@pytest.mark.parametrize("order", [0, 1, 2, 16 + 1, 32 + 2])
def test_epilogues(order):
    m, n, k, rows = 1300, 1152, 1152, 256
    g = torch.Generator().manual_seed(5)
    a = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().cuda()
    bias = torch.randn(n, generator=g).cuda()
    gate = torch.randn((m + rows - 1) // rows, 6 * n, generator=g).cuda()[:, 2 * n: 3 * n]  # a chunk of the adaLN vector: strided rows
    resid = torch.randn(m, n, generator=g).bfloat16().cuda()
    # GELU(tanh)
    got = _run(a, w, bias, act=1, order=order).float()
    want = _ref(a, w, bias, act=1)
    assert (got - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3
    # gate x value + residual; the gate rows are `rows` tokens each and 6 n floats apart
    from fastgen_amd import _lib

    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().fg_op_gemm_bf16(p(a), p(w), p(bias), p(out), m, n, k, 0, p(gate), 6 * n, rows, p(resid), order,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    want = _ref(a, w, bias, gate=gate.contiguous(), gate_rows=rows, resid=resid)
    assert (out.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3


@pytest.mark.parametrize("m,n,k", [(4680, 1536, 1536), (4680, 1536, 8960), (1000, 512, 2048), (256, 256, 1024)])
def test_split_k_on_short_grids(m, n, k):
    """The video DiT's one-sample chunk: 4 680 tokens x 1 536 outputs = 114 tiles for 256 CUs; with scratch the launcher cuts K."""
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().cuda()
    bias = torch.randn(n, generator=g).cuda()
    rows = 1560
    gate = torch.randn((m + rows - 1) // rows, n, generator=g).cuda()
    resid = torch.randn(m, n, generator=g).bfloat16().cuda()
    want = _ref(a, w, bias, gate=gate, gate_rows=rows, resid=resid)
    got = _run(a, w, bias, gate=gate, gate_rows=rows, resid=resid, order=64 + 32 + 1)
    assert (got.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3
    want = _ref(a, w, bias, act=1)
    got = _run(a, w, bias, act=1, order=64 + 1)
    assert (got.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3


@pytest.mark.parametrize("m,n,k", [(4680, 1536, 1536), (4680, 4608, 1536), (4680, 1536, 8960), (300, 128, 128), (1000, 1152, 1152), (513, 400, 256)])
def test_narrow_tile_kernel(m, n, k):
    """tile_order bit 256: the 256 x 128 form of the ping-pong kernel (what the launcher gives short token counts - one sample of the
    video DiT is 4 680 tokens - instead of split-K or two-thirds-empty rounds of 256-wide tiles): every epilogue, ragged token and
    output counts, gate periods that straddle a tile, in place."""
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().cuda()
    bias = torch.randn(n, generator=g).cuda()
    rows = 1560 if m > 2000 else 256
    gate = torch.randn((m + rows - 1) // rows, n, generator=g).cuda()
    resid = torch.randn(m, n, generator=g).bfloat16().cuda()
    want = _ref(a, w, bias, gate=gate, gate_rows=rows, resid=resid)
    got = _run(a, w, bias, gate=gate, gate_rows=rows, resid=resid, order=256 + 1)
    assert torch.isfinite(got.float()).all()
    assert (got.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3
    assert torch.equal(got, _run(a, w, bias, gate=gate, gate_rows=rows, resid=resid, order=256 + 1))
    want = _ref(a, w, bias, act=1)
    got = _run(a, w, bias, act=1, order=256 + 0)
    assert (got.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3
    got = _run(a, w, order=256 + 1)
    want = _ref(a, w)
    assert (got.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-6
    # in place: out aliases the residual
    from fastgen_amd import _lib

    want = _ref(a, w, bias, gate=gate, gate_rows=rows, resid=resid)
    x = resid.clone()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().fg_op_gemm_bf16(p(a), p(w), p(bias), p(x), m, n, k, 0, p(gate), n, rows, p(x), 256 + 1,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert (x.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3


@pytest.mark.parametrize("m,n,k", [(4680, 1536, 1536), (1024, 1152, 4608), (300, 256, 128), (1000, 1152, 1152), (513, 400, 256)])
def test_one_wave_per_simd_kernel(m, n, k):
    """tile_order bit 512: gemm_bf16_w4_kernel (4 waves x 128 x 128 outputs, accumulators in the accumulator file, one barrier per K-step):
    every token epilogue, ragged token and output counts (shifted last tiles), gate periods that straddle a tile, in place."""
    g = torch.Generator().manual_seed(m + n + k + 1)
    a = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().cuda()
    bias = torch.randn(n, generator=g).cuda()
    rows = 1560 if m > 2000 else 256
    gate = torch.randn((m + rows - 1) // rows, n, generator=g).cuda()
    resid = torch.randn(m, n, generator=g).bfloat16().cuda()
    want = _ref(a, w, bias, gate=gate, gate_rows=rows, resid=resid)
    got = _run(a, w, bias, gate=gate, gate_rows=rows, resid=resid, order=512 + 1)
    assert torch.isfinite(got.float()).all()
    assert (got.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3
    assert torch.equal(got, _run(a, w, bias, gate=gate, gate_rows=rows, resid=resid, order=512 + 1))
    want = _ref(a, w, bias, act=1)
    got = _run(a, w, bias, act=1, order=512 + 0)
    assert (got.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3
    got = _run(a, w, order=512 + 1)
    want = _ref(a, w)
    assert (got.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-6
    from fastgen_amd import _lib

    want = _ref(a, w, bias, gate=gate, gate_rows=rows, resid=resid)
    x = resid.clone()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().fg_op_gemm_bf16(p(a), p(w), p(bias), p(x), m, n, k, 0, p(gate), n, rows, p(x), 512 + 1,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert (x.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3


def test_unsupported_shapes_are_refused():
    from fastgen_amd import _lib

    a = torch.zeros(64, 96, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(64, 96, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(_lib.FastGenAMDError):
        _run(a, w)  # k % 64 != 0
    # the timing experiments (no stores / no epilogue / cycle stamps: act bits 4 / 8 / 16, tile_order bit 128) are not in the product
    # library: anything but act 0 / 1 is refused instead of silently computing garbage
    a = torch.zeros(256, 128, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(256, 128, dtype=torch.bfloat16, device="cuda")
    for act in (2, 4, 8, 16, 5, -1):
        with pytest.raises(_lib.FastGenAMDError, match="act must be"):
            _run(a, w, act=act)
    for order in (128, 128 + 33, 16 + 32, 1024, 256 + 32, 512 + 32, 512 + 256):
        with pytest.raises(_lib.FastGenAMDError, match="tile_order"):
            _run(a, w, order=order)
    assert torch.equal(_run(a, w, act=1), torch.zeros(256, 256, dtype=torch.bfloat16, device="cuda"))


@pytest.mark.parametrize("m,n,k,rows", [(4680, 3072, 1536, 1560), (2900, 1152, 1152, 300), (1024, 512, 256, 256)])
def test_gate_periods_that_straddle_a_tile(m, n, k, rows):
    """The ping-pong kernel's token epilogue reads the gate rows of a tile's first token and of the next gate period: periods that are
    no multiple of the 256-token tile (the video DiT's 1 560 tokens per frame), a ragged last token tile, in-place residual."""
    g = torch.Generator().manual_seed(m + rows)
    a = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().cuda()
    bias = torch.randn(n, generator=g).cuda()
    gate = torch.randn((m + rows - 1) // rows, n, generator=g).cuda()
    resid = torch.randn(m, n, generator=g).bfloat16().cuda()
    want = _ref(a, w, bias, gate=gate, gate_rows=rows, resid=resid)
    got = _run(a, w, bias, gate=gate, gate_rows=rows, resid=resid, order=32 + 1)
    assert (got.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3
    # in place: out aliases the residual (rows / columns a shifted edge tile shares with its neighbour are stored once)
    from fastgen_amd import _lib

    x = resid.clone()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().fg_op_gemm_bf16(p(a), p(w), p(bias), p(x), m, n, k, 0, p(gate), n, rows, p(x), 32 + 1,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert (x.float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-3
