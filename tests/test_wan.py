"""Causal video DiT (SURVEY 8(f)3 + the CausVid row of 8(a)).  PARITY UNPINNED: the network's arithmetic lives in un-vendored
diffusers (oracle/wan_ref.py's header); these tests hold the HIP path (fg_wan_*, through the drop-in module) to the restatement.

CPU: properties of the restatement itself (RoPE frame offset, cache-append attention == attention over the concatenated frames),
the module's state dict against the restated key list, loud failure without a GPU.
GPU (-m gpu): attention kernel vs torch SDPA; network calls of the autoregressive sampler (chunk 0 with store_kv, chunk 1 over the
cache) and the whole CausVid student loop against the oracle.  Tolerance: the HIP path runs bf16 operands (the reference's
precision for this network) against the fp32 oracle: relative L2 <= 2e-2 per call, <= 3e-2 after the multi-chunk loop."""
import ctypes

import pytest
import torch

from oracle import wan_ref as R

KW = dict(num_attention_heads=2, attention_head_dim=128, text_dim=128, ffn_dim=512, num_layers=2, chunk_size=2, total_num_frames=6)


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def test_rope_offset_is_a_slice_of_the_full_table():
    cfg = R.TINY
    c_all, s_all = R.rope_for_chunk(cfg, 6, 3, 4, 0)
    c, s = R.rope_for_chunk(cfg, 2, 3, 4, 2)
    assert torch.equal(c, c_all[2 * 12: 4 * 12]) and torch.equal(s, s_all[2 * 12: 4 * 12])
    # axis split of head dim 128: 44 | 42 | 42, every frequency twice
    cos, _, dims = R.rope_tables(cfg)
    assert dims == (44, 42, 42) and torch.equal(cos[0][:, 0::2], cos[0][:, 1::2])


def test_oracle_cache_append_equals_attention_over_all_frames():
    """Chunk 1 computed over the cache that chunk 0's store_kv call left == chunk 1's rows of one call over both chunks in which
    chunk 0's queries are irrelevant: run the two-chunk call with the same per-frame inputs and compare the later frames' attention
    inputs through the first block's output."""
    cfg = R.TINY
    net = R.CausalWanRef(R.random_state_dict(cfg, 3), cfg)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 16, 4, 4, 6, generator=g)
    text = torch.randn(1, 8, cfg.text_dim, generator=g)
    t = torch.tensor([0.0])
    net.forward(x[:, :, :2], t, text, 0, store_kv=True)
    tr1 = {}
    net.forward(x[:, :, 2:], t, text, 2, store_kv=False, trace=tr1)
    net2 = R.CausalWanRef(R.random_state_dict(cfg, 3), cfg)
    tr2 = {}
    net2.forward(x, t, text, 0, store_kv=False, trace=tr2)  # all four frames at once: every query sees all 4 frames (no mask)
    # block 0's self-attention of frames 2-3 differs (they now also see each other AND frames 0-1 exactly as before): same keys, so
    # the first block's output rows of frames 2-3 agree
    L = 2 * 2 * 3
    assert torch.allclose(tr1["block0"][:, :L], tr2["block0"][:, L:], atol=2e-5)


def test_module_state_dict_matches_the_restated_key_list():
    from fastgen_amd.networks.Wan.network_causal import CausalWan

    net = CausalWan(**KW)
    want = R.state_dict_shapes(R.TINY)
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want and list(got) == list(want)
    res = net.load_state_dict(R.random_state_dict(R.TINY, 5), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    with pytest.raises(RuntimeError, match="HIP GPU only"):
        with torch.no_grad():
            net(torch.zeros(1, 16, 2, 4, 6), torch.tensor([0.5]), condition=torch.zeros(1, 8, 128), is_ar=True)
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 16, 2, 4, 6), torch.tensor([0.5]), condition=torch.zeros(1, 8, 128), is_ar=True)  # grad mode


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,hd,Lq,Lkv", [(1, 2, 128, 128, 128), (2, 3, 128, 200, 333), (1, 12, 128, 96, 1000), (1, 2, 128, 48, 17),
                                            (1, 1, 128, 4680, 4680), (2, 3, 128, 200, 1100), (3, 16, 72, 256, 256), (1, 2, 72, 100, 77),
                                            (2, 4, 72, 300, 250), (2, 4, 72, 300, 1000)])
def test_attention_matches_sdpa(B, H, hd, Lq, Lkv):
    """head dim 128: the video DiT (one sample takes the key-split path; >= 1024 keys: fa2_kernel with its uneven two-way split, here also
    for two samples, a ragged query tile and a ragged key tile); 72: DiT-XL/2 (half-empty contraction step, output tile computed from
    row spill-over; <= 256 keys: the whole-sequence kernel, also with masked keys and three query tiles)."""
    from fastgen_amd import _lib

    g = torch.Generator().manual_seed(Lq + Lkv)
    q = torch.randn(B, Lq, H * hd, generator=g).bfloat16().cuda()
    k = torch.randn(B, Lkv, H * hd, generator=g).bfloat16().cuda()
    v = torch.randn(B, Lkv, H * hd, generator=g).bfloat16().cuda()
    out = torch.full_like(q, float("nan"))
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().fg_op_attention(p(q), p(k), p(v), p(out), B, H, hd, Lq, Lkv, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    want = R.sdpa(q.float().view(B, Lq, H, hd), k.float().view(B, Lkv, H, hd), v.float().view(B, Lkv, H, hd))
    assert torch.isfinite(out.float()).all() and _rel(out, want) < 1e-2, _rel(out, want)


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["rising", "falling", "spikes"])
def test_attention_when_the_scores_outgrow_the_running_reference(order):
    """fa2_kernel (head dim 128, >= 1024 keys) exponentiates against a reference maximum that only moves when a tile's scores leave a
    window of 2^40 above it, rescaling O and the row sum in a rarely taken block outside its tile loop.  Random keys of one scale never
    take that block after the first tile; here the key norms grow (fall, spike) along the sequence so that it is taken again and again,
    over a ragged last tile and over key splits."""
    from fastgen_amd import _lib

    B, H, hd, Lq, Lkv = 1, 2, 128, 300, 2000 + 13
    g = torch.Generator().manual_seed(3)
    q = (4.0 * torch.randn(B, Lq, H * hd, generator=g)).bfloat16().cuda()
    k = torch.randn(B, Lkv, H * hd, generator=g)
    ramp = torch.linspace(0.25, 8.0, Lkv)
    if order == "falling":
        ramp = ramp.flip(0)
    elif order == "spikes":
        ramp = torch.where(torch.arange(Lkv) % 97 == 5, 8.0, 0.5) * (1.0 + torch.arange(Lkv) / Lkv)
    k = (k * ramp[None, :, None]).bfloat16().cuda()
    v = torch.randn(B, Lkv, H * hd, generator=g).bfloat16().cuda()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    want = R.sdpa(q.float().view(B, Lq, H, hd), k.float().view(B, Lkv, H, hd), v.float().view(B, Lkv, H, hd))
    for ns in (1, 4):
        out = torch.full_like(q, float("nan"))
        _lib.check(_lib.lib().fg_op_attention_split(p(q), p(k), p(v), p(out), B, H, hd, Lq, Lkv, ns,
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all() and _rel(out, want) < 1e-2, (ns, _rel(out, want))


def _nets(seed=7):
    from fastgen_amd.networks.Wan.network_causal import CausalWan

    sd = R.random_state_dict(R.TINY, seed)
    ref = R.CausalWanRef(sd, R.TINY)
    net = CausalWan(**KW)
    net.load_state_dict(sd, strict=True)
    return ref, net.cuda().eval()


@pytest.mark.gpu
def test_autoregressive_calls_against_oracle():
    ref, net = _nets()
    g = torch.Generator().manual_seed(8)
    B, H, W = 2, 16, 24
    x = torch.randn(B, 16, 4, H, W, generator=g)
    text = torch.randn(B, 40, 128, generator=g)
    with torch.inference_mode():
        # chunk 0 at t = 0.8 (no cache yet), then its cache-fill call at t = 0, then chunk 1 over the cache
        for (lo, hi, t, store) in [(0, 2, 0.8, False), (0, 2, 0.0, True), (2, 4, 0.6, False), (2, 4, 0.0, True)]:
            tt = torch.full((B,), t, dtype=torch.float64)
            want = ref.forward(x[:, :, lo:hi], tt, text, cur_start_frame=lo, store_kv=store)
            got = net(x[:, :, lo:hi].cuda(), tt.cuda(), condition=text.cuda(), fwd_pred_type="flow", cur_start_frame=lo, store_kv=store, is_ar=True)
            assert got.shape == want.shape
            assert _rel(got.cpu(), want) < 2e-2, (lo, t, store, _rel(got.cpu(), want))
        # a store_kv = False call on frames whose keys / values are stored would clobber them for later chunks (this build writes the
        # chunk's K / V to its cache rows in every call; the reference leaves its cache alone there): refused, not answered differently
        from fastgen_amd import _lib

        tt = torch.full((B,), 0.6, dtype=torch.float64)
        with pytest.raises(_lib.FastGenAMDError, match="store_kv = 0"):
            net(x[:, :, 2:4].cuda(), tt.cuda(), condition=text.cuda(), fwd_pred_type="x0", cur_start_frame=2, is_ar=True)
        # the x0 conversion of the RF schedule: x0 = x_t - t * flow (frames 4-5: behind everything stored)
        x45 = torch.randn(B, 16, 2, H, W, generator=g)
        got = net(x45.cuda(), tt.cuda(), condition=text.cuda(), fwd_pred_type="x0", cur_start_frame=4, is_ar=True)
        want = x45 - 0.6 * ref.forward(x45, tt, text, cur_start_frame=4)
        assert _rel(got.cpu(), want) < 2e-2
        net.clear_caches()
        # ... and after clear_caches nothing is stored any more
        net(x[:, :, 2:4].cuda(), tt.cuda(), condition=text.cuda(), fwd_pred_type="x0", cur_start_frame=2, is_ar=True)
        net.clear_caches()


@pytest.mark.gpu
def test_width_of_the_1p3b_network_against_oracle():
    """Inner dim 1536 (12 heads), MLP 8960: the widths the 1.3B network instantiates (RMSNorm / LayerNorm / output kernels at 12 x 128,
    whole 256-token GEMM tiles, split-K on the short grids, key-split attention) - two layers, two chunks of two 16 x 16-token frames."""
    from fastgen_amd.networks.Wan.network_causal import CausalWan

    cfg = R.WanConfig(num_heads=12, head_dim=128, text_dim=256, ffn_dim=8960, num_layers=2, chunk_size=2, total_num_frames=4)
    sd = R.random_state_dict(cfg, 21)
    ref = R.CausalWanRef(sd, cfg)
    net = CausalWan(num_attention_heads=12, attention_head_dim=128, text_dim=256, ffn_dim=8960, num_layers=2, chunk_size=2, total_num_frames=4)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(22)
    x = torch.randn(1, 16, 4, 32, 32, generator=g)
    text = torch.randn(1, 64, 256, generator=g)
    with torch.inference_mode():
        for (lo, hi, t, store) in [(0, 2, 0.0, True), (2, 4, 0.5, False)]:
            tt = torch.full((1,), t, dtype=torch.float64)
            want = ref.forward(x[:, :, lo:hi], tt, text, cur_start_frame=lo, store_kv=store)
            got = net(x[:, :, lo:hi].cuda(), tt.cuda(), condition=text.cuda(), fwd_pred_type="flow", cur_start_frame=lo, store_kv=store, is_ar=True)
            assert _rel(got.cpu(), want) < 2e-2, (lo, _rel(got.cpu(), want))


@pytest.mark.gpu
def test_width_of_the_14b_network_against_oracle():
    """Inner dim 5120 (40 heads), MLP 13824 (`Wan2.1-T2V-14B`, fastgen/configs/net.py:180-181): the widest configuration the C ABI lists
    (include/fastgen_amd.h, fg_wan_config) - RMSNorm / LayerNorm / output kernels at 40 x 128, GEMMs with K = 5120 / 13824 - two layers,
    two chunks of two 16 x 16-token frames."""
    from fastgen_amd.networks.Wan.network_causal import CausalWan

    cfg = R.WanConfig(num_heads=40, head_dim=128, text_dim=256, ffn_dim=13824, num_layers=2, chunk_size=2, total_num_frames=4)
    sd = R.random_state_dict(cfg, 31)
    ref = R.CausalWanRef(sd, cfg)
    net = CausalWan(num_attention_heads=40, attention_head_dim=128, text_dim=256, ffn_dim=13824, num_layers=2, chunk_size=2, total_num_frames=4)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(32)
    x = torch.randn(1, 16, 4, 32, 32, generator=g)
    text = torch.randn(1, 64, 256, generator=g)
    with torch.inference_mode():
        for (lo, hi, t, store) in [(0, 2, 0.0, True), (2, 4, 0.5, False)]:
            tt = torch.full((1,), t, dtype=torch.float64)
            want = ref.forward(x[:, :, lo:hi], tt, text, cur_start_frame=lo, store_kv=store)
            got = net(x[:, :, lo:hi].cuda(), tt.cuda(), condition=text.cuda(), fwd_pred_type="flow", cur_start_frame=lo, store_kv=store, is_ar=True)
            assert _rel(got.cpu(), want) < 2e-2, (lo, _rel(got.cpu(), want))


@pytest.mark.gpu
@pytest.mark.parametrize("frames,H,W", [(21, 60, 104), (16, 90, 160)])
def test_full_size_video_shapes_properties(frames, H, W):
    """The 1.3B network (30 layers, 12 heads, MLP 8960, text 512 x 4096) on the real latent shapes - 480p / 21 frames
    (configs/experiments/WanT2V/config_sf.py:19-43: `[16, 21, 60, 104]`, chunks of 3) and BASELINE.json's 720p / 16 frames (`[16, 16, 90, 160]`:
    the remainder frame joins the first chunk) - which no oracle run reaches (32 760 / 57 600 tokens).  Size-independent properties:
    finite, run-to-run identical, and the block-wise causal call over all frames == the autoregressive calls chunk by chunk over the
    cache that the earlier chunks' store_kv calls left (same keys, same RoPE positions, same per-frame timestep) within the bf16
    tolerance.  Only these sizes reach the key-split attention at 7+ chunks of keys, split-K token GEMMs beside full grids, and the
    32-bit buffer-offset guards."""
    from fastgen_amd.networks.Wan.network_causal import CausalWan

    dev = torch.device("cuda:0")
    torch.manual_seed(frames)
    net = CausalWan(total_num_frames=frames).to(dev).eval()  # defaults = Wan2.1-T2V-1.3B, chunk_size 3
    g = torch.Generator(device=dev).manual_seed(100 + frames)
    x = torch.randn(1, 16, frames, H, W, generator=g, device=dev)
    text = torch.randn(1, 512, 4096, generator=g, device=dev)
    t = torch.full((1, frames), 0.4, dtype=torch.float64, device=dev)
    with torch.inference_mode():
        full = net(x, t, condition=text, fwd_pred_type="flow", is_ar=False)
        assert full.shape == x.shape and torch.isfinite(full).all()
        assert float(full.std()) > 1e-3  # (not a degenerate all-equal output)
        assert torch.equal(full, net(x, t, condition=text, fwd_pred_type="flow", is_ar=False))  # fixed-order reductions: bit-identical
        rem = frames % 3
        bounds = [0] + [3 * (i + 1) + rem for i in range(frames // 3)]
        outs = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            o = net(x[:, :, a:b], t[:, 0], condition=text, fwd_pred_type="flow", cur_start_frame=a, store_kv=True, is_ar=True)
            assert torch.isfinite(o).all()
            r = _rel(o, full[:, :, a:b])
            assert r < 2e-2, (a, b, r)
            outs.append(o)
        # the last chunk once more over the complete cache: the same bits (the K / V it rewrites are the ones already there)
        o2 = net(x[:, :, bounds[-2]:], t[:, 0], condition=text, fwd_pred_type="flow", cur_start_frame=bounds[-2], store_kv=True, is_ar=True)
        assert torch.equal(o2, outs[-1])
        net.clear_caches()


@pytest.mark.gpu
@pytest.mark.parametrize("Lq,Lkv", [(4680, 32760), (10800, 57600)])
def test_key_split_attention_at_full_length(Lq, Lkv):
    """The last chunk's self-attention of the 480p / 720p video shapes (one sample, 12 heads x 128): the launcher's key-split choice
    (cost model) against one workgroup per query tile walking all keys, against a forced 8-way split, and against fp32 softmax
    attention of the same bf16 operands."""
    from fastgen_amd import _lib

    B, H, hd = 1, 12, 128
    g = torch.Generator(device="cuda").manual_seed(Lkv)
    q = torch.randn(B, Lq, H * hd, generator=g, device="cuda").bfloat16()
    k = torch.randn(B, Lkv, H * hd, generator=g, device="cuda").bfloat16()
    v = torch.randn(B, Lkv, H * hd, generator=g, device="cuda").bfloat16()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    outs = {}
    for ns in (0, 1, 8):
        out = torch.full_like(q, float("nan"))
        _lib.check(_lib.lib().fg_op_attention_split(p(q), p(k), p(v), p(out), B, H, hd, Lq, Lkv, ns,
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        outs[ns] = out
    want = torch.empty(B, Lq, H * hd, device="cuda")
    for h in range(H):  # fp32, one head at a time (Lq x Lkv scores: 0.6 / 2.5 GB)
        sl = slice(h * hd, (h + 1) * hd)
        sc = torch.softmax(q[0, :, sl].float() @ k[0, :, sl].float().t() * hd ** -0.5, dim=-1)
        want[0, :, sl] = sc @ v[0, :, sl].float()
        del sc
    for ns, out in outs.items():
        assert _rel(out, want) < 1e-2, (ns, _rel(out, want))
    assert _rel(outs[0], outs[1]) < 4e-3 and _rel(outs[8], outs[1]) < 4e-3  # (two bf16 roundings of nearly equal fp32 values)


@pytest.mark.gpu
def test_causvid_student_loop_against_oracle():
    from fastgen_amd.methods.distribution_matching.causvid import CausVidModel

    ref, net = _nets(9)
    g = torch.Generator().manual_seed(10)
    B, H, W = 1, 16, 16
    noise = torch.randn(B, 16, 5, H, W, generator=g)  # 5 frames, chunk_size 2: chunks of 3 + 2 frames (the remainder goes first)
    text = torch.randn(B, 16, 128, generator=g)
    t_list = [0.999, 0.7, 0.3, 0.0]
    got = CausVidModel.generator_fn(net, noise.cuda(), student_sample_steps=3, t_list=t_list, condition=text.cuda(), student_sample_type="ode")
    tl = torch.tensor(t_list, dtype=torch.float64)
    want = R.student_sample_loop(ref, noise * 0.999, tl, text, sample_type="ode")  # latents = noise * sigma(t0) on the RF schedule
    assert got.shape == want.shape
    assert _rel(got.cpu(), want) < 3e-2, _rel(got.cpu(), want)


def _per_call_loop(net, x, tl, text, sample_type, context_noise=0.0, eps=None, exits=None):
    """The chunked student loop as separate network calls through the module's forward and the schedule mirror (what `fg_wan_sampler_run`
    fuses): chunks of chunk_size frames with the remainder in the first one; per chunk the denoising calls down to the exit step, then the
    cache-fill call on the result (re-noised to context_noise if asked)."""
    B, F, cs, sch = x.shape[0], x.shape[2], net.chunk_size, net.noise_scheduler
    n, rem = F // cs, F % cs
    bounds = [(0, rem)] if n == 0 else [(0 if i == 0 else cs * i + rem, cs * (i + 1) + rem) for i in range(n)]
    call = dict(condition=text, fwd_pred_type="x0", is_ar=True)
    net.clear_caches()
    for ci, (a, b) in enumerate(bounds):
        cur = x[:, :, a:b]
        last = len(tl) - 2 if exits is None else exits[ci]
        for i in range(last + 1):
            t = tl[i].expand(B)
            x0 = net(cur, t, cur_start_frame=a, store_kv=False, **call)
            if i < last and tl[i + 1] > 0:
                e = eps[i][:, :, a:b] if sample_type == "sde" else sch.x0_to_eps(xt=cur, x0=x0, t=t)
                x0 = sch.forward_process(x0, e, tl[i + 1].expand(B))
            cur = x0
        x[:, :, a:b] = cur
        tc, xc = tl[-1].expand(B), cur
        if context_noise > 0:
            tc = torch.full((B,), context_noise, device=x.device, dtype=x.dtype)
            xc = sch.forward_process(cur, eps[-1][:, :, a:b], tc)
        net(xc, tc, cur_start_frame=a, store_kv=True, **call)
    net.clear_caches()
    return x


@pytest.mark.gpu
def test_fused_student_loop_equals_the_per_call_loop():
    """`fg_wan_sampler_run` (the whole CausVid loop in the library, one hipGraph per chunk) against the same sequence of `CausalWan.forward`
    calls and schedule-mirror arithmetic: bit-identical - 'ode'; 'sde' with injected noise and a context-noise cache call; chunks of 3 + 2
    frames; graph, graph replayed (other timesteps), eager; Self-Forcing's per-chunk exit steps."""
    _, net = _nets(23)
    g = torch.Generator().manual_seed(24)
    B, F, H, W = 2, 5, 16, 24
    lat = (torch.randn(B, 16, F, H, W, generator=g) * 0.999).cuda()
    text = torch.randn(B, 16, 128, generator=g).cuda()
    eps = torch.randn(3, B, 16, F, H, W, generator=g).cuda()
    tl = torch.tensor([0.999, 0.7, 0.3, 0.0], dtype=torch.float64, device="cuda")
    with torch.inference_mode():
        want = _per_call_loop(net, lat.clone(), tl, text, "ode")
        got = net.student_sample(lat.clone(), tl, text, sample_type="ode")
        assert torch.isfinite(got).all() and torch.equal(got, want)
        assert torch.equal(net.student_sample(lat.clone(), tl, text, sample_type="ode"), want)  # replay of the cached chunk graphs
        assert torch.equal(net.student_sample(lat.clone(), tl, text, sample_type="ode", use_graph=False), want)
        tl2 = torch.tensor([0.95, 0.6, 0.2, 0.0], dtype=torch.float64, device="cuda")
        assert torch.equal(net.student_sample(lat.clone(), tl2, text, sample_type="ode"), _per_call_loop(net, lat.clone(), tl2, text, "ode"))
        want = _per_call_loop(net, lat.clone(), tl, text, "sde", context_noise=0.1, eps=eps)
        got = net.student_sample(lat.clone(), tl, text, sample_type="sde", context_noise=0.1, eps=eps)
        assert torch.equal(got, want)
        assert torch.equal(net.student_sample(lat.clone(), tl, text, sample_type="sde", context_noise=0.1, eps=eps, use_graph=False), want)
        want = _per_call_loop(net, lat.clone(), tl, text, "ode", exits=[0, 2])
        assert torch.equal(net.student_sample(lat.clone(), tl, text, sample_type="ode", exit_steps=[0, 2]), want)
        # device RNG: seed control
        a = net.student_sample(lat.clone(), tl, text, sample_type="sde", seed=3)
        assert torch.equal(a, net.student_sample(lat.clone(), tl, text, sample_type="sde", seed=3))
        assert not torch.equal(a, net.student_sample(lat.clone(), tl, text, sample_type="sde", seed=4))
        # the per-call entry points still work afterwards (caches cleared, text set again on demand)
        t0 = torch.full((B,), 0.5, dtype=torch.float64, device="cuda")
        assert torch.isfinite(net(lat[:, :, :3], t0, condition=text, cur_start_frame=0, store_kv=True, is_ar=True)).all()
        net.clear_caches()


class _SFConfig:
    """The ModelConfig fields SelfForcingModel reads (configs/methods/config_self_forcing.py:23-29 + the DMD2 sampling fields)."""

    class _T:
        t_list = [0.999, 0.7, 0.3, 0.0]

    student_sample_steps, student_sample_type, sample_t_cfg = 3, "ode", _T
    same_step_across_blocks, last_step_only, context_noise = True, False, 0.0
    enable_gradient_in_rollout, start_gradient_frame = False, 0


def test_self_forcing_exit_steps():
    from fastgen_amd.methods.distribution_matching.self_forcing import SelfForcingModel

    cfg = _SFConfig()
    m = SelfForcingModel(cfg, net=None, device=torch.device("cpu"))
    torch.manual_seed(0)
    steps = m._sample_denoising_end_steps(64)
    assert len(steps) == 64 and set(steps) == {0, 1, 2}  # uniform over [0, student_sample_steps)
    cfg.last_step_only = True
    assert m._sample_denoising_end_steps(5) == [2] * 5


def test_oracle_self_forcing_rollout_exit_at_last_step_is_the_causvid_loop():
    """With every chunk leaving at the last step the rollout is CausVid's student loop (same calls, same cache fills at t = 0)."""
    cfg = R.TINY
    g = torch.Generator().manual_seed(12)
    noise = torch.randn(1, 16, 4, 4, 6, generator=g)
    text = torch.randn(1, 8, cfg.text_dim, generator=g)
    tl = torch.tensor([0.999, 0.6, 0.0], dtype=torch.float64)
    a = R.self_forcing_rollout(R.CausalWanRef(R.random_state_dict(cfg, 3), cfg), noise, tl, text, [1], sample_type="ode")
    b = R.student_sample_loop(R.CausalWanRef(R.random_state_dict(cfg, 3), cfg), noise, tl, text, sample_type="ode")
    assert torch.allclose(a, b, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("same,ends", [(True, [1, 2]), (False, [2, 0])])
def test_self_forcing_rollout_against_oracle(same, ends):
    from fastgen_amd.methods.distribution_matching.self_forcing import SelfForcingModel

    ref, net = _nets(11)
    g = torch.Generator().manual_seed(13)
    noise = torch.randn(1, 16, 5, 16, 16, generator=g)  # 5 frames, chunk_size 2: blocks of 3 + 2 frames
    text = torch.randn(1, 16, 128, generator=g)
    cfg = _SFConfig()
    cfg.same_step_across_blocks = same
    m = SelfForcingModel(cfg, net=net)
    m._sample_denoising_end_steps = lambda n: ends[:n]
    got = m.rollout_with_gradient(noise.cuda(), condition=text.cuda(), enable_gradient=False)
    want = R.self_forcing_rollout(ref, noise, torch.tensor(cfg._T.t_list, dtype=torch.float64), text, ends, same_step_across_blocks=same,
                                  sample_type="ode")
    assert got.shape == want.shape == noise.shape
    assert _rel(got.cpu(), want) < 3e-2, _rel(got.cpu(), want)
    # gradients requested at the exit step: the causal video DiT has no backward and says so
    with pytest.raises(NotImplementedError):
        m.rollout_with_gradient(noise.cuda(), condition=text.cuda(), enable_gradient=True)


def test_oracle_blockwise_causal_mask():
    """5 frames of 3 tokens, chunk_size 2: chunks of 3 + 2 frames (remainder in front, network_causal.py:163-174)."""
    m = R.blockwise_causal_mask(5, 3, 2)
    assert m.shape == (15, 15)
    assert m[:9, :9].all() and not m[:9, 9:].any()  # the first chunk sees itself only
    assert m[9:, :].all()                           # the last chunk sees everything
    # one chunk of all frames when there are fewer frames than a chunk
    assert R.blockwise_causal_mask(2, 3, 4).all()


def test_oracle_block_causal_call_equals_the_autoregressive_calls_at_t0():
    """With every frame at the same t and the cache filled by the same inputs, chunk c of the block-causal call = the autoregressive
    call on chunk c over the cache of chunks < c (same keys, same RoPE positions)."""
    cfg = R.TINY  # chunk_size 2, total_num_frames 6
    sd = R.random_state_dict(cfg, 5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 16, 6, 4, 6, generator=g)
    text = torch.randn(1, 8, cfg.text_dim, generator=g)
    t = torch.tensor([0.4])
    full = R.CausalWanRef(sd, cfg).forward(x, t, text, block_causal=True)
    ar = R.CausalWanRef(sd, cfg)
    for c in range(3):
        o = ar.forward(x[:, :, 2 * c:2 * c + 2], t, text, cur_start_frame=2 * c, store_kv=True)
        assert torch.allclose(o, full[:, :, 2 * c:2 * c + 2], atol=2e-5), c


@pytest.mark.gpu
def test_block_causal_call_against_oracle():
    """`is_ar=False` over all total_num_frames frames with per-frame timesteps (diffusion forcing), caches untouched."""
    ref, net = _nets(14)
    g = torch.Generator().manual_seed(15)
    B, H, W = 2, 16, 16
    x = torch.randn(B, 16, 6, H, W, generator=g)
    text = torch.randn(B, 24, 128, generator=g)
    t = torch.rand(B, 6, generator=g, dtype=torch.float64) * 0.9 + 0.05
    with torch.inference_mode():
        # an autoregressive call first: its cache rows must survive the full-length call
        a0 = net(x[:, :, :2].cuda(), t[:, 0].cuda(), condition=text.cuda(), fwd_pred_type="flow", cur_start_frame=0, store_kv=True, is_ar=True)
        got = net(x.cuda(), t.cuda(), condition=text.cuda(), fwd_pred_type="flow", is_ar=False)
        a1 = net(x[:, :, 2:4].cuda(), t[:, 0].cuda(), condition=text.cuda(), fwd_pred_type="flow", cur_start_frame=2, store_kv=False, is_ar=True)
    want = ref.forward(x, t, text, block_causal=True)
    assert _rel(got.cpu(), want) < 2e-2, _rel(got.cpu(), want)
    ref.clear_caches()
    w0 = ref.forward(x[:, :, :2], t[:, 0], text, cur_start_frame=0, store_kv=True)
    w1 = ref.forward(x[:, :, 2:4], t[:, 0], text, cur_start_frame=2, store_kv=False)
    assert _rel(a0.cpu(), w0) < 2e-2 and _rel(a1.cpu(), w1) < 2e-2
    with pytest.raises(NotImplementedError):
        net(x[:, :, :4].cuda(), t[:, :4].cuda(), condition=text.cuda(), is_ar=False)


class _ToyVAE:
    """decode / encode pair for the segment bridge (any pair works for the call sequence: pixels = 2 x latents + 1)."""

    @staticmethod
    def decode(z):
        return 2.0 * z + 1.0

    @staticmethod
    def encode(px):
        return 0.5 * (px - 1.0) * 0.9  # (not an exact inverse: the bridged first latent differs from the reused ones)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [0, 2])
def test_causvid_extrapolation_against_oracle(overlap, monkeypatch):
    from fastgen_amd.methods.distribution_matching.causvid import CausVidModel

    ref, net = _nets(16)
    net.vae = _ToyVAE()
    g = torch.Generator().manual_seed(17)
    B, T, H, W = 1, 4, 16, 16  # two chunks of 2 frames per segment
    noise = torch.randn(B, 16, T, H, W, generator=g)
    text = torch.randn(B, 16, 128, generator=g)
    fresh = [torch.randn(B, 16, T, H, W, generator=g) for _ in range(2)]
    t_list = [0.999, 0.6, 0.0]
    pending = [f.clone() for f in fresh]
    monkeypatch.setattr(torch, "randn_like", lambda x, **kw: pending.pop(0).to(x.device, x.dtype))  # 'ode': the only draws are the new segments' latents
    got = CausVidModel.generator_fn_extrapolation(net, noise.cuda(), condition=text.cuda(), num_segments=3, overlap_frames=overlap,
                                                  student_sample_steps=2, student_sample_type="ode", t_list=t_list)
    want = R.extrapolate(ref, noise, torch.tensor(t_list, dtype=torch.float64), text, 3, overlap,
                         lambda z: _ToyVAE.encode(_ToyVAE.decode(z)), fresh)
    assert got.shape == want.shape == (B, 16, 3 * T - 2 * overlap, H, W)
    assert _rel(got.cpu(), want) < 4e-2, _rel(got.cpu(), want)
    with pytest.raises(ValueError):
        CausVidModel.generator_fn_extrapolation(net, noise.cuda(), condition=text.cuda(), num_segments=2, overlap_frames=1, t_list=t_list,
                                                student_sample_steps=2)


@pytest.mark.gpu
def test_block_causal_call_with_a_frame_remainder():
    """5 frames at chunk_size 2: the mask's first chunk holds 3 frames (network_causal.py:163-174); odd token counts per frame
    (10 x 6 latents = 15 tokens: query tiles and key tiles end inside a chunk)."""
    import dataclasses

    from fastgen_amd.networks.Wan.network_causal import CausalWan

    cfg = dataclasses.replace(R.TINY, total_num_frames=5)
    sd = R.random_state_dict(cfg, 21)
    ref = R.CausalWanRef(sd, cfg)
    net = CausalWan(**dict(KW, total_num_frames=5))
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(22)
    B = 2
    x = torch.randn(B, 16, 5, 10, 6, generator=g)
    text = torch.randn(B, 9, 128, generator=g)
    t = torch.rand(B, 5, generator=g, dtype=torch.float64) * 0.9 + 0.05
    with torch.inference_mode():
        got = net(x.cuda(), t.cuda(), condition=text.cuda(), fwd_pred_type="flow", is_ar=False)
    want = ref.forward(x, t, text, block_causal=True)
    assert _rel(got.cpu(), want) < 2e-2, _rel(got.cpu(), want)
    # the chunked loop over the same 5 frames: chunks of 3 + 2
    from fastgen_amd.methods.distribution_matching.causvid import CausVidModel

    tl = [0.999, 0.5, 0.0]
    noise = torch.randn(B, 16, 5, 10, 6, generator=g)
    got = CausVidModel.generator_fn(net, noise.cuda(), student_sample_steps=2, t_list=tl, condition=text.cuda(), student_sample_type="ode")
    want = R.student_sample_loop(ref, noise * 0.999, torch.tensor(tl, dtype=torch.float64), text, sample_type="ode")
    assert _rel(got.cpu(), want) < 3e-2, _rel(got.cpu(), want)
