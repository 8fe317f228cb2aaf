"""DiT.  CPU: the oracle restatement (oracle/dit_ref.py) against vectors recorded from the reference's own DiT class
(built on restated timm stand-ins: oracle/_timm_restated.py, `python oracle/gen_golden.py dit`), and the drop-in module's state dict.
GPU (-m gpu): the HIP path through the C ABI (fg_dit_forward) against those vectors and the oracle.

Tolerances on the O(1) output: bf16x3 (default, fp32 tensors) and exact fp32: max |err| <= 2e-4, relative L2 <= 5e-5 (28 blocks of
width 1152: three times the contraction length and twice the depth of the EDM U-Net); bf16: relative L2 <= 2e-2."""
import os

import pytest
import torch

from oracle import dit_ref as R

CFGS = {"xl": R.XL_2, "s": R.S_2, "xl_r": R.DiTConfig(r_timestep=True)}
KW = {"xl": dict(hidden_size=1152, depth=28, num_heads=16), "s": dict(hidden_size=384, depth=12, num_heads=6),
      "xl_r": dict(hidden_size=1152, depth=28, num_heads=16, r_timestep=True)}


def _inputs(fx, tag):
    B = 2
    x = torch.randn((B, 4, 32, 32), generator=torch.Generator().manual_seed(501))
    cond = torch.zeros(B, 1000)
    cond[0, 417] = 1.0  # row 1 all-zero: the unconditional class
    r = fx.get(f"{tag}/r")
    return x, fx[f"{tag}/t"], cond, r


@pytest.mark.parametrize("tag", ["s", "xl", "xl_r"])
def test_oracle_against_reference_golden(golden_dir, tag):
    fx = torch.load(os.path.join(golden_dir, "dit_forward_b2.pt"), weights_only=True)
    cfg = CFGS[tag]
    sd = R.random_state_dict(cfg, seed=77)
    x, t, cond, r = _inputs(fx, tag)
    tr = {}
    out = R.dit_forward(sd, cfg, x, t, cond, r=r, trace=tr)
    assert (out - fx[f"{tag}/out"]).abs().max() < 5e-5
    assert (tr["c"] - fx[f"{tag}/c"]).abs().max() < 1e-5
    for i in (0, cfg.depth // 2, cfg.depth - 1):
        v = tr[f"block{i}"].reshape(-1)
        smp = v[:: max(1, v.numel() // 1024)][:1024]
        assert float((smp - fx[f"{tag}/block{i}/sample"]).norm() / fx[f"{tag}/block{i}/sample"].norm()) < 1e-5


@pytest.mark.parametrize("tag", ["s", "xl"])
def test_module_state_dict_is_the_references(golden_dir, tag):
    from fastgen_amd.networks.DiT.network import DiT

    net = DiT(**KW[tag])
    want = {}
    for line in open(os.path.join(golden_dir, f"dit_{tag}_state_dict_keys.txt")):
        p = line.split()
        want[p[0]] = tuple(int(v) for v in p[1:])
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want and list(got) == list(want)
    sd = R.random_state_dict(CFGS[tag], seed=77)
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # the positional table is the reference's (recorded there as a persistent buffer)
    assert torch.allclose(net.pos_embed, R.pos_embed_2d(KW[tag]["hidden_size"], 16).unsqueeze(0), atol=1e-6)
    with pytest.raises(RuntimeError, match="HIP GPU only"):
        with torch.no_grad():
            net(torch.zeros(1, 4, 32, 32), torch.full((1,), 0.5, dtype=torch.float64), condition=torch.zeros(1, 1000))
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 4, 32, 32), torch.full((1,), 0.5, dtype=torch.float64), condition=torch.zeros(1, 1000))  # autograd


TOL = {"fp32": (2e-4, 5e-5), "bf16x3": (2e-4, 5e-5), "bf16": (1e-1, 2e-2)}


@pytest.mark.gpu
@pytest.mark.parametrize("tag,mode", [("s", "fp32"), ("s", "bf16x3"), ("s", "bf16"), ("xl", "bf16x3"), ("xl", "bf16"), ("xl_r", "bf16x3")])
def test_forward_against_reference_golden(golden_dir, tag, mode):
    from fastgen_amd.networks.DiT.network import DiT

    fx = torch.load(os.path.join(golden_dir, "dit_forward_b2.pt"), weights_only=True)
    dev = torch.device("cuda:0")
    net = DiT(compute_dtype=mode, **KW[tag])
    net.load_state_dict(R.random_state_dict(CFGS[tag], seed=77), strict=True)
    net = net.to(dev).eval()
    x, t, cond, r = _inputs(fx, tag)
    with torch.inference_mode():
        out = net(x.to(dev), t.to(dev), condition=cond.to(dev), r=None if r is None else r.to(dev)).cpu()
    want = fx[f"{tag}/out"]
    err, rel = float((out - want).abs().max()), float((out - want).norm() / want.norm())
    assert torch.isfinite(out).all() and err <= TOL[mode][0] and rel <= TOL[mode][1], (tag, mode, err, rel)
    if tag == "s" and mode == "bf16x3":
        with torch.inference_mode():
            # class indices instead of one-hot rows, x0 conversion of the flow prediction, ragged batch, determinism
            ids = torch.tensor([417, 1000], device=dev)
            assert torch.equal(net(x.to(dev), t.to(dev), condition=ids).cpu(), out)
            x0 = net(x.to(dev), t.to(dev), condition=ids, fwd_pred_type="x0").cpu()
            assert torch.allclose(x0, x - t.reshape(2, 1, 1, 1).float() * out, atol=1e-5)
            x5 = torch.randn((5, 4, 32, 32), generator=torch.Generator().manual_seed(9))
            t5 = torch.tensor([0.9, 0.7, 0.5, 0.3, 0.1], dtype=torch.float64)
            c5 = torch.nn.functional.one_hot(torch.tensor([1, 2, 3, 4, 5]), 1000).float()
            got5 = net(x5.to(dev), t5.to(dev), condition=c5.to(dev)).cpu()
            want5 = R.dit_forward(R.random_state_dict(CFGS[tag], seed=77), CFGS[tag], x5, t5, c5)
            assert float((got5 - want5).abs().max()) <= TOL[mode][0]
            assert torch.equal(got5[1:3], net(x5[1:3].to(dev), t5[1:3].to(dev), condition=c5[1:3].to(dev)).cpu())  # batch independence


@pytest.mark.gpu
@pytest.mark.parametrize("tag,mode", [("s", "bf16"), ("xl", "bf16"), ("s", "bf16x3")])
def test_large_batch_paths_against_reference_golden(golden_dir, tag, mode):
    """Batch 256 takes code the batch-2 fixtures do not reach (bf16: all blocks' conditioning linears as one stacked GEMM; whole
    256-token tiles of the ping-pong GEMM; DiT-XL: the LDS-staged attention): the golden pair, repeated, must come back."""
    from fastgen_amd.networks.DiT.network import DiT

    fx = torch.load(os.path.join(golden_dir, "dit_forward_b2.pt"), weights_only=True)
    dev = torch.device("cuda:0")
    net = DiT(compute_dtype=mode, **KW[tag])
    net.load_state_dict(R.random_state_dict(CFGS[tag], seed=77), strict=True)
    net = net.to(dev).eval()
    x, t, cond, r = _inputs(fx, tag)
    rep = 128
    with torch.inference_mode():
        out = net(x.repeat(rep, 1, 1, 1).to(dev), t.repeat(rep).to(dev), condition=cond.repeat(rep, 1).to(dev)).cpu()
    want = fx[f"{tag}/out"]
    for i in (0, 1, 126, 127):
        got = out[2 * i: 2 * i + 2]
        err, rel = float((got - want).abs().max()), float((got - want).norm() / want.norm())
        assert err <= TOL[mode][0] and rel <= TOL[mode][1], (tag, mode, i, err, rel)


@pytest.mark.gpu
def test_flow_sampler_against_oracle():
    """RF Euler sampler of the reference (`DiT._sample_flow`, :605-651) with classifier-free guidance, 4 steps, DiT-S/2: the module's
    `sample()` on the HIP network against the same loop run on the oracle network."""
    from fastgen_amd.networks.DiT.network import DiT

    dev = torch.device("cuda:0")
    cfg = R.S_2
    sd = R.random_state_dict(cfg, seed=77)
    net = DiT(**KW["s"])
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    noise = torch.randn((2, 4, 32, 32), generator=torch.Generator().manual_seed(3))
    cond = torch.nn.functional.one_hot(torch.tensor([7, 99]), 1000).float()
    neg = torch.zeros(2, 1000)
    got = net.sample(noise.to(dev), condition=cond.to(dev), neg_condition=neg.to(dev), guidance_scale=2.0, num_steps=4).cpu()
    tl = net.noise_scheduler.get_t_list(4)
    x = net.noise_scheduler.latents(noise=noise, t_init=tl[0])
    for t, tn in zip(tl[:-1], tl[1:]):
        tb = t.expand(2)
        v = R.dit_forward(sd, cfg, torch.cat([x, x]), torch.cat([tb, tb]), torch.cat([neg, cond]))
        vu, vc = v.chunk(2)
        x = x + (tn - t).to(x.dtype) * (vu + 2.0 * (vc - vu))
    assert float((got - x).abs().max()) <= 5e-4 and float((got - x).norm() / x.norm()) <= 1e-4


@pytest.mark.gpu
def test_class_index_outside_the_embedding_table_is_reported():
    """`y_embedder.class_embeddings` has the extra "unconditional" row only for class_dropout_prob > 0 (DiT/network.py:116-118).  Without
    it an all-zero one-hot row (-> index num_classes) or any out-of-range id would read past the table; the reference's nn.Embedding
    device-asserts.  Here: the one-hot form is refused up front; an id tensor poisons that sample (NaN, nothing is clamped or read) and
    the handle's next call reports it."""
    from fastgen_amd import _lib
    from fastgen_amd.networks.DiT.network import DiT

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DiT(class_dropout_prob=0.0, compute_dtype="bf16x3", **KW["s"]).to(dev).eval()
    assert net.state_dict()["y_embedder.class_embeddings.weight"].shape[0] == 1000
    x = torch.randn(2, 4, 32, 32, device=dev)
    t = torch.tensor([0.7, 0.3], dtype=torch.float64, device=dev)
    cond = torch.zeros(2, 1000, device=dev)
    cond[0, 5] = 1.0
    with torch.inference_mode():
        with pytest.raises(ValueError, match="class_dropout_prob"):
            net(x, t, condition=cond)
        good = net(x, t, condition=torch.tensor([5, 999], device=dev))
        assert torch.isfinite(good).all()
        bad = net(x, t, condition=torch.tensor([5, 1000], device=dev))
        torch.cuda.synchronize()
        assert torch.equal(bad[0], good[0]) and torch.isnan(bad[1]).all()
        with pytest.raises(_lib.FastGenAMDError, match="class index"):
            net(x, t, condition=torch.tensor([5, 999], device=dev))
        assert torch.equal(net(x, t, condition=torch.tensor([5, 999], device=dev)), good)  # reported once, then back to normal


def _euler_per_step(net, noise, cond, neg, g, steps):
    """`DiT._sample_flow` (DiT/network.py:605-651) step by step through the module's forward: the loop `fg_dit_sampler_run(FG_LOOP_EULER)` fuses."""
    sch = net.noise_scheduler
    tl = sch.get_t_list(steps, device=noise.device)
    x = sch.latents(noise=noise, t_init=tl[0])
    n = x.shape[0]
    for t, tn in zip(tl[:-1], tl[1:]):
        if neg is not None:
            vu, vc = net(torch.cat([x, x]), torch.cat([t.expand(n)] * 2), condition=torch.cat([neg, cond]), fwd_pred_type="flow").chunk(2)
            v = vu + g * (vc - vu)
        else:
            v = net(x, t.expand(n), condition=cond, fwd_pred_type="flow")
        x = x + (tn - t).to(x.dtype) * v
    return x


@pytest.mark.gpu
@pytest.mark.parametrize("tag,mode,B", [("s", "bf16x3", 3), ("xl_r", "bf16", 2)])
def test_fused_sampler_loops_equal_the_per_step_loops(tag, mode, B, monkeypatch):
    """`fg_dit_sampler_run` - the x0 student loop (methods/model.py:315-420), the MeanFlow loop (mean_flow.py:336-381) and the Euler
    sampler with classifier-free guidance (DiT/network.py:605-651) as one library call each, replayed as a hipGraph - against the same
    loops run step by step through `DiT.forward` and the schedule mirror: bit-identical ('sde' with injected noise, 'ode'; graph and
    eager; a replay with other timesteps)."""
    from fastgen_amd.methods.consistency_model.mean_flow import MeanFlowModel
    from fastgen_amd.methods.model import FastGenModel
    from fastgen_amd.networks.DiT.network import DiT

    dev = torch.device("cuda:0")
    net = DiT(compute_dtype=mode, **KW[tag])
    net.load_state_dict(R.random_state_dict(CFGS[tag], seed=77), strict=True)
    net = net.to(dev).eval()
    g = torch.Generator().manual_seed(31)
    noise = torch.randn((B, 4, 32, 32), generator=g).to(dev)
    cond = torch.nn.functional.one_hot(torch.tensor([3, 500, 999][:B]), 1000).float().to(dev)
    eps = torch.randn((3, B, 4, 32, 32), generator=g).to(dev)
    sch = net.noise_scheduler

    def per_step(model_cls, steps, kind, t_list=None):
        tl = sch.get_t_list(steps, device=dev) if t_list is None else torch.tensor(t_list, dtype=torch.float64, device=dev)
        pending = [e for e in eps[: steps - 1]]
        monkeypatch.setattr(torch, "randn_like", lambda x, **kw: pending.pop(0))
        with torch.inference_mode():
            out = model_cls._student_sample_loop(net, sch.latents(noise, tl[0]), tl, condition=cond, student_sample_type=kind)
        monkeypatch.undo()
        return out

    gf = FastGenModel.generator_fn
    for kind in ("sde", "ode"):
        want = per_step(FastGenModel, 4, kind)
        got = gf(net, noise, student_sample_steps=4, condition=cond, student_sample_type=kind, eps=eps)
        assert torch.isfinite(got).all() and torch.equal(got, want), kind
        assert torch.equal(gf(net, noise, student_sample_steps=4, condition=cond, student_sample_type=kind, eps=eps, use_graph=False), want)
    # the cached graph replayed with another timestep list (same steps / zero pattern), then 1 step
    tl2 = [0.9, 0.61, 0.33, 0.12, 0.0]
    assert torch.equal(gf(net, noise, student_sample_steps=4, t_list=tl2, condition=cond, student_sample_type="ode"), per_step(FastGenModel, 4, "ode", tl2))
    assert torch.equal(gf(net, noise, student_sample_steps=1, condition=cond), per_step(FastGenModel, 1, "sde"))
    # device RNG: seed control
    a = gf(net, noise, student_sample_steps=3, condition=cond, seed=5)
    assert torch.equal(a, gf(net, noise, student_sample_steps=3, condition=cond, seed=5))
    assert not torch.equal(a, gf(net, noise, student_sample_steps=3, condition=cond, seed=6))
    if CFGS[tag].r_timestep:
        mf = MeanFlowModel.generator_fn
        for kind in ("sde", "ode"):
            want = per_step(MeanFlowModel, 4, kind)
            got = mf(net, noise, student_sample_steps=4, condition=cond, student_sample_type=kind, eps=eps)
            assert torch.isfinite(got).all() and torch.equal(got, want), kind
        assert torch.equal(mf(net, noise, student_sample_steps=2, t_list=[0.999, 0.5, 0.0], condition=cond, student_sample_type="ode"),
                           per_step(MeanFlowModel, 2, "ode", [0.999, 0.5, 0.0]))
    # Euler sampler, guided and plain
    neg = torch.zeros(B, 1000, device=dev)
    with torch.inference_mode():
        want = _euler_per_step(net, noise, cond, neg, 2.5, 4)
        assert torch.equal(net.sample(noise, condition=cond, neg_condition=neg, guidance_scale=2.5, num_steps=4), want)
        assert torch.equal(net.sample(noise, condition=cond, neg_condition=neg, guidance_scale=2.5, num_steps=4, use_graph=False), want)
        want = _euler_per_step(net, noise, cond, None, 1.0, 3)
        assert torch.equal(net.sample(noise, condition=cond, guidance_scale=None, num_steps=3), want)
    # refused, not approximated
    from fastgen_amd import _lib

    with pytest.raises(_lib.FastGenAMDError):
        gf(net, noise, student_sample_steps=1, t_list=[1.5, 0.0], condition=cond)
    if not CFGS[tag].r_timestep:
        with pytest.raises(NotImplementedError):
            net.few_step_sample(noise, cond, [0.999, 0.0], loop="meanflow")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_feature_taps(golden_dir, mode):
    """`feature_indices` / `return_features_early` (DiT/network.py:483-484, 536-543, 563-566): the token tensors behind the requested
    blocks, against the oracle's trace of the same forward and the strided samples recorded from the reference's own blocks."""
    from fastgen_amd.networks.DiT.network import DiT

    fx = torch.load(os.path.join(golden_dir, "dit_forward_b2.pt"), weights_only=True)
    dev = torch.device("cuda:0")
    cfg = CFGS["s"]
    sd = R.random_state_dict(cfg, seed=77)
    net = DiT(compute_dtype=mode, **KW["s"])
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    x, t, cond, _ = _inputs(fx, "s")
    tr = {}
    R.dit_forward(sd, cfg, x, t, cond, trace=tr)
    xd, td, cd = x.to(dev), t.to(dev), cond.to(dev)
    tol = 5e-5 if mode == "bf16x3" else 2e-2
    with torch.inference_mode():
        plain = net(xd, td, condition=cd)
        out, feats = net(xd, td, condition=cd, feature_indices={0, 6, 11})
        early = net(xd, td, condition=cd, feature_indices={0, 6}, return_features_early=True)
        (out_lv, feats_lv), logvar = net(xd, td, condition=cd, feature_indices={11}, return_logvar=True)
        assert net(xd, td, condition=cd, return_features_early=True) == []
        out_past, feats_past = net(xd, td, condition=cd, feature_indices={6, 40}, return_features_early=True)  # block 40 never comes: no early return
    assert torch.equal(out, plain) and torch.equal(out_lv, plain) and torch.equal(out_past, plain) and logvar.shape == (2, 1)
    assert [tuple(f.shape) for f in feats] == [(2, 256, 384)] * 3 and len(early) == 2 and len(feats_lv) == 1 and len(feats_past) == 1
    for f, i in zip(feats, (0, 6, 11)):
        want = tr[f"block{i}"]
        assert float((f.cpu() - want).norm() / want.norm()) <= tol, i
        v = f.float().cpu().reshape(-1)
        smp = v[:: max(1, v.numel() // 1024)][:1024]
        assert float((smp - fx[f"s/block{i}/sample"]).norm() / fx[f"s/block{i}/sample"].norm()) <= tol, i
    assert torch.equal(early[0], feats[0]) and torch.equal(early[1], feats[1]) and torch.equal(feats_lv[0], feats[2]) and torch.equal(feats_past[0], feats[1])
