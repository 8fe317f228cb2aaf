"""Pins oracle/edm_ref.py (the CPU restatement) to golden vectors recorded from the reference itself
(oracle/gen_golden.py).  CPU only."""
import os

import pytest
import torch

from oracle import edm_ref as R


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=True)


def _cs(t):
    t = t.detach().to(torch.float64).reshape(-1)
    w = torch.arange(1, t.numel() + 1, dtype=torch.float64) % 7 + 1
    return torch.stack([t.sum(), (t * w).sum(), t.abs().max()])


def _sd_cs(sd):
    return torch.stack([_cs(v) for _, v in sorted(sd.items())]).sum(0)


def _seeded(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.fixture(scope="module")
def full_sd():
    return R.random_state_dict(R.CIFAR10, seed=1234)


def test_state_dict_names_match_reference(golden_dir):
    """429 entries, names and shapes exactly as the reference's EDMPrecond.state_dict()."""
    want = {}
    for line in open(os.path.join(golden_dir, "state_dict_keys.txt")):
        parts = line.split()
        want[parts[0]] = tuple(int(p) for p in parts[1:])
    got = R.param_shapes(R.CIFAR10)
    assert len(want) == 429
    assert got == want


def test_schedule(golden_dir):
    fx = _load(golden_dir, "schedule.pt")
    sig = R.edm_sigmas()
    assert torch.equal(sig[:3], fx["sigmas_head"]) and torch.equal(sig[-3:], fx["sigmas_tail"])
    for n in (1, 2, 4):
        assert torch.equal(R.edm_t_list(n), fx[f"t_list_{n}"])
    x, e = _seeded((2, 3, 8, 8), 11), _seeded((2, 3, 8, 8), 12)
    t = torch.tensor([17.498123, 0.1726], dtype=torch.float64)
    assert torch.equal(R.forward_process(x, e, t), fx["fp_out"])
    assert torch.equal(R.latents(x, torch.tensor(79.5638, dtype=torch.float64)), fx["lat_out"])
    assert torch.equal(R.x0_to_eps(x, e, t), fx["x0eps_out"])


def test_schedule_analytic_identities():
    """The reference's own schedule checks (tests/test_network.py:112-131): EDM has alpha=1, sigma=t."""
    x, e = _seeded((4, 3, 8, 8), 1), _seeded((4, 3, 8, 8), 2)
    t = torch.tensor([0.002, 1.0, 10.0, 80.0], dtype=torch.float64)
    xt = R.forward_process(x, e, t)
    assert torch.allclose(xt, x + e * t.float().reshape(-1, 1, 1, 1), atol=1e-5)
    assert torch.allclose(R.x0_to_eps(xt, x, t), e, atol=2e-3)
    tl = R.edm_t_list(4)
    assert tl[-1] == 0 and torch.all(tl[:-1] > tl[1:]) and tl[0] <= 80.0


def test_forward_small(golden_dir):
    fx = _load(golden_dir, "forward_small.pt")
    cfg = R.SongUNetConfig(img_resolution=8, model_channels=32, channel_mult=(1, 2), num_blocks=1, attn_resolutions=(4,))
    sd = R.random_state_dict(cfg, seed=77)
    assert torch.allclose(_sd_cs(sd), fx["sd_checksum"])
    x = _seeded((3, 3, 8, 8), 31)
    t = torch.tensor([80.0, 2.5, 0.002], dtype=torch.float64)
    c = torch.nn.functional.one_hot(torch.tensor([0, 5, 9]), 10).float()
    with torch.inference_mode():
        assert torch.allclose(R.edm_precond_forward(sd, cfg, x, t, c), fx["out"], rtol=1e-5, atol=1e-5)
        assert torch.allclose(R.edm_precond_forward(sd, cfg, x, t, None), fx["out_nocond"], rtol=1e-5, atol=1e-5)


def test_forward_full_b2(golden_dir, full_sd):
    fx = _load(golden_dir, "forward_full_b2.pt")
    assert torch.allclose(_sd_cs(full_sd), fx["sd_checksum"])
    x = _seeded((2, 3, 32, 32), 21)
    assert torch.allclose(_cs(x), fx["x_checksum"])
    trace = {}
    with torch.inference_mode():
        out = R.edm_precond_forward(full_sd, R.CIFAR10, x * fx["t"].reshape(2, 1, 1, 1).float(), fx["t"], fx["cond"], trace)
    assert torch.allclose(trace["emb"], fx["emb"], rtol=1e-5, atol=1e-6)
    for key, v in trace.items():
        if key == "emb":
            continue
        name = key.replace("model.", "")
        want = fx[f"blk/{name}/sample"]
        got = v.reshape(-1)[:: max(1, v.numel() // 4096)][:4096]
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-4), name
    assert torch.allclose(out, fx["out"], rtol=1e-4, atol=1e-5)


def test_blocks_full(golden_dir, full_sd):
    fx = _load(golden_dir, "blocks_full.pt")
    enc, dec = R.layout(R.CIFAR10)
    by_key = {b.key: b for b in enc + dec}
    names = sorted({k.split("/")[0] for k in fx if "/" in k})
    assert len(names) == 9
    for n in names:
        b = by_key[fx[f"{n}/key"]]
        bs = fx[f"{n}/out"].shape[0]
        rin = b.res * 2 if b.down else (b.res // 2 if b.up else b.res)
        x = _seeded((bs, b.cin, rin, rin), int(fx[f"{n}/seed"]))
        assert torch.allclose(_cs(x), fx[f"{n}/x_checksum"])
        with torch.inference_mode():
            y = R.unet_block(full_sd, b, x, fx["emb"][:bs])
        assert torch.allclose(y, fx[f"{n}/out"], rtol=1e-5, atol=1e-5), n


def test_sampler_full_b2(golden_dir, full_sd):
    fx = _load(golden_dir, "sampler_full_b2.pt")
    noise = _seeded((2, 3, 32, 32), 0)
    assert torch.allclose(_cs(noise), fx["noise_checksum"])
    eps = [_seeded((2, 3, 32, 32), s) for s in (1, 2, 3)]
    tr = {}
    out = R.generator_fn(full_sd, R.CIFAR10, noise, fx["cond"], 4, sample_type="sde", eps_list=eps, trace=tr)
    assert torch.allclose(torch.stack(tr["x_pred"]), fx["x_pred_sde"], rtol=1e-4, atol=2e-5)
    assert torch.allclose(out, fx["out_sde"], rtol=1e-4, atol=2e-5)
    assert torch.allclose(R.generator_fn(full_sd, R.CIFAR10, noise, fx["cond"], 4, sample_type="ode"), fx["out_ode"],
                          rtol=1e-4, atol=2e-5)
    assert torch.allclose(R.generator_fn(full_sd, R.CIFAR10, noise, fx["cond"], 1), fx["out_1step"], rtol=1e-4, atol=2e-5)
    assert torch.allclose(R.generator_fn(full_sd, R.CIFAR10, noise, fx["cond"], 2, t_list=[40.0, 1.5, 0.0],
                                         sample_type="ode"), fx["out_tlist"], rtol=1e-4, atol=2e-5)


# ---- MeanFlow student: r_timestep network, drop_precond, rectified-flow schedule -----------------------------------


@pytest.fixture(scope="module")
def mf_sd():
    return R.random_state_dict(R.CIFAR10_MEANFLOW, seed=4321)


def test_meanflow_state_dict_names_match_reference(golden_dir):
    want = {}
    for line in open(os.path.join(golden_dir, "state_dict_keys_meanflow.txt")):
        parts = line.split()
        want[parts[0]] = tuple(int(p) for p in parts[1:])
    got = R.param_shapes(R.CIFAR10_MEANFLOW)
    assert got == want
    assert got["model.map_layer0.weight"] == (512, 256) and "model.map_label.weight" not in got


def test_schedule_rf(golden_dir):
    fx = _load(golden_dir, "schedule_rf.pt")
    for n in (1, 2, 4):
        assert torch.equal(R.rf_t_list(n), fx[f"t_list_{n}"])
    x, e = _seeded((2, 3, 8, 8), 11), _seeded((2, 3, 8, 8), 12)
    t = torch.tensor([0.7492, 0.2497], dtype=torch.float64)
    assert torch.equal(R.forward_process(x, e, t, "rf"), fx["fp_out"])
    assert torch.equal(R.latents(x, torch.tensor(0.999, dtype=torch.float64)), fx["lat_out"])
    assert torch.equal(R.x0_to_eps(x, e, t, schedule="rf"), fx["x0eps_out"])


def test_meanflow_forward_and_drop_precond(golden_dir, mf_sd):
    fx = _load(golden_dir, "meanflow_full_b2.pt")
    assert torch.allclose(_sd_cs(mf_sd), fx["sd_checksum"])
    x = _seeded((2, 3, 32, 32), 41)
    assert torch.allclose(_cs(x), fx["x_checksum"])
    cfg = R.CIFAR10_MEANFLOW
    with torch.inference_mode():
        tr = {}
        out = R.edm_precond_forward(mf_sd, cfg, x, fx["t"], None, trace=tr, r=fx["r"])
        assert torch.allclose(tr["emb"], fx["emb"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(out, fx["out"], rtol=1e-4, atol=2e-5)
        for dp in (None, "input", "output"):
            cfg_v = R.SongUNetConfig(**{**cfg.__dict__, "drop_precond": dp})
            got = R.edm_precond_forward(mf_sd, cfg_v, x, fx["t"], None, r=fx["r"])
            assert torch.allclose(got, fx[f"out_drop_{dp}"], rtol=1e-4, atol=2e-5), dp
        with pytest.raises(ValueError):  # EDM/network.py:510
            R.edm_precond_forward(R.random_state_dict(R.CIFAR10, 1), R.CIFAR10, x, fx["t"], None, r=fx["r"])


def test_meanflow_sampler(golden_dir, mf_sd):
    fx = _load(golden_dir, "meanflow_full_b2.pt")
    cfg = R.CIFAR10_MEANFLOW
    noise = _seeded((2, 3, 32, 32), 5)
    assert torch.allclose(_cs(noise), fx["noise_checksum"])
    eps = [_seeded((2, 3, 32, 32), s) for s in (6, 7, 8)]
    kw = dict(rtol=1e-4, atol=2e-5)
    assert torch.allclose(R.generator_fn(mf_sd, cfg, noise, None, 4, sample_type="sde", eps_list=eps, loop="meanflow"),
                          fx["out_sde"], **kw)
    assert torch.allclose(R.generator_fn(mf_sd, cfg, noise, None, 4, sample_type="ode", loop="meanflow"), fx["out_ode"], **kw)
    assert torch.allclose(R.generator_fn(mf_sd, cfg, noise, None, 1, sample_type="ode", loop="meanflow"), fx["out_1step"], **kw)
    assert torch.allclose(R.generator_fn(mf_sd, cfg, noise, None, 2, t_list=[0.999, 0.5, 0.0], sample_type="ode",
                                         loop="meanflow"), fx["out_tlist"], **kw)


def test_teacher_euler_sampler(golden_dir, full_sd):
    """EDMPrecond.sample with and without classifier-free guidance (EDM/network.py:976-1026)."""
    fx = _load(golden_dir, "teacher_sample_b2.pt")
    noise = _seeded((2, 3, 32, 32), 50)
    assert torch.allclose(_cs(noise), fx["noise_checksum"])
    with torch.inference_mode():
        got = R.edm_sample(full_sd, R.CIFAR10, noise, fx["cond"], torch.zeros(2, 10), 2.0, 4)
        assert got.dtype == fx["out_cfg"].dtype == torch.float64
        assert torch.allclose(got, fx["out_cfg"], rtol=1e-6, atol=1e-7)
        got = R.edm_sample(full_sd, R.CIFAR10, noise, fx["cond"], None, None, 3)
        assert torch.allclose(got, fx["out_plain"], rtol=1e-6, atol=1e-7)


def test_images_to_uint8_known_answers():
    """Sample-writer conversion (scripts/fid/compute_fid_from_ckpts.py:199): hand-derived bytes.  -1 -> 0.5 -> 0;
    0 -> 128; 1 -> 255.5 -> clip 255; (k - 128) / 127.5 lands on or just around the integer k; out-of-range saturates."""
    vals = torch.tensor([-1.0, 0.0, 1.0, -5.0, 7.0, -1.0 + 1 / 127.5, 0.999, 0.5, -0.5, 2 / 255 - 1], dtype=torch.float32)
    want = [0, 128, 255, 0, 255, 1, 255, 191, 64, 1]
    # 0.999 * 127.5 + 128 = 255.3725 -> 255;  0.5 -> 191.75 -> 191;  -0.5 -> 64.25 -> 64;  2/255 - 1 -> 1.5 -> 1
    x = vals.reshape(1, 1, 1, -1).expand(2, 3, 1, -1)
    got = R.images_to_uint8(x)
    assert got.dtype == torch.uint8 and got.shape == (2, 1, vals.numel(), 3)
    for c in range(3):
        assert got[0, 0, :, c].tolist() == want
    # layout: channel is innermost in the output
    y = torch.stack([torch.full((2, 2), -1.0), torch.zeros(2, 2), torch.ones(2, 2)]).unsqueeze(0)
    assert R.images_to_uint8(y)[0, 1, 1].tolist() == [0, 128, 255]


def test_conv_weight_grad_matches_reference_autograd(golden_dir):
    """oracle.conv_weight_grad against the weight gradients recorded from the reference's Conv2d under autograd
    (oracle/gen_golden.py backward): same seeded operands, fp32."""
    fx = torch.load(os.path.join(golden_dir, "backward_conv.pt"), weights_only=True)
    for ks in (3, 1):
        B, cin, cout, res, k = fx[f"k{ks}/shape"].tolist()
        assert k == ks
        x = torch.randn((B, cin, res, res), generator=torch.Generator().manual_seed(53 + ks))
        dy = torch.randn((B, cout, res, res), generator=torch.Generator().manual_seed(54 + ks))
        got = R.conv_weight_grad(x, dy, ks)
        want = fx[f"k{ks}/weight_grad"]
        assert got.shape == want.shape
        assert (got - want).abs().max() <= 1e-4 * want.abs().max()
        # the bias gradient is the plain sum of dy over batch and pixels
        assert torch.allclose(dy.sum((0, 2, 3)), fx[f"k{ks}/bias_grad"], rtol=1e-5, atol=1e-4)


def _seeded_discriminator(idx):
    """The module with the parameters of tests/golden/discriminator_edm.pt: drawn in named_parameters() order from one
    generator (oracle/gen_golden.py discriminator_fixture).  The constructor is plain torch and runs without a GPU."""
    from fastgen_amd.networks.discriminators import Discriminator_EDM

    d = Discriminator_EDM(feature_indices=idx)
    g = torch.Generator().manual_seed(501)
    with torch.no_grad():
        for n, p in d.named_parameters():
            if p.ndim == 1 and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.ndim == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / (p[0].numel() ** 0.5))
    return d


@pytest.mark.parametrize("tag,idx", [("default", None), ("all", {0, 1, 2})])
def test_discriminator_oracle_and_module_tree_match_reference(golden_dir, tag, idx):
    """State-dict keys of the drop-in module equal the reference's, and the oracle's forward / autograd reproduce the logits and
    gradients recorded from the reference's Discriminator_EDM (fp32, 1e-4)."""
    fx = torch.load(os.path.join(golden_dir, "discriminator_edm.pt"), weights_only=True)
    d = _seeded_discriminator(idx)
    assert list(d.state_dict().keys()) == fx[f"{tag}/keys"]
    assert d.in_res == fx[f"{tag}/in_res"].tolist()
    bs = int(fx[f"{tag}/bs"])
    sd = {k: v.clone().requires_grad_(True) for k, v in d.state_dict().items()}
    feats = [torch.randn((bs, 256, r, r), generator=torch.Generator().manual_seed(510 + r)).requires_grad_(True) for r in d.in_res]
    with torch.enable_grad():
        logits = R.discriminator_edm(sd, feats, d.in_res)
        assert torch.allclose(logits, fx[f"{tag}/logits"], rtol=1e-4, atol=1e-5)
        dl = torch.randn(tuple(logits.shape), generator=torch.Generator().manual_seed(520))
        logits.backward(dl)
    for r, f in zip(d.in_res, feats):
        assert abs(float(f.grad.double().norm()) / float(fx[f"{tag}/dfeat{r}/norm"]) - 1) <= 1e-4
    for k, v in sd.items():
        assert abs(float(v.grad.double().norm()) / float(fx[f"{tag}/{k}/norm"]) - 1) <= 1e-4, k


def test_oracle_jvp_matches_reference_func_jvp(golden_dir, full_sd, mf_sd):
    """Forward-mode derivative of the oracle against `torch.func.jvp` through the reference's modules (with its hand-written
    AttentionOp.jvp), tests/golden/jvp_b2.pt: the MeanFlow network with tangents (v, 1, 0) and the preconditioned network."""
    fx = torch.load(os.path.join(golden_dir, "jvp_b2.pt"), weights_only=True)
    v = torch.randn((2, 3, 32, 32), generator=torch.Generator().manual_seed(72))
    x = torch.randn((2, 3, 32, 32), generator=torch.Generator().manual_seed(71))
    t, r = fx["mf/t"], fx["mf/r"]
    out, jv = R.edm_precond_jvp(mf_sd, R.CIFAR10_MEANFLOW, x, t, None, v, torch.ones_like(t), r=r, vr=torch.zeros_like(r))
    assert torch.allclose(out, fx["mf/out"], rtol=1e-4, atol=1e-5)
    assert float((jv - fx["mf/jvp"]).norm() / fx["mf/jvp"].norm()) <= 1e-4
    t = fx["edm/t"]
    x = torch.randn((2, 3, 32, 32), generator=torch.Generator().manual_seed(21)) * t.reshape(2, 1, 1, 1)
    out, jv = R.edm_precond_jvp(full_sd, R.CIFAR10, x, t, fx["edm/cond"], v, fx["edm/vt"])
    assert torch.allclose(out, fx["edm/out"], rtol=1e-4, atol=1e-5)
    assert float((jv - fx["edm/jvp"]).norm() / fx["edm/jvp"].norm()) <= 1e-4


def test_oracle_augmentation_labels(golden_dir, full_sd):
    fx = torch.load(os.path.join(golden_dir, "augment_b2.pt"), weights_only=True)
    x = torch.randn((2, 3, 32, 32), generator=torch.Generator().manual_seed(91)) * fx["t"].reshape(2, 1, 1, 1).float()
    aug = torch.randn((2, 9), generator=torch.Generator().manual_seed(92))
    out = R.edm_precond_forward(full_sd, R.CIFAR10, x, fx["t"], fx["cond"], augment_labels=aug)
    assert torch.allclose(out, fx["out"], rtol=1e-4, atol=1e-5)


def test_oracle_training_mode_dropout(golden_dir, full_sd):
    """Where the training-mode dropout sits and how it scales (EDM/network.py:283-284): the oracle with explicit keep factors
    against the reference in train() mode whose F.dropout was replaced by the same seeded factors (tests/golden/dropout_b2.pt)."""
    fx = torch.load(os.path.join(golden_dir, "dropout_b2.pt"), weights_only=True)
    p = float(fx["p"])
    enc, dec = R.layout(R.CIFAR10)
    blocks = [b for b in enc + dec if b.kind == "block"]
    keeps = {b.key: (torch.rand((2, b.cout, b.res, b.res), generator=torch.Generator().manual_seed(600 + i)) >= p).float() / (1 - p)
             for i, b in enumerate(blocks)}
    x = torch.randn((2, 3, 32, 32), generator=torch.Generator().manual_seed(95)) * fx["t"].reshape(2, 1, 1, 1).float()
    out = R.edm_precond_forward(full_sd, R.CIFAR10, x, fx["t"], fx["cond"], drop_keeps=keeps)
    assert torch.allclose(out, fx["out"], rtol=1e-4, atol=1e-5)
    assert not torch.allclose(R.edm_precond_forward(full_sd, R.CIFAR10, x, fx["t"], fx["cond"]), fx["out"], atol=1e-3)


def test_oracle_sigma_shift_is_an_eval_mode_term(golden_dir):
    """`sigma_shift = None if self.training else self.sigma_shift` (EDM/network.py:956): the reference with sigma_shift = 0.003 in
    eval() and train() mode (tests/golden/sigma_shift_b2.pt, recorded by oracle/gen_golden.py sigma_shift)."""
    import dataclasses

    fx = torch.load(os.path.join(golden_dir, "sigma_shift_b2.pt"), weights_only=True)
    cfg = dataclasses.replace(R.CIFAR10, sigma_shift=float(fx["sigma_shift"]))
    sd = R.random_state_dict(cfg, seed=1234)
    x = torch.randn((2, 3, 32, 32), generator=torch.Generator().manual_seed(131)) * (0.25 + fx["t"].reshape(2, 1, 1, 1).float())
    for mode in ("eval", "train"):
        out = R.edm_precond_forward(sd, cfg, x, fx["t"], fx["cond"], training=(mode == "train"))
        assert (out - fx[f"out_{mode}"]).abs().max() < 1e-4, mode
    assert (fx["out_eval"] - fx["out_train"]).abs().max() > 1e-3
