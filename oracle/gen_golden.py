"""Golden-vector generator (test infrastructure; runs ONLY in the build container).

Imports the *reference itself* from /root/reference (with inert stubs for absent third-party imports,
oracle/_ref_import.py), runs its EDM network / noise schedule / sampler on seeded inputs and writes
small fixtures to tests/golden/.  The reference never travels to the GPU box; the fixtures do.

Weights and inputs are not stored: they are regenerated from seeds by oracle/edm_ref.py
(random_state_dict, seeded torch CPU generator); every fixture stores checksums of what it was fed so a
drift of the RNG stream is detected instead of silently comparing different problems.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.pt)
        python oracle/gen_golden.py meanflow   (only the MeanFlow / rectified-flow fixtures)
        python oracle/gen_golden.py sample     (only the teacher Euler-sampler fixture)
        python oracle/gen_golden.py train_schedule   (only the training-side schedule helpers)
        python oracle/gen_golden.py dit        (only the DiT fixtures: reference DiT class on restated timm stand-ins)
        python oracle/gen_golden.py sigma_shift   (only the eval- vs train-mode sigma_shift fixture)
        python oracle/gen_golden.py backward   (only the training-step fixtures: conv weight gradients)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import edm_ref  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def checksum(t: torch.Tensor) -> torch.Tensor:
    t = t.detach().to(torch.float64).reshape(-1)
    w = torch.arange(1, t.numel() + 1, dtype=torch.float64) % 7 + 1
    return torch.stack([t.sum(), (t * w).sum(), t.abs().max()])


def sd_checksum(sd) -> torch.Tensor:
    return torch.stack([checksum(v) for _, v in sorted(sd.items())]).sum(0)


def ref_net(edm_net, cfg: edm_ref.SongUNetConfig, sd, **extra):
    net = edm_net.EDMPrecond(
        img_resolution=cfg.img_resolution, img_channels=cfg.img_channels, label_dim=cfg.label_dim,
        sigma_shift=cfg.sigma_shift, sigma_data=cfg.sigma_data, model_type="SongUNet", augment_dim=cfg.augment_dim,
        model_channels=cfg.model_channels, channel_mult=list(cfg.channel_mult), channel_mult_noise=cfg.channel_mult_noise,
        num_blocks=cfg.num_blocks, attn_resolutions=list(cfg.attn_resolutions),
        embedding_type="positional", encoder_type="standard", decoder_type="standard", resample_filter=[1, 1],
        dropout=0.0, label_dropout=0, r_timestep=cfg.r_timestep, drop_precond=cfg.drop_precond,
        schedule_type=cfg.schedule, **extra,
    )
    ref_sd = net.state_dict()
    assert set(ref_sd) == set(sd), (set(ref_sd) ^ set(sd))
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd, strict=True)
    return net.eval()


def seeded(shape, seed, dtype=torch.float32):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)).to(dtype)


def meanflow_fixtures(edm_net, ns):
    """MeanFlow student on CIFAR-10 (configs/experiments/EDM/config_mf_cifar10.py): r_timestep network without
    preconditioning on the rectified-flow schedule, sampled by MeanFlowModel._student_sample_loop."""
    from fastgen.methods import MeanFlowModel

    # ---- rectified-flow schedule ------------------------------------------------------------------------
    sched = ns.RFNoiseSchedule()
    fx = {f"t_list_{n}": sched.get_t_list(n) for n in (1, 2, 4)}
    x = seeded((2, 3, 8, 8), 11)
    e = seeded((2, 3, 8, 8), 12)
    t = torch.tensor([0.7492, 0.2497], dtype=torch.float64)
    fx["fp_out"] = sched.forward_process(x, e, t)
    fx["lat_out"] = sched.latents(x, t_init=torch.tensor(0.999, dtype=torch.float64))
    fx["x0eps_out"] = sched.x0_to_eps(x, e, t)
    fx["flow_out"] = sched.x0_to_flow(x, e, t)
    fx["max_sigma"] = torch.tensor(sched.max_sigma, dtype=torch.float64)
    torch.save(fx, os.path.join(OUT, "schedule_rf.pt"))

    # ---- full-width network: names, forward with (t, r), sampler -------------------------------------------
    cfg = edm_ref.CIFAR10_MEANFLOW
    sd = edm_ref.random_state_dict(cfg, seed=4321)
    net = ref_net(edm_net, cfg, sd, net_pred_type="flow")
    with open(os.path.join(OUT, "state_dict_keys_meanflow.txt"), "w") as f:
        for k, v in net.state_dict().items():
            f.write(f"{k} {' '.join(str(d) for d in v.shape)}\n")
    B = 2
    xin = seeded((B, 3, 32, 32), 41)
    tt = torch.tensor([0.999, 0.4995], dtype=torch.float64)
    rr = torch.tensor([0.0, 0.2497], dtype=torch.float64)
    emb = {}
    hk = net.model.map_layer1.register_forward_hook(
        lambda m, i, o: emb.__setitem__("emb", torch.nn.functional.silu(o.detach().clone())))
    with torch.inference_mode():
        out = net(xin, tt, condition=None, r=rr, fwd_pred_type="flow")
        out_x0 = net(xin, tt, condition=None, r=rr, fwd_pred_type="x0")
    hk.remove()
    # the other drop_precond settings on the same weights (each leaves one half of the preconditioning on)
    variants = {}
    for dp in (None, "input", "output"):
        cfg_v = edm_ref.SongUNetConfig(**{**cfg.__dict__, "drop_precond": dp})
        nv = ref_net(edm_net, cfg_v, sd, net_pred_type="flow")
        with torch.inference_mode():
            variants[str(dp)] = nv(xin, tt, condition=None, r=rr, fwd_pred_type="flow").clone()
    noise = seeded((B, 3, 32, 32), 5)
    eps_list = [seeded((B, 3, 32, 32), s) for s in (6, 7, 8)]
    it = iter(eps_list)
    orig_randn_like = torch.randn_like
    try:
        torch.randn_like = lambda x, **k: next(it).to(x.dtype)
        out_sde = MeanFlowModel.generator_fn(net, noise, student_sample_steps=4, student_sample_type="sde")
    finally:
        torch.randn_like = orig_randn_like
    out_ode = MeanFlowModel.generator_fn(net, noise, student_sample_steps=4, student_sample_type="ode")
    out_1 = MeanFlowModel.generator_fn(net, noise, student_sample_steps=1, student_sample_type="ode")
    out_tl = MeanFlowModel.generator_fn(net, noise, student_sample_steps=2, t_list=[0.999, 0.5, 0.0],
                                        student_sample_type="ode")  # the config's recommended 2-step list (:15-17)
    torch.save({"sd_checksum": sd_checksum(sd), "x_checksum": checksum(xin), "noise_checksum": checksum(noise),
                "t": tt, "r": rr, "emb": emb["emb"], "out": out.clone(), "out_x0": out_x0.clone(),
                "out_drop_None": variants["None"], "out_drop_input": variants["input"],
                "out_drop_output": variants["output"], "out_sde": out_sde.clone(), "out_ode": out_ode.clone(),
                "out_1step": out_1.clone(), "out_tlist": out_tl.clone()},
               os.path.join(OUT, "meanflow_full_b2.pt"))


def teacher_sample_fixture(edm_net):
    """EDMPrecond.sample (Euler sampler with classifier-free guidance, EDM/network.py:976-1026), full width, B = 2."""
    cfg = edm_ref.CIFAR10
    sd = edm_ref.random_state_dict(cfg, seed=1234)
    net = ref_net(edm_net, cfg, sd)
    noise = seeded((2, 3, 32, 32), 50)
    cond = torch.nn.functional.one_hot(torch.tensor([3, 7]), 10).float()
    neg = torch.zeros(2, 10)
    with torch.inference_mode():
        out_cfg = net.sample(noise, condition=cond, neg_condition=neg, guidance_scale=2.0, num_steps=4)
        out_plain = net.sample(noise, condition=cond, guidance_scale=None, num_steps=3)
    torch.save({"sd_checksum": sd_checksum(sd), "noise_checksum": checksum(noise), "cond": cond,
                "out_cfg": out_cfg.clone(), "out_plain": out_plain.clone()}, os.path.join(OUT, "teacher_sample_b2.pt"))


def train_schedule_fixture(ns):
    """The schedule members only the training callers read (dmd2.py:100-120, sCM.py, mean_flow.py): derivatives,
    conditional velocity, SNR maps, table look-ups, t_list stepping and the seeded timestep draws."""
    fx = {}
    x = seeded((3, 3, 4, 4), 31)
    e = seeded((3, 3, 4, 4), 32)
    for name, sched, t in (("edm", ns.EDMNoiseSchedule(), torch.tensor([63.2, 1.7, 0.0021], dtype=torch.float64)),
                           ("rf", ns.RFNoiseSchedule(), torch.tensor([0.93, 0.4, 0.0], dtype=torch.float64))):
        fx[f"{name}/t"] = t
        fx[f"{name}/rescale_t"] = sched.rescale_t(t)
        fx[f"{name}/alpha_prime"] = sched.alpha_prime(t)
        fx[f"{name}/sigma_prime"] = sched.sigma_prime(t)
        fx[f"{name}/cond_velocity"] = sched.cond_velocity(x, e, t)
        fx[f"{name}/sqrt_snr"] = sched.sqrt_snr(t)
        fx[f"{name}/sqrt_snr_to_t"] = sched.sqrt_snr_to_t(torch.tensor([0.0, 0.3, 7.5, 2e-7], dtype=torch.float32))
        probe = torch.cat([t, sched.sigmas[[0, 1, 500, 999]], (sched.sigmas[10:12].mean()).reshape(1),
                           torch.tensor([1e3, -1.0], dtype=torch.float64)])
        fx[f"{name}/closest_probe"] = probe
        fx[f"{name}/closest_idx"] = sched.closest_sigma_idx(probe)
        fx[f"{name}/closest_idx_4d"] = sched.closest_sigma_idx(probe[:3].reshape(3, 1, 1, 1))
        fx[f"{name}/sigma_idx_to_t"] = sched.sigma_idx_to_t(torch.tensor([0, 17, 999]))
        ids = torch.tensor([0, 2, 1, 3])
        fx[f"{name}/next_default"] = sched.next_in_t_list(ids, 4, None)
        fx[f"{name}/next_custom_stride2"] = sched.next_in_t_list(torch.tensor([0, 1]), 3, [0.9, 0.5, 0.2, 0.0], stride=2)
        torch.manual_seed(77)
        tt, ii = sched.sample_from_t_list(16, 4, return_ids=True)
        fx[f"{name}/sample_from_t_list"], fx[f"{name}/sample_from_t_list_ids"] = tt, ii
        torch.manual_seed(78)
        fx[f"{name}/sample_from_custom"] = sched.sample_from_t_list(8, 3, t_list=[0.9, 0.5, 0.2, 0.0])
        kinds = ("polynomial", "uniform", "lognormal") if name == "edm" else ("logitnormal", "uniform", "shifted")
        for k in kinds:
            torch.manual_seed(79)
            fx[f"{name}/sample_t_{k}"] = sched.sample_t(32, time_dist_type=k)
        torch.manual_seed(80)
        fx[f"{name}/sample_t_bounded"] = sched.sample_t(32, time_dist_type="uniform", min_t=0.0001, max_t=0.5)
        fx[f"{name}/safe_clamp_f32"] = sched.safe_clamp(torch.tensor([-1.0, 0.4, 90.0], dtype=torch.float32), 0.002, 0.999)
        fx[f"{name}/safe_clamp_bf16"] = sched.safe_clamp(torch.tensor([-1.0, 0.4, 90.0], dtype=torch.bfloat16), 0.002, 0.999)
    torch.save(fx, os.path.join(OUT, "schedule_train.pt"))


def backward_fixture(edm_net):
    """Training-step pieces, recorded from the reference's own modules under autograd: the weight gradient
    of `Conv2d` (EDM/network.py:54-126) for a 3x3 and a 1x1 kernel."""
    fx = {}
    for ks, cin, cout, res in ((3, 64, 128, 16), (1, 128, 128, 8)):
        conv = edm_net.Conv2d(in_channels=cin, out_channels=cout, kernel=ks)
        with torch.no_grad():
            conv.weight.copy_(seeded(tuple(conv.weight.shape), 51 + ks) * 0.05)
            conv.bias.copy_(seeded(tuple(conv.bias.shape), 52 + ks) * 0.1)
        x = seeded((2, cin, res, res), 53 + ks).requires_grad_(True)
        dy = seeded((2, cout, res, res), 54 + ks)
        conv(x).backward(dy)
        fx[f"k{ks}/weight_grad"] = conv.weight.grad.clone()
        fx[f"k{ks}/bias_grad"] = conv.bias.grad.clone()
        fx[f"k{ks}/input_grad"] = x.grad.clone()
        fx[f"k{ks}/shape"] = torch.tensor([2, cin, cout, res, ks])
    torch.save(fx, os.path.join(OUT, "backward_conv.pt"))


BWD_CASES = {
    "enc_first": ("32x32_block0", 1),  # 128 -> 256, 1x1 skip
    "enc_plain": ("8x8_block1", 2),  # 256 -> 256 @ 8x8, identity skip
    "dec_cat512": ("16x16_block1_dec", 1),  # 512 -> 256 (concat), 1x1 skip
    "dec_cat384": ("32x32_block4_dec", 1),  # 384 -> 256 (12 channels per group), 1x1 skip
    "enc_down": ("16x16_down", 1),  # 32 -> 16: 2x2 average in conv0 and in the skip conv
    "dec_up": ("16x16_up_dec", 2),  # 8 -> 16: nearest up-sampling in conv0 and in the skip conv
    "enc_attn": ("16x16_block1", 2),  # 256 -> 256 + attention, T = 256
    "dec_in0": ("8x8_in0_dec", 2),  # attention, T = 64
    "dec_cat_attn": ("16x16_block4_dec", 1),  # 512 -> 256 (concat) + attention
}


def summarise(t: torch.Tensor):
    """L2 norm + a strided sample (<= 4096 entries): enough to pin a gradient tensor without storing megabytes."""
    t = t.detach().reshape(-1)
    return {"norm": t.double().norm().float(), "sample": t[:: max(1, t.numel() // 4096)][:4096].clone()}


def block_backward_fixtures(edm_net):
    """UNetBlock backward recorded from the reference's modules under autograd: d/dx, d/demb and every
    parameter gradient for four block variants without attention / resampling, full width, seeded operands."""
    cfg = edm_ref.CIFAR10
    sd = edm_ref.random_state_dict(cfg, seed=1234)
    net = ref_net(edm_net, cfg, sd)
    enc, dec = edm_ref.layout(cfg)
    spec = {b.key.split(".")[-1] + ("_dec" if ".dec." in b.key else ""): b for b in enc + dec}
    fx = {"sd_checksum": sd_checksum(sd)}
    for i, (cname, (bname, bs)) in enumerate(BWD_CASES.items()):
        b = spec[bname]
        mod = (net.model.enc if ".enc." in b.key else net.model.dec)[b.key.split(".")[-1]]
        for p in mod.parameters():
            p.requires_grad_(True)
            p.grad = None
        rin = b.res * 2 if b.down else (b.res // 2 if b.up else b.res)
        x = seeded((bs, b.cin, rin, rin), 300 + i).requires_grad_(True)
        emb = (seeded((bs, 512), 320 + i) * 0.5).requires_grad_(True)
        dout = seeded((bs, b.cout, b.res, b.res), 340 + i)
        mod(x, emb).backward(dout)
        grads = {"dx": x.grad, "demb": emb.grad}
        grads.update({n: p.grad for n, p in mod.named_parameters()})
        # own restatement under autograd must agree already here
        sdg = {k: v.clone().requires_grad_(k.startswith(b.key + ".")) for k, v in sd.items()}
        xo = x.detach().clone().requires_grad_(True)
        eo = emb.detach().clone().requires_grad_(True)
        edm_ref.unet_block(sdg, b, xo, eo).backward(dout)
        assert torch.allclose(xo.grad, x.grad, rtol=1e-4, atol=1e-5), (cname, "dx")
        assert torch.allclose(eo.grad, emb.grad, rtol=1e-4, atol=1e-4), (cname, "demb")
        for n, p in mod.named_parameters():
            assert torch.allclose(sdg[f"{b.key}.{n}"].grad, p.grad, rtol=1e-4, atol=1e-4 * float(p.grad.abs().max())), (cname, n)
        fx[f"{cname}/key"] = b.key
        fx[f"{cname}/bs"] = torch.tensor(bs)
        fx[f"{cname}/seeds"] = torch.tensor([300 + i, 320 + i, 340 + i])
        for n, g in grads.items():
            sm = summarise(g)
            fx[f"{cname}/{n}/norm"], fx[f"{cname}/{n}/sample"] = sm["norm"], sm["sample"]
    torch.save(fx, os.path.join(OUT, "blocks_backward.pt"))


def full_backward_fixture(edm_net):
    """Whole-network backward: the reference's EDMPrecond under autograd, full width, B = 2, seeded operands;
    norm and a 512-entry strided sample of every parameter gradient (parameters the forward does not use have none)."""
    cfg = edm_ref.CIFAR10
    sd = edm_ref.random_state_dict(cfg, seed=1234)
    net = ref_net(edm_net, cfg, sd)
    for p in net.parameters():
        p.requires_grad_(True)
        p.grad = None
    tt = torch.tensor([17.4981, 0.1726], dtype=torch.float64)
    x = seeded((2, 3, 32, 32), 21) * tt.reshape(2, 1, 1, 1).float()
    cond = torch.nn.functional.one_hot(torch.tensor([3, 7]), 10).float()
    dout = seeded((2, 3, 32, 32), 401)
    out = net(x, tt, condition=cond, fwd_pred_type="x0")
    out.backward(dout)
    # own restatement under autograd
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "resample_filter" not in k) for k, v in sd.items()}
    edm_ref.edm_precond_forward(sdg, cfg, x, tt, cond).backward(dout)
    fx = {"sd_checksum": sd_checksum(sd), "out": out.detach().clone(), "t": tt, "cond": cond}
    names = []
    for n, p in net.named_parameters():
        if p.grad is None:
            assert sdg[n].grad is None or float(sdg[n].grad.abs().max()) == 0.0, n
            continue
        assert torch.allclose(sdg[n].grad, p.grad, rtol=1e-3, atol=1e-4 * float(p.grad.abs().max())), n
        g = p.grad.reshape(-1)
        fx[f"{n}/norm"] = g.double().norm().float()
        fx[f"{n}/sample"] = g[:: max(1, g.numel() // 512)][:512].clone()
        names.append(n)
    with open(os.path.join(OUT, "full_backward_names.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    # ---- the gradient paths of DMD2's GAN branch (dmd2.py:137-146): feature taps and the network input ---------------------
    probe = ["model.enc.32x32_block3.conv1.weight", "model.enc.16x16_block0.norm0.weight", "model.enc.8x8_block3.skip.weight"
             if "model.enc.8x8_block3.skip.weight" in dict(net.named_parameters()) else "model.enc.8x8_block3.conv0.weight",
             "model.map_layer0.weight", "model.enc.32x32_conv.weight"]

    def clear():
        for p in net.parameters():
            p.grad = None

    def shapes(i):
        return [(2, 256, 32, 32), (2, 256, 16, 16), (2, 256, 8, 8)][i]

    # (a) d out / d x_t
    clear()
    xg = x.clone().requires_grad_(True)
    net(xg, tt, condition=cond, fwd_pred_type="x0").backward(dout)
    fx["gan/dx_out"] = xg.grad.clone()
    # (b) taps returned early, all three, gradients into x_t and the encoder
    clear()
    xg = x.clone().requires_grad_(True)
    feats = net(xg, tt, condition=cond, return_features_early=True, feature_indices={0, 1, 2})
    dfs = [seeded(shapes(i), 410 + i) for i in range(3)]
    torch.autograd.backward(feats, dfs)
    fx["gan/dx_early"] = xg.grad.clone()
    for n in probe:
        g = dict(net.named_parameters())[n].grad.reshape(-1)
        fx[f"gan/early/{n}/norm"] = g.double().norm().float()
        fx[f"gan/early/{n}/sample"] = g[:: max(1, g.numel() // 512)][:512].clone()
    assert dict(net.named_parameters())["model.dec.8x8_in0.conv0.weight"].grad is None
    for i, f in enumerate(feats):
        fx[f"gan/feat{i}/sample"] = f.detach().reshape(-1)[:: f.numel() // 512][:512].clone()
    # (c) prediction and the bottleneck tap together
    clear()
    xg = x.clone().requires_grad_(True)
    o, fe = net(xg, tt, condition=cond, feature_indices={2}, fwd_pred_type="x0")
    torch.autograd.backward([o, fe[0]], [dout, dfs[2]])
    fx["gan/dx_both"] = xg.grad.clone()
    g = dict(net.named_parameters())[probe[0]].grad.reshape(-1)
    fx["gan/both/probe0/sample"] = g[:: max(1, g.numel() // 512)][:512].clone()
    fx["gan/probe_names"] = probe
    torch.save(fx, os.path.join(OUT, "full_backward_b2.pt"))


def meanflow_backward_fixture(edm_net):
    """Backward of the MeanFlow CIFAR-10 network (r_timestep embedding, no preconditioning, unconditional, flow prediction;
    configs/experiments/EDM/config_mf_cifar10.py) under autograd: what its finite-difference training mode differentiates
    (mean_flow.py:162-218).  Norm + 512-entry sample of every parameter gradient and the input gradient, B = 2."""
    cfg = edm_ref.CIFAR10_MEANFLOW
    sd = edm_ref.random_state_dict(cfg, seed=4321)
    net = ref_net(edm_net, cfg, sd)
    for p in net.parameters():
        p.requires_grad_(True)
        p.grad = None
    tt = torch.tensor([0.83, 0.31], dtype=torch.float64)
    rr = torch.tensor([0.40, 0.0], dtype=torch.float64)
    x = seeded((2, 3, 32, 32), 61).requires_grad_(True)
    dout = seeded((2, 3, 32, 32), 62)
    out = net(x, tt, r=rr)
    out.backward(dout)
    fx = {"sd_checksum": sd_checksum(sd), "out": out.detach().clone(), "t": tt, "r": rr, "dx": x.grad.clone()}
    names = []
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.reshape(-1)
        fx[f"{n}/norm"] = g.double().norm().float()
        fx[f"{n}/sample"] = g[:: max(1, g.numel() // 512)][:512].clone()
        names.append(n)
    fx["names"] = names
    torch.save(fx, os.path.join(OUT, "meanflow_backward_b2.pt"))


def discriminator_fixture():
    """`Discriminator_EDM` (fastgen/networks/discriminators.py:62-137) recorded from the reference: logits, the gradient with
    respect to the feature maps and every parameter gradient (norm + 512-entry sample), for the default bottleneck head and
    for all three heads; weights from the reference's own default init under a fixed seed, stored as a seed (regenerated by the
    same constructor call in the test would need the reference - so the state dict itself is stored for the small default head
    and as a seed recipe (N(0, 0.02) re-randomisation) for the three-head case)."""
    import importlib

    disc_mod = importlib.import_module("fastgen.networks.discriminators")
    fx = {}
    for tag, idx, bs in (("default", None, 4), ("all", {0, 1, 2}, 2)):
        torch.manual_seed(500)
        d = disc_mod.Discriminator_EDM(feature_indices=idx)
        # deterministic, seed-reproducible parameters: every tensor ~ N(0, s), gains around 1
        g = torch.Generator().manual_seed(501)
        with torch.no_grad():
            for n, p in d.named_parameters():
                if p.ndim == 1 and n.endswith("weight"):
                    p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
                elif p.ndim == 1:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(torch.randn(p.shape, generator=g) / (p[0].numel() ** 0.5))
        feats = [seeded((bs, 256, r, r), 510 + r).requires_grad_(True) for r in d.in_res]
        logits = d(feats)
        dl = seeded(tuple(logits.shape), 520)
        logits.backward(dl)
        fx[f"{tag}/keys"] = list(d.state_dict().keys())
        fx[f"{tag}/in_res"] = torch.tensor(d.in_res)
        fx[f"{tag}/bs"] = torch.tensor(bs)
        fx[f"{tag}/logits"] = logits.detach().clone()
        for r, f in zip(d.in_res, feats):
            sm = summarise(f.grad)
            fx[f"{tag}/dfeat{r}/norm"], fx[f"{tag}/dfeat{r}/sample"] = sm["norm"], sm["sample"]
        for n, p in d.named_parameters():
            sm = summarise(p.grad)
            fx[f"{tag}/{n}/norm"], fx[f"{tag}/{n}/sample"] = sm["norm"], sm["sample"][:512].clone()
    torch.save(fx, os.path.join(OUT, "discriminator_edm.pt"))


def jvp_fixture(edm_net):
    """Forward-mode derivative of the network, recorded with the reference's own call
    `torch.func.jvp(net_wrapper, (x_t, t, r), tangents)` (consistency_model/mean_flow.py:240-250): the MeanFlow CIFAR-10 network
    with tangents (v, 1, 0), and the preconditioned DMD2 network with tangents (v, vt)."""
    fx = {}
    cfg = edm_ref.CIFAR10_MEANFLOW
    sd = edm_ref.random_state_dict(cfg, seed=4321)
    net = ref_net(edm_net, cfg, sd)
    x, v = seeded((2, 3, 32, 32), 71), seeded((2, 3, 32, 32), 72)
    t, r = torch.tensor([0.83, 0.31]), torch.tensor([0.40, 0.0])
    with torch.no_grad():
        out, jv = torch.func.jvp(lambda a, b, c: net(a, b, r=c), (x, t, r), (v, torch.ones_like(t), torch.zeros_like(r)))
    oo, oj = edm_ref.edm_precond_jvp(sd, cfg, x, t, None, v, torch.ones_like(t), r=r, vr=torch.zeros_like(r))
    assert torch.allclose(oo, out, rtol=1e-4, atol=1e-5) and torch.allclose(oj, jv, rtol=1e-3, atol=1e-4 * float(jv.abs().max()))
    fx.update({"mf/out": out.clone(), "mf/jvp": jv.clone(), "mf/t": t, "mf/r": r})
    cfg = edm_ref.CIFAR10
    sd = edm_ref.random_state_dict(cfg, seed=1234)
    net = ref_net(edm_net, cfg, sd)
    t = torch.tensor([17.4981, 0.1726])
    vt = torch.tensor([1.0, -0.05])
    x = seeded((2, 3, 32, 32), 21) * t.reshape(2, 1, 1, 1)
    cond = torch.nn.functional.one_hot(torch.tensor([3, 7]), 10).float()
    with torch.no_grad():
        out, jv = torch.func.jvp(lambda a, b: net(a, b, condition=cond, fwd_pred_type="x0"), (x, t), (v, vt))
    oo, oj = edm_ref.edm_precond_jvp(sd, cfg, x, t, cond, v, vt)
    assert torch.allclose(oo, out, rtol=1e-4, atol=1e-5) and torch.allclose(oj, jv, rtol=1e-3, atol=1e-4 * float(jv.abs().max()))
    fx.update({"edm/out": out.clone(), "edm/jvp": jv.clone(), "edm/t": t, "edm/vt": vt, "edm/cond": cond})
    # near the data end, where c_out -> 0 (the raw network output must not be recovered by dividing by it)
    t = torch.tensor([0.004, 0.05])
    vt = torch.tensor([0.001, -0.01])
    x = seeded((2, 3, 32, 32), 73) * 0.5
    with torch.no_grad():
        out, jv = torch.func.jvp(lambda a, b: net(a, b, condition=cond, fwd_pred_type="x0"), (x, t), (v, vt))
    fx.update({"edm_small/out": out.clone(), "edm_small/jvp": jv.clone(), "edm_small/t": t, "edm_small/vt": vt})
    torch.save(fx, os.path.join(OUT, "jvp_b2.pt"))


def trigflow_fixture(edm_net, ns):
    """sCM-family network interface (consistency_model/sCM.py:21-83, 150-181): `TrigNoiseSchedule` members and `TrigFlowPrecond`
    around the CIFAR-10 denoiser - forward and `torch.func.jvp` along seeded tangents, B = 2."""
    import importlib

    scm = importlib.import_module("fastgen.methods.consistency_model.sCM")
    fx = {}
    sched = ns.TrigNoiseSchedule()
    t = torch.tensor([0.2, 0.9, 1.5], dtype=torch.float64)
    x, e = seeded((3, 3, 4, 4), 81), seeded((3, 3, 4, 4), 82)
    fx["sched/t"] = t
    fx["sched/sqrt_snr"] = sched.sqrt_snr(t)
    fx["sched/sqrt_snr_to_t"] = sched.sqrt_snr_to_t(torch.tensor([0.0, 0.4, 3.0, 1e3], dtype=torch.float32))
    fx["sched/forward_process"] = sched.forward_process(x, e, t)
    fx["sched/x0_to_flow"] = sched.x0_to_flow(x, e, t)
    fx["sched/flow_to_x0"] = sched.flow_to_x0(x, e, t)
    fx["sched/sigma_idx_to_t"] = sched.sigma_idx_to_t(torch.tensor([0, 17, 999]))
    fx["sched/max_sigma"] = torch.tensor(sched.max_sigma, dtype=torch.float64)
    for k in ("uniform", "logitnormal"):
        torch.manual_seed(83)
        fx[f"sched/sample_t_{k}"] = sched.sample_t(16, time_dist_type=k)
    cfg = edm_ref.CIFAR10
    sd = edm_ref.random_state_dict(cfg, seed=1234)
    net = ref_net(edm_net, cfg, sd)
    wrap = scm.TrigFlowPrecond(net, sigma_data=0.5)
    t_hat = torch.tensor([0.35, 1.25])
    xh = seeded((2, 3, 32, 32), 84) * 0.5
    cond = torch.nn.functional.one_hot(torch.tensor([3, 7]), 10).float()
    vx, vt = seeded((2, 3, 32, 32), 85), torch.tensor([0.16, 0.14])
    with torch.no_grad():
        F0 = wrap(xh, t_hat, condition=cond)
        F1, dF = torch.func.jvp(lambda a, b: wrap(a, b.clamp(min=-torch.pi / 2 + 1e-4, max=torch.pi / 2 - 1e-4), condition=cond),
                                (xh, t_hat), (vx, vt))
        x_t, tt = wrap._convert_trigflow_to_net_input(xh, t_hat)
    fx.update({"wrap/t_hat": t_hat, "wrap/vt": vt, "wrap/cond": cond, "wrap/F": F0.clone(), "wrap/dF": dF.clone(), "wrap/x_t": x_t.clone(),
               "wrap/t": tt.clone()})
    assert torch.allclose(F0, F1)
    torch.save(fx, os.path.join(OUT, "trigflow_b2.pt"))


def augment_fixture(edm_net):
    """Training-time augmentation labels (EDM/network.py:495, 518-519, 903-915): forward and autograd of the reference with
    condition = {"aug_condition", "orig_condition"}; output, map_augment's gradient and two other parameter gradients, B = 2."""
    cfg = edm_ref.CIFAR10
    sd = edm_ref.random_state_dict(cfg, seed=1234)
    net = ref_net(edm_net, cfg, sd)
    for p in net.parameters():
        p.requires_grad_(True)
        p.grad = None
    tt = torch.tensor([3.3, 0.45], dtype=torch.float64)
    x = seeded((2, 3, 32, 32), 91) * tt.reshape(2, 1, 1, 1).float()
    cond = torch.nn.functional.one_hot(torch.tensor([1, 8]), 10).float()
    aug = seeded((2, cfg.augment_dim), 92)
    dout = seeded((2, 3, 32, 32), 93)
    out = net(x, tt, condition={"aug_condition": aug, "orig_condition": cond}, fwd_pred_type="x0")
    out.backward(dout)
    oo = edm_ref.edm_precond_forward(sd, cfg, x, tt, cond, augment_labels=aug)
    assert torch.allclose(oo, out, rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        plain = net(x, tt, condition=cond, fwd_pred_type="x0")
    assert not torch.allclose(plain, out)  # the labels do change the result
    ps = dict(net.named_parameters())
    fx = {"t": tt, "cond": cond, "out": out.detach().clone(), "map_augment_grad": ps["model.map_augment.weight"].grad.clone()}
    for n in ("model.map_layer0.weight", "model.dec.32x32_block2.conv1.weight"):
        g = ps[n].grad.reshape(-1)
        fx[f"{n}/sample"] = g[:: max(1, g.numel() // 512)][:512].clone()
    torch.save(fx, os.path.join(OUT, "augment_b2.pt"))


def dit_fixture():
    """DiT: the reference's own `DiT` class (fastgen/networks/DiT/network.py) built on restated stand-ins of the
    three timm classes it imports (oracle/_timm_restated.py: timm is un-vendored and absent here - parity 'restated' for those
    three, reference code for the rest), seeded re-randomised weights (oracle/dit_ref.random_state_dict).  Recorded: state-dict
    names / shapes, and for DiT-XL/2 (configs/net.py:124-127: hidden 1152, depth 28, 16 heads of 72) and DiT-S/2 (384, 12, 6 x 64)
    a forward at B = 2 - output, conditioning vector, strided samples of every block output; XL also with the r embedding."""
    import _timm_restated
    import dit_ref

    _ref_import.install_stubs()
    _timm_restated.install()
    if _ref_import.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, _ref_import.REFERENCE_ROOT)
    import fastgen.networks.DiT.network as dit_net

    fx = {}
    for tag, cfg in (("xl", dit_ref.XL_2), ("s", dit_ref.S_2), ("xl_r", dit_ref.DiTConfig(r_timestep=True))):
        sd = dit_ref.random_state_dict(cfg, seed=77)
        net = dit_net.DiT(input_size=cfg.input_size, patch_size=cfg.patch_size, in_channels=cfg.in_channels, hidden_size=cfg.hidden_size,
                          depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, class_dropout_prob=cfg.class_dropout_prob,
                          enable_class_dropout=False, num_classes=cfg.num_classes, learn_sigma=False, r_timestep=cfg.r_timestep,
                          scale_t=cfg.scale_t)
        ref_sd = net.state_dict()
        assert list(ref_sd.keys()) == list(sd.keys()), (set(ref_sd) ^ set(sd))
        assert torch.allclose(ref_sd["pos_embed"], sd["pos_embed"], atol=1e-6)  # the restated 2-D sinusoidal table
        if tag != "xl_r":
            with open(os.path.join(OUT, f"dit_{tag}_state_dict_keys.txt"), "w") as f:
                for k, v in ref_sd.items():
                    f.write(f"{k} {' '.join(str(d) for d in v.shape)}\n")
        net.load_state_dict(sd, strict=True)
        net.eval()
        B = 2
        x = seeded((B, 4, 32, 32), 501)
        t = torch.tensor([0.731, 0.094], dtype=torch.float64)
        r = torch.tensor([0.352, 0.0], dtype=torch.float64) if cfg.r_timestep else None
        cond = torch.zeros(B, cfg.num_classes)
        cond[0, 417] = 1.0  # row 1 stays all-zero: the unconditional class (DiT/network.py:493-498)
        trace, hooks = {}, []
        for i, blk in enumerate(net.blocks):
            hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: trace.__setitem__(i, o.detach().clone())))
        with torch.inference_mode():
            out = net(x, t, condition=cond, r=r)
        for h in hooks:
            h.remove()
        tr = {}
        oo = dit_ref.dit_forward(sd, cfg, x, t, cond, r=r, trace=tr)
        assert torch.allclose(oo, out, rtol=1e-4, atol=2e-5), float((oo - out).abs().max())
        fx.update({f"{tag}/out": out.clone(), f"{tag}/t": t, f"{tag}/cond_class": torch.tensor([417, -1]), f"{tag}/c": tr["c"].clone(),
                   f"{tag}/sd_checksum": sd_checksum(sd), f"{tag}/x_checksum": checksum(x)})
        if r is not None:
            fx[f"{tag}/r"] = r
        for i in range(cfg.depth):
            v = trace[i].reshape(-1)
            fx[f"{tag}/block{i}/sample"] = v[:: max(1, v.numel() // 1024)][:1024].clone()
            fx[f"{tag}/block{i}/norm"] = v.double().norm().float()
    torch.save(fx, os.path.join(OUT, "dit_forward_b2.pt"))


def sigma_shift_fixture(edm_net):
    """sigma_shift is applied in eval mode only (EDM/network.py:956: `None if self.training else self.sigma_shift`): the
    reference with sigma_shift = 0.003 (the value suggested in its consistency-model configs) in eval() and in train() mode
    (dropout 0, so train() is deterministic), output and the gradient of <out, dout> with respect to x_t, B = 2, timesteps
    small enough for the shift to matter."""
    import dataclasses
    cfg = dataclasses.replace(edm_ref.CIFAR10, sigma_shift=0.003)
    sd = edm_ref.random_state_dict(cfg, seed=1234)
    net = ref_net(edm_net, cfg, sd)
    tt = torch.tensor([0.5, 0.0045], dtype=torch.float64)
    xin = seeded((2, 3, 32, 32), 131)
    cond = torch.nn.functional.one_hot(torch.tensor([2, 9]), 10).float()
    dout = seeded((2, 3, 32, 32), 132)
    fx = {"t": tt, "cond": cond, "sigma_shift": torch.tensor(cfg.sigma_shift)}
    for mode in ("eval", "train"):
        net.train(mode == "train")
        x = (xin * (0.25 + tt.reshape(2, 1, 1, 1).float())).requires_grad_(True)
        out = net(x, tt, condition=cond, fwd_pred_type="x0")
        (dx,) = torch.autograd.grad((out * dout).sum(), x)
        oo = edm_ref.edm_precond_forward(sd, cfg, x.detach(), tt, cond, training=(mode == "train"))
        assert torch.allclose(oo, out, rtol=1e-4, atol=1e-5), mode
        fx[f"out_{mode}"], fx[f"dx_{mode}"] = out.detach().clone(), dx.clone()
    assert (fx["out_eval"] - fx["out_train"]).abs().max() > 1e-3  # the shift does change the result at these timesteps
    torch.save(fx, os.path.join(OUT, "sigma_shift_b2.pt"))


def dropout_fixture(edm_net):
    """Training-mode dropout (EDM/network.py:283-284; the SFT config trains with p = 0.13): the reference in train() mode with
    `torch.nn.functional.dropout` replaced by a deterministic stand-in that multiplies by explicit keep factors drawn from seeded
    generators (call i uses seed 600 + i), so that WHERE the dropout sits and how it scales are pinned; the oracle given the same
    factors must reproduce output and gradients."""
    import torch.nn.functional as TF

    p_drop = 0.13
    cfg = edm_ref.CIFAR10
    sd = edm_ref.random_state_dict(cfg, seed=1234)
    net = edm_net.EDMPrecond(
        img_resolution=32, img_channels=3, label_dim=10, sigma_shift=0.0, sigma_data=0.5, model_type="SongUNet", augment_dim=9,
        model_channels=128, channel_mult=[2, 2, 2], channel_mult_noise=1, num_blocks=4, attn_resolutions=[16],
        embedding_type="positional", encoder_type="standard", decoder_type="standard", resample_filter=[1, 1], dropout=p_drop,
        label_dropout=0, r_timestep=False, drop_precond=None)
    net.load_state_dict(sd, strict=True)
    net.train()
    calls = []
    orig = TF.dropout

    def fake_dropout(x, p=0.5, training=True, inplace=False):
        assert training and abs(p - p_drop) < 1e-12
        i = len(calls)
        keep = (torch.rand(x.shape, generator=torch.Generator().manual_seed(600 + i)) >= p).to(x.dtype) / (1 - p)
        calls.append(tuple(x.shape))
        return x * keep

    tt = torch.tensor([2.2, 0.31], dtype=torch.float64)
    x = seeded((2, 3, 32, 32), 95) * tt.reshape(2, 1, 1, 1).float()
    cond = torch.nn.functional.one_hot(torch.tensor([0, 5]), 10).float()
    dout = seeded((2, 3, 32, 32), 96)
    for p in net.parameters():
        p.requires_grad_(True)
    TF.dropout = fake_dropout
    torch.nn.functional.dropout = fake_dropout
    try:
        out = net(x, tt, condition=cond, fwd_pred_type="x0")
        out.backward(dout)
    finally:
        TF.dropout = orig
        torch.nn.functional.dropout = orig
    enc, dec = edm_ref.layout(cfg)
    blocks = [b for b in enc + dec if b.kind == "block"]
    assert len(calls) == len(blocks) == 33, len(calls)   # one call per UNetBlock, in execution order
    keeps = {}
    for i, b in enumerate(blocks):
        assert calls[i] == (2, b.cout, b.res, b.res)
        keeps[b.key] = (torch.rand(calls[i], generator=torch.Generator().manual_seed(600 + i)) >= p_drop).float() / (1 - p_drop)
    oo = edm_ref.edm_precond_forward(sd, cfg, x, tt, cond, drop_keeps=keeps)
    assert torch.allclose(oo, out, rtol=1e-4, atol=1e-5), float((oo - out).abs().max())
    ps = dict(net.named_parameters())
    fx = {"p": torch.tensor(p_drop), "t": tt, "cond": cond, "out": out.detach().clone()}
    for n in ("model.enc.32x32_block1.conv1.weight", "model.enc.16x16_block2.norm1.weight", "model.dec.8x8_block1.conv0.weight"):
        g = ps[n].grad.reshape(-1)
        fx[f"{n}/sample"] = g[:: max(1, g.numel() // 512)][:512].clone()
    torch.save(fx, os.path.join(OUT, "dropout_b2.pt"))


def main():
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["dit"]:
        dit_fixture()
        print("DiT fixtures written to", OUT)
        return
    edm_net, ns, model = _ref_import.import_reference()
    torch.manual_seed(0)
    if sys.argv[1:] == ["augment"]:
        augment_fixture(edm_net)
        dropout_fixture(edm_net)
        print("augment fixture written to", OUT)
        return
    if sys.argv[1:] == ["sigma_shift"]:
        sigma_shift_fixture(edm_net)
        print("sigma_shift fixture written to", OUT)
        return
    if sys.argv[1:] == ["jvp"]:
        jvp_fixture(edm_net)
        trigflow_fixture(edm_net, ns)
        print("jvp fixture written to", OUT)
        return
    if sys.argv[1:] == ["backward"]:
        backward_fixture(edm_net)
        block_backward_fixtures(edm_net)
        full_backward_fixture(edm_net)
        meanflow_backward_fixture(edm_net)
        discriminator_fixture()
        print("backward fixtures written to", OUT)
        return
    if sys.argv[1:] == ["train_schedule"]:
        train_schedule_fixture(ns)
        print("training-side schedule fixture written to", OUT)
        return
    if sys.argv[1:] == ["sample"]:
        teacher_sample_fixture(edm_net)
        print("teacher sample fixture written to", OUT)
        return
    if sys.argv[1:] == ["meanflow"]:
        meanflow_fixtures(edm_net, ns)
        print("MeanFlow fixtures written to", OUT)
        return

    # ---- (i)+(ii) schedule ------------------------------------------------------------------
    sched = ns.EDMNoiseSchedule()
    fx = {"sigmas_head": sched.sigmas[:3].clone(), "sigmas_tail": sched.sigmas[-3:].clone()}
    for n in (1, 2, 4):
        fx[f"t_list_{n}"] = sched.get_t_list(n)
    x = seeded((2, 3, 8, 8), 11)
    e = seeded((2, 3, 8, 8), 12)
    t = torch.tensor([17.498123, 0.1726], dtype=torch.float64)
    fx["fp_out"] = sched.forward_process(x, e, t)
    fx["lat_out"] = sched.latents(x, t_init=torch.tensor(79.5638, dtype=torch.float64))
    fx["x0eps_out"] = sched.x0_to_eps(x, e, t)
    torch.save(fx, os.path.join(OUT, "schedule.pt"))

    # ---- names / shapes of the full CIFAR-10 network --------------------------------------------
    cfg = edm_ref.CIFAR10
    sd = edm_ref.random_state_dict(cfg, seed=1234)
    net = ref_net(edm_net, cfg, sd)
    with open(os.path.join(OUT, "state_dict_keys.txt"), "w") as f:
        for k, v in net.state_dict().items():
            f.write(f"{k} {' '.join(str(d) for d in v.shape)}\n")

    # ---- (iii) embedding + (v) full forward, full width, B=2 ---------------------------------------
    B = 2
    xin = seeded((B, 3, 32, 32), 21)
    cond = torch.nn.functional.one_hot(torch.tensor([3, 7]), 10).float()
    tt = torch.tensor([17.4981, 0.1726], dtype=torch.float64)
    trace = {}
    hooks = []
    named = [("enc." + n, m) for n, m in net.model.enc.items()] + [("dec." + n, m) for n, m in net.model.dec.items()]
    for name, blk in named:
        hooks.append(blk.register_forward_hook(lambda m, i, o, name=name: trace.__setitem__(name, o.detach().clone())))
    hooks.append(net.model.map_layer1.register_forward_hook(
        lambda m, i, o: trace.__setitem__("emb", torch.nn.functional.silu(o.detach().clone()))))
    with torch.inference_mode():
        out = net(xin * tt.reshape(B, 1, 1, 1).float(), tt, condition=cond, fwd_pred_type="x0")
    for h in hooks:
        h.remove()
    fx = {"sd_checksum": sd_checksum(sd), "x_checksum": checksum(xin), "out": out.clone(), "emb": trace["emb"],
          "t": tt, "cond": cond}
    # strided samples + moments of every block output (full tensors would be ~60 MB)
    for name, v in trace.items():
        if name == "emb":
            continue
        fx[f"blk/{name}/moments"] = torch.stack([v.double().mean(), v.double().std(), v.double().abs().max()])
        fx[f"blk/{name}/sample"] = v.reshape(-1)[:: max(1, v.numel() // 4096)][:4096].clone()
    torch.save(fx, os.path.join(OUT, "forward_full_b2.pt"))

    # ---- (iv) UNetBlock variants, full width, small batch ---------------------------------------------
    enc, dec = edm_ref.layout(cfg)
    spec = {b.key.split(".")[-1] + ("_dec" if ".dec." in b.key else ""): b for b in enc + dec}
    emb = trace["emb"]
    cases = {
        "enc_first": ("32x32_block0", 1),  # 128 -> 256, 1x1 skip
        "enc_plain": ("8x8_block1", 2),  # 256 -> 256 @ 8x8
        "enc_attn": ("16x16_block1", 2),  # 256 -> 256 + attention, T = 256
        "enc_down": ("16x16_down", 1),  # 32 -> 16 avg-pool + skip 1x1
        "dec_in0": ("8x8_in0_dec", 2),  # attention, T = 64
        "dec_up": ("16x16_up_dec", 2),  # 8 -> 16 nearest
        "dec_cat512": ("16x16_block1_dec", 1),  # 512 -> 256
        "dec_cat384": ("32x32_block4_dec", 1),  # 384 -> 256 (12 channels per group)
        "dec_cat_attn": ("16x16_block4_dec", 1),  # 512 -> 256 + attention
    }
    fxb = {"sd_checksum": sd_checksum(sd)}
    for cname, (bname, bs) in cases.items():
        b = spec[bname]
        mod = (net.model.enc if ".enc." in b.key else net.model.dec)[b.key.split(".")[-1]]
        rin = b.res * 2 if b.down else (b.res // 2 if b.up else b.res)
        xb = seeded((bs, b.cin, rin, rin), 100 + len(fxb))
        with torch.inference_mode():
            yb = mod(xb, emb[:bs])
        fxb[f"{cname}/key"] = b.key
        fxb[f"{cname}/seed"] = torch.tensor(100 + len(fxb) - 1)
        fxb[f"{cname}/x_checksum"] = checksum(xb)
        fxb[f"{cname}/out"] = yb.clone()
        # own restatement must agree already here (fails loudly at generation time)
        yo = edm_ref.unet_block(sd, b, xb, emb[:bs])
        assert torch.allclose(yo, yb, rtol=1e-5, atol=1e-5), (cname, (yo - yb).abs().max())
    fxb["emb"] = emb
    torch.save(fxb, os.path.join(OUT, "blocks_full.pt"))

    # ---- (vi) 4-step generator_fn trace with injected eps, full width, B=2 ----------------------------------
    noise = seeded((B, 3, 32, 32), 0)
    eps_list = [seeded((B, 3, 32, 32), s) for s in (1, 2, 3)]
    it = iter(eps_list)
    orig_randn_like = torch.randn_like
    preds = []
    hook = net.register_forward_hook(lambda m, i, o: preds.append(o.detach().clone()))
    try:
        torch.randn_like = lambda x, **k: next(it).to(x.dtype)
        out_sde = model.FastGenModel.generator_fn(net, noise, condition=cond, student_sample_steps=4,
                                                 student_sample_type="sde")
    finally:
        torch.randn_like = orig_randn_like
    sde_preds = [p.clone() for p in preds]
    preds.clear()
    out_ode = model.FastGenModel.generator_fn(net, noise, condition=cond, student_sample_steps=4,
                                             student_sample_type="ode")
    out_1 = model.FastGenModel.generator_fn(net, noise, condition=cond, student_sample_steps=1)
    out_tl = model.FastGenModel.generator_fn(net, noise, condition=cond, student_sample_steps=2,
                                            t_list=[40.0, 1.5, 0.0], student_sample_type="ode")
    hook.remove()
    torch.save({"sd_checksum": sd_checksum(sd), "noise_checksum": checksum(noise), "cond": cond,
                "out_sde": out_sde.clone(), "x_pred_sde": torch.stack(sde_preds), "out_ode": out_ode.clone(),
                "out_1step": out_1.clone(), "out_tlist": out_tl.clone()},
               os.path.join(OUT, "sampler_full_b2.pt"))

    # ---- reduced-width whole-net trace (oracle unit test; the reference's own test sizes, tests/test_dmd2model.py:13-44)
    small = edm_ref.SongUNetConfig(img_resolution=8, model_channels=32, channel_mult=(1, 2), num_blocks=1,
                                   attn_resolutions=(4,))
    sds = edm_ref.random_state_dict(small, seed=77)
    nets = ref_net(edm_net, small, sds)
    xs = seeded((3, 3, 8, 8), 31)
    ts = torch.tensor([80.0, 2.5, 0.002], dtype=torch.float64)
    cs = torch.nn.functional.one_hot(torch.tensor([0, 5, 9]), 10).float()
    with torch.inference_mode():
        outs = nets(xs, ts, condition=cs, fwd_pred_type="x0")
        outs_nocond = nets(xs, ts, condition=None, fwd_pred_type="x0")
    torch.save({"sd_checksum": sd_checksum(sds), "out": outs.clone(), "out_nocond": outs_nocond.clone()},
               os.path.join(OUT, "forward_small.pt"))

    meanflow_fixtures(edm_net, ns)
    train_schedule_fixture(ns)
    backward_fixture(edm_net)
    block_backward_fixtures(edm_net)
    full_backward_fixture(edm_net)
    meanflow_backward_fixture(edm_net)
    discriminator_fixture()
    jvp_fixture(edm_net)
    trigflow_fixture(edm_net, ns)
    augment_fixture(edm_net)
    dropout_fixture(edm_net)
    teacher_sample_fixture(edm_net)

    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
