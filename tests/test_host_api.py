"""CPU-only tests of the host side: the C-ABI library loads and exports what include/*.h declares, the nn.Module
drop-in reproduces the reference's state-dict, the schedule mirror matches the golden vectors, loud failures."""
import ctypes
import os
import re

import pytest
import torch

from fastgen_amd import _lib
from fastgen_amd.methods.model import FastGenModel
from fastgen_amd.networks.EDM.network import EDMPrecond
from fastgen_amd.methods.consistency_model.mean_flow import MeanFlowModel
from fastgen_amd.networks.noise_schedule import EDMNoiseSchedule, RFNoiseSchedule, get_noise_schedule
from oracle import edm_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(img_resolution=32, img_channels=3, label_dim=10, sigma_shift=0.0, sigma_data=0.5, model_type="SongUNet",
          augment_dim=9, model_channels=128, channel_mult=[2, 2, 2], channel_mult_noise=1, embedding_type="positional",
          encoder_type="standard", decoder_type="standard", resample_filter=[1, 1], dropout=0.0, label_dropout=0,
          r_timestep=False, drop_precond=None)
# MeanFlow CIFAR-10 student (configs/methods/config_mean_flow.py:137-145 + experiments/EDM/config_mf_cifar10.py)
KW_MF = {**KW, "label_dim": 0, "augment_dim": 6, "r_timestep": True, "drop_precond": "both", "schedule_type": "rf",
         "net_pred_type": "flow"}


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fastgen_amd.h")).read()
    declared = set(re.findall(r"\b(fg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.fg_version()


def test_library_exports_nothing_but_its_api():
    """Built with -fvisibility=hidden: the only functions in the dynamic symbol table are the fg_* entry points (no engine or
    launcher helper that could collide with another library inside a PyTorch process)."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    funcs = [ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in ("T", "t")]
    assert funcs and all(f.startswith("fg_") for f in funcs), [f for f in funcs if not f.startswith("fg_")]


def test_no_timing_switches_in_the_product_library():
    """The attention kernel's FASTGEN_AMD_FA_ABL switches and the token GEMM's act bits 4 / 8 / 16 skip work and compute garbage: they
    exist only in libfastgen_amd_timing.so (`make -C fastgen_amd/csrc timing`, -DFG_TIMING_BUILD), never in the product library - no
    environment variable may corrupt its outputs.  (act values other than 0 / 1 are refused: tests/test_gemm.py, on the GPU.)"""
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"FASTGEN_AMD_FA_ABL" not in blob
    assert b"act must be 0 (none) or 1 (GELU tanh)" in blob  # the refusal itself is in


def test_create_rejects_unsupported_configs():
    L = _lib.lib()
    cfg = _lib.fg_edm_config()
    cfg.img_resolution, cfg.img_channels, cfg.label_dim, cfg.model_channels = 32, 3, 10, 32  # 64-wide blocks
    cfg.num_levels, cfg.channel_mult_emb, cfg.num_blocks, cfg.channel_mult_noise = 1, 4, 1, 1
    cfg.channel_mult[0] = 2
    cfg.sigma_data = 0.5
    h = ctypes.c_void_p()
    assert L.fg_edm_create(ctypes.byref(cfg), ctypes.byref(h)) == 1
    assert b"unsupported" in L.fg_last_error()
    cfg.compute_dtype = 7
    assert L.fg_edm_create(ctypes.byref(cfg), ctypes.byref(h)) == 1
    with pytest.raises(_lib.FastGenAMDError):
        _lib.check(L.fg_edm_create(None, None))


def test_t_list_c_abi(golden_dir):
    fx = torch.load(os.path.join(golden_dir, "schedule.pt"), weights_only=True)
    for n in (1, 2, 4):
        arr = (ctypes.c_double * (n + 1))()
        _lib.check(_lib.lib().fg_edm_t_list(n, arr))
        got = torch.tensor(list(arr), dtype=torch.float64)
        assert torch.allclose(got, fx[f"t_list_{n}"], rtol=1e-14, atol=0)
        assert got[-1] == 0


@pytest.fixture(scope="module")
def net():
    return EDMPrecond(**KW)


def test_state_dict_is_the_references(net, golden_dir):
    want = {}
    for line in open(os.path.join(golden_dir, "state_dict_keys.txt")):
        p = line.split()
        want[p[0]] = tuple(int(v) for v in p[1:])
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want and list(got) == list(want)  # same names, shapes AND order
    assert sum(p.numel() for p in net.parameters()) == 55_735_428
    # a reference-format checkpoint loads strictly, and strict=False reports nothing missing/unexpected
    sd = R.random_state_dict(R.CIFAR10, seed=5)
    res = net.load_state_dict(sd, strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(net.state_dict()["model.enc.32x32_block0.conv0.weight"], sd["model.enc.32x32_block0.conv0.weight"])


def test_default_init_follows_reference_scales(net):
    m = EDMPrecond(**KW)
    sd = m.state_dict()
    assert sd["model.enc.32x32_block0.conv1.weight"].abs().max() < 1e-5 * 0.1  # init_zero: xavier * 1e-5
    assert sd["model.dec.32x32_aux_conv.weight"].abs().max() < 1e-5
    assert torch.all(sd["model.enc.32x32_block0.norm0.weight"] == 1) and torch.all(sd["model.enc.32x32_conv.bias"] == 0)
    w = sd["model.enc.32x32_block0.conv0.weight"]
    bound = (6 / (128 * 9 + 256 * 9)) ** 0.5
    assert w.abs().max() <= bound and w.abs().max() > 0.9 * bound
    assert sd["model.enc.16x16_down.conv0.resample_filter"].tolist() == [[[[0.25, 0.25], [0.25, 0.25]]]]


def test_module_surface(net):
    assert net.net_pred_type == "x0" and net.schedule_type == "edm" and net.label_dim == 10
    assert isinstance(net.noise_scheduler, EDMNoiseSchedule)
    assert net.noise_scheduler.max_t == 80.0 and net.noise_scheduler.t_precision == torch.float64
    assert len(net.noise_scheduler.state_dict()) == 0
    for attr in ("forward", "sample", "fully_shard", "reset_parameters", "few_step_sample"):
        assert callable(getattr(net, attr))
    assert net(torch.zeros(1, 3, 32, 32), torch.ones(1), return_features_early=True) == []


def test_constructor_errors():
    with pytest.raises(ValueError):
        EDMPrecond(**{**KW, "model_type": "DhariwalUNet"})
    with pytest.raises(ValueError):
        EDMPrecond(**{**KW, "drop_precond": "sideways"})
    with pytest.raises(NotImplementedError):
        EDMPrecond(**{**KW, "embedding_type": "fourier"})
    with pytest.raises(ValueError):
        EDMPrecond(**{**KW, "net_pred_type": "score"})
    with pytest.raises(KeyError):
        get_noise_schedule("nope")
    with pytest.raises(_lib.FastGenAMDError):
        EDMPrecond(**{**KW, "model_channels": 32})  # 64-wide blocks: kernels are tiled for 256


def test_product_path_fails_loudly_without_gpu(net):
    x, t = torch.zeros(2, 3, 32, 32), torch.ones(2, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="HIP GPU only"):  # autograd requested: the training path has no CPU fallback either
        net(x, t)
    prev, net.compute_dtype = net.compute_dtype, "fp32"
    try:
        with pytest.raises(NotImplementedError):  # the exact-fp32 mode has no backward pass
            net(x, t)
    finally:
        net.compute_dtype = prev
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="HIP GPU only"):
            net(x, t)
        with pytest.raises(RuntimeError, match="HIP GPU only"):
            FastGenModel.generator_fn(net, x, student_sample_steps=4)
        with pytest.raises(ValueError):
            net(x, t, r=t)
        with pytest.raises(RuntimeError, match="HIP GPU only"):
            net(x, t, feature_indices={0})
        with pytest.raises(RuntimeError, match="HIP GPU only"):
            net(x, t, feature_indices={0, 2}, return_features_early=True)


def test_schedule_mirror_matches_golden(golden_dir):
    fx = torch.load(os.path.join(golden_dir, "schedule.pt"), weights_only=True)
    s = EDMNoiseSchedule()
    assert torch.equal(s.sigmas[:3], fx["sigmas_head"]) and torch.equal(s.sigmas[-3:], fx["sigmas_tail"])
    for n in (1, 2, 4):
        assert torch.equal(s.get_t_list(n), fx[f"t_list_{n}"])
    g = lambda seed: torch.randn((2, 3, 8, 8), generator=torch.Generator().manual_seed(seed))  # noqa: E731
    x, e = g(11), g(12)
    t = torch.tensor([17.498123, 0.1726], dtype=torch.float64)
    assert torch.equal(s.forward_process(x, e, t), fx["fp_out"])
    assert torch.equal(s.latents(x, t_init=torch.tensor(79.5638, dtype=torch.float64)), fx["lat_out"])
    assert torch.equal(s.x0_to_eps(x, e, t), fx["x0eps_out"])
    # round trips of the prediction-type conversions
    x0 = g(13)
    xt = s.forward_process(x0, e, t)
    assert torch.allclose(s.convert_model_output(xt, x0, t, "x0", "eps"), e, atol=2e-3)
    assert torch.allclose(s.convert_model_output(xt, s.convert_model_output(xt, x0, t, "x0", "flow"), t, "flow", "x0"), x0, atol=1e-4)
    assert s.convert_model_output(xt, x0, t, "x0", "x0") is x0
    with pytest.raises(AssertionError):
        s.convert_model_output(xt, x0, t, "x0", "v")
    with pytest.raises(AssertionError):
        s.forward_process(x, e, torch.tensor([100.0, 1.0], dtype=torch.float64))


class _FakeNet(torch.nn.Module):
    """FastGenNetwork-shaped CPU module for the generic (non-fused) sampler loop."""

    def __init__(self):
        super().__init__()
        self.noise_scheduler = EDMNoiseSchedule()
        self.calls = []

    def forward(self, x, t, condition=None, fwd_pred_type=None):
        self.calls.append((t.clone(), self.training, torch.is_inference_mode_enabled()))
        return 0.5 * x / (1 + t.float().reshape(-1, 1, 1, 1))


def test_generic_sampler_loop_matches_reference_semantics():
    net = _FakeNet().train()
    noise = torch.randn(3, 3, 8, 8, generator=torch.Generator().manual_seed(0))
    out = FastGenModel.generator_fn(net, noise, student_sample_steps=4, student_sample_type="ode")
    assert net.training  # restored
    assert len(net.calls) == 4 and all((not tr) and inf for _, tr, inf in net.calls)
    tl = EDMNoiseSchedule().get_t_list(4)
    assert all(torch.equal(c[0], tl[i].expand(3)) for i, c in enumerate(net.calls))
    # same loop written out with the oracle's schedule functions
    x = R.latents(noise, tl[0])
    for t_cur, t_next in zip(tl[:-1], tl[1:]):
        xp = 0.5 * x / (1 + t_cur.float())
        if t_next > 0:
            x = R.forward_process(xp, R.x0_to_eps(x, xp, t_cur.expand(3)), t_next.expand(3))
    assert torch.equal(out, xp)
    with pytest.raises(AssertionError):
        FastGenModel.generator_fn(net, noise, student_sample_steps=2, t_list=[10.0, 1.0, 0.5])
    with pytest.raises(NotImplementedError):
        FastGenModel.generator_fn(net, noise, student_sample_steps=2, student_sample_type="euler")


# ---- MeanFlow student: r_timestep / drop_precond / rectified flow ---------------------------------------------------


def test_meanflow_module_and_state_dict(golden_dir):
    net = EDMPrecond(**KW_MF)
    want = {}
    for line in open(os.path.join(golden_dir, "state_dict_keys_meanflow.txt")):
        parts = line.split()
        want[parts[0]] = tuple(int(p) for p in parts[1:])
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == want
    assert net.net_pred_type == "flow" and net.schedule_type == "rf" and net.r_timestep and net.drop_precond == "both"
    assert isinstance(net.noise_scheduler, RFNoiseSchedule) and net.noise_scheduler.max_t == 0.999
    assert net.fused_loop() == "meanflow" and EDMPrecond(**KW).fused_loop() == "x0"
    assert EDMPrecond(**{**KW_MF, "net_pred_type": "x0"}).fused_loop() is None
    x, t = torch.zeros(2, 3, 32, 32), torch.full((2,), 0.5, dtype=torch.float64)
    with torch.no_grad():
        with pytest.raises(ValueError, match="needs r"):
            net(x, t)
        with pytest.raises(RuntimeError, match="HIP GPU only"):
            net(x, t, r=t)
        with pytest.raises(RuntimeError, match="HIP GPU only"):
            MeanFlowModel.generator_fn(net, x, student_sample_steps=2, student_sample_type="ode")
        with pytest.raises(NotImplementedError):  # the x0 loop has no meaning for an r_timestep flow network
            net.few_step_sample(x, None, [0.999, 0.0], loop="x0")


def test_rf_schedule_mirror_matches_golden(golden_dir):
    fx = torch.load(os.path.join(golden_dir, "schedule_rf.pt"), weights_only=True)
    s = get_noise_schedule("rf")
    for n in (1, 2, 4):
        assert torch.equal(s.get_t_list(n), fx[f"t_list_{n}"])
        arr = (ctypes.c_double * (n + 1))()
        _lib.check(_lib.lib().fg_rf_t_list(n, arr))
        assert torch.equal(torch.tensor(list(arr), dtype=torch.float64), fx[f"t_list_{n}"])
    g = lambda seed: torch.randn((2, 3, 8, 8), generator=torch.Generator().manual_seed(seed))  # noqa: E731
    x, e = g(11), g(12)
    t = torch.tensor([0.7492, 0.2497], dtype=torch.float64)
    assert torch.equal(s.forward_process(x, e, t), fx["fp_out"])
    assert torch.equal(s.latents(x, t_init=torch.tensor(0.999, dtype=torch.float64)), fx["lat_out"])
    assert torch.equal(s.x0_to_eps(x, e, t), fx["x0eps_out"])
    assert torch.equal(s.x0_to_flow(x, e, t), fx["flow_out"])
    assert s.max_sigma == fx["max_sigma"].item()
    assert torch.equal(s.alpha(t), 1 - t) and torch.equal(s.sigma(t), t)
    with pytest.raises(AssertionError):
        s.forward_process(x, e, torch.tensor([1.0, 0.5], dtype=torch.float64))


class _FakeFlowNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.noise_scheduler = RFNoiseSchedule()
        self.calls = []

    def forward(self, x, t, condition=None, r=None, fwd_pred_type=None):
        assert fwd_pred_type == "flow"
        self.calls.append((t.clone(), r.clone()))
        return x * (0.3 + t.float().reshape(-1, 1, 1, 1)) - r.float().reshape(-1, 1, 1, 1)


@pytest.mark.parametrize("name", ["edm", "rf"])
def test_training_side_schedule_members_match_golden(golden_dir, name):
    """The members the DMD2 / sCM / MeanFlow training callers read, against values recorded from the reference
    (oracle/gen_golden.py train_schedule): exact, including the seeded draws (same generator calls in the same order)."""
    fx = {k.split("/", 1)[1]: v for k, v in torch.load(os.path.join(golden_dir, "schedule_train.pt"), weights_only=True).items()
          if k.startswith(name + "/")}
    sched = get_noise_schedule(name)
    x = torch.randn((3, 3, 4, 4), generator=torch.Generator().manual_seed(31))
    e = torch.randn((3, 3, 4, 4), generator=torch.Generator().manual_seed(32))
    t = fx["t"]

    def same(got, key):
        want = fx[key]
        assert got.dtype == want.dtype and got.shape == want.shape, (key, got.dtype, want.dtype, got.shape, want.shape)
        assert torch.equal(got, want), key

    same(sched.rescale_t(t), "rescale_t")
    same(sched.alpha_prime(t), "alpha_prime")
    same(sched.sigma_prime(t), "sigma_prime")
    same(sched.cond_velocity(x, e, t), "cond_velocity")
    same(sched.sqrt_snr(t), "sqrt_snr")
    same(sched.sqrt_snr_to_t(torch.tensor([0.0, 0.3, 7.5, 2e-7], dtype=torch.float32)), "sqrt_snr_to_t")
    same(sched.closest_sigma_idx(fx["closest_probe"]), "closest_idx")
    same(sched.closest_sigma_idx(fx["closest_probe"][:3].reshape(3, 1, 1, 1)), "closest_idx_4d")
    same(sched.sigma_idx_to_t(torch.tensor([0, 17, 999])), "sigma_idx_to_t")
    same(sched.next_in_t_list(torch.tensor([0, 2, 1, 3]), 4, None), "next_default")
    same(sched.next_in_t_list(torch.tensor([0, 1]), 3, [0.9, 0.5, 0.2, 0.0], stride=2), "next_custom_stride2")
    with pytest.raises(ValueError):
        sched.next_in_t_list(torch.tensor([3]), 4, None, stride=2)
    with pytest.raises(AssertionError):
        sched.next_in_t_list(torch.tensor([0]), 4, [0.9, 0.5, 0.0])
    torch.manual_seed(77)
    tt, ii = sched.sample_from_t_list(16, 4, return_ids=True)
    same(tt, "sample_from_t_list")
    same(ii, "sample_from_t_list_ids")
    assert int(ii.max()) <= 3  # never the clean (t = 0) entry
    torch.manual_seed(78)
    same(sched.sample_from_t_list(8, 3, t_list=[0.9, 0.5, 0.2, 0.0]), "sample_from_custom")
    for kind in (("polynomial", "uniform", "lognormal") if name == "edm" else ("logitnormal", "uniform", "shifted")):
        torch.manual_seed(79)
        same(sched.sample_t(32, time_dist_type=kind), f"sample_t_{kind}")
    torch.manual_seed(80)
    got = sched.sample_t(32, time_dist_type="uniform", min_t=0.0001, max_t=0.5)
    same(got, "sample_t_bounded")
    assert float(got.min()) >= sched.min_t and float(got.max()) <= 0.5
    same(sched.safe_clamp(torch.tensor([-1.0, 0.4, 90.0], dtype=torch.float32), 0.002, 0.999), "safe_clamp_f32")
    same(sched.safe_clamp(torch.tensor([-1.0, 0.4, 90.0], dtype=torch.bfloat16), 0.002, 0.999), "safe_clamp_bf16")
    with pytest.raises(ValueError):
        sched.sample_t(4, time_dist_type="no-such-distribution")


def test_meanflow_generic_loop_matches_reference_semantics():
    net = _FakeFlowNet()
    noise = torch.randn(3, 3, 8, 8, generator=torch.Generator().manual_seed(0))
    tl = RFNoiseSchedule().get_t_list(3)
    out = MeanFlowModel.generator_fn(net, noise, student_sample_steps=3, student_sample_type="ode")
    assert [c[0][0].item() for c in net.calls] == tl[:-1].tolist()
    assert [c[1][0].item() for c in net.calls] == tl[1:].tolist()  # ode: r = t_next
    x = R.latents(noise, tl[0])
    for t_cur, t_next in zip(tl[:-1], tl[1:]):
        u = x * (0.3 + t_cur.float()) - t_next.float()
        x = x - (t_cur - t_next).float() * u
    assert torch.equal(out, x)
    net.calls.clear()
    torch.manual_seed(3)
    out = MeanFlowModel.generator_fn(net, noise, student_sample_steps=3, student_sample_type="sde")
    assert all((c[1] == 0).all() for c in net.calls)  # sde: r = 0
    torch.manual_seed(3)
    x = R.latents(noise, tl[0])
    for t_cur, t_next in zip(tl[:-1], tl[1:]):
        x = x - t_cur.float() * (x * (0.3 + t_cur.float()))
        if t_next > 0:
            x = R.forward_process(x, torch.randn_like(x), t_next.expand(3), "rf")
    assert torch.equal(out, x)
    with pytest.raises(NotImplementedError):
        MeanFlowModel.generator_fn(net, noise, student_sample_steps=2, student_sample_type="euler")


def test_schedule_mirrors_equal_oracle_on_random_inputs():
    """Property test: the host-side schedule mirrors and the oracle's schedule functions are the same maps (bit for bit) on
    random tensors and timesteps, for both schedules."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=25, deadline=None)
    @given(st.integers(0, 2**31 - 1), st.floats(0.002, 80.0), st.floats(0.0, 0.999))
    def prop(seed, t_edm, t_rf):
        g = torch.Generator().manual_seed(seed)
        x, e = torch.randn(2, 3, 4, 4, generator=g), torch.randn(2, 3, 4, 4, generator=g)
        for name, tv in (("edm", t_edm), ("rf", t_rf)):
            s = get_noise_schedule(name)
            t = torch.full((2,), tv, dtype=torch.float64)
            assert torch.equal(s.forward_process(x, e, t), R.forward_process(x, e, t, name))
            assert torch.equal(s.x0_to_eps(x, e, t), R.x0_to_eps(x, e, t, schedule=name))
            if tv > 0:
                assert torch.equal(s.latents(x, t[0]), R.latents(x, t[0]))
            # conversions invert each other up to rounding
            xt = s.forward_process(x, e, t)
            if tv > 1e-3:
                fl = s.convert_model_output(xt, x, t, "x0", "flow")
                assert torch.allclose(s.convert_model_output(xt, fl, t, "flow", "x0"), x, atol=1e-4 * max(1.0, tv))

    prop()


def test_training_abi_argument_checks_without_gpu():
    """The training-side entry points validate their arguments on the host before any kernel is launched (an undersized
    scratch or a wrong shape must be an error code, never an out-of-bounds access)."""
    L = _lib.lib()
    cfg = _lib.fg_edm_config.from_buffer_copy(EDMPrecond(**KW)._cfg)
    cfg.compute_dtype = _lib.FG_DTYPE_BF16
    h = ctypes.c_void_p()
    _lib.check(L.fg_edm_create(ctypes.byref(cfg), ctypes.byref(h)))
    try:
        with pytest.raises(_lib.FastGenAMDError, match="unknown parameter"):
            _lib.check(L.fg_edm_bind_grad(h, b"model.no_such.weight", None, 0))
        buf = (ctypes.c_float * 4)()
        with pytest.raises(_lib.FastGenAMDError, match="gradient elements"):
            _lib.check(L.fg_edm_bind_grad(h, b"model.enc.32x32_conv.bias", ctypes.cast(buf, ctypes.c_void_p), 4))
        _lib.check(L.fg_edm_bind_grad(h, b"model.enc.32x32_conv.bias", None, 0))  # unbinding is always accepted
        _lib.check(L.fg_edm_set_augment(h, None))  # resetting is always accepted
        assert L.fg_edm_backward_workspace_bytes(h, 0) == 0
        assert L.fg_edm_backward_workspace_bytes(h, 4) > L.fg_edm_workspace_bytes(h, 4)
        assert L.fg_edm_block_backward_workspace_bytes(h, 999, 2) == 0
        one = ctypes.cast(buf, ctypes.c_void_p)
        with pytest.raises(_lib.FastGenAMDError, match="not packed"):
            _lib.check(L.fg_edm_backward(h, one, one, None, one, one, one, 0, 2, one, 1 << 20, None))
        with pytest.raises(_lib.FastGenAMDError, match="not packed"):
            _lib.check(L.fg_edm_run_block_backward(h, 0, one, 128, None, 0, one, one, one, None, one, 2, one, 1 << 20, None))
        # the stand-alone weight-gradient op: shapes outside the kernel's tiling are rejected, sizes are reported
        assert L.fg_op_conv_wgrad_workspace_bytes(4, 32, 256, 256, 3) > 0
        assert L.fg_op_conv_wgrad_workspace_bytes(4, 32, 256, 200, 3) == 0
        assert L.fg_op_conv_wgrad_workspace_bytes(4, 24, 256, 256, 3) == 0
        with pytest.raises(_lib.FastGenAMDError, match="unsupported shape"):
            _lib.check(L.fg_op_conv_wgrad(one, one, one, 4, 32, 256, 256, 5, 0, one, 1 << 30, None))
        with pytest.raises(_lib.FastGenAMDError, match="workspace too small"):
            _lib.check(L.fg_op_conv_wgrad(one, one, one, 4, 32, 256, 256, 3, 0, one, 16, None))
    finally:
        L.fg_edm_destroy(h)


def test_trig_schedule_matches_golden(golden_dir):
    """TrigNoiseSchedule (the sCM-family time axis) against values recorded from the reference: exact."""
    fx = torch.load(os.path.join(golden_dir, "trigflow_b2.pt"), weights_only=True)
    s = get_noise_schedule("trig")
    t = fx["sched/t"]
    x = torch.randn((3, 3, 4, 4), generator=torch.Generator().manual_seed(81))
    e = torch.randn((3, 3, 4, 4), generator=torch.Generator().manual_seed(82))
    assert torch.equal(s.sqrt_snr(t), fx["sched/sqrt_snr"])
    assert torch.equal(s.sqrt_snr_to_t(torch.tensor([0.0, 0.4, 3.0, 1e3], dtype=torch.float32)), fx["sched/sqrt_snr_to_t"])
    assert torch.equal(s.forward_process(x, e, t), fx["sched/forward_process"])
    assert torch.equal(s.x0_to_flow(x, e, t), fx["sched/x0_to_flow"])
    assert torch.equal(s.flow_to_x0(x, e, t), fx["sched/flow_to_x0"])
    assert torch.equal(s.sigma_idx_to_t(torch.tensor([0, 17, 999])), fx["sched/sigma_idx_to_t"])
    assert s.max_sigma == float(fx["sched/max_sigma"])
    for k in ("uniform", "logitnormal"):
        torch.manual_seed(83)
        assert torch.equal(s.sample_t(16, time_dist_type=k), fx[f"sched/sample_t_{k}"])


def test_trigflow_wrapper_composition_with_the_oracle_as_network(golden_dir):
    """TrigFlowPrecond's input map, forward and composed jvp (torch.func.jvp of the elementwise maps around the network's own
    jvp) against the reference's wrapper under torch.func.jvp - with the CPU oracle standing in for the network, so the
    composition logic is checked without a GPU."""
    from fastgen_amd.methods.consistency_model.sCM import TrigFlowPrecond

    fx = torch.load(os.path.join(golden_dir, "trigflow_b2.pt"), weights_only=True)
    sd = R.random_state_dict(R.CIFAR10, seed=1234)

    class OracleNet(torch.nn.Module):
        noise_scheduler = get_noise_schedule("edm")

        def forward(self, x, t, condition=None, return_logvar=False, fwd_pred_type="x0"):
            return R.edm_precond_forward(sd, R.CIFAR10, x, t, condition)

        def jvp(self, x, t, vx, vt, condition=None, fwd_pred_type=None):
            return R.edm_precond_jvp(sd, R.CIFAR10, x, t, condition, vx, vt)

    w = TrigFlowPrecond(OracleNet(), sigma_data=0.5)
    xh = torch.randn((2, 3, 32, 32), generator=torch.Generator().manual_seed(84)) * 0.5
    vx = torch.randn((2, 3, 32, 32), generator=torch.Generator().manual_seed(85))
    x_t, t = w._convert_trigflow_to_net_input(xh, fx["wrap/t_hat"])
    assert torch.equal(x_t, fx["wrap/x_t"]) and torch.equal(t, fx["wrap/t"])
    with torch.no_grad():
        F = w(xh, fx["wrap/t_hat"], condition=fx["wrap/cond"])
    assert torch.allclose(F, fx["wrap/F"], rtol=1e-4, atol=1e-5)
    F2, dF = w.jvp(xh, fx["wrap/t_hat"], vx, fx["wrap/vt"], condition=fx["wrap/cond"])
    assert torch.allclose(F2, fx["wrap/F"], rtol=1e-4, atol=1e-5)
    assert float((dF - fx["wrap/dF"]).norm() / fx["wrap/dF"].norm()) <= 1e-4


def test_label_dropout_follows_reference_semantics():
    """label_dropout zeroes whole class-label rows in training mode only, with the reference's draw (torch.rand([B, 1]) on the
    input's device, EDM/network.py:515-516); eval mode leaves the labels alone."""
    n = EDMPrecond(**{**KW, "label_dropout": 0.5})
    c = torch.eye(10)[:8]
    n.train()
    torch.manual_seed(3)
    got = n._labels(c, 8, torch.device("cpu"))
    torch.manual_seed(3)
    want = c * (torch.rand([8, 1]) >= 0.5).to(c.dtype)
    assert torch.equal(got, want) and 0 < int(got.sum()) < 8
    n.eval()
    assert torch.equal(n._labels(c, 8, torch.device("cpu")), c)
