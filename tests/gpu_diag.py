"""GPU diagnostic (test infrastructure, not collected by pytest; run as `python tests/gpu_diag.py`): prints per-op / per-block error of the HIP path against the oracle + golden fixtures
without asserting, so one run on the GPU box localises a wrong kernel."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastgen_amd import _lib  # noqa: E402
from fastgen_amd.networks.EDM.network import EDMPrecond  # noqa: E402
from oracle import edm_ref as R  # noqa: E402

dev = torch.device("cuda:0")
G = os.path.join(ROOT, "tests", "golden")


def seeded(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def report(tag, got, want):
    d = (got - want).abs()
    rel = ((got - want).norm() / want.norm().clamp_min(1e-12)).item()
    print(f"{tag:34s} max_abs={d.max().item():.3e} rel_l2={rel:.3e} |want|max={want.abs().max().item():.3f} "
          f"nan={int(torch.isnan(got).sum())}", flush=True)


def main():
    cfg = R.CIFAR10
    sd = R.random_state_dict(cfg, seed=1234)
    kw = dict(img_resolution=32, img_channels=3, label_dim=10, sigma_shift=0.0, sigma_data=0.5, model_type="SongUNet",
              augment_dim=9, model_channels=128, channel_mult=[2, 2, 2], channel_mult_noise=1,
              embedding_type="positional", encoder_type="standard", decoder_type="standard", resample_filter=[1, 1],
              dropout=0.0, label_dropout=0, r_timestep=False, drop_precond=None)
    L = _lib.lib()
    fx = torch.load(os.path.join(G, "blocks_full.pt"), weights_only=True)
    enc, dec = R.layout(cfg)
    blocks = [b for b in enc + dec if b.kind == "block"]
    for mode in (sys.argv[1:] or ("fp32", "bf16x3", "bf16")):
        print(f"==== {mode} ====")
        net = EDMPrecond(compute_dtype=mode, **kw)
        net.load_state_dict(sd)
        net = net.to(dev).eval()
        with torch.inference_mode():
            dt, h = net._engine(dev)
            # embedding
            f = torch.load(os.path.join(G, "forward_full_b2.pt"), weights_only=True)
            x = seeded((2, 3, 32, 32), 21) * f["t"].reshape(2, 1, 1, 1).float()
            ws = net._workspace(dt, h, 2, dev)
            out = torch.empty(2, 3, 32, 32, device=dev)
            emb = torch.empty(2, 512, device=dev)
            xd, td, cd = x.to(dev), f["t"].to(dev), f["cond"].to(dev)
            _lib.check(L.fg_edm_forward(h, xd.data_ptr(), td.data_ptr(), None, cd.data_ptr(), out.data_ptr(), emb.data_ptr(), 2,
                                        ws.data_ptr(), ws.numel(), None))
            torch.cuda.synchronize()
            report("emb", emb.cpu(), f["emb"])
            report("forward_full_b2", out.cpu(), f["out"])
            # blocks
            names = sorted({k.split("/")[0] for k in fx if "/" in k})
            for n in names:
                key = fx[f"{n}/key"]
                bi = [i for i, b in enumerate(blocks) if b.key == key][0]
                b = blocks[bi]
                bs = fx[f"{n}/out"].shape[0]
                rin = b.res * 2 if b.down else (b.res // 2 if b.up else b.res)
                xb = seeded((bs, b.cin, rin, rin), int(fx[f"{n}/seed"]))
                c2 = b.skip_from or 0
                c1 = b.cin - c2
                x1 = nhwc(xb[:, :c1]).to(dev)
                x2 = nhwc(xb[:, c1:]).to(dev) if c2 else None
                e = fx["emb"][:bs].to(dev).contiguous()
                o = torch.empty(bs, b.res, b.res, b.cout, device=dev)
                ws = net._workspace(dt, h, bs, dev)
                _lib.check(L.fg_edm_run_block(h, bi, x1.data_ptr(), c1, x2.data_ptr() if c2 else None, c2, e.data_ptr(),
                                              o.data_ptr(), bs, ws.data_ptr(), ws.numel(), None))
                torch.cuda.synchronize()
                report(f"block {n} ({key.split('.')[-1]})", nchw(o.cpu()), fx[f"{n}/out"])
            # sampler
            s = torch.load(os.path.join(G, "sampler_full_b2.pt"), weights_only=True)
            noise = seeded((2, 3, 32, 32), 0).to(dev)
            eps = torch.stack([seeded((2, 3, 32, 32), k) for k in (1, 2, 3)]).to(dev)
            tl = net.noise_scheduler.get_t_list(4)
            for ug in (False, True):
                o = net.few_step_sample(noise, s["cond"].to(dev), tl, "sde", eps=eps, use_graph=ug)
                torch.cuda.synchronize()
                report(f"sampler sde graph={ug}", o.cpu(), s["out_sde"])
            o = net.few_step_sample(noise, s["cond"].to(dev), tl, "ode")
            report("sampler ode", o.cpu(), s["out_ode"])
            # timing at a few batch sizes
            for B in (16, 128, 512):
                nz = torch.randn(B, 3, 32, 32, device=dev)
                cnd = torch.nn.functional.one_hot(torch.arange(B) % 10, 10).float().to(dev)
                for _ in range(2):
                    net.few_step_sample(nz, cnd, tl, "sde", seed=1)
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(3):
                    net.few_step_sample(nz, cnd, tl, "sde", seed=1)
                torch.cuda.synchronize()
                dtm = (time.time() - t0) / 3
                print(f"  B={B}: {dtm * 1e3:.1f} ms / 4-step batch -> {B / dtm:.1f} img/s", flush=True)


if __name__ == "__main__":
    main()
