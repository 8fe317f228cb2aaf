"""Time one DMD2 training iteration of the EDM CIFAR-10 config on one MI355X with the three U-Nets (student, frozen teacher,
fake score) running on the fastgen_amd module, synthetic data.  Mirrors the structure of the reference's
`DMD2Model._student_update_step` / `_fake_score_discriminator_update_step` (fastgen/methods/distribution_matching/dmd2.py:186-247,
319-400, config: gan_loss_weight_gen = 1e-3, student_update_freq = 5, no CFG, bf16 AMP):

  student step        : student fwd+bwd; fake score fwd; teacher fwd WITH bottleneck tap, discriminator on the tap, GAN loss
                        differentiated through the teacher to its input (-> student); VSD loss
  fake-score/disc step: student fwd (no grad); fake score fwd+bwd (denoising loss); teacher encoder twice (fake / real taps, no
                        grad); discriminator fwd+bwd

The discriminator is fastgen_amd's Discriminator_EDM (networks/discriminators.py:62-137, bottleneck head); optimizers are
torch.optim.AdamW.  A measurement script: the losses are the reference's formulas written with torch ops.
Usage: python scripts/dmd2_step_bench.py [batch ...]   (default 64 256; the config's per-GPU batch on 8 GPUs is 256)"""
import sys
import time

import torch
import torch.nn.functional as F

from fastgen_amd.networks.discriminators import Discriminator_EDM
from fastgen_amd.networks.EDM.network import EDMPrecond

KW = dict(img_resolution=32, img_channels=3, label_dim=10, model_type="SongUNet", augment_dim=9, model_channels=128,
          channel_mult=[2, 2, 2], num_blocks=4, attn_resolutions=[16], embedding_type="positional", encoder_type="standard",
          decoder_type="standard", resample_filter=[1, 1], dropout=0.0)
dev = torch.device("cuda")


def make(seed, train):
    n = EDMPrecond(compute_dtype="bf16", **KW).randomize_parameters_(seed=seed).to(dev).eval()
    n.requires_grad_(train)
    return n


student, teacher, fake = make(1, True), make(2, False), make(3, True)
disc = Discriminator_EDM().to(dev)  # the bottleneck head (the reference's default feature_indices)
opt_s = torch.optim.AdamW(student.parameters(), lr=1e-5)
opt_f = torch.optim.AdamW(fake.parameters(), lr=1e-5)
opt_d = torch.optim.AdamW(disc.parameters(), lr=1e-5)
sched = student.noise_scheduler


def vsd_loss(gen, teacher_x0, fake_x0):
    with torch.no_grad():
        w = 1 / ((gen.float() - teacher_x0.float()).abs().mean(dim=(1, 2, 3), keepdim=True) + 1e-6)
        target = gen - (fake_x0 - teacher_x0) * w
    return 0.5 * F.mse_loss(gen, target)


def student_step(B, noise, cond, eps, t):
    t_student = torch.full((B,), sched.max_t, dtype=torch.float64, device=dev)
    gen = student(noise * sched.max_t, t_student, condition=cond, fwd_pred_type="x0")
    xt = sched.forward_process(gen, eps, t)
    with torch.no_grad():
        fake_x0 = fake(xt, t, condition=cond, fwd_pred_type="x0")
    teacher_x0, feat = teacher(xt, t, condition=cond, feature_indices={2}, fwd_pred_type="x0")
    gan_gen = F.softplus(-disc([feat[0]])).mean()
    loss = vsd_loss(gen, teacher_x0.detach(), fake_x0) + 1e-3 * gan_gen
    opt_s.zero_grad(set_to_none=True)
    loss.backward()
    opt_s.step()


def fake_score_step(B, noise, cond, eps, t, real):
    with torch.no_grad():
        t_student = torch.full((B,), sched.max_t, dtype=torch.float64, device=dev)
        gen = student(noise * sched.max_t, t_student, condition=cond, fwd_pred_type="x0")
        xt = sched.forward_process(gen, eps, t)
    loss_f = F.mse_loss(fake(xt, t, condition=cond, fwd_pred_type="x0"), gen)
    with torch.no_grad():
        fake_feat = teacher(xt, t, condition=cond, return_features_early=True, feature_indices={2})
        real_feat = teacher(sched.forward_process(real, eps, t), t, condition=cond, return_features_early=True, feature_indices={2})
    loss_d = F.softplus(disc([fake_feat[0]])).mean() + F.softplus(-disc([real_feat[0]])).mean()
    opt_f.zero_grad(set_to_none=True)
    opt_d.zero_grad(set_to_none=True)
    (loss_f + loss_d).backward()
    opt_f.step()
    opt_d.step()


for B in [int(a) for a in sys.argv[1:]] or [64, 256]:
    g = torch.Generator(device=dev).manual_seed(B)
    noise = torch.randn(B, 3, 32, 32, device=dev, generator=g)
    eps = torch.randn(B, 3, 32, 32, device=dev, generator=g)
    real = torch.randn(B, 3, 32, 32, device=dev, generator=g).clamp(-1, 1)
    cond = F.one_hot(torch.arange(B, device=dev) % 10, 10).float()
    t = sched.sample_t(B, device=dev)
    steps = {"student step": lambda: student_step(B, noise, cond, eps, t),
             "fake-score / discriminator step": lambda: fake_score_step(B, noise, cond, eps, t, real)}
    res = {}
    for name, fn in steps.items():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / n
        print(f"B={B:4d} {name:32s}: {res[name] * 1e3:8.2f} ms  {B / res[name]:8.1f} img/s")
    it = (res["student step"] + 4 * res["fake-score / discriminator step"]) / 5  # student_update_freq = 5
    print(f"B={B:4d} average iteration (1 student : 4 fake-score steps): {it * 1e3:8.2f} ms  {B / it:8.1f} img/s per GPU")
    assert all(torch.isfinite(p).all() for p in student.parameters())
