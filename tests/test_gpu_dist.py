"""The drop-in EDMPrecond under the reference's two data-parallel wrappers, on ONE GPU (run with -m gpu):

  * FSDP2 (`net.fully_shard(mesh=...)`, fastgen/networks/EDM/network.py:861-879 + fastgen/utils/distributed/fsdp.py:67-220) on a
    1-rank device mesh: parameters are sharded DTensors between calls, the fused engine runs on all-gathered parameters, gradients
    come back through FSDP2's reduce-scatter — everything bit-equal to the bare module.
  * DDP (fastgen/utils/distributed/ddp.py:44-72) with TWO ranks: two child processes share cuda:0 over the gloo backend (RCCL
    refuses two ranks on one device; gloo reduces CUDA tensors through the host), each with its own half of the batch; the
    averaged gradients must equal the mean of the two single-rank gradients.
"""
import os
import socket

import pytest
import torch

from fastgen_amd.methods.model import FastGenModel
from fastgen_amd.networks.EDM.network import EDMPrecond
from oracle import edm_ref as R

pytestmark = pytest.mark.gpu

KW = dict(img_resolution=32, img_channels=3, label_dim=10, sigma_shift=0.0, sigma_data=0.5, model_type="SongUNet",
          augment_dim=9, model_channels=128, channel_mult=[2, 2, 2], channel_mult_noise=1, embedding_type="positional",
          encoder_type="standard", decoder_type="standard", resample_filter=[1, 1], dropout=0.0, label_dropout=0,
          r_timestep=False, drop_precond=None)


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _seeded(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _make(sd, dev, mode="bf16"):
    net = EDMPrecond(compute_dtype=mode, **KW)
    net.load_state_dict(sd, strict=True)
    return net.to(dev)


def test_fsdp2_one_rank_mesh_equals_bare_module():
    import torch.distributed as dist
    from torch.distributed.device_mesh import init_device_mesh
    from torch.distributed.tensor import DTensor

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sd = R.random_state_dict(R.CIFAR10, seed=1234)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
        created = True
    try:
        mesh = init_device_mesh("cuda", (1,))
        bare, net = _make(sd, dev), _make(sd, dev)
        net.fully_shard(mesh=mesh)
        assert all(isinstance(p, DTensor) for p in net.parameters())
        assert list(net.state_dict().keys()) == list(bare.state_dict().keys())  # checkpoint names are untouched
        with pytest.raises(RuntimeError, match="sharded DTensor"):
            net._engine(dev)  # the engine itself refuses sharded parameters: only the wrapped entry points gather them

        B = 4
        t = torch.tensor([3.1, 0.4, 41.0, 1.2], dtype=torch.float64, device=dev)
        x = (_seeded((B, 3, 32, 32), 5) * t.reshape(B, 1, 1, 1).float().cpu()).to(dev)
        cond = torch.nn.functional.one_hot(torch.arange(B) % 10, 10).float().to(dev)
        dout = _seeded((B, 3, 32, 32), 6).to(dev)

        # inference forward, feature taps, fused sampler, forward-mode derivative: bit-equal
        for m in (bare, net):
            m.eval()
        with torch.no_grad():
            assert torch.equal(net(x, t, condition=cond), bare(x, t, condition=cond))
            fa, fb = net(x, t, condition=cond, feature_indices={0, 2}), bare(x, t, condition=cond, feature_indices={0, 2})
            assert torch.equal(fa[0], fb[0]) and all(torch.equal(a, b) for a, b in zip(fa[1], fb[1]))
        assert all(isinstance(p, DTensor) for p in net.parameters())  # every group resharded after the call
        noise = _seeded((B, 3, 32, 32), 7).to(dev)
        ga = FastGenModel.generator_fn(net, noise, condition=cond, student_sample_steps=2, student_sample_type="ode")
        gb = FastGenModel.generator_fn(bare, noise, condition=cond, student_sample_steps=2, student_sample_type="ode")
        assert torch.equal(ga, gb)
        v = _seeded((B, 3, 32, 32), 8).to(dev)
        ja, jb = net.jvp(x, t, v, condition=cond), bare.jvp(x, t, v, condition=cond)
        assert torch.equal(ja[0], jb[0]) and torch.equal(ja[1], jb[1])

        # training step: loss.backward() leaves sharded .grad DTensors equal to the bare module's gradients, AdamW steps the
        # sharded parameters, the next forward sees the update
        for m in (bare, net):
            m.train()
        oa, ob = torch.optim.AdamW(net.parameters(), lr=1e-3), torch.optim.AdamW(bare.parameters(), lr=1e-3)
        for it in range(2):
            for m, o in ((net, oa), (bare, ob)):
                o.zero_grad(set_to_none=True)
                out, lv = m(x, t, condition=cond, fwd_pred_type="x0", return_logvar=True)
                ((out * dout).sum() + lv.sum()).backward()
            ga = {n: p.grad for n, p in net.named_parameters()}
            for n, p in bare.named_parameters():
                if p.grad is None:
                    continue
                g = ga[n]
                assert isinstance(g, DTensor), n
                assert torch.equal(g.full_tensor(), p.grad), (it, n)
            oa.step()
            ob.step()
            for (n, p), (_, q) in zip(net.named_parameters(), bare.named_parameters()):
                assert torch.equal(p.full_tensor(), q), (it, n)
        with torch.no_grad():
            assert torch.equal(net(x, t, condition=cond), bare(x, t, condition=cond))
    finally:
        if created:
            dist.destroy_process_group()


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        sd = R.random_state_dict(R.CIFAR10, seed=1234)
        net = _make(sd, dev).train()
        B = 2
        t = torch.tensor([[3.1, 0.4], [41.0, 1.2]][rank], dtype=torch.float64, device=dev)
        x = (_seeded((B, 3, 32, 32), 50 + rank) * t.reshape(B, 1, 1, 1).float().cpu()).to(dev)
        cond = torch.nn.functional.one_hot(torch.tensor([[1, 2], [3, 4]][rank]), 10).float().to(dev)
        dout = _seeded((B, 3, 32, 32), 60 + rank).to(dev)
        # this rank's own gradients, bare module
        (net(x, t, condition=cond, fwd_pred_type="x0") * dout).sum().backward()
        own = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        net.zero_grad(set_to_none=True)
        # expected DDP result: the mean over ranks (gathered on the host)
        want = {}
        for n, g in own.items():
            parts = [torch.zeros_like(g, device="cpu") for _ in range(world)]
            dist.all_gather(parts, g.cpu())
            want[n] = (parts[0].double() + parts[1].double()) / world
        ddp = DDP(net, device_ids=[0], find_unused_parameters=True)  # as the reference wraps it (ddp.py:44-52)
        (ddp(x, t, condition=cond, fwd_pred_type="x0") * dout).sum().backward()
        worst = 0.0
        for n, p in net.named_parameters():
            assert (p.grad is not None) == (n in want), n  # every hook of a used parameter fired
            if n in want:
                w = want[n]
                err = float((p.grad.cpu().double() - w).abs().max() / w.abs().max().clamp_min(1e-30))
                worst = max(worst, err)
        # both ranks hold the same reduced gradients
        probe = net.model.enc._modules["16x16_block1"]._modules["conv1"].weight.grad.cpu()
        parts = [torch.zeros_like(probe) for _ in range(world)]
        dist.all_gather(parts, probe)
        q.put((rank, worst, bool(torch.equal(parts[0], parts[1])), float((own["model.enc.16x16_block1.conv1.weight"].cpu() - probe).abs().max())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_ddp_two_ranks_average_gradients():
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=600) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    for rank, worst, same, moved in res:
        assert worst <= 1e-6, (rank, worst)   # fp32 mean of two fp32 gradients, up to the order of one addition
        assert same, rank                    # identical on both ranks
        assert moved > 0, rank               # and different from this rank's own gradient: a reduction did happen


def _one_rank_mesh():
    import torch.distributed as dist
    from torch.distributed.device_mesh import init_device_mesh

    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
        created = True
    return init_device_mesh("cuda", (1,)), created


def test_fsdp2_dit_one_rank_mesh_equals_bare_module():
    """`DiT.fully_shard` with the reference's grouping (DiT/network.py:402-420: every block, the embedders, the output layer) on a 1-rank
    mesh: parameters are sharded DTensors before and after every call, the engine gathers / packs / reshards group by group, and forward,
    the fused student loop and the Euler sampler are bit-equal to the bare module; an in-place update of one block's parameters is seen
    (only that group is gathered again).  Multi-rank: unmeasured (one GPU here)."""
    import torch.distributed as dist
    from torch.distributed.tensor import DTensor

    from fastgen_amd.networks import _weights
    from fastgen_amd.networks.DiT.network import DiT
    from oracle import dit_ref as DR

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    mesh, created = _one_rank_mesh()
    try:
        kw = dict(hidden_size=384, depth=12, num_heads=6, compute_dtype="bf16")
        sd = DR.random_state_dict(DR.S_2, seed=77)
        bare, net = DiT(**kw), DiT(**kw)
        for m in (bare, net):
            m.load_state_dict(sd, strict=True)
            m.to(dev).eval()
        net.fully_shard(mesh=mesh)
        grouped = [n for n, _ in net.named_parameters() if not n.startswith("logvar_linear")]
        assert all(isinstance(p, DTensor) for n, p in net.named_parameters() if n in grouped)
        assert list(net.state_dict().keys()) == list(bare.state_dict().keys())
        groups = _weights.weight_groups(net, net._names)
        assert sum(g[2] is not None for g in groups) == 12 + 4 and {g[0] for g in groups if g[2] is None} >= {"pos_embed"}
        B = 3
        x = _seeded((B, 4, 32, 32), 5).to(dev)
        t = torch.tensor([0.9, 0.5, 0.1], dtype=torch.float64, device=dev)
        cond = torch.nn.functional.one_hot(torch.tensor([1, 2, 3]), 1000).float().to(dev)
        with torch.inference_mode():
            assert torch.equal(net(x, t, condition=cond), bare(x, t, condition=cond))
            assert all(isinstance(p, DTensor) for n, p in net.named_parameters() if n in grouped)  # every group resharded
            assert torch.equal(FastGenModel.generator_fn(net, x, condition=cond, student_sample_steps=3, student_sample_type="ode"),
                               FastGenModel.generator_fn(bare, x, condition=cond, student_sample_steps=3, student_sample_type="ode"))
            neg = torch.zeros(B, 1000, device=dev)
            assert torch.equal(net.sample(x, condition=cond, neg_condition=neg, guidance_scale=2.0, num_steps=3),
                               bare.sample(x, condition=cond, neg_condition=neg, guidance_scale=2.0, num_steps=3))
        # an optimizer-style in-place update of one block on both modules: the sharded module re-packs that group and follows
        with torch.no_grad():
            for m in (bare, net):
                dict(m.named_parameters())["blocks.5.feed_forward.fc1.weight"].mul_(1.5)
        with torch.inference_mode():
            a, b = net(x, t, condition=cond), bare(x, t, condition=cond)
        assert torch.equal(a, b)
    finally:
        if created:
            dist.destroy_process_group()


def test_fsdp2_causal_video_dit_one_rank_mesh_equals_bare_module():
    """`CausalWan.fully_shard` (Wan/network.py:761-782: every block, then the transformer as the root group) on a 1-rank mesh: the
    autoregressive calls and the fused chunk loop are bit-equal to the bare module, parameters sharded before and after."""
    import torch.distributed as dist
    from torch.distributed.tensor import DTensor

    from fastgen_amd.methods.distribution_matching.causvid import CausVidModel
    from fastgen_amd.networks import _weights
    from fastgen_amd.networks.Wan.network_causal import CausalWan
    from oracle import wan_ref as WR

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    mesh, created = _one_rank_mesh()
    try:
        kw = dict(num_attention_heads=2, attention_head_dim=128, text_dim=128, ffn_dim=512, num_layers=2, chunk_size=2, total_num_frames=6)
        sd = WR.random_state_dict(WR.TINY, 7)
        bare, net = CausalWan(**kw), CausalWan(**kw)
        for m in (bare, net):
            m.load_state_dict(sd, strict=True)
            m.to(dev).eval()
        net.fully_shard(mesh=mesh)
        assert all(isinstance(p, DTensor) for p in net.parameters())
        assert list(net.state_dict().keys()) == list(bare.state_dict().keys())
        groups = _weights.weight_groups(net, [n for n in net._names if "logvar" not in n])
        assert [g[0] for g in groups] == ["transformer.blocks.0.", "transformer.blocks.1.", "transformer."] and groups[2][1] == "transformer.blocks."
        g = torch.Generator().manual_seed(8)
        x = torch.randn(1, 16, 4, 16, 16, generator=g).to(dev)
        text = torch.randn(1, 16, 128, generator=g).to(dev)
        t = torch.tensor([0.6], dtype=torch.float64, device=dev)
        with torch.inference_mode():
            for lo, store in ((0, True), (2, False)):
                a = net(x[:, :, lo:lo + 2], t, condition=text, cur_start_frame=lo, store_kv=store, is_ar=True)
                b = bare(x[:, :, lo:lo + 2], t, condition=text, cur_start_frame=lo, store_kv=store, is_ar=True)
                assert torch.equal(a, b)
            assert all(isinstance(p, DTensor) for p in net.parameters())
            net.clear_caches(), bare.clear_caches()
            tl = [0.999, 0.5, 0.0]
            ga = CausVidModel.generator_fn(net, x, student_sample_steps=2, t_list=tl, condition=text, student_sample_type="ode")
            gb = CausVidModel.generator_fn(bare, x, student_sample_steps=2, t_list=tl, condition=text, student_sample_type="ode")
            assert torch.equal(ga, gb)
    finally:
        if created:
            dist.destroy_process_group()
