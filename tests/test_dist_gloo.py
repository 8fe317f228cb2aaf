"""World-size-2 test of the multi-GPU measurement path on CPU (gloo): sampling shards by image with no data-path
collective, so the only cross-rank logic is bench.timed_region's fences and the MAX-over-ranks reduction."""
import os
import socket
import time

import torch
import torch.multiprocessing as mp

import bench


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    assert bench.dist_env() == (rank, rank, world)
    dist = bench.init_dist(world, "gloo")
    calls = [0]

    def step():  # rank 1 is the slow replica
        calls[0] += 1
        time.sleep(0.02 * (1 + 2 * rank))

    dt = bench.timed_region(step, steps=5, warmup=2, world=world, sync_fn=lambda: None)
    # each rank draws different noise (seed + rank) — replicas must not generate the same images
    g = torch.Generator().manual_seed(1000 + rank)
    sample = torch.randn(4, generator=g)
    gathered = [torch.zeros(4) for _ in range(world)]
    dist.all_gather(gathered, sample)
    q.put((rank, dt, calls[0], [t.tolist() for t in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_timed_region_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, c0, g0), (r1, dt1, c1, g1) = res
    assert c0 == c1 == 7  # exactly warmup + steps calls on every rank
    assert abs(dt0 - dt1) < 1e-9  # both ranks report the SAME number: the max over ranks
    assert dt0 >= 5 * 0.06 * 0.9  # ... which is the slow rank's time (5 steps x 60 ms), not the fast rank's 100 ms
    assert g0 == g1 and g0[0] != g0[1]  # replicas hold different noise


def test_single_rank_needs_no_process_group():
    n = [0]
    dt = bench.timed_region(lambda: n.__setitem__(0, n[0] + 1), steps=3, warmup=1, world=1, sync_fn=lambda: None)
    assert n[0] == 4 and dt >= 0


def test_bench_line_contract():
    """The JSON line bench.py prints carries every field the driver reads, with the agreed types."""
    import json

    import bench

    roof = {"bound": "mfma", "kernel": "k", "achieved": 1.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.0004, "traffic": None}
    cpu = {"value": 10.0, "unit": "img/s", "cores": 16, "kind": "port", "sample": "s"}
    line = json.loads(json.dumps(bench.result_line(6000.0, 2, 10, 3, 1.7, "bf16", 512, 4, True, roof, cpu)))
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[k], typ), k
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 1024 and "workload" in line["config"]
    assert "model" not in line["config"]
    assert set(roof) <= set(line["roofline"]) and {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert abs(line["ms_per_step"] - 170.0) < 1e-9
