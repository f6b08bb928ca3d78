"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py is 'contiguous shards, no data-path collective, max of the
elapsed times' (SURVEY.md §8e). Each rank solves its shard (the oracle stands in for the GPU here) and the union must be
exactly the single-process result; the only collectives are the barrier and the MAX all-reduce of the timing."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, B, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401
    import common
    import oracle
    import wbc_shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wx, _ = common.models()
    cfg = common.config("c3", wx)
    d = common.tick_inputs(wx, cfg, B, seed=77)                    # every rank can regenerate the global batch
    lo, hi = wbc_shard.shard_range(B, rank, world)
    sub = {k: v[lo:hi] for k, v in d.items()}
    out = oracle.tick([wx], [cfg], sub, 0.002, hi - lo)
    dist.barrier()
    worst = wbc_shard.max_over_ranks(1.0 + rank, dist)             # rank r pretends to have taken 1 + r seconds
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), qdot=out["qdot"], lo=lo, hi=hi, worst=worst)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_reproduce_the_single_process_batch(tmp_path):
    B, world = 96, 2
    mp.spawn(_worker, args=(world, 29531, B, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    import common
    import oracle
    wx, _ = common.models()
    cfg = common.config("c3", wx)
    d = common.tick_inputs(wx, cfg, B, seed=77)
    ref = oracle.tick([wx], [cfg], d, 0.002, B)["qdot"]
    got = np.zeros_like(ref)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        got[int(z["lo"]):int(z["hi"])] = z["qdot"]
        assert float(z["worst"]) == 2.0                               # max over ranks of (1, 2)
    assert np.array_equal(got, ref)                                   # shards are independent: bit-identical


class _OracleEngine:
    """Stands in for bench.GpuEngine on a CPU rank: same interface, the tick is the oracle's."""
    name = "oracle-cpu"

    def __init__(self):
        import common
        self.model = common.models()[0]
        self.cfg = common.config("c3", self.model)
        self.options = {}
        self.steps_run = 0

    def fk(self, q):
        import oracle
        return oracle.fk([self.model], q, want_com=False)["oMf"]

    def load(self, host_in):
        self.inp = host_in

    def step(self):
        import oracle
        B = self.inp["q"].shape[0]
        self.out = oracle.tick([self.model], [self.cfg], self.inp, 0.002, B, want_q_next=False)
        self.steps_run += 1

    def sync(self):
        pass

    def timed_block(self, steps):
        import time
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        ms = 1e3 * (time.perf_counter() - t0)
        return lambda: ms

    def results(self):
        return self.out

    def path(self):
        return "oracle"


def _bench_worker(rank, world, port, out_dir):
    import argparse
    import json
    root = os.path.dirname(HERE)
    sys.path.insert(0, HERE)
    sys.path.insert(0, root)
    import conftest  # noqa: F401
    import bench
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    args = argparse.Namespace(batch=40, steps=2, warmup=1, repeats=3, posture="PREV")
    comm = bench.Comm("gloo", rank, world, None)
    eng = _OracleEngine()
    line, res = bench.run_rank(args, comm, eng, bench.make_inputs)
    assert eng.steps_run == args.warmup + args.steps * args.repeats         # exactly K steps per timed block
    np.savez(os.path.join(out_dir, "bench_r%d.npz" % rank), qdot=res["qdot"], q=eng.host_in["q"])
    if rank == 0:
        assert line is not None
        with open(os.path.join(out_dir, "line.json"), "w") as f:
            json.dump(line, f)
    else:
        assert line is None
    comm.close()


def test_bench_rank_body_over_gloo(tmp_path):
    """bench.py's per-rank body (seeding by rank, K-step blocks between barriers, MAX over ranks, the JSON line) on two CPU
    ranks over gloo with the oracle as the engine: the code the multi-GPU run executes, minus the GPU."""
    import json
    world = 2
    mp.spawn(_bench_worker, args=(world, 29547, str(tmp_path)), nprocs=world, join=True)
    line = json.load(open(os.path.join(str(tmp_path), "line.json")))
    assert line["metric"] == "wbc_qp_solves_per_sec" and line["unit"] == "ticks/s" and line["n_gpus"] == 2
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["vs_baseline"] is None and line["dtype"] == "f64"
    assert line["config"]["global_batch"] == 80 and line["config"]["batch_per_gpu"] == 40 and line["config"]["parallelism"].startswith("shard2")
    assert line["steps"] == 2 and line["warmup"] == 1 and line["repeats"]["n"] == 3 and len(line["repeats"]["ms_per_step"]) == 3
    # whole-job aggregate: all ranks' ticks over the (max-over-ranks) time of the median block
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 80.0) < 1e-6
    assert sorted(line["repeats"]["ms_per_step"])[1] == pytest.approx(line["ms_per_step"])
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"]
    # every rank drew its own shard (seed = rank) and solved it
    sys.path.insert(0, HERE)
    import common
    import oracle
    import wbc_workload
    wx = common.models()[0]
    cfg = common.config("c3", wx)
    z = [np.load(os.path.join(str(tmp_path), "bench_r%d.npz" % r)) for r in range(world)]
    assert not np.array_equal(z[0]["q"], z[1]["q"])
    for r in range(world):
        d = wbc_workload.make_tick_inputs(wx, cfg, 40, seed=r, fk=common.OracleFK([wx]), stress=True)
        assert np.array_equal(d["q"], z[r]["q"])
        assert np.array_equal(oracle.tick([wx], [cfg], d, 0.002, 40, want_q_next=False)["qdot"], z[r]["qdot"])
