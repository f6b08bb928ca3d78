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


from bench_engine_stub import OracleEngine as _OracleEngine  # noqa: E402  (the same stub `bench.py --gpus 2` is run with below)


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


def test_bench_command_line_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it (what the driver's scaling run types): the parent starts two fresh
    child ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, no exec, no torch or GPU in the parent), forwards rank 0's
    JSON line and exits with the worst child's code. CPU ranks: gloo + the oracle engine stub."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["WBC_BENCH_ENGINE_STUB"] = os.path.join(HERE, "bench_engine_stub.py") + ":OracleEngine"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "3",
                        "--batch", "40"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # ONE line, rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 80 and line["config"]["engine"] == "oracle-cpu"
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 80.0) < 1e-6
    assert "cpu_baseline" not in line                                 # rank 0 at N = 1 only


def test_bench_launcher_returns_the_worst_rank_exit_code(tmp_path):
    """a rank that fails (here: an engine stub that does not exist) makes the launcher return non-zero instead of hanging"""
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["WBC_BENCH_ENGINE_STUB"] = os.path.join(str(tmp_path), "missing.py") + ":Nope"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_under_an_external_launcher_still_works(tmp_path):
    """the torch.distributed.run path: WORLD_SIZE already set -> no children, this process IS a rank (world 1 here)"""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29563",
               WBC_BENCH_ENGINE_STUB=os.path.join(HERE, "bench_engine_stub.py") + ":OracleEngine")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--repeats", "1",
                        "--batch", "16", "--no-cpu-baseline", "--rollout-ticks", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 16
