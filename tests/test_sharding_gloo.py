"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py is 'contiguous shards, no data-path collective, max of the
elapsed times' (SURVEY.md §8e). Each rank solves its shard (the oracle stands in for the GPU here) and the union must be
exactly the single-process result; the only collectives are the barrier and the MAX all-reduce of the timing."""
import os
import sys

import numpy as np
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
