"""GPU: bench.py's RCCL branch executed once on ONE GPU (world size 1): `torch.distributed` with backend "nccl" (= RCCL on ROCm),
the barrier and the MAX all-reduce on a device tensor, and the rank-0 JSON line — with the real GpuEngine — so that the code an 8-GPU node
runs has executed on hardware before such a node ever sees it. (The N > 1 logic itself is covered over gloo: test_sharding_gloo.py.)"""
import argparse
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_rank_body_over_rccl_at_world_size_one():
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29600 + os.getpid() % 300)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = argparse.Namespace(batch=8192, steps=3, warmup=1, repeats=3, posture="PREV", jtj_mfma=-1)
    engine = bench.GpuEngine(args, 0)
    comm = bench.Comm("nccl", 0, 1, engine.dev, force=True)
    try:
        assert comm.dist is not None and dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        comm.barrier()
        assert comm.max([1.5, -2.0]) == [1.5, -2.0]                     # all_reduce(MAX) on a device tensor through RCCL
        line, res = bench.run_rank(args, comm, engine, bench.make_inputs)
        assert line["n_gpus"] == 1 and line["steps"] == 3 and line["repeats"]["n"] == 3 and line["value"] > 1e6
        assert line["config"]["engine"] == "hip" and "sim3p" in line["config"]["kernel_path"]
        assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 8192.0) < 1e-3
        assert (res["status"] == 0).mean() > 0.8
        t = torch.ones(4, device=engine.dev)
        dist.all_reduce(t)
        assert float(t.sum()) == 4.0
    finally:
        comm.close()
        engine.close()
    assert not dist.is_initialized()
