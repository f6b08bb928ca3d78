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


def test_bench_command_line_starts_two_ranks_on_the_gpu_box():
    """`python bench.py --gpus 2` with no launcher around it, on real hardware: the parent (which never touches the GPU) starts two child ranks with the
    real GpuEngine; on a one-GPU box both share device 0 and meet over gloo (WBC_BENCH_SHARE_GPU: RCCL refuses two ranks on one device), on a box
    with two or more GPUs each rank takes its own and they meet over RCCL. One JSON line, n_gpus 2, the aggregate of both shards, rc 0."""
    import json
    import subprocess
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    shared = torch.cuda.device_count() < 2
    if shared:
        env["WBC_BENCH_SHARE_GPU"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "3", "--batch", "8192"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]           # nothing but rank 0's line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16384 and line["config"]["engine"] == "hip"
    assert line["config"]["backend"].startswith("gloo" if shared else "nccl") and "sim3p" in line["config"]["kernel_path"]
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 16384.0) < 1e-3 and line["value"] > 1e6
    assert "cpu_baseline" not in line and line["scaling"] == "weak"
