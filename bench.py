#!/usr/bin/env python3
"""bench.py — WBC control ticks per second on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is ONE pass of the fused hot path (wbc_tick: FK -> Jacobians -> task stack -> H, g, C, bounds -> QP -> q̇)
over one batch of synthetic robot states already resident in HBM. Workload (config.workload): BASELINE configs[2],
"Batch=65536 A1+wx200 with 4-foot contact + ... inequalities, 1 MI355X" = SURVEY.md §8(d) C3: the reference's sim3
switch set (Grip task + posture; 12 contact equalities, 4 trunk-box rows, 26 velocity-damper bounds, 3 locked DoF).
With --gpus N (launched under torch.distributed.run) every rank runs its own 65536-instance shard on its own GPU —
no data-path collective — and the job value is the sum: scaling "weak".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]

Extra objects on the line: "roofline" (algorithmic HBM bytes/tick x ticks / kernel time from HIP events on the launch
stream, against 8 TB/s) and "cpu_baseline" (the CPU oracle, oracle/wbc_oracle.c, timed on this host's cores on a bounded
sample of the same inputs — the oracle is only the yardstick/checker here, never the thing measured).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd"))

import numpy as np  # noqa: E402

# q 27*8 + targets 18*8 in, q̇ 26*8 + status 4 out
ALGO_BYTES_PER_TICK = 572
# what wbc_tick really streams per instance (adds prev targets 144 B, box centre 32 B, trunk refs 96 B, iters 4 B)
ACTUAL_BYTES_PER_TICK = 216 + 120 + 120 + 32 + 208 + 4 + 4
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector = matrix peak
ALGO_FLOP_PER_TICK = 5.5e4     # + ~1e4 per working-set change beyond the equalities


def pmc_traffic_bytes():
    """HBM bytes per bench step from the committed rocprofv3 PMC passes of this same command (profiles/r01_pmc_summary_v12.txt:
    FETCH_SIZE and WRITE_SIZE in KiB, separate --pmc runs, tools/gpu_profile.sh), summed over the kernels one step launches
    (the sim3 tick kernel + the deferred pass). 8-byte-per-lane accesses: the gfx950 x2
    FETCH_SIZE correction for 16-byte streams does not apply (uncalibrated width)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary_v12.txt")
    try:
        total = 0.0
        for line in open(path):
            if "wbc_tick_sim3_kernel" in line or "wbc_tick_deferred_kernel" in line:
                for key in ("FETCH_SIZE", "WRITE_SIZE"):
                    if " %s " % key in line:
                        total += float(line.split("per_dispatch=")[1])
        return total * 1024.0 if total else None
    except Exception:
        return None


def pmc_issue_busy():
    """VALU / LDS busy fractions of the sim3 tick kernel from the same committed PMC passes: SQ_ACTIVE_INST_VALU and
    SQ_LDS_IDX_ACTIVE per CU-cycle (GRBM_GUI_ACTIVE counts the 8 XCDs' cycles; 256 CUs)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary_v12.txt")
    try:
        v = {}
        for line in open(path):
            if "wbc_tick_sim3_kernel" in line:
                for key in ("SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVES",
                            "SQ_LDS_BANK_CONFLICT"):
                    if " %s " % key in line:
                        v[key] = float(line.split("per_dispatch=")[1])
        cu_cycles = 256.0 * v["GRBM_GUI_ACTIVE"] / 8.0
        return {"valu_busy": v["SQ_ACTIVE_INST_VALU"] / cu_cycles, "lds_busy": v["SQ_LDS_IDX_ACTIVE"] / cu_cycles,
                "valu_insts_per_tick": v["SQ_INSTS_VALU"] / v["SQ_WAVES"], "lds_insts_per_tick": v["SQ_INSTS_LDS"] / v["SQ_WAVES"],
                "salu_insts_per_tick": v["SQ_INSTS_SALU"] / v["SQ_WAVES"],
                "lds_bank_conflict_rate": v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"],
                "lds_bytes_per_instance": 13056, "vgprs": 168, "waves_per_simd": 3,      # tools/kernel_stats.sh (wbc_tick_sim3_kernel)
                "source": "profiles/r01_pmc_summary_v12.txt"}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--cpu-sample", type=int, default=0, help="instances for the CPU baseline (0 = auto, ~15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--jtj-mfma", type=int, default=0)
    ap.add_argument("--rollout-ticks", type=int, default=10, help="closed-loop ticks of the extra wbc_rollout measurement (0 = skip)")
    ap.add_argument("--posture", default="PREV", choices=["PREV", "HYBRID", "MANI"],
                    help="posture mode of the tick (default PREV = the BASELINE workload; HYBRID is what sim3.py:145 sets)")
    args = ap.parse_args()

    import torch
    import wbc_model
    import wbc_workload
    from wbc_batch import WbcBatch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    B, DT = args.batch, 0.002
    model = wbc_model.load_model("a1_wx200")
    cfg = wbc_model.sim3_config(model, Joint=args.posture)
    bt = WbcBatch(model, B, device_id=local)
    bt.configure(cfg)
    if args.jtj_mfma:
        bt.set_option("jtj_mfma", 1)
    if os.environ.get("WBC_PRESOLVE") is not None:        # diagnostic A/B: 0 = general path only (no structural presolve, no sim3 kernel)
        bt.set_option("presolve", int(os.environ["WBC_PRESOLVE"]))
    if os.environ.get("WBC_SIM3_KERNEL") is not None:     # diagnostic A/B: 0 = presolve inside the general kernel
        bt.set_option("sim3_kernel", int(os.environ["WBC_SIM3_KERNEL"]))
    if os.environ.get("WBC_DBG_ALIAS"):
        bt.set_option("dbg_alias_inputs", 1)

    class GpuFK:   # inputs are placed on the robot with the product's own FK (never the oracle's)
        def __call__(self, q):
            return bt.fk(q, want=("oMf",))["oMf"]

    host_in = wbc_workload.make_tick_inputs(model, cfg, B, seed=rank, fk=GpuFK(), stress=True)
    dev_in = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in host_in.items()}
    dev_out = dict(qdot=torch.zeros((B, 26), dtype=torch.float64, device=dev),
                   status=torch.zeros(B, dtype=torch.int32, device=dev), iters=torch.zeros(B, dtype=torch.int32, device=dev))
    step = bt.make_tick_call(dev_in, dev_out, DT)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                      # same (current) stream the C-ABI launches on
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # closed loop: K chained ticks on the device = tick + updateState/trunkWorldPos + reference-state advance
    closed = None
    if args.rollout_ticks > 0 and world == 1:     # extra leg, single-GPU runs only
        step_t = torch.zeros((B, 5, 3), dtype=torch.float64, device=dev)
        step_t[:, 4, 0] = 1e-4
        bt.rollout(dev_in, DT, 2, ee_target_step=step_t, want_trace=False)
        barrier()
        t1 = time.perf_counter()
        ro = bt.rollout(dev_in, DT, args.rollout_ticks, ee_target_step=step_t, want_trace=False)
        barrier()
        t_roll = time.perf_counter() - t1
        closed = {"value": float(B) * args.rollout_ticks / t_roll, "unit": "closed-loop ticks/s per GPU", "ticks": args.rollout_ticks,
                  "ms_per_tick": 1e3 * t_roll / args.rollout_ticks, "optimal_frac": float((ro["status"] == 0).double().mean().item()),
                  "what": "wbc_rollout: wbc_tick + wbc_update_state (FK + trunkWorldPos) + prev-target state advance, targets moving 0.1 mm/tick"}

    status = dev_out["status"].cpu().numpy()
    iters = dev_out["iters"].cpu().numpy()
    qdot = dev_out["qdot"].cpu().numpy()

    if rank == 0:
        ticks = float(B) * world * args.steps
        value = ticks / elapsed
        ach = ALGO_BYTES_PER_TICK * B / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "wbc_qp_solves_per_sec", "value": value, "unit": "ticks/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] / SURVEY C3: A1+wx200 (nq 27, nv 26), sim3 switch set: Grip task + " + args.posture + " posture, "
                                   "12 contact equalities + 4 trunk-box rows + 26 damper bounds (3 locked), m=32 p=16 n=26",
                       "batch_per_gpu": B, "global_batch": B * world, "dt": DT, "parallelism": "shard%d (no collective)" % world,
                       "jtj": "mfma_f64" if args.jtj_mfma else "valu_f64"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": pmc_traffic_bytes(), "traffic_unit": "bytes/launch (PMC, batch 65536)", "kernel": "wbc_tick_sim3_kernel (+ wbc_tick_deferred_kernel)", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_tick": ALGO_BYTES_PER_TICK, "streamed_bytes_per_tick": ACTUAL_BYTES_PER_TICK,
                         "fp64_tflops": ALGO_FLOP_PER_TICK * B / (kernel_ms * 1e-3) / 1e12,
                         "fp64_frac": ALGO_FLOP_PER_TICK * B / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                         "issue": pmc_issue_busy(),
                         "note": "tiny-dense LDS-resident fp64 work: VALU- and LDS-issue bound, neither HBM nor MFMA is approachable (SURVEY.md §8d); see profiles/"},
            "solver": {"optimal_frac": float((status == 0).mean()), "iters_mean": float(iters.mean()),
                       "iters_p95": float(np.percentile(iters, 95)), "iters_max": int(iters.max())},
        }
        if closed is not None:
            line["closed_loop"] = closed
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0 at N = 1 only
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle   # checker / yardstick only
            cores = len(os.sched_getaffinity(0))
            n = args.cpu_sample or min(B, 16384)
            sub = {k: v[:n] for k, v in host_in.items()}
            oracle.tick([model], [cfg], {k: v[:256] for k, v in host_in.items()}, DT, 256, nthreads=cores, want_q_next=False)
            reps, t_cpu, ref = 0, 0.0, None
            while t_cpu < 10.0 and reps < 64:
                t1 = time.perf_counter()
                ref = oracle.tick([model], [cfg], sub, DT, n, nthreads=cores, want_q_next=False)
                t_cpu += time.perf_counter() - t1
                reps += 1
            t1 = time.perf_counter()
            oracle.tick([model], [cfg], {k: v[:2048] for k, v in host_in.items()}, DT, 2048, nthreads=1, want_q_next=False)
            one = 2048 / (time.perf_counter() - t1)
            ok = (ref["status"] == 0) & (status[:n] == 0)
            line["cpu_baseline"] = {"value": n * reps / t_cpu, "unit": "ticks/s", "cores": cores, "kind": "port",
                                    "sample": "first %d instances of rank 0's batch x %d repeats (%.1f s), OpenMP over %d threads; "
                                              "single-thread rate %.0f ticks/s; reference design rate 500 ticks/s (paced, not measured)" % (n, reps, t_cpu, cores, one),
                                    "single_thread": one}
            line["accuracy"] = {"qdot_max_abs_err_vs_cpu": float(np.abs(ref["qdot"] - qdot[:n])[ok].max()),
                                "status_agree_frac": float((ref["status"] == status[:n]).mean()), "tolerance": 1e-5, "instances": int(n)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    bt.close()


if __name__ == "__main__":
    main()
