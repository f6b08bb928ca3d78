#!/usr/bin/env python3
"""Regenerate tests/golden/tick_*.npz: seeded tick inputs + the CPU oracle's outputs for BASELINE configs 1/2/3/5.
Runs the oracle only (no reference, no GPU). The reference itself cannot be executed here (pinocchio/qpOASES absent),
so these are the oracle's own regression anchors, not reference outputs (tests/golden/README.md)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: F401,E402  (sets sys.path)
import common  # noqa: E402
import oracle  # noqa: E402

DT = 0.002


def make(name, cfg_name, B, seed, mixed=False):
    wx, px = common.models()
    if mixed:
        models = [wx, px]
        cfgs = [common.config(cfg_name, m) for m in models]
        mid = (np.arange(B) % 2).astype(np.int32)
        parts = [common.tick_inputs(m, c, B, seed + i) for i, (m, c) in enumerate(zip(models, cfgs))]
        d = {k: np.where(mid.reshape((B,) + (1,) * (parts[0][k].ndim - 1)) == 0, parts[0][k], parts[1][k]) for k in parts[0]}
        d["model_id"] = mid
    else:
        models, cfgs = [wx], [common.config(cfg_name, wx)]
        d = common.tick_inputs(wx, cfgs[0], B, seed, with_rot=(cfg_name in ("full", "everything")))
    out = oracle.tick(models, cfgs, d, DT, B)
    asm = oracle.assemble(models, cfgs, d, DT, B) if not mixed else {}
    np.savez_compressed(os.path.join(HERE, "golden", name + ".npz"), **{"in_" + k: v for k, v in d.items()},
                        **{"out_" + k: v for k, v in out.items()}, **{"asm_" + k: v for k, v in asm.items()})
    print(name, "status", np.bincount(out["status"]), "iters", out["iters"].tolist())


def make_rollout(name, cfg_name, B, seed, K):
    """K closed-loop ticks: inputs, per-tick steps and the oracle's final state / gripper trace."""
    wx, _ = common.models()
    cfg = common.config(cfg_name, wx)
    d = common.tick_inputs(wx, cfg, B, seed)
    rng = np.random.default_rng(seed)
    step = np.zeros((B, 5, 3))
    step[:, 4] = rng.normal(0, 1e-4, (B, 3))
    imu = d["q"][:, 3:7].copy()
    out = oracle.rollout([wx], [cfg], d, DT, B, K, ee_target_step=step, imu=imu)
    np.savez_compressed(os.path.join(HERE, "golden", name + ".npz"), **{"in_" + k: v for k, v in d.items()},
                        step=step, imu=imu, ticks=K, **{"out_" + k: v for k, v in out.items()})
    print(name, "status", np.bincount(out["status"]), "iters", out["iters"].tolist())


if __name__ == "__main__":
    make("tick_c3_hybrid", "c3_hybrid", 8, 7)
    make("tick_c3_mani", "c3_mani", 4, 8)
    make_rollout("rollout_c3", "c3", 8, 9, 6)
    make_rollout("rollout_c3_hybrid", "c3_hybrid", 4, 10, 4)
    make("tick_c1", "c1", 1, 1)
    make("tick_c2", "c2", 8, 2)
    make("tick_c3", "c3", 16, 3)
    make("tick_c5_mixed", "c3", 8, 5, mixed=True)
    make("tick_everything", "everything", 8, 6)
