#!/usr/bin/env python3
"""Per-phase shader cycles of the fused tick (diagnostic build only):
    WBC_HIP_LIB=mech5845m-wbc-for-legged-manipulator_amd/csrc/build/libwbc_hip_prof.so python tools/phase_cycles.py [B] [mfma]
Shares, not run time: the stamps fence the phases (cdna_hip_programming.md §7 'In-kernel stamps')."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mech5845m-wbc-for-legged-manipulator_amd"))
import torch
import wbc_model, wbc_workload
from wbc_batch import WbcBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mfma = len(sys.argv) > 2 and sys.argv[2] == "mfma"
model = wbc_model.load_model("a1_wx200")
_name = os.environ.get("WBC_CFG", "sim3")          # WBC_CFG=c2: BASELINE configs[1]; full / everything / ...: tests/common.py's named configurations
if _name in ("sim3", "c2"):
    cfg = {"c2": wbc_model.equality_only_config, "sim3": wbc_model.sim3_config}[_name](model)
else:
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    import common
    cfg = common.config(_name, model)
bt = WbcBatch(model, B)
assert b"PROFILE" in bt.lib.wbc_version(), "set WBC_HIP_LIB to the profile build"
bt.configure(cfg)
bt.set_option("jtj_mfma", int(mfma))
for opt in ("presolve", "presolve_orth", "sim3_kernel"):
    if os.environ.get("WBC_" + opt.upper()) is not None:
        bt.set_option(opt, int(os.environ["WBC_" + opt.upper()]))
class _Fk:
    def __call__(self, q): return bt.fk(q, want=("oMf",))["oMf"]
    def com(self, q): return bt.fk(q, want=("com",))["com"]
fk = _Fk()
if _name in ("full", "everything"):
    d = common.tick_inputs(model, cfg, B, 0, with_rot=True)
else:
    d = wbc_workload.make_tick_inputs(model, cfg, B, 0, fk)
dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
bt.debug_cycles()          # arm + reset
for _ in range(3):
    bt.tick(dev, 0.002)
torch.cuda.synchronize()
c = bt.debug_cycles()
n = max(1, c[0])
names = ["ticks", "fk_jac", "task_stack", "cholesky", "inverse_x0", "eq_phase", "ineq_phase", "output"]
out = {"ticks": c[0], "iters_mean": c[8] / n}
for i in range(1, 8):
    out[names[i]] = c[i] / n
out["total"] = sum(c[1:8]) / n
out["task_sub"] = dict(rows_targets=c[9] / n, jtj_posture=c[10] / n, constraint_rows=c[11] / n, damper=c[12] / n)
e = max(1, c[14])
out["presolve"] = dict(engaged_frac=c[14] / n, cycles=c[13] / e, G=c[16] / e, Hred_gred=c[17] / e, C_rows=c[18] / e, store=c[19] / e)
out["entry_to_start"] = c[23] / n
out["fk_sub"] = dict(levels=c[20] / n, frames_columns=c[21] / n, com_trunk=c[22] / n)
print(json.dumps(out))
