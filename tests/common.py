"""Shared helpers for the tests: models, configurations of BASELINE.json, oracle-backed FK for input generation."""
import numpy as np

import oracle
import wbc_capi as capi
import wbc_model
import wbc_workload


class OracleFK:
    """fk callable for wbc_workload.make_tick_inputs backed by the CPU oracle."""

    def __init__(self, models, model_id=None):
        self.models, self.model_id = models, model_id

    def __call__(self, q):
        return oracle.fk(self.models, q, self.model_id, want_com=False)["oMf"]

    def com(self, q):
        return oracle.fk(self.models, q, self.model_id)["com"]


def models():
    return wbc_model.load_model("a1_wx200"), wbc_model.load_model("a1_px100_pin_ver")


def config(name, model):
    """BASELINE.json configs: c1/c3 = sim3 switch set, c2 = equality-only, full = every task on."""
    if name in ("c1", "c3"):
        return wbc_model.sim3_config(model)
    if name == "c2":
        return wbc_model.equality_only_config(model)
    if name == "full":   # the warm-up problem of setInitialState: all 6 Cartesian tasks + Tikhonov, bounds only
        return wbc_model.make_config(model, Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=True)
    if name == "everything":  # every task and every constraint type at once (coverage, not a reference preset)
        return wbc_model.make_config(model, Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint="PREV",
                                     task_com=True, cCoM=True, cTrunk=True, cFR=True, cFL=True, cRR=True, cRL=True,
                                     mode="static_reach")
    if name in ("c3_hybrid", "c3_hybrid_clean", "c3_mani"):   # sim3.py:145 sets Joint="HYBRID"; literal = to the letter (C.4)
        return wbc_model.sim3_config(model, Joint="MANI" if name == "c3_mani" else "HYBRID",
                                     posture_literal=not name.endswith("clean"))
    if name == "hybrid_grip_com":  # HYBRID + constraints that DO depend on the arm: the leaked perturbed state shows in C
        return wbc_model.make_config(model, Grip=True, Joint="HYBRID", cCoM=True, cTrunk=True, cFR=True, cFL=True, cRR=True,
                                     cRL=True, cGrip=True, mode="static_reach")
    if name == "c3_nobounds":   # sim3 switch set without the velocity-damper box: the presolve adds no leg-bound rows
        return wbc_model.make_config(model, Grip=True, Joint="PREV", cTrunk=True, cFR=True, cFL=True, cRR=True, cRL=True,
                                     mode="static_reach", use_bounds=False)
    if name == "c3_two_feet":   # only two stance feet: 26 - 6 = 20 unknowns > 16 -> general path, no presolve
        return wbc_model.make_config(model, Grip=True, Joint="PREV", cTrunk=True, cFR=True, cRL=True, mode="static_reach")
    if name == "c3_trunk_task":  # trunk task on top (base-only support: the plan stays enabled, two task blocks)
        return wbc_model.make_config(model, Grip=True, Trunk=True, Joint="PREV", cTrunk=True, cFR=True, cFL=True, cRR=True,
                                     cRL=True, mode="static_reach")
    if name == "c3_custom":
        return wbc_model.sim3_config(model, Joint="CUSTOM")
    raise KeyError(name)


def tick_inputs(model, cfg, B, seed, stress=True, with_rot=False):
    d = wbc_workload.make_tick_inputs(model, cfg, B, seed, OracleFK([model]), stress=stress)
    if with_rot:   # exercise the orientation feed-forward terms with a moving reference
        rng = np.random.default_rng(seed + 1000)
        from scipy.spatial.transform import Rotation as R
        e = rng.uniform(-0.3, 0.3, (B, 5, 3))
        de = rng.normal(0, 1e-3, (B, 5, 3))
        d["ee_ref_rot"] = R.from_euler("xyz", e.reshape(-1, 3)).as_matrix().reshape(B, 5, 9)
        d["ee_prev_rot"] = R.from_euler("xyz", (e - de).reshape(-1, 3)).as_matrix().reshape(B, 5, 9)
        d["trunk_ref_euler"] = d["trunk_ref_euler"] + rng.normal(0, 0.02, (B, 3))
        d["trunk_prev_rot"] = R.from_euler("xyz", d["trunk_ref_euler"] - rng.normal(0, 1e-3, (B, 3))).as_matrix().reshape(B, 9)
    return d


def kkt_residuals(H, g, C, lb, ub, cl, cu, x, act_tol=1e-7):
    """Solver-independent optimality certificate: (primal violation, stationarity residual with sign-correct multipliers)."""
    from scipy.optimize import lsq_linear
    n = len(g)
    viol = 0.0
    if lb is not None:
        viol = max(viol, (lb - x).max(), (x - ub).max())
    if C is not None and len(cl):
        v = C @ x
        viol = max(viol, (cl - v).max(), (v - cu).max())
    r = H @ x + g
    rows, free_sign = [], []
    if lb is not None:
        for k in range(n):
            e = np.zeros(n)
            e[k] = 1
            if lb[k] == ub[k]:
                rows.append(e), free_sign.append(True)
            else:
                if abs(x[k] - lb[k]) < act_tol * max(1, abs(lb[k])):
                    rows.append(e), free_sign.append(False)
                if abs(x[k] - ub[k]) < act_tol * max(1, abs(ub[k])):
                    rows.append(-e), free_sign.append(False)
    if C is not None and len(cl):
        v = C @ x
        for i in range(len(cl)):
            if cl[i] == cu[i]:
                rows.append(C[i]), free_sign.append(True)
            else:
                if abs(v[i] - cl[i]) < act_tol * max(1, abs(cl[i])):
                    rows.append(C[i]), free_sign.append(False)
                if abs(v[i] - cu[i]) < act_tol * max(1, abs(cu[i])):
                    rows.append(-C[i]), free_sign.append(False)
    if rows:
        N = np.array(rows).T
        lo = [-np.inf if f else 0.0 for f in free_sign]
        res = lsq_linear(N, r, bounds=(lo, np.inf), tol=1e-14)
        stat = np.abs(N @ res.x - r).max()
    else:
        stat = np.abs(r).max()
    return viol, stat


def exact_normal_equations(A, b):
    """H = A'A and g = -A'b of double-precision A, b in EXACT rational arithmetic (lists of Fractions): the least-squares problem itself,
    free of the rounding that forming H in doubles adds (1e-5 relative on the posture block of the benchmark tick)."""
    from fractions import Fraction
    m, n = A.shape
    Af = [[Fraction(float(A[i, j])) for j in range(n)] for i in range(m)]
    bf = [Fraction(float(v)) for v in b]
    nz = [[k for k in range(m) if Af[k][i] != 0] for i in range(n)]
    H = [[sum(Af[k][i] * Af[k][j] for k in nz[i]) for j in range(n)] for i in range(n)]
    g = [-sum(Af[k][i] * bf[k] for k in nz[i]) for i in range(n)]
    return H, g


def exact_kkt(H, g, rows, rhs):
    """Exact (rational) solution of  H x + N'lam = -g,  N x = rhs  for double-precision data (or H, g already given as Fractions:
    exact_normal_equations); returns (x, lam) as floats."""
    from fractions import Fraction
    n, p = len(g), len(rhs)
    N = n + p
    M = [[Fraction(0)] * (N + 1) for _ in range(N)]
    for i in range(n):
        for j in range(n):
            M[i][j] = Fraction(H[i][j])
        for j in range(p):
            M[i][n + j] = M[n + j][i] = Fraction(float(rows[j][i]))
        M[i][N] = -Fraction(g[i])
    for j in range(p):
        M[n + j][N] = Fraction(float(rhs[j]))
    for c in range(N):
        piv = max(range(c, N), key=lambda r: abs(M[r][c]))
        assert M[piv][c] != 0, "active rows are dependent"
        M[c], M[piv] = M[piv], M[c]
        inv = 1 / M[c][c]
        for r in range(N):
            if r != c and M[r][c] != 0:
                f = M[r][c] * inv
                M[r] = [a - f * b for a, b in zip(M[r], M[c])]
    sol = [float(M[i][N] / M[i][i]) for i in range(N)]
    return np.array(sol[:n]), np.array(sol[n:])



def exact_ls_optimum(A, b, C, lb, ub, Clb, Cub, x_float):
    """exact_optimum for the least-squares form  min 1/2 |A x - b|^2  (QP_Wrapper.py:17-18: H = A'A, g = -A'b) with H and g formed in
    rational arithmetic from the double-precision A, b — the optimum every correct rounding of H approximates."""
    H, g = exact_normal_equations(np.asarray(A), np.asarray(b))
    return exact_optimum(H, g, C, lb, ub, Clb, Cub, x_float)


def exact_optimum(H, g, C, lb, ub, Clb, Cub, x_float):
    """The exact optimum of the double-precision QP data on the active set read off `x_float`, with the optimality checks
    (primal feasibility, multiplier signs) asserted. Returns x_exact."""
    n = len(g)
    lo, hi = np.concatenate([lb, Clb]), np.concatenate([ub, Cub])
    Nall = np.vstack([np.eye(n), C])
    v = Nall @ x_float
    rows, rhs, kind = [], [], []
    for i in range(len(lo)):
        if lo[i] == hi[i]:
            rows.append(Nall[i]); rhs.append(lo[i]); kind.append(0)
        elif abs(v[i] - lo[i]) < 1e-7 * max(1, abs(lo[i])):
            rows.append(Nall[i]); rhs.append(lo[i]); kind.append(-1)
        elif abs(v[i] - hi[i]) < 1e-7 * max(1, abs(hi[i])):
            rows.append(Nall[i]); rhs.append(hi[i]); kind.append(+1)
    x, lam = exact_kkt(H, g, rows, rhs)
    vx = Nall @ x
    assert (vx >= lo - 1e-9 * np.maximum(1, np.abs(lo))).all() and (vx <= hi + 1e-9 * np.maximum(1, np.abs(hi))).all()
    for k, l in zip(kind, lam):              # H x + g + N'lam = 0: at a lower bound lam <= 0, at an upper bound lam >= 0
        assert not (k == -1 and l > 1e-9 * (1 + abs(l))) and not (k == +1 and l < -1e-9 * (1 + abs(l)))
    return x
