"""CPU: certifies the oracle's QP (Goldfarb–Idnani restatement) independently of how it iterates: KKT residuals with
sign-correct multipliers, scipy.optimize cross-solves, the toy problem of the reference's tests_NOT_FOR_USE/qp_tests.py,
infeasibility detection; and checks the algebra of the wavefront variant (tests/gi_variant.py) against it."""
import numpy as np
import pytest
from scipy.optimize import minimize

import common
import gi_variant
import oracle

DT = 0.002


def _random_qp(rng, n=None, p=None):
    n = n or int(rng.integers(3, 27))
    p = int(rng.integers(0, 16)) if p is None else p
    A = rng.normal(size=(n + int(rng.integers(0, 10)), n))
    H = A.T @ A + 1e-3 * np.eye(n)
    g = rng.normal(size=n) * 4
    C = rng.normal(size=(p, n)) if p else None
    lb, ub = -rng.uniform(0.02, 0.8, n), rng.uniform(0.02, 0.8, n)
    k = rng.integers(0, n, 2)
    lb[k] = ub[k] = 0
    if p:
        cl, cu = -rng.uniform(0.02, 0.8, p), rng.uniform(0.02, 0.8, p)
        ne = min(p, max(0, (n - 3) // 3))
        cl[:ne] = cu[:ne] = rng.normal(size=ne) * 0.1
    else:
        cl = cu = None
    return H, g, C, lb, ub, cl, cu


def test_kkt_certificate_on_random_qps():
    rng = np.random.default_rng(0)
    solved = 0
    for _ in range(300):
        H, g, C, lb, ub, cl, cu = _random_qp(rng)
        x, st, it = oracle.qp_solve(H, g, C, lb, ub, cl, cu)
        if st != 0:
            continue
        solved += 1
        viol, stat = common.kkt_residuals(H, g, C, lb, ub, cl, cu, x)
        assert viol < 1e-8 and stat < 1e-8
    assert solved > 280


def test_against_scipy_slsqp():
    rng = np.random.default_rng(1)
    agreed = 0
    for _ in range(40):
        H, g, C, lb, ub, cl, cu = _random_qp(rng, n=8, p=4)
        x, st, _ = oracle.qp_solve(H, g, C, lb, ub, cl, cu)
        if st != 0:
            continue
        cons = []
        for i in range(4):
            if cl[i] == cu[i]:
                cons.append({"type": "eq", "fun": lambda z, i=i: C[i] @ z - cl[i]})
            else:
                cons.append({"type": "ineq", "fun": lambda z, i=i: C[i] @ z - cl[i]})
                cons.append({"type": "ineq", "fun": lambda z, i=i: cu[i] - C[i] @ z})
        res = minimize(lambda z: 0.5 * z @ H @ z + g @ z, np.zeros(8), jac=lambda z: H @ z + g, bounds=list(zip(lb, ub)),
                       constraints=cons, method="SLSQP", options={"ftol": 1e-13, "maxiter": 500})
        if not res.success:      # SLSQP's line search gives up on some instances (fixed variables): not a verdict on the oracle
            continue
        agreed += 1
        f_or, f_sp = 0.5 * x @ H @ x + g @ x, res.fun
        assert f_or <= f_sp + 1e-9            # the oracle is at least as good as SLSQP ...
        assert np.abs(x - res.x).max() < 1e-5  # ... and they agree on the unique minimiser
    assert agreed >= 15


def test_reference_toy_qp():
    """tests_NOT_FOR_USE/qp_tests.py:4-13, restated literally: P = M'M, q = [3,2,3] M, G x <= [1,1,1], x1+x2+x3 = 1.
    The reference holds no expected output for it; SURVEY.md §8c computed x* ~ [0, 0, 1] with scipy. Certified here by
    KKT residuals and an SLSQP cross-solve."""
    M = np.array([[1.0, 2.0, 0.0], [-8.0, 3.0, 2.0], [0.0, 1.0, 1.0]])
    P, qv = M.T @ M, np.array([3.0, 2.0, 3.0]) @ M
    G = np.array([[1.0, 2.0, 1.0], [2.0, 0.0, 1.0], [-1.0, 2.0, -1.0]])
    C = np.vstack([G, np.ones((1, 3))])
    cl = np.array([-1e30, -1e30, -1e30, 1.0])
    cu = np.array([1.0, 1.0, 1.0, 1.0])
    x, st, _ = oracle.qp_solve(P, qv, C, None, None, cl, cu)
    assert st == 0
    viol, stat = common.kkt_residuals(P, qv, C, np.full(3, -1e30), np.full(3, 1e30), cl, cu, x)
    assert viol < 1e-9 and stat < 1e-9
    cons = [{"type": "eq", "fun": lambda z: z.sum() - 1}] + [{"type": "ineq", "fun": lambda z, i=i: 1 - G[i] @ z} for i in range(3)]
    res = minimize(lambda z: 0.5 * z @ P @ z + qv @ z, np.zeros(3), jac=lambda z: P @ z + qv, constraints=cons, method="SLSQP",
                   options={"ftol": 1e-14})
    assert np.abs(res.x - x).max() < 1e-6      # (SLSQP may flag its line search at the vertex; its point is what is compared)
    assert np.abs(x - np.array([0.0, 0.0, 1.0])).max() < 0.05


def test_infeasible_and_nonconvex_are_reported():
    H, g = np.eye(3), np.zeros(3)
    C = np.array([[1.0, 0, 0], [1.0, 0, 0]])
    x, st, _ = oracle.qp_solve(H, g, C, -np.ones(3), np.ones(3), np.array([0.5, -2.0]), np.array([2.0, 0.0]))
    assert st == 2                                                     # 0.5 <= x0 and x0 <= 0
    x, st, _ = oracle.qp_solve(np.diag([1.0, -1.0, 1.0]), g, None, -np.ones(3), np.ones(3))
    assert st == 3


def test_wavefront_variant_algebra_matches_oracle():
    """Householder add / explicit R^-1 / row-of-T drop (what the HIP kernel runs) == textbook Givens + R."""
    rng = np.random.default_rng(2)
    n_ok = 0
    for _ in range(150):
        H, g, C, lb, ub, cl, cu = _random_qp(rng)
        x, st, it = oracle.qp_solve(H, g, C, lb, ub, cl, cu)
        x2, st2, it2 = gi_variant.solve(H, g, C, lb, ub, cl, cu)
        assert st == st2
        if st == 0:
            n_ok += 1
            assert np.abs(x - x2).max() < 1e-10 and it == it2
    assert n_ok > 130


def test_tick_problems_satisfy_kkt():
    """The actual WBC problems (cond(H) ~ 3e9): certificate on the oracle's tick outputs for configs 2 and 3."""
    wx, _ = common.models()
    for name, B in (("c3", 48), ("c2", 16), ("everything", 16)):
        cfg = common.config(name, wx)
        d = common.tick_inputs(wx, cfg, B, seed=5, with_rot=(name == "everything"))
        a = oracle.assemble([wx], [cfg], d, DT, B)
        out = oracle.tick([wx], [cfg], d, DT, B)
        assert (out["status"] == 0).mean() > 0.9
        for b in np.nonzero(out["status"] == 0)[0]:
            x = out["qdot"][b]
            viol, stat = common.kkt_residuals(a["H"][b], a["g"][b], a["C"][b], a["lb"][b], a["ub"][b], a["Clb"][b], a["Cub"][b], x, act_tol=1e-6)
            scale = 1 + np.abs(a["g"][b]).max()
            assert viol < 1e-7 and stat / scale < 1e-7, (name, b, viol, stat)


def test_adversarial_qp_fixture():
    """tests/golden/qp_cases.npz: every bound active, duplicated and contradictory equality rows, an
    infeasible box, locked DoF — the oracle reproduces its stored answers, and each feasible answer carries a
    solver-independent KKT certificate with multipliers of the right sign."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qp_cases.npz"))
    names = [str(s) for s in z["names"]]
    x, st, it = oracle.qp_solve(z["H"], z["g"], z["C"], z["lb"], z["ub"], z["Clb"], z["Cub"])
    assert (st == z["status"]).all() and (it == z["iters"]).all()
    assert dict(zip(names, st.tolist())) == {"all_bounds_active": 0, "bounds_only": 0, "contradictory_equalities": 2,
                                             "duplicate_equalities": 0, "infeasible_box": 2, "locked_and_mixed": 0,
                                             "unconstrained": 0}
    for i, nm in enumerate(names):
        if st[i] != 0:
            continue
        assert np.abs(x[i] - z["x"][i]).max() < 1e-10, nm
        viol, stat = common.kkt_residuals(z["H"][i], z["g"][i], z["C"][i], z["lb"][i], z["ub"][i], z["Clb"][i], z["Cub"][i], x[i])
        scale = max(1.0, np.abs(z["H"][i] @ x[i]).max())
        assert viol < 1e-8 and stat < 1e-7 * scale, (nm, viol, stat)
    i = names.index("all_bounds_active")
    assert (np.abs(z["active"][i][:26]) == 1).all()                      # every variable sits on a bound
    i = names.index("unconstrained")
    assert np.abs(x[i] + np.linalg.solve(z["H"][i], z["g"][i])).max() < 1e-9 and not z["active"][i].any()


def test_oracle_solution_is_the_exact_optimum_on_benchmark_ticks():
    """Strongest solver-independent pin available here: for ticks of the benchmark configuration (cond(H) ~ 3e9), take the active set
    the oracle ended with, solve the KKT system EXACTLY (rational arithmetic) and check that (i) this point is primal feasible,
    (ii) every inequality multiplier has the right sign — so it IS the unique optimum (H > 0) — and (iii) the oracle's answer is on it.
    Two exact problems are in play: the least-squares problem itself (H = A'A, g = -A'b formed in rationals from the doubles A, b) and
    the QP of the ROUNDED H, g that QP_Wrapper.py:17-18 hands its solver. They differ by ~1e-7 here (the rounding of fl(A'A) on the
    1.5e-9 posture block) — as much as the plain dual method's own error. The oracle's iterative refinement (qp_refine: the analogue of
    QP_Wrapper.py:37 numRefinementSteps, residual from A and b) lands on the FORMER to 1e-8; without it the error is ~1e-6."""
    import wbc_model
    wx = wbc_model.load_model("a1_wx200")
    cfg = common.config("c3", wx)
    B = 8
    d = common.tick_inputs(wx, cfg, B, seed=71)
    a = oracle.assemble([wx], [cfg], d, 0.002, B)
    t = oracle.tick([wx], [cfg], d, 0.002, B)
    old = oracle.set_refine_steps(0)
    try:
        t0 = oracle.tick([wx], [cfg], d, 0.002, B)
    finally:
        oracle.set_refine_steps(old)
    assert old == 1 and (t0["status"] == t["status"]).all() and (t0["iters"] == t["iters"]).all()
    e_ref, e_plain, e_round = [], [], []
    for b in range(B):
        if t["status"][b] != 0:
            continue
        args = (a["C"][b], a["lb"][b], a["ub"][b], a["Clb"][b], a["Cub"][b], t["qdot"][b])
        x_ls = common.exact_ls_optimum(a["A"][b], a["b"][b], *args)
        x_h = common.exact_optimum(a["H"][b], a["g"][b], *args)
        e_ref.append(np.abs(t["qdot"][b] - x_ls).max())
        e_plain.append(np.abs(t0["qdot"][b] - x_ls).max())
        e_round.append(np.abs(x_h - x_ls).max())
    print("refined %.2e  plain %.2e  exact(rounded H) - exact(least squares) %.2e" % (max(e_ref), max(e_plain), max(e_round)))
    assert len(e_ref) >= 6 and max(e_ref) < 1e-8, e_ref            # (1e-7 asked; 1.4e-9 measured)
    assert max(e_plain) < 2e-6 and max(e_plain) > 20 * max(e_ref)   # what the refinement buys
    assert max(e_round) > 10 * max(e_ref)                           # and why its residual must not come from fl(A'A)


def test_kernel_refinement_step_restated_in_numpy():
    """The refinement the kernels run (csrc/wbc_common.h qp_refine: R rebuilt from the final J and the active constraints, multipliers by back
    substitution, r1 = -(grad f - N'u) with grad f = A'(A x - b), correction through J) stated in plain numpy (gi_variant.refine_step) on top of
    the numpy model of the kernels' dual method: on benchmark ticks — 12 contact equalities, 3 fixed variables, cond(H) ~ 3e9 — it lands
    on the exact least-squares optimum and on the oracle's refined answer, from 1e-6 away; with the residual taken from H instead it does not."""
    import wbc_model
    wx = wbc_model.load_model("a1_wx200")
    cfg = common.config("c3", wx)
    B = 6
    d = common.tick_inputs(wx, cfg, B, seed=72)
    a = oracle.assemble([wx], [cfg], d, 0.002, B)
    t = oracle.tick([wx], [cfg], d, 0.002, B)
    worst_ref, worst_plain, worst_h = 0.0, 0.0, 0.0
    for b in range(B):
        if t["status"][b] != 0:
            continue
        A, bb, H, g = a["A"][b], a["b"][b], a["H"][b], a["g"][b]
        args = (a["C"][b], a["lb"][b], a["ub"][b], a["Clb"][b], a["Cub"][b])
        x0, s0, it0 = gi_variant.solve(H, g, *args)
        x1, s1, it1 = gi_variant.solve(H, g, *args, neg_grad=lambda x: A.T @ (bb - A @ x))
        xh, sh, ith = gi_variant.solve(H, g, *args, neg_grad=lambda x: -(H @ x + g))
        assert s0 == s1 == sh == 0 and it0 == it1 == ith == t["iters"][b]
        x_ls = common.exact_ls_optimum(A, bb, *args, x1)
        worst_ref = max(worst_ref, np.abs(x1 - x_ls).max(), np.abs(x1 - t["qdot"][b]).max())
        worst_plain = max(worst_plain, np.abs(x0 - x_ls).max())
        worst_h = max(worst_h, np.abs(xh - x_ls).max())
    print("numpy model: refined %.2e, plain %.2e, refined with the residual from H %.2e" % (worst_ref, worst_plain, worst_h))
    assert worst_ref < 1e-8 and worst_plain > 20 * worst_ref and worst_h > 20 * worst_ref


def test_refinement_on_the_qp_entry_points():
    """QP(A, b, ...) (QP_Wrapper.py:10-53 -> oracle.qp_solve_ls: least-squares residual) reaches the exact least-squares optimum on an
    ill-conditioned random problem with active bounds and rows; QP(H, g) (oracle.qp_solve) is not refined — its residual could only come from H,
    whose rounding is the error (gi_variant's numpy model shows it: test_kernel_refinement_step_restated_in_numpy) — and stays at the plain method's
    distance from the exact optimum of ITS data; refinement never changes status or iteration count."""
    rng = np.random.default_rng(3)
    n, m, p = 10, 14, 4
    worst_ls, worst_h, worst_plain = 0.0, 0.0, 0.0
    for trial in range(12):
        A = np.vstack([rng.normal(size=(4, n)), 3e-5 * np.eye(n)])          # four O(1) rows + a 1e-9 Tikhonov block: cond(H) ~ 1e10
        b = np.concatenate([rng.normal(size=4), 3e-5 * rng.normal(size=n)])
        C = rng.normal(size=(p, n))
        lb, ub = -rng.uniform(0.5, 2.0, n), rng.uniform(0.5, 2.0, n)
        cl, cu = -rng.uniform(0.1, 1.0, p), rng.uniform(0.1, 1.0, p)
        H, g = A.T @ A, -A.T @ b
        x, st, it = oracle.qp_solve_ls(A, b, C, lb, ub, cl, cu)
        xh, sth, ith = oracle.qp_solve(H, g, C, lb, ub, cl, cu)
        old = oracle.set_refine_steps(0)
        try:
            x0, st0, it0 = oracle.qp_solve(H, g, C, lb, ub, cl, cu)
        finally:
            oracle.set_refine_steps(old)
        assert st == st0 == sth and it == it0 == ith
        if st != 0:
            continue
        x_ls = common.exact_ls_optimum(A, b, C, lb, ub, cl, cu, x)
        x_h = common.exact_optimum(H, g, C, lb, ub, cl, cu, xh)
        worst_ls = max(worst_ls, np.abs(x - x_ls).max())
        worst_h = max(worst_h, np.abs(xh - x_h).max())
        worst_plain = max(worst_plain, np.abs(x0 - x_h).max())
    print("ls-refined vs exact ls %.2e; H-refined vs exact(H) %.2e; plain vs exact(H) %.2e" % (worst_ls, worst_h, worst_plain))
    assert worst_ls < 1e-9 and worst_h <= 2 * worst_plain + 1e-12


def test_warm_started_variant_reaches_the_same_optimum():
    """tests/gi_variant.py solve_v3 (the warm start the kernels run): seeded with (a) the cold solve's own
    final working set, (b) that of a perturbed problem (the previous tick), (c) garbage, it returns the cold optimum; (a) needs no
    dual iteration at all, and wrong seeds are removed by the restoration step, never kept."""
    rng = np.random.default_rng(77)
    n, p = 14, 10
    cold_it, warm_it, warm_prev_it = 0, 0, 0
    for trial in range(60):
        A = rng.normal(size=(n + 4, n))
        H = A.T @ A + 1e-3 * np.eye(n)
        g = rng.normal(size=n) * 4
        C = rng.normal(size=(p, n))
        lb, ub = -rng.uniform(0.02, 0.5, n), rng.uniform(0.02, 0.5, n)
        lb[-1] = ub[-1] = 0.0                                     # a locked variable
        cl, cu = -rng.uniform(0.02, 0.5, p), rng.uniform(0.02, 0.5, p)
        cl[0] = cu[0] = 0.05                                      # an equality row
        xr, sr, _ = oracle.qp_solve(H[None], g[None], C[None], lb[None], ub[None], cl[None], cu[None])
        xr, sr = xr[0], int(sr[0])
        x0, s0, it0, ws0 = gi_variant.solve_v3(H, g, C, lb, ub, cl, cu)
        assert s0 == sr
        if sr != 0:
            continue
        assert np.abs(x0 - xr).max() < 1e-9
        cold_it += it0
        # (a) its own working set: everything is seeded, nothing left to do
        x1, s1, it1, ws1 = gi_variant.solve_v3(H, g, C, lb, ub, cl, cu, seeds=ws0)
        assert s1 == 0 and np.abs(x1 - xr).max() < 1e-9 and sorted(ws1) == sorted(ws0)
        # without the seed filter: the two equalities + one step per seed, nothing else (with it, a seed that the equalities-only
        # minimiser is far from is turned away and comes back through a dual iteration)
        x1n, s1n, it1n, _ = gi_variant.solve_v3(H, g, C, lb, ub, cl, cu, seeds=ws0, far=None)
        assert s1n == 0 and np.abs(x1n - xr).max() < 1e-9 and it1n == 2 + len(ws0)
        warm_it += it1
        # (b) the working set of the "previous tick" (g moved a little)
        xp, sp, itp, wsp = gi_variant.solve_v3(H, g + rng.normal(size=n) * 0.15, C, lb, ub, cl, cu)
        if sp == 0:
            x2, s2, it2, _ = gi_variant.solve_v3(H, g, C, lb, ub, cl, cu, seeds=wsp)
            assert s2 == 0 and np.abs(x2 - xr).max() < 1e-9
            warm_prev_it += it2 - it0
        # (c) garbage: random constraints on random sides, duplicates, equalities, out-of-range ids
        junk = [(int(rng.integers(-2, n + p + 2)), int(rng.integers(0, 2))) for _ in range(rng.integers(1, 12))]
        x3, s3, _, _ = gi_variant.solve_v3(H, g, C, lb, ub, cl, cu, seeds=junk)
        assert s3 == 0 and np.abs(x3 - xr).max() < 1e-8, (trial, junk)
        # (d) the opposite sides of the true working set: every seed must be thrown out again
        x4, s4, _, ws4 = gi_variant.solve_v3(H, g, C, lb, ub, cl, cu, seeds=[(c, 1 - s) for c, s in ws0])
        assert s4 == 0 and np.abs(x4 - xr).max() < 1e-8 and sorted(ws4) == sorted(ws0)
    assert cold_it > 0
    print("working-set changes: cold %d, seeded with the own set %d, with the previous tick's set %+d vs cold" % (cold_it, warm_it, warm_prev_it))


def test_packed_qp_kernel_flow_restated_in_numpy():
    """tests/gi_variant.py solve_v5 — the packed stand-alone QP kernel's flow (csrc/wbc_k_qpp.hip): fixed variables presolved,
    equality rows through the add step with R'y1 = b_e riding along and x_eq from the factors, seeds by add steps with x / u read off the factors and
    restoration by drop + rebuild, dual iterations in which only inequality slots block — against the oracle's textbook loop: same status, same
    iteration count cold, same optimum cold and hot-started (own set: nothing but one step per seed; a perturbed problem's set; garbage; the opposite
    sides), contradictory equality rows infeasible, a redundant one skipped."""
    rng = np.random.default_rng(505)
    n, p = 12, 9
    cold_it = own_it = 0
    for trial in range(60):
        A = rng.normal(size=(n + 3, n))
        H = A.T @ A + 1e-3 * np.eye(n)
        g = rng.normal(size=n) * 4
        C = rng.normal(size=(p, n))
        lb, ub = -rng.uniform(0.02, 0.6, n), rng.uniform(0.02, 0.6, n)
        lb[3] = ub[3] = 0.04                                      # a fixed variable with a non-zero value
        cl, cu = -rng.uniform(0.02, 0.6, p), rng.uniform(0.02, 0.6, p)
        cl[1] = cu[1] = 0.05                                      # an equality row
        kind = trial % 6
        if kind == 1:
            C[4] = C[1]; cl[4] = cu[4] = -0.2                     # contradicts row 1: infeasible
        if kind == 2:
            C[4] = 3.0 * C[1]; cl[4] = cu[4] = 0.15               # redundant with row 1: skipped
        if kind == 3:
            cl[6], cu[6] = 30.0, 40.0                             # out of the box's reach: infeasible through the dual iterations
        xr, sr, ir = oracle.qp_solve(H[None], g[None], C[None], lb[None], ub[None], cl[None], cu[None])
        xr, sr, ir = xr[0], int(sr[0]), int(ir[0])
        x0, s0, it0, ws0 = gi_variant.solve_v5(H, g, C, lb, ub, cl, cu)
        assert s0 == sr, (trial, kind, s0, sr)
        if sr != 0:
            assert kind in (1, 3) and (x0 == 0).all()
            continue
        assert np.abs(x0 - xr).max() < 1e-9 and it0 == ir, (trial, it0, ir)
        assert all(c != 3 and c != n + 1 for c, _ in ws0)         # neither the fixed variable nor the equality row is carried
        cold_it += it0
        x1, s1, it1, ws1 = gi_variant.solve_v5(H, g, C, lb, ub, cl, cu, seeds=ws0)
        assert s1 == 0 and np.abs(x1 - xr).max() < 1e-9 and sorted(ws1) == sorted(ws0)
        x1n, s1n, it1n, _ = gi_variant.solve_v5(H, g, C, lb, ub, cl, cu, seeds=ws0, far=np.inf)
        n_eq = 2 if kind == 2 else 1                              # (the redundant row still counts as a working-set change)
        assert s1n == 0 and np.abs(x1n - xr).max() < 1e-9 and it1n == 1 + n_eq + len(ws0)
        own_it += it1
        xp, sp, _, wsp = gi_variant.solve_v5(H, g + rng.normal(size=n) * 0.15, C, lb, ub, cl, cu)
        if sp == 0:
            x2, s2, _, ws2 = gi_variant.solve_v5(H, g, C, lb, ub, cl, cu, seeds=wsp)
            assert s2 == 0 and np.abs(x2 - xr).max() < 1e-9 and sorted(ws2) == sorted(ws0)
        junk = [(int(rng.integers(-2, n + p + 2)), int(rng.integers(0, 2))) for _ in range(rng.integers(1, 12))]
        x3, s3, _, _ = gi_variant.solve_v5(H, g, C, lb, ub, cl, cu, seeds=junk)
        assert s3 == 0 and np.abs(x3 - xr).max() < 1e-8, (trial, junk)
        x4, s4, _, ws4 = gi_variant.solve_v5(H, g, C, lb, ub, cl, cu, seeds=[(c, 1 - s) for c, s in ws0])
        assert s4 == 0 and np.abs(x4 - xr).max() < 1e-8 and sorted(ws4) == sorted(ws0)
    assert cold_it > 0 and own_it <= cold_it
    print("working-set changes: cold %d, seeded with the own set %d" % (cold_it, own_it))


def _householder_null_basis_leg_first(E_base, K_blocks):
    """The device's construction (contact_presolve_orth) restated: Householder QR of E' with the coordinates ordered
    [leg 0, .., leg f-1, base]; every vector carried as (its entries at the CURRENT leg's coordinates, its base part) only — what a
    reflector leaves in other legs' coordinates of a later column is part of R and never read again. Returns Z [(6 + 3 f) x 6] in the
    order [base; leg 0; ..] whose columns span null([B K])."""
    nf = len(K_blocks)
    # vectors: the 3 nf columns of E' (contact rows), then the 6 + 3 nf unit vectors (-> rows of Q)
    vecs = []
    for f in range(nf):
        for r in range(3):
            vecs.append(dict(base=E_base[f][r].astype(float).copy(), own=f, kown=K_blocks[f][r].astype(float).copy()))
    for i in range(6):
        vecs.append(dict(base=np.eye(6)[i].copy(), own=-1, kown=np.zeros(3)))
    for f in range(nf):
        for t in range(3):
            vecs.append(dict(base=np.zeros(6), own=f, kown=np.eye(3)[t].copy()))
    for f in range(nf):
        cur = [v["kown"].copy() if v["own"] == f else np.zeros(3) for v in vecs]
        for r in range(3):
            piv = 3 * f + r
            x = np.concatenate([vecs[piv]["base"], cur[piv][r:]])
            sig = x @ x
            ek = cur[piv][r]
            alpha = -np.sqrt(sig) if ek > 0 else np.sqrt(sig)
            vb, vl = vecs[piv]["base"].copy(), np.zeros(3)
            vl[r] = ek - alpha
            vl[r + 1:] = cur[piv][r + 1:]
            beta = 1.0 / (sig - alpha * ek)
            for j, v in enumerate(vecs):
                w = beta * (vb @ v["base"] + vl[r:] @ cur[j][r:])
                v["base"] = v["base"] - w * vb
                cur[j][r:] = cur[j][r:] - w * vl[r:]
    return np.array([v["base"] for v in vecs[3 * nf:]])


def test_orthonormal_contact_presolve_restated_in_numpy():
    """Host-side statement of what wbc_tick_kernel<.., ORTH> does for configurations whose tasks touch the stance legs (BASELINE
    configs[1], "everything"): Z from the leg-first Householder QR is an orthonormal basis of the contact rows' null space, the
    reduced QP (H' = Z'HZ, bound rows = rows of Z) has cond(H') <= cond(H) and the same minimiser as the full problem — while the
    explicit basis [I; -K^-1 B] the sim3 presolve uses is orders of magnitude worse conditioned here."""
    wx, _ = common.models()
    for name, B, rot in (("c2", 24, False), ("everything", 12, True)):
        cfg = common.config(name, wx)
        d = common.tick_inputs(wx, cfg, B, seed=5, with_rot=rot)
        qp = oracle.assemble([wx], [cfg], d, DT, B)
        x_ref, st, it = oracle.qp_solve(qp["H"], qp["g"], qp["C"], qp["lb"], qp["ub"], qp["Clb"], qp["Cub"])
        worse = []
        for b in np.nonzero(st == 0)[0]:
            H, g, Cm, lb, ub, cl, cu = (qp[k][b] for k in ("H", "g", "C", "lb", "ub", "Clb", "Cub"))
            eq = [i for i in range(Cm.shape[0]) if cl[i] == cu[i] == 0.0 and np.any(Cm[i, 6:18] != 0)]
            assert len(eq) == 12
            feet = [eq[3 * f:3 * f + 3] for f in range(4)]
            legd = [[j for j in range(6, 18) if Cm[rows[0], j] != 0 or Cm[rows[1], j] != 0 or Cm[rows[2], j] != 0] for rows in feet]
            assert all(len(l) == 3 for l in legd)
            Zbl = _householder_null_basis_leg_first([Cm[rows][:, :6] for rows in feet], [Cm[np.ix_(rows, l)] for rows, l in zip(feet, legd)])
            bl = list(range(6)) + [j for l in legd for j in l]
            E = Cm[np.ix_(eq, bl)]
            assert np.abs(E @ Zbl).max() < 1e-12 * np.abs(E).max() * 10 and np.abs(Zbl.T @ Zbl - np.eye(6)).max() < 1e-13
            locked = [j for j in range(26) if lb[j] == ub[j] == 0.0]
            rest = [j for j in range(26) if j not in bl and j not in locked]
            Z = np.zeros((26, 6 + len(rest)))
            Z[np.ix_(bl, range(6))] = Zbl
            for k, j in enumerate(rest):
                Z[j, 6 + k] = 1.0
            keep = [i for i in range(Cm.shape[0]) if i not in eq]
            Hr = Z.T @ H @ Z
            y, s2, _ = oracle.qp_solve(Hr, Z.T @ g, np.vstack([Cm[keep] @ Z, Z[bl]]),
                                       np.concatenate([np.full(6, -1e30), lb[rest]]), np.concatenate([np.full(6, 1e30), ub[rest]]),
                                       np.concatenate([cl[keep], lb[bl]]), np.concatenate([cu[keep], ub[bl]]))
            assert s2 == 0
            assert np.abs(Z @ y - x_ref[b]).max() < 1e-6
            Hfree = H[np.ix_([j for j in range(26) if j not in locked], [j for j in range(26) if j not in locked])]
            assert np.linalg.cond(Hr) <= 1.0001 * np.linalg.cond(Hfree)
            # the explicit basis, for the record
            G = np.vstack([-np.linalg.solve(Cm[np.ix_(rows, l)], Cm[rows][:, :6]) for rows, l in zip(feet, legd)])
            # ... and normalised through the 6 x 6 Cholesky factor of I + G'G (what the device does for well-conditioned leg blocks,
            # orth_null_basis): the same null space, orthonormal to ~eps |G|^2
            S = np.linalg.inv(np.linalg.cholesky(np.eye(6) + G.T @ G)).T
            Zf = np.vstack([S, G @ S])
            assert np.abs(E @ Zf).max() < 1e-9 * np.abs(E).max() and np.abs(Zf.T @ Zf - np.eye(6)).max() < 1e-7
            assert np.abs(Zbl @ (Zbl.T @ Zf) - Zf).max() < 1e-9
            Ze = Z.copy()
            Ze[np.ix_(bl, range(6))] = np.vstack([np.eye(6), G])
            worse.append(np.linalg.cond(Ze.T @ H @ Ze) / np.linalg.cond(Hr))
        assert np.median(worse) > 30, np.median(worse)


def test_packed_warm_start_variant_reaches_the_same_optimum():
    """tests/gi_variant.py solve_v4 = the warm start of the packed kernel (wbc_tick_sim3p_kernel<WARM>: seeds through the add step, x / u
    rebuilt from the factors, restoration, dual iterations) on inequality-only problems of the reduced sim3 size: seeded with its own
    set, the previous tick's, garbage or the opposite sides, it returns the cold optimum and the cold working set."""
    rng = np.random.default_rng(5)
    n, p = 11, 16
    cold_it, own_it, prev_d = 0, 0, 0
    for trial in range(80):
        A = rng.normal(size=(n + 4, n))
        H = A.T @ A + 1e-3 * np.eye(n)
        g = rng.normal(size=n) * 4
        C = rng.normal(size=(p, n))
        lb, ub = -rng.uniform(0.02, 0.5, n), rng.uniform(0.02, 0.5, n)
        cl, cu = -rng.uniform(0.02, 0.5, p), rng.uniform(0.02, 0.5, p)
        xr, sr, _ = oracle.qp_solve(H[None], g[None], C[None], lb[None], ub[None], cl[None], cu[None])
        xr, sr = xr[0], int(sr[0])
        x0, s0, it0, ws0 = gi_variant.solve_v4(H, g, C, lb, ub, cl, cu)
        assert s0 == sr
        if sr != 0:
            continue
        assert np.abs(x0 - xr).max() < 1e-9
        x1, s1, it1, ws1 = gi_variant.solve_v4(H, g, C, lb, ub, cl, cu, seeds=ws0)
        assert s1 == 0 and np.abs(x1 - xr).max() < 1e-9 and sorted(ws1) == sorted(ws0)
        x1n, s1n, it1n, _ = gi_variant.solve_v4(H, g, C, lb, ub, cl, cu, seeds=ws0, far=None)
        assert s1n == 0 and np.abs(x1n - xr).max() < 1e-9 and it1n == len(ws0)       # one add step per seed, nothing else
        cold_it += it0
        own_it += it1
        xp, sp, itp, wsp = gi_variant.solve_v4(H, g + rng.normal(size=n) * 0.15, C, lb, ub, cl, cu)
        if sp == 0:
            x2, s2, it2, _ = gi_variant.solve_v4(H, g, C, lb, ub, cl, cu, seeds=wsp)
            assert s2 == 0 and np.abs(x2 - xr).max() < 1e-9
            prev_d += it2 - it0
        junk = [(int(rng.integers(-2, n + p + 2)), int(rng.integers(0, 2))) for _ in range(rng.integers(1, 12))]
        x3, s3, _, _ = gi_variant.solve_v4(H, g, C, lb, ub, cl, cu, seeds=junk)
        assert s3 == 0 and np.abs(x3 - xr).max() < 1e-8, (trial, junk)
        x4, s4, _, ws4 = gi_variant.solve_v4(H, g, C, lb, ub, cl, cu, seeds=[(c, 1 - s) for c, s in ws0])
        assert s4 == 0 and np.abs(x4 - xr).max() < 1e-8 and sorted(ws4) == sorted(ws0)
    assert cold_it > 0 and own_it <= cold_it
    print("working-set changes: cold %d, seeded with the own set %d, with the previous tick's set %+d vs cold" % (cold_it, own_it, prev_d))


def test_packed_orth_kernel_formulation_restated_in_numpy():
    """What wbc_tick_orthp_kernel computes for BASELINE configs[1], in numpy from the oracle's A, b and contact rows:
    G = -K^-1 B per foot, the Cholesky factor of M = I + G'G row by row (the cooperative form: a finished row and 1 / L_jj are broadcast),
    S = L^-T by one forward substitution per column, Z = [S; G S]; then block by block (CoM first, then the EE tasks) H' += (A_blk Z)'(A_blk Z)
    and g' -= (A_blk Z)'b_blk, H' += d^2 I; the 16 x 16 Cholesky sweep with the substitutions L y_s = e_s and — on the padding lane — L z = g';
    y = -L^-T z as a dot product per variable; qd = Z y. Same answer as the oracle's full-size solve with its 12 equality rows."""
    wx, px = common.models()
    for model in (wx, px):
        cfg = common.config("c2", model)
        B = 16
        d = common.tick_inputs(model, cfg, B, seed=9)
        qp = oracle.assemble([model], [cfg], d, DT, B)
        ref = oracle.tick([model], [cfg], d, DT, B)
        nv = model.nv
        for b in range(B):
            A, bv, Cm = qp["A"][b][:, :nv], qp["b"][b], qp["C"][b][:, :nv]
            mcart = A.shape[0] - 26
            feet = [list(range(3 * f, 3 * f + 3)) for f in range(4)]
            legd = [[j for j in range(6, 18) if np.any(Cm[rows][:, j] != 0)] for rows in feet]
            bl = list(range(6)) + [j for l in legd for j in l]
            rest = [j for j in range(nv) if j not in bl]
            G = np.vstack([-np.linalg.solve(Cm[np.ix_(rows, l)], Cm[rows][:, :6]) for rows, l in zip(feet, legd)])
            # cooperative Cholesky of M = I + G'G: lane r keeps row r, step j finalises row j and the rows below take their column j
            M = np.eye(6) + G.T @ G
            L = M.copy()
            rinv = np.zeros(6)
            for j in range(6):
                v = L[j, j] - L[j, :j] @ L[j, :j]
                rinv[j] = 1.0 / np.sqrt(v)
                L[j, j] = v * rinv[j]
                for r in range(j + 1, 6):
                    L[r, j] = (L[r, j] - L[r, :j] @ L[j, :j]) * rinv[j]
            L = np.tril(L)
            S = np.zeros((6, 6))
            for c in range(6):                     # column c of L^-1 = row c of S
                x = np.zeros(6)
                for i in range(c, 6):
                    x[i] = ((1.0 if i == c else 0.0) - L[i, :i] @ x[:i]) * rinv[i]
                S[c] = x
            Zbl = np.vstack([S, G @ S])
            assert np.abs(Zbl.T @ Zbl - np.eye(6)).max() < 1e-7 and np.abs(Cm[:, bl] @ Zbl).max() < 1e-9 * np.abs(Cm).max()
            n = 6 + len(rest)
            assert n <= 15
            Z = np.zeros((nv, n))
            Z[np.ix_(bl, range(6))] = Zbl
            for k, j in enumerate(rest):
                Z[j, 6 + k] = 1.0
            dpost = A[mcart + 7, 7]                # the posture block's diagonal
            H = np.zeros((16, 16))
            g = np.zeros(16)
            blocks = [range(30, 33)] + [range(6 * e, 6 * e + 6) for e in range(5)]      # CoM first, then FR, FL, RR, RL, Grip
            for rows in blocks:
                AZ = A[list(rows)] @ Z
                H[:n, :n] += AZ.T @ AZ
                g[:n] -= AZ.T @ bv[list(rows)]
            H[:n, :n] += dpost * dpost * np.eye(n)
            for k in range(n, 16):
                H[k, k] = 1.0
            # the sweep: right-looking Cholesky, every "lane" s carries its row of H and the right-hand side e_s; lane 15 carries g'
            Hs, Y = H.copy(), np.eye(16)
            Y[15] = g
            for j in range(16):
                pj = Hs[j, j]
                assert pj > 0
                rj = 1.0 / np.sqrt(pj)
                col = Hs[:, j].copy()
                for s in range(16):
                    th, ty = Hs[s, j] / pj, Y[s, j] / pj
                    Hs[s, j + 1:] -= th * col[j + 1:]
                    Y[s, j + 1:] -= ty * col[j + 1:]
                    Y[s, j] = Y[s, j] * rj
            z = Y[15]                               # L^-1 g'
            y = -(Y[:n] @ z)                        # row s of L^-T against it
            qd = Z @ y
            assert np.abs(qd - ref["qdot"][b][:nv]).max() < 1e-6, np.abs(qd - ref["qdot"][b][:nv]).max()


def test_packed_box_kernel_formulation_restated_in_numpy():
    """What wbc_tick_boxp_kernel computes for the warm-up problem, in numpy from the oracle's A, b and velocity box: the block-arrow
    structure it relies on (a limb DoF meets another limb only through the base), the eliminated set E = base (+ the DoF with the widest box where 16
    lanes do not hold the rest), the cooperative Cholesky of H_EE with the forward substitutions of W~ = L^-1 H_EK and L^-1 g_E riding on it,
    H' = (limb blocks of H_KK) - W~'W~, g' = g_K - W~'(L^-1 g_E), the dual active-set method on <= 16 bounded unknowns, x_E = -L^-T(L^-1 g_E + W~ x_K)
    and the check of the eliminated DoF's own bounds. Same answer AND the same working-set changes as the oracle's solve of the full 26-unknown problem
    (the reduced problem's dual iterates are the full problem's)."""
    wx, px = common.models()
    for model, cfg_kw in ((wx, dict(Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=True)),
                          (px, dict(Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=True)),
                          (wx, dict(FR=True, RL=True, Grip=True, Joint="PREV"))):
        import wbc_model
        cfg = wbc_model.make_config(model, **cfg_kw)
        B = 24
        d = common.tick_inputs(model, cfg, B, seed=19, with_rot=True)
        qp = oracle.assemble([model], [cfg], d, DT, B)
        ref = oracle.tick([model], [cfg], d, DT, B)
        nv = model.nv
        assert qp["C"].shape[1] == 0
        lock = [k for k in range(6, nv) if k >= cfg.lock_from]
        free = [k for k in range(6, nv) if k not in lock]
        # the plan's choice of the extra eliminated DoF: widest box = position range x velocity limit of the damper entry the DoF looks at
        extra = sorted(free, key=lambda k: (-(cfg.damper_hi[k] - cfg.damper_lo[k]) * cfg.damper_vmax[k], k))[:max(0, len(free) - 16)]
        E = list(range(6)) + extra
        K = [k for k in free if k not in extra]
        assert len(K) <= 16 and len(E) <= 8
        worst, conds = 0.0, []
        for b in range(B):
            if ref["status"][b] != 0:
                continue
            H, g, lb, ub = qp["H"][b][:nv, :nv], qp["g"][b][:nv], qp["lb"][b][:nv], qp["ub"][b][:nv]
            assert (lb[lock] == 0).all() and (ub[lock] == 0).all()
            # block-arrow: two kept DoF are coupled only inside a limb (here: |H_KK| vanishes between different task supports)
            A = qp["A"][b][:, :nv]
            mcart = A.shape[0] - 26
            sup = [frozenset(np.nonzero(np.abs(A[:mcart, k]) > 0)[0] // 6) for k in K]
            for i, ki in enumerate(K):
                for j, kj in enumerate(K):
                    if i != j and not (sup[i] & sup[j]):
                        assert H[ki, kj] == 0.0
            HEE, HEK, HKK = H[np.ix_(E, E)], H[np.ix_(E, K)], H[np.ix_(K, K)]
            ne = len(E)
            # cooperative Cholesky; step j's row of L also finishes entry j of W~ and of L^-1 g_E
            L = HEE.copy(); rinv = np.zeros(ne); Wt = HEK.copy(); gt = g[E].copy()
            for j in range(ne):
                v = L[j, j] - L[j, :j] @ L[j, :j]
                assert v > 0
                rinv[j] = 1.0 / np.sqrt(v)
                L[j, j] = v * rinv[j]
                for r in range(j + 1, ne):
                    L[r, j] = (L[r, j] - L[r, :j] @ L[j, :j]) * rinv[j]
                Wt[j] = (Wt[j] - L[j, :j] @ Wt[:j]) * rinv[j]
                gt[j] = (gt[j] - L[j, :j] @ gt[:j]) * rinv[j]
            L = np.tril(L)
            assert np.abs(L @ L.T - HEE).max() < 1e-9 * np.abs(HEE).max()
            Hp = HKK - Wt.T @ Wt
            gp = g[K] - Wt.T @ gt
            conds.append(np.linalg.cond(Hp) / np.linalg.cond(H))
            xK, st, it = gi_variant.solve(Hp, gp, lb=lb[K], ub=ub[K])[:3]
            assert st == 0
            xE = -np.linalg.solve(L.T, gt + Wt @ xK)
            x = np.zeros(26)
            x[K], x[E] = xK, xE
            # the eliminated DoF's own bounds hold (else the kernel hands the instance to the general path)
            held = ((xE < lb[E] - 1e-9 * np.maximum(1, np.abs(lb[E]))) | (xE > ub[E] + 1e-9 * np.maximum(1, np.abs(ub[E])))).any()
            if held:
                continue
            worst = max(worst, np.abs(x - ref["qdot"][b]).max())
            assert it + len(lock) + (26 - nv) == ref["iters"][b]      # (the oracle also counts the padded DoF of the smaller model as a locked bound)
        assert worst < 1e-9, worst
        print("%s %s: worst |x - oracle| %.2e, cond(H') / cond(H) median %.1e" % (model.name, sorted(cfg_kw), worst, float(np.median(conds))))
        assert np.median(conds) < 1.0          # (never worse; 1e-3 on a typical wx200 warm-up instance: the locked DoF and the base leave the problem)
