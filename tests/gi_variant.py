"""numpy model of the QP algorithm VARIANT the HIP kernel runs (test infrastructure).

The CPU oracle (oracle/wbc_oracle.c) is the textbook Goldfarb–Idnani dual active-set: Givens rotations to add a
constraint, an explicit triangular factor R with back-substitution. The wavefront kernel uses forms that map to
64 lanes without sequential triangular solves:
  * add:  ONE Householder reflector on the trailing columns of J (J2 <- J2 P), using z = J2 d2 already at hand;
  * T = R^-1 is kept explicitly, so r = T d1 is a mat-vec; adding a constraint appends the column (-r/delta, 1/delta);
  * drop: the Givens sequence is read off the removed ROW l of T (it is orthogonal to range(R without column l)),
          applied to adjacent columns of T and of J1.
This file states that variant in plain numpy so its algebra is checked against the oracle on the CPU, lane
mapping aside; tests/test_qp_variant.py runs it. Decision rules and tolerances are the oracle's.
"""
import numpy as np

INF = 1e20
EPS2 = 2.220446049250313e-16 ** 2


def refine_step(J, q, act, act_side, x, neg_grad, normal, rhs):
    """The kernels' refinement step at the final working set (csrc/wbc_common.h qp_refine; oracle: qp_refine) in plain numpy. Needs nothing of
    the dual method but its final J (J J' = H^-1, the first q columns spanning the active normals) and the list of active constraints:
    R = J1'N' is REBUILT from them (so equality slots, whose part of R the register-resident QR never stores, are covered), then
        u = R^-1 J1' grad f,   r1 = -(grad f - N'u),   r2_k = b_k - n_k'x,   x += J1 R^-T r2 + J2 J2' r1.
    neg_grad(x) returns -grad f(x) from the caller's UNFACTORED data (A'(b - A x) for a least-squares problem)."""
    n = len(x)
    gneg = neg_grad(x)
    N = np.array([normal(c, sd) for c, sd in zip(act, act_side)]).reshape(q, n)
    R = np.triu((J.T @ N.T)[:q, :q])                       # what rounding leaves below the diagonal is dropped, as on the device
    w = J.T @ gneg
    u = np.linalg.solve(R, -w[:q]) if q else np.zeros(0)   # back substitution
    r1 = gneg + N.T @ u
    r2 = np.array([rhs(c, sd) for c, sd in zip(act, act_side)]) - N @ x if q else np.zeros(0)
    dy = J.T @ r1
    if q:
        dy[:q] = np.linalg.solve(R.T, r2)                  # forward substitution
    return x + J @ dy


def solve(H, g, C=None, lb=None, ub=None, Clb=None, Cub=None, max_iter=None, neg_grad=None):
    """neg_grad: x -> -grad f(x) from the unfactored data; given, one refinement step (refine_step) is applied at the final working set."""
    n = len(g)
    p = 0 if C is None else C.shape[0]
    ncon = n + p

    def lo(c):
        return (lb[c] if lb is not None else -1e30) if c < n else Clb[c - n]

    def hi(c):
        return (ub[c] if ub is not None else 1e30) if c < n else Cub[c - n]

    def normal(c, side):
        s = -1.0 if side else 1.0
        if c < n:
            e = np.zeros(n)
            e[c] = s
            return e
        return s * C[c - n]

    def value(c, x):
        return x[c] if c < n else C[c - n] @ x

    try:
        L = np.linalg.cholesky(H)
    except np.linalg.LinAlgError:
        return np.zeros(n), 3, 0
    J = np.linalg.inv(L).T.copy()
    jf2 = (J * J).sum()
    x = -J @ (J.T @ g)
    T = np.zeros((n, n))
    u = np.zeros(n + 1)
    act, act_eq, act_side, active = [], [], [], np.zeros(ncon, bool)
    q = 0
    iters = 0
    max_iter = max_iter or 10 * (n + p) + 20
    eqs = [c for c in range(ncon) if lo(c) == hi(c) and abs(lo(c)) < INF]
    eqi = 0
    while True:
        ip = -1
        if eqi < len(eqs):
            ip, side, is_eq = eqs[eqi], 0, True
            eqi += 1
            b_ip = lo(ip)
            s_ip = value(ip, x) - b_ip
        else:
            is_eq = False
            worst = 0.0
            for c in range(ncon):
                if active[c] or (lo(c) == hi(c) and abs(lo(c)) < INF):
                    continue
                v = value(c, x)
                if lo(c) > -INF:
                    s = v - lo(c)
                    if s < -1e-9 * max(1.0, abs(lo(c))) and s < worst:
                        worst, ip, side, b_ip = s, c, 0, lo(c)
                if hi(c) < INF:
                    s = hi(c) - v
                    if s < -1e-9 * max(1.0, abs(hi(c))) and s < worst:
                        worst, ip, side, b_ip = s, c, 1, -hi(c)
            if ip < 0:
                if neg_grad is not None:
                    x = refine_step(J, q, act, act_side, x, neg_grad, normal, lambda c, sd: -hi(c) if sd else lo(c))
                return x, 0, iters
            s_ip = worst
        npv = normal(ip, side)
        np2 = npv @ npv
        u_ip = 0.0
        while True:
            iters += 1
            if iters > max_iter:
                return x, 1, iters
            d = J.T @ npv
            zn = d[q:] @ d[q:]
            z = J[:, q:] @ d[q:]
            r = T[:q, :q] @ d[:q]
            have_step = zn > 100.0 * n * EPS2 * jf2 * np2
            t1, l = np.inf, -1
            for k in range(q):
                if not act_eq[k] and r[k] > 0 and u[k] / r[k] < t1:
                    t1, l = u[k] / r[k], k
            t2 = -s_ip / zn if have_step else np.inf
            if is_eq and not have_step:
                if abs(s_ip) <= 1e-9 * max(1.0, abs(b_ip)):
                    break
                return x, 2, iters
            t = t2 if is_eq else min(t1, t2)
            if not np.isfinite(t):
                return x, 2, iters
            if have_step:
                x = x + t * z
            u[:q] -= t * r
            u_ip += t
            if have_step and t == t2:
                # --- add with one Householder reflector: P d2 = delta e1, J2 <- J2 P
                dq = d[q]
                delta = -np.sqrt(zn) if dq >= 0 else np.sqrt(zn)
                v = d[q:].copy()
                v[0] -= delta
                vv = 2.0 * (zn - delta * dq)           # v'v
                if vv > 0:
                    w = z - delta * J[:, q]            # J2 v
                    J[:, q:] -= np.outer(w, (2.0 / vv) * v)
                T[:q, q] = -r / delta
                T[q, q] = 1.0 / delta
                u[q] = u_ip
                act.append(ip)
                act_eq.append(is_eq)
                act_side.append(side)
                active[ip] = True
                q += 1
                break
            # --- drop blocking constraint l: rotations from the removed row of T
            trow = T[l, l:q].copy()
            active[act[l]] = False
            del act[l], act_eq[l], act_side[l]
            u[l:q - 1] = u[l + 1:q].copy()
            Tt = np.delete(T[:q, :q], l, axis=0)       # (q-1) x q
            h = trow[0]
            for k in range(l, q - 1):
                a, b = h, trow[k - l + 1]
                rho = np.hypot(a, b)
                c_, s_ = (b / rho, -a / rho) if rho > 0 else (1.0, 0.0)
                h = rho
                ck, ck1 = Tt[:, k].copy(), Tt[:, k + 1].copy()
                Tt[:, k], Tt[:, k + 1] = c_ * ck + s_ * ck1, -s_ * ck + c_ * ck1
                jk, jk1 = J[:, k].copy(), J[:, k + 1].copy()
                J[:, k], J[:, k + 1] = c_ * jk + s_ * jk1, -s_ * jk + c_ * jk1
            T[:, :] = np.triu(T) * 0
            T[:q - 1, :q - 1] = np.triu(Tt[:, :q - 1])
            q -= 1
            s_ip = (-1.0 if side else 1.0) * value(ip, x) - b_ip



def solve_v2(H, g, C=None, lb=None, ub=None, Clb=None, Cub=None, max_iter=None):
    """Kernel v2 algebra: the equalities are absorbed first by a Householder QR of J'N_e that updates only J
    (x and the multipliers are not tracked meanwhile); x_eq = J1 y1 - J2 J2'g with R'y1 = b_e solved incrementally;
    T = R^-1 is kept only for the inequality slots (its lower-right block, which is all r = T d1 and the drop ever read)."""
    n = len(g)
    p = 0 if C is None else C.shape[0]
    ncon = n + p

    def lo(c):
        return (lb[c] if lb is not None else -1e30) if c < n else Clb[c - n]

    def hi(c):
        return (ub[c] if ub is not None else 1e30) if c < n else Cub[c - n]

    def normal(c, side):
        sgn = -1.0 if side else 1.0
        if c < n:
            e = np.zeros(n)
            e[c] = sgn
            return e
        return sgn * C[c - n]

    def value(c, x):
        return x[c] if c < n else C[c - n] @ x

    try:
        L = np.linalg.cholesky(H)
    except np.linalg.LinAlgError:
        return np.zeros(n), 3, 0
    J = np.linalg.inv(L).T.copy()
    jf2 = (J * J).sum()
    iters = 0
    max_iter = max_iter or 10 * (n + p) + 20
    # ---- equality block
    q = 0
    y1 = np.zeros(n)
    active = np.zeros(ncon, bool)
    for c in range(ncon):
        if not (lo(c) == hi(c) and abs(lo(c)) < INF):
            continue
        iters += 1
        npv = normal(c, 0)
        b_e = lo(c)
        d = J.T @ npv
        zn = d[q:] @ d[q:]
        dy = d[:q] @ y1[:q]
        if not zn > 100.0 * n * EPS2 * jf2 * (npv @ npv):
            if abs(dy - b_e) <= 1e-9 * max(1.0, abs(b_e)):
                continue
            return np.zeros(n), 2, iters
        dq = d[q]
        delta = -np.sqrt(zn) if dq >= 0 else np.sqrt(zn)
        v = d[q:].copy()
        v[0] -= delta
        vv = 2.0 * (zn - delta * dq)
        if vv > 0:
            w = J[:, q:] @ v
            J[:, q:] -= np.outer(w, (2.0 / vv) * v)
        y1[q] = (b_e - dy) / delta
        active[c] = True
        q += 1
    qe = q
    dg = J.T @ g
    yv = np.concatenate([y1[:qe], -dg[qe:]])
    x = J @ yv
    # ---- inequality phase: slots qe.. ; T holds only the (q-qe) x (q-qe) block
    T = np.zeros((n, n))
    u = np.zeros(n + 1)
    act = []
    while True:
        worst, ip = 0.0, -1
        for c in range(ncon):
            if active[c] or (lo(c) == hi(c) and abs(lo(c)) < INF):
                continue
            vv_ = value(c, x)
            if lo(c) > -INF:
                s = vv_ - lo(c)
                if s < -1e-9 * max(1.0, abs(lo(c))) and s < worst:
                    worst, ip, side, b_ip = s, c, 0, lo(c)
            if hi(c) < INF:
                s = hi(c) - vv_
                if s < -1e-9 * max(1.0, abs(hi(c))) and s < worst:
                    worst, ip, side, b_ip = s, c, 1, -hi(c)
        if ip < 0:
            return x, 0, iters
        s_ip = worst
        npv = normal(ip, side)
        np2 = npv @ npv
        u_ip = 0.0
        while True:
            iters += 1
            if iters > max_iter:
                return x, 1, iters
            d = J.T @ npv
            zn = d[q:] @ d[q:]
            z = J[:, q:] @ d[q:]
            r = T[qe:q, qe:q] @ d[qe:q]                     # only the inequality slots
            have_step = zn > 100.0 * n * EPS2 * jf2 * np2
            t1, l = np.inf, -1
            for k in range(q - qe):
                if r[k] > 0 and u[k] / r[k] < t1:
                    t1, l = u[k] / r[k], k
            t2 = -s_ip / zn if have_step else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return x, 2, iters
            if have_step:
                x = x + t * z
            u[:q - qe] -= t * r
            u_ip += t
            if have_step and t == t2:
                dq = d[q]
                delta = -np.sqrt(zn) if dq >= 0 else np.sqrt(zn)
                v = d[q:].copy()
                v[0] -= delta
                vv = 2.0 * (zn - delta * dq)
                if vv > 0:
                    w = z - delta * J[:, q]
                    J[:, q:] -= np.outer(w, (2.0 / vv) * v)
                T[qe:q, q] = -r / delta
                T[q, q] = 1.0 / delta
                u[q - qe] = u_ip
                act.append(ip)
                active[ip] = True
                q += 1
                break
            # drop inequality slot l (absolute slot qe + l)
            la = qe + l
            trow = T[la, la:q].copy()
            active[act[l]] = False
            del act[l]
            u[l:q - qe - 1] = u[l + 1:q - qe].copy()
            Tt = np.delete(T[qe:q, qe:q], l, axis=0)
            h = trow[0]
            for k in range(la, q - 1):
                a_, b_ = h, trow[k - la + 1]
                rho = np.hypot(a_, b_)
                c_, s_ = (b_ / rho, -a_ / rho) if rho > 0 else (1.0, 0.0)
                h = rho
                kk = k - qe
                ck, ck1 = Tt[:, kk].copy(), Tt[:, kk + 1].copy()
                Tt[:, kk], Tt[:, kk + 1] = c_ * ck + s_ * ck1, -s_ * ck + c_ * ck1
                jk, jk1 = J[:, k].copy(), J[:, k + 1].copy()
                J[:, k], J[:, k + 1] = c_ * jk + s_ * jk1, -s_ * jk + c_ * jk1
            T[qe:, qe:] = 0
            m_ = q - qe - 1
            T[qe:qe + m_, qe:qe + m_] = np.triu(Tt[:, :m_])
            q -= 1
            s_ip = (-1.0 if side else 1.0) * value(ip, x) - b_ip


def solve_v3(H, g, C=None, lb=None, ub=None, Clb=None, Cub=None, seeds=(), max_iter=None, refresh=True, far=0.1):
    """Kernel v3 algebra = solve_v2 + WARM START (the hot-start analogue of QP_Wrapper.py:55-73):
    `seeds` = [(constraint, side)] carried over from the previous tick's final working set.
      1. the seeds go through the same Householder QR as the equalities (cheap, register resident on the wavefront), each one
         appending its column (-T r / delta, 1 / delta) to T = R22^-1 so that they stay droppable; dependent seeds are skipped;
      2. x = J1 y1 - J2 J2'g, multipliers of the seeded slots u = T (y1 + J'g)[qe:q];
      3. RESTORATION: while a seeded multiplier is negative, drop the most negative slot l (Givens, as in the dual method)
         and move to the minimiser on the reduced set: with d, z, r of the dropped constraint taken on the NEW factors,
         x <- x - u_l z, u <- u + u_l r (the add step of the dual method read backwards). What is left is an S-pair
         (x minimises on W, u >= 0) — the dual iterations carry on from there.
    Returns (x, status, iters, final working set [(constraint, side)])."""
    n = len(g)
    p = 0 if C is None else C.shape[0]
    ncon = n + p

    def lo(c):
        return (lb[c] if lb is not None else -1e30) if c < n else Clb[c - n]

    def hi(c):
        return (ub[c] if ub is not None else 1e30) if c < n else Cub[c - n]

    def is_eq(c):
        return lo(c) == hi(c) and abs(lo(c)) < INF

    def normal(c, side):
        sgn = -1.0 if side else 1.0
        if c < n:
            e = np.zeros(n)
            e[c] = sgn
            return e
        return sgn * C[c - n]

    def value(c, x):
        return x[c] if c < n else C[c - n] @ x

    try:
        L = np.linalg.cholesky(H)
    except np.linalg.LinAlgError:
        return np.zeros(n), 3, 0, []
    J = np.linalg.inv(L).T.copy()
    jf2 = (J * J).sum()
    iters = 0
    max_iter = max_iter or 10 * (n + p) + 20
    q = 0
    y1 = np.zeros(n)
    active = np.zeros(ncon, bool)
    T = np.zeros((n, n))
    act = []                                    # inequality slots: (constraint, side)
    todo = [(c, 0, True) for c in range(ncon) if is_eq(c)]
    qe = None
    for c, side in seeds:
        if 0 <= c < ncon and not is_eq(c) and ((side == 0 and lo(c) > -INF) or (side == 1 and hi(c) < INF)) and \
                not any(t[0] == c for t in todo):
            todo.append((c, side, False))
    todo = todo[:n]
    x0f = None
    for c, side, eq in todo:
        if not eq and qe is None:
            qe = q
        npv = normal(c, side)
        b_e = lo(c) if side == 0 else -hi(c)
        if not eq and far is not None:
            # a seed is taken only if the equalities-only minimiser violates it or comes close to it: a constraint that is
            # active at the solution almost always is, and a seed far on the feasible side (a velocity bound of tens of rad/s
            # for a joint that hardly moves) would drag the iterate far away — harmless in exact arithmetic, digits at cond(H) ~ 1e9
            if x0f is None:
                dg0 = J.T @ g
                x0f = J @ np.concatenate([y1[:q], -dg0[q:]])
            if npv @ x0f - b_e > far * max(1.0, np.abs(x0f).max()):
                continue
        iters += 1
        d = J.T @ npv
        zn = d[q:] @ d[q:]
        dy = d[:q] @ y1[:q]
        if not zn > 100.0 * n * EPS2 * jf2 * (npv @ npv):
            if not eq or abs(dy - b_e) <= 1e-9 * max(1.0, abs(b_e)):
                continue                         # a dependent seed is simply not taken
            return np.zeros(n), 2, iters, []
        dq = d[q]
        delta = -np.sqrt(zn) if dq >= 0 else np.sqrt(zn)
        v = d[q:].copy()
        v[0] -= delta
        vv = 2.0 * (zn - delta * dq)
        if vv > 0:
            w = J[:, q:] @ v
            J[:, q:] -= np.outer(w, (2.0 / vv) * v)
        y1[q] = (b_e - dy) / delta
        if not eq:
            T[qe:q, q] = -(T[qe:q, qe:q] @ d[qe:q]) / delta
            T[q, q] = 1.0 / delta
            act.append((c, side))
        active[c] = True
        q += 1
    if qe is None:
        qe = q
    dg = J.T @ g
    x = J @ np.concatenate([y1[:q], -dg[q:]])
    # the equalities-only minimiser (independent of whatever happens to the columns >= qe): anchor of the refresh below
    x0 = J @ np.concatenate([y1[:qe], -dg[qe:]])
    u = np.zeros(n + 1)
    u[:q - qe] = T[qe:q, qe:q] @ (y1[qe:q] + dg[qe:q])

    def drop(l):
        nonlocal q
        la = qe + l
        trow = T[la, la:q].copy()
        active[act[l][0]] = False
        del act[l]
        u[l:q - qe - 1] = u[l + 1:q - qe].copy()
        u[q - qe - 1] = 0.0
        Tt = np.delete(T[qe:q, qe:q], l, axis=0)
        h = trow[0]
        for k in range(la, q - 1):
            a_, b_ = h, trow[k - la + 1]
            rho = np.hypot(a_, b_)
            c_, s_ = (b_ / rho, -a_ / rho) if rho > 0 else (1.0, 0.0)
            h = rho
            kk = k - qe
            ck, ck1 = Tt[:, kk].copy(), Tt[:, kk + 1].copy()
            Tt[:, kk], Tt[:, kk + 1] = c_ * ck + s_ * ck1, -s_ * ck + c_ * ck1
            jk, jk1 = J[:, k].copy(), J[:, k + 1].copy()
            J[:, k], J[:, k + 1] = c_ * jk + s_ * jk1, -s_ * jk + c_ * jk1
        T[qe:, qe:] = 0
        m_ = q - qe - 1
        T[qe:qe + m_, qe:qe + m_] = np.triu(Tt[:, :m_])
        q -= 1

    # ---- restoration of dual feasibility
    restored = False
    while q > qe and u[:q - qe].min() < 0.0:
        iters += 1
        if iters > max_iter:
            return x, 1, iters, list(act)
        l = int(np.argmin(u[:q - qe]))
        ul = u[l]
        c, side = act[l]
        drop(l)
        d = J.T @ normal(c, side)
        z = J[:, q:] @ d[q:]
        r = T[qe:q, qe:q] @ d[qe:q]
        x = x - ul * z
        u[:q - qe] += ul * r
        restored = True
    if refresh and restored:
        # x and u went through iterates as far away as the wrong seeds put them (a velocity bound is tens of rad/s), and with
        # cond(H) ~ 1e9 that costs digits. The factors J, T did not (orthogonal updates only): rebuild x and u from them.
        # With s_j = b_j - n_j'x0 the slacks of the remaining inequality slots at the equalities-only minimiser x0:
        #   w = T's,  x = x0 + J[:, qe:q] w,  u = T w
        sl = np.array([(lo(c) if sd == 0 else -hi(c)) - normal(c, sd) @ x0 for c, sd in act])
        Tb = T[qe:q, qe:q]
        w = Tb.T @ sl
        x = x0 + J[:, qe:q] @ w
        u[:q - qe] = Tb @ w
        # (the refreshed multipliers may show a sign the drifted ones hid: restore once more on the accurate values)
        while q > qe and u[:q - qe].min() < 0.0:
            iters += 1
            l = int(np.argmin(u[:q - qe]))
            ul = u[l]
            c, side = act[l]
            drop(l)
            d = J.T @ normal(c, side)
            z = J[:, q:] @ d[q:]
            r = T[qe:q, qe:q] @ d[qe:q]
            x = x - ul * z
            u[:q - qe] += ul * r
    # ---- dual iterations (solve_v2's inequality phase)
    while True:
        worst, ip = 0.0, -1
        for c in range(ncon):
            if active[c] or is_eq(c):
                continue
            vv_ = value(c, x)
            if lo(c) > -INF:
                s = vv_ - lo(c)
                if s < -1e-9 * max(1.0, abs(lo(c))) and s < worst:
                    worst, ip, side, b_ip = s, c, 0, lo(c)
            if hi(c) < INF:
                s = hi(c) - vv_
                if s < -1e-9 * max(1.0, abs(hi(c))) and s < worst:
                    worst, ip, side, b_ip = s, c, 1, -hi(c)
        if ip < 0:
            return x, 0, iters, list(act)
        s_ip = worst
        npv = normal(ip, side)
        np2 = npv @ npv
        u_ip = 0.0
        while True:
            iters += 1
            if iters > max_iter:
                return x, 1, iters, list(act)
            d = J.T @ npv
            zn = d[q:] @ d[q:]
            z = J[:, q:] @ d[q:]
            r = T[qe:q, qe:q] @ d[qe:q]
            have_step = zn > 100.0 * n * EPS2 * jf2 * np2
            t1, l = np.inf, -1
            for k in range(q - qe):
                if r[k] > 0 and u[k] / r[k] < t1:
                    t1, l = u[k] / r[k], k
            t2 = -s_ip / zn if have_step else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return x, 2, iters, list(act)
            if have_step:
                x = x + t * z
            u[:q - qe] -= t * r
            u_ip += t
            if have_step and t == t2:
                dq = d[q]
                delta = -np.sqrt(zn) if dq >= 0 else np.sqrt(zn)
                v = d[q:].copy()
                v[0] -= delta
                vv = 2.0 * (zn - delta * dq)
                if vv > 0:
                    w = z - delta * J[:, q]
                    J[:, q:] -= np.outer(w, (2.0 / vv) * v)
                T[qe:q, q] = -r / delta
                T[q, q] = 1.0 / delta
                u[q - qe] = u_ip
                act.append((ip, side))
                active[ip] = True
                q += 1
                break
            drop(l)
            s_ip = (-1.0 if side else 1.0) * value(ip, x) - b_ip


def solve_v4(H, g, C=None, lb=None, ub=None, Clb=None, Cub=None, seeds=(), max_iter=None, far=0.25):
    """The PACKED kernel's warm start (wbc_tick_sim3p_kernel<WARM>; SURVEY.md §8 f2, QP_Wrapper.py:55-73) for a problem WITHOUT
    equalities (the contact presolve and the locked DoF removed them):
      1. x0 = -J0 J0'g; a seed is kept only if x0 violates it or comes within far * max(1, |x0|_inf) of it;
      2. every kept seed goes through the dual method's ADD step alone (Householder on J2, column (-T r / delta, 1 / delta) of T):
         no search, no ratio test, no partial step; dependent seeds are skipped;
      3. x, u from the factors: s_j = b_j - n_j'x0, w = T's, x = x0 + J1 w, u = T w;
      4. restoration (most negative multiplier dropped, x <- x - u_l z, u <- u + u_l r on the new factors), then 3. once more and a
         second restoration pass on the accurate multipliers;
      5. solve()'s dual iterations from that S-pair.
    Returns (x, status, iters, final working set [(constraint, side)])."""
    n = len(g)
    p = 0 if C is None else C.shape[0]
    ncon = n + p

    def lo(c):
        return (lb[c] if lb is not None else -1e30) if c < n else Clb[c - n]

    def hi(c):
        return (ub[c] if ub is not None else 1e30) if c < n else Cub[c - n]

    def normal(c, side):
        sgn = -1.0 if side else 1.0
        if c < n:
            e = np.zeros(n)
            e[c] = sgn
            return e
        return sgn * C[c - n]

    def value(c, x):
        return x[c] if c < n else C[c - n] @ x

    assert not any(lo(c) == hi(c) and abs(lo(c)) < INF for c in range(ncon)), "solve_v4: no equalities"
    try:
        L = np.linalg.cholesky(H)
    except np.linalg.LinAlgError:
        return np.zeros(n), 3, 0, []
    J = np.linalg.inv(L).T.copy()
    jf2 = (J * J).sum()
    x0 = -J @ (J.T @ g)
    x = x0.copy()
    T = np.zeros((n, n))
    u = np.zeros(n + 1)
    act, active = [], np.zeros(ncon, bool)
    q = 0
    iters = 0
    max_iter = max_iter or 10 * (n + p) + 20

    def add(c, side, d, zn, z, r, u_new):
        nonlocal q
        dq = d[q]
        delta = -np.sqrt(zn) if dq >= 0 else np.sqrt(zn)
        v = d[q:].copy()
        v[0] -= delta
        vv = 2.0 * (zn - delta * dq)
        if vv > 0:
            w = z - delta * J[:, q]
            J[:, q:] -= np.outer(w, (2.0 / vv) * v)
        T[:q, q] = -r / delta
        T[q, q] = 1.0 / delta
        u[q] = u_new
        act.append((c, side))
        active[c] = True
        q += 1

    def drop(l):
        nonlocal q
        trow = T[l, l:q].copy()
        active[act[l][0]] = False
        del act[l]
        u[l:q - 1] = u[l + 1:q].copy()
        u[q - 1] = 0.0
        Tt = np.delete(T[:q, :q], l, axis=0)
        h = trow[0]
        for k in range(l, q - 1):
            a_, b_ = h, trow[k - l + 1]
            rho = np.hypot(a_, b_)
            c_, s_ = (b_ / rho, -a_ / rho) if rho > 0 else (1.0, 0.0)
            h = rho
            ck, ck1 = Tt[:, k].copy(), Tt[:, k + 1].copy()
            Tt[:, k], Tt[:, k + 1] = c_ * ck + s_ * ck1, -s_ * ck + c_ * ck1
            jk, jk1 = J[:, k].copy(), J[:, k + 1].copy()
            J[:, k], J[:, k + 1] = c_ * jk + s_ * jk1, -s_ * jk + c_ * jk1
        T[:, :] = 0
        T[:q - 1, :q - 1] = np.triu(Tt[:, :q - 1])
        q -= 1

    # ---- 1, 2: seeds, bounds first then rows, lowest index first (the kernel's order); one side per constraint
    near = far * max(1.0, np.abs(x0).max()) if far is not None else np.inf
    todo, seen = [], {}
    for c, side in seeds:
        if 0 <= c < ncon:
            seen[c] = None if (c in seen and seen[c] != side) else side      # both sides claimed: dropped (the kernel's code 3)
    for c in sorted(seen):
        side = seen[c]
        if side is None:
            continue
        if (side == 0 and lo(c) > -INF) or (side == 1 and hi(c) < INF):
            todo.append((c, side))
    for c, side in todo:
        slack = value(c, x0) - lo(c) if side == 0 else hi(c) - value(c, x0)
        if not slack <= near:
            continue
        npv = normal(c, side)
        d = J.T @ npv
        zn = d[q:] @ d[q:]
        if not zn > 100.0 * n * EPS2 * jf2 * (npv @ npv):
            continue
        iters += 1
        add(c, side, d, zn, J[:, q:] @ d[q:], T[:q, :q] @ d[:q], 0.0)

    def refresh():
        nonlocal x
        sl = np.array([(lo(c) if sd == 0 else -hi(c)) - normal(c, sd) @ x0 for c, sd in act])
        w = T[:q, :q].T @ sl
        x = x0 + J[:, :q] @ w
        u[:q] = T[:q, :q] @ w

    def restore():
        nonlocal x, iters
        did = False
        while q > 0 and u[:q].min() < 0.0:
            iters += 1
            l = int(np.argmin(u[:q]))
            ul = u[l]
            c, side = act[l]
            drop(l)
            d = J.T @ normal(c, side)
            x = x - ul * (J[:, q:] @ d[q:])
            u[:q] += ul * (T[:q, :q] @ d[:q])
            did = True
        return did

    if q > 0:
        refresh()
        if restore():
            refresh()
            restore()
    # ---- 5: dual iterations
    while True:
        worst, ip = 0.0, -1
        for c in range(ncon):
            if active[c]:
                continue
            v = value(c, x)
            if lo(c) > -INF:
                s = v - lo(c)
                if s < -1e-9 * max(1.0, abs(lo(c))) and s < worst:
                    worst, ip, side, b_ip = s, c, 0, lo(c)
            if hi(c) < INF:
                s = hi(c) - v
                if s < -1e-9 * max(1.0, abs(hi(c))) and s < worst:
                    worst, ip, side, b_ip = s, c, 1, -hi(c)
        if ip < 0:
            return x, 0, iters, list(act)
        s_ip = worst
        npv = normal(ip, side)
        np2 = npv @ npv
        u_ip = 0.0
        while True:
            iters += 1
            if iters > max_iter:
                return x, 1, iters, list(act)
            d = J.T @ npv
            zn = d[q:] @ d[q:]
            z = J[:, q:] @ d[q:]
            r = T[:q, :q] @ d[:q]
            have_step = zn > 100.0 * n * EPS2 * jf2 * np2
            t1, l = np.inf, -1
            for k in range(q):
                if r[k] > 0 and u[k] / r[k] < t1:
                    t1, l = u[k] / r[k], k
            t2 = -s_ip / zn if have_step else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return x, 2, iters, list(act)
            if have_step:
                x = x + t * z
            u[:q] -= t * r
            u_ip += t
            if have_step and t == t2:
                add(ip, side, d, zn, z, r, u_ip)
                break
            drop(l)
            s_ip = (-1.0 if side else 1.0) * value(ip, x) - b_ip


def solve_v5(H, g, C=None, lb=None, ub=None, Clb=None, Cub=None, seeds=(), max_iter=None, far=0.25):
    """The PACKED STAND-ALONE QP kernel's flow (csrc/wbc_k_qpp.hip wbc_qp_packed_kernel) in plain numpy — a general problem, cold or
    hot-started:
      0. variables with lb == ub are presolved out (H_kk = 1, g_k = -value; their columns leave H and C, the value moves into g and the row bounds);
      1. equality rows (Clb == Cub) enter first, in index order, through the dual method's ADD step alone (never droppable); their right-hand sides
         ride along as R'y1 = b_e (y1_q = (b_e - d1'y1) / delta), a dependent row is skipped if consistent and is INFEASIBLE otherwise;
         x_eq = J1 y1 - J2 J2'g comes from the factors;
      2. hot start: a carried inequality is taken (add step, no primal step) only if x_eq violates it or comes within far * max(1, |x_eq|_inf) of it;
         x and u are then read off the factors: b = the slots' right-hand sides, y1 = T'b, x = J1 y1 - J2 J2'g, u = T (y1 + J1'g);
         while a seeded multiplier is negative its slot is dropped and x, u are read off again;
      3. the dual iterations; only the inequality slots (>= qe) can block.
    T = R^-1 covers every slot (the kernel keeps it packed). Returns (x, status, iters, final working set [(constraint, side)] of INEQUALITIES)."""
    n = len(g)
    p = 0 if C is None else C.shape[0]
    ncon = n + p
    H = np.array(H, dtype=float)
    g = np.array(g, dtype=float)
    C = np.zeros((0, n)) if C is None else np.array(C, dtype=float)
    lo_b = np.full(n, -1e30) if lb is None else np.array(lb, dtype=float)
    hi_b = np.full(n, 1e30) if ub is None else np.array(ub, dtype=float)
    lo_r = np.zeros(0) if Clb is None else np.array(Clb, dtype=float)
    hi_r = np.zeros(0) if Cub is None else np.array(Cub, dtype=float)
    if np.isnan(lo_b).any() or np.isnan(hi_b).any() or np.isnan(lo_r).any() or np.isnan(hi_r).any():
        return np.zeros(n), 3, 0, []
    # ---- 0: presolve
    fix = (lo_b == hi_b) & (np.abs(lo_b) < INF)
    iters = int(fix.sum())
    if fix.any():
        fv = np.where(fix, lo_b, 0.0)
        g = g + H @ fv
        cs = C @ fv
        lo_r, hi_r = lo_r - cs, hi_r - cs
        H[fix, :] = 0.0
        H[:, fix] = 0.0
        C[:, fix] = 0.0
        for k in np.nonzero(fix)[0]:
            H[k, k] = 1.0
            g[k] = -fv[k]
        lo_b, hi_b = np.where(fix, -1e30, lo_b), np.where(fix, 1e30, hi_b)

    def lo(c):
        return lo_b[c] if c < n else lo_r[c - n]

    def hi(c):
        return hi_b[c] if c < n else hi_r[c - n]

    def normal(c, side):
        sgn = -1.0 if side else 1.0
        if c < n:
            e = np.zeros(n)
            e[c] = sgn
            return e
        return sgn * C[c - n]

    def value(c, x):
        return x[c] if c < n else C[c - n] @ x

    try:
        L = np.linalg.cholesky(H)
    except np.linalg.LinAlgError:
        return np.zeros(n), 3, 0, []
    J = np.linalg.inv(L).T.copy()
    jf2 = (J * J).sum()
    T = np.zeros((n, n))
    u = np.zeros(n + 1)
    act, active = [], np.zeros(ncon, bool)
    q = 0
    max_iter = max_iter or 10 * (n + p) + 20

    def add(c, side, d, zn, z, r, u_new):
        nonlocal q
        dq = d[q]
        delta = -np.sqrt(zn) if dq >= 0 else np.sqrt(zn)
        v = d[q:].copy()
        v[0] -= delta
        vv = 2.0 * (zn - delta * dq)
        if vv > 0:
            w = z - delta * J[:, q]
            J[:, q:] -= np.outer(w, (2.0 / vv) * v)
        T[:q, q] = -r / delta
        T[q, q] = 1.0 / delta
        u[q] = u_new
        act.append((c, side))
        active[c] = True
        q += 1
        return delta

    def drop(l):
        nonlocal q
        trow = T[l, l:q].copy()
        active[act[l][0]] = False
        del act[l]
        u[l:q - 1] = u[l + 1:q].copy()
        u[q - 1] = 0.0
        Tt = np.delete(T[:q, :q], l, axis=0)
        h = trow[0]
        for k in range(l, q - 1):
            a_, b_ = h, trow[k - l + 1]
            rho = np.hypot(a_, b_)
            c_, s_ = (b_ / rho, -a_ / rho) if rho > 0 else (1.0, 0.0)
            h = rho
            ck, ck1 = Tt[:, k].copy(), Tt[:, k + 1].copy()
            Tt[:, k], Tt[:, k + 1] = c_ * ck + s_ * ck1, -s_ * ck + c_ * ck1
            jk, jk1 = J[:, k].copy(), J[:, k + 1].copy()
            J[:, k], J[:, k + 1] = c_ * jk + s_ * jk1, -s_ * jk + c_ * jk1
        T[:, :] = 0
        T[:q - 1, :q - 1] = np.triu(Tt[:, :q - 1])
        q -= 1

    # ---- 1: equality rows
    eq_r = [(lo_r[r_] == hi_r[r_]) and abs(lo_r[r_]) < INF for r_ in range(p)]
    y1 = np.zeros(n)
    for r_ in range(p):
        if not eq_r[r_]:
            continue
        iters += 1
        npv = normal(n + r_, 0)
        d = J.T @ npv
        zn = d[q:] @ d[q:]
        dy = d[:q] @ y1[:q]
        b_e = lo_r[r_]
        if zn > 100.0 * n * EPS2 * jf2 * (npv @ npv):
            qold = q
            delta = add(n + r_, 0, d, zn, J[:, q:] @ d[q:], T[:q, :q] @ d[:q], 0.0)
            y1[qold] = (b_e - dy) / delta
        elif not abs(dy - b_e) <= 1e-9 * max(1.0, abs(b_e)):
            return np.zeros(n), 2, iters, []
    qe = q
    x = J[:, :q] @ y1[:q] - J[:, q:] @ (J[:, q:].T @ g)

    def rebuild():
        nonlocal x
        b = np.array([lo(c) if sd == 0 else -hi(c) for c, sd in act])
        y = T[:q, :q].T @ b
        jg = J.T @ g
        x = J[:, :q] @ y - J[:, q:] @ jg[q:]
        u[:q] = T[:q, :q] @ (y + jg[:q])
        u[:qe] = 0.0

    # ---- 2: hot start
    if len(seeds):
        near = far * max(1.0, np.abs(x).max())
        seen = {}
        for c, side in seeds:
            if 0 <= c < ncon:
                seen[c] = None if (c in seen and seen[c] != side) else side
        x_eq = x.copy()
        seeded = False
        for c in sorted(seen):
            side = seen[c]
            if side is None or (c >= n and eq_r[c - n]):
                continue
            if not ((side == 0 and lo(c) > -INF) or (side == 1 and hi(c) < INF)):
                continue
            slack = value(c, x_eq) - lo(c) if side == 0 else hi(c) - value(c, x_eq)
            if not slack <= near:
                continue
            npv = normal(c, side)
            d = J.T @ npv
            zn = d[q:] @ d[q:]
            if not zn > 100.0 * n * EPS2 * jf2 * (npv @ npv):
                continue
            iters += 1
            add(c, side, d, zn, J[:, q:] @ d[q:], T[:q, :q] @ d[:q], 0.0)
            seeded = True
        if seeded:
            rebuild()
            while q > qe and u[qe:q].min() < 0.0:
                iters += 1
                if iters > max_iter:
                    return np.zeros(n), 1, iters, []
                drop(qe + int(np.argmin(u[qe:q])))
                rebuild()
    # ---- 3: dual iterations
    while True:
        worst, ip = 0.0, -1
        for c in range(ncon):
            if active[c] or (c >= n and eq_r[c - n]):
                continue
            v = value(c, x)
            if lo(c) > -INF:
                s = v - lo(c)
                if s < -1e-9 * max(1.0, abs(lo(c))) and s < worst:
                    worst, ip, side, b_ip = s, c, 0, lo(c)
            if hi(c) < INF:
                s = hi(c) - v
                if s < -1e-9 * max(1.0, abs(hi(c))) and s < worst:
                    worst, ip, side, b_ip = s, c, 1, -hi(c)
        if ip < 0:
            if not np.isfinite(x).all():
                return np.zeros(n), 3, iters, []
            return x, 0, iters, [a for a in act[qe:]]
        s_ip = worst
        npv = normal(ip, side)
        np2 = npv @ npv
        u_ip = 0.0
        while True:
            iters += 1
            if iters > max_iter:
                return np.zeros(n), 1, iters, []
            d = J.T @ npv
            zn = d[q:] @ d[q:]
            z = J[:, q:] @ d[q:]
            r = T[:q, :q] @ d[:q]
            have_step = zn > 100.0 * n * EPS2 * jf2 * np2
            t1, l = np.inf, -1
            for k in range(qe, q):
                if r[k] > 0 and u[k] / r[k] < t1:
                    t1, l = u[k] / r[k], k
            t2 = -s_ip / zn if have_step else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return np.zeros(n), 2, iters, []
            if have_step:
                x = x + t * z
            u[:q] -= t * r
            u_ip += t
            if have_step and t == t2:
                add(ip, side, d, zn, z, r, u_ip)
                break
            drop(l)
            s_ip = (-1.0 if side else 1.0) * value(ip, x) - b_ip
