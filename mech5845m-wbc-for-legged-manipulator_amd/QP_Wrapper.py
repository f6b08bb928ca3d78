"""MI355X mirror of the reference's ``wrappers/QP_Wrapper.py`` (class ``QP``): same constructor, methods and attributes.

    QP(A, b, lb, ub, C=None, Clb=None, Cub=None, n_of_velocity_dimensions=None)
    QP.solveQP() -> xOpt                                   (reference QP_Wrapper.py:23-53)
    QP.solveQPHotstart(A, b, lb, ub, C, Clb, Cub) -> xOpt  (reference QP_Wrapper.py:55-73)

The arithmetic — ``H = A'A``, ``g = -A'b`` (QP_Wrapper.py:17-18, 66-67) and the QP that the reference hands to qpOASES —
runs in the HIP kernel behind ``wbc_qp_solve_ls`` (include/wbc.h); this is its B = 1 use. There is no CPU fallback.

Kept from the reference: ``C`` is what ``findConstraints`` returns, i.e. the (n, p) transposed view of a row-major
p x n array, and the solver reads it as p rows of n; ``lb``/``ub`` longer than n are read up to n
(Robot_Wrapper2 passes 2n values); ``xOpt`` is allocated once by ``solveQP`` and the SAME ndarray is overwritten and
returned by every ``solveQPHotstart``; misuse of ``solveQPHotstart`` on a bounds-only QP prints and exits.
Added (the reference drops qpOASES' return codes): ``status`` (0 optimal, 1 iteration cap, 2 infeasible,
3 numerical) and ``nWSR`` holding the number of working-set changes performed, qpOASES' in/out convention.
"""
import numpy as np

import wbc_capi as capi
from wbc_batch import WbcBatch

_shared = None


def _batch():
    """One QP-only handle (no robot model) shared by every QP object of the process."""
    global _shared
    if _shared is None:
        _shared = WbcBatch([], max_batch=1)
    return _shared


class QP:
    def __init__(self, A, b, lb, ub, C=None, Clb=None, Cub=None, n_of_velocity_dimensions=None):
        self.lb = lb
        self.ub = ub
        self.Clb = Clb
        self.Cub = Cub
        self.C = C
        self._A, self._b = A, b
        self._H = self._g = None
        self.no_solutions = n_of_velocity_dimensions if n_of_velocity_dimensions is not None else np.asarray(A).shape[1]
        self.nWSR = np.array([100000])
        self.qp = None
        self.status = None
        self.use_mfma = None     # None: the library decides by the row count of A; True / False force the path
        self._ws = None          # the last solve's working set (qpOASES keeps its own between init and hotstart, :45-48, :70)

    # H, g are attributes in the reference (computed eagerly with numpy); here the device forms them on demand.
    @property
    def H(self):
        if self._H is None:
            self._solve(form_only=True)
        return self._H

    @property
    def g(self):
        if self._g is None:
            self._solve(form_only=True)
        return self._g

    def _solve(self, form_only=False, hot=False):
        n = int(self.no_solutions)
        A = np.ascontiguousarray(self._A, dtype=np.float64)
        if A.ndim != 2 or A.shape[1] != n:
            raise ValueError("A must be (m, %d), got %s" % (n, A.shape))
        b = np.ascontiguousarray(self._b, dtype=np.float64).reshape(-1)
        if b.shape[0] != A.shape[0]:
            raise ValueError("b has %d entries for %d rows of A" % (b.shape[0], A.shape[0]))
        lb = np.ascontiguousarray(self.lb, dtype=np.float64).reshape(-1)[:n]
        ub = np.ascontiguousarray(self.ub, dtype=np.float64).reshape(-1)[:n]
        bounded_only = self.C is None or self.Clb is None or self.Cub is None
        Cr = cl = cu = None
        if not bounded_only and not form_only:
            Cr = np.ascontiguousarray(np.asarray(self.C, dtype=np.float64).T)      # (p, n) rows
            cl = np.ascontiguousarray(self.Clb, dtype=np.float64).reshape(-1)
            cu = np.ascontiguousarray(self.Cub, dtype=np.float64).reshape(-1)
            if Cr.shape != (cl.shape[0], n) or cu.shape != cl.shape:
                raise ValueError("C must be (n, p) with p = len(Clb) = len(Cub)")
            Cr, cl, cu = Cr[None], cl[None], cu[None]
        bt = _batch()
        self.qp = bt
        ws_in = self._ws if (hot and self._ws is not None and not form_only) else None
        x, st, it, H, g, ws = bt.qp_solve_ls(A[None], b[None], Cr, lb[None], ub[None], cl, cu, use_mfma=self.use_mfma, want_Hg=True,
                                             working_set=ws_in, want_working_set=True)
        self._H, self._g = H[0], g[0]
        if form_only:
            return None
        if int(st[0]) == 0:         # (an unsolved QP keeps the set of the last solved one, like the stale xOpt)
            self._ws = ws
        self.status = int(st[0])
        self.nWSR = np.array([int(it[0])])
        return x[0]

    def solveQP(self):
        self._ws = None                 # a cold solve builds a fresh QProblem (QP_Wrapper.py:26-29): no working set is carried into it or past a failed one
        x = self._solve()
        self.xOpt = np.zeros((self.no_solutions,))
        if self.status == 0:            # qpOASES' getPrimalSolution leaves its argument alone for an unsolved QP (:50-51)
            self.xOpt[:] = x
        return self.xOpt

    def solveQPHotstart(self, A, b, lb, ub, C, Clb, Cub):
        if self.Clb is None or self.Cub is None:
            print("Error, cannot hotstart simply bounded QP")
            exit()
        self.lb = lb
        self.ub = ub
        self.Clb = Clb
        self.Cub = Cub
        self.C = C
        self._A, self._b = A, b
        self._H = self._g = None
        self.nWSR = np.array([100000])
        x = self._solve(hot=True)       # qp.hotstart (:70): seeded with the previous solve's working set (wbc_qp_solve_ls working_set_in)
        if self.status == 0:            # unsolved: xOpt keeps the previous tick's answer, as in the reference (:71-73)
            self.xOpt[:] = x
        return self.xOpt
