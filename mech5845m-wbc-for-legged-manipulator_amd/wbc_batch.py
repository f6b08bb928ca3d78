"""Batched front end of the HIP library: the many-instance form of ``RobotModel`` / ``QP``.

``WbcBatch`` owns the C handles and maps dictionaries of arrays onto the ``WbcTickIn`` / ``WbcTickOut`` /
``WbcQpData`` / ``WbcFkOut`` structs of include/wbc.h. Arrays may be numpy (host: the library stages them over
PCIe) or torch CUDA tensors (device: pointers pass straight through, nothing is copied or synchronised).
The single-instance classes in QP_Wrapper.py / Robot_Wrapper4.py are the B = 1 case of this object.
"""
import ctypes as C

import numpy as np

import wbc_capi as capi

NV, NQS = capi.V_STRIDE, capi.Q_STRIDE


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _mem_of(arrays):
    kinds = {("dev" if (_is_torch(a) and a.is_cuda) else "host") for a in arrays if a is not None}
    if len(kinds) > 1:
        raise capi.WbcError("mixing host and device arrays in one call")
    return capi.MEM_DEVICE if kinds == {"dev"} else capi.MEM_HOST


def _prep(a, dtype, keep):
    """-> (pointer, array kept alive). numpy arrays are made contiguous float64/int32; torch tensors are checked."""
    if a is None:
        return None
    if _is_torch(a):
        import torch
        want = torch.float64 if dtype == np.float64 else torch.int32
        if a.dtype != want or not a.is_contiguous():
            raise capi.WbcError("device tensors must be contiguous %s" % want)
        keep.append(a)
        return a.data_ptr()
    arr = np.ascontiguousarray(a, dtype=dtype)
    keep.append(arr)
    return arr.ctypes.data


def _stream(mem):
    if mem == capi.MEM_DEVICE:
        import torch
        return torch.cuda.current_stream().cuda_stream
    return None


class WbcBatch:
    def __init__(self, models, max_batch, device_id=0):
        self.lib = capi.load_library()
        self.models = list(models) if isinstance(models, (list, tuple)) else [models]
        self.max_batch = int(max_batch)
        self._mh = []
        for m in self.models:
            h = C.c_void_p()
            capi.check(self.lib.wbc_model_create(C.byref(m.blob), C.byref(h)), self.lib)
            self._mh.append(h)
        arr = (C.c_void_p * len(self._mh))(*[h.value for h in self._mh])
        self._h = C.c_void_p()
        capi.check(self.lib.wbc_batch_create(arr, len(self._mh), self.max_batch, device_id, C.byref(self._h)), self.lib)
        self.cfgs = [None] * len(self.models)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.wbc_batch_destroy(self._h)
            self._h = None
        for h in getattr(self, "_mh", []):
            self.lib.wbc_model_destroy(h)
        self._mh = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- settings
    def configure(self, cfg, model_index=0):
        capi.check(self.lib.wbc_batch_configure(self._h, model_index, C.byref(cfg)), self.lib)
        self.cfgs[model_index] = cfg

    def set_option(self, name, value):
        capi.check(self.lib.wbc_batch_set_option(self._h, name.encode(), int(value)), self.lib)

    @property
    def task_rows(self):
        return self.lib.wbc_task_rows(self._h)

    @property
    def constraint_rows(self):
        return self.lib.wbc_constraint_rows(self._h)

    def debug_cycles(self):
        """per-phase cycle sums since the last call (profile build only; see include/wbc.h)."""
        out = (C.c_uint64 * 24)()
        capi.check(self.lib.wbc_debug_cycles(self._h, out), self.lib)
        return [int(v) for v in out]

    def synchronize(self, stream=None):
        capi.check(self.lib.wbc_batch_synchronize(self._h, stream), self.lib)

    # ---- helpers
    def _tick_in(self, inputs, keep):
        t = capi.WbcTickIn()
        for name, _ in capi.WbcTickIn._fields_:
            a = inputs.get(name)
            if a is not None:
                setattr(t, name, _prep(a, np.int32 if name == "model_id" else np.float64, keep))
        return t

    def _alloc(self, like, shape, dtype=np.float64):
        if like is not None and _is_torch(like):
            import torch
            return torch.empty(shape, dtype=torch.float64 if dtype == np.float64 else torch.int32, device=like.device)
        return np.empty(shape, dtype=dtype)

    # ---- entry points
    def fk(self, q, model_id=None, want=("oMi", "oMf", "J", "com", "Jcom")):
        """updateState's kinematics: returns dict(oMi [B,nj,12], oMf [B,nf,12], J [B,6,26], com [B,3], Jcom [B,3,26])."""
        keep = []
        mem = _mem_of([q, model_id])
        B = q.shape[0]
        nj, nf = self.models[0].njoints, self.models[0].blob.nframes
        shapes = dict(oMi=(B, nj, 12), oMf=(B, nf, 12), J=(B, 6, NV), com=(B, 3), Jcom=(B, 3, NV))
        out = {k: self._alloc(q, shapes[k]) for k in want}
        o = capi.WbcFkOut()
        for k, v in out.items():
            setattr(o, k, _prep(v, np.float64, keep))
        capi.check(self.lib.wbc_fk_jacobians(self._h, B, _prep(q, np.float64, keep), _prep(model_id, np.int32, keep), mem,
                                              C.byref(o), _stream(mem)), self.lib)
        return out

    def assemble(self, inputs, dt, want=("A", "b", "H", "g", "C", "Clb", "Cub", "lb", "ub")):
        """qpA/qpb/findConstraints/velDamperJointConstraints + H, g for every instance."""
        keep = []
        mem = _mem_of(list(inputs.values()))
        q = inputs["q"]
        B = q.shape[0]
        m, p = self.task_rows, self.constraint_rows
        shapes = dict(A=(B, m, NV), b=(B, m), H=(B, NV, NV), g=(B, NV), C=(B, p, NV), Clb=(B, p), Cub=(B, p), lb=(B, NV), ub=(B, NV))
        out = {k: self._alloc(q, shapes[k]) for k in want}
        o = capi.WbcQpData()
        for k, v in out.items():
            setattr(o, k, _prep(v, np.float64, keep))
        tin = self._tick_in(inputs, keep)
        capi.check(self.lib.wbc_assemble(self._h, B, C.byref(tin), float(dt), mem, C.byref(o), _stream(mem)), self.lib)
        return out

    def tick(self, inputs, dt, want_q_next=False, out=None):
        """One runWBC tick per instance up to the QP (+ integrate): returns dict(qdot, status, iters[, q_next])."""
        keep = []
        mem = _mem_of(list(inputs.values()))
        q = inputs["q"]
        B = q.shape[0]
        if out is None:
            out = dict(qdot=self._alloc(q, (B, NV)), status=self._alloc(q, (B,), np.int32), iters=self._alloc(q, (B,), np.int32))
            if want_q_next:
                out["q_next"] = self._alloc(q, (B, NQS))
        o = capi.WbcTickOut()
        for k, v in out.items():
            setattr(o, k, _prep(v, np.int32 if k in ("status", "iters") else np.float64, keep))
        tin = self._tick_in(inputs, keep)
        capi.check(self.lib.wbc_tick(self._h, B, C.byref(tin), float(dt), mem, C.byref(o), _stream(mem)), self.lib)
        return out

    def make_tick_call(self, inputs, out, dt):
        """Bind device tensors once; the returned closure issues exactly one wbc_tick on the current stream."""
        keep = []
        if _mem_of(list(inputs.values()) + list(out.values())) != capi.MEM_DEVICE:
            raise capi.WbcError("make_tick_call wants device tensors")
        B = inputs["q"].shape[0]
        o = capi.WbcTickOut()
        for k, v in out.items():
            setattr(o, k, _prep(v, np.int32 if k in ("status", "iters") else np.float64, keep))
        tin = self._tick_in(inputs, keep)
        lib, h, dtv = self.lib, self._h, float(dt)
        import torch

        def call():
            rc = lib.wbc_tick(h, B, C.byref(tin), dtv, capi.MEM_DEVICE, C.byref(o), torch.cuda.current_stream().cuda_stream)
            if rc:
                capi.check(rc, lib)
        call._keep = keep
        return call

    def qp_solve(self, H, g, C_=None, lb=None, ub=None, Clb=None, Cub=None):
        """Batched QP.solveQP given H, g. H [B,n,n], C_ [B,p,n] (row-major rows), returns (x, status, iters)."""
        keep = []
        mem = _mem_of([H, g, C_, lb, ub, Clb, Cub])
        B, n = H.shape[0], H.shape[-1]
        p = 0 if C_ is None else C_.shape[-2]
        x, st, it = self._alloc(H, (B, n)), self._alloc(H, (B,), np.int32), self._alloc(H, (B,), np.int32)
        f = np.float64
        capi.check(self.lib.wbc_qp_solve(self._h, B, n, p, _prep(H, f, keep), _prep(g, f, keep), _prep(C_, f, keep),
                                          _prep(lb, f, keep), _prep(ub, f, keep), _prep(Clb, f, keep), _prep(Cub, f, keep), mem,
                                          _prep(x, f, keep), _prep(st, np.int32, keep), _prep(it, np.int32, keep), _stream(mem)), self.lib)
        return x, st, it

    def qp_solve_ls(self, A, b, C_=None, lb=None, ub=None, Clb=None, Cub=None, use_mfma=False, want_Hg=False):
        """Batched QP(A, b, ...).solveQP(): H = A'A and g = -A'b are formed on the device (QP_Wrapper.py:17-18)."""
        keep = []
        mem = _mem_of([A, b, C_, lb, ub, Clb, Cub])
        B, m, n = A.shape
        p = 0 if C_ is None else C_.shape[-2]
        x, st, it = self._alloc(A, (B, n)), self._alloc(A, (B,), np.int32), self._alloc(A, (B,), np.int32)
        Ho = self._alloc(A, (B, n, n)) if want_Hg else None
        go = self._alloc(A, (B, n)) if want_Hg else None
        f = np.float64
        capi.check(self.lib.wbc_qp_solve_ls(self._h, B, m, n, p, _prep(A, f, keep), _prep(b, f, keep), _prep(C_, f, keep),
                                             _prep(lb, f, keep), _prep(ub, f, keep), _prep(Clb, f, keep), _prep(Cub, f, keep), mem,
                                             int(bool(use_mfma)), _prep(x, f, keep), _prep(st, np.int32, keep), _prep(it, np.int32, keep),
                                             _prep(Ho, f, keep), _prep(go, f, keep), _stream(mem)), self.lib)
        return (x, st, it, Ho, go) if want_Hg else (x, st, it)

    def posture_target(self, q, model_id=None, want_q_after=True):
        """qpJointb's "MANI" / "HYBRID" posture target u [B,26] under the configured mode (Robot_Wrapper4.py:1220-1260),
        and the configuration the reference's state is left at, q_after [B,27]."""
        keep = []
        mem = _mem_of([q, model_id])
        B = q.shape[0]
        u = self._alloc(q, (B, NV))
        qa = self._alloc(q, (B, NQS)) if want_q_after else None
        f = np.float64
        capi.check(self.lib.wbc_posture_target(self._h, B, _prep(q, f, keep), _prep(model_id, np.int32, keep), mem,
                                                _prep(u, f, keep), _prep(qa, f, keep), _stream(mem)), self.lib)
        return (u, qa) if want_q_after else u

    def update_state(self, q_cur, q_next, foot_targets, imu=None, model_id=None):
        """The tail of runWBC (Robot_Wrapper4.py:1397-1399): updateState(joint_config, base_config, running=True) with the
        foot-anchored base estimator trunkWorldPos (:1297-1327). Returns the new current_joint_config [B,27]."""
        keep = []
        mem = _mem_of([q_cur, q_next, foot_targets, imu, model_id])
        B = q_cur.shape[0]
        qn = self._alloc(q_cur, (B, NQS))
        f = np.float64
        capi.check(self.lib.wbc_update_state(self._h, B, _prep(q_cur, f, keep), _prep(q_next, f, keep), _prep(imu, f, keep),
                                              _prep(foot_targets, f, keep), _prep(model_id, np.int32, keep), mem,
                                              _prep(qn, f, keep), _stream(mem)), self.lib)
        return qn

    def rollout(self, inputs, dt, ticks, ee_target_step=None, trunk_target_step=None, imu=None, want_trace=True):
        """K closed-loop ticks on the device: tick -> update_state -> reference-state side effects ->
        targets advance by their step. Returns dict(q, qdot, ee_target, status, iters[, grip_trace [K,B,3]])."""
        keep = []
        extra = [ee_target_step, trunk_target_step, imu]
        mem = _mem_of(list(inputs.values()) + extra)
        q = inputs["q"]
        B = q.shape[0]
        out = dict(q=self._alloc(q, (B, NQS)), qdot=self._alloc(q, (B, NV)), ee_target=self._alloc(q, (B, 5, 3)),
                   status=self._alloc(q, (B,), np.int32), iters=self._alloc(q, (B,), np.int32))
        if want_trace:
            out["grip_trace"] = self._alloc(q, (int(ticks), B, 3))
        r = capi.WbcRollout()
        r.ticks = int(ticks)
        f = np.float64
        r.ee_target_step, r.trunk_target_step, r.imu = _prep(ee_target_step, f, keep), _prep(trunk_target_step, f, keep), _prep(imu, f, keep)
        r.q_final, r.qdot_last, r.ee_target_final = _prep(out["q"], f, keep), _prep(out["qdot"], f, keep), _prep(out["ee_target"], f, keep)
        r.status_max, r.iters_sum = _prep(out["status"], np.int32, keep), _prep(out["iters"], np.int32, keep)
        if want_trace:
            r.grip_trace = _prep(out["grip_trace"], f, keep)
        tin = self._tick_in(inputs, keep)
        capi.check(self.lib.wbc_rollout(self._h, B, C.byref(tin), float(dt), C.byref(r), mem, _stream(mem)), self.lib)
        return out

    def integrate(self, q, v, dt, model_id=None):
        """pin.integrate(model, q, v * dt) for every instance."""
        keep = []
        mem = _mem_of([q, v, model_id])
        B = q.shape[0]
        qn = self._alloc(q, (B, NQS))
        f = np.float64
        capi.check(self.lib.wbc_integrate(self._h, B, _prep(q, f, keep), _prep(v, f, keep), _prep(model_id, np.int32, keep),
                                           float(dt), mem, _prep(qn, f, keep), _stream(mem)), self.lib)
        return qn
