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


# doubles (int32 for model_id) per instance of every WbcTickIn field (include/wbc.h)
TICK_IN_WIDTH = dict(q=NQS, ee_target=15, prev_ee_target=15, trunk_target=3, prev_trunk_target=3, trunk_box_center=4,
                     ee_ref_rot=45, ee_prev_rot=45, trunk_ref_euler=3, trunk_prev_rot=9, com_target=3, com_target_vel=3,
                     model_id=1, posture_u=NV, q_con=NQS, working_set=2)
_INT32_FIELDS, _INT64_FIELDS = ("model_id", "status", "iters"), ("working_set",)


def _dtype_of(name):
    return np.int32 if name in _INT32_FIELDS else (np.int64 if name in _INT64_FIELDS else np.float64)


def _prep(a, dtype, keep, B=None, width=None, name="array", device_id=None):
    """-> pointer (the array is kept alive in `keep`). numpy arrays are made contiguous float64/int32; torch tensors are
    checked. With B / width given the array must hold exactly B x width values with B as its leading dimension: the
    kernels index raw pointers with those strides, so a wrong shape would read (or write) its neighbours' data."""
    if a is None:
        return None
    if B is not None:
        shape = tuple(a.shape)
        if len(shape) < 1 or shape[0] != B or int(np.prod(shape)) != B * width:
            raise capi.WbcError("%s: shape %s does not hold %d x %d values (leading dimension = batch)" % (name, shape, B, width))
    if _is_torch(a):
        import torch
        want = {np.float64: torch.float64, np.int32: torch.int32, np.int64: torch.int64}[dtype]
        if a.dtype != want or not a.is_contiguous():
            raise capi.WbcError("%s: device tensors must be contiguous %s" % (name, want))
        if a.is_cuda and device_id is not None and a.device.index != device_id:
            raise capi.WbcError("%s lives on cuda:%s, the handle on cuda:%d" % (name, a.device.index, device_id))
        keep.append(a)
        return a.data_ptr()
    arr = np.ascontiguousarray(a, dtype=dtype)
    keep.append(arr)
    return arr.ctypes.data


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
        self.device_id = int(device_id)
        capi.check(self.lib.wbc_batch_create(arr, len(self._mh), self.max_batch, device_id, C.byref(self._h)), self.lib)
        self.cfgs = [None] * len(self.models)
        self.max_nj = max((m.njoints for m in self.models), default=0)          # FK output strides (include/wbc.h, WbcFkOut)
        self.max_nf = max((m.blob.nframes for m in self.models), default=0)

    def _stream(self, mem):
        """the handle's device's current torch stream for device buffers, the null stream for host buffers"""
        if mem == capi.MEM_DEVICE:
            import torch
            return torch.cuda.current_stream(self.device_id).cuda_stream
        return None

    def _p(self, a, dtype, keep, B=None, width=None, name="array"):
        return _prep(a, dtype, keep, B, width, name, self.device_id)

    def _batch_of(self, q, name="q"):
        if q is None or len(q.shape) != 2 or q.shape[1] != NQS:
            raise capi.WbcError("%s must be [B, %d] (padded to the largest model's nq), got %s" % (
                name, NQS, None if q is None else tuple(q.shape)))
        B = int(q.shape[0])
        if B < 1 or B > self.max_batch:
            raise capi.WbcError("B = %d outside [1, max_batch = %d]" % (B, self.max_batch))
        return B

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
        self._options = dict(getattr(self, "_options", {}), **{name: int(value)})

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

    def stat(self, name, stream=None):
        """wbc_batch_get_stat: "last_path" (0 general, 1 compact sim3, 2 packed sim3, 3 packed orth — equality-only or INEQ variant —, 4 packed box), "last_orth", "last_posture_par",
        "last_update_packed", "last_qp_path" (problems per wavefront of the last stand-alone QP call: 1, 2 or 4), "deferred_last", "pivoted_last", "sim3_lds_bytes", "tick_lds_bytes", "orthp_lds_bytes"."""
        v = C.c_int64()
        capi.check(self.lib.wbc_batch_get_stat(self._h, name.encode(), stream, C.byref(v)), self.lib)
        return int(v.value)

    # ---- helpers
    def _tick_in(self, inputs, keep, B):
        unknown = set(inputs) - set(TICK_IN_WIDTH)
        if unknown:
            raise capi.WbcError("unknown tick inputs %s" % sorted(unknown))
        t = capi.WbcTickIn()
        for name, _ in capi.WbcTickIn._fields_:
            a = inputs.get(name)
            if a is not None:
                setattr(t, name, self._p(a, _dtype_of(name), keep, B, TICK_IN_WIDTH[name], name))
        return t

    def _outs(self, out, widths, keep, B, struct):
        for k, v in out.items():
            if k not in widths:
                raise capi.WbcError("unknown output %r" % k)
            setattr(struct, k, self._p(v, _dtype_of(k), keep, B, widths[k], k))
        return struct

    def _alloc(self, like, shape, dtype=np.float64):
        if like is not None and _is_torch(like):
            import torch
            return torch.empty(shape, dtype={np.float64: torch.float64, np.int32: torch.int32, np.int64: torch.int64}[dtype], device=like.device)
        return np.empty(shape, dtype=dtype)

    # ---- entry points
    def fk(self, q, model_id=None, want=("oMi", "oMf", "J", "com", "Jcom")):
        """updateState's kinematics: returns dict(oMi [B,nj,12], oMf [B,nf,12], J [B,6,26], com [B,3], Jcom [B,3,26]);
        nj / nf are the LARGEST model's joint / frame counts (rows beyond an instance's own model are zero)."""
        keep = []
        mem = _mem_of([q, model_id])
        B = self._batch_of(q)
        nj, nf = self.max_nj, self.max_nf
        shapes = dict(oMi=(B, nj, 12), oMf=(B, nf, 12), J=(B, 6, NV), com=(B, 3), Jcom=(B, 3, NV))
        out = {k: self._alloc(q, shapes[k]) for k in want}
        o = capi.WbcFkOut()
        for k, v in out.items():
            setattr(o, k, self._p(v, np.float64, keep))
        capi.check(self.lib.wbc_fk_jacobians(self._h, B, self._p(q, np.float64, keep), self._p(model_id, np.int32, keep, B, 1, "model_id"),
                                              mem, C.byref(o), self._stream(mem)), self.lib)
        return out

    def assemble(self, inputs, dt, want=("A", "b", "H", "g", "C", "Clb", "Cub", "lb", "ub")):
        """qpA/qpb/findConstraints/velDamperJointConstraints + H, g for every instance."""
        keep = []
        mem = _mem_of(list(inputs.values()))
        q = inputs.get("q")
        B = self._batch_of(q)
        m, p = self.task_rows, self.constraint_rows
        shapes = dict(A=(B, m, NV), b=(B, m), H=(B, NV, NV), g=(B, NV), C=(B, p, NV), Clb=(B, p), Cub=(B, p), lb=(B, NV), ub=(B, NV))
        out = {k: self._alloc(q, shapes[k]) for k in want}
        o = capi.WbcQpData()
        for k, v in out.items():
            setattr(o, k, self._p(v, np.float64, keep))
        tin = self._tick_in(inputs, keep, B)
        capi.check(self.lib.wbc_assemble(self._h, B, C.byref(tin), float(dt), mem, C.byref(o), self._stream(mem)), self.lib)
        return out

    _TICK_OUT_WIDTH = dict(qdot=NV, status=1, iters=1, q_next=NQS, working_set=2)

    def tick(self, inputs, dt, want_q_next=False, out=None, want_working_set=False):
        """One runWBC tick per instance up to the QP (+ integrate): returns dict(qdot, status, iters[, q_next][, working_set]).
        inputs["working_set"] (int64 [B,2], the previous tick's out["working_set"]) warm-starts the QP (include/wbc.h)."""
        keep = []
        q = inputs.get("q")
        B = self._batch_of(q)
        if out is None:
            out = dict(qdot=self._alloc(q, (B, NV)), status=self._alloc(q, (B,), np.int32), iters=self._alloc(q, (B,), np.int32))
            if want_q_next:
                out["q_next"] = self._alloc(q, (B, NQS))
            if want_working_set:
                out["working_set"] = self._alloc(q, (B, 2), np.int64)
        mem = _mem_of(list(inputs.values()) + list(out.values()))
        o = self._outs(out, self._TICK_OUT_WIDTH, keep, B, capi.WbcTickOut())
        tin = self._tick_in(inputs, keep, B)
        capi.check(self.lib.wbc_tick(self._h, B, C.byref(tin), float(dt), mem, C.byref(o), self._stream(mem)), self.lib)
        return out

    def make_tick_call(self, inputs, out, dt):
        """Bind device tensors once; the returned closure issues exactly one wbc_tick on the handle's current stream."""
        keep = []
        if _mem_of(list(inputs.values()) + list(out.values())) != capi.MEM_DEVICE:
            raise capi.WbcError("make_tick_call wants device tensors")
        B = self._batch_of(inputs.get("q"))
        o = self._outs(out, self._TICK_OUT_WIDTH, keep, B, capi.WbcTickOut())
        tin = self._tick_in(inputs, keep, B)
        lib, h, dtv, dev = self.lib, self._h, float(dt), self.device_id
        import torch

        def call():
            rc = lib.wbc_tick(h, B, C.byref(tin), dtv, capi.MEM_DEVICE, C.byref(o), torch.cuda.current_stream(dev).cuda_stream)
            if rc:
                capi.check(rc, lib)
        call._keep = keep
        return call

    def qp_solve(self, H, g, C_=None, lb=None, ub=None, Clb=None, Cub=None, working_set=None, want_working_set=False):
        """Batched QP.solveQP given H, g. H [B,n,n], C_ [B,p,n] (row-major rows), returns (x, status, iters).
        working_set [B,2] int64 (include/wbc.h wbc_qp_solve) hot-starts the solve; want_working_set appends the final one to the result."""
        keep = []
        mem = _mem_of([H, g, C_, lb, ub, Clb, Cub, working_set])
        if len(H.shape) != 3 or H.shape[1] != H.shape[2]:
            raise capi.WbcError("H must be [B, n, n], got %s" % (tuple(H.shape),))
        B, n = H.shape[0], H.shape[-1]
        if C_ is not None and (len(C_.shape) != 3 or C_.shape[2] != n):
            raise capi.WbcError("C must be [B, p, %d], got %s" % (n, tuple(C_.shape)))
        p = 0 if C_ is None else C_.shape[-2]
        x, st, it = self._alloc(H, (B, n)), self._alloc(H, (B,), np.int32), self._alloc(H, (B,), np.int32)
        wso = self._alloc(H, (B, 2), np.int64) if want_working_set else None
        f = np.float64
        P = self._p
        capi.check(self.lib.wbc_qp_solve(self._h, B, n, p, P(H, f, keep), P(g, f, keep, B, n, "g"), P(C_, f, keep, B, p * n, "C"),
                                          P(lb, f, keep, B, n, "lb"), P(ub, f, keep, B, n, "ub"), P(Clb, f, keep, B, p, "Clb"),
                                          P(Cub, f, keep, B, p, "Cub"), mem,
                                          P(x, f, keep), P(st, np.int32, keep), P(it, np.int32, keep),
                                          P(working_set, np.int64, keep, B, 2, "working_set"), P(wso, np.int64, keep), self._stream(mem)), self.lib)
        return (x, st, it, wso) if want_working_set else (x, st, it)

    def qp_solve_ls(self, A, b, C_=None, lb=None, ub=None, Clb=None, Cub=None, use_mfma=None, want_Hg=False, working_set=None,
                    want_working_set=False):
        """Batched QP(A, b, ...).solveQP(): H = A'A and g = -A'b are formed on the device (QP_Wrapper.py:17-18);
        use_mfma: True / False, None = matrix cores from WBC_MFMA_AUTO_ROWS rows of A on. working_set / want_working_set: as in
        qp_solve (solveQPHotstart); the final set comes last in the returned tuple."""
        keep = []
        mem = _mem_of([A, b, C_, lb, ub, Clb, Cub, working_set])
        if len(A.shape) != 3:
            raise capi.WbcError("A must be [B, m, n], got %s" % (tuple(A.shape),))
        B, m, n = A.shape
        if C_ is not None and (len(C_.shape) != 3 or C_.shape[2] != n):
            raise capi.WbcError("C must be [B, p, %d], got %s" % (n, tuple(C_.shape)))
        p = 0 if C_ is None else C_.shape[-2]
        x, st, it = self._alloc(A, (B, n)), self._alloc(A, (B,), np.int32), self._alloc(A, (B,), np.int32)
        Ho = self._alloc(A, (B, n, n)) if want_Hg else None
        go = self._alloc(A, (B, n)) if want_Hg else None
        wso = self._alloc(A, (B, 2), np.int64) if want_working_set else None
        f = np.float64
        P = self._p
        capi.check(self.lib.wbc_qp_solve_ls(self._h, B, m, n, p, P(A, f, keep), P(b, f, keep, B, m, "b"), P(C_, f, keep, B, p * n, "C"),
                                             P(lb, f, keep, B, n, "lb"), P(ub, f, keep, B, n, "ub"), P(Clb, f, keep, B, p, "Clb"),
                                             P(Cub, f, keep, B, p, "Cub"), mem,
                                             -1 if use_mfma is None else int(bool(use_mfma)), P(x, f, keep), P(st, np.int32, keep), P(it, np.int32, keep),
                                             P(Ho, f, keep), P(go, f, keep),
                                             P(working_set, np.int64, keep, B, 2, "working_set"), P(wso, np.int64, keep), self._stream(mem)), self.lib)
        out = (x, st, it, Ho, go) if want_Hg else (x, st, it)
        return out + (wso,) if want_working_set else out

    def posture_target(self, q, model_id=None, want_q_after=True):
        """qpJointb's "MANI" / "HYBRID" posture target u [B,26] under the configured mode (Robot_Wrapper4.py:1220-1260),
        and the configuration the reference's state is left at, q_after [B,27]."""
        keep = []
        mem = _mem_of([q, model_id])
        B = self._batch_of(q)
        u = self._alloc(q, (B, NV))
        qa = self._alloc(q, (B, NQS)) if want_q_after else None
        f = np.float64
        capi.check(self.lib.wbc_posture_target(self._h, B, self._p(q, f, keep), self._p(model_id, np.int32, keep, B, 1, "model_id"), mem,
                                                self._p(u, f, keep), self._p(qa, f, keep), self._stream(mem)), self.lib)
        return (u, qa) if want_q_after else u

    def update_state(self, q_cur, q_next, foot_targets, imu=None, model_id=None):
        """The tail of runWBC (Robot_Wrapper4.py:1397-1399): updateState(joint_config, base_config, running=True) with the
        foot-anchored base estimator trunkWorldPos (:1297-1327). Returns the new current_joint_config [B,27]."""
        keep = []
        mem = _mem_of([q_cur, q_next, foot_targets, imu, model_id])
        B = self._batch_of(q_cur, "q_cur")
        qn = self._alloc(q_cur, (B, NQS))
        f = np.float64
        P = self._p
        capi.check(self.lib.wbc_update_state(self._h, B, P(q_cur, f, keep), P(q_next, f, keep, B, NQS, "q_next"), P(imu, f, keep, B, 4, "imu"),
                                              P(foot_targets, f, keep, B, 15, "foot_targets"), P(model_id, np.int32, keep, B, 1, "model_id"), mem,
                                              P(qn, f, keep), self._stream(mem)), self.lib)
        return qn

    def rollout(self, inputs, dt, ticks, ee_target_step=None, trunk_target_step=None, imu=None, want_trace=True,
                mode=capi.ROLLOUT_RUNNING, hold_ticks=0):
        """K closed-loop ticks on the device: tick -> update_state -> reference-state side effects ->
        targets advance by their step; then `hold_ticks` more ticks with the targets held. mode: ROLLOUT_RUNNING
        (updateState(running=True): IMU fed back, base re-estimated from the stance feet) or ROLLOUT_WARMUP
        (updateState(running=False), the loop of setInitialState). Returns dict(q, qdot, ee_target, status, iters[,
        grip_trace [K + hold, B, 3]])."""
        keep = []
        extra = [ee_target_step, trunk_target_step, imu]
        mem = _mem_of(list(inputs.values()) + extra)
        q = inputs.get("q")
        B = self._batch_of(q)
        out = dict(q=self._alloc(q, (B, NQS)), qdot=self._alloc(q, (B, NV)), ee_target=self._alloc(q, (B, 5, 3)),
                   status=self._alloc(q, (B,), np.int32), iters=self._alloc(q, (B,), np.int32))
        if want_trace:
            out["grip_trace"] = self._alloc(q, (int(ticks) + int(hold_ticks), B, 3))
        r = capi.WbcRollout()
        r.ticks, r.mode, r.hold_ticks = int(ticks), int(mode), int(hold_ticks)
        f = np.float64
        P = self._p
        r.ee_target_step, r.trunk_target_step, r.imu = (P(ee_target_step, f, keep, B, 15, "ee_target_step"),
                                                        P(trunk_target_step, f, keep, B, 3, "trunk_target_step"), P(imu, f, keep, B, 4, "imu"))
        r.q_final, r.qdot_last, r.ee_target_final = P(out["q"], f, keep), P(out["qdot"], f, keep), P(out["ee_target"], f, keep)
        r.status_max, r.iters_sum = P(out["status"], np.int32, keep), P(out["iters"], np.int32, keep)
        if want_trace:
            r.grip_trace = P(out["grip_trace"], f, keep)
        tin = self._tick_in(inputs, keep, B)
        capi.check(self.lib.wbc_rollout(self._h, B, C.byref(tin), float(dt), C.byref(r), mem, self._stream(mem)), self.lib)
        return out

    def integrate(self, q, v, dt, model_id=None):
        """pin.integrate(model, q, v * dt) for every instance."""
        keep = []
        mem = _mem_of([q, v, model_id])
        B = self._batch_of(q)
        qn = self._alloc(q, (B, NQS))
        f = np.float64
        P = self._p
        capi.check(self.lib.wbc_integrate(self._h, B, P(q, f, keep), P(v, f, keep, B, NV, "v"), P(model_id, np.int32, keep, B, 1, "model_id"),
                                           float(dt), mem, P(qn, f, keep), self._stream(mem)), self.lib)
        return qn

    # ---- setInitialState for B robots
    def warm_up(self, q0, model_id=None, dt=0.002, ticks_per_segment=1000, foot_radius=0.0, configure=True):
        """``RobotModel.setInitialState`` (reference Robot_Wrapper4.py:196-351) for every instance of a batch: from q0 [B,27]
        (the reference starts from pin.neutral with the joints clamped to their upper limits, :199-208) the six Cartesian
        tasks + Tikhonov posture drag the feet and the gripper along straight lines to the crouched stance — 2 x
        ticks_per_segment bounds-only QPs per robot, ONE wbc_rollout call (mode WARMUP) for the whole batch — then the base
        orientation is reset and the trunk height set from the foot heights (:328-338).
        configure=True puts the warm-up's own switch set on the handle (setTasks(all True), no constraints, default weights: :272)
        and leaves it there. Host arrays in, dict(q [B,27], status, iters, start, goal) out."""
        import wbc_model
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        B = self._batch_of(q0, "q0")
        mid = None if model_id is None else np.ascontiguousarray(model_id, dtype=np.int32)
        if configure:
            for i, m in enumerate(self.models):
                self.configure(wbc_model.make_config(m, Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=True), i)
        f = self.fk(q0, mid, want=("oMf",))["oMf"]                        # updateState(q, feedback=False), :211
        pos, rot = f[:, :, 9:12], f[:, :, 0:9]
        ee, trunk = pos[:, capi.FR_EE0:capi.FR_EE0 + 5].copy(), pos[:, capi.FR_TRUNK].copy()
        Rt = rot[:, capi.FR_TRUNK].reshape(B, 3, 3)
        Ree = rot[:, capi.FR_EE0:capi.FR_EE0 + 5].reshape(B, 5, 3, 3)
        goal = ee.copy()                                                   # :238-262
        goal[:, :4, 0] = pos[:, capi.FR_HIP0:capi.FR_HIP0 + 4, 0]          # feet under their hips ...
        goal[:, :4, 2] *= 0.9                                              # ... and 10 % closer to the trunk (multiplier_F / _R)
        goal[:, 4, 2] = pos[:, capi.FR_ARM_BASE, 2]                        # gripper: height of oMi[arm_base_id] (G_base, :37, :253 — a constructor
        goal[:, 4, 0] = pos[:, capi.FR_HIP0, 0]                            # argument of its own, not the fifth hip_waist name), x of the FR hip
        goal[:, 4, 0] *= 1.1                                               # multiplier_G = diag(1.1, 1, 1.5)
        goal[:, 4, 2] *= 1.5
        n = int(ticks_per_segment)
        d = dict(q=q0, ee_target=ee, prev_ee_target=ee.copy(), trunk_target=trunk, prev_trunk_target=trunk.copy(),
                 ee_ref_rot=Ree.reshape(B, 5, 9).copy(),                   # R* = from_euler(as_euler(R_EE)) = R_EE (:222-226)
                 ee_prev_rot=np.einsum("bji,bejk->beik", Rt, Ree).reshape(B, 5, 9),     # prev_EE_CoM_rot = R_trunk' R_EE (:219)
                 trunk_ref_euler=np.stack([np.arctan2(Rt[:, 2, 1], Rt[:, 2, 2]), -np.arcsin(np.clip(Rt[:, 2, 0], -1, 1)),
                                           np.arctan2(Rt[:, 1, 0], Rt[:, 0, 0])], axis=1),
                 trunk_prev_rot=np.zeros((B, 9)))                          # old_ref_trunk_rot_matrix = zeros before initialiseWBC (:150)
        if mid is not None:
            d["model_id"] = mid
        warm = getattr(self, "_options", {}).get("warm_start", 0)
        self.set_option("warm_start", 0)                                   # the reference builds a fresh QP object every iteration (:320): cold
        try:
            ro = self.rollout(d, dt, n, ee_target_step=(goal - ee) / n, want_trace=False, mode=capi.ROLLOUT_WARMUP, hold_ticks=n)
        finally:
            self.set_option("warm_start", warm)
        q = ro["q"].copy()
        q[:, 3:6] = 0.0                                                    # "reset base orientation" (:328-330): x, y, z of the quaternion
        feet_z = self.fk(q, mid, want=("oMf",))["oMf"][:, capi.FR_EE0:capi.FR_EE0 + 4, 11]
        q[:, 2] = -feet_z.mean(axis=1) + foot_radius                       # :336-337
        return dict(q=q, status=ro["status"], iters=ro["iters"], start=ee, goal=goal)
