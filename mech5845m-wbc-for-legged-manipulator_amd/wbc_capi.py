"""ctypes mirror of include/wbc.h and loader of the HIP shared library.

This directory is a flat module directory, like the reference's ``wrappers/`` (its modules import each
other as ``from QP_Wrapper import QP``, reference wrappers/Robot_Wrapper4.py:6): put it on ``sys.path``
in place of ``wrappers/`` and ``QP_Wrapper`` / ``Robot_Wrapper4`` resolve to the MI355X versions.

There is no CPU fallback: if ``csrc/build/libwbc_hip.so`` is missing or does not load, importing the
product classes fails loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WBC_HIP_LIB") or os.path.join(HERE, "csrc", "build", "libwbc_hip.so")   # WBC_HIP_LIB: e.g. the profile build

# ---- limits (include/wbc.h)
MAX_JOINTS, MAX_NQ, MAX_NV, NEE, MAX_FRAMES, MAX_P, MAX_M, MAX_MODELS = 24, 28, 26, 5, 16, 24, 96, 4
Q_STRIDE, V_STRIDE = 27, 26
JT = dict(UNIVERSE=0, FF=1, RX=2, RY=3, RZ=4, PX=5, PY=6, PZ=7)
FR_EE0, FR_TRUNK, FR_HIP0, FR_ARM_BASE, FR_NROLES = 0, 5, 6, 11, 12
MEM_HOST, MEM_DEVICE = 0, 1
QP_OPTIMAL, QP_MAX_ITER, QP_INFEASIBLE, QP_NUMERICAL = 0, 1, 2, 3
JOINT_OFF, JOINT_TIKHONOV, JOINT_PREV, JOINT_MANI, JOINT_HYBRID, JOINT_CUSTOM = 0, 1, 2, 3, 4, 5

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class WbcModelBlob(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("nv", C.c_int32), ("njoints", C.c_int32),
        ("jtype", C.c_int32 * MAX_JOINTS), ("parent", C.c_int32 * MAX_JOINTS),
        ("idx_q", C.c_int32 * MAX_JOINTS), ("idx_v", C.c_int32 * MAX_JOINTS),
        ("place_R", (C.c_double * 9) * MAX_JOINTS), ("place_p", (C.c_double * 3) * MAX_JOINTS),
        ("mass", C.c_double * MAX_JOINTS), ("com", (C.c_double * 3) * MAX_JOINTS),
        ("q_lo", C.c_double * MAX_NQ), ("q_hi", C.c_double * MAX_NQ), ("v_max", C.c_double * MAX_NV),
        ("nframes", C.c_int32), ("frame_joint", C.c_int32 * MAX_FRAMES),
        ("frame_R", (C.c_double * 9) * MAX_FRAMES), ("frame_p", (C.c_double * 3) * MAX_FRAMES),
        ("ee_joint", C.c_int32 * NEE),
    ]


class WbcConfig(C.Structure):
    _fields_ = [
        ("task_ee", C.c_int32 * NEE), ("task_trunk", C.c_int32), ("task_com", C.c_int32), ("task_joint", C.c_int32),
        ("con_com", C.c_int32), ("con_trunk", C.c_int32), ("con_ee", C.c_int32 * NEE),
        ("use_bounds", C.c_int32), ("lock_from", C.c_int32),
        ("arm_base_id", C.c_int32), ("posture_literal", C.c_int32),
        ("damper_qidx", C.c_int32 * MAX_NV),
        ("damper_lo", C.c_double * MAX_NV), ("damper_hi", C.c_double * MAX_NV), ("damper_vmax", C.c_double * MAX_NV),
        ("damper_coef", C.c_double), ("damper_qi", C.c_double), ("damper_qs", C.c_double),
        ("ee_W", (C.c_double * 6) * NEE), ("ee_w", C.c_double * NEE), ("ee_gain", (C.c_double * 6) * NEE),
        ("trunk_W", C.c_double * 6), ("trunk_w", C.c_double), ("trunk_gain", C.c_double * 6),
        ("com_W", C.c_double * 3), ("com_gain", C.c_double * 3),
        ("joint_w", C.c_double),
        ("trunk_box_z_frac", C.c_double), ("trunk_box_ang", C.c_double), ("trunk_box_scale", C.c_double),
        ("com_box_scale", C.c_double),
    ]


class WbcTickIn(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "q", "ee_target", "prev_ee_target", "trunk_target", "prev_trunk_target", "trunk_box_center",
        "ee_ref_rot", "ee_prev_rot", "trunk_ref_euler", "trunk_prev_rot", "com_target", "com_target_vel", "model_id",
        "posture_u", "q_con", "working_set")]


class WbcQpData(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("A", "b", "H", "g", "C", "Clb", "Cub", "lb", "ub")]


class WbcTickOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qdot", "status", "iters", "q_next", "working_set")]


ROLLOUT_RUNNING, ROLLOUT_WARMUP = 0, 1


class WbcRollout(C.Structure):
    _fields_ = [("ticks", C.c_int32), ("mode", C.c_int32)] + [(n, C.c_void_p) for n in (
        "ee_target_step", "trunk_target_step", "imu", "q_final", "qdot_last", "ee_target_final", "grip_trace",
        "status_max", "iters_sum")] + [("hold_ticks", C.c_int32), ("pad_", C.c_int32)]


class WbcFkOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("oMi", "oMf", "J", "com", "Jcom")]


# every symbol include/wbc.h declares, with its ctypes signature
_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
SIGNATURES = {
    "wbc_model_create": (_i, [C.POINTER(WbcModelBlob), C.POINTER(_vp)]),
    "wbc_model_destroy": (None, [_vp]),
    "wbc_batch_create": (_i, [C.POINTER(_vp), _i, _i, _i, C.POINTER(_vp)]),
    "wbc_batch_destroy": (None, [_vp]),
    "wbc_batch_configure": (_i, [_vp, _i, C.POINTER(WbcConfig)]),
    "wbc_task_rows": (_i, [_vp]),
    "wbc_constraint_rows": (_i, [_vp]),
    "wbc_fk_jacobians": (_i, [_vp, _i, _vp, _vp, _i, C.POINTER(WbcFkOut), _vp]),
    "wbc_assemble": (_i, [_vp, _i, C.POINTER(WbcTickIn), _d, _i, C.POINTER(WbcQpData), _vp]),
    "wbc_qp_solve": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wbc_qp_solve_ls": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wbc_posture_target": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "wbc_tick": (_i, [_vp, _i, C.POINTER(WbcTickIn), _d, _i, C.POINTER(WbcTickOut), _vp]),
    "wbc_update_state": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "wbc_rollout": (_i, [_vp, _i, C.POINTER(WbcTickIn), _d, C.POINTER(WbcRollout), _i, _vp]),
    "wbc_integrate": (_i, [_vp, _i, _vp, _vp, _vp, _d, _i, _vp, _vp]),
    "wbc_batch_set_option": (_i, [_vp, C.c_char_p, _i]),
    "wbc_batch_get_stat": (_i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64)]),
    "wbc_batch_synchronize": (_i, [_vp, _vp]),
    "wbc_debug_cycles": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "wbc_last_error": (C.c_char_p, []),
    "wbc_version": (C.c_char_p, []),
    "wbc_abi_sizes": (_i, [c_int32_p, c_int32_p]),
}

_lib = None


class WbcError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen the HIP library and bind every entry point; raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise WbcError("HIP extension not built: %s is missing (run `python __graft_entry__.py build`)" % p)
    # PyTorch ships its own copy of the HIP runtime. If this library (linked against /opt/rocm's) is loaded first and torch
    # afterwards, the process ends up with two runtimes and the second one to initialise sees no device. Loading torch
    # first makes both share one runtime; without torch installed there is only ours.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    sb, sc = C.c_int32(), C.c_int32()
    lib.wbc_abi_sizes(C.byref(sb), C.byref(sc))
    if sb.value != C.sizeof(WbcModelBlob) or sc.value != C.sizeof(WbcConfig):
        raise WbcError("ABI mismatch: library (%d, %d) vs ctypes (%d, %d)" % (
            sb.value, sc.value, C.sizeof(WbcModelBlob), C.sizeof(WbcConfig)))
    if path is None:
        _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_library()
        raise WbcError("wbc call failed (%d): %s" % (rc, (lib.wbc_last_error() or b"").decode()))
