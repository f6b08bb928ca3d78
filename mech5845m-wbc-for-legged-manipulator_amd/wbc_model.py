"""Model blobs and batch-uniform settings for the WBC hot path.

``load_model`` replaces ``pin.buildModelFromUrdf(urdf, pin.JointModelFreeFlyer())`` + the frame/joint id
look-ups of ``RobotModel.__init__`` (reference wrappers/Robot_Wrapper4.py:21-52) with a baked kinematic
tree (models/*.json, produced at build time by tools/bake_model.py from the reference's URDFs).
``make_config`` gathers the attributes ``__init__`` / ``setTasks`` / ``setConstraints`` /
``staticReachMode`` set (Robot_Wrapper4.py:72-125, 176-193, 1415-1464) into the C-ABI ``WbcConfig``.
"""
import json
import math
import os

import numpy as np

import wbc_capi as capi

MODELS_DIR = os.path.join(capi.HERE, "models")

# the names sim3.py passes (reference wrappers/sim3.py:56-60)
A1_ROLES = dict(
    EE_frame_names=["FR_foot_fixed", "FL_foot_fixed", "RR_foot_fixed", "RL_foot_fixed", "gripper_bar"],
    EE_joint_names=["FR_calf_joint", "FL_calf_joint", "RR_calf_joint", "RL_calf_joint", "gripper"],
    hip_waist_joint_names=["FR_hip_joint", "FL_hip_joint", "RR_hip_joint", "RL_hip_joint", "waist"],
    imu="imu_joint", G_base="waist")


def _num(x):
    return math.inf if x == "inf" else -math.inf if x == "-inf" else float(x)


class Model:
    """A baked model: the JSON (names -> ids) and the flat C blob."""

    def __init__(self, data, roles):
        self.data = data
        self.roles = roles
        self.name = data["name"]
        self.nq, self.nv, self.njoints = data["nq"], data["nv"], data["njoints"]
        self.joint_names = [j["name"] for j in data["joints"]]
        self.frames = data["frames"]
        if self.njoints > capi.MAX_JOINTS or self.nv > capi.MAX_NV or self.nq > capi.Q_STRIDE:
            raise capi.WbcError("model %s exceeds compiled limits" % self.name)
        b = capi.WbcModelBlob()
        b.nq, b.nv, b.njoints = self.nq, self.nv, self.njoints
        for i, j in enumerate(data["joints"]):
            b.jtype[i], b.parent[i] = j["type_id"], j["parent"]
            b.idx_q[i], b.idx_v[i] = j["idx_q"], j["idx_v"]
            for r in range(3):
                for c in range(3):
                    b.place_R[i][3 * r + c] = j["placement_R"][r][c]
                b.place_p[i][r] = j["placement_p"][r]
                b.com[i][r] = j["com"][r]
            b.mass[i] = j["mass"]
        self.q_lo = np.array([_num(x) for x in data["q_lo"]])
        self.q_hi = np.array([_num(x) for x in data["q_hi"]])
        self.v_max = np.array([_num(x) for x in data["v_max"]])
        for i in range(self.nq):
            b.q_lo[i], b.q_hi[i] = self.q_lo[i], self.q_hi[i]
        for i in range(self.nv):
            b.v_max[i] = self.v_max[i]
        # controller frames, in WBC_FR_* role order
        names = list(roles["EE_frame_names"]) + [roles["imu"]] + list(roles["hip_waist_joint_names"]) + [roles["G_base"]]
        kinds = ["FIXED_JOINT"] * 6 + ["JOINT"] * 6     # getFrameId(name, type): Robot_Wrapper4.py:30,38,47,51
        self.role_frames = [self.frame_id(n, k) for n, k in zip(names, kinds)]
        b.nframes = len(self.role_frames)
        for i, fid in enumerate(self.role_frames):
            f = self.frames[fid]
            b.frame_joint[i] = f["parent_joint"]
            for r in range(3):
                for c in range(3):
                    b.frame_R[i][3 * r + c] = f["R"][r][c]
                b.frame_p[i][r] = f["p"][r]
        self.ee_joint = [self.joint_id(n) for n in roles["EE_joint_names"]]
        for i in range(capi.NEE):
            b.ee_joint[i] = self.ee_joint[i]
        self.blob = b

    def frame_id(self, name, kind=None):
        for i, f in enumerate(self.frames):
            if f["name"] == name and (kind is None or f["type"] == kind):
                return i
        raise KeyError("frame %r (%s) not in model %s" % (name, kind, self.name))

    def joint_id(self, name):
        return self.joint_names.index(name)

    def neutral(self):
        """pin.neutral(model): zeros with identity quaternion (x, y, z, w) = (0, 0, 0, 1)."""
        q = np.zeros(capi.Q_STRIDE)
        q[6] = 1.0
        return q


def load_model(name_or_path="a1_wx200", roles=None):
    path = name_or_path if os.path.exists(name_or_path) else os.path.join(MODELS_DIR, name_or_path + ".json")
    with open(path) as f:
        data = json.load(f)
    return Model(data, dict(A1_ROLES if roles is None else roles))


def model_for_urdf(urdf_path, roles=None):
    """Map the reference's ``urdf_path`` constructor argument (sim3.py:48-51) to a baked blob by file stem."""
    stem = os.path.splitext(os.path.basename(urdf_path))[0]
    if not os.path.exists(os.path.join(MODELS_DIR, stem + ".json")):
        raise capi.WbcError("no baked model for %s: run tools/bake_model.py %s %s/%s.json" % (
            urdf_path, urdf_path, MODELS_DIR, stem))
    return load_model(stem, roles)


def damper_tables(model, compat=True):
    """Index map and limits of velDamperJointConstraints (Robot_Wrapper4.py:583-598).

    compat=True reproduces the reference literally: DoF i compares ``current_joint_config[i]``
    (nq-indexed, so the quaternion w sits at 6 and every actuated joint is tested against its predecessor's
    angle) against limits from which index 6 was deleted, and ``vel_lim[i] = 5`` is applied for i < 7 on the
    nv-sized array (so FL_hip's 52.4 rad/s becomes 5).  compat=False is the intended map.
    """
    nq, nv = model.nq, model.nv
    lo, hi, vm = model.q_lo.copy(), model.q_hi.copy(), model.v_max.copy()
    lock_q = model.ee_joint[4] - 2 + 7
    for i in range(nq):
        if i < 7:
            lo[i], hi[i] = -5.0, 5.0
            if compat or i < 6:
                vm[i] = 5.0
        if i >= lock_q:
            lo[i], hi[i] = 0.0, 0.0
    lo, hi = np.delete(lo, 6), np.delete(hi, 6)
    qidx = np.arange(nv) if compat else np.array([i if i < 6 else i + 1 for i in range(nv)])
    return qidx, lo, hi, vm, model.ee_joint[4] - 2 + 6


def make_config(model, Trunk=False, FR=False, FL=False, RR=False, RL=False, Grip=False, Joint=False,
                cCoM=False, cTrunk=False, cFR=False, cFL=False, cRR=False, cRL=False, cGrip=False,
                task_com=False, mode="default", use_bounds=True, damper_compat=True, posture_literal=True):
    """WbcConfig for the given task / constraint switches.

    mode "default" = the weights of ``RobotModel.__init__`` (Robot_Wrapper4.py:72-125);
    mode "static_reach" = ``staticReachMode()`` (Robot_Wrapper4.py:1415-1464).
    Joint: False / True (Tikhonov) / "PREV" / "MANI" / "HYBRID" as in ``setTasks`` (Robot_Wrapper4.py:176-183,
    1209-1260), or "CUSTOM" (u given per instance). posture_literal: MANI/HYBRID to the letter of the reference
    (index quirks, accumulating perturbations, perturbed state leaking into the rest of the tick — SURVEY.md C.4)
    or, if False, the intended central difference.
    """
    c = capi.WbcConfig()
    for i, on in enumerate((FR, FL, RR, RL, Grip)):
        c.task_ee[i] = int(bool(on))
    c.task_trunk, c.task_com = int(bool(Trunk)), int(bool(task_com))
    if Joint is True:
        c.task_joint = capi.JOINT_TIKHONOV
    elif Joint == "PREV":
        c.task_joint = capi.JOINT_PREV
    elif Joint == "MANI":
        c.task_joint = capi.JOINT_MANI
    elif Joint == "HYBRID":
        c.task_joint = capi.JOINT_HYBRID
    elif Joint == "CUSTOM":
        c.task_joint = capi.JOINT_CUSTOM
    elif Joint is False or Joint is None:
        c.task_joint = capi.JOINT_OFF
    else:
        raise ValueError("unknown posture mode %r" % (Joint,))
    c.arm_base_id = model.joint_id(model.roles["G_base"])     # getJointId(G_base), Robot_Wrapper4.py:37
    c.posture_literal = int(bool(posture_literal))
    c.con_com, c.con_trunk = int(bool(cCoM)), int(bool(cTrunk))
    for i, on in enumerate((cFR, cFL, cRR, cRL, cGrip)):
        c.con_ee[i] = int(bool(on))
    c.use_bounds = int(bool(use_bounds))
    qidx, lo, hi, vm, lock_from = damper_tables(model, damper_compat)
    c.lock_from = lock_from
    for i in range(capi.MAX_NV):
        if i < model.nv:
            c.damper_qidx[i], c.damper_lo[i], c.damper_hi[i], c.damper_vmax[i] = int(qidx[i]), lo[i], hi[i], vm[i]
    c.damper_coef, c.damper_qi, c.damper_qs = 0.01, 0.026, 0.015
    static = mode == "static_reach"
    if mode not in ("default", "static_reach"):
        raise ValueError(mode)
    for i in range(capi.NEE):
        for r in range(6):
            c.ee_W[i][r] = 1.0
            c.ee_gain[i][r] = (0.05 if i == 4 else 0.8) if static else 0.5
        c.ee_w[i] = (1.0 if i == 4 else 100.0) if static else 1.0
    for r in range(6):
        c.trunk_W[r] = 1.0
        c.trunk_gain[r] = 0.8 if static else 0.5
    c.trunk_w = 1.0
    for r in range(3):
        c.com_W[r], c.com_gain[r] = 1.0, 1.0        # Robot_Wrapper2.py:71, 96
    c.joint_w = 0.001 if static else 0.05
    c.trunk_box_z_frac, c.trunk_box_ang, c.trunk_box_scale, c.com_box_scale = 0.25, 0.15, 0.5, 0.8
    return c


def sim3_config(model, damper_compat=True, Joint="PREV", posture_literal=True):
    """The switch set of the reference's sim3.py tick (sim3.py:145-148 + staticReachMode): tasks {Grip, Joint},
    constraints {Trunk, FR, FL, RR, RL}. sim3.py itself sets Joint="HYBRID"; "PREV" is the benchmark default
    (BASELINE configs[2] / SURVEY.md §8d C3)."""
    return make_config(model, Grip=True, Joint=Joint, cTrunk=True, cFR=True, cFL=True, cRR=True, cRL=True,
                       mode="static_reach", damper_compat=damper_compat, posture_literal=posture_literal)


def equality_only_config(model):
    """BASELINE config 2: 5 EE tasks + CoM task (Robot_Wrapper2) + Tikhonov posture, 4-foot contact equalities,
    no box bounds."""
    return make_config(model, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=True, task_com=True,
                       cFR=True, cFL=True, cRR=True, cRL=True, use_bounds=False)
