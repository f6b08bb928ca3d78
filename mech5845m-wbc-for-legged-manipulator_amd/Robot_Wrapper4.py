"""MI355X mirror of the reference's ``wrappers/Robot_Wrapper4.py`` (class ``RobotModel``): one robot instance (B = 1)
of the batched HIP pipeline behind the reference's constructor, methods and attributes, so that a driver written for
the reference (``wrappers/sim3.py``) finds the same surface:

    RobotModel(urdf_path, mesh_dir_path, EE_frame_names, EE_joint_names, G_base, imu, FR_hip_joint,
               hip_waist_joint_names, foot_offset=False)                       (reference Robot_Wrapper4.py:19)
    setTasks / setConstraints / staticReachMode / initialiseWBC / updateState / runWBC            (:176, :186, :1415, :354, :387, :1330)
    qpA / qpb / findConstraints / velDamperJointConstraints / endEffectorA2 / trunkA / EEConstraint /
    trunkConstraint / CoMConstraint / jointVelocitiestoConfig / trunkWorldPos                      (:1271 ... :1297)
    attributes read by the sims: current_joint_config, FR_leg/FL_leg/RR_leg/RL_leg/grip, prev_EE_pos,
    robot_data.oMf[i].translation/.rotation, robot_data.oMi[j], robot_data.com[0], trunk_frame_index,
    end_effector_index_list_frame, n_velocity_dimensions, ...

Every number comes from the device through the C-ABI (wbc_batch.WbcBatch -> include/wbc.h); pinocchio, qpOASES and
klampt are not used. Differences from the reference, all deliberate and documented in DESIGN.md:
  * ``dt`` is the fixed ``step_time`` (the reference busy-waits and measures wall-clock) unless
    ``pace_realtime = True`` is set on the object;
  * the per-tick debug prints of calcTargetVelEE3 (Robot_Wrapper4.py:1075-1085) are not emitted;
  * posture modes "MANI"/"HYBRID" run on the device to the letter of the reference, including the
    perturbed configuration qpJointb leaves in current_joint_config / robot_data; ``posture_literal = False`` on the
    object switches to the clean central difference;
  * ``solver_status`` / ``solver_iters`` expose what the reference throws away.
"""
import time

import numpy as np

import wbc_capi as capi
import wbc_model
from QP_Wrapper import QP  # noqa: F401  (the reference module imports it too: Robot_Wrapper4.py:6)
from wbc_batch import WbcBatch


# ------------------------------------------------------------------ small shims for robot_model / robot_data
class _SE3:
    __slots__ = ("rotation", "translation")

    def __init__(self, M12):
        self.rotation = np.array(M12[:9]).reshape(3, 3)
        self.translation = np.array(M12[9:12])


class _Placements:
    """robot_data.oMf / robot_data.oMi: index -> object with .translation and .rotation."""

    def __init__(self, getter, n):
        self._get, self._n = getter, n

    def __getitem__(self, i):
        return self._get(int(i))

    def __len__(self):
        return self._n


class _Data:
    def __init__(self, owner):
        self._o = owner
        self.oMi = _Placements(lambda j: _SE3(owner._oMi[j]), owner._model.njoints)
        self.oMf = _Placements(owner._frame_placement, len(owner._model.frames))
        self.com = [np.zeros(3)]


class _ModelView:
    def __init__(self, model):
        self.nq, self.nv = model.nq, model.nv
        self.names = list(model.joint_names)
        self.lowerPositionLimit = model.q_lo.copy()
        self.upperPositionLimit = model.q_hi.copy()
        self.velocityLimit = model.v_max.copy()
        self._m = model

    def getFrameId(self, name, kind=None):
        return self._m.frame_id(name)

    def getJointId(self, name):
        return self._m.joint_id(name)


def _euler_xyz_from_R(R):
    """scipy Rotation.from_matrix(R).as_euler('xyz') for a proper rotation (extrinsic x-y-z)."""
    return np.array([np.arctan2(R[2, 1], R[2, 2]), -np.arcsin(np.clip(R[2, 0], -1.0, 1.0)), np.arctan2(R[1, 0], R[0, 0])])


def _R_from_euler_xyz(e):
    a, b, c = np.asarray(e, dtype=float).reshape(3)
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    return np.array([[cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa],
                     [sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa],
                     [-sb, cb * sa, cb * ca]])


def _diag6(M, what):
    M = np.asarray(M, dtype=float)
    if M.shape != (6, 6) or np.abs(M - np.diag(np.diag(M))).max() != 0.0:
        raise NotImplementedError("%s must be a diagonal 6x6 matrix (every preset of the reference is)" % what)
    return np.diag(M)


class _LinearTrajectory:
    """klampt ``trajectory.Trajectory(milestones=...)``: piecewise-linear, unit time per segment, clamped ends
    (what Robot_Wrapper4.py:264-283 and sim3.py:221-295 use)."""

    def __init__(self, milestones):
        self.m = [np.asarray(x, dtype=float) for x in milestones]

    def eval(self, t):
        if t <= 0:
            return self.m[0].tolist()
        if t >= len(self.m) - 1:
            return self.m[-1].tolist()
        i = int(np.floor(t))
        u = t - i
        return ((1 - u) * self.m[i] + u * self.m[i + 1]).tolist()


class RobotModel:
    def __init__(self, urdf_path, mesh_dir_path, EE_frame_names, EE_joint_names, G_base, imu, FR_hip_joint,
                 hip_waist_joint_names, foot_offset=False, device_id=0, warm_up=True):
        roles = dict(EE_frame_names=list(EE_frame_names), EE_joint_names=list(EE_joint_names),
                     hip_waist_joint_names=list(hip_waist_joint_names), imu=imu, G_base=G_base)
        self._model = wbc_model.model_for_urdf(urdf_path, roles)
        self._bt = WbcBatch(self._model, 1, device_id=device_id)
        self.robot_model = _ModelView(self._model)
        self.joint_names = self.robot_model.names
        self.foot_radius = 0
        self._oMi = np.zeros((self._model.njoints, 12))
        self.robot_data = _Data(self)
        self.J = np.zeros((6, self._model.nv))

        # frame / joint ids (reference :28-52)
        self.trunk_frame_index = self._model.frame_id(imu, "FIXED_JOINT")
        self.current_joint_config = 0
        self.EE_frame_names, self.EE_joint_names = list(EE_frame_names), list(EE_joint_names)
        self.hip_waist_joint_names = list(hip_waist_joint_names)
        self.arm_base_id = self._model.joint_id(G_base)
        self.arm_base_frame_id = self._model.frame_id(G_base, "JOINT")
        self.FR_hip_joint = self._model.joint_id(FR_hip_joint)
        self.n_velocity_dimensions = self._model.nv
        self.n_configuration_dimensions = self._model.nq
        self.n_of_EE = 5
        self.end_effector_index_list_frame = [self._model.frame_id(n, "FIXED_JOINT") for n in EE_frame_names]
        self.end_effector_index_list_joint = [self._model.joint_id(n) for n in EE_joint_names]
        self.hip_waist_joint_index_list_frame = [self._model.frame_id(n, "JOINT") for n in hip_waist_joint_names]
        if foot_offset:                                   # reference :55-58 reads the foot collision sphere
            link = EE_frame_names[0].replace("_fixed", "")
            self.foot_radius = self._model.data["collision_spheres"].get(link, 0.0)

        self.initialised = False
        self.EE_frame_pos = [0, 0, 0, 0, 0]
        self.trunk_frame_pos = 0
        self.default_trunk_ori = np.zeros((3, 1))
        self.default_EE_ori_list = [np.zeros((3, 1)) for _ in range(5)]
        self.updateState(self._model.neutral()[:self._model.nq], feedback=False)
        arm_base = self.robot_data.oMf[self.arm_base_frame_id].translation
        grip = self.robot_data.oMf[self.end_effector_index_list_frame[4]].translation
        self.arm_reach = np.sum(grip - arm_base)

        self._set_default_weights()
        self.task_active_Trunk = self.task_active_FR_foot = self.task_active_FL_foot = False
        self.task_active_RR_foot = self.task_active_RL_foot = self.task_active_GRIP = False
        self.task_active_Joint = False
        self.const_active_CoM = self.const_active_Trunk = self.const_active_FR_foot = False
        self.const_active_FL_foot = self.const_active_RR_foot = self.const_active_RL_foot = self.const_active_GRIP = False

        self.previous_time = 0
        self.step_time = 0.002
        self.dt = 0.002
        self.pace_realtime = False
        self.damper_compat = True          # reproduce the reference's index quirk
        self.posture_literal = True        # MANI/HYBRID to the letter, state leak included

        self.prev_trunk_ref = np.array([0, 0, 0])
        self.old_ref_trunk_rot_matrix = np.zeros((3, 3))
        self.prev_EE_pos = [0, 0, 0, 0, 0]
        self.prev_EE_CoM_rot = [0, 0, 0, 0, 0]
        self.EE_A_list, self.EE_b_list = [0] * 5, [0] * 5
        self.firstQP = True
        self.hotstart = True               # runWBC's QPs after the first are warm-started, as the reference's are (:1389-1394)
        self._working_set = None
        self.qp = None
        self.solver_status, self.solver_iters = None, None
        self.q_vel = None
        self.FL_base_pos = np.copy(self.robot_data.oMf[self.end_effector_index_list_frame[1]].translation)
        self.print_ = False
        self.end_effector_A = self.end_effector_B = self.trunk_A = self.trunk_B = 0
        self.initial_trunk_pos = np.zeros(3)
        self.initial_trunk_ori_euler = np.zeros((3, 1))

        if warm_up:
            self.setInitialState()
        self.initialised = True
        self.FL_base_pos = np.copy(self.robot_data.oMf[self.end_effector_index_list_frame[1]].translation)
        self.trunk_base_pos = np.copy(self.robot_data.oMf[self.trunk_frame_index].translation)
        self.prev_trunk_pos = np.copy(self.trunk_base_pos)
        self._log_previous_state()

    # ------------------------------------------------------------------ settings
    def _set_default_weights(self):
        """reference :72-125"""
        self.trunk_weight = np.identity(6) * 1
        self.FR_weight = self.FL_weight = self.RR_weight = self.RL_weight = np.identity(6) * 1
        self.grip_weight = np.identity(6) * 1
        self.EE_weight = [self.FR_weight, self.FL_weight, self.RR_weight, self.RL_weight, self.grip_weight]
        self.cart_task_weight_FR = self.cart_task_weight_FL = self.cart_task_weight_RR = self.cart_task_weight_RL = 1
        self.cart_task_weight_GRIP = 1
        self.cart_task_weight_Trunk = 1
        self.cart_task_weight_EE_list = [1, 1, 1, 1, 1]
        self.joint_task_weight = 0.05
        self.trunk_gain = np.identity(6) * 0.5
        self.FL_gain = self.FR_gain = self.RL_gain = self.RR_gain = np.identity(6) * 0.5
        self.GRIP_gain = np.identity(6) * 0.5
        self.EE_gains = [self.FL_gain, self.FR_gain, self.RL_gain, self.RR_gain, self.GRIP_gain]
        self.com_weight, self.com_gain = np.identity(3), np.identity(3)     # Robot_Wrapper2.py:71, 96

    def staticReachMode(self):
        """reference :1415-1464"""
        self.trunk_weight = np.identity(6) * 1
        self.FR_weight = self.FL_weight = self.RR_weight = self.RL_weight = np.identity(6) * 1
        self.grip_weight = np.identity(6) * 1
        self.EE_weight = [self.FR_weight, self.FL_weight, self.RR_weight, self.RL_weight, self.grip_weight]
        self.cart_task_weight_FR = self.cart_task_weight_FL = self.cart_task_weight_RR = self.cart_task_weight_RL = 100
        self.cart_task_weight_GRIP = 1
        self.cart_task_weight_Trunk = 1
        self.cart_task_weight_EE_list = [100, 100, 100, 100, 1]
        self.joint_task_weight = 0.001
        self.trunk_gain = np.identity(6) * 0.8
        self.FL_gain = self.FR_gain = self.RL_gain = self.RR_gain = np.identity(6) * 0.8
        self.GRIP_gain = np.identity(6) * 0.05
        self.EE_gains = [self.FL_gain, self.FR_gain, self.RL_gain, self.RR_gain, self.GRIP_gain]

    def setTasks(self, Trunk=False, FR=False, FL=False, RR=False, RL=False, Grip=False, Joint=False):
        self.task_active_Trunk, self.task_active_FR_foot, self.task_active_FL_foot = Trunk, FR, FL
        self.task_active_RR_foot, self.task_active_RL_foot, self.task_active_GRIP = RR, RL, Grip
        self.task_active_Joint = Joint

    def setConstraints(self, CoM=False, Trunk=False, FR=False, FL=False, RR=False, RL=False, Grip=False):
        self.const_active_CoM, self.const_active_Trunk = CoM, Trunk
        self.const_active_FR_foot, self.const_active_FL_foot = FR, FL
        self.const_active_RR_foot, self.const_active_RL_foot, self.const_active_GRIP = RR, RL, Grip

    def _config(self, tasks=None, cons=None, joint="same"):
        """WbcConfig from the current attributes; `tasks`/`cons` override the switches (for the single-block accessors)."""
        t = tasks if tasks is not None else dict(Trunk=self.task_active_Trunk, FR=self.task_active_FR_foot,
                                                 FL=self.task_active_FL_foot, RR=self.task_active_RR_foot,
                                                 RL=self.task_active_RL_foot, Grip=self.task_active_GRIP)
        c = cons if cons is not None else dict(cCoM=self.const_active_CoM, cTrunk=self.const_active_Trunk,
                                               cFR=self.const_active_FR_foot, cFL=self.const_active_FL_foot,
                                               cRR=self.const_active_RR_foot, cRL=self.const_active_RL_foot,
                                               cGrip=self.const_active_GRIP)
        j = self.task_active_Joint if joint == "same" else joint
        if j is False or j is None:
            j = True     # the device needs H > 0; callers that asked for no posture rows get them sliced off again
        cfg = wbc_model.make_config(self._model, Joint=j, damper_compat=self.damper_compat,
                                    posture_literal=self.posture_literal, **t, **c)
        for i in range(5):
            W = _diag6(self.EE_weight[i], "EE_weight[%d]" % i)
            G = _diag6(self.EE_gains[i], "EE_gains[%d]" % i)
            for r in range(6):
                cfg.ee_W[i][r], cfg.ee_gain[i][r] = W[r], G[r]
            cfg.ee_w[i] = float(self.cart_task_weight_EE_list[i])
        W, G = _diag6(self.trunk_weight, "trunk_weight"), _diag6(self.trunk_gain, "trunk_gain")
        for r in range(6):
            cfg.trunk_W[r], cfg.trunk_gain[r] = W[r], G[r]
        cfg.trunk_w = float(self.cart_task_weight_Trunk)
        cfg.joint_w = float(self.joint_task_weight)
        return cfg

    # ------------------------------------------------------------------ state
    def _frame_placement(self, fid):
        f = self._model.frames[fid]
        Mj = self._oMi[f["parent_joint"]]
        R = Mj[:9].reshape(3, 3) @ np.array(f["R"])
        p = Mj[9:] + Mj[:9].reshape(3, 3) @ np.array(f["p"])
        return _SE3(np.concatenate([R.reshape(9), p]))

    def _fk(self, config):
        q = np.zeros((1, capi.Q_STRIDE))
        q[0, :self._model.nq] = config
        out = self._bt.fk(q)
        self._oMi = out["oMi"][0]
        self.J = out["J"][0][:, :self._model.nv]
        self.robot_data.com[0] = out["com"][0]
        self._Jcom = out["Jcom"][0][:, :self._model.nv]
        self.trunk_frame_pos = np.copy(self.robot_data.oMf[self.trunk_frame_index].translation)
        for i in range(5):
            self.EE_frame_pos[i] = np.copy(self.robot_data.oMf[self.end_effector_index_list_frame[i]].translation)

    def updateState(self, joint_config, imu_data=0, feedback=True, running=False):
        """reference :387-428"""
        if feedback and running:
            config = np.concatenate((self.current_joint_config[:3], imu_data, joint_config), axis=0)
        else:
            config = np.asarray(joint_config, dtype=float)
        self.previous_joint_config = self.current_joint_config
        self.current_joint_config = config
        self._fk(config)
        if running:
            base_pos = self.trunkWorldPos()
            config = np.concatenate((base_pos, self.current_joint_config[3:]), axis=0)
            self.previous_joint_config = self.current_joint_config
            self.current_joint_config = config
            self._fk(config)

    def jointVelocitiestoConfig(self, joint_vel, update_model=False):
        """reference :440-449: pin.integrate(q, qdot * dt)"""
        q = np.zeros((1, capi.Q_STRIDE))
        q[0, :self._model.nq] = self.current_joint_config
        v = np.zeros((1, capi.V_STRIDE))
        v[0, :self._model.nv] = joint_vel
        new_config = self._bt.integrate(q, v, self.dt)[0, :self._model.nq]
        if update_model:
            self.updateState(new_config, feedback=False, running=bool(self.initialised))
            return None
        return new_config

    def trunkWorldPos(self):
        """reference :1297-1327: base position from the (assumed static) foot targets"""
        WRB = self.robot_data.oMf[self.trunk_frame_index].rotation
        trunk = self.robot_data.oMf[self.trunk_frame_index].translation
        targets = [self.FR_target_cartesian_pos, self.FL_target_cartesian_pos, self.RR_target_cartesian_pos, self.RL_target_cartesian_pos]
        WPA = sum(np.asarray(t, dtype=float).reshape(3) for t in targets) / 4
        BPA = sum(self.robot_data.oMf[self.end_effector_index_list_frame[i]].translation - trunk for i in range(4)) / 4
        return WPA - WRB @ BPA

    def _log_previous_state(self):
        """reference :167-173 / :369-376"""
        self.prev_trunk_ref = np.copy(self.robot_data.oMf[self.trunk_frame_index].translation)
        Rt = self.robot_data.oMf[self.trunk_frame_index].rotation
        for i in range(5):
            M = self.robot_data.oMf[self.end_effector_index_list_frame[i]]
            self.prev_EE_pos[i] = np.copy(M.translation)
            self.prev_EE_CoM_rot[i] = Rt.T @ M.rotation

    def _capture_default_orientations(self):
        """reference :222-226 / :363-367"""
        self.default_trunk_ori = _euler_xyz_from_R(self.robot_data.oMf[self.trunk_frame_index].rotation).reshape(3, 1)
        for i in range(5):
            R = self.robot_data.oMf[self.end_effector_index_list_frame[i]].rotation
            self.default_EE_ori_list[i] = _euler_xyz_from_R(R).reshape(3, 1)

    def initialiseWBC(self, imu_data):
        """reference :354-383"""
        self.updateState(self.current_joint_config, imu_data, running=False)
        self._capture_default_orientations()
        self._log_previous_state()
        self.old_ref_trunk_rot_matrix = np.copy(self.robot_data.oMf[self.trunk_frame_index].rotation)
        self.initial_trunk_pos = np.copy(self.robot_data.oMf[self.trunk_frame_index].translation)
        self.initial_trunk_ori = np.copy(self.robot_data.oMf[self.trunk_frame_index].rotation)
        self.initial_trunk_ori_euler = _euler_xyz_from_R(self.initial_trunk_ori).reshape(3, 1)

    # ------------------------------------------------------------------ device inputs for the current state
    def _tick_inputs(self, target_EE, target_trunk):
        nq = self._model.nq
        d = {}
        q = np.zeros((1, capi.Q_STRIDE))
        q[0, :nq] = self.current_joint_config
        d["q"] = q
        d["ee_target"] = np.array([np.asarray(t, dtype=float).reshape(3) for t in target_EE]).reshape(1, 5, 3)
        d["prev_ee_target"] = np.array([np.asarray(t, dtype=float).reshape(3) for t in self.prev_EE_pos]).reshape(1, 5, 3)
        tt = np.asarray(target_trunk if target_trunk is not None else self.trunk_frame_pos, dtype=float).reshape(1, 3)
        d["trunk_target"] = tt
        d["prev_trunk_target"] = np.asarray(self.prev_trunk_ref, dtype=float).reshape(1, 3)
        d["trunk_box_center"] = np.concatenate([[self.initial_trunk_pos[2]], np.asarray(self.initial_trunk_ori_euler).reshape(3)]).reshape(1, 4)
        d["ee_ref_rot"] = np.array([_R_from_euler_xyz(e) for e in self.default_EE_ori_list]).reshape(1, 5, 9)
        d["ee_prev_rot"] = np.array([np.asarray(R, dtype=float) for R in self.prev_EE_CoM_rot]).reshape(1, 5, 9)
        d["trunk_ref_euler"] = np.asarray(self.default_trunk_ori, dtype=float).reshape(1, 3)
        d["trunk_prev_rot"] = np.asarray(self.old_ref_trunk_rot_matrix, dtype=float).reshape(1, 9)
        return d

    def _advance_reference_state(self, target_EE, target_trunk):
        """the side effects of qpb(): calcTargetVelEE3 :1151-1152 and calcTargetVelTrunk2 :995-996"""
        active = [self.task_active_FR_foot, self.task_active_FL_foot, self.task_active_RR_foot, self.task_active_RL_foot, self.task_active_GRIP]
        for i in range(5):
            if active[i]:
                self.prev_EE_pos[i] = target_EE[i]
                self.prev_EE_CoM_rot[i] = _R_from_euler_xyz(self.default_EE_ori_list[i])
        if self.task_active_Trunk:
            self.prev_trunk_ref = target_trunk
            self.old_ref_trunk_rot_matrix = _R_from_euler_xyz(self.default_trunk_ori)

    def _assemble(self, cfg, target_EE=None, target_trunk=None, want=("A", "b", "C", "Clb", "Cub", "lb", "ub")):
        self._bt.configure(cfg)
        tEE = target_EE if target_EE is not None else self.prev_EE_pos
        out = self._bt.assemble(self._tick_inputs(tEE, target_trunk), self.dt, want=want)
        return {k: v[0] for k, v in out.items()}

    # ------------------------------------------------------------------ accessors (same returns as the reference)
    def qpA(self):
        """reference :1271-1280: (6 n_tasks [+ nv]) x nv"""
        nv = self._model.nv
        a = self._assemble(self._config(joint=True), want=("A",))["A"]   # qpJointA does not depend on the posture mode
        m_cart = a.shape[0] - capi.V_STRIDE          # the device always carries the (padded) posture block
        return a[:m_cart + (nv if self._has_posture() else 0), :nv]

    def qpb(self, target_cartesian_pos_EE, target_cartesian_pos_trunk):
        """reference :1283-1294 (column vector); advances prev_EE_pos / prev_EE_CoM_rot / prev_trunk_ref like the reference"""
        nv = self._model.nv
        b = self._assemble(self._config(), target_cartesian_pos_EE, target_cartesian_pos_trunk, want=("b",))["b"]
        m_cart = b.shape[0] - capi.V_STRIDE
        b = b[:m_cart + nv] if self._has_posture() else b[:m_cart]
        if self.task_active_Joint in ("MANI", "HYBRID") and self.posture_literal:
            # qpJointb leaves the model at the last perturbed configuration (:1231-1236, :1252-1257; updateState aliases
            # current_joint_config to the array being perturbed): everything called after qpb sees that state
            q = np.zeros((1, capi.Q_STRIDE))
            q[0, :self._model.nq] = self.current_joint_config
            self._bt.configure(self._config())
            _, q_after = self._bt.posture_target(q)
            self.previous_joint_config = self.current_joint_config
            self.current_joint_config = q_after[0, :self._model.nq].copy()
            self._fk(self.current_joint_config)
        self._advance_reference_state(target_cartesian_pos_EE, target_cartesian_pos_trunk)
        return b.reshape(-1, 1)

    def _has_posture(self):
        return self.task_active_Joint is True or self.task_active_Joint in ("PREV", "MANI", "HYBRID")

    def findConstraints(self):
        """reference :764-836: returns (C.T, Clb, Cub) with C.T of shape (nv, p)"""
        o = self._assemble(self._config(joint=True), want=("C", "Clb", "Cub"))   # independent of the posture mode
        return o["C"][:, :self._model.nv].T, o["Clb"], o["Cub"]

    def velDamperJointConstraints(self):
        """reference :572-637"""
        o = self._assemble(self._config(joint=True), want=("lb", "ub"))
        return o["lb"][:self._model.nv], o["ub"][:self._model.nv]

    def _one_task(self, **which):
        base = dict(Trunk=False, FR=False, FL=False, RR=False, RL=False, Grip=False)
        base.update(which)
        none = dict(cCoM=False, cTrunk=False, cFR=False, cFL=False, cRR=False, cRL=False, cGrip=False)
        return self._assemble(self._config(tasks=base, cons=none, joint=True), want=("A",))["A"][:6, :self._model.nv]

    def _one_constraint(self, rows, **which):
        base = dict(cCoM=False, cTrunk=False, cFR=False, cFL=False, cRR=False, cRL=False, cGrip=False)
        base.update(which)
        none = dict(Trunk=False, FR=False, FL=False, RR=False, RL=False, Grip=False)
        o = self._assemble(self._config(tasks=none, cons=base, joint=True), want=("C", "Clb", "Cub"))
        return o["C"][:rows, :self._model.nv], o["Clb"][:rows], o["Cub"][:rows]

    def endEffectorA2(self, frame_index):
        """reference :474-484 (stores into EE_A_list)"""
        key = ("FR", "FL", "RR", "RL", "Grip")[frame_index]
        self.EE_A_list[frame_index] = self._one_task(**{key: True})

    def trunkA(self):
        """reference :487-490 (stores into trunk_A)"""
        self.trunk_A = self._one_task(Trunk=True)

    def EEConstraint(self, frame_index):
        """reference :757-761"""
        key = ("cFR", "cFL", "cRR", "cRL", "cGrip")[frame_index]
        return self._one_constraint(3, **{key: True})

    def trunkConstraint(self):
        """reference :707-754"""
        return self._one_constraint(4, cTrunk=True)

    def CoMConstraint(self):
        """reference :669-694"""
        return self._one_constraint(2, cCoM=True)

    # ------------------------------------------------------------------ the tick
    def _pace(self):
        if self.pace_realtime:                           # reference :1338-1342 (busy-wait, dt = measured)
            while (time.time() - self.previous_time) < self.step_time:
                self.dt = time.time() - self.previous_time
            self.previous_time = time.time()
        else:
            self.dt = self.step_time

    def _solve_tick(self, cfg, target_EE, target_trunk):
        self._bt.configure(cfg)
        inp = self._tick_inputs(target_EE, target_trunk)
        if self.hotstart and not self.firstQP and self._working_set is not None:
            inp["working_set"] = self._working_set        # solveQPHotstart (reference :1392-1394): the previous tick's working set seeds this one
        out = self._bt.tick(inp, self.dt, want_q_next=True, want_working_set=True)
        self._working_set = out["working_set"]
        self.solver_status, self.solver_iters = int(out["status"][0]), int(out["iters"][0])
        if self.solver_status != 0 and getattr(self, "q_vel", None) is not None:
            # an unsolved QP leaves qpOASES' xOpt at the previous tick's answer (QP_Wrapper.py:71-73): the reference keeps
            # moving with the stale velocity; the device returns 0, so the stale vector is integrated here
            q = np.zeros((1, capi.Q_STRIDE))
            q[0, :self._model.nq] = self.current_joint_config
            v = np.zeros((1, capi.V_STRIDE))
            v[0, :self._model.nv] = self.q_vel
            return np.array(self.q_vel), self._bt.integrate(q, v, self.dt)[0, :self._model.nq]
        return out["qdot"][0, :self._model.nv], out["q_next"][0, :self._model.nq]

    def runWBC(self, base_config, target_cartesian_pos_EE=None, target_cartesian_pos_trunk=None):
        """reference :1330-1412: one control tick; returns (FL_leg, FR_leg, RL_leg, RR_leg, grip)"""
        self.FR_target_cartesian_pos, self.FL_target_cartesian_pos = target_cartesian_pos_EE[0], target_cartesian_pos_EE[1]
        self.RR_target_cartesian_pos, self.RL_target_cartesian_pos = target_cartesian_pos_EE[2], target_cartesian_pos_EE[3]
        self._pace()
        q_vel, q_next = self._solve_tick(self._config(), target_cartesian_pos_EE, target_cartesian_pos_trunk)
        self._advance_reference_state(target_cartesian_pos_EE, target_cartesian_pos_trunk)
        self.firstQP = False
        self.q_vel = q_vel
        joint_config = q_next[7:]
        self.updateState(joint_config, np.asarray(base_config, dtype=float), running=True)
        return joint_config[0:3], joint_config[3:6], joint_config[6:9], joint_config[9:12], joint_config[12:]

    # ------------------------------------------------------------------ warm-up (reference :196-351)
    def setInitialState(self, iterations_per_segment=1000):
        """Drag the neutral pose to the crouched stance with the bounds-only QP (all six Cartesian tasks + Tikhonov
        posture) along the straight-line foot / gripper trajectories, then fix the base: the B = 1 call of
        ``WbcBatch.warm_up`` — the 2 x 1000 QPs run as ONE wbc_rollout (mode WBC_ROLLOUT_WARMUP) on the device. dt is the fixed
        step_time here (the reference measures wall-clock inside its busy-wait).
        Differences by design: ``solver_status`` is the WORST status over the 2000 QPs (the reference keeps none: qpOASES' return code
        is dropped, QP_Wrapper.py:49), ``solver_iters`` their sum; the last ``q_vel`` is a local of the reference's loop (:321) and is
        not kept here either. The gripper goal's height comes from frame role WBC_FR_ARM_BASE = oMi[arm_base_id] (G_base, :37, :253)."""
        nq = self._model.nq
        q0 = self._model.neutral()[:nq]
        for i in range(self.n_velocity_dimensions):          # reference :201-208: only the upper clamp has an effect
            if q0[i] > self.robot_model.upperPositionLimit[i]:   # (this is what bends the knees: calf upper limit < 0)
                q0[i] = self.robot_model.upperPositionLimit[i]
        self.updateState(q0, feedback=False)
        self._log_previous_state()
        self._capture_default_orientations()
        Trunk_target_pos = np.copy(self.trunk_frame_pos)
        self.setTasks(Trunk=True, FR=True, FL=True, RR=True, RL=True, Grip=True, Joint=True)
        saved = (self.const_active_CoM, self.const_active_Trunk, self.const_active_FR_foot, self.const_active_FL_foot,
                 self.const_active_RR_foot, self.const_active_RL_foot, self.const_active_GRIP)
        self.setConstraints()
        self._bt.configure(self._config())                   # the object's own weights / gains (defaults at construction, :72-125)
        q = np.zeros((1, capi.Q_STRIDE))
        q[0, :nq] = q0
        self.dt = self.step_time
        out = self._bt.warm_up(q, None, self.step_time, iterations_per_segment, foot_radius=self.foot_radius, configure=False)
        self.setConstraints(*saved)
        self.solver_status, self.solver_iters = int(out["status"][0]), int(out["iters"][0])
        # the attribute state the reference's loop leaves behind: targets at their last milestone, reference state advanced
        goal = out["goal"][0]
        target = [goal[i].reshape(3, 1) for i in range(5)]
        self.FR_target_cartesian_pos, self.FL_target_cartesian_pos = target[0], target[1]
        self.RR_target_cartesian_pos, self.RL_target_cartesian_pos = target[2], target[3]
        self._advance_reference_state(target, Trunk_target_pos)
        self.updateState(out["q"][0, :nq].copy(), feedback=False)      # base orientation reset + trunk height fix included (:328-338)
        joints = self.current_joint_config[7:]
        self.FL_leg, self.FR_leg, self.RL_leg, self.RR_leg, self.grip = joints[0:3], joints[3:6], joints[6:9], joints[9:12], joints[12:]
        self.fristQP = False                                  # (sic) reference :347
        self.dt = self.step_time
        print("Initial state set successfully")

    def Rot2Euler(self, Rot):
        """reference :1495-1499"""
        import math
        roll = math.atan2(Rot[2, 1], Rot[2, 2])
        pitch = math.atan2(-Rot[2, 0], math.sqrt(Rot[2, 1] ** 2 + Rot[2, 2] ** 2))
        yaw = math.atan2(Rot[1, 0], Rot[0, 0])
        return np.array([[roll, pitch, yaw]]).T
