"""Synthetic tick inputs for the benchmark configurations of BASELINE.json.

The reference has no input generator (its inputs come from a PyBullet GUI simulation); this module draws
robot states from the distribution SURVEY.md §8(d) fixes so that tests, bench.py and the oracle all see the
same seeded inputs. It needs a forward-kinematics callable to place the targets on the robot
(``fk(q[B,27]) -> oMf[B, nframes, 12]``): tests pass the oracle's, bench.py passes the GPU's.
"""
import os

import numpy as np

import wbc_capi as capi

_MOCAP = os.path.join(capi.HERE, "data", "mocap_wx200_legs.csv")   # ships with the package: the product never reads tests/
_mocap_cache = None


def mocap_legs():
    """Rows of 12 leg angles in the log's FR, FL, RR, RL order (fixture sampled from the reference's
    tests_NOT_FOR_USE/mocap_wx200.txt)."""
    global _mocap_cache
    if _mocap_cache is None:
        _mocap_cache = np.loadtxt(_MOCAP, delimiter=",", comments="#")
    return _mocap_cache


def euler_xyz_to_quat(e):
    """(x, y, z, w) of Rz(c) Ry(b) Rx(a), vectorised over rows."""
    a, b, c = e[:, 0] / 2, e[:, 1] / 2, e[:, 2] / 2
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    return np.stack([sa * cb * cc - ca * sb * sc, ca * sb * cc + sa * cb * sc,
                     ca * cb * sc - sa * sb * cc, ca * cb * cc + sa * sb * sc], axis=1)


def R_to_euler_xyz(R):
    """extrinsic xyz angles of [B, 9] row-major rotations (scipy as_euler('xyz'))."""
    return np.stack([np.arctan2(R[:, 7], R[:, 8]), -np.arcsin(np.clip(R[:, 6], -1, 1)), np.arctan2(R[:, 3], R[:, 0])], axis=1)


def sample_q(model, B, rng):
    """Robot configurations: mocap gait legs + noise, arm in the middle 80 % of its range, small base tilt."""
    nq = model.nq
    q = np.zeros((B, capi.Q_STRIDE))
    q[:, 0:2] = rng.uniform(-0.05, 0.05, (B, 2))
    q[:, 2] = 0.30 + rng.uniform(-0.03, 0.03, B)
    q[:, 3:7] = euler_xyz_to_quat(rng.uniform(-0.1, 0.1, (B, 3)))
    legs = mocap_legs()[rng.integers(0, len(mocap_legs()), B)]
    legs = legs.reshape(B, 4, 3)[:, [1, 0, 3, 2], :].reshape(B, 12)      # FR,FL,RR,RL -> FL,FR,RL,RR
    legs = legs + rng.normal(0, 0.02, (B, 12))
    lo, hi = model.q_lo[7:19], model.q_hi[7:19]
    q[:, 7:19] = np.clip(legs, lo + 0.03, hi - 0.03)
    n_arm = nq - 19 - 3                                                   # waist .. last wrist joint
    lo, hi = model.q_lo[19:19 + n_arm], model.q_hi[19:19 + n_arm]
    mid, half = 0.5 * (lo + hi), 0.4 * (hi - lo)
    q[:, 19:19 + n_arm] = mid + half * rng.uniform(-1, 1, (B, n_arm))
    q[:, nq - 3] = 0.0                                                    # gripper
    q[:, nq - 2], q[:, nq - 1] = 0.02, -0.02                              # fingers, as in the mocap log
    return q


def make_tick_inputs(model, cfg, B, seed, fk, stress=True):
    """dict of numpy arrays keyed like WbcTickIn for `B` instances of `model` under `cfg`.

    stress=True applies the C3 recipe: 25 % of the instances get one arm DoF's damper inside its limit zone
    and 25 % get the trunk at the edge of (a few just outside) its z box, so bounds and box rows activate.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    q = sample_q(model, B, rng)
    if stress and cfg.use_bounds:
        pick = rng.random(B) < 0.25
        arm_dofs = np.arange(19, model.nv - 3)
        for b in np.nonzero(pick)[0]:
            i = int(rng.choice(arm_dofs))
            qi = cfg.damper_qidx[i]
            side = rng.random() < 0.5
            v = (cfg.damper_lo[i] + rng.uniform(0.0, 0.01)) if side else (cfg.damper_hi[i] - rng.uniform(0.0, 0.01))
            if model.q_lo[qi] <= v <= model.q_hi[qi]:
                q[b, qi] = v
    oMf = fk(q)
    pos = oMf[:, :, 9:12]
    ee = pos[:, capi.FR_EE0:capi.FR_EE0 + 5, :].copy()
    trunk = pos[:, capi.FR_TRUNK, :].copy()
    ee_target = ee.copy()
    ee_target[:, 4, :] += rng.normal(0, 0.01, (B, 3))
    d = dict(q=q, ee_target=ee_target,
             prev_ee_target=ee_target - rng.normal(0, 0.0005, (B, 5, 3)),
             trunk_target=trunk.copy(), prev_trunk_target=trunk - rng.normal(0, 0.0005, (B, 3)))
    eul = R_to_euler_xyz(oMf[:, capi.FR_TRUNK, 0:9])
    box = np.concatenate([trunk[:, 2:3], eul], axis=1)
    if stress and cfg.con_trunk:
        pick = rng.random(B) < 0.25
        frac = rng.uniform(0.245, 0.2505, B) * np.where(rng.random(B) < 0.5, 1.0, -1.0)
        # centre z0 such that the current z sits frac*z0 away from it: z = z0 (1 + frac)
        box[pick, 0] = trunk[pick, 2] / (1.0 + frac[pick])
    d["trunk_box_center"] = box
    d["trunk_ref_euler"] = eul.copy()
    Rt = oMf[:, capi.FR_TRUNK, 0:9]
    d["trunk_prev_rot"] = Rt.copy()           # steady state: R*_prev == R* 
    com = None
    if cfg.task_com:
        com = fk.com(q) if hasattr(fk, "com") else None
        if com is None:
            raise ValueError("CoM task needs fk.com(q)")
        d["com_target"] = com + rng.normal(0, 0.002, (B, 3))
        d["com_target_vel"] = rng.normal(0, 0.05, (B, 3))
    return d
