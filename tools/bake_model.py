#!/usr/bin/env python3
"""Bake a URDF into the flat kinematic-tree blob (JSON) the WBC kernels and the CPU oracle consume.

The reference builds its model with ``pin.buildModelFromUrdf(urdf_path, pin.JointModelFreeFlyer())``
(reference wrappers/Robot_Wrapper4.py:21).  Pinocchio is not installable here, so this tool restates
what that call produces for the two URDFs the benchmark uses:

* a free-flyer ``root_joint`` (jid 1, nq 7 / nv 6) is put above the URDF root link;
* the URDF tree is walked depth-first; the children of a link are visited in ASCII order of their
  *joint* name (urdfdom keeps joints in a ``std::map``), which is what makes the leg order FL, FR, RL, RR
  (confirmed by the reference's own dump tests_NOT_FOR_USE/Jacobians.py:18 "Joint 4 (FL_Calf_joint)");
* ``fixed`` joints create no model joint: the child link's inertia is lumped into the supporting
  joint's body and the joint/link survive as FIXED_JOINT / BODY frames;
* ``revolute``/``continuous`` about a coordinate axis -> RX/RY/RZ, ``prismatic`` -> PX/PY/PZ.

Only mass and centre of mass are kept from the inertials (the controller is velocity-level; inertia
tensors never enter).

Usage:  python tools/bake_model.py <urdf> <out.json>
This is a build-time tool (it reads the URDF text); nothing at run time needs the URDF.
"""
import json
import math
import sys
import xml.etree.ElementTree as ET

JT_UNIVERSE, JT_FF, JT_RX, JT_RY, JT_RZ, JT_PX, JT_PY, JT_PZ = range(8)
JT_NAMES = ["UNIVERSE", "FF", "RX", "RY", "RZ", "PX", "PY", "PZ"]


def _vec(s, n=3):
    v = [float(x) for x in s.split()]
    assert len(v) == n, s
    return v


def rpy_to_R(r, p, y):
    """URDF fixed-axis roll/pitch/yaw: R = Rz(y) Ry(p) Rx(r) (row-major 3x3)."""
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr]]


def matmul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(3)) for j in range(3)] for i in range(3)]


def matvec(A, v):
    return [sum(A[i][k] * v[k] for k in range(3)) for i in range(3)]


def se3_mul(M1, M2):
    R1, p1 = M1
    R2, p2 = M2
    Rp = matvec(R1, p2)
    return matmul(R1, R2), [p1[i] + Rp[i] for i in range(3)]


IDENT = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]


def origin_of(elem):
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return IDENT, [0.0, 0.0, 0.0]
    xyz = _vec(o.get("xyz", "0 0 0"))
    rpy = _vec(o.get("rpy", "0 0 0"))
    R = IDENT if rpy == [0.0, 0.0, 0.0] else rpy_to_R(*rpy)
    return R, xyz


def bake(urdf_path):
    root = ET.parse(urdf_path).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {j.get("name"): j for j in root.findall("joint")}
    children = {}   # parent link -> [joint names], ASCII-sorted
    child_links = set()
    for name in sorted(joints):
        j = joints[name]
        children.setdefault(j.find("parent").get("link"), []).append(name)
        child_links.add(j.find("child").get("link"))
    roots = [l for l in links if l not in child_links]
    assert len(roots) == 1, roots
    root_link = roots[0]

    mj = [dict(name="universe", type=JT_UNIVERSE, parent=0, idx_q=-1, idx_v=-1, R=IDENT, p=[0.0] * 3,
               q_lo=[], q_hi=[], v_max=[], bodies=[])]
    mj.append(dict(name="root_joint", type=JT_FF, parent=0, idx_q=0, idx_v=0, R=IDENT, p=[0.0] * 3,
                   q_lo=[-math.inf] * 7, q_hi=[math.inf] * 7, v_max=[math.inf] * 6, bodies=[]))
    frames = [dict(name="universe", type="FIXED_JOINT", parent_joint=0, R=IDENT, p=[0.0] * 3),
              dict(name="root_joint", type="JOINT", parent_joint=1, R=IDENT, p=[0.0] * 3)]
    nq, nv = 7, 6

    def add_body(jid, M, link_name):
        frames.append(dict(name=link_name, type="BODY", parent_joint=jid, R=M[0], p=M[1]))
        inertial = links[link_name].find("inertial")
        if inertial is None:
            return
        mass = float(inertial.find("mass").get("value"))
        _, c = se3_mul(M, origin_of(inertial))
        mj[jid]["bodies"].append((mass, c))

    def walk(link_name, jid, M):
        """link_name is attached to model joint jid with placement M (R, p) in that joint's frame."""
        nonlocal nq, nv
        add_body(jid, M, link_name)
        for jn in children.get(link_name, []):
            j = joints[jn]
            jt = j.get("type")
            Mj = se3_mul(M, origin_of(j))
            child = j.find("child").get("link")
            if jt == "fixed":
                frames.append(dict(name=jn, type="FIXED_JOINT", parent_joint=jid, R=Mj[0], p=Mj[1]))
                walk(child, jid, Mj)
                continue
            axis = _vec(j.find("axis").get("xyz")) if j.find("axis") is not None else [1.0, 0.0, 0.0]
            k = [abs(a) for a in axis].index(1.0)
            assert axis[k] == 1.0 and sum(abs(a) for a in axis) == 1.0, "only +x/+y/+z joint axes: %s" % jn
            if jt in ("revolute", "continuous"):
                typ = (JT_RX, JT_RY, JT_RZ)[k]
            elif jt == "prismatic":
                typ = (JT_PX, JT_PY, JT_PZ)[k]
            else:
                raise ValueError("unsupported joint type %s (%s)" % (jt, jn))
            lim = j.find("limit")
            lo = float(lim.get("lower", "0")) if lim is not None else 0.0
            hi = float(lim.get("upper", "0")) if lim is not None else 0.0
            vmax = float(lim.get("velocity", "0")) if lim is not None else 0.0
            if jt == "continuous":
                lo, hi = -math.inf, math.inf
            new = len(mj)
            mj.append(dict(name=jn, type=typ, parent=jid, idx_q=nq, idx_v=nv, R=Mj[0], p=Mj[1],
                           q_lo=[lo], q_hi=[hi], v_max=[vmax], bodies=[]))
            nq += 1
            nv += 1
            frames.append(dict(name=jn, type="JOINT", parent_joint=new, R=IDENT, p=[0.0] * 3))
            walk(child, new, (IDENT, [0.0] * 3))

    walk(root_link, 1, (IDENT, [0.0] * 3))

    q_lo, q_hi, v_max = [], [], []
    out_joints = []
    for j in mj:
        m = sum(b[0] for b in j["bodies"])
        c = [sum(b[0] * b[1][i] for b in j["bodies"]) / m if m > 0 else 0.0 for i in range(3)]
        q_lo += j["q_lo"]
        q_hi += j["q_hi"]
        v_max += j["v_max"]
        out_joints.append(dict(name=j["name"], type=JT_NAMES[j["type"]], type_id=j["type"], parent=j["parent"],
                               idx_q=j["idx_q"], idx_v=j["idx_v"], placement_R=j["R"], placement_p=j["p"],
                               mass=m, com=c))

    def enc(v):
        return [("inf" if x == math.inf else "-inf" if x == -math.inf else x) for x in v]

    # collision spheres (the reference reads the foot radius from the collision geometry, Robot_Wrapper4.py:55-58)
    spheres = {}
    for lname, l in links.items():
        for col in l.findall("collision"):
            sph = col.find("geometry/sphere")
            if sph is not None:
                spheres[lname] = float(sph.get("radius"))

    return dict(name=root.get("name"), source=urdf_path.split("/Robot_Descriptions/")[-1], collision_spheres=spheres,
                nq=nq, nv=nv, njoints=len(out_joints), joints=out_joints, frames=frames,
                q_lo=enc(q_lo), q_hi=enc(q_hi), v_max=enc(v_max),
                total_mass=sum(j["mass"] for j in out_joints))


if __name__ == "__main__":
    blob = bake(sys.argv[1])
    with open(sys.argv[2], "w") as f:
        json.dump(blob, f, indent=1)
    print("%s: nq=%d nv=%d njoints=%d nframes=%d mass=%.6f" % (
        blob["name"], blob["nq"], blob["nv"], blob["njoints"], len(blob["frames"]), blob["total_mass"]))
