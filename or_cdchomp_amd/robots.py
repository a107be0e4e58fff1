"""Robot descriptions for the build's own kinematic model.

OpenRAVE (the reference's FK / Jacobian provider, /root/reference
src/orcdchomp_mod.cpp:1016-1048) is third party and absent, so the build defines
its own kinematic tree:

    link frame = parent link frame o pose_parent_joint o motion(axis, q[dof])

Links are stored in topological order.  Spheres use the reference's XML
vocabulary (``<sphere link= pos= radius=/>``, src/orcdchomp_kdata.cpp:79-89).
The WAM sphere table is the data of scripts/barrettwam_withspheres.robot.xml:24-45
(data, not code); the link frames are a WAM-like arm of our own because the
OpenRAVE model files (robots/wam7.kinbody.xml) are not part of the reference repo.
"""
import math
import re

import numpy as np

JOINT_FIXED, JOINT_REVOLUTE, JOINT_PRISMATIC = 0, 1, 2


def quat_from_axis_angle(axis, angle):
    """[qx qy qz qw] (libcd pose convention, src/libcd/kin.c:42-52)."""
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / np.linalg.norm(ax)
    s = math.sin(0.5 * angle)
    return [ax[0] * s, ax[1] * s, ax[2] * s, math.cos(0.5 * angle)]


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz]


class RobotModel:
    def __init__(self, name):
        self.name = name
        self.link_names = []
        self.adjacent = []          # (link, link): pairs the robot description declares adjacent (<adjacent> tags)
        self.parent = []
        self.pose_parent_joint = []
        self.joint_type = []
        self.axis = []
        self.dof_index = []
        self.n_dof = 0
        self.limit_lower = []
        self.limit_upper = []
        self.spheres = []          # (link name, [x y z], radius), XML order
        self.manipulators = []     # (name, end-effector link name, tool pose [x y z qx qy qz qw]); the first is the active one

    def add_link(self, name, parent=None, xyz=(0, 0, 0), quat=(0, 0, 0, 1),
                 joint=JOINT_FIXED, axis=(0, 0, 1), limits=None, dof=None):
        """dof=None allocates a new robot dof for a moving joint."""
        pidx = -1 if parent is None else self.link_names.index(parent)
        self.link_names.append(name)
        self.parent.append(pidx)
        self.pose_parent_joint.append([float(v) for v in xyz] + [float(v) for v in quat])
        self.joint_type.append(joint)
        a = np.asarray(axis, dtype=np.float64)
        self.axis.append(list(a / np.linalg.norm(a)))
        if joint == JOINT_FIXED:
            self.dof_index.append(-1)
        else:
            if dof is None:
                dof = self.n_dof
                self.n_dof += 1
                lo, hi = limits if limits is not None else (-math.inf, math.inf)
                self.limit_lower.append(float(lo))
                self.limit_upper.append(float(hi))
            self.dof_index.append(dof)
        return len(self.link_names) - 1

    def add_sphere(self, link, pos, radius):
        if link not in self.link_names:
            raise ValueError("link %s in <orcdchomp> does not exist." % link)
        self.spheres.append((link, [float(v) for v in pos], float(radius)))

    def add_spheres_xml(self, xml_text):
        """Accepts the body of an <orcdchomp><spheres> block."""
        for m in re.finditer(r"<sphere\s+([^>]*?)/?>", xml_text):
            attrs = dict(re.findall(r'(\w+)\s*=\s*"([^"]*)"', m.group(1)))
            self.add_sphere(attrs["link"], [float(v) for v in attrs["pos"].split()],
                            float(attrs["radius"]))

    def link_frames(self, base_pose, dof_values):
        """world frames (R [n_links][3][3], t [n_links][3]) of all links: link frame = parent link frame o
        pose_parent_joint o motion(axis, q[dof]); base_pose and pose_parent_joint are x y z qx qy qz qw"""
        def rot(q):
            x, y, z, w = q
            return np.array([[1 - 2*(y*y + z*z), 2*(x*y - z*w), 2*(x*z + y*w)],
                             [2*(x*y + z*w), 1 - 2*(x*x + z*z), 2*(y*z - x*w)],
                             [2*(x*z - y*w), 2*(y*z + x*w), 1 - 2*(x*x + y*y)]])
        Rb, tb = rot(base_pose[3:7]), np.asarray(base_pose[:3], dtype=np.float64)
        R, t = [], []
        for li in range(len(self.link_names)):
            Rp, tp = (Rb, tb) if self.parent[li] < 0 else (R[self.parent[li]], t[self.parent[li]])
            pj = self.pose_parent_joint[li]
            Rj, tj = Rp @ rot(pj[3:7]), Rp @ np.asarray(pj[:3]) + tp
            if self.joint_type[li] == JOINT_REVOLUTE:
                a = np.asarray(self.axis[li]); q = float(dof_values[self.dof_index[li]])
                Rj = Rj @ rot(list(a * math.sin(0.5 * q)) + [math.cos(0.5 * q)])
            elif self.joint_type[li] != JOINT_FIXED:
                tj = tj + float(dof_values[self.dof_index[li]]) * (Rj @ np.asarray(self.axis[li]))
            R.append(Rj); t.append(tj)
        return np.array(R), np.array(t)

    # flat arrays for the C structs -------------------------------------------------
    def arrays(self):
        return dict(
            n_links=len(self.link_names),
            parent=np.asarray(self.parent, dtype=np.int32),
            pose_parent_joint=np.asarray(self.pose_parent_joint, dtype=np.float64).reshape(-1, 7),
            joint_type=np.asarray(self.joint_type, dtype=np.int32),
            axis=np.asarray(self.axis, dtype=np.float64).reshape(-1, 3),
            dof_index=np.asarray(self.dof_index, dtype=np.int32),
            n_dof=self.n_dof,
            limit_lower=np.asarray(self.limit_lower, dtype=np.float64),
            limit_upper=np.asarray(self.limit_upper, dtype=np.float64),
            n_spheres=len(self.spheres),
            sphere_link=np.asarray([self.link_names.index(s[0]) for s in self.spheres], dtype=np.int32),
            sphere_pos=np.asarray([s[1] for s in self.spheres], dtype=np.float64).reshape(-1, 3),
            sphere_radius=np.asarray([s[2] for s in self.spheres], dtype=np.float64),
            n_adjacent=len(self.adjacent),
            adjacent=np.asarray([[self.link_names.index(a), self.link_names.index(b)] for a, b in self.adjacent],
                                dtype=np.int32).reshape(-1, 2),
        )


# sphere table: data of /root/reference scripts/barrettwam_withspheres.robot.xml:24-45
WAM_SPHERES_XML = """
<sphere link="wam0"      pos=" 0.22  0.14 0.346" radius="0.15" />
<sphere link="wam2"      pos=" 0.0   0.0  0.2 " radius="0.06" />
<sphere link="wam2"      pos=" 0.0   0.0  0.3 " radius="0.06" />
<sphere link="wam2"      pos=" 0.0   0.0  0.4 " radius="0.06" />
<sphere link="wam2"      pos=" 0.0   0.0  0.5 " radius="0.06" />
<sphere link="wam3"      pos="0.0  0.0  0.0" radius="0.06" />
<sphere link="wam4"      pos="0.0 0.0  0.2 " radius="0.06" />
<sphere link="wam4"      pos="0.0 0.0  0.1 " radius="0.06" />
<sphere link="wam4"      pos="0.0 0.0  0.3 " radius="0.06" />
<sphere link="wam6"      pos=" 0.0   0.0  0.1 " radius="0.06" />
<sphere link="Finger0-1" pos=" 0.05  -0.01 0.0 " radius="0.04" />
<sphere link="Finger1-1" pos=" 0.05  -0.01 0.0 " radius="0.04" />
<sphere link="Finger2-1" pos=" 0.05  -0.01 0.0 " radius="0.04" />
<sphere link="Finger0-2" pos=" 0.05   0.0  0.0 " radius="0.04" />
<sphere link="Finger1-2" pos=" 0.05   0.0  0.0 " radius="0.04" />
<sphere link="Finger2-2" pos=" 0.05   0.0  0.0 " radius="0.04" />
"""

# start configuration of the demo, /root/reference scripts/test_wam7.py:66
WAM_START = [2.5, -1.8, 0.0, 2.0, 0.0, 0.2, 0.0]
# robot base pose of the demo (test_wam7.py:40: quaternion w,x,y,z = .70711,0,.70711,0;
# translation -1,0,1) in libcd order [x y z qx qy qz qw]
WAM_BASE_POSE = [-1.0, 0.0, 1.0, 0.0, 0.70711, 0.0, 0.70711]
# synthetic goal standing in for the ikfast solution
WAM_GOAL = [0.6, -1.2, 0.3, 1.6, -0.4, 0.5, 0.2]


def wam7():
    """WAM-like 7-dof arm + 3-finger hand (11 dofs; the arm is dofs 0..6).

    All link frames are aligned with the base at the zero configuration, z along
    the arm, joint axes alternate z / y (the layout the sphere table assumes:
    upper-arm spheres at z = 0.2..0.5 of wam2, elbow sphere at the wam3 origin).
    """
    r = RobotModel("BarrettWAM")
    R, F = JOINT_REVOLUTE, JOINT_FIXED
    r.add_link("wam0")
    r.add_link("wam1", "wam0", (0.22, 0.14, 0.346), joint=R, axis=(0, 0, 1), limits=(-2.6, 2.6))
    r.add_link("wam2", "wam1", (0, 0, 0), joint=R, axis=(0, 1, 0), limits=(-2.0, 2.0))
    r.add_link("wam3", "wam2", (0.045, 0, 0.55), joint=R, axis=(0, 0, 1), limits=(-2.8, 2.8))
    r.add_link("wam4", "wam3", (-0.045, 0, 0), joint=R, axis=(0, 1, 0), limits=(-0.9, 3.1))
    r.add_link("wam5", "wam4", (0, 0, 0.3), joint=R, axis=(0, 0, 1), limits=(-4.76, 1.24))
    r.add_link("wam6", "wam5", (0, 0, 0), joint=R, axis=(0, 1, 0), limits=(-1.6, 1.6))
    r.add_link("wam7", "wam6", (0, 0, 0.06), joint=R, axis=(0, 0, 1), limits=(-3.0, 3.0))
    r.add_link("handbase", "wam7", (0, 0, 0.0), joint=F)
    # fingers: local x points along the finger (up the hand at zero config)
    up = quat_from_axis_angle((0, 1, 0), -0.5 * math.pi)
    for k, phi in enumerate((math.radians(60.0), math.radians(-60.0), math.radians(180.0))):
        q = quat_mul(quat_from_axis_angle((0, 0, 1), phi), up)
        base = (0.04 * math.cos(phi), 0.04 * math.sin(phi), 0.10)
        r.add_link("Finger%d-1" % k, "handbase", base, q, joint=R, axis=(0, 1, 0), limits=(0.0, 2.44))
        r.add_link("Finger%d-2" % k, "Finger%d-1" % k, (0.07, 0, 0), joint=F)
    # spread joint exists as a dof (JF4) but moves no sphere link in this model
    r.n_dof += 1
    r.limit_lower.append(0.0)
    r.limit_upper.append(math.pi)
    r.add_spheres_xml(WAM_SPHERES_XML)
    # links a self-collision check never tests against each other, as a robot file lists them in <adjacent> tags:
    # the links either side of the arm's short links (the elbow wam3, the wrist wam5 / wam6) and the hand with the
    # links next to it (the sphere model is fat around those joints)
    for a, b in (("wam1", "wam3"), ("wam2", "wam4"), ("wam3", "wam5"), ("wam4", "wam6"), ("wam5", "wam7"), ("wam6", "handbase")):
        r.adjacent.append((a, b))
    for k in range(3):
        r.adjacent += [("wam7", "Finger%d-1" % k), ("wam6", "Finger%d-1" % k), ("handbase", "Finger%d-2" % k), ("wam7", "Finger%d-2" % k)]
    # the arm's manipulator: end effector = the hand's base, tool frame 0.16 m along the arm
    r.manipulators.append(("arm", "handbase", [0.0, 0.0, 0.16, 0.0, 0.0, 0.0, 1.0]))
    return r


def tree30():
    """Synthetic 30-dof tree of SURVEY.md 8d config 5: 2-dof torso, two 7-dof arms
    continued by two 7-dof 'hands' (serial revolute chains), link length 0.15 m,
    alternating z / y axes, 2 spheres (r=0.05) per moving link -> 60 spheres."""
    r = RobotModel("tree30")
    R = JOINT_REVOLUTE
    r.add_link("base")
    names = []
    prev = "base"
    for i in range(2):
        nm = "torso%d" % i
        r.add_link(nm, prev, (0, 0, 0.15 if i else 0.3), joint=R,
                   axis=(0, 0, 1) if i % 2 == 0 else (0, 1, 0), limits=(-1.5, 1.5))
        names.append(nm)
        prev = nm
    for side, sgn in (("L", 1.0), ("R", -1.0)):
        prev = "torso1"
        for i in range(14):
            nm = "%s%d" % (side, i)
            off = (0, sgn * 0.2, 0.15) if i == 0 else (0, 0, 0.15)
            r.add_link(nm, prev, off, joint=R,
                       axis=(0, 0, 1) if i % 2 == 0 else (0, 1, 0), limits=(-2.0, 2.0))
            names.append(nm)
            prev = nm
    for nm in names:
        r.add_sphere(nm, (0, 0, 0.04), 0.05)
        r.add_sphere(nm, (0, 0, 0.11), 0.05)
    return r


class Tsr:
    """A task space region in the wire format `create` parses (tsr_create_parse, reference
    src/orcdchomp_mod.cpp:3068-3111): "manipindex bodyandlink" + T0w (rotation column by column, then
    translation) + Twe (the same) + Bw [6][2] (x y z roll pitch yaw; a row [0 0] is a hard constraint)."""

    def __init__(self, T0w_R=None, T0w_d=(0, 0, 0), Twe_R=None, Twe_d=(0, 0, 0), Bw=None, manipindex=0, bodyandlink="NULL"):
        self.T0w_R = np.eye(3) if T0w_R is None else np.asarray(T0w_R, dtype=np.float64).reshape(3, 3)
        self.T0w_d = np.asarray(T0w_d, dtype=np.float64)
        self.Twe_R = np.eye(3) if Twe_R is None else np.asarray(Twe_R, dtype=np.float64).reshape(3, 3)
        self.Twe_d = np.asarray(Twe_d, dtype=np.float64)
        self.Bw = np.zeros((6, 2)) if Bw is None else np.asarray(Bw, dtype=np.float64).reshape(6, 2)
        self.manipindex = manipindex
        self.bodyandlink = bodyandlink

    def serialize(self):
        vals = []
        for Rm, d in ((self.T0w_R, self.T0w_d), (self.Twe_R, self.Twe_d)):
            vals += [repr(float(Rm[r, c])) for c in range(3) for r in range(3)]
            vals += [repr(float(v)) for v in d]
        vals += [repr(float(v)) for v in self.Bw.reshape(12)]
        return "%d %s %s" % (self.manipindex, self.bodyandlink, " ".join(vals))
