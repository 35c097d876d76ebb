""""Compile the robot": GraphChain (+ world obstacles + allowed pairs) -> flat, GPU-friendly arrays.

This is the host-side one-time step that replaces the reference's per-frame flattening in
``Arm.__init__`` (numbotics/robots/arm.py:17-71) and the per-query pair/shape bookkeeping in
``Arm.collisions`` (arm.py:555-580).  The output is plain NumPy (a few KB) and is the ONLY thing the
C-ABI library (include/nbk.h) and the CPU oracle (oracle/) consume; both read the same numbers.

Moving-frame tree
  One frame per movable joint, in topological order.  Consecutive FIXED joints are pre-multiplied into
  the next moving joint's offset in the same association order as arm.py:38-52
  (``((I @ F1) @ F2) @ J``), so the constants are bit-identical to the reference's per-frame
  ``offsets``; a trailing FIXED run becomes the constant ``local`` pose of a link in its frame
  (arm.py:54-58).

Joint constants
  For a revolute joint the reference evaluates ``R_off @ (K - cos*(K - I) + sin*[a]x)``
  (robots/helpers.py:43-55).  The three constant 3x3 products M0 = R_off K, M1 = R_off (K - I),
  M2 = R_off [a]x are formed here once, so the per-configuration work is ``M0 - c*M1 + s*M2``.
  Prismatic joints (broken upstream) get M0 = R_off, M1 = M2 = 0 and a slide
  vector R_off @ axis.
"""
from dataclasses import dataclass, field

import numpy as np
import networkx as nx

from numbotics_amd.physics import Constraint
from numbotics_amd.utils import Shape

# device / oracle shape type codes
SH_SPHERE, SH_CAPSULE, SH_BOX, SH_CYLINDER, SH_PLANE, SH_HULL = 0, 1, 2, 3, 4, 5
JT_REVOLUTE, JT_PRISMATIC = 0, 1
MAX_JOINTS = 32


def _T34(T: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64)[:3, :4]).reshape(12)


def _skew(a):
    return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])


@dataclass
class FrameRef:
    """Where a named link frame lives: moving-joint frame index (-1 = base) + constant local pose."""
    joint: int
    local: np.ndarray                      # (4,4)
    path: np.ndarray                       # joint indices root -> this frame (int32)
    trailing_fixed: bool                   # reference FK/Jacobian are only right for these (Q1)


@dataclass
class KinematicModel:
    """Flat kinematic tree (no shapes)."""
    n_q: int
    joint_parent: np.ndarray               # (J,) int32
    joint_type: np.ndarray                 # (J,) int32
    joint_qidx: np.ndarray                 # (J,) int32
    joint_offset: np.ndarray               # (J,12) f64: merged fixed offsets @ joint offset (3x4)
    joint_axis: np.ndarray                 # (J,3)  f64
    joint_rot: np.ndarray                  # (J,27) f64: M0, M1, M2 row-major
    joint_trans: np.ndarray                # (J,3)  f64: offset translation
    joint_slide: np.ndarray                # (J,3)  f64: R_off @ axis for prismatic, 0 otherwise
    base_pose: np.ndarray                  # (12,)
    frames: dict = field(default_factory=dict)     # link name -> FrameRef
    link_names: list = field(default_factory=list)

    @property
    def n_joints(self):
        return int(self.joint_parent.shape[0])


def compile_kinematics(chain) -> KinematicModel:
    G = chain._G
    order = list(nx.topological_sort(G))
    root = order[0]
    jidx = chain.joint_index
    frame_of = {root: -1}
    local_of = {root: np.eye(4)}
    path_of = {root: []}
    trailing = {root: False}
    parents, types, qidx, offsets, axes = [], [], [], [], []
    for node in order[1:]:
        (u,) = list(G.predecessors(node))
        joint = G.edges[(u, node)]["joint"]
        if joint.type == Constraint.FIXED:
            frame_of[node] = frame_of[u]
            local_of[node] = local_of[u] @ joint.offset
            path_of[node] = path_of[u]
            trailing[node] = True
        elif joint.type in (Constraint.REVOLUTE, Constraint.PRISMATIC):
            k = len(parents)
            parents.append(frame_of[u])
            types.append(JT_REVOLUTE if joint.type == Constraint.REVOLUTE else JT_PRISMATIC)
            qidx.append(jidx[joint])
            offsets.append(local_of[u] @ joint.offset)
            axes.append(joint.axis)
            frame_of[node] = k
            local_of[node] = np.eye(4)
            path_of[node] = path_of[u] + [k]
            trailing[node] = False
        else:
            raise NotImplementedError(
                "SPHERICAL joints are not evaluable upstream either (robots/helpers.py:67-85 fails, "
                "tests/golden/golden_meta.json g1_reference_raised); not built")
    J = len(parents)
    if J > MAX_JOINTS:
        raise ValueError(f"at most {MAX_JOINTS} movable joints are supported, got {J}")
    rot = np.zeros((J, 27))
    trans = np.zeros((J, 3))
    slide = np.zeros((J, 3))
    for k in range(J):
        R = offsets[k][:3, :3]
        a = axes[k]
        trans[k] = offsets[k][:3, 3]
        if types[k] == JT_REVOLUTE:
            K = np.outer(a, a)
            rot[k, 0:9] = (R @ K).reshape(9)
            rot[k, 9:18] = (R @ (K - np.eye(3))).reshape(9)
            rot[k, 18:27] = (R @ _skew(a)).reshape(9)
        else:
            rot[k, 0:9] = R.reshape(9)
            slide[k] = R @ a
    # exact zeros are stored as +0.0: the device evaluates axis-aligned joints with literal zeros in place of the table entries that
    # are zero by construction (csrc/nbk.hip joint_apply), and x + 0.0 turns a -0.0 into the +0.0 that literal is
    rot, trans, slide = rot + 0.0, trans + 0.0, slide + 0.0
    km = KinematicModel(
        n_q=chain.dof,
        joint_parent=np.array(parents, dtype=np.int32).reshape(J),
        joint_type=np.array(types, dtype=np.int32).reshape(J),
        joint_qidx=np.array(qidx, dtype=np.int32).reshape(J),
        joint_offset=np.array([_T34(o) for o in offsets], dtype=np.float64).reshape(J, 12),
        joint_axis=np.array(axes, dtype=np.float64).reshape(J, 3),
        joint_rot=rot, joint_trans=trans, joint_slide=slide,
        base_pose=_T34(chain.base_pose),
        link_names=[l._name for l in chain._links],
    )
    for node in order:
        km.frames[node] = FrameRef(joint=frame_of[node], local=local_of[node],
                                   path=np.array(path_of[node], dtype=np.int32), trailing_fixed=trailing[node])
    return km


class HullTable:
    """Vertex / face-plane tables of the convex hulls of a scene (MESH shapes), shared by the device descriptor and the
    oracle.  A shape of type SH_HULL names its hull in param[0]."""

    def __init__(self):
        self.verts, self.planes = [], []
        self.vert_begin, self.face_begin = [0], [0]

    def add(self, part) -> int:
        self.verts.append(np.asarray(part.vertices, dtype=np.float64).reshape(-1, 3))
        self.planes.append(np.asarray(part.planes, dtype=np.float64).reshape(-1, 4))
        self.vert_begin.append(self.vert_begin[-1] + len(self.verts[-1]))
        self.face_begin.append(self.face_begin[-1] + len(self.planes[-1]))
        return len(self.verts) - 1

    def arrays(self):
        V = np.concatenate(self.verts) if self.verts else np.zeros((0, 3))
        P = np.concatenate(self.planes) if self.planes else np.zeros((0, 4))
        return (np.array(self.vert_begin, dtype=np.int32), np.ascontiguousarray(V, dtype=np.float64),
                np.array(self.face_begin, dtype=np.int32), np.ascontiguousarray(P, dtype=np.float64))


BULLET_MARGIN = 0.04          # btCollisionShape's CONVEX_DISTANCE_MARGIN
BULLET_HULL_MARGIN = 0.001    # what pybullet's createCollisionShape gives the convex hulls of a GEOM_MESH


def bullet_margin(shape, info) -> float:
    """The collision margin Bullet itself would give this shape (``bullet_margins=True`` scenes; third-party behaviour restated
    from Bullet's public sources, NOT pinned: pybullet is absent).  btBoxShape / btCylinderShape shrink their implicit
    dimensions by the margin and add it back as a rounding -- exactly this build's ``core (+) ball(margin)`` -- with the margin
    limited to a tenth of the smallest half extent (setSafeMargin); spheres and capsules are exact; a hull is inflated."""
    if shape in (Shape.CUBE, Shape.CUBOID):
        return float(min(BULLET_MARGIN, 0.1 * np.min(np.asarray(info['half_extents'], dtype=np.float64))))
    if shape == Shape.CYLINDER:
        return float(min(BULLET_MARGIN, 0.1 * min(float(info['radius']), float(info['height']) / 2.0)))
    if shape == Shape.MESH:
        return BULLET_HULL_MARGIN
    return 0.0


def _margin(cs, bullet_margins: bool) -> float:
    info = cs._shape_info
    if 'collision_margin' in info:
        return float(info['collision_margin'])
    return bullet_margin(cs.shape, info) if bullet_margins else 0.0


def _shape_records(cs, owner_pose_local: np.ndarray, hulls: HullTable, bullet_margins: bool = False):
    """[(type, 3x4 pose in the owner frame, params[4])] for one CollisionShape: one record, except for a MESH with
    ``convex_decomposition=True``, which yields one convex hull per object of its file (numbotics/utils/shape.py:81-94 -> Bullet
    GEOM_MESH; without the flag a mesh is one hull of all its vertices, utils/mesh.py:mesh_hulls)."""
    if cs.shape == Shape.MESH:
        from numbotics_amd.utils.mesh import mesh_hulls
        info = cs._shape_info
        kw = {k: info[k] for k in ('mesh_scale', 'auto_center', 'convex_decomposition') if k in info}
        T = owner_pose_local @ cs.offset
        out = []
        for part in mesh_hulls(info['filename'], **kw):
            Tc = T.copy()
            Tc[:3, 3] = T[:3, :3] @ part.center + T[:3, 3]          # the hull's local origin = the mean of its vertices
            p = np.zeros(4)
            p[0] = float(hulls.add(part))
            p[3] = _margin(cs, bullet_margins)
            out.append((SH_HULL, Tc, p))
        return out
    return [_shape_record(cs, owner_pose_local, bullet_margins)]


def chain_link_poses(chain) -> dict:
    """World poses {link name: 4x4} of every link of ``chain`` at ``chain.configuration`` -- how the links of ANOTHER chain enter an
    arm's scene: as static obstacles (upstream: ``Arm.collision_pairs`` pairs every link with every link of the other chains in the
    world, robots/arm.py:226-243, and Bullet holds those bodies at their current joint state).  Host-side scene set-up in NumPy (a
    few dozen 4x4 products, once per compile), same joint formula as the kernels: L = M0 - cos(q) M1 + sin(q) M2."""
    kin = compile_kinematics(chain)
    q = np.asarray(chain.configuration, dtype=np.float64)
    base = np.vstack([kin.base_pose.reshape(3, 4), [0.0, 0.0, 0.0, 1.0]])
    frames = []
    for k in range(kin.n_joints):
        M = kin.joint_rot[k].reshape(3, 3, 3)
        qk = q[kin.joint_qidx[k]]
        X = np.eye(4)
        if kin.joint_type[k] == JT_REVOLUTE:
            X[:3, :3] = M[0] - np.cos(qk) * M[1] + np.sin(qk) * M[2]
            X[:3, 3] = kin.joint_trans[k]
        else:
            X[:3, :3] = M[0]
            X[:3, 3] = kin.joint_trans[k] + qk * kin.joint_slide[k]
        parent = base if kin.joint_parent[k] < 0 else frames[kin.joint_parent[k]]
        frames.append(parent @ X)
    return {name: (base if fr.joint < 0 else frames[fr.joint]) @ fr.local for name, fr in kin.frames.items()}


def _shape_record(cs, owner_pose_local: np.ndarray, bullet_margins: bool = False):
    """(type, 3x4 pose in the owner frame, params[4]) for one primitive CollisionShape."""
    info = cs._shape_info
    T = owner_pose_local @ cs.offset
    p = np.zeros(4)
    p[3] = _margin(cs, bullet_margins)
    if cs.shape in (Shape.CUBE, Shape.CUBOID):
        p[0:3] = np.asarray(info['half_extents'], dtype=np.float64)
        return SH_BOX, T, p
    if cs.shape == Shape.SPHERE:
        p[0] = float(info['radius'])
        return SH_SPHERE, T, p
    if cs.shape == Shape.CYLINDER:
        p[0], p[1] = float(info['radius']), float(info['height']) / 2.0
        return SH_CYLINDER, T, p
    if cs.shape == Shape.CAPSULE:
        p[0], p[1] = float(info['radius']), float(info['height']) / 2.0
        return SH_CAPSULE, T, p
    if cs.shape == Shape.PLANE:
        n = np.asarray(info['normal'], dtype=np.float64)
        nw = T[:3, :3] @ (n / np.linalg.norm(n))
        p[0:3] = nw
        return SH_PLANE, T, p
    raise ValueError(f"shape {cs.shape} has no collision geometry")


@dataclass
class SceneModel:
    """Shapes + allowed pairs on top of a KinematicModel."""
    kin: KinematicModel
    rshape_frame: np.ndarray               # (S,) int32
    rshape_type: np.ndarray                # (S,) int32
    rshape_local: np.ndarray               # (S,12) f64
    rshape_param: np.ndarray               # (S,4)  f64
    rshape_link: np.ndarray                # (S,) int32  index into chain._links
    wshape_type: np.ndarray                # (W,) int32
    wshape_pose: np.ndarray                # (W,12) f64 (world)
    wshape_param: np.ndarray               # (W,4)
    wshape_obj: np.ndarray                 # (W,) int32 index into `objects`
    pair_a: np.ndarray                     # (P,) int32 robot shape
    pair_b: np.ndarray                     # (P,) int32 robot shape, or S + world shape
    objects: list = field(default_factory=list)
    links: list = field(default_factory=list)
    # convex hulls of MESH shapes (a shape of type SH_HULL names its hull in param[0])
    hull_vert_begin: np.ndarray = field(default_factory=lambda: np.zeros(1, dtype=np.int32))
    hull_verts: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    hull_face_begin: np.ndarray = field(default_factory=lambda: np.zeros(1, dtype=np.int32))
    hull_planes: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))

    @property
    def n_hulls(self):
        return int(self.hull_vert_begin.shape[0]) - 1

    @property
    def n_rshapes(self):
        return int(self.rshape_type.shape[0])

    @property
    def n_wshapes(self):
        return int(self.wshape_type.shape[0])

    @property
    def n_pairs(self):
        return int(self.pair_a.shape[0])

    def pair_members(self, p: int):
        """(subject link, target link-or-object) of pair ``p``."""
        a, b = int(self.pair_a[p]), int(self.pair_b[p])
        subj = self.links[self.rshape_link[a]]
        if b < self.n_rshapes:
            return subj, self.links[self.rshape_link[b]]
        return subj, self.objects[self.wshape_obj[b - self.n_rshapes]]


def compile_scene(chain, kin: KinematicModel, pairs, compound: bool = True, bullet_margins: bool = True) -> SceneModel:
    """``pairs``: iterable of (Link, Link | PhysicsObject) as produced by ``Arm.collision_pairs()``.
    ``bullet_margins``: shapes without an explicit ``collision_margin`` get the margin Bullet would give them
    (``bullet_margin`` above) instead of 0 -- the setting closest to what the reference's ``getClosestPoints`` measures."""
    from numbotics_amd.physics import PhysicsObject, Link
    links = chain._links
    link_index = {l._name: i for i, l in enumerate(links)}
    r_frame, r_type, r_local, r_param, r_link = [], [], [], [], []
    hulls = HullTable()
    shapes_of_link = {}
    for i, link in enumerate(links):
        shapes = link._collision_shapes if compound else (
            [link._collision_shape] if link._collision_shape.shape != Shape.EMPTY else [])
        fr = kin.frames[link._name]
        ids = []
        for cs in shapes:
            if cs.shape == Shape.EMPTY:
                continue
            for t, T, p in _shape_records(cs, fr.local, hulls, bullet_margins):
                if t == SH_PLANE:
                    raise ValueError("a PLANE cannot be a robot link shape")
                ids.append(len(r_type))
                r_frame.append(fr.joint)
                r_type.append(t)
                r_local.append(_T34(T))
                r_param.append(p)
                r_link.append(i)
        shapes_of_link[i] = ids
    S = len(r_type)
    objects, w_type, w_pose, w_param, w_obj = [], [], [], [], []
    shapes_of_obj = {}
    other_poses = {}                     # id(other chain) -> {link name: world pose at its current configuration}
    pa, pb = [], []
    def in_chain(x):
        return isinstance(x, Link) and x._body_id == chain._pyb_id and x._world_name == chain._world_name

    for a, b in pairs:
        if not in_chain(a):
            a, b = b, a
        if not in_chain(a):
            raise ValueError("a collision pair must contain a link of this chain")
        ia = link_index[a._name]
        if in_chain(b):
            ib = link_index[b._name]
            for sa in shapes_of_link[ia]:
                for sb in shapes_of_link[ib]:
                    pa.append(sa)
                    pb.append(sb)
        elif isinstance(b, PhysicsObject):
            key = id(b)
            if key not in shapes_of_obj:
                ids = []
                if b._collision_shape.shape != Shape.EMPTY:
                    for t, T, p in _shape_records(b._collision_shape, b.pose, hulls, bullet_margins):
                        ids.append(len(w_type))
                        w_type.append(t)
                        w_pose.append(_T34(T))
                        w_param.append(p)
                        w_obj.append(len(objects))
                objects.append(b)
                shapes_of_obj[key] = ids
            for sa in shapes_of_link[ia]:
                for sb in shapes_of_obj[key]:
                    pa.append(sa)
                    pb.append(S + sb)
        elif isinstance(b, Link):
            # a link of another chain in the same world: a static obstacle at that chain's current configuration
            key = id(b)
            if key not in shapes_of_obj:
                other = next((o for o in chain.world.objects() if not isinstance(o, PhysicsObject) and getattr(o, '_pyb_id', None) == b._body_id), None)
                if other is None:
                    raise ValueError(f"link {b._name} belongs to no chain of this world")
                if id(other) not in other_poses:
                    other_poses[id(other)] = chain_link_poses(other)
                ids = []
                for cs in (b._collision_shapes if compound else [b._collision_shape]):
                    if cs.shape == Shape.EMPTY:
                        continue
                    for t, T, p in _shape_records(cs, other_poses[id(other)][b._name], hulls, bullet_margins):
                        if t == SH_PLANE:
                            raise ValueError("a PLANE cannot be a robot link shape")
                        ids.append(len(w_type))
                        w_type.append(t)
                        w_pose.append(_T34(T))
                        w_param.append(p)
                        w_obj.append(len(objects))
                objects.append(b)
                shapes_of_obj[key] = ids
            for sa in shapes_of_link[ia]:
                for sb in shapes_of_obj[key]:
                    pa.append(sa)
                    pb.append(S + sb)
        else:
            raise ValueError(f"cannot pair a link with {type(b).__name__}")
    # stable order: by subject shape, then target (the device loops shape-A-major)
    if pa:
        order = np.lexsort((np.array(pb), np.array(pa)))
        pa = np.array(pa, dtype=np.int32)[order]
        pb = np.array(pb, dtype=np.int32)[order]
    W = len(w_type)
    hvb, hv, hfb, hp = hulls.arrays()
    return SceneModel(
        kin=kin,
        rshape_frame=np.array(r_frame, dtype=np.int32).reshape(S),
        rshape_type=np.array(r_type, dtype=np.int32).reshape(S),
        rshape_local=np.array(r_local, dtype=np.float64).reshape(S, 12),
        rshape_param=np.array(r_param, dtype=np.float64).reshape(S, 4),
        rshape_link=np.array(r_link, dtype=np.int32).reshape(S),
        wshape_type=np.array(w_type, dtype=np.int32).reshape(W),
        wshape_pose=np.array(w_pose, dtype=np.float64).reshape(W, 12),
        wshape_param=np.array(w_param, dtype=np.float64).reshape(W, 4),
        wshape_obj=np.array(w_obj, dtype=np.int32).reshape(W),
        pair_a=np.array(pa, dtype=np.int32).reshape(-1),
        pair_b=np.array(pb, dtype=np.int32).reshape(-1),
        objects=objects, links=list(links),
        hull_vert_begin=hvb, hull_verts=hv, hull_face_begin=hfb, hull_planes=hp,
    )
