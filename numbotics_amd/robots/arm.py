"""``Arm``: the reference's robot front-end (numbotics/robots/arm.py) over the MI355X engine.

Same method names, argument order, defaults and exception types as upstream.  What changes:

  * ``forward_kinematics`` / ``jacobian`` run as HIP kernels (one configuration per lane) instead of
    serial numba loops (robots/helpers.py:91-187); they accept NumPy arrays or torch CUDA tensors.
  * ``in_collision`` / ``closest_to`` / ``collisions`` no longer push one ``q`` at a time through
    PyBullet (arm.py:555-604, physics/chain.py:944-969): the allowed pairs are compiled once into a
    device descriptor and whole batches are checked per launch.  ``in_collision`` additionally accepts
    ``(B, dof)`` input and then returns ``(B,)`` bool; the ``(dof,)`` form returns a Python bool as
    upstream.
  * Reference quirks: Q1 -- FK/Jacobian here are correct for every link, upstream
    only for frames that end in a FIXED joint; Q2 -- ``global_pose=False`` is treated as ``None``;
    Q3 -- the default self-collision rule is upstream's *effective* one (all non-adjacent shaped
    pairs); ``weld_filter=True`` enables the rule upstream intended.
"""
import contextlib

import numpy as np
import networkx as nx

from numbotics_amd.physics import Chain, Constraint, Link, PhysicsObject, Proximity
from numbotics_amd.utils import Shape, logger
from .robot import Robot
from .model import compile_kinematics, compile_scene


def _is_tensor(x):
    return type(x).__module__.startswith("torch")


class Arm(Robot):

    def __init__(self, chain, weld_filter: bool = False, compound: bool = True, bullet_margins: bool = True):
        """``bullet_margins`` (additive, default True): every box / cylinder / mesh hull that has no explicit ``collision_margin``
        gets the margin Bullet itself applies to the shapes ``pybullet.createCollisionShape`` builds for the reference
        (numbotics/utils/shape.py:60-109; robots/model.py ``bullet_margin``: boxes / cylinders are rounded by
        min(0.04, a tenth of the smallest half extent), hulls inflated by 0.001) -- the restatement closest to the reference's
        ``getClosestPoints`` answers.  ``False`` = the sharp analytic shapes (margin 0); can be switched later through the
        ``bullet_margins`` property."""
        super().__init__(chain)
        self._weld_filter = weld_filter
        self._compound = compound
        self._bullet_margins = bool(bullet_margins)
        self._kin = compile_kinematics(chain)
        # per-frame flattened sequences in the reference's own format (arm.py:61), kept for inspection
        self._link_joint_sequence = {}
        self._links_from_nodes = {}
        root = next(iter(nx.topological_sort(chain._G)))
        for node in chain._G.nodes:
            self._links_from_nodes[node] = chain._G.nodes[node]["link"]
            fr = self._kin.frames[node]
            if node == root:
                self._link_joint_sequence[node] = tuple()
                continue
            offs = [self._T44(self._kin.joint_offset[k]) for k in fr.path]
            if fr.trailing_fixed:
                offs.append(fr.local)
            self._link_joint_sequence[node] = (
                np.array(offs).reshape(-1, 4, 4),
                self._kin.joint_axis[fr.path].reshape(-1, 3),
                np.array([0 if self._kin.joint_type[k] == 0 else 1 for k in fr.path], dtype=np.int64),
                self._kin.joint_qidx[fr.path].astype(np.int64),
            )
        self._additional_self_collision_pairs = set()
        self._void_self_collision_pairs = set()
        self._additional_collision_pairs = set()
        self._void_collision_pairs = set()
        self._pairs_version = 0
        self._kin_dev = None
        self._scene_cache = None       # (key, SceneModel, DeviceModel)
        self.self_collision_pairs()

    @staticmethod
    def _T44(p12):
        T = np.eye(4)
        T[:3, :4] = np.asarray(p12).reshape(3, 4)
        return T

    # ---- passthrough properties (arm.py:73-124) ----------------------------------------------------
    @property
    def base_pose(self):
        return self._chain.base_pose

    @property
    def joint_limits(self):
        return self._chain.joint_limits

    @property
    def dof(self):
        return self._chain.dof

    @property
    def configuration(self):
        return self._chain.configuration

    @configuration.setter
    def configuration(self, q):
        self._chain.configuration = q

    @contextlib.contextmanager
    def stateless(self):
        """Upstream saves/restores PyBullet state around a query (arm.py:128-146); queries here are pure."""
        yield

    @contextlib.contextmanager
    def pool(self, poolsize: int = 1):
        """Upstream clones the world per CPU thread (arm.py:149-187).  The device path batches instead;
        the engine is re-entrant, so the pool is ``poolsize`` references to this Arm."""
        yield tuple(self for _ in range(poolsize))

    # ---- collision pair bookkeeping (arm.py:190-366) -----------------------------------------------
    def _in_chain(self, x):
        return isinstance(x, Link) and x._body_id == self._chain._pyb_id and x._world_name == self._chain._world_name

    def self_collision_pairs(self):
        if not hasattr(self, '_self_collision_pairs'):
            pairs = set()
            G_u = self._chain._G.to_undirected()
            shaped = [l for l in self._chain._links if l._collision_shape.shape != Shape.EMPTY]
            for a in shaped:
                for b in shaped:
                    if a is b:
                        continue
                    if self._weld_filter:
                        path = nx.shortest_path(G_u, a._name, b._name)
                        moving = sum(1 for u, v in zip(path[:-1], path[1:])
                                     if G_u.edges[(u, v)]["joint"].type != Constraint.FIXED)
                        if moving < 2:
                            continue
                    if not G_u.has_edge(a._name, b._name) and (b, a) not in pairs:
                        pairs.add((a, b))
            self._self_collision_pairs = pairs
        return self._self_collision_pairs.union(self._additional_self_collision_pairs).difference(
            self._void_self_collision_pairs)

    def collision_pairs(self):
        pairs = set()
        for link in self._chain._links:
            if link._collision_shape.shape == Shape.EMPTY:
                continue
            for obj in self._chain.world.objects():
                if obj == self._chain:
                    continue
                if isinstance(obj, PhysicsObject):
                    if obj._collision_shape.shape == Shape.EMPTY:
                        continue
                    pairs.add((link, obj))
                elif isinstance(obj, Chain):
                    for other in obj._links:
                        pairs.add((link, other))
        return pairs.union(self.self_collision_pairs()).union(self._additional_collision_pairs).difference(
            self._void_collision_pairs)

    def _resolve(self, x):
        if isinstance(x, str):
            found = self._links_from_nodes.get(x)
            if found is None:
                found = self._chain.world.get_object(x)
                if found is None:
                    raise ValueError(f"Object name: {x} must be a valid object in the world")
            return found
        return x

    def _ordered(self, link_a, link_b, verb):
        link_a, link_b = self._resolve(link_a), self._resolve(link_b)
        a_in, b_in = self._in_chain(link_a), self._in_chain(link_b)
        if not a_in and not b_in:
            logger.warning(f"Did not {verb} collision pair between {link_a.name} and {link_b.name} "
                           "because neither is in the chain")
            return None
        if not a_in:
            link_a, link_b = link_b, link_a
            b_in = False
        return link_a, link_b, b_in

    @staticmethod
    def _has(pairs, a, b):
        return (a, b) in pairs or (b, a) in pairs

    @staticmethod
    def _drop(pairs, a, b):
        if (a, b) in pairs:
            pairs.remove((a, b))
        elif (b, a) in pairs:
            pairs.remove((b, a))

    def add_collision_pair(self, link_a, link_b):
        r = self._ordered(link_a, link_b, "add")
        if r is None:
            return
        a, b, is_self = r
        if is_self:
            if self._has(self.self_collision_pairs(), a, b):
                logger.warning(f"Did not add collision pair between {a.name} and {b.name} because it is "
                               "already a self collision pair")
            else:
                # upstream files this under _additional_collision_pairs, where its own self-collision
                # filter (arm.py:569-572) never looks; here the pair becomes active
                self._additional_self_collision_pairs.add((a, b))
            self._drop(self._void_self_collision_pairs, a, b)
        else:
            if self._has(self.collision_pairs(), a, b):
                logger.warning(f"Did not add collision pair between {a.name} and {b.name} because it is "
                               "already a collision pair")
            else:
                self._additional_collision_pairs.add((a, b))
            self._drop(self._void_collision_pairs, a, b)
        self._pairs_version += 1

    def remove_collision_pair(self, link_a, link_b):
        r = self._ordered(link_a, link_b, "remove")
        if r is None:
            return
        a, b, is_self = r
        if is_self:
            self._drop(self._additional_self_collision_pairs, a, b)
            if self._has(self._void_self_collision_pairs, a, b):
                logger.warning(f"Did not remove collision pair between {a.name} and {b.name} because it is "
                               "has already been removed")
            elif (a, b) in self.self_collision_pairs():
                self._void_self_collision_pairs.add((a, b))
            elif (b, a) in self.self_collision_pairs():
                self._void_self_collision_pairs.add((b, a))
        else:
            self._drop(self._additional_collision_pairs, a, b)
            if self._has(self._void_collision_pairs, a, b):
                logger.warning(f"Did not remove collision pair between {a.name} and {b.name} because it is "
                               "has already been removed")
            elif (a, b) in self.collision_pairs():
                self._void_collision_pairs.add((a, b))
            elif (b, a) in self.collision_pairs():
                self._void_collision_pairs.add((b, a))
        self._pairs_version += 1

    # ---- descriptors ----------------------------------------------------------------------------------
    def _sorted_pairs(self, pairs):
        order = {l._name: i for i, l in enumerate(self._chain._links)}

        def key(p):
            a, b = p
            if not self._in_chain(a):
                a, b = b, a
            kb = (0, order[b._name]) if self._in_chain(b) else (1, b.name)
            return (order[a._name], kb)
        return sorted(pairs, key=key)

    @property
    def bullet_margins(self) -> bool:
        return self._bullet_margins

    @bullet_margins.setter
    def bullet_margins(self, on: bool):
        self._bullet_margins = bool(on)
        self._scene_cache = None

    def scene_model(self, pairs=None):
        """The flat SceneModel (robots/model.py) of the current world and pair set."""
        if pairs is not None:
            return compile_scene(self._chain, self._refreshed_kin(), self._sorted_pairs(pairs), self._compound, self._bullet_margins)
        key = (self._chain.world._revision, self._pairs_version, self._bullet_margins)
        if self._scene_cache is None or self._scene_cache[0] != key:
            sm = compile_scene(self._chain, self._refreshed_kin(), self._sorted_pairs(self.collision_pairs()),
                               self._compound, self._bullet_margins)
            self._scene_cache = (key, sm, None)
        return self._scene_cache[1]

    def _refreshed_kin(self):
        bp = np.ascontiguousarray(self._chain.base_pose[:3, :4]).reshape(12)
        if not np.array_equal(bp, self._kin.base_pose):
            self._kin.base_pose = bp
            self._kin_dev = None
        return self._kin

    def _kin_device(self):
        from numbotics_amd.engine import DeviceModel
        self._refreshed_kin()
        if self._kin_dev is None:
            self._kin_dev = DeviceModel(self._kin)
        return self._kin_dev

    def _scene_device(self):
        from numbotics_amd.engine import DeviceModel
        sm = self.scene_model()
        key, _, dev = self._scene_cache
        if dev is None:
            dev = DeviceModel(sm)
            self._scene_cache = (key, sm, dev)
        return sm, dev

    # ---- kinematics -----------------------------------------------------------------------------------
    def _check_frame_q(self, q, frame):
        if frame not in self._kin.frames:
            raise ValueError(f"Frame {frame} not found in chain")
        if q.shape[-1] != self.dof:
            raise ValueError(f"q must have {self.dof} elements")

    @staticmethod
    def _reshape(x, shape):
        return x.reshape(shape)

    def _pose_arg(self, pose, q_shape, name):
        """-> (single 4x4 ndarray | None, batched (B,4,4) | None)"""
        if pose is None:
            return None, None
        if tuple(pose.shape[-2:]) != (4, 4):
            raise ValueError(f"{name} must be a 4x4 matrix")
        if pose.ndim == 2:
            single = pose.detach().cpu().numpy() if _is_tensor(pose) else np.asarray(pose, dtype=np.float64)
            return single, None
        if tuple(pose.shape[:-2]) != tuple(q_shape[:-1]):
            raise ValueError(f"{name} must have the same batch dimensions as q")
        return None, pose.reshape(-1, 4, 4)

    def forward_kinematics(self, q, frame: str, use_com: bool = False, local_pose=None):
        self._check_frame_q(q, frame)
        shape = tuple(q.shape)
        single, batched = self._pose_arg(local_pose, shape, "local_pose")
        extra = None
        if use_com:
            extra = self._links_from_nodes[frame]._offset
        if single is not None:
            extra = single if extra is None else extra @ single
        T = self._kin_device().fk(q.reshape(-1, self.dof), frame, extra_local=extra, local_pose=batched)
        if len(shape) == 1:
            return T[0]
        return T.reshape(*shape[:-1], 4, 4)

    def forward_kinematics_all(self, q, frames=None, use_com: bool = False):
        """Poses of many links in one launch (additive; upstream loops ``forward_kinematics`` over the link names):
        ``(..., dof)`` -> ``(..., L, 4, 4)`` and the list of frame names (default: every link of the chain, chain order).
        Each pose is bit-identical to ``forward_kinematics(q, name, use_com)``."""
        if q.shape[-1] != self.dof:
            raise ValueError(f"q must have {self.dof} elements")
        names = list(self._kin.frames.keys()) if frames is None else list(frames)
        for f in names:
            if f not in self._kin.frames:
                raise ValueError(f"Frame {f} not found in chain")
        extra = {f: self._links_from_nodes[f]._offset for f in names} if use_com else None
        shape = tuple(q.shape)
        T = self._kin_device().fk_frames(q.reshape(-1, self.dof), names, extra_locals=extra)
        if len(shape) == 1:
            return T[0], names
        return T.reshape(*shape[:-1], len(names), 4, 4), names

    def jacobian(self, q, frame: str, use_com: bool = False, local_pose=None, global_pose=False):
        self._check_frame_q(q, frame)
        if global_pose is False:          # upstream's default crashes on `.shape` (App. A Q2)
            global_pose = None
        shape = tuple(q.shape)
        if global_pose is not None and local_pose is not None:
            raise ValueError("local_pose and global_pose cannot both be provided")
        l_single, l_batched = self._pose_arg(local_pose, shape, "local_pose")
        g_single, g_batched = self._pose_arg(global_pose, shape, "global_pose")
        q2 = q.reshape(-1, self.dof)
        B = q2.shape[0]
        extra = self._links_from_nodes[frame]._offset if use_com else None
        if l_single is not None:
            extra = l_single if extra is None else extra @ l_single
        if g_single is not None:
            g_batched = np.tile(g_single[None], (B, 1, 1))
        if len(self._kin.frames[frame].path) == 0:
            J = np.zeros((B, 6, self.dof))      # arm.py:455-457
            if _is_tensor(q):
                import torch
                J = torch.zeros((B, 6, self.dof), dtype=torch.float64, device=q.device)
        else:
            J = self._kin_device().jacobian(q2, frame, extra_local=extra, local_pose=l_batched, global_pose=g_batched)
        if len(shape) == 1:
            return J[0]
        return J.reshape(*shape[:-1], 6, self.dof)

    def inverse_kinematics(self, pose, q0, frame: str, use_com: bool = False, use_limits: bool = False,
                           tol: float = 1e-6, max_iter: int = 100, max_failures: int = 15):
        """Damped-least-squares IK, signature and return shapes of upstream (arm.py:464-552): ``(success, q)`` with
        ``success`` a flat bool array and ``q`` shaped ``(dof,)`` or ``(*batch_dims, dof)``.  All problems iterate
        inside one launch (``nbk_ik_batch``).  Upstream's own loop cannot run (its ``jacobian`` call hits Q2); the
        per-element arithmetic it writes down is what the kernel follows, with a Cholesky solve for LAPACK's LU."""
        if frame not in self._kin.frames:
            raise ValueError(f"Frame {frame} not found in chain")
        if q0.shape[-1] != self.dof:
            raise ValueError(f"q0 must have {self.dof} elements")
        if max_iter < 1:
            raise ValueError("max_iter must be greater than 0")
        if tuple(pose.shape[-2:]) != (4, 4):
            raise ValueError("pose must be a 4x4 matrix")
        batch_dims = None
        if pose.ndim > 2:
            batch_dims = tuple(pose.shape[:-2])
            pose = pose.reshape(-1, 4, 4)
        if q0.ndim > 1:
            if batch_dims:
                if tuple(q0.shape[:-1]) != batch_dims:
                    raise ValueError("pose and q0 must have the same batch dimensions")
            else:
                batch_dims = tuple(q0.shape[:-1])
        q0f = q0.reshape(-1, self.dof)
        posef = pose.reshape(-1, 4, 4)
        n = max(int(q0f.shape[0]), int(posef.shape[0]))
        if _is_tensor(q0f) or _is_tensor(posef):
            import torch
            q0f = q0f if _is_tensor(q0f) else torch.from_numpy(np.ascontiguousarray(q0f, dtype=np.float64))
            posef = posef if _is_tensor(posef) else torch.from_numpy(np.ascontiguousarray(posef, dtype=np.float64))
            if q0f.shape[0] != n:
                q0f = q0f.expand(n, self.dof)
            if posef.shape[0] != n:
                posef = posef.expand(n, 4, 4)
            q0f, posef = q0f.cuda(), posef.cuda()
        else:
            if q0f.shape[0] != n:
                q0f = np.tile(q0f, (n, 1))
            if posef.shape[0] != n:
                posef = np.tile(posef, (n, 1, 1))
        extra = self._links_from_nodes[frame]._offset if use_com else None
        limits = np.ascontiguousarray(self.joint_limits, dtype=np.float64) if use_limits else None
        ok, q, _, _ = self._kin_device().ik(posef, q0f, frame, extra_local=extra, limits=limits, tol=tol,
                                            max_iter=max_iter, max_failures=max_failures)
        if batch_dims is None:
            return ok, q[0]
        return ok, q.reshape(*batch_dims, self.dof)

    # ---- collision queries ------------------------------------------------------------------------------
    def _proximities(self, q, sm, dist, wit):
        out = []
        for p in range(sm.n_pairs):
            subj, targ = sm.pair_members(p)
            out.append(Proximity(subject=subj, target=targ, position_on_subject=wit[p, 0:3].copy(),
                                 position_on_target=wit[p, 3:6].copy(), normal_target_to_subject=wit[p, 6:9].copy(),
                                 distance=float(dist[p])))
        return out

    def collisions(self, q):
        """One Proximity per allowed primitive pair (compound links contribute one per element pair)."""
        if tuple(q.shape) != (self.dof,):
            raise ValueError(f"q must be a 1D array with {self.dof} elements")
        sm, dev = self._scene_device()
        if sm.n_pairs == 0:
            return []
        qn = q.detach().cpu().numpy() if _is_tensor(q) else np.asarray(q, dtype=np.float64)
        dist, wit = dev.pair_distances(qn.reshape(1, -1), witness=True)
        return self._proximities(qn, sm, dist[0], wit[0])

    def self_collisions(self, q):
        if tuple(q.shape) != (self.dof,):
            raise ValueError(f"q must be a 1D array with {self.dof} elements")
        return [p for p in self.collisions(q) if self._in_chain(p.target)
                and self._has(self.self_collision_pairs(), p.subject, p.target)]

    def closest_to(self, q):
        """``min(collisions(q), key=distance)`` (arm.py:599-600) without building the other Proximity records."""
        if tuple(q.shape) != (self.dof,):
            raise ValueError(f"q must be a 1D array with {self.dof} elements")
        sm, dev = self._scene_device()
        if sm.n_pairs == 0:
            raise ValueError("min() arg is an empty sequence")
        qn = q.detach().cpu().numpy() if _is_tensor(q) else np.asarray(q, dtype=np.float64)
        dist, wit = dev.pair_distances(qn.reshape(1, -1), witness=True)
        p = int(np.argmin(dist[0]))                       # first minimum, as min() over the list
        subj, targ = sm.pair_members(p)
        return Proximity(subject=subj, target=targ, position_on_subject=wit[0, p, 0:3].copy(),
                         position_on_target=wit[0, p, 3:6].copy(), normal_target_to_subject=wit[0, p, 6:9].copy(),
                         distance=float(dist[0, p]))

    def in_collision(self, q, threshold: float = 0.0):
        """``(dof,)`` -> bool as upstream (arm.py:603-604); ``(..., dof)`` -> bool array/tensor (additive)."""
        if q.shape[-1] != self.dof:
            raise ValueError(f"q must have {self.dof} elements")
        sm, dev = self._scene_device()
        if q.ndim == 1:
            if sm.n_pairs == 0:
                raise ValueError("min() arg is an empty sequence")       # what upstream's closest_to raises
            if isinstance(q, np.ndarray):
                return dev.validity_scalar(q, threshold)                 # host fast path: pinned staging, one wait
            return bool(dev.validity(q.reshape(1, -1), threshold)[0])
        mask = dev.validity(q.reshape(-1, self.dof), threshold)
        return mask.reshape(tuple(q.shape[:-1]))

    def pair_distances(self, q):
        """(B, P) signed distances of every allowed primitive pair (additive, batched ``collisions``)."""
        sm, dev = self._scene_device()
        return dev.pair_distances(q.reshape(-1, self.dof))

    def closest_distance(self, q):
        """(B,) min signed distance and (B,) pair index (additive, batched ``closest_to``)."""
        sm, dev = self._scene_device()
        return dev.closest(q.reshape(-1, self.dof))

    def _pair_selection(self, sm, obj, link):
        """Indices (into the scene's primitive pair list) of the proximities ``distance_to(q, obj, link)`` returns."""
        pairs = self.collision_pairs()
        sel = []
        for p in range(sm.n_pairs):
            subj, targ = sm.pair_members(p)
            if not (targ == obj or (isinstance(obj, Chain) and self._in_chain(targ) and obj == self._chain)):
                continue
            if link is None:
                if self._has(pairs, subj, targ):
                    sel.append(p)
            elif subj == link:
                sel.append(p)
        return sel

    def distance_to(self, q, obj, link=None):
        if link is not None and not self._has(self.collision_pairs(), link, obj):
            raise ValueError(f"Collision pair ({link.name}, {obj.name}) not valid")
        sm, _ = self._scene_device()
        prox = self.collisions(q)
        return [prox[p] for p in self._pair_selection(sm, obj, link)]

    def proximity_jacobians(self, q):
        """Batched ``collisions`` + ``jacobian_proximity`` over every allowed primitive pair (additive):
        ``(..., dof)`` -> signed distances ``(B, P)``, witnesses ``(B, P, 9)`` (point on subject, point on target,
        normal target->subject) and rows ``(B, P, dof)`` with row = n . Jv_subject(p_s) - n . Jv_target(p_t)
        (arm.py:620-632), one launch."""
        sm, dev = self._scene_device()
        return dev.proximity_jacobian(q.reshape(-1, self.dof))

    def jacobian_proximity(self, q, obj, link=None):
        """As upstream (arm.py:620-632): one row per proximity of ``distance_to(q, obj, link)``, a 1-D row when
        there is exactly one."""
        if tuple(q.shape) != (self.dof,):
            raise ValueError(f"q must be a 1D array with {self.dof} elements")
        if link is not None and not self._has(self.collision_pairs(), link, obj):
            raise ValueError(f"Collision pair ({link.name}, {obj.name}) not valid")
        sm, dev = self._scene_device()
        sel = self._pair_selection(sm, obj, link)
        qn = q.detach().cpu().numpy() if _is_tensor(q) else np.asarray(q, dtype=np.float64)
        if not sel:
            return np.zeros((0, self.dof))
        _, _, rows = dev.proximity_jacobian(qn.reshape(1, -1))
        J = np.ascontiguousarray(rows[0][sel])
        if J.shape[0] == 1:
            return J[0]
        return J
