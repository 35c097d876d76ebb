"""Device engine: owns an ``nbk_model`` handle and launches the HIP kernels on torch-managed buffers.

PyTorch is plumbing only (device memory, the current HIP stream); every number is produced by
libnbk.so.  NumPy inputs are staged to the GPU and results are returned as NumPy; torch CUDA tensors
stay on the device.  Without a GPU every compute call raises ``NbkError`` -- there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import NbkError


def _torch():
    import torch
    return torch


def _require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise NbkError("no GPU visible: the numbotics_amd device path has no CPU fallback")
    return torch


class _Staged:
    """A float64 (B, n) device tensor plus how to hand results back."""

    def __init__(self, x, n_cols, what="q"):
        torch = _require_gpu()
        self.numpy = not torch.is_tensor(x)
        if self.numpy:
            a = np.ascontiguousarray(np.asarray(x, dtype=np.float64)).reshape(-1, n_cols)
            self.t = torch.from_numpy(a).to("cuda", non_blocking=False)
        else:
            if not x.is_cuda:
                x = x.to("cuda")
            self.t = x.to(torch.float64).contiguous().reshape(-1, n_cols)
        self.B = int(self.t.shape[0])
        self.device = self.t.device

    def out(self, t):
        return t.cpu().numpy() if self.numpy else t


def _host_f64(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(-1)
    if a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


class DeviceModel:
    """Immutable device descriptor built from a KinematicModel or SceneModel (robots/model.py)."""

    def __init__(self, model):
        _require_gpu()
        lib = _lib.load()
        kin = getattr(model, "kin", model)
        scene = model if hasattr(model, "kin") else None
        self.kin, self.scene = kin, scene
        keep = []

        def ptr(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data
        d = _lib.ModelDesc()
        d.n_q, d.n_joints = kin.n_q, kin.n_joints
        d.joint_parent = ptr(kin.joint_parent, np.int32)
        d.joint_type = ptr(kin.joint_type, np.int32)
        d.joint_qidx = ptr(kin.joint_qidx, np.int32)
        d.joint_rot = ptr(kin.joint_rot, np.float64)
        d.joint_trans = ptr(kin.joint_trans, np.float64)
        d.joint_slide = ptr(kin.joint_slide, np.float64)
        d.joint_axis = ptr(kin.joint_axis, np.float64)
        d.base_pose = ptr(kin.base_pose, np.float64)
        if scene is not None:
            d.n_rshapes, d.n_wshapes, d.n_pairs = scene.n_rshapes, scene.n_wshapes, scene.n_pairs
            d.rshape_frame = ptr(scene.rshape_frame, np.int32)
            d.rshape_type = ptr(scene.rshape_type, np.int32)
            d.rshape_local = ptr(scene.rshape_local, np.float64)
            d.rshape_param = ptr(scene.rshape_param, np.float64)
            d.wshape_type = ptr(scene.wshape_type, np.int32)
            d.wshape_pose = ptr(scene.wshape_pose, np.float64)
            d.wshape_param = ptr(scene.wshape_param, np.float64)
            d.pair_a = ptr(scene.pair_a, np.int32)
            d.pair_b = ptr(scene.pair_b, np.int32)
            d.n_hulls = scene.n_hulls
            d.hull_vert_begin = ptr(scene.hull_vert_begin, np.int32)
            d.hull_verts = ptr(scene.hull_verts, np.float64)
            d.hull_face_begin = ptr(scene.hull_face_begin, np.int32)
            d.hull_planes = ptr(scene.hull_planes, np.float64)
        h = C.c_void_p()
        _lib.check(lib.nbk_model_create(C.byref(d), C.byref(h)), "nbk_model_create")
        self._h = h
        self._lib = lib
        self.n_q = kin.n_q
        self.n_pairs = scene.n_pairs if scene is not None else 0

    def __del__(self):
        h = getattr(self, "_h", None)
        for fs in getattr(self, "_framesets", {}).values():
            try:
                self._lib.nbk_frameset_destroy(fs)
            except Exception:
                pass
        if h:
            try:
                self._lib.nbk_model_destroy(h)
            except Exception:
                pass
            self._h = None

    @staticmethod
    def _stream():
        return C.c_void_p(_torch().cuda.current_stream().cuda_stream)

    # ---- kinematics ----------------------------------------------------------------------------
    def _frame_args(self, frame, extra_local):
        fr = self.kin.frames[frame]
        local = fr.local if extra_local is None else fr.local @ extra_local
        path = np.ascontiguousarray(fr.path, dtype=np.int32)
        return path, np.ascontiguousarray(local[:3, :4], dtype=np.float64).reshape(12)

    def fk(self, q, frame, extra_local=None, local_pose=None):
        torch = _require_gpu()
        qs = _Staged(q, self.n_q)
        path, local = self._frame_args(frame, extra_local)
        out = torch.empty((qs.B, 4, 4), dtype=torch.float64, device=qs.device)
        lp = None
        if local_pose is not None:
            lp = _Staged(local_pose, 16, "local_pose")
            if lp.B != qs.B:
                raise ValueError("local_pose must have one 4x4 per configuration")
        _lib.check(self._lib.nbk_fk_batch(self._h, qs.t.data_ptr(), qs.B, path.ctypes.data, len(path),
                                          local.ctypes.data, None if lp is None else lp.t.data_ptr(),
                                          out.data_ptr(), self._stream()), "nbk_fk_batch")
        return qs.out(out)

    def fk_frames(self, q, frames, extra_locals=None):
        """Poses of several frames per configuration in one sweep: (B, len(frames), 4, 4), bit-identical to ``fk`` per frame.
        ``extra_locals``: optional {frame: 4x4} right factors (e.g. COM offsets)."""
        torch = _require_gpu()
        qs = _Staged(q, self.n_q)
        frames = list(frames)
        ex = extra_locals or {}
        key = (tuple(frames), tuple(sorted((k, np.asarray(v, dtype=np.float64).tobytes()) for k, v in ex.items())))
        cache = self.__dict__.setdefault("_framesets", {})
        fs = cache.get(key)
        if fs is None:
            joints = np.empty((len(frames),), dtype=np.int32)
            locs = np.empty((len(frames), 12), dtype=np.float64)
            for i, f in enumerate(frames):
                fr = self.kin.frames[f]
                local = fr.local if f not in ex else fr.local @ np.asarray(ex[f], dtype=np.float64)
                joints[i] = fr.joint
                locs[i] = np.ascontiguousarray(local[:3, :4]).reshape(12)
            h = C.c_void_p()
            _lib.check(self._lib.nbk_frameset_create(self._h, len(frames), joints.ctypes.data, locs.ctypes.data, C.byref(h)),
                       "nbk_frameset_create")
            fs = cache[key] = h
        out = torch.empty((qs.B, len(frames), 4, 4), dtype=torch.float64, device=qs.device)
        _lib.check(self._lib.nbk_fk_frames_batch(self._h, fs, qs.t.data_ptr(), qs.B, out.data_ptr(), self._stream()),
                   "nbk_fk_frames_batch")
        return qs.out(out)

    def jacobian(self, q, frame, extra_local=None, local_pose=None, global_pose=None):
        torch = _require_gpu()
        qs = _Staged(q, self.n_q)
        path, local = self._frame_args(frame, extra_local)
        out = torch.empty((qs.B, 6, self.n_q), dtype=torch.float64, device=qs.device)
        mode, pose = 0, None
        if local_pose is not None:
            mode, pose = 1, _Staged(local_pose, 16)
        elif global_pose is not None:
            mode, pose = 2, _Staged(global_pose, 16)
        if pose is not None and pose.B != qs.B:
            raise ValueError("pose must have one 4x4 per configuration")
        _lib.check(self._lib.nbk_jacobian_batch(self._h, qs.t.data_ptr(), qs.B, path.ctypes.data, len(path),
                                                local.ctypes.data, mode, None if pose is None else pose.t.data_ptr(),
                                                out.data_ptr(), self._stream()), "nbk_jacobian_batch")
        return qs.out(out)

    def ik(self, pose, q0, frame, extra_local=None, limits=None, tol=1e-6, max_iter=100, max_failures=15):
        """B damped-least-squares IK problems of one frame -> (success (B,) bool, q (B,n_q), |diff| (B,), steps (B,))."""
        torch = _require_gpu()
        qs = _Staged(q0, self.n_q)
        ps = _Staged(pose, 16, "pose")
        if ps.B != qs.B:
            raise ValueError("pose and q0 must have the same number of rows")
        path, local = self._frame_args(frame, extra_local)
        lim = None if limits is None else _host_f64(limits, 2 * self.n_q)
        q = torch.empty((qs.B, self.n_q), dtype=torch.float64, device=qs.device)
        ok = torch.empty((qs.B,), dtype=torch.uint8, device=qs.device)
        nrm = torch.empty((qs.B,), dtype=torch.float64, device=qs.device)
        it = torch.empty((qs.B,), dtype=torch.int32, device=qs.device)
        _lib.check(self._lib.nbk_ik_batch(self._h, ps.t.data_ptr(), qs.t.data_ptr(), qs.B, path.ctypes.data, len(path),
                                          local.ctypes.data, None if lim is None else lim.ctypes.data, float(tol),
                                          int(max_iter), int(max_failures), q.data_ptr(), ok.data_ptr(), nrm.data_ptr(),
                                          it.data_ptr(), self._stream()), "nbk_ik_batch")
        return qs.out(ok.bool()), qs.out(q), qs.out(nrm), qs.out(it)

    # ---- collision -----------------------------------------------------------------------------
    def validity_workspace_bytes(self, B: int) -> int:
        return int(self._lib.nbk_validity_workspace_bytes(self._h, int(B)))

    def validity(self, q, threshold=0.0, packed=False, workspace=None):
        """In-collision flags.  packed=False: (B,) bool; packed=True: (ceil(B/64),) int64 words
        (bit b%64 of word b//64), the form the multi-GPU all-gather moves.
        `workspace`: optional torch uint8 CUDA tensor of at least validity_workspace_bytes(B) bytes
        (caller-owned scratch: concurrent streams / graph capture); default = the descriptor's own."""
        torch = _require_gpu()
        qs = _Staged(q, self.n_q)
        words = mask = None
        if packed:
            words = torch.empty(((qs.B + 63) // 64,), dtype=torch.int64, device=qs.device)   # every word is written
        else:
            mask = torch.empty((qs.B,), dtype=torch.uint8, device=qs.device)
        wp = None if words is None else words.data_ptr()
        mp = None if mask is None else mask.data_ptr()
        if workspace is None:
            _lib.check(self._lib.nbk_validity_batch(self._h, qs.t.data_ptr(), qs.B, float(threshold), wp, mp,
                                                    self._stream()), "nbk_validity_batch")
        else:
            _lib.check(self._lib.nbk_validity_batch_ws(self._h, qs.t.data_ptr(), qs.B, float(threshold), wp, mp,
                                                       workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                                       self._stream()), "nbk_validity_batch_ws")
        return qs.out(words) if packed else qs.out(mask.bool())

    NARROW_BUILDS = ("k_narrow_bool", "k_narrow_pos", "k_narrow_pred", "k_narrow")

    def narrow_build(self, threshold=0.0) -> str:
        """Diagnostic: the narrowphase build ``validity`` launches for this scene at this threshold (bench.py reports it)."""
        return self.NARROW_BUILDS[int(self._lib.nbk_debug_narrow_variant(self._h, float(threshold)))]

    def validity_scalar(self, q, threshold=0.0) -> bool:
        """One configuration from host memory (the reference's scalar ``in_collision(q)``): pinned, device-mapped staging
        inside the library, one wait -- no torch tensors on the way."""
        a = np.ascontiguousarray(q, dtype=np.float64).reshape(-1)
        if a.size != self.n_q:
            raise ValueError(f"expected {self.n_q} values, got {a.size}")
        out = C.c_int32(0)
        _lib.check(self._lib.nbk_validity_scalar_host(self._h, a.ctypes.data, float(threshold), C.byref(out)),
                   "nbk_validity_scalar_host")
        return bool(out.value)

    def edge_validity_scalar(self, start, goal, resolution, max_distance, mode="connect", threshold=0.0, dist=None):
        """One edge from host memory -> (valid, end state (n_q,), samples)."""
        s = np.ascontiguousarray(start, dtype=np.float64).reshape(-1)
        g = np.ascontiguousarray(goal, dtype=np.float64).reshape(-1)
        if s.size != self.n_q or g.size != self.n_q:
            raise ValueError(f"expected {self.n_q} values per end point")
        end = np.empty((self.n_q,), dtype=np.float64)
        ok, ns = C.c_int32(0), C.c_int32(0)
        _lib.check(self._lib.nbk_edge_validity_scalar_host(
            self._h, s.ctypes.data, g.ctypes.data, -1.0 if dist is None else float(dist), float(resolution), float(max_distance),
            0 if mode == "connect" else 1, float(threshold), C.byref(ok), end.ctypes.data, C.byref(ns)),
            "nbk_edge_validity_scalar_host")
        return bool(ok.value), end, int(ns.value)

    def closest(self, q):
        torch = _require_gpu()
        qs = _Staged(q, self.n_q)
        d = torch.empty((qs.B,), dtype=torch.float64, device=qs.device)
        idx = torch.empty((qs.B,), dtype=torch.int32, device=qs.device)
        _lib.check(self._lib.nbk_closest_batch(self._h, qs.t.data_ptr(), qs.B, d.data_ptr(), idx.data_ptr(),
                                               self._stream()), "nbk_closest_batch")
        return qs.out(d), qs.out(idx)

    def pair_distances(self, q, witness=False):
        torch = _require_gpu()
        qs = _Staged(q, self.n_q)
        d = torch.empty((qs.B, self.n_pairs), dtype=torch.float64, device=qs.device)
        w = torch.empty((qs.B, self.n_pairs, 9), dtype=torch.float64, device=qs.device) if witness else None
        _lib.check(self._lib.nbk_pair_distances_batch(self._h, qs.t.data_ptr(), qs.B, d.data_ptr(),
                                                      None if w is None else w.data_ptr(), self._stream()),
                   "nbk_pair_distances_batch")
        return (qs.out(d), qs.out(w)) if witness else qs.out(d)

    def proximity_jacobian(self, q):
        """(B,P) signed distances, (B,P,9) witnesses and (B,P,n_q) proximity-Jacobian rows of every allowed pair."""
        torch = _require_gpu()
        qs = _Staged(q, self.n_q)
        d = torch.empty((qs.B, self.n_pairs), dtype=torch.float64, device=qs.device)
        w = torch.empty((qs.B, self.n_pairs, 9), dtype=torch.float64, device=qs.device)
        j = torch.empty((qs.B, self.n_pairs, self.n_q), dtype=torch.float64, device=qs.device)
        _lib.check(self._lib.nbk_proximity_jacobian_batch(self._h, qs.t.data_ptr(), qs.B, d.data_ptr(), w.data_ptr(),
                                                          j.data_ptr(), self._stream()), "nbk_proximity_jacobian_batch")
        return qs.out(d), qs.out(w), qs.out(j)

    def edge_validity(self, starts, goals, resolution, max_distance, mode="connect", threshold=0.0, dist=None):
        torch = _require_gpu()
        s = _Staged(starts, self.n_q)
        g = _Staged(goals, self.n_q)
        if s.B != g.B:
            raise ValueError("starts and goals must have the same number of rows")
        dd = None
        if dist is not None:
            dd = _Staged(dist, 1)
            if dd.B != s.B:
                raise ValueError("dist must have one value per edge")
        valid = torch.empty((s.B,), dtype=torch.uint8, device=s.device)
        end = torch.empty((s.B, self.n_q), dtype=torch.float64, device=s.device)
        ns = torch.empty((s.B,), dtype=torch.int32, device=s.device)
        _lib.check(self._lib.nbk_edge_validity_batch(
            self._h, s.t.data_ptr(), g.t.data_ptr(), None if dd is None else dd.t.data_ptr(), s.B,
            float(resolution), float(max_distance), 0 if mode == "connect" else 1, float(threshold),
            valid.data_ptr(), end.data_ptr(), ns.data_ptr(), self._stream()), "nbk_edge_validity_batch")
        return s.out(valid.bool()), s.out(end), s.out(ns)


def selftest_math(a, b):
    """sincos(a), sqrt(a), a/b as the kernels compute them (arithmetic-contract check)."""
    torch = _require_gpu()
    lib = _lib.load()
    ta = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    tb = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float64)).cuda()
    outs = [torch.empty_like(ta) for _ in range(4)]
    _lib.check(lib.nbk_selftest_math(ta.data_ptr(), tb.data_ptr(), ta.numel(), *[o.data_ptr() for o in outs],
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nbk_selftest_math")
    return [o.cpu().numpy() for o in outs]


def knn_prefix_device(points, k):
    """(N, d) float32 -> (N, k) int64 neighbour lists among the points inserted before (nbk_knn_prefix), -1 padded."""
    torch = _require_gpu()
    lib = _lib.load()
    x = np.ascontiguousarray(points, dtype=np.float32)
    n, dim = x.shape
    t = torch.from_numpy(x).cuda()
    out = torch.empty((n, k), dtype=torch.int32, device="cuda")
    _lib.check(lib.nbk_knn_prefix(t.data_ptr(), n, dim, int(k), out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "nbk_knn_prefix")
    return out.cpu().numpy().astype(np.int64)
