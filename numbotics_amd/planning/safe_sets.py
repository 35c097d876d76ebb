"""The data-parallel inner steps of the reference's IRIS solver on the device
(reference: numbotics/planning/safe_sets.py:124-134,164-201).

Only the collision-bound steps are provided; the convex programmes around them (hit-and-run sampling,
cvxpy/MOSEK ellipsoid, SLSQP) are out of scope.

* ``collision_mask(arm, points, tol)`` = ``[subject.in_collision(q, tol) for q in points]``, which upstream maps
  over a pool of cloned PyBullet worlds (safe_sets.py:186-191) -- here one launch.
* ``counter_example_bisection(arm, centre, points, num_bisections, tol)`` = ``counter_ex_search_bisection`` for
  every colliding sample at once (safe_sets.py:124-134): ``num_bisections`` launches on shrinking intervals
  instead of ``M * num_bisections`` single-configuration PyBullet queries.
* ``distance_and_gradient(arm, points, link, obj)`` = the constraint value and Jacobian that
  ``counter_ex_search_nlp`` hands to SLSQP (safe_sets.py:86-121: ``distance_to(x, link, obj)[0].distance`` and
  ``jacobian_proximity(x, link=link, obj=obj)``), for M points in one launch.
"""
import numpy as np


def collision_mask(arm, points, collision_tolerance: float = 1e-6):
    """(M, dof) -> (M,) bool, True where the configuration's closest pair is nearer than the tolerance."""
    return arm.in_collision(points, collision_tolerance)


def counter_example_bisection(arm, centre, points, num_bisections: int = 15, collision_tolerance: float = 1e-6, graph: bool = False):
    """For each colliding point q: bisect [centre, q] keeping the colliding end, return the final upper ends.

    Same arithmetic per sample as upstream: ``midpoint = (lo + hi) / 2.0``; a colliding midpoint replaces
    ``hi``, a free one replaces ``lo``.

    ``points`` as a torch CUDA tensor keeps every round on the device -- midpoints, the validity launch and the interval
    update (``torch.where`` on the returned mask) with no copy through the host -- and returns a CUDA tensor, bit-equal to the
    NumPy form.  ``graph=True`` replays one captured round (midpoint, validity kernels, update: one graph launch instead of six)
    from a graph kept on the Arm per (scene, M, tolerance); the first such call pays for the capture."""
    if type(points).__module__.startswith("torch") and points.is_cuda:
        return _bisection_on_device(arm, centre, points, num_bisections, collision_tolerance, graph)
    points = np.asarray(points, dtype=np.float64)
    lo = np.tile(np.asarray(centre, dtype=np.float64)[None], (points.shape[0], 1))
    hi = points.copy()
    for _ in range(num_bisections):
        mid = (lo + hi) / 2.0
        hit = np.asarray(arm.in_collision(mid, collision_tolerance))
        hi[hit] = mid[hit]
        lo[~hit] = mid[~hit]
    return hi


class _BisectionGraph:
    """One captured bisection round over persistent buffers (lo, hi, mid, mask) on a private stream, for one (scene, M, tolerance):
    capturing costs more than a whole 15-round search, so the graph is kept on the Arm and replayed by later calls."""

    def __init__(self, dev, M, nq, tol, device):
        import torch
        from numbotics_amd import _lib
        self.dev, self.M, self.tol = dev, M, float(tol)
        self.lo = torch.empty((M, nq), dtype=torch.float64, device=device)
        self.hi = torch.empty_like(self.lo)
        self.mid = torch.empty_like(self.lo)
        self.mask = torch.empty((M,), dtype=torch.uint8, device=device)
        self.stream = torch.cuda.Stream()
        self._check = _lib.check
        self.graph = None

    def round(self):
        import torch
        torch.add(self.lo, self.hi, out=self.mid)
        self.mid.div_(2.0)                                # (lo + hi) / 2.0, as upstream
        self._check(self.dev._lib.nbk_validity_batch(self.dev._h, self.mid.data_ptr(), self.M, self.tol, None, self.mask.data_ptr(),
                                                     self.dev._stream()), "nbk_validity_batch")
        hit = self.mask.bool().unsqueeze(1)
        torch.where(hit, self.mid, self.hi, out=self.hi)
        torch.where(hit, self.lo, self.mid, out=self.lo)

    def run(self, centre, pts, rounds):
        import torch
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.lo.copy_(centre.expand_as(self.lo))
            self.hi.copy_(pts)
            if self.graph is None:
                self.round()                              # allocates the stream's scratch: a capture cannot
                rounds -= 1
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.stream):     # records one round, executes nothing
                    self.round()
                self.graph = g
            for _ in range(rounds):
                self.graph.replay()
            out = self.hi.clone()
        torch.cuda.current_stream().wait_stream(self.stream)
        return out


def _bisection_on_device(arm, centre, points, num_bisections, tol, graph):
    import torch
    pts = points.to(torch.float64).contiguous()
    M, nq = pts.shape
    c = torch.as_tensor(np.asarray(centre, dtype=np.float64), device=pts.device).reshape(1, nq)
    if M == 0 or num_bisections <= 0:
        return pts.clone()
    _, dev = arm._scene_device()
    if graph:
        cache = arm.__dict__.setdefault("_bisection_graphs", {})
        key = (id(dev), M, float(tol), pts.device.index)
        if key not in cache:
            if len(cache) >= 8:
                cache.clear()
            cache[key] = _BisectionGraph(dev, M, nq, tol, pts.device)
        return cache[key].run(c, pts, num_bisections)
    state = _BisectionGraph(dev, M, nq, tol, pts.device)
    state.lo.copy_(c.expand(M, nq))
    state.hi.copy_(pts)
    for _ in range(num_bisections):
        state.round()
    return state.hi


def distance_and_gradient(arm, points, link, obj):
    """(M, dof) -> signed distance (M,) of the body pair (link, obj) and its gradient rows (M, dof).

    A body pair of compound links has several primitive pairs: the closest one is reported per point (for
    single-primitive links this is upstream's ``[0]``)."""
    sm, _ = arm._scene_device()
    if not arm._has(arm.collision_pairs(), link, obj):
        raise ValueError(f"Collision pair ({link.name}, {obj.name}) not valid")
    sel = arm._pair_selection(sm, obj, link)
    points = np.asarray(points, dtype=np.float64).reshape(-1, arm.dof)
    dist, _, rows = arm.proximity_jacobians(points)
    dist, rows = np.asarray(dist)[:, sel], np.asarray(rows)[:, sel]
    k = np.argmin(dist, axis=1)
    i = np.arange(points.shape[0])
    return dist[i, k], rows[i, k]
