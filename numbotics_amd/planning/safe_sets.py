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


def counter_example_bisection(arm, centre, points, num_bisections: int = 15, collision_tolerance: float = 1e-6):
    """For each colliding point q: bisect [centre, q] keeping the colliding end, return the final upper ends.

    Same arithmetic per sample as upstream: ``midpoint = (lo + hi) / 2.0``; a colliding midpoint replaces
    ``hi``, a free one replaces ``lo``."""
    points = np.asarray(points, dtype=np.float64)
    lo = np.tile(np.asarray(centre, dtype=np.float64)[None], (points.shape[0], 1))
    hi = points.copy()
    for _ in range(num_bisections):
        mid = (lo + hi) / 2.0
        hit = np.asarray(arm.in_collision(mid, collision_tolerance))
        hi[hit] = mid[hit]
        lo[~hit] = mid[~hit]
    return hi


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
