"""``unit_bspline`` (reference: numbotics/planning/trajectories.py:6-22).

Clamped uniform B-spline on [0, 1] with knots ``zeros(k) ++ linspace(0, 1, n-k+1) ++ ones(k)``.
Upstream returns a ``scipy.interpolate.BSpline``; this is a small de Boor evaluator with the same call
signature ``spline(t) -> point``.  For two control points and degree 1 (the connector default) it
evaluates ``(1-t)*c0 + t*c1`` with three separate roundings, which is bit-identical to what SciPy
produces for that case (tests/golden: g5_bspline_*), and is the form the device edge kernel uses.
"""
import numpy as np


class UnitBSpline:
    def __init__(self, knots, control_points, degree):
        self.t = np.asarray(knots, dtype=np.float64)
        self.c = np.asarray(control_points, dtype=np.float64)
        self.k = int(degree)

    def __call__(self, x):
        x = float(x)
        t, c, k = self.t, self.c, self.k
        n = c.shape[0]
        if k == 1 and n == 2:
            return (1.0 - x) * c[0] + x * c[1]
        # knot span: t[ell] <= x < t[ell+1], clamped to the last non-empty interval
        ell = int(np.searchsorted(t, x, side='right')) - 1
        ell = min(max(ell, k), n - 1)
        d = [c[j + ell - k].copy() for j in range(k + 1)]
        for r in range(1, k + 1):
            for j in range(k, r - 1, -1):
                den = t[j + 1 + ell - r] - t[j + ell - k]
                alpha = 0.0 if den == 0.0 else (x - t[j + ell - k]) / den
                d[j] = (1.0 - alpha) * d[j - 1] + alpha * d[j]
        return d[k]


def unit_bspline(control_points: np.ndarray, degree: int = 1):
    control_points = np.asarray(control_points)
    if control_points.ndim != 2:
        raise ValueError("control_points must be a 2D array (B x n)")
    B, _ = control_points.shape
    if degree >= B:
        raise ValueError("Degree must be less than the number of control points")
    knots = np.concatenate((np.zeros(degree), np.linspace(0, 1, B - degree + 1), np.ones(degree)))
    return UnitBSpline(knots, control_points, degree)
