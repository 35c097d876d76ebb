"""Independent ground truth for convex-shape distances (test helper).

Solves  min |x - y|^2  s.t. x in coreA, y in coreB  with SciPy SLSQP in each shape's local
coordinates, then subtracts the margins.  Shares no code with oracle/ or numbotics_amd/csrc.
Only meaningful when the cores are disjoint (positive core distance).
"""
import numpy as np
from scipy.optimize import minimize

SPHERE, CAPSULE, BOX, CYLINDER, PLANE, HULL = 0, 1, 2, 3, 4, 5


def random_pose(rng, scale=0.5):
    A = rng.normal(size=(3, 3))
    Q, R = np.linalg.qr(A)
    Q = Q * np.sign(np.diag(R))
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    T = np.eye(4)
    T[:3, :3] = Q
    T[:3, 3] = rng.uniform(-scale, scale, 3)
    return T


def random_param(rng, t):
    p = np.zeros(4)
    if t == SPHERE:
        p[0] = rng.uniform(0.02, 0.2)
    elif t in (CAPSULE, CYLINDER):
        p[0], p[1] = rng.uniform(0.02, 0.15), rng.uniform(0.02, 0.3)
    elif t == BOX:
        p[:3] = rng.uniform(0.02, 0.3, 3)
    return p


def _core(t, p):
    """(constraints builder, margin, x0) in local coordinates."""
    if t == SPHERE:
        return ("point", None), p[0]
    if t == CAPSULE:
        return ("seg", p[1]), p[0]
    if t == BOX:
        return ("box", p[:3] - p[3]), p[3]
    if t == CYLINDER:
        return ("cyl", (p[0] - p[3], p[1] - p[3])), p[3]
    if t == HULL:
        return ("hull", p[4]), p[3]          # p[4] = (vertices (m,3), planes (f,4)) of a ConvexPart, p[3] = margin
    raise ValueError(t)


def _bounds_cons(kind, data, off):
    """bounds for the 3 local coords starting at variable index `off`, plus inequality constraints."""
    cons = []
    if kind == "point":
        b = [(0, 0)] * 3
    elif kind == "seg":
        b = [(0, 0), (0, 0), (-data, data)]
    elif kind == "box":
        b = [(-data[0], data[0]), (-data[1], data[1]), (-data[2], data[2])]
    elif kind == "hull":
        V, P = data
        lo, hi = V.min(axis=0), V.max(axis=0)
        b = [(lo[0], hi[0]), (lo[1], hi[1]), (lo[2], hi[2])]
        for pl in P:               # the H-representation: n.x <= d for every face (independent of the vertex list GJK uses)
            def jac(z, o=off, pl=pl):
                g = np.zeros(len(z))
                g[o:o + 3] = -pl[:3]
                return g
            cons.append({'type': 'ineq', 'fun': lambda z, o=off, pl=pl: pl[3] - pl[:3] @ z[o:o + 3], 'jac': jac})
    else:
        R, h = data
        b = [(-R, R), (-R, R), (-h, h)]
        cons.append({'type': 'ineq', 'fun': lambda z, o=off, R=R: R * R - z[o] ** 2 - z[o + 1] ** 2,
                     'jac': lambda z, o=off: np.concatenate([np.zeros(o), [-2 * z[o], -2 * z[o + 1], 0.0],
                                                             np.zeros(len(z) - o - 3)])})
    return b, cons


def truth_distance(ta, Ta, pa, tb, Tb, pb, restarts=6, seed=0):
    """Signed distance for shapes whose cores are disjoint; returns (d, core_distance)."""
    (ka, da), ma = _core(ta, pa)
    (kb, db), mb = _core(tb, pb)
    ba, ca = _bounds_cons(ka, da, 0)
    bb, cb = _bounds_cons(kb, db, 3)
    Ra, ta_, Rb, tb_ = Ta[:3, :3], Ta[:3, 3], Tb[:3, :3], Tb[:3, 3]

    def f(z):
        e = (Ra @ z[:3] + ta_) - (Rb @ z[3:] + tb_)
        return float(e @ e)

    def g(z):
        e = (Ra @ z[:3] + ta_) - (Rb @ z[3:] + tb_)
        return np.concatenate([2 * Ra.T @ e, -2 * Rb.T @ e])
    rng = np.random.default_rng(seed)
    best = np.inf
    for r in range(restarts):
        z0 = np.array([rng.uniform(lo, hi) if hi > lo else lo for lo, hi in ba + bb]) * (0.5 if r else 0.0)
        res = minimize(f, z0, jac=g, bounds=ba + bb, constraints=ca + cb, method='SLSQP',
                       options={'ftol': 1e-16, 'maxiter': 500})
        if res.fun < best and all(c['fun'](res.x) > -1e-9 for c in ca + cb):
            best = res.fun
    dc = np.sqrt(max(best, 0.0))
    return dc - ma - mb, dc
