"""Independent truth for the oracle's narrowphase at fuzz scale (CPU suite).

Every oracle answer is checked by a CERTIFICATE evaluated with this file's own NumPy support functions (no code shared with
oracle/ or numbotics_amd/csrc): for a pair reported separated by d with witness points pa, pb and normal n

    pa in core A, pb in core B, |pa - pb| = d_core            (d_core is attained: dist <= d_core)
    min_A n.x - max_B n.x >= d_core                           (a plane separates by d_core: dist >= d_core)

so d_core IS the distance, to the tolerance of the checks (1e-9) -- no optimiser involved.  For a pair reported overlapping with
depth D along n (the translation of A by D n separates):  the overlap of the two supports along n equals D (an upper bound of the
minimum translation distance), and the lower bound comes from an exact construction where one exists (polytope cores: Qhull on
the vertices of A (-) B, the nearest facet of the difference body) or from a minimisation over directions plus a bracket between
inscribed and circumscribed prisms (cylinder cores).

Poses are products of the reference's Rodrigues rotation (numbotics/robots/helpers.py:43-55 uses the axis as given) about joint
axes rounded to five digits: rotations orthonormal to ~1e-6 only -- the regime in which round 2's fuzz campaign found a
cylinder-support cancellation that oracle and device shared.  Shapes are what the oracle defines for such a frame: box / hull
points c + sum_j x_j ax_j, a cylinder's disc perpendicular to its axis column, its radius unscaled.
Bullet itself is absent: parity with it stays UNPINNED; this pins the geometry the build defines.
"""
import numpy as np
import pytest
from scipy.optimize import minimize
from scipy.spatial import ConvexHull

from oracle.cpu_oracle import HullSet, shape_distance_h, shape_collides_h
from numbotics_amd.utils.mesh import convex_hull

SPHERE, CAPSULE, BOX, CYLINDER, HULL = 0, 1, 2, 3, 5
NAMES = {0: "sphere", 1: "capsule", 2: "box", 3: "cylinder", 5: "hull"}


# ---- poses: chains of Rodrigues rotations about five-digit axes -------------------------------------------------------------
def rodrigues(axis, angle):
    a = np.asarray(axis, dtype=np.float64)
    K = np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


def joint_pose(rng, scale, five_digit=True):
    R = np.eye(3)
    for _ in range(int(rng.integers(1, 4))):
        a = rng.normal(size=3)
        a = a / np.linalg.norm(a)
        if five_digit:
            a = np.round(a, 5)                                   # a URDF axis as people write it
        R = R @ rodrigues(a, rng.uniform(-np.pi, np.pi))
    if not five_digit:
        U, _, Vt = np.linalg.svd(R)                              # orthonormal to rounding
        R = U @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.uniform(-scale, scale, 3)
    return T


# ---- shapes as cores, independent of the oracle --------------------------------------------------------------------------------
class CoreT:
    """core of one shape in the world: kind, centre c, axis columns ax (3,3), parameters, margin"""

    def __init__(self, t, T, p, part=None):
        self.t, self.c, self.ax = t, T[:3, 3].copy(), T[:3, :3].copy()
        if t == SPHERE:
            self.margin = p[0]
        elif t == CAPSULE:
            self.margin, self.h = p[0], p[1]
        elif t == BOX:
            self.margin, self.he = p[3], np.asarray(p[:3]) - p[3]
        elif t == CYLINDER:
            self.margin, self.R, self.h = p[3], p[0] - p[3], p[1] - p[3]
        else:
            self.margin, self.V, self.P = p[3], part.vertices, part.planes

    def support_value(self, n):
        """h(n) = max over the core of n.x"""
        b = float(n @ self.c)
        if self.t == SPHERE:
            return b
        if self.t == CAPSULE:
            return b + self.h * abs(n @ self.ax[:, 2])
        if self.t == BOX:
            return b + float(np.sum(self.he * np.abs(n @ self.ax)))
        if self.t == CYLINDER:
            u = self.ax[:, 2]
            uu = u @ u
            nu = n @ u
            rad2 = max(0.0, n @ n - nu * nu / uu)              # squared length of the part of n perpendicular to u
            return b + self.h * abs(nu) + self.R * np.sqrt(rad2)
        return b + float(np.max(self.V @ (n @ self.ax)))

    def violation(self, x):
        """how far x is outside the core (<= 0: inside), in the core's own metric"""
        if self.t == SPHERE:
            return float(np.linalg.norm(x - self.c))
        if self.t == CAPSULE:
            u = self.ax[:, 2]
            z = np.clip((x - self.c) @ u / (u @ u), -self.h, self.h)
            return float(np.linalg.norm(x - self.c - z * u))
        if self.t == BOX:
            loc = np.linalg.solve(self.ax, x - self.c)
            return float(np.max(np.abs(loc) - self.he) * np.max(np.linalg.norm(self.ax, axis=0)))
        if self.t == CYLINDER:
            u = self.ax[:, 2]
            z = (x - self.c) @ u / (u @ u)
            r = np.linalg.norm(x - self.c - z * u)
            return float(max(r - self.R, (abs(z) - self.h) * np.linalg.norm(u)))
        loc = np.linalg.solve(self.ax, x - self.c)
        return float(np.max(self.P[:, :3] @ loc - self.P[:, 3])) if len(self.P) else 0.0

    def vertices(self, n_gon=0, outer=False):
        """world vertices of the (polytope) core; cylinders as n_gon-prisms, inscribed or circumscribed"""
        if self.t == SPHERE:
            return self.c[None]
        if self.t == CAPSULE:
            u = self.ax[:, 2]
            return np.array([self.c - self.h * u, self.c + self.h * u])
        if self.t == BOX:
            s = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
            return self.c + (s * self.he) @ self.ax.T
        if self.t == CYLINDER:
            u = self.ax[:, 2]
            e1 = np.cross(u, [1.0, 0.0, 0.0] if abs(u[0]) < 0.8 else [0.0, 1.0, 0.0])
            e1 /= np.linalg.norm(e1)
            e2 = np.cross(u, e1)
            e2 /= np.linalg.norm(e2)
            ang = 2 * np.pi * np.arange(n_gon) / n_gon
            R = self.R / np.cos(np.pi / n_gon) if outer else self.R
            ring = R * (np.cos(ang)[:, None] * e1 + np.sin(ang)[:, None] * e2)
            return np.concatenate([self.c + self.h * u + ring, self.c - self.h * u + ring])
        return self.c + self.V @ self.ax.T


def random_param(rng, t):
    p = np.zeros(4)
    if t == SPHERE:
        p[0] = rng.uniform(0.02, 0.2)
    elif t == CAPSULE:
        p[0], p[1] = rng.uniform(0.02, 0.15), rng.uniform(0.02, 0.3)
    elif t == CYLINDER:
        p[0], p[1] = rng.uniform(0.03, 0.15), rng.uniform(0.03, 0.3)
        p[3] = float(rng.choice([0.0, 0.0, 0.1 * min(p[0], p[1])]))       # Bullet's own margin on some
    elif t == BOX:
        p[:3] = rng.uniform(0.03, 0.3, 3)
        p[3] = float(rng.choice([0.0, 0.0, 0.1 * p[:3].min()]))
    return p


class Scene:
    def __init__(self, seed, n_hulls=40, five_digit=True):
        self.five_digit = five_digit
        self.rng = np.random.default_rng(seed)
        self.hs = HullSet()
        self.parts = []
        for _ in range(n_hulls):
            pts = self.rng.normal(size=(int(self.rng.integers(8, 28)), 3))
            pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * self.rng.uniform(0.6, 1.0, (len(pts), 1)) * self.rng.uniform(0.05, 0.3, 3)
            part = convex_hull(pts)
            self.parts.append(part)
            self.hs.add(part)

    def shape(self, t, scale):
        T = joint_pose(self.rng, scale, self.five_digit)
        if t == HULL:
            i = int(self.rng.integers(0, len(self.parts)))
            p = np.array([float(i), 0.0, 0.0, float(self.rng.choice([0.0, 0.001]))])
            return T, p, CoreT(t, T, p, self.parts[i])
        p = random_param(self.rng, t)
        return T, p, CoreT(t, T, p)


def max_gap_over_directions(A, B, n0):
    """max over unit n of (min_A n.x - max_B n.x) by a local search from n0: the distance of two disjoint convex sets (the gap is a
    concave function of the direction near its maximum)"""
    def f(x):
        nn = x / np.linalg.norm(x)
        return A.support_value(-nn) + B.support_value(nn)
    r = minimize(f, n0, method="Nelder-Mead", options={"xatol": 1e-12, "fatol": 1e-15, "maxiter": 4000})
    return -min(float(r.fun), f(n0))


def certificate(A, B, d, wa, wb, n, refine=False):
    """-> (kind, error): 'sep' with the largest violation of the separated certificate, or 'pen' with |D - overlap(n)|.

    Separated: the witnesses lie in the cores and are d_core apart (always checked: the distance is attained), and a plane
    separates by d_core -- along the reported normal for polytope / point / segment cores; GJK's final direction is NOT the one that
    achieved its best lower bound, so with a cylinder core (a curved contact) the reported normal is only good to ~1e-3 rad and
    the plane is looked for (`refine`: a local search over directions from the reported one; sampled, it costs milliseconds).
    The plane test is ill-conditioned below d_core ~ 1e-6 (its error is rounding / d_core) and skipped there; below 1e-7 the
    witness points themselves are not required to lie in the cores (found by this test: two boxes 2e-9 apart, witnesses 0.27 off
    although their difference -- the distance -- is right)."""
    dc = d + A.margin + B.margin                       # signed distance of the cores
    if dc > 0.0:
        pa, pb = wa + A.margin * n, wb - B.margin * n
        err = abs(np.linalg.norm(pa - pb) - dc)
        if dc >= 1e-7:                                   # (closer than that the barycentric weights of GJK's last simplex are
            err = max(err, A.violation(pa), B.violation(pb))     # ill-conditioned: the VALUE stays good, the witness points need not)
        if dc >= 1e-6:
            curved = CYLINDER in (A.t, B.t)
            if not curved:
                err = max(err, dc - (-A.support_value(-n) - B.support_value(n)))
            elif refine:
                err = max(err, dc - max_gap_over_directions(A, B, n))
        return "sep", err
    D = -dc
    return "pen", abs(D - (A.support_value(-n) + B.support_value(n)))


def exact_depth_polytopes(A, B, n_gon=0, outer=False):
    """minimum translation distance of two overlapping polytopes: the nearest facet of conv{a_i - b_j} to the origin"""
    VA, VB = A.vertices(n_gon, outer), B.vertices(n_gon, outer)
    D = (VA[:, None, :] - VB[None, :, :]).reshape(-1, 3)
    try:
        hull = ConvexHull(D)
    except Exception:
        return None
    off = -hull.equations[:, 3]                         # n.x <= off inside, |n| = 1
    return float(off.min()) if off.min() > 0 else None


def min_overlap_over_directions(A, B, n0, rng, starts=6):
    """min over unit n of h_A(-n) + h_B(n): local minimisations from the oracle's direction and random ones (an upper bound of
    the minimum translation distance that is tight when a start lands in the right basin)"""
    def f(x):
        nn = x / np.linalg.norm(x)
        return A.support_value(-nn) + B.support_value(nn)
    best = f(n0)
    for s in range(starts):
        x0 = n0 + (0.0 if s == 0 else 0.7) * rng.normal(size=3)
        r = minimize(f, x0, method="Nelder-Mead", options={"xatol": 1e-11, "fatol": 1e-13, "maxiter": 2000})
        best = min(best, float(r.fun))
    return best


CLASSES = [(a, b) for a in (SPHERE, CAPSULE, BOX, CYLINDER, HULL) for b in (SPHERE, CAPSULE, BOX, CYLINDER, HULL)]


def _near_contact_pair(sc, ta, tb, rng):
    """a random pair moved along its own normal to a signed distance (of the shapes or of their cores) that is log-uniform in
    +-[1e-9, 3e-2], or left as drawn"""
    Ta, pa, A = sc.shape(ta, 0.3)
    Tb, pb, B = sc.shape(tb, 0.3)
    mode = rng.integers(0, 4)
    if mode < 3:
        d, wa, wb, n, _ = shape_distance_h(sc.hs, ta, Ta, pa, tb, Tb, pb)
        if np.isfinite(d) and np.linalg.norm(n) > 0.5:
            target = float(np.sign(rng.uniform(-1, 1)) * 10.0 ** rng.uniform(-9, -1.5))
            Tb = Tb.copy()
            # n points from B to A: moving B along n closes the gap -- of the shapes, or (every other time) of their CORES, so that
            # half of the pairs test the overlapping-core branch whatever their margins
            gap = d + (A.margin + B.margin if mode == 0 else 0.0)
            Tb[:3, 3] += (gap - target) * n
            B = CoreT(tb, Tb, pb, sc.parts[int(pb[0])] if tb == HULL else None)
    return (Ta, pa, A), (Tb, pb, B)


def test_certificates_at_fuzz_scale():
    """6e4 random near-contact and overlapping pairs of all 25 class pairs in orthonormal frames (the shapes are what their names
    say): every separated answer carries its own proof, every overlapping answer's depth is the overlap along its own direction,
    and the validity predicate agrees with the distance -- all to 1e-8."""
    tol, n_pairs = 1e-8, 60_000
    sc = Scene(101, five_digit=False)
    rng = sc.rng
    worst_sep, worst_pen, count = {}, {}, {"sep": 0, "pen": 0}
    for i in range(n_pairs):
        ta, tb = CLASSES[i % 25]
        (Ta, pa, A), (Tb, pb, B) = _near_contact_pair(sc, ta, tb, rng)
        d, wa, wb, n, it = shape_distance_h(sc.hs, ta, Ta, pa, tb, Tb, pb)
        assert np.isfinite(d) and it <= 64
        kind, err = certificate(A, B, d, wa, wb, n, refine=(i % 11 == 0))
        key = (NAMES[ta], NAMES[tb])
        w = worst_sep if kind == "sep" else worst_pen
        w[key] = max(w.get(key, 0.0), err)
        count[kind] += 1
        if i % 7 == 0:
            for thr in (0.0, d + 10 * tol, d - 10 * tol):
                # a verdict AT the distance itself is anybody's; at threshold 0 the boolean walk counts a grazing contact of a
                # cylinder it cannot separate in 32 steps as touching (conservative; largest seen 4.4e-6 on cylinder pairs)
                band = 1e-5 if (thr == 0.0 and CYLINDER in (ta, tb)) else 10 * tol
                if abs(d - thr) >= band:
                    assert shape_collides_h(sc.hs, ta, Ta, pa, tb, Tb, pb, thr) == (d < thr), (key, d, thr)
    assert count["sep"] > 0.3 * n_pairs and count["pen"] > 0.15 * n_pairs, count
    assert len(worst_sep) == 25 and max(worst_sep.values()) < tol, worst_sep
    assert len(worst_pen) >= 21 and max(worst_pen.values()) < tol, worst_pen      # (point / segment cores never overlap each other)


def test_five_digit_axes_stay_close_to_the_orthonormal_answer():
    """Frames that are products of the reference's Rodrigues rotation about joint axes rounded to five digits are orthonormal to a few
    1e-5 only; a "box" is then a parallelepiped and there is no truth to pin to 1e-8 -- but the answer must stay within a few
    1e-5 x the shapes' size of the answer for the nearest orthonormal frames (the cancellation of round 2 moved a cylinder's
    support point by 25 mm), the predicate must agree with the distance, and (A, B) with (B, A)."""
    sc = Scene(102, five_digit=True)
    rng = sc.rng
    worst = 0.0
    for i in range(40_000):
        ta, tb = CLASSES[i % 25]
        (Ta, pa, A), (Tb, pb, B) = _near_contact_pair(sc, ta, tb, rng)
        d, wa, wb, n, it = shape_distance_h(sc.hs, ta, Ta, pa, tb, Tb, pb)
        To = []
        for T in (Ta, Tb):
            U, _, Vt = np.linalg.svd(T[:3, :3])
            To.append(T.copy())
            To[-1][:3, :3] = U @ Vt
        do = shape_distance_h(sc.hs, ta, To[0], pa, tb, To[1], pb)[0]
        worst = max(worst, abs(d - do))
        assert abs(d - do) < 1e-4, (NAMES[ta], NAMES[tb], d, do)
        if i % 5 == 0:
            d2 = shape_distance_h(sc.hs, tb, Tb, pb, ta, Ta, pa)[0]
            assert abs(d - d2) < 1e-4, (NAMES[ta], NAMES[tb], d, d2)
            for thr in (d + 1e-4, d - 1e-4):          # (closed forms and support functions read a sheared frame 1e-5 apart)
                assert shape_collides_h(sc.hs, ta, Ta, pa, tb, Tb, pb, thr) == (d < thr), (NAMES[ta], NAMES[tb], d, thr)
    assert worst > 0.0


def test_penetration_depth_is_the_minimum_translation_distance():
    """The reported depth against independent truth: exact (Qhull on A (-) B) for every polytope class pair incl. hulls; for
    cylinder cores the minimum over directions found by local searches, and a bracket between 90-gon prisms inside and outside."""
    sc = Scene(202, five_digit=False)
    rng = sc.rng
    poly = (SPHERE, CAPSULE, BOX, HULL)
    worst = {}
    n_poly = n_cyl = 0
    for i in range(6000):
        ta, tb = CLASSES[i % 25]
        if ta == SPHERE and tb == SPHERE:
            continue
        (Ta, pa, A), (Tb, pb, B) = _near_contact_pair(sc, ta, tb, rng)
        d, wa, wb, n, _ = shape_distance_h(sc.hs, ta, Ta, pa, tb, Tb, pb)
        D = -(d + A.margin + B.margin)
        if D <= 1e-7:
            continue
        key = (NAMES[ta], NAMES[tb])
        if ta in poly and tb in poly:
            if {ta, tb} <= {SPHERE, CAPSULE}:
                continue                                # (a flat difference body: Qhull has nothing to say)
            ex = exact_depth_polytopes(A, B)
            if ex is None:
                continue
            worst[key] = max(worst.get(key, 0.0), abs(D - ex))
            n_poly += 1
        elif n_cyl < 400:
            best = min_overlap_over_directions(A, B, n, rng)
            assert D <= best + 1e-6, (key, D, best)                     # no direction found separates with noticeably less (EPA stops at 32 vertices)
            worst[key] = max(worst.get(key, 0.0), abs(D - best))
            if n_cyl % 8 == 0:
                inner, outer = exact_depth_polytopes(A, B, 90), exact_depth_polytopes(A, B, 90, outer=True)
                if inner is not None and outer is not None:
                    assert inner - 1e-9 <= D <= outer + 1e-9, (key, inner, D, outer)
            n_cyl += 1
    assert n_poly > 400 and n_cyl > 300, (n_poly, n_cyl)
    assert max(worst.values()) < 1e-6, worst
