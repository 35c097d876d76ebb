"""The oracle's narrowphase against an independent ground truth (SciPy SLSQP on the convex programme
min |x - y|, tests/geom_truth.py) and its own internal consistency.  Bullet parity is UNPINNED (pybullet
is absent and the reference pins no collision value); this pins the GEOMETRY the build defines."""
import warnings

import numpy as np
import pytest

from oracle.cpu_oracle import shape_distance, shape_collides
from geom_truth import SPHERE, CAPSULE, BOX, CYLINDER, PLANE, random_pose, random_param, truth_distance

NAMES = {0: "sphere", 1: "capsule", 2: "box", 3: "cylinder"}


def test_separated_distances_match_truth():
    rng = np.random.default_rng(5)
    worst = {}
    n_sep = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(700):
            ta, tb = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            Ta, Tb = random_pose(rng, 0.3), random_pose(rng, 0.3)
            pa, pb = random_param(rng, ta), random_param(rng, tb)
            d, wa, wb, n, it = shape_distance(ta, Ta, pa, tb, Tb, pb)
            dt, dc = truth_distance(ta, Ta, pa, tb, Tb, pb)
            key = (NAMES[ta], NAMES[tb])
            if dc > 1e-6:
                n_sep += 1
                worst[key] = max(worst.get(key, 0.0), abs(d - dt))
                if d > 0:       # witness points realise the distance along the normal
                    assert abs(np.linalg.norm(wa - wb) - d) < 1e-9
                    assert np.abs((wa - wb) - d * n).max() < 1e-9
            else:               # truth: cores overlap -> we must report penetration of the cores
                mA = pa[0] if ta < 2 else pa[3]
                mB = pb[0] if tb < 2 else pb[3]
                assert d <= -(mA + mB) + 1e-6, key
    assert n_sep > 300 and len(worst) == 16
    assert max(worst.values()) < 1e-8, worst          # SLSQP itself is good to ~1e-9


def test_symmetry_and_predicate_consistency():
    rng = np.random.default_rng(11)
    for _ in range(3000):
        ta, tb = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        Ta, Tb = random_pose(rng, 0.25), random_pose(rng, 0.25)
        pa, pb = random_param(rng, ta), random_param(rng, tb)
        d, wa, wb, n, it = shape_distance(ta, Ta, pa, tb, Tb, pb)
        d2, wa2, wb2, n2, _ = shape_distance(tb, Tb, pb, ta, Ta, pa)
        # (penetration depths with a cylinder core come from EPA, which stops at a relative gap of 1e-8 or 32 vertices: the two
        # orders, and the predicate's canonical order, agree to that -- 1e-6 is the bar the truth tests hold it to)
        overlap_tol = 1e-6 if (d < 0 and 3 in (ta, tb)) else 1e-9 * max(1.0, abs(d))
        assert abs(d - d2) < overlap_tol
        assert it <= 64
        for thr in (0.0, 0.05, -0.01, d + max(1e-6 * abs(d), 2 * overlap_tol), d - max(1e-6 * abs(d), 2 * overlap_tol)):
            assert shape_collides(ta, Ta, pa, tb, Tb, pb, thr) == (d < thr)


def test_closed_form_known_answers():
    I = np.eye(4)

    def at(x, y, z):
        T = np.eye(4); T[:3, 3] = [x, y, z]; return T
    # sphere-sphere
    d = shape_distance(SPHERE, I, [0.1, 0, 0, 0], SPHERE, at(1, 0, 0), [0.2, 0, 0, 0])[0]
    assert d == pytest.approx(0.7, abs=1e-15)
    # capsule along z (half length 0.5, r 0.1) vs sphere beside its end cap
    d = shape_distance(CAPSULE, I, [0.1, 0.5, 0, 0], SPHERE, at(0, 0, 1.0), [0.1, 0, 0, 0])[0]
    assert d == pytest.approx(0.3, abs=1e-15)
    # two boxes face to face, and deep overlap (SAT depth along x)
    d = shape_distance(BOX, I, [0.5, 0.5, 0.5, 0], BOX, at(2, 0, 0), [0.5, 0.5, 0.5, 0])[0]
    assert d == pytest.approx(1.0, abs=1e-15)
    d = shape_distance(BOX, I, [0.5, 0.5, 0.5, 0], BOX, at(0.8, 0.1, 0), [0.5, 0.5, 0.5, 0])[0]
    assert d == pytest.approx(-0.2, abs=1e-15)
    # sphere centre inside a box: -(distance to the nearest face) - r
    d = shape_distance(SPHERE, at(0.3, 0, 0), [0.05, 0, 0, 0], BOX, I, [0.5, 0.6, 0.7, 0])[0]
    assert d == pytest.approx(-0.25, abs=1e-15)
    # cylinders: coaxial, stacked with a gap; side by side
    d = shape_distance(CYLINDER, I, [0.2, 0.5, 0, 0], CYLINDER, at(0, 0, 1.3), [0.3, 0.5, 0, 0])[0]
    assert d == pytest.approx(0.3, abs=1e-12)
    d = shape_distance(CYLINDER, I, [0.2, 0.5, 0, 0], CYLINDER, at(1.0, 0, 0), [0.3, 0.5, 0, 0])[0]
    assert d == pytest.approx(0.5, abs=1e-9)
    # cylinder standing on a plane z = 0 lifted by 0.1
    d = shape_distance(CYLINDER, at(0, 0, 0.6), [0.2, 0.5, 0, 0], PLANE, I, [0, 0, 1, 0])[0]
    assert d == pytest.approx(0.1, abs=1e-15)
    # margin: a box rounded by 0.04 keeps its faces, loses its corner
    d_face = shape_distance(BOX, I, [0.5, 0.5, 0.5, 0.04], SPHERE, at(1.0, 0, 0), [0.1, 0, 0, 0])[0]
    assert d_face == pytest.approx(0.4, abs=1e-15)
    d_sharp = shape_distance(BOX, I, [0.5, 0.5, 0.5, 0.0], SPHERE, at(1, 1, 1), [0.1, 0, 0, 0])[0]
    d_round = shape_distance(BOX, I, [0.5, 0.5, 0.5, 0.04], SPHERE, at(1, 1, 1), [0.1, 0, 0, 0])[0]
    assert d_round > d_sharp and d_round - d_sharp == pytest.approx(0.04 * (np.sqrt(3) - 1), abs=1e-12)


def test_degenerate_inputs_do_not_blow_up():
    I = np.eye(4)
    # coincident centres, every pair class: finite negative distance
    for ta in range(4):
        for tb in range(4):
            pa, pb = random_param(np.random.default_rng(ta), ta), random_param(np.random.default_rng(tb + 9), tb)
            d = shape_distance(ta, I, pa, tb, I, pb)[0]
            assert np.isfinite(d) and d < 0
    # parallel capsules (degenerate segment-segment denominator)
    T = np.eye(4); T[:3, 3] = [0.5, 0, 0.2]
    d = shape_distance(CAPSULE, I, [0.1, 0.4, 0, 0], CAPSULE, T, [0.1, 0.4, 0, 0])[0]
    assert d == pytest.approx(0.3, abs=1e-15)


def test_bullet_margins_match_truth_on_rounded_shapes():
    """``bullet_margins=True`` scenes: boxes / cylinders carry margin = min(0.04, a tenth of the smallest half extent) and are the
    Minkowski sum core (+) ball(margin) -- what Bullet's btBoxShape / btCylinderShape are.  Distances of such rounded shapes against
    the SLSQP truth, and the margin rule itself."""
    from numbotics_amd.robots.model import bullet_margin
    from numbotics_amd.utils import Shape
    assert bullet_margin(Shape.CUBE, {'half_extents': np.array([0.4, 0.4, 0.4])}) == 0.04
    assert np.isclose(bullet_margin(Shape.CUBOID, {'half_extents': np.array([0.03, 0.06, 0.045])}), 0.003)
    assert np.isclose(bullet_margin(Shape.CYLINDER, {'radius': 0.05, 'height': 0.12}), 0.005)
    assert bullet_margin(Shape.SPHERE, {'radius': 0.1}) == 0.0 and bullet_margin(Shape.CAPSULE, {'radius': 0.1, 'height': 1.0}) == 0.0
    assert bullet_margin(Shape.MESH, {}) == 0.001
    rng = np.random.default_rng(29)
    worst, n = 0.0, 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(250):
            ta, tb = int(rng.choice([BOX, CYLINDER])), int(rng.integers(0, 4))
            Ta, Tb = random_pose(rng, 0.35), random_pose(rng, 0.35)
            pa, pb = random_param(rng, ta), random_param(rng, tb)
            pa[3] = min(0.04, 0.1 * (pa[:3].min() if ta == BOX else min(pa[0], pa[1])))
            if tb in (BOX, CYLINDER):
                pb[3] = min(0.04, 0.1 * (pb[:3].min() if tb == BOX else min(pb[0], pb[1])))
            d = shape_distance(ta, Ta, pa, tb, Tb, pb)[0]
            dt, dc = truth_distance(ta, Ta, pa, tb, Tb, pb)
            if dc > 1e-6:
                n += 1
                worst = max(worst, abs(d - dt))
            for thr in (0.0, 0.02, d * (1 + 1e-6), d * (1 - 1e-6)):
                assert shape_collides(ta, Ta, pa, tb, Tb, pb, thr) == (d < thr)
    assert n > 100 and worst < 1e-8, (n, worst)
