"""MESH collision shapes on the CPU side: file readers, the reference's load_mesh transform sequence
(numbotics/utils/mesh.py:18-37), hull building, scene compilation (URDF <mesh>, numbotics/physics/helpers.py:252-255;
Mesh obstacles, numbotics/physics/object.py:425-447) and the oracle's hull geometry against an independent SLSQP solution
over the hulls' H-representation.  trimesh / vhacdx / pybullet are absent: parity with them is UNPINNED; what is pinned
is the geometry this build defines (one convex hull per mesh object, which is what Bullet's GEOM_MESH builds)."""
import os
import warnings

import numpy as np
import pytest

from numbotics_amd.utils.mesh import (read_obj, read_stl, load_mesh, convex_hull, mesh_hulls, center_of_mass, write_obj,
                                      hull_faces)
from numbotics_amd.scenes import MESH_DIR, build_scene, sample_q
from oracle.cpu_oracle import Oracle, HullSet, shape_distance_h, shape_collides_h, shape_distance
from geom_truth import SPHERE, CAPSULE, BOX, CYLINDER, HULL, random_pose, random_param, truth_distance

CUBE = np.array([[x, y, z] for x in (-1.0, 1.0) for y in (-1.0, 1.0) for z in (-1.0, 1.0)])


def test_obj_reader_groups_indices_and_polygons(tmp_path):
    p = tmp_path / "two.obj"
    p.write_text("# comment\n"
                 "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\n"
                 "o tet\nf 1 2 3\nf 1/1 2/2 4/4\nf 1//1 3//3 4//4\nf 2 3 4\n"
                 "v 5 0 0\nv 6 0 0\nv 6 1 0\nv 5 1 0\nv 5 0 1\nv 6 0 1\nv 6 1 1\nv 5 1 1\n"
                 "g box\nf -8 -7 -6 -5\nf -4 -3 -2 -1\nf 5 6 10 9\n"
                 "o empty_object_without_faces\n")
    parts = read_obj(str(p))
    assert [q.name for q in parts] == ["tet", "box"]
    assert parts[0].vertices.shape == (4, 3) and len(parts[0].faces) == 4
    assert parts[1].vertices.shape == (8, 3) and len(parts[1].faces[0]) == 4        # quads survive, negative indices resolved
    assert np.array_equal(parts[1].vertices.min(axis=0), [5, 0, 0]) and np.array_equal(parts[1].vertices.max(axis=0), [6, 1, 1])
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(ValueError):
        load_mesh(str(bad))
    with pytest.raises(ValueError):
        load_mesh(str(tmp_path / "missing.obj"))
    with pytest.raises(ValueError):
        load_mesh(str(tmp_path / "mesh.dae"))


def test_stl_binary_and_ascii(tmp_path):
    parts = read_stl(os.path.join(MESH_DIR, "wedge.stl"))
    assert len(parts) == 1 and parts[0].vertices.shape == (24, 3)                  # 8 triangles, vertices not merged
    h = convex_hull(parts[0].vertices)
    assert h.vertices.shape == (6, 3) and h.planes.shape == (5, 4)                  # a triangular prism
    a = tmp_path / "a.stl"
    a.write_text("solid one\nfacet normal 0 0 1\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nendloop\nendfacet\nendsolid one\n"
                 "solid two\nfacet normal 0 0 1\nouter loop\nvertex 0 0 2\nvertex 1 0 2\nvertex 0 1 2\nendloop\nendfacet\nendsolid two\n")
    parts = read_stl(str(a))
    assert [q.name for q in parts] == ["one", "two"] and parts[1].vertices[0, 2] == 2.0


def test_load_mesh_transform_sequence_and_center_of_mass(tmp_path):
    """auto_center -> scale -> offset, in that order (numbotics/utils/mesh.py:26-31)."""
    V, F = hull_faces(CUBE * [0.5, 1.0, 2.0] + [3.0, -1.0, 0.5])
    f = write_obj(str(tmp_path / "c.obj"), [("c", V, F)])
    assert np.allclose(center_of_mass(read_obj(f)), [3.0, -1.0, 0.5], atol=1e-12)            # closed: volume centroid
    T = random_pose(np.random.default_rng(1), 0.4)
    parts = load_mesh(f, mesh_scale=np.array([2.0, 1.0, 0.5]), offset=T, auto_center=True)
    want = ((V - [3.0, -1.0, 0.5]) * [2.0, 1.0, 0.5]) @ T[:3, :3].T + T[:3, 3]
    assert np.allclose(parts[0].vertices, want, atol=1e-14)
    # an open surface (one face missing) falls back to the area-weighted centroid: not the volume centroid of a skewed solid
    tetra = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    Vt, Ft = hull_faces(tetra)
    g = write_obj(str(tmp_path / "open.obj"), [("t", Vt, Ft[:-1])])
    assert not np.allclose(center_of_mass(read_obj(g)), tetra.mean(axis=0))
    # convex_decomposition: a multi-object file IS the decomposition; a single object cannot be decomposed here
    assert len(load_mesh(os.path.join(MESH_DIR, "table.obj"), convex_decomposition=True)) == 5
    assert len(mesh_hulls(os.path.join(MESH_DIR, "table.obj"), convex_decomposition=True)) == 5
    # without it the objects of a file are ONE mesh (trimesh merges them before the reference exports the file for Bullet):
    # one hull of all the vertices -- the table's top and the space between its legs are inside it
    one = mesh_hulls(os.path.join(MESH_DIR, "table.obj"))
    allv = np.concatenate([p.vertices for p in load_mesh(os.path.join(MESH_DIR, "table.obj"))])
    assert len(one) == 1 and (((allv - one[0].center) @ one[0].planes[:, :3].T) <= one[0].planes[:, 3] + 1e-12).all()
    under_the_top = np.array([allv[:, 0].mean(), allv[:, 1].mean(), 0.5 * (allv[:, 2].min() + allv[:, 2].max())])
    assert ((under_the_top - one[0].center) @ one[0].planes[:, :3].T <= one[0].planes[:, 3]).all()
    parts = mesh_hulls(os.path.join(MESH_DIR, "table.obj"), convex_decomposition=True)
    assert not any((((under_the_top - h.center) @ h.planes[:, :3].T) <= h.planes[:, 3]).all() for h in parts)
    with pytest.raises(NotImplementedError):
        load_mesh(f, convex_decomposition=True)


def test_convex_hull_parts():
    h = convex_hull(np.concatenate([CUBE, CUBE * 0.5, np.zeros((1, 3))]) + [1.0, 2.0, 3.0])
    assert h.vertices.shape == (8, 3) and h.planes.shape == (6, 4)                            # interior points dropped, facets merged
    assert np.allclose(h.center, [1.0, 2.0, 3.0]) and np.isclose(h.radius, np.sqrt(3.0))
    assert np.allclose(np.abs(h.planes[:, :3]).sum(axis=1), 1.0) and np.allclose(h.planes[:, 3], 1.0)
    assert (h.vertices @ h.planes[:, :3].T <= h.planes[:, 3] + 1e-12).all()
    flat = convex_hull(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.5, 0.5, 0]], dtype=float))
    assert flat.planes.shape == (0, 4) and flat.vertices.shape == (5, 3)                       # degenerate: support function only
    one = convex_hull(np.array([[1.0, 2.0, 3.0]]))
    assert one.vertices.shape == (1, 3) and one.radius == 0.0


def test_scene_compilation_with_mesh_links_and_mesh_obstacles(fresh_world):
    arm, chain, obs = build_scene("c5m")
    sm = arm.scene_model()
    assert sm.n_rshapes == 11 and (sm.rshape_type == 5).sum() == 10 and (sm.rshape_type == 0).sum() == 1
    # bracelet_link: two <collision> elements, one single-object file each -> two hull primitives on that link (compound)
    li = [l._name for l in sm.links].index("bracelet_link")
    assert (sm.rshape_link == li).sum() == 2
    # obstacles: rock 1 + table 5 (its five objects, convex_decomposition=True) + wedge 1 hulls + 1 box
    assert sm.n_wshapes == 8 and (sm.wshape_type == 5).sum() == 7 and sm.n_hulls == 17
    assert sm.hull_vert_begin[-1] == len(sm.hull_verts) and sm.hull_face_begin[-1] == len(sm.hull_planes)
    # every hull is centred on the mean of its vertices and its planes contain all of its vertices
    for h in range(sm.n_hulls):
        V = sm.hull_verts[sm.hull_vert_begin[h]:sm.hull_vert_begin[h + 1]]
        P = sm.hull_planes[sm.hull_face_begin[h]:sm.hull_face_begin[h + 1]]
        assert np.abs(V.mean(axis=0)).max() < 1e-12 and len(P) >= 4
        assert (V @ P[:, :3].T <= P[:, 3] + 1e-12).all() and np.allclose(np.linalg.norm(P[:, :3], axis=1), 1.0)
    # the wedge's shape kwargs (mesh_scale, offset) and the body pose all reached the hull: its lowest vertex sits on z = 0
    w = int(np.flatnonzero(sm.wshape_obj == 2)[0])
    h = int(sm.wshape_param[w, 0])
    V = sm.hull_verts[sm.hull_vert_begin[h]:sm.hull_vert_begin[h + 1]]
    T = np.vstack([sm.wshape_pose[w].reshape(3, 4), [0, 0, 0, 1]])
    Vw = V @ T[:3, :3].T + T[:3, 3]
    assert abs(Vw[:, 2].min()) < 1e-12 and abs(Vw[:, 2].max() - 0.25 * 1.2) < 1e-12
    # mesh hulls are inscribed in the cylinders they replace: separated pairs are never closer than with the primitives
    orc_m = Oracle(sm)
    q = sample_q(chain, 400, seed=3)
    assert 0.02 < orc_m.validity(q).mean() < 0.6
    d, wit = orc_m.pair_distances(q[:50], witness=True)
    sep = d > 1e-9
    assert sep.mean() > 0.9
    assert np.abs(np.linalg.norm(wit[..., 0:3] - wit[..., 3:6], axis=-1) - d)[sep].max() < 1e-9      # witnesses realise the distance


def test_mesh_arm_is_never_closer_than_the_cylinder_arm(fresh_world):
    from numbotics_amd.physics.world import _reset_worlds
    from numbotics_amd.physics import World
    arm, chain, obs = build_scene("c2m", bullet_margins=False)         # the sharp shapes: hull vertices on the cylinders' surfaces
    q = sample_q(chain, 300, seed=5)
    dm = Oracle(arm.scene_model()).pair_distances(q)
    _reset_worlds(); World()
    arm2, chain2, obs2 = build_scene("c2", bullet_margins=False)
    dc = Oracle(arm2.scene_model()).pair_distances(q)
    assert dm.shape == dc.shape                 # same pair list: the compound bracelet link yields the same two primitives
    both = (dm > 1e-6) & (dc > 1e-6)
    assert both.mean() > 0.9 and (dm - dc)[both].min() > -1e-9


def _random_hull(rng, hs, n=24):
    pts = rng.normal(size=(n, 3))
    pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.6, 1.0, (n, 1)) * rng.uniform(0.05, 0.3, 3)
    part = convex_hull(pts)
    return hs.add(part), part


def test_hull_distances_match_slsqp_truth():
    """hull-hull and hull-primitive core distances, witnesses and the validity predicate against the convex programme
    solved over the hulls' face planes (the oracle's GJK only ever touches the vertex lists)."""
    rng = np.random.default_rng(17)
    hs = HullSet()
    worst = {}
    n_sep = 0
    names = {0: "sphere", 1: "capsule", 2: "box", 3: "cylinder", 5: "hull"}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(260):
            tb = int(rng.choice([0, 1, 2, 3, 5]))
            ia, part_a = _random_hull(rng, hs)
            Ta, Tb = random_pose(rng, 0.35), random_pose(rng, 0.35)
            ma = float(rng.choice([0.0, 0.0, 0.01]))
            pa = [ia, 0.0, 0.0, ma]
            pa_truth = [0.0, 0.0, 0.0, ma, (part_a.vertices, part_a.planes)]
            if tb == 5:
                ib, part_b = _random_hull(rng, hs)
                pb, pb_truth = [ib, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, (part_b.vertices, part_b.planes)]
            else:
                pb = random_param(rng, tb)
                pb_truth = pb
            swap = bool(rng.integers(0, 2))
            args = (tb, Tb, pb, 5, Ta, pa) if swap else (5, Ta, pa, tb, Tb, pb)
            d, wa, wb, n, iters = shape_distance_h(hs, *args)
            dt, dc = truth_distance(5, Ta, pa_truth, tb, Tb, pb_truth)
            assert iters <= 64
            key = names[tb]
            if dc > 1e-6:
                n_sep += 1
                worst[key] = max(worst.get(key, 0.0), abs(d - dt))
                if d > 0:
                    assert abs(np.linalg.norm(wa - wb) - d) < 1e-9 and np.abs((wa - wb) - d * n).max() < 1e-9
            else:
                mb = pb[0] if tb in (0, 1) else pb[3]
                assert d <= -(ma + mb) + 1e-6, key
            for thr in (0.0, 0.02, -0.003, d * (1 + 1e-6), d * (1 - 1e-6)):
                assert shape_collides_h(hs, *args, thr) == (d < thr), (key, thr, d)
    assert n_sep > 120 and len(worst) == 5
    assert max(worst.values()) < 1e-8, worst


def test_hull_of_a_box_equals_the_box_primitive():
    """A hull whose vertices are a box's corners: separated distances equal the box primitive's to rounding; overlapping
    ones too against a sphere (a point inside a hull: the face planes give the exact depth), otherwise the hull's depth is
    the documented upper bound (face normals + the partner's axes + the centre line: no edge-edge axes)."""
    rng = np.random.default_rng(23)
    hs = HullSet()
    for _ in range(300):
        he = rng.uniform(0.05, 0.3, 3)
        ia = hs.add(convex_hull(CUBE * he))
        tb = int(rng.integers(0, 4))
        Ta, Tb = random_pose(rng, 0.25), random_pose(rng, 0.25)
        pb = random_param(rng, tb)
        dh = shape_distance_h(hs, 5, Ta, [ia, 0, 0, 0], tb, Tb, pb)[0]
        db = shape_distance(2, Ta, [he[0], he[1], he[2], 0.0], tb, Tb, pb)[0]
        if db > 0 or tb == 0:
            assert abs(dh - db) < 1e-9, (tb, dh, db)
        else:
            assert dh < 0 and dh <= db + 1e-9, (tb, dh, db)


def _sheared(T, eps):
    """The pose a link gets when its joint axis is given to a few digits only: Rodrigues with a non-unit axis leaves the
    rotation orthonormal to ~eps (the reference uses the axis as given, robots/helpers.py:43-55)."""
    S = T.copy()
    S[:3, 2] *= 1.0 + eps
    return S


def test_cylinder_support_with_an_axis_that_is_not_quite_unit():
    """Captured from the fuzz campaign (seed 70503, row 41921): a cylinder on a link whose rotation is orthonormal to 2.4e-6
    only, 7.2 mm from a flat hull.  The boolean walk reached a search direction parallel to the cylinder axis to 1e-9; the
    old support routine then returned a point 25 mm beyond the cap and both GJK predicates reported a collision."""
    Ta = np.eye(4)
    Ta[:3] = [[-0.3517763815178784, -0.5800229888189756, -0.7347301184496722, 0.4215516531568988],
              [-0.2353265310503352, -0.7048913061458668, 0.6691408572229722, 0.765730064901918],
              [-0.9060293326209685, 0.40829119121473895, 0.111465674169479, 0.21667821304131982]]
    Tb = np.eye(4)
    Tb[:3] = [[0.5764151840484136, -0.8030006847075268, -0.15144449794454842, 0.5383620147343717],
              [0.6486553841895438, 0.5623445676167756, -0.5128496659195719, 0.592557737901488],
              [0.49698262360000955, 0.19737904557258373, 0.8450146650848772, 0.13278055792270793]]
    V = np.array([[-0.13120378261111507, -0.020409024073481954, 0.010804724290791042], [-0.0997475749780061, -0.06307594605645662, -0.004016755366807444],
                  [-0.07617097416858774, -0.03600921341436852, -0.017783099370204822], [-0.07058986061864665, 0.05285513033391884, -0.016917463431851895],
                  [-0.048740045190185974, -0.0862752884480052, 0.011847596983291896], [-0.04576513014462036, 0.04475304731113082, 0.020094230150352152],
                  [-0.02409959318614671, 0.03631267947495086, 0.020791416960678908], [-0.015527361410270682, -0.03444917188470428, -0.022012430250863347],
                  [-0.01397888116297892, 0.012336178246702356, 0.022173955965907648], [0.002025428536813499, -0.04194707187471403, -0.020320262228817408],
                  [0.01046466592576778, 0.06851273887540982, 0.0011497061531134693], [0.016992773884378382, 0.03058686127917226, -0.018051213524855732],
                  [0.02701514434902899, -0.10461295799492545, 0.003752622236137681], [0.07184753131671641, 0.08195956001253224, 0.006286442592893948],
                  [0.08279216210149143, -0.04719529790758362, 0.01876537070002667], [0.08566116288233198, 0.08699239942202616, 0.0027088916083875436],
                  [0.11379277563853701, -0.02608700210476586, -0.012887066137633192], [0.11523155883549291, 0.04575237880316217, -0.006386667330547111]])
    assert abs(np.linalg.norm(Ta[:3, 2]) ** 2 - 1.0) > 1e-6          # the axis of the captured pose is not unit
    hs = HullSet()
    ih = hs.add(convex_hull(V))
    cyl = [0.0299, 0.1126, 0.0, 0.0]
    d = shape_distance_h(hs, 3, Ta, cyl, 5, Tb, [ih, 0, 0, 0])[0]
    assert 0.0070 < d < 0.0075
    for thr in (0.0, 1e-6, 0.007):
        assert not shape_collides_h(hs, 3, Ta, cyl, 5, Tb, [ih, 0, 0, 0], thr)
        assert not shape_collides_h(hs, 5, Tb, [ih, 0, 0, 0], 3, Ta, cyl, thr)
    assert shape_collides_h(hs, 3, Ta, cyl, 5, Tb, [ih, 0, 0, 0], 0.0075)


def test_predicates_agree_with_the_distance_under_sheared_poses():
    """Cylinders against every solid under poses orthonormal to 1e-5 / 1e-3 only, with the partner placed on the cylinder's
    axis (the walk's first direction is then axial): the boolean walk (threshold 0), the distance predicate (1e-6) and the
    distance routine have to tell the same story."""
    rng = np.random.default_rng(91)
    hs = HullSet()
    _random_hull(rng, hs)
    n_close = 0
    for it in range(1500):
        eps = float(rng.choice([0.0, 2.4e-6, 1e-5, 1e-3]))
        Ta = _sheared(random_pose(rng, 0.2), eps)
        rad, hh = rng.uniform(0.02, 0.1), rng.uniform(0.05, 0.3)
        tb = int(rng.choice([2, 3, 5]))
        Tb = _sheared(random_pose(rng, 0.2), eps)
        if it % 2 == 0:       # partner centred on the cylinder's axis, beyond a cap
            Tb[:3, 3] = Ta[:3, 3] + Ta[:3, 2] * (hh + rng.uniform(0.0, 0.25)) * rng.choice([-1.0, 1.0])
        if tb == 5:
            ib, _ = _random_hull(rng, hs)
            pb = [ib, 0.0, 0.0, 0.0]
        else:
            pb = random_param(rng, tb)
        pa = [rad, hh, 0.0, 0.0]
        d = shape_distance_h(hs, 3, Ta, pa, tb, Tb, pb)[0]
        if abs(d) < 1e-5:
            continue
        n_close += abs(d) < 0.02
        for thr in (0.0, 1e-6):
            assert shape_collides_h(hs, 3, Ta, pa, tb, Tb, pb, thr) == (d < thr), (it, eps, tb, d, thr)
            assert shape_collides_h(hs, tb, Tb, pb, 3, Ta, pa, thr) == (d < thr), (it, eps, tb, d, thr)
    assert n_close > 50
