"""Proximity-Jacobian rows of the oracle (Arm.jacobian_proximity, reference arm.py:620-632).

The reference needs PyBullet for the proximities, so the rows are pinned two ways instead: (1) against the
composition the reference writes down -- n @ J_lin(global_pose at the witness point) of subject minus target --
using orc_jacobian, which is pinned by the golden Jacobians (G2/G3); (2) against central differences of the signed
distance (the row is its gradient wherever the pair is separated)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.cpu_oracle import Oracle, lib, _p
from numbotics_amd.scenes import build_scene, sample_q

TREE_URDF = os.path.join(os.path.dirname(__file__), "models", "tree_gripper.urdf")


def _path_of(kin, joint):
    path = []
    while joint >= 0:
        path.append(joint)
        joint = int(kin.joint_parent[joint])
    return np.array(path[::-1], dtype=np.int32)


def _lin_jac(orc, kin, joint, q, point):
    """Linear rows of orc_jacobian(mode 2) of the moving frame `joint` at the world point."""
    path = _path_of(kin, joint)
    local = np.eye(4)[:3].reshape(12).copy()
    pose = np.eye(4).reshape(1, 16).copy()
    pose[0, [3, 7, 11]] = point
    out = np.empty((1, 6, kin.n_q))
    qq = np.ascontiguousarray(q.reshape(1, -1))
    lib().orc_jacobian(C.byref(orc._m), _p(qq), C.c_int64(1), _p(path), C.c_int32(len(path)), _p(local), C.c_int32(2),
                       _p(pose), _p(out))
    return out[0, :3]


def _check_rows(arm, chain, n=24, seed=5):
    sm = arm.scene_model()
    kin = sm.kin
    orc = Oracle(sm)
    q = sample_q(chain, n, seed=seed)
    d, w, rows = orc.proximity_jacobian(q)
    d0, w0 = orc.pair_distances(q, witness=True)
    assert np.array_equal(d, d0) and np.array_equal(w, w0)
    S = sm.n_rshapes
    worst = 0.0
    for b in range(n):
        for p in range(sm.n_pairs):
            a, t = int(sm.pair_a[p]), int(sm.pair_b[p])
            nrm = w[b, p, 6:9]
            ref = np.zeros(kin.n_q)
            fa = int(sm.rshape_frame[a])
            if fa >= 0:
                ref += nrm @ _lin_jac(orc, kin, fa, q[b], w[b, p, 0:3])
            if t < S and int(sm.rshape_frame[t]) >= 0:
                ref -= nrm @ _lin_jac(orc, kin, int(sm.rshape_frame[t]), q[b], w[b, p, 3:6])
            worst = max(worst, np.abs(rows[b, p] - ref).max())
    assert worst < 1e-12, worst
    # gradient of the signed distance (separated pairs; h balances GJK's 1e-10 convergence against truncation)
    h = 1e-4
    num = np.zeros_like(rows)
    for j in range(kin.n_q):
        qp, qm = q.copy(), q.copy()
        qp[:, j] += h
        qm[:, j] -= h
        num[:, :, j] = (orc.pair_distances(qp) - orc.pair_distances(qm)) / (2 * h)
    sep = d > 5e-3
    assert sep.mean() > 0.5
    err = np.abs(num - rows)[sep]
    assert err.max() < 2e-5, err.max()
    return d, rows


def test_rows_kinova_scene(fresh_world):
    arm, chain, obs = build_scene("c2")
    d, rows = _check_rows(arm, chain)
    assert np.abs(rows).max() > 0.1                          # not trivially zero


def test_rows_tree_robot_with_prismatic_joints(fresh_world):
    from numbotics_amd.physics import GraphChain, Cube, Sphere, Capsule
    from numbotics_amd.robots import Arm
    chain = GraphChain.from_urdf(TREE_URDF)
    arm = Arm(chain)
    obs = [Cube(0.0, 0.08, position=np.array([0.35, 0.0, 0.55])), Sphere(0.0, 0.05, position=np.array([0.2, 0.2, 0.4])),
           Capsule(0.0, 0.03, 0.3, position=np.array([-0.2, 0.1, 0.6]))]
    _check_rows(arm, chain, n=16, seed=9)
    assert len(obs) == 3
