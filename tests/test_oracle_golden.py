"""The CPU oracle against the golden vectors produced by the reference's own source
(tests/golden/make_golden.py).  Tolerance 1e-12: the oracle fuses multiply-adds and uses its own
sincos, the reference is plain NumPy; the north-star bar for poses is 1e-6."""
import numpy as np
import pytest

from oracle.cpu_oracle import Oracle, edge_samples, sincos
from numbotics_amd.robots.model import compile_kinematics
from numbotics_amd.planning import unit_bspline
from helpers import kin_from_sequence

TOL = 1e-12


def test_sincos_accuracy():
    x = np.random.default_rng(0).uniform(-100, 100, 200000)
    s, c = sincos(x)
    assert np.abs(s - np.sin(x)).max() < 3e-16
    assert np.abs(c - np.cos(x)).max() < 3e-16
    s, c = sincos(np.array([0.0, np.pi / 2, np.pi, -np.pi / 2, 1e9, np.inf, np.nan, 3e9]))
    assert s[0] == 0.0 and c[0] == 1.0
    assert abs(s[4] - np.sin(1e9)) < 1e-9
    assert np.isnan(s[5]) and np.isnan(s[6]) and np.isnan(c[7])     # documented domain |x| < 2^31


@pytest.mark.parametrize("name", ["rev7", "mixed_fixed"])
def test_g1_chain_sweep(g12, name):
    """nb_compute_transformation (robots/helpers.py:91-113)."""
    off, ax = g12[f"g1_{name}_offsets"], g12[f"g1_{name}_axes"]
    ty, ix, q, T0 = g12[f"g1_{name}_types"], g12[f"g1_{name}_idxs"], g12[f"g1_{name}_q"], g12[f"g1_{name}_T0"]
    km = kin_from_sequence(off, ax, ty, ix, T0[0], q.shape[1])
    out = Oracle(km).fk(q, "end")
    assert np.abs(out - g12[f"g1_{name}_T"]).max() < TOL


def test_g1_reference_cannot_do_prismatic_or_spherical(golden_meta):
    """Documented upstream breakage: both branches raise under NumPy."""
    raised = golden_meta["g1_reference_raised"]
    assert raised["prismatic"].startswith("ValueError") and raised["spherical"].startswith("ValueError")


def test_prismatic_is_a_rigid_motion():
    """Our prismatic joint (fixed, Q5): rotation stays orthonormal, translation moves along R_off @ axis."""
    rng = np.random.default_rng(3)
    from geom_truth import random_pose
    off = np.stack([random_pose(rng) for _ in range(2)])
    ax = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
    km = kin_from_sequence(off, ax, [0, 1], [0, 1], np.eye(4), 2)
    q = rng.uniform(-1, 1, (16, 2))
    T = Oracle(km).fk(q, "end")
    assert np.abs(T[:, :3, :3] @ np.swapaxes(T[:, :3, :3], 1, 2) - np.eye(3)).max() < 1e-14
    q2 = q.copy(); q2[:, 1] += 0.25
    T2 = Oracle(km).fk(q2, "end")
    step = T2[:, :3, 3] - T[:, :3, 3]
    assert np.abs(np.linalg.norm(step, axis=1) - 0.25).max() < 1e-14
    assert np.abs(T2[:, :3, :3] - T[:, :3, :3]).max() < 1e-15


@pytest.mark.parametrize("name", ["rev7", "mixed_fixed"])
def test_g2_jacobian(g12, name):
    """nb_compute_jacobian (robots/helpers.py:117-187) incl. use_com / local_pose / global_pose."""
    pre = f"g2_{name}"
    off, ax, ty, ix = g12[f"{pre}_offsets"], g12[f"{pre}_axes"], g12[f"{pre}_types"], g12[f"{pre}_idxs"]
    q, T0, com = g12[f"{pre}_q"], g12[f"{pre}_T0"], g12[f"{pre}_com"]
    km = kin_from_sequence(off[:-1], ax, ty, ix, T0[0], q.shape[1], trailing=off[-1])
    orc = Oracle(km)
    assert np.abs(orc.jacobian(q, "end") - g12[f"{pre}_J_plain"]).max() < TOL
    assert np.abs(orc.jacobian(q, "end", extra_local=com) - g12[f"{pre}_J_com"]).max() < TOL
    assert np.abs(orc.jacobian(q, "end", local_pose=g12[f"{pre}_local_pose"]) - g12[f"{pre}_J_local"]).max() < TOL
    assert np.abs(orc.jacobian(q, "end", global_pose=g12[f"{pre}_global_pose"]) - g12[f"{pre}_J_global"]).max() < TOL


def test_g3_flattening_matches_reference(kinova, g3, golden_meta):
    """Arm.__init__ flattening (robots/arm.py:17-71): per-frame offsets/axes/types/idxs, bit for bit."""
    arm, chain, _ = kinova
    for f in golden_meta["g3_frames"]:
        offsets, axes, types, idxs = arm._link_joint_sequence[f]
        assert np.array_equal(offsets, g3[f"g3_seq_{f}_offsets"]), f
        assert np.array_equal(axes, g3[f"g3_seq_{f}_axes"]), f
        assert np.array_equal(types, g3[f"g3_seq_{f}_types"]), f
        assert np.array_equal(idxs, g3[f"g3_seq_{f}_idxs"]), f


def test_g3_kinova_fk_and_jacobian(kinova, g3, golden_meta):
    """Config 1 of BASELINE.json: Kinova batched FK, 1024 random q, on the CPU path."""
    arm, chain, _ = kinova
    orc = Oracle(compile_kinematics(chain))
    q = g3["g3_q"]
    assert q.shape == (1024, 7)
    for f in golden_meta["g3_trailing_fixed_frames"]:
        ref = g3[f"g3_fk_{f}"]
        assert np.abs(orc.fk(q[:ref.shape[0]], f) - ref).max() < TOL, f
        jr = g3[f"g3_jac_{f}"]
        assert np.abs(orc.jacobian(q[:jr.shape[0]], f) - jr).max() < TOL, f
    assert g3["g3_fk_tool_frame"].shape == (1024, 4, 4)
    # option coverage
    lp, gp, lpb = g3["g3_local_pose"], g3["g3_global_pose"], g3["g3_local_pose_batch"]
    assert np.abs(orc.fk(q[:128], "tool_frame", extra_local=lp) - g3["g3_fk_tool_frame_local"]).max() < TOL
    assert np.abs(orc.fk(q[:128], "tool_frame", local_pose=lpb) - g3["g3_fk_tool_frame_local_batch"]).max() < TOL
    assert np.abs(orc.jacobian(q[:128], "tool_frame", extra_local=lp) - g3["g3_jac_tool_frame_local"]).max() < TOL
    assert np.abs(orc.jacobian(q[:128], "tool_frame", global_pose=gp) - g3["g3_jac_tool_frame_global"]).max() < TOL


def test_g3_q1_frames_position_only(kinova, g3, golden_meta):
    """for links that are the direct child of a moving joint the reference drops the last
    joint's rotation.  Positions agree; the orientation is ours (correct) and differs from upstream."""
    arm, chain, _ = kinova
    orc = Oracle(compile_kinematics(chain))
    q = g3["g3_q"]
    checked = 0
    for f in golden_meta["g3_frames"]:
        if f in golden_meta["g3_trailing_fixed_frames"]:
            continue
        ref = g3[f"g3_fk_{f}"]
        out = orc.fk(q[:ref.shape[0]], f)
        assert np.abs(out[:, :3, 3] - ref[:, :3, 3]).max() < TOL
        assert np.abs(out[:, :3, :3] - ref[:, :3, :3]).max() > 0.1
        assert np.abs(out[:, :3, :3] @ np.swapaxes(out[:, :3, :3], 1, 2) - np.eye(3)).max() < 1e-13
        checked += 1
    assert checked == 7


def test_fk_jacobian_consistency(kinova):
    """Independent of the reference: the linear block of J is the derivative of the FK position."""
    arm, chain, _ = kinova
    orc = Oracle(compile_kinematics(chain))
    q = np.random.default_rng(4).uniform(-2, 2, (8, 7))
    for f in ("tool_frame", "forearm_link", "gripper"):
        J = orc.jacobian(q, f)
        h = 1e-6
        for j in range(7):
            qp, qm = q.copy(), q.copy()
            qp[:, j] += h; qm[:, j] -= h
            num = (orc.fk(qp, f)[:, :3, 3] - orc.fk(qm, f)[:, :3, 3]) / (2 * h)
            assert np.abs(num - J[:, :3, j]).max() < 1e-8


def test_g4_default_self_collision_pairs(kinova, golden_meta, fresh_world):
    """Arm.self_collision_pairs (robots/arm.py:190-223), effective rule incl. dead weld filter (Q3)."""
    from numbotics_amd.physics import GraphChain
    from numbotics_amd.robots import Arm
    from conftest import URDF
    arm = Arm(GraphChain.from_urdf(URDF))
    mine = sorted(tuple(sorted((a._name, b._name))) for a, b in arm.self_collision_pairs())
    assert [list(p) for p in mine] == golden_meta["g4_self_collision_pairs"]
    strict = Arm(GraphChain.from_urdf(URDF), weld_filter=True)
    assert len(strict.self_collision_pairs()) < len(mine)


def test_g5_edge_sampling(g5, golden_meta):
    """DiscreteConnector.connect/steer discretisation (connectors.py:57-100): counts and sample points
    bit-for-bit, including d = 1, 0.5, 0 and d < float32 eps."""
    n_cmp = 0
    for c in golden_meta["g5_cases"]:
        k = c["id"]
        s, g, ref = g5[f"g5_{k}_start"], g5[f"g5_{k}_goal"], g5[f"g5_{k}_samples"]
        mine = edge_samples(s, g, c["resolution"], c["max_distance"], c["mode"])
        n = ref.shape[0]
        if c["fail_at"] is None:
            assert mine.shape[0] == n, (k, mine.shape[0], n)          # full walk: the counts must agree
        if n:
            assert np.array_equal(mine[:n], ref), k                    # bit-exact sample points
            n_cmp += 1
    assert n_cmp > 400


def test_g5_bspline(g5):
    cp, ts = g5["g5_bspline_cp"], g5["g5_bspline_t"]
    spl = unit_bspline(cp)
    assert np.array_equal(np.stack([spl(t) for t in ts]), g5["g5_bspline_val"])        # degree 1: bit-exact
    spl2 = unit_bspline(g5["g5_bspline2_cp"], degree=2)
    assert np.abs(np.stack([spl2(t) for t in ts]) - g5["g5_bspline2_val"]).max() < 1e-14
