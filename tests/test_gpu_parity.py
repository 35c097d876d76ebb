"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle, bit for bit, and against the
reference-generated golden vectors.  Needs a real MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.cpu_oracle import Oracle, sincos, sqrt_div
from numbotics_amd.scenes import build_scene, sample_q
from numbotics_amd._lib import debug_option


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


class fused_path:
    """Route small batches through the fused kernel k_validity (the default sends every size through broadphase + narrowphase)."""

    def __enter__(self):
        from numbotics_amd._lib import set_debug_option
        set_debug_option("two_kernel_min_b", 10 ** 9)

    def __exit__(self, *exc):
        from numbotics_amd._lib import set_debug_option
        set_debug_option("two_kernel_min_b", 1)
        return False


def assert_bitwise(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        idx = np.argwhere(~same)[0]
        raise AssertionError(f"{what}: {(~same).sum()} of {same.size} values differ; first at {tuple(idx)}: "
                             f"{a[tuple(idx)]!r} vs {b[tuple(idx)]!r}; max abs diff {np.nanmax(np.abs(a - b))}")


def test_arithmetic_contract(torch_cuda):
    """sincos / sqrt / divide on the device round exactly like the host oracle."""
    from numbotics_amd.engine import selftest_math
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(-10, 10, 200000), rng.uniform(0, 1e-3, 50000), rng.uniform(-1e6, 1e6, 50000),
                        np.array([0.0, -0.0, 1.0, np.pi, np.pi / 2, 1e-300, 1e300, np.inf, np.nan, 4e9])])
    b = rng.uniform(-3, 3, a.size)
    b[b == 0] = 1.0
    s, c, sq, dv = selftest_math(a, b)
    so, co = sincos(a)
    with np.errstate(invalid="ignore"):
        sqo, dvo = sqrt_div(a, b)
    assert_bitwise(s, so, "sin")
    assert_bitwise(c, co, "cos")
    assert_bitwise(sq, sqo, "sqrt")
    assert_bitwise(dv, dvo, "div")


def test_fk_bitwise_and_golden(kinova, g3, golden_meta, torch_cuda):
    arm, chain, _ = kinova
    orc = Oracle(arm._kin)
    q = g3["g3_q"]
    for f in golden_meta["g3_frames"]:
        T = arm.forward_kinematics(q, f)
        assert_bitwise(T, orc.fk(q, f), f"fk {f}")
    for f in golden_meta["g3_trailing_fixed_frames"]:
        ref = g3[f"g3_fk_{f}"]
        T = arm.forward_kinematics(q[:ref.shape[0]], f)
        assert np.abs(T - ref).max() < 1e-12          # north-star bar: 1e-6
    # shapes and options, as the reference returns them
    assert arm.forward_kinematics(q[0], "tool_frame").shape == (4, 4)
    assert np.abs(arm.forward_kinematics(q[0], "tool_frame") - g3["g3_fk_tool_frame_1d"]).max() < 1e-12
    T3 = arm.forward_kinematics(q[:12].reshape(3, 4, 7), "tool_frame")
    assert T3.shape == (3, 4, 4, 4) and np.abs(T3 - g3["g3_fk_tool_frame_3d"]).max() < 1e-12
    lp, lpb = g3["g3_local_pose"], g3["g3_local_pose_batch"]
    assert np.abs(arm.forward_kinematics(q[:128], "tool_frame", local_pose=lp) - g3["g3_fk_tool_frame_local"]).max() < 1e-12
    assert np.abs(arm.forward_kinematics(q[:128], "tool_frame", local_pose=lpb) - g3["g3_fk_tool_frame_local_batch"]).max() < 1e-12
    # ragged sizes around the 64-lane block
    for B in (1, 63, 64, 65, 127, 129):
        assert_bitwise(arm.forward_kinematics(q[:B], "gripper"), orc.fk(q[:B], "gripper"), f"fk B={B}")
    # torch in -> torch out, stays on the device
    tq = torch_cuda.from_numpy(q).cuda()
    Tt = arm.forward_kinematics(tq, "tool_frame")
    assert Tt.is_cuda and Tt.shape == (1024, 4, 4)
    assert_bitwise(Tt.cpu().numpy(), orc.fk(q, "tool_frame"), "fk torch")


def test_jacobian_bitwise_and_golden(kinova, g3, golden_meta, torch_cuda):
    arm, chain, _ = kinova
    orc = Oracle(arm._kin)
    q = g3["g3_q"]
    for f in golden_meta["g3_frames"]:
        assert_bitwise(arm.jacobian(q[:200], f), orc.jacobian(q[:200], f), f"jac {f}")
    for f in golden_meta["g3_trailing_fixed_frames"]:
        jr = g3[f"g3_jac_{f}"]
        assert np.abs(arm.jacobian(q[:jr.shape[0]], f, global_pose=None) - jr).max() < 1e-12
    lp, gp = g3["g3_local_pose"], g3["g3_global_pose"]
    assert np.abs(arm.jacobian(q[:128], "tool_frame", local_pose=lp) - g3["g3_jac_tool_frame_local"]).max() < 1e-12
    assert np.abs(arm.jacobian(q[:128], "tool_frame", global_pose=gp) - g3["g3_jac_tool_frame_global"]).max() < 1e-12
    assert arm.jacobian(q[0], "tool_frame").shape == (6, 7)                       # default global_pose=False (Q2)
    assert np.array_equal(arm.jacobian(q[:4], "base_link"), np.zeros((4, 6, 7)))  # arm.py:455-457


def test_configurations_sampled_on_the_device(kinova, torch_cuda):
    """scenes.sample_q_device: q ~ U(limits) as a CUDA tensor (no PCIe); the mask of those q equals the oracle's on the same values."""
    from numbotics_amd.scenes import sample_q_device
    arm, chain, _ = kinova
    lim = np.asarray(chain.joint_limits, dtype=np.float64)
    q = sample_q_device(chain, 30000, seed=5)
    assert q.is_cuda and q.dtype == torch_cuda.float64 and q.shape == (30000, chain.dof)
    qh = q.cpu().numpy()
    assert (qh >= lim[:, 0]).all() and (qh <= lim[:, 1]).all()
    assert np.abs(qh.mean(axis=0) - lim.mean(axis=1)).max() < 0.05 * (lim[:, 1] - lim[:, 0]).max()
    assert torch_cuda.equal(q, sample_q_device(chain, 30000, seed=5)) and not torch_cuda.equal(q, sample_q_device(chain, 30000, seed=6))
    buf = torch_cuda.empty((30000, chain.dof), dtype=torch_cuda.float64, device="cuda")
    assert sample_q_device(chain, 30000, seed=5, out=buf).data_ptr() == buf.data_ptr() and torch_cuda.equal(buf, q)
    mask = arm.in_collision(q)
    assert mask.is_cuda
    assert np.array_equal(mask.cpu().numpy(), Oracle(arm.scene_model()).validity(qh, 0.0))


def test_fk_and_jacobian_q_paths_agree(kinova, g3, torch_cuda):
    """k_fk / k_jacobian_reg read q straight into registers for n_q <= 8; the LDS-staged form (robots beyond 8 DoF, or the
    ``fk_lds_q`` switch) must give the same bits, ragged block tails included."""
    from numbotics_amd._lib import debug_option
    arm, chain, _ = kinova
    orc = Oracle(arm._kin)
    q = g3["g3_q"]
    for B in (1, 63, 64, 65, 1000):
        T, J = arm.forward_kinematics(q[:B], "tool_frame"), arm.jacobian(q[:B], "tool_frame")
        with debug_option("fk_lds_q", 1):
            T2, J2 = arm.forward_kinematics(q[:B], "tool_frame"), arm.jacobian(q[:B], "tool_frame")
        assert_bitwise(T, T2, f"fk q paths B={B}")
        assert_bitwise(J, J2, f"jacobian q paths B={B}")
        assert_bitwise(T, orc.fk(q[:B], "tool_frame"), f"fk B={B}")
        assert_bitwise(J, orc.jacobian(q[:B], "tool_frame"), f"jacobian B={B}")


@pytest.mark.parametrize("margins", [True, False], ids=["bullet", "sharp"])
@pytest.mark.parametrize("scene", ["c1", "c2", "c3"])
def test_validity_mask_bitwise(fresh_world, scene, margins, torch_cuda):
    """c1: the arm alone (self-collision pairs only, no world shape at all), c2 / c3: BASELINE configs 2 and 3; in the default
    mode (Bullet's shape margins, ``Arm(chain)``) and with sharp shapes (``bullet_margins=False``)."""
    arm, chain, obs = build_scene(scene, bullet_margins=margins)
    assert arm.bullet_margins == margins
    sm = arm.scene_model()
    assert ((sm.rshape_param[:, 3] > 0).sum() >= 9) == margins
    assert (sm.n_wshapes == 0) == (scene == "c1")
    orc = Oracle(sm)
    q = sample_q(chain, 20000, seed=2)
    for thr in (0.0, 1e-6, 0.02, -0.005):
        mask = arm.in_collision(q, thr)
        ref = orc.validity(q, thr)
        assert mask.dtype == bool and mask.shape == (20000,)
        assert np.array_equal(mask, ref), (scene, thr, int((mask != ref).sum()))
        with fused_path():
            assert np.array_equal(arm.in_collision(q[:3000], thr), ref[:3000]), (scene, thr, "fused")
    assert 0.005 < orc.validity(q, 0.0).mean() < 0.9
    # packed bit mask == bytes
    _, dev = arm._scene_device()
    words = dev.validity(q, 0.0, packed=True)
    unpacked = ((words.view(np.uint64)[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).reshape(-1)
    assert np.array_equal(unpacked[:20000], orc.validity(q, 0.0)) and not unpacked[20000:].any()
    # scalar contract
    assert isinstance(arm.in_collision(q[0]), bool) and arm.in_collision(q[0]) == bool(ref_first(orc, q[0]))
    with pytest.raises(ValueError):
        arm.in_collision(q[:, :6])
    for B in (1, 63, 64, 65, 130):
        assert np.array_equal(arm.in_collision(q[:B]), orc.validity(q[:B]))


def test_fused_and_two_kernel_paths_agree(fresh_world, torch_cuda):
    """The fused kernel (NBK_TWO_KERNEL_MIN_B raises the hand-over size; it is also what a nbk_validity_batch_ws call without
    a workspace runs) and broadphase + narrowphase: same predicate, same bits -- also with a caller-owned workspace."""
    import os
    torch = torch_cuda
    arm, chain, obs = build_scene("c3")
    _, dev = arm._scene_device()
    orc = Oracle(arm.scene_model())
    q = sample_q(chain, 40000, seed=12)
    for thr in (0.0, 0.01, -0.002):
        big = dev.validity(q, thr)                                        # broadphase + narrowphase
        small = np.concatenate([dev.validity(q[i:i + 4096], thr) for i in range(0, 40000, 4096)])
        with debug_option("two_kernel_min_b", 1000000):
            assert dev.validity_workspace_bytes(4096) == 0
            fused = np.concatenate([dev.validity(q[i:i + 4096], thr) for i in range(0, 40000, 4096)])   # fused kernel
        assert np.array_equal(big, small) and np.array_equal(big, fused)
        assert np.array_equal(big, orc.validity(q, thr, nthreads=8))
    need = dev.validity_workspace_bytes(40000)
    assert need > 0 and dev.validity_workspace_bytes(100) > 0 and dev.validity_workspace_bytes(0) == 0
    ws = torch.empty((need,), dtype=torch.uint8, device="cuda")
    assert np.array_equal(dev.validity(q, 0.0, workspace=ws), big if thr == 0.0 else dev.validity(q, 0.0))
    words = dev.validity(q, 0.0, packed=True, workspace=ws)
    from numbotics_amd.parallel import unpack_mask
    assert np.array_equal(unpack_mask(words, 40000), dev.validity(q, 0.0))
    from numbotics_amd._lib import NbkError
    with pytest.raises(NbkError):
        dev.validity(q, 0.0, workspace=ws[:1024])


def ref_first(orc, q0):
    return orc.validity(q0.reshape(1, -1))[0]


@pytest.mark.parametrize("margins", [True, False], ids=["bullet", "sharp"])
@pytest.mark.parametrize("scene", ["c2", "c3"])
def test_distances_bitwise(fresh_world, scene, margins, torch_cuda):
    arm, chain, obs = build_scene(scene, bullet_margins=margins)
    sm = arm.scene_model()
    orc = Oracle(sm)
    q = sample_q(chain, 3000, seed=3)
    D = arm.pair_distances(q)
    Dref = orc.pair_distances(q)
    assert_bitwise(D, Dref, "pair distances")
    dmin, idx = arm.closest_distance(q)
    dref, iref = orc.closest(q)
    assert_bitwise(dmin, dref, "closest distance")
    assert np.array_equal(idx, iref)
    # validity predicate == (closest distance < thr) away from the threshold
    for thr in (0.0, 0.01):
        assert np.array_equal(arm.in_collision(q, thr), dmin < thr)
    # Proximity records of the scalar API
    _, dev = arm._scene_device()
    Dw, W = dev.pair_distances(q[:300], witness=True)
    Dr, Wr = orc.pair_distances(q[:300], witness=True)
    assert_bitwise(Dw, Dr, "distances (witness kernel)")
    assert_bitwise(W, Wr, "witness points")
    prox = arm.collisions(q[0])
    assert len(prox) == sm.n_pairs
    best = arm.closest_to(q[0])
    assert best.distance == dref[0]
    with pytest.raises(ValueError):
        arm.collisions(q[:2])


def test_shape_zoo_distances(fresh_world, torch_cuda):
    """Every primitive pair class, incl. capsule / sphere / cylinder / plane obstacles and margins."""
    from numbotics_amd.physics import GraphChain, Cube, Cuboid, Sphere, Capsule, Cylinder, Plane
    from numbotics_amd.robots import Arm
    from conftest import URDF
    chain = GraphChain.from_urdf(URDF)
    arm = Arm(chain)
    rng = np.random.default_rng(9)
    from geom_truth import random_pose
    obs = [Cube(0.0, 0.2, position=np.array([0.6, 0.1, 0.3])),
           Cuboid(0.0, np.array([0.1, 0.3, 0.05]), pose=random_pose(rng, 0.7)),
           Sphere(0.0, 0.15, position=np.array([-0.5, 0.3, 0.6])),
           Capsule(0.0, 0.08, 0.4, pose=random_pose(rng, 0.7)),
           Cylinder(0.0, 0.12, 0.3, pose=random_pose(rng, 0.7)),
           Plane(0.0, np.array([0.0, 0.0, 1.0]), position=np.array([0.0, 0.0, -0.05])),
           Cuboid(0.0, np.array([0.2, 0.2, 0.2]), position=np.array([0.0, -0.7, 0.5]), collision_margin=0.04)]
    sm = arm.scene_model()
    assert sm.n_wshapes == 7
    orc = Oracle(sm)
    q = sample_q(chain, 12000, seed=5)
    assert_bitwise(arm.pair_distances(q[:2000]), orc.pair_distances(q[:2000]), "zoo distances")
    for thr in (0.0, 0.03, -0.004):
        ref = orc.validity(q, thr, nthreads=8)
        assert np.array_equal(arm.in_collision(q, thr), ref)                    # broadphase + narrowphase kernels
        with fused_path():
            assert np.array_equal(arm.in_collision(q[:2000], thr), ref[:2000])      # fused kernel


def test_one_wave_per_edge_kernel_still_agrees(fresh_world, torch_cuda):
    """k_edges (one wave per edge) is no longer the default for small edge counts; NBK_EDGE_BATCH_MIN_E brings it back."""
    import os
    arm, chain, obs = build_scene("c3")
    orc = Oracle(arm.scene_model())
    _, dev = arm._scene_device()
    q = sample_q(chain, 60, seed=44)
    with debug_option("edge_batch_min_e", 1000):
        for mode in ("connect", "steer"):
            ok, end, ns = dev.edge_validity(q[:30], q[30:], 0.02, 1.0, mode=mode)
            okr, endr, nsr = orc.edge_validity(q[:30], q[30:], 0.02, 1.0, mode=mode)
            assert np.array_equal(ok, okr) and np.array_equal(ns, nsr)
            assert_bitwise(end, endr, "k_edges end states")


@pytest.mark.parametrize("margins", [True, False], ids=["bullet", "sharp"])
@pytest.mark.parametrize("mode", ["connect", "steer"])
def test_edge_validity(fresh_world, mode, margins, torch_cuda):
    from numbotics_amd.planning.sampling_based import ConnectorParams, DiscreteConnector
    arm, chain, obs = build_scene("c3", bullet_margins=margins)
    orc = Oracle(arm.scene_model())
    rng = np.random.default_rng(6)
    lim = chain.joint_limits
    E = 300
    s = rng.uniform(lim[:, 0], lim[:, 1], (E, 7))
    g = s + rng.normal(scale=0.6, size=(E, 7))
    s[0] = g[0]                                  # degenerate edge -> invalid
    for res, maxd in ((0.05, np.pi), (0.01, 1.0)):
        conn = DiscreteConnector(ConnectorParams(resolution=res, max_distance=maxd, arm=arm))
        _, dev = arm._scene_device()
        ok, end, ns = dev.edge_validity(s, g, res, maxd, mode=mode)
        okr, endr, nsr = orc.edge_validity(s, g, res, maxd, mode=mode)
        assert np.array_equal(ok, okr) and np.array_equal(ns, nsr)
        assert_bitwise(end, endr, "edge end states")
        assert not ok[0] and ns[0] == 0
        assert 0 < ok.mean() < 1
        # scalar contract: goal copy / traj(T_f) / None
        for e in range(12):
            r = getattr(conn, mode)(s[e], g[e])
            if okr[e]:
                assert r is not None and np.array_equal(r, endr[e])
            else:
                assert r is None


def test_edge_matches_per_sample_walk(fresh_world, torch_cuda):
    """The batched edge kernel equals the reference algorithm run sample by sample with the scalar
    validity checker `not arm.in_collision(q)` (README.md:107)."""
    from numbotics_amd.planning.sampling_based import ConnectorParams, DiscreteConnector
    arm, chain, obs = build_scene("c2")
    rng = np.random.default_rng(8)
    lim = chain.joint_limits
    walk = DiscreteConnector(ConnectorParams(resolution=0.1, max_distance=np.pi,
                                             validity_checker=lambda q: not arm.in_collision(q)))
    fast = DiscreteConnector(ConnectorParams(resolution=0.1, max_distance=np.pi, arm=arm))
    n_none = 0
    for _ in range(25):
        a, b = rng.uniform(lim[:, 0], lim[:, 1]), rng.uniform(lim[:, 0], lim[:, 1])
        for mode in ("connect", "steer"):
            r1, r2 = getattr(walk, mode)(a, b), getattr(fast, mode)(a, b)
            assert (r1 is None) == (r2 is None)
            if r1 is not None:
                assert np.array_equal(r1, r2)
            else:
                n_none += 1
    assert n_none > 0


@pytest.mark.parametrize("margins", [True, False], ids=["bullet", "sharp"])
def test_full_size_properties(fresh_world, margins, torch_cuda):
    """BASELINE config 2 at full size (1e6 q): size-independent properties + oracle on a slice."""
    torch = torch_cuda
    arm, chain, obs = build_scene("c2", bullet_margins=margins)
    orc = Oracle(arm.scene_model())
    B = 1_000_000
    q = sample_q(chain, B, seed=1)
    tq = torch.from_numpy(q).cuda()
    m1 = arm.in_collision(tq)
    assert m1.is_cuda and m1.dtype == torch.bool and m1.shape == (B,)
    m1 = m1.cpu().numpy()
    # idempotence / determinism
    assert np.array_equal(m1, arm.in_collision(tq).cpu().numpy())
    # sharding invariance: any split point gives the same bits (what the multi-GPU path relies on)
    cut = 333_333
    m2 = np.concatenate([arm.in_collision(tq[:cut]).cpu().numpy(), arm.in_collision(tq[cut:]).cpu().numpy()])
    assert np.array_equal(m1, m2)
    # permutation equivariance
    perm = np.random.default_rng(0).permutation(B)
    assert np.array_equal(arm.in_collision(tq[torch.from_numpy(perm).cuda()]).cpu().numpy(), m1[perm])
    # oracle on a 50k slice spread over the batch
    sl = np.arange(0, B, 20)
    assert np.array_equal(m1[sl], orc.validity(q[sl], nthreads=8))
    # threshold monotonicity
    m_hi = arm.in_collision(tq, 0.05).cpu().numpy()
    assert (m_hi | ~m1).all() and m_hi.sum() > m1.sum()


TREE_URDF = __import__("os").path.join(__import__("os").path.dirname(__file__), "models", "tree_gripper.urdf")


def test_tree_robot_with_prismatic_joints(fresh_world, torch_cuda):
    """Branching chain (saved frames), prismatic joints, non-z axes: FK / Jacobian / masks / distances / edges,
    through the fused kernel, the LDS broadphase and the register broadphase."""
    import os
    from numbotics_amd.physics import GraphChain, Cube, Sphere, Plane, Capsule
    from numbotics_amd.robots import Arm
    chain = GraphChain.from_urdf(TREE_URDF)
    arm = Arm(chain)
    assert chain.dof == 5
    obs = [Cube(0.0, 0.08, position=np.array([0.35, 0.0, 0.55])), Sphere(0.0, 0.05, position=np.array([0.2, 0.2, 0.4])),
           Plane(0.0, np.array([0.0, 0.0, 1.0]), position=np.array([0.0, 0.0, -0.01])),
           Capsule(0.0, 0.03, 0.3, position=np.array([-0.2, 0.1, 0.6]))]
    arm.remove_collision_pair("base", obs[2].name)          # the base sits on the ground
    sm = arm.scene_model()
    assert arm._kin.n_joints == 5 and sm.n_pairs > 20
    orc_k, orc = Oracle(arm._kin), Oracle(sm)
    q = sample_q(chain, 12000, seed=21)
    for f in ("tip_a", "tip_b", "wrist_cam", "finger_b", "column"):
        assert_bitwise(arm.forward_kinematics(q[:500], f), orc_k.fk(q[:500], f), f"tree fk {f}")
        assert_bitwise(arm.jacobian(q[:500], f), orc_k.jacobian(q[:500], f), f"tree jac {f}")
    # FK / Jacobian agree with finite differences (prismatic columns are pure translations)
    J = arm.jacobian(q[:4], "tip_b")
    h = 1e-6
    for j in range(5):
        qp, qm = q[:4].copy(), q[:4].copy()
        qp[:, j] += h; qm[:, j] -= h
        num = (arm.forward_kinematics(qp, "tip_b")[:, :3, 3] - arm.forward_kinematics(qm, "tip_b")[:, :3, 3]) / (2 * h)
        assert np.abs(num - J[:, :3, j]).max() < 1e-8
    for thr in (0.0, 0.01):
        ref = orc.validity(q, thr, nthreads=8)
        assert np.array_equal(arm.in_collision(q, thr), ref)                        # register broadphase
        with fused_path():
            assert np.array_equal(np.concatenate([arm.in_collision(q[i:i + 4000], thr) for i in range(0, 12000, 4000)]), ref)  # fused
        with debug_option("no_reg_broad", 1):
            assert np.array_equal(arm.in_collision(q, thr), ref)                    # LDS broadphase
    assert 0.02 < orc.validity(q).mean() < 0.98
    assert_bitwise(arm.pair_distances(q[:1000]), orc.pair_distances(q[:1000]), "tree distances")
    _, dev = arm._scene_device()
    s, g = q[:200], q[200:400]
    ok, end, ns = dev.edge_validity(s, g, 0.02, 1.0, mode="steer")
    okr, endr, nsr = orc.edge_validity(s, g, 0.02, 1.0, mode="steer")
    assert np.array_equal(ok, okr) and np.array_equal(ns, nsr)
    assert_bitwise(end, endr, "tree edge ends")


def test_many_shapes_fall_back_to_lds_broadphase(fresh_world, torch_cuda, tmp_path):
    """More than 16 robot primitives: k_broad (centres in LDS) instead of k_broad_reg."""
    from numbotics_amd.physics import GraphChain, Cube
    from numbotics_amd.robots import Arm
    n = 9
    parts = ['<?xml version="1.0"?><robot name="snake">', '<link name="l0"><collision><geometry><sphere radius="0.03"/></geometry></collision></link>']
    for i in range(1, n + 1):
        parts.append(f'<link name="l{i}"><collision><origin xyz="0 0 0.05"/><geometry><cylinder radius="0.02" length="0.1"/></geometry></collision>'
                     f'<collision><origin xyz="0 0 0.1"/><geometry><sphere radius="0.025"/></geometry></collision></link>')
        ax = ["1 0 0", "0 1 0", "0 0 1"][i % 3]
        parts.append(f'<joint name="j{i}" type="revolute"><origin xyz="0 0 {0.0 if i == 1 else 0.1}"/><parent link="l{i-1}"/><child link="l{i}"/>'
                     f'<axis xyz="{ax}"/><limit lower="-1.5" upper="1.5" effort="1" velocity="1"/></joint>')
    parts.append('</robot>')
    path = tmp_path / "snake.urdf"
    path.write_text("\n".join(parts))
    chain = GraphChain.from_urdf(str(path))
    arm = Arm(chain)
    cube = Cube(0.0, 0.1, position=np.array([0.25, 0.0, 0.4]))
    sm = arm.scene_model()
    assert sm.n_rshapes == 2 * n + 1 and sm.n_rshapes > 16
    orc = Oracle(sm)
    q = sample_q(chain, 10000, seed=5)
    ref = orc.validity(q, 0.0, nthreads=8)
    assert np.array_equal(arm.in_collision(q), ref) and 0.05 < ref.mean() < 0.95
    assert np.array_equal(arm.in_collision(q[:3000]), ref[:3000])


def test_iris_inner_steps(fresh_world, torch_cuda):
    """The two collision-bound steps of IrisSolver.seperating_hyperplanes (safe_sets.py:124-134,186-201) on the
    device, against the same loop run configuration by configuration on the oracle."""
    from numbotics_amd.planning import collision_mask, counter_example_bisection
    arm, chain, obs = build_scene("c2")
    orc = Oracle(arm.scene_model())
    M = 10071                                  # upstream's first-iteration sample count (safe_sets.py:176-182)
    pts = sample_q(chain, M, seed=31)
    seed_q = np.zeros(7)
    assert not arm.in_collision(seed_q, 1e-6)
    mask = collision_mask(arm, pts, 1e-6)
    assert np.array_equal(mask, orc.validity(pts, 1e-6, nthreads=8)) and mask.any()
    col = pts[mask]
    hi = counter_example_bisection(arm, seed_q, col, 15, 1e-6)
    # reference loop, one sample at a time, on the oracle
    for i in range(0, col.shape[0], max(1, col.shape[0] // 40)):
        lo_i, hi_i = seed_q.copy(), col[i].copy()
        for _ in range(15):
            mid = (lo_i + hi_i) / 2.0
            if orc.validity(mid[None], 1e-6)[0]:
                hi_i = mid
            else:
                lo_i = mid
        assert np.array_equal(hi_i, hi[i])
    assert np.asarray(arm.in_collision(hi, 1e-6)).all()          # the returned ends are still colliding
    # the same search with device tensors in and out: every round stays on the device (no host copy), bit-equal to the host loop
    torch = torch_cuda
    col_d = torch.from_numpy(col).cuda()
    hi_d = counter_example_bisection(arm, seed_q, col_d, 15, 1e-6)
    assert hi_d.is_cuda and np.array_equal(hi_d.cpu().numpy(), hi)
    hi_g = counter_example_bisection(arm, seed_q, col_d, 15, 1e-6, graph=True)       # rounds 3..15 replayed from one hipGraph
    assert np.array_equal(hi_g.cpu().numpy(), hi)


def _tree_scene():
    from numbotics_amd.physics import GraphChain, Cube, Sphere, Capsule
    from numbotics_amd.robots import Arm
    chain = GraphChain.from_urdf(TREE_URDF)
    arm = Arm(chain)
    obs = [Cube(0.0, 0.08, position=np.array([0.35, 0.0, 0.55])), Sphere(0.0, 0.05, position=np.array([0.2, 0.2, 0.4])),
           Capsule(0.0, 0.03, 0.3, position=np.array([-0.2, 0.1, 0.6]))]
    return arm, chain, obs


@pytest.mark.parametrize("scene", ["c2", "c3", "tree"])
def test_proximity_jacobian_rows(fresh_world, scene, torch_cuda):
    """batched Proximity records + jacobian_proximity rows (arm.py:607-632) in one launch,
    bit-exact against the oracle; the scalar API equals the composition the reference writes down."""
    import ctypes as C
    from numbotics_amd.math import trans_mat
    from numbotics_amd.planning.safe_sets import distance_and_gradient
    arm, chain, obs = _tree_scene() if scene == "tree" else build_scene(scene)
    sm = arm.scene_model()
    orc = Oracle(sm)
    M = 10071 if scene == "c2" else 1500                       # config 5's sample count on the benchmark scene
    q = sample_q(chain, M, seed=41)
    d, w, rows = arm.proximity_jacobians(q)
    dr, wr, rr = orc.proximity_jacobian(q)
    assert_bitwise(d, dr, "proximity distances")
    assert_bitwise(w, wr, "proximity witnesses")
    assert_bitwise(rows, rr, "proximity jacobian rows")
    assert rows.shape == (M, sm.n_pairs, chain.dof) and np.abs(rows).max() > 0.1
    # torch in -> torch out, same bits
    qt = torch_cuda.from_numpy(q[:256]).cuda()
    dt, wt, rt = arm.proximity_jacobians(qt)
    assert rt.is_cuda and np.array_equal(rt.cpu().numpy(), rows[:256])
    # scalar API of the reference: rows for one obstacle / for the chain itself, built from the device Jacobian
    target = obs[0]
    J = np.atleast_2d(arm.jacobian_proximity(q[0], target))
    prox = arm.distance_to(q[0], target)
    assert J.shape == (len(prox), chain.dof) and len(prox) >= 1
    for i, p in enumerate(prox):
        n = p.normal_target_to_subject
        ref = n @ arm.jacobian(q[0], p.subject._name, global_pose=trans_mat(pos=p.position_on_subject))[:3]
        assert np.abs(J[i] - ref).max() < 1e-12
    Js = np.atleast_2d(arm.jacobian_proximity(q[1], chain))
    prox = arm.distance_to(q[1], chain)
    assert Js.shape[0] == len(prox) >= 1
    for i, p in enumerate(prox[:6]):
        n = p.normal_target_to_subject
        ref = n @ arm.jacobian(q[1], p.subject._name, global_pose=trans_mat(pos=p.position_on_subject))[:3] \
            - n @ arm.jacobian(q[1], p.target._name, global_pose=trans_mat(pos=p.position_on_target))[:3]
        assert np.abs(Js[i] - ref).max() < 1e-12
    # the NLP constraint and its Jacobian for a body pair (safe_sets.py:86-121), batched
    link = arm.distance_to(q[0], target)[0].subject
    dist, grad = distance_and_gradient(arm, q[:64], link, target)
    sel = arm._pair_selection(sm, target, link)
    assert np.array_equal(dist, dr[:64][:, sel].min(axis=1))
    k = dr[:64][:, sel].argmin(axis=1)
    assert np.array_equal(grad, rr[:64][:, sel][np.arange(64), k])
    # C boundary: status codes
    from numbotics_amd import _lib
    lib = _lib.load()
    _, dev = arm._scene_device()
    st = C.c_void_p(torch_cuda.cuda.current_stream().cuda_stream)
    assert lib.nbk_proximity_jacobian_batch(dev._h, qt.data_ptr(), 256, dt.data_ptr(), wt.data_ptr(), None, st) == -1
    assert lib.nbk_proximity_jacobian_batch(dev._h, qt.data_ptr(), 0, None, None, None, st) == 0


def test_inverse_kinematics(kinova, torch_cuda):
    """batched Levenberg-Marquardt IK (arm.py:464-552), all problems inside one launch;
    iterates, flags, residuals and step counts bit-exact against the oracle (tests/test_oracle_ik.py ties the oracle
    to the NumPy restatement of the reference loop)."""
    import ctypes as C
    arm, chain, obs = kinova
    orc = Oracle(arm._kin)
    _ = arm._kin_device()
    rng = np.random.default_rng(3)
    n = 5000
    qt = sample_q(chain, n, seed=13, margin=0.2)
    limits = np.asarray(chain.joint_limits, dtype=np.float64)
    dev = arm._kin_device()
    for frame, spread, lim in (("tool_frame", 0.5, None), ("tool_frame", 3.0, limits), ("bracelet_link", 1.0, None)):
        pose = orc.fk(qt, frame)
        q0 = qt + rng.uniform(-spread, spread, qt.shape)
        ok, q, nrm, it = dev.ik(pose, q0, frame, limits=lim)
        okr, qr, nrmr, itr = orc.ik(pose, q0, frame, limits=lim)
        assert np.array_equal(ok, okr) and np.array_equal(it, itr)
        assert_bitwise(q, qr, f"ik q {frame}")
        assert_bitwise(nrm, nrmr, f"ik residual {frame}")
        assert ok.mean() > (0.9 if spread < 1.0 else 0.2)
        T = arm.forward_kinematics(q[ok], frame)
        assert np.abs(T[:, :3, 3] - pose[ok][:, :3, 3]).max() < 1e-6
    # upstream's signature and return shapes (arm.py:548-552)
    pose = orc.fk(qt[:8], "tool_frame")
    ok1, q1 = arm.inverse_kinematics(pose[0], qt[0] + 0.1, "tool_frame")
    assert ok1.shape == (1,) and q1.shape == (7,) and ok1[0]
    okb, qb = arm.inverse_kinematics(pose, qt[:8] + 0.1, "tool_frame", use_limits=True)
    assert okb.shape == (8,) and qb.shape == (8, 7) and okb.all()
    okc, qc = arm.inverse_kinematics(pose[0], qt[:8] + 0.1, "tool_frame")            # one pose, many starts
    assert okc.shape == (8,) and qc.shape == (8, 7)
    okd, qd = arm.inverse_kinematics(pose.reshape(2, 4, 4, 4), (qt[:8] + 0.1).reshape(2, 4, 7), "tool_frame", use_com=True)
    assert okd.shape == (8,) and qd.shape == (2, 4, 7)
    Tc = arm.forward_kinematics(qd[0, 0], "tool_frame", use_com=True)
    assert np.abs(Tc[:3, 3] - pose[0][:3, 3]).max() < 1e-6
    qt_t = torch_cuda.from_numpy(qt[:8] + 0.1).cuda()
    okt, qtt = arm.inverse_kinematics(torch_cuda.from_numpy(pose).cuda(), qt_t, "tool_frame")
    assert qtt.is_cuda and np.array_equal(qtt.cpu().numpy(), arm.inverse_kinematics(pose, qt[:8] + 0.1, "tool_frame")[1])
    with pytest.raises(ValueError):
        arm.inverse_kinematics(pose[0], qt[0], "nope")
    with pytest.raises(ValueError):
        arm.inverse_kinematics(pose[0], qt[0][:6], "tool_frame")
    with pytest.raises(ValueError):
        arm.inverse_kinematics(pose[0], qt[0], "tool_frame", max_iter=0)
    with pytest.raises(ValueError):
        arm.inverse_kinematics(pose[:4], qt[:8], "tool_frame")
    from numbotics_amd import _lib
    lib = _lib.load()
    assert lib.nbk_ik_batch(dev._h, None, qt_t.data_ptr(), 8, None, 0, None, None, 1e-6, 100, 15, None, None, None, None, None) == -1


def test_inverse_kinematics_tree_robot(fresh_world, torch_cuda):
    arm, chain, obs = _tree_scene()
    orc = Oracle(arm._kin)
    dev = arm._kin_device()
    rng = np.random.default_rng(5)
    qt = sample_q(chain, 2000, seed=17)
    for frame in ("tip_b", "wrist_cam"):          # 3-4 joints on the path: the damping carries the rank-deficient system
        pose = orc.fk(qt, frame)
        q0 = qt + rng.uniform(-0.2, 0.2, qt.shape)
        ok, q, nrm, it = dev.ik(pose, q0, frame)
        okr, qr, nrmr, itr = orc.ik(pose, q0, frame)
        assert np.array_equal(ok, okr) and np.array_equal(it, itr)
        assert_bitwise(q, qr, f"tree ik q {frame}")
        assert_bitwise(nrm, nrmr, f"tree ik residual {frame}")
        # joints off the path are never touched
        path_cols = set(int(arm._kin.joint_qidx[k]) for k in arm._kin.frames[frame].path)
        for j in range(chain.dof):
            if j not in path_cols:
                assert np.array_equal(q[:, j], q0[:, j])


def test_batched_prm_on_the_device(fresh_world, torch_cuda):
    """every candidate edge of a PRM roadmap in one connect_batch call; the accepted edges are
    exactly those the scalar connector.connect accepts, which in turn is the oracle's edge predicate."""
    from numbotics_amd.planning.sampling_based import (ConnectorParams, DiscreteConnector, EuclideanSpace, PlannerParams, PRM)
    arm, chain, obs = build_scene("c3")
    lim = np.asarray(chain.joint_limits, dtype=np.float64)
    space = EuclideanSpace(lim[:, 0].copy(), lim[:, 1].copy())
    conn = DiscreteConnector(ConnectorParams(resolution=0.05, max_distance=np.pi, arm=arm))
    rng = np.random.default_rng(8)
    free = [s for s in sample_q(chain, 4000, seed=5) if conn.is_valid(s)]
    start, goal = free[0], free[1]
    params = PlannerParams(max_iters=600, k_nearest=10, goal_bias=0.02)
    samples = [goal.copy() if rng.random() < params.goal_bias else s for s in free[2:2 + params.max_iters]]
    prm = PRM(space, conn, params)
    prm.add_start(start)
    prm.add_goal(goal)
    prm.plan(samples)
    assert prm.n_candidate_edges > 5000 and 0.02 < prm.edges.shape[0] / prm.n_candidate_edges < 0.98
    V, nodes, cand, dist = prm.candidate_edges(samples)
    live = dist > np.finfo(np.float32).eps
    orc = Oracle(arm.scene_model())
    ok_ref, _, _ = orc.edge_validity(nodes[cand[live, 0]], nodes[cand[live, 1]], 0.05, np.pi, mode="connect", dist=dist[live], nthreads=8)
    accepted = set(map(tuple, prm.edges.tolist()))
    assert accepted == set(map(tuple, cand[live][ok_ref].tolist()))
    for a, b in cand[live][:40]:                                   # the scalar contract agrees edge by edge
        assert (conn.connect(nodes[a], nodes[b], distance_func=space.distance) is not None) == ((int(a), int(b)) in accepted)
    path = prm.solution()
    if path is not None:
        assert path[0].id == "v_0" and path[-1].id == "g_0"
        for a, b in zip(path[:-1], path[1:]):
            assert conn.connect(a.state, b.state) is not None or conn.connect(b.state, a.state) is not None


def test_rrt_star_on_the_device(fresh_world, torch_cuda):
    """SURVEY.md 8(f) rank 2, RRT*: the steer and the per-iteration batch of neighbour connects run on the device; the
    tree (vertices, parents, costs, rewires, goal links) is the one the same planner grows over the CPU oracle's edge
    predicate."""
    from numbotics_amd.planning.sampling_based import (ConnectorParams, DiscreteConnector, EuclideanSpace, PlannerParams, RRTStar)
    # the start / goal / sample script below was laid out on the sharp shapes (the goal is reached and >= 3 rewires happen there);
    # tree equality is asserted first and holds in either shape mode
    arm, chain, obs = build_scene("c3", bullet_margins=False)
    lim = np.asarray(chain.joint_limits, dtype=np.float64)
    space = EuclideanSpace(lim[:, 0].copy(), lim[:, 1].copy())
    res, maxd = 0.05, 1.0
    conn = DiscreteConnector(ConnectorParams(resolution=res, max_distance=maxd, arm=arm))
    orc = Oracle(arm.scene_model())

    class OracleConnector:
        def is_valid(self, s):
            return not bool(orc.validity(np.asarray(s)[None], 0.0)[0])

        def steer(self, a, b, distance_func):
            d = distance_func(a, b)
            if d <= np.finfo(np.float32).eps:
                return None
            ok, end, _ = orc.edge_validity(a[None], b[None], res, maxd, mode="steer", dist=np.array([d]))
            return np.copy(end[0]) if bool(ok[0]) else None

        def connect_batch(self, A, B, dist=None):
            return orc.edge_validity(A, B, res, maxd, mode="connect", dist=dist)[0]

    rng = np.random.default_rng(21)
    Q = sample_q(chain, 3000, seed=9)
    free = Q[~np.asarray(arm.in_collision(Q, 0.0))]
    start = free[0]
    # a goal several steers away whose straight edge from the start is free: goal-biased samples then walk to it and the
    # last steer lands on it exactly (distance 0 < goal_tolerance: the vertex is dropped, its best parent linked to the goal)
    far = np.linalg.norm(free[1:400] - start, axis=1)
    clear = np.asarray(conn.connect_batch(np.tile(start, (399, 1)), free[1:400], far))
    gi = 1 + int(np.nonzero(clear & (far > 2.5))[0][0])
    goal = free[gi]
    pool = np.delete(free, [0, gi], axis=0)
    params = PlannerParams(max_iters=400, k_nearest=8, goal_bias=0.1, rewire_factor=5.0, goal_tolerance=1e-6)
    samples = [goal.copy() if rng.random() < params.goal_bias else s for s in pool[:params.max_iters]]
    trees = []
    for c in (conn, OracleConnector()):
        t = RRTStar(space, c, params)
        t.add_start(start)
        t.add_goal(goal)
        t.plan(samples)
        trees.append(t)
    dev, ref = trees
    assert dev.states.shape[0] > 50 and dev.n_candidate_edges > 300
    assert_bitwise(dev.states, ref.states, "rrt* vertices")
    assert dev.parent == ref.parent and dev._wpar == ref._wpar and dev.cost == ref.cost
    assert dev.goal_edges == ref.goal_edges and dev.n_rewired == ref.n_rewired and dev.n_candidate_edges == ref.n_candidate_edges
    assert len(dev.goal_edges[0]) >= 1 and dev.n_rewired >= 3 and len(dev.parent) > dev.states.shape[0]
    path = dev.solution()
    assert path is not None and path[0].id == "v_0" and path[-1].id == "g_0"
    for j in range(1, min(40, len(dev.parent))):                      # tree edges hold under the scalar contract too
        if dev.parent[j] >= 0:
            assert conn.connect(dev.states[np.searchsorted(dev.vertex_ids, dev.parent[j])],
                                dev.states[np.searchsorted(dev.vertex_ids, j)], distance_func=space.distance) is not None


@pytest.mark.parametrize("seed", [101, 102, 104, 106, 109, 110, 120])
def test_random_mechanisms_and_scenes(fresh_world, seed, torch_cuda, tmp_path):
    """Fuzz: random trees (3-10 links, mixed joint types, compound links) among random obstacles (all primitive kinds,
    planes, margins): every entry point bit-exact against the oracle, through all three validity paths."""
    import os
    from numbotics_amd.physics import GraphChain
    from numbotics_amd.robots import Arm
    from random_scenes import random_urdf, random_obstacles
    rng = np.random.default_rng(seed)
    n_links = int(rng.integers(3, 11))
    if seed == 120:
        n_links = 19                                  # more than 16 robot primitives: the LDS broadphase serves it
    chain = GraphChain.from_urdf(random_urdf(rng, n_links, str(tmp_path / "fuzz.urdf")))
    if chain.dof == 0:
        pytest.skip("all joints fixed")
    arm = Arm(chain)
    obs = random_obstacles(rng, int(rng.integers(1, 7)))
    sm = arm.scene_model()
    if sm.n_pairs == 0:
        pytest.skip("no collision pairs")
    orc = Oracle(sm)
    lim = np.asarray(chain.joint_limits, dtype=np.float64)
    lim = np.where(np.isfinite(lim), lim, np.sign(lim) * np.pi)
    q = rng.uniform(lim[:, 0], lim[:, 1], (9000, chain.dof))
    for thr in (0.0, float(rng.choice([0.02, -0.003]))):
        ref = orc.validity(q, thr, nthreads=8)
        assert np.array_equal(arm.in_collision(q, thr), ref), f"two-kernel path, thr {thr}"
        with fused_path():
            assert np.array_equal(arm.in_collision(q[:3000], thr), ref[:3000]), f"fused path, thr {thr}"
        with debug_option("no_reg_broad", 1):
            assert np.array_equal(arm.in_collision(q, thr), ref), f"LDS broadphase, thr {thr}"
    d, w, rows = arm.proximity_jacobians(q[:600])
    dr, wr, rr = orc.proximity_jacobian(q[:600])
    assert_bitwise(d, dr, "fuzz distances")
    assert_bitwise(w, wr, "fuzz witnesses")
    assert_bitwise(rows, rr, "fuzz jacobian rows")
    dmin, idx = arm.closest_distance(q[:1500])
    dref, iref = orc.closest(q[:1500])
    assert_bitwise(dmin, dref, "fuzz closest")
    assert np.array_equal(idx, iref)
    _, dev = arm._scene_device()
    s_, g_ = q[:150], q[150:300]
    for mode in ("connect", "steer"):
        ok, end, ns = dev.edge_validity(s_, g_, 0.03, 1.5, mode=mode)
        okr, endr, nsr = orc.edge_validity(s_, g_, 0.03, 1.5, mode=mode)
        assert np.array_equal(ok, okr) and np.array_equal(ns, nsr)
        assert_bitwise(end, endr, "fuzz edge ends")
    # kinematics of the last link
    frame = [n for n in arm._kin.frames][-1]
    orc_k = Oracle(arm._kin)
    assert_bitwise(arm.forward_kinematics(q[:400], frame), orc_k.fk(q[:400], frame), "fuzz fk")
    if len(arm._kin.frames[frame].path):
        assert_bitwise(arm.jacobian(q[:400], frame), orc_k.jacobian(q[:400], frame), "fuzz jacobian")
    assert len(obs) >= 1


@pytest.mark.parametrize("margins", [True, False], ids=["bullet", "sharp"])
@pytest.mark.parametrize("scene", ["c2", "c3"])
def test_contact_transitions_are_resolved_exactly(fresh_world, scene, margins, torch_cuda):
    """The float32 broadphase may only cull what is certainly free: sweep one joint in steps of 1e-10 rad across
    collision / free transitions found by bisection on the oracle, and ask for the same mask bit for bit."""
    arm, chain, obs = build_scene(scene, bullet_margins=margins)
    orc = Oracle(arm.scene_model())
    base = sample_q(chain, 3000, seed=77)
    m0 = orc.validity(base, 0.0, nthreads=8)
    rng = np.random.default_rng(7)
    sweeps = []
    for i in np.nonzero(m0)[0][:400]:
        j = int(rng.integers(0, chain.dof))
        lo, hi = base[i].copy(), base[i].copy()          # hi collides; look for a free value of joint j nearby
        for step in (0.05, 0.2, 0.6):
            lo[j] = base[i][j] + step
            if not orc.validity(lo[None], 0.0)[0]:
                break
        else:
            continue
        for _ in range(60):                              # bisect the transition down to the last bits
            mid = 0.5 * (lo + hi)
            if orc.validity(mid[None], 0.0)[0]:
                hi = mid
            else:
                lo = mid
        fine = np.tile(hi, (400, 1))
        fine[:, j] = hi[j] + (np.arange(400) - 200) * 1e-10
        sweeps.append(fine)
        if len(sweeps) == 24:
            break
    assert len(sweeps) >= 20
    q = np.concatenate(sweeps)                           # 9 600 configurations: the two-kernel path
    ref = orc.validity(q, 0.0, nthreads=8)
    assert 0.2 < ref.mean() < 0.8
    assert np.array_equal(arm.in_collision(q, 0.0), ref)
    with fused_path():
        assert np.array_equal(arm.in_collision(q[:4000], 0.0), ref[:4000])          # fused path
    for thr in (1e-7, -1e-7):
        assert np.array_equal(arm.in_collision(q, thr), orc.validity(q, thr, nthreads=8))


def test_thresholds_at_the_distance_itself(fresh_world, torch_cuda):
    """Positive thresholds within 1e-9 (relative) of a configuration's own clearance: the inflated boolean walk cannot decide
    these in 64 steps when the closest pair is curved, and the distance iteration has to -- on the device (k_narrow_pos and the
    fused kernel) as in the oracle, whose counter confirms the fallback was taken."""
    import ctypes
    from oracle.cpu_oracle import lib as orc_lib
    arm, chain, obs = build_scene("c2")
    orc = Oracle(arm.scene_model())
    base = sample_q(chain, 6000, seed=91)
    dmin = np.asarray(orc.pair_distances(base)).min(axis=1)
    sel = np.nonzero((dmin > 2e-3) & (dmin < 0.25))[0][:250]
    assert len(sel) >= 200
    stats = (ctypes.c_longlong * 8)()
    orc_lib().orc_stats(stats, 1)
    n_checked = 0
    for i in sel:
        q = np.repeat(base[i:i + 1], 64, axis=0)
        for f in (1.0 - 1e-9, 1.0, 1.0 + 1e-9):
            thr = float(dmin[i] * f)
            ref = bool(orc.validity(base[i:i + 1], thr)[0])
            got = np.asarray(arm.in_collision(q, thr))
            assert got.all() == got.any() and bool(got[0]) == ref, (i, f, thr)
            with fused_path():
                assert bool(np.asarray(arm.in_collision(q[:2], thr))[0]) == ref, (i, f, thr, "fused")
            n_checked += 1
    orc_lib().orc_stats(stats, 0)
    assert stats[3] > 0 and stats[4] > 0, ("inflated walks / undecided", stats[3], stats[4], n_checked)


def test_broadphase_boundary_on_sphere_pairs(fresh_world, torch_cuda):
    """Sphere link against sphere obstacles: the shapes fill their bounding spheres, so the contact transition IS the
    boundary of the broadphase's bounding-sphere test -- the place where a non-conservative float32 cull would show."""
    from numbotics_amd.physics import GraphChain, Sphere
    from numbotics_amd.robots import Arm
    from numbotics_amd.utils import Shape
    from conftest import URDF
    chain = GraphChain.from_urdf(URDF)
    arm = Arm(chain)
    rng = np.random.default_rng(12)
    ball_links = [l for l in chain._links if l._collision_shape.shape == Shape.SPHERE]
    assert len(ball_links) == 1
    obs = [Sphere(0.0, float(rng.uniform(0.05, 0.25)), position=rng.uniform(-0.7, 0.7, 3) + np.array([0.0, 0.0, 0.5])) for _ in range(10)]
    for a, b in list(arm.self_collision_pairs()):
        arm.remove_collision_pair(a, b)
    for a, b in list(arm.collision_pairs()):
        if a != ball_links[0] and b != ball_links[0]:
            arm.remove_collision_pair(a, b)
    sm = arm.scene_model()
    assert sm.n_pairs == 10
    orc = Oracle(sm)
    base = sample_q(chain, 20000, seed=78)
    m0 = orc.validity(base, 0.0, nthreads=8)
    assert 0.01 < m0.mean() < 0.9
    sweeps = []
    for i in np.nonzero(m0)[0]:
        j = int(rng.integers(0, 4))
        lo, hi = base[i].copy(), base[i].copy()
        for step in (0.3, 0.8, 1.5):
            lo[j] = base[i][j] + step
            if not orc.validity(lo[None], 0.0)[0]:
                break
        else:
            continue
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            if orc.validity(mid[None], 0.0)[0]:
                hi = mid
            else:
                lo = mid
        fine = np.tile(hi, (400, 1))
        fine[:, j] = hi[j] + (np.arange(400) - 200) * 1e-10
        sweeps.append(fine)
        if len(sweeps) == 30:
            break
    assert len(sweeps) >= 20
    q = np.concatenate(sweeps)
    ref = orc.validity(q, 0.0, nthreads=8)
    assert 0.2 < ref.mean() < 0.8
    assert np.array_equal(arm.in_collision(q, 0.0), ref)
    d = arm.pair_distances(q)
    assert np.abs(d.min(axis=1)).max() < 1e-7            # every configuration sits on the boundary of some pair
    for thr in (1e-9, -1e-9, 1e-6):
        assert np.array_equal(arm.in_collision(q, thr), orc.validity(q, thr, nthreads=8))


def test_large_joint_values_stay_exact(fresh_world, torch_cuda):
    """Angles of thousands of radians (continuous joints) and metres of prismatic travel: the float32 broadphase's slack
    grows with |q| and with the configuration's extent, the masks stay those of the oracle."""
    arm, chain, obs = build_scene("c2")
    orc = Oracle(arm.scene_model())
    rng = np.random.default_rng(21)
    q = sample_q(chain, 9000, seed=79)
    q_big = q + 2.0 * np.pi * rng.integers(-500, 500, q.shape)              # same poses, |q| up to ~3000 rad
    ref = orc.validity(q_big, 0.0, nthreads=8)
    assert np.array_equal(arm.in_collision(q_big, 0.0), ref) and 0.02 < ref.mean() < 0.5
    q_wild = q * rng.choice([1.0, 30.0, 1000.0], size=q.shape)
    assert np.array_equal(arm.in_collision(q_wild, 0.0), orc.validity(q_wild, 0.0, nthreads=8))


def test_large_prismatic_travel_stays_exact(fresh_world, torch_cuda):
    arm2, chain2, obs2 = _tree_scene()
    orc2 = Oracle(arm2.scene_model())
    rng = np.random.default_rng(22)
    q2 = sample_q(chain2, 9000, seed=80)
    q2[:, ::2] *= rng.choice([1.0, 1.0, 40.0], size=q2[:, ::2].shape)       # joints pushed metres / radians past their limits
    assert np.array_equal(arm2.in_collision(q2, 0.0), orc2.validity(q2, 0.0, nthreads=8))


@pytest.mark.parametrize("robot", ["kinova", "tree"])
def test_fk_of_all_links_in_one_sweep(fresh_world, robot, torch_cuda):
    """nbk_fk_frames_batch: every link pose of every configuration from one tree sweep, bit-identical to the per-frame
    kernel and to the oracle."""
    if robot == "kinova":
        arm, chain, obs = build_scene("c1")
    else:
        arm, chain, obs = _tree_scene()
    orc = Oracle(arm._kin)
    q = sample_q(chain, 3000, seed=90)
    T, names = arm.forward_kinematics_all(q)
    assert T.shape == (3000, len(names), 4, 4) and len(names) == len(arm._kin.frames) >= 8
    for i, f in enumerate(names):
        assert_bitwise(T[:, i], orc.fk(q, f), f"all-links fk {f}")
        assert_bitwise(T[:64, i], arm.forward_kinematics(q[:64], f), f"all-links vs per-frame {f}")
    sub = [names[-1], names[0], names[3]]
    Ts, n2 = arm.forward_kinematics_all(q[:100], frames=sub, use_com=True)
    assert n2 == sub
    for i, f in enumerate(sub):
        assert_bitwise(Ts[:, i], arm.forward_kinematics(q[:100], f, use_com=True), f"subset {f}")
    T1, _ = arm.forward_kinematics_all(q[0])
    assert T1.shape == (len(names), 4, 4) and np.array_equal(T1, T[0])
    Tt, _ = arm.forward_kinematics_all(torch_cuda.from_numpy(q[:130]).cuda().reshape(2, 65, chain.dof))
    assert Tt.is_cuda and tuple(Tt.shape) == (2, 65, len(names), 4, 4) and np.array_equal(Tt.cpu().numpy().reshape(130, -1), T[:130].reshape(130, -1))
    with pytest.raises(ValueError):
        arm.forward_kinematics_all(q, frames=["nope"])


def test_device_knn_matches_the_host_scan(torch_cuda):
    """nbk_knn_prefix == the NumPy restatement of the insert-then-query flat L2 index, index for
    index (ties, duplicates, fewer than k points, k up to 64)."""
    from numbotics_amd.planning.sampling_based import knn_prefix
    rng = np.random.default_rng(3)
    for n, dim, k in ((3000, 7, 50), (2000, 2, 8), (1500, 13, 64), (700, 7, 1), (40, 5, 64)):
        x = rng.normal(size=(n, dim))
        assert np.array_equal(knn_prefix(x, k, device=True), knn_prefix(x, k, device=False)), (n, dim, k)
    grid = rng.integers(0, 4, size=(1200, 3)).astype(np.float64)           # many exact ties and duplicate points
    assert np.array_equal(knn_prefix(grid, 20, device=True), knn_prefix(grid, 20, device=False))
    from numbotics_amd import _lib
    assert _lib.load().nbk_knn_prefix(None, 10, 3, 65, None, None) == -1


def test_non_finite_joint_values_count_as_colliding(fresh_world, torch_cuda):
    """NaN / inf joint values: every validity path (fused, two-kernel float32 and float64 broadphase, edges) reports the
    configuration as colliding, like the oracle."""
    import os
    arm, chain, obs = build_scene("c2")
    orc = Oracle(arm.scene_model())
    q = sample_q(chain, 9000, seed=91)
    rng = np.random.default_rng(1)
    bad_rows = rng.choice(9000, 300, replace=False)
    q[bad_rows, rng.integers(0, 7, 300)] = rng.choice([np.nan, np.inf, -np.inf], 300)
    ref = orc.validity(q, 0.0, nthreads=8)
    assert ref[bad_rows].all()
    assert np.array_equal(arm.in_collision(q, 0.0), ref)                       # float32 broadphase
    with fused_path():
        assert np.array_equal(arm.in_collision(q[:3000], 0.0), ref[:3000])         # fused kernel
    for flag in ("f64_broad", "no_reg_broad"):                         # float64 register / LDS broadphase
        with debug_option(flag, 1):
            assert np.array_equal(arm.in_collision(q, 0.0), ref), flag
    _, dev = arm._scene_device()
    s_, g_ = q[:200].copy(), q[200:400].copy()
    ok, end, ns = dev.edge_validity(s_, g_, 0.05, 1.5, mode="connect")
    okr, endr, nsr = orc.edge_validity(s_, g_, 0.05, 1.5, mode="connect")
    finite = np.isfinite(s_).all(axis=1) & np.isfinite(g_).all(axis=1)
    assert np.array_equal(ok[finite], okr[finite]) and not ok[~finite].any() and not okr[~finite].any()


def test_scene_with_hundreds_of_obstacles(fresh_world, torch_cuda):
    """400 world shapes (4 444 pairs): the register broadphase reads its per-call tables from global memory, so the number
    of obstacles is not bounded by LDS; most of them are out of reach and dropped per call by the static reach test."""
    from numbotics_amd.physics import GraphChain
    from numbotics_amd.robots import Arm
    from random_scenes import random_obstacles
    from conftest import URDF
    chain = GraphChain.from_urdf(URDF)
    arm = Arm(chain)
    rng = np.random.default_rng(31)
    obs = random_obstacles(rng, 400, reach=2.5)
    sm = arm.scene_model()
    assert sm.n_wshapes == 400 and sm.n_pairs > 4000
    orc = Oracle(sm)
    q = sample_q(chain, 12000, seed=33)
    for thr in (0.0, 0.02):
        ref = orc.validity(q, thr, nthreads=8)
        assert np.array_equal(arm.in_collision(q, thr), ref)
        with fused_path():
            assert np.array_equal(arm.in_collision(q[:2000], thr), ref[:2000])          # fused kernel
    assert 0.05 < orc.validity(q, 0.0, nthreads=8).mean() < 0.95
    # a threshold at which many more of the 400 shapes come within reach (the queues are sized per threshold), and a batch
    # large enough for several queue tiles: the library's own scratch stays within 1 GiB (+ the float32 tables)
    ref = orc.validity(q, 0.6, nthreads=8)
    assert np.array_equal(arm.in_collision(q, 0.6), ref) and ref.mean() > 0.5
    torch = torch_cuda
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    qb = torch.from_numpy(sample_q(chain, 400000, seed=34)).cuda()
    got = arm.in_collision(qb, 0.0)
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info()[0]
    assert used < (1 << 30) + (200 << 20), used
    sl = np.arange(0, 400000, 37)
    assert np.array_equal(got.cpu().numpy()[sl], orc.validity(qb.cpu().numpy()[sl], 0.0, nthreads=8))
    dmin, idx = arm.closest_distance(q[:300])
    dref, iref = orc.closest(q[:300])
    assert_bitwise(dmin, dref, "closest among 4444 pairs")
    assert np.array_equal(idx, iref)


def test_robot_with_forty_primitives(fresh_world, torch_cuda, tmp_path):
    """More primitives than the LDS-parked kernels can hold: validity (every batch size) and edges still run, through the
    broadphase + narrowphase kernels; the per-pair distance entry points say UNSUPPORTED instead of failing a launch."""
    from numbotics_amd.physics import GraphChain
    from numbotics_amd.robots import Arm
    from numbotics_amd._lib import NbkError
    from random_scenes import random_urdf, random_obstacles
    rng = np.random.default_rng(124)
    chain = GraphChain.from_urdf(random_urdf(rng, 36, str(tmp_path / "big.urdf"), max_back=1))       # a 36-link serial chain, 39 primitives
    arm = Arm(chain)
    obs = random_obstacles(rng, 3)
    sm = arm.scene_model()
    assert sm.n_rshapes >= 25
    orc = Oracle(sm)
    lim = np.asarray(chain.joint_limits, dtype=np.float64)
    lim = np.where(np.isfinite(lim), lim, np.sign(lim) * np.pi)
    q = rng.uniform(lim[:, 0], lim[:, 1], (9000, chain.dof))
    for thr in (0.0, 0.01):
        ref = orc.validity(q, thr, nthreads=8)
        assert np.array_equal(arm.in_collision(q, thr), ref)
        assert np.array_equal(arm.in_collision(q[:100], thr), ref[:100])         # small batch: same kernels
        assert bool(arm.in_collision(q[7], thr)) == bool(ref[7])                   # scalar contract
    _, dev = arm._scene_device()
    for n_e in (5, 200):
        ok, end, ns = dev.edge_validity(q[:n_e], q[n_e:2 * n_e], 0.05, 1.0, mode="steer")
        okr, endr, nsr = orc.edge_validity(q[:n_e], q[n_e:2 * n_e], 0.05, 1.0, mode="steer")
        assert np.array_equal(ok, okr) and np.array_equal(ns, nsr)
        assert_bitwise(end, endr, "big robot edge ends")
    with pytest.raises(NbkError):
        arm.pair_distances(q[:10])
    with pytest.raises(NbkError):
        arm.closest_distance(q[:10])


def test_misaligned_views_and_extreme_thresholds(fresh_world, torch_cuda):
    """Device views that start on an odd row (8-byte but not 16-byte aligned slabs), and thresholds far outside the
    geometry (everything / nothing collides): same masks as the oracle on every path."""
    arm, chain, obs = build_scene("c2")
    orc = Oracle(arm.scene_model())
    q = sample_q(chain, 20001, seed=95)
    qt = torch_cuda.from_numpy(q).cuda()
    ref = orc.validity(q, 0.0, nthreads=8)
    for lo, hi in ((1, 20001), (3, 9003), (1, 4001), (5, 70)):
        view = qt[lo:hi]
        assert view.data_ptr() % 16 != 0 or lo % 2 == 0
        got = arm.in_collision(view, 0.0)
        assert np.array_equal(got.cpu().numpy(), ref[lo:hi]), (lo, hi)
    T = arm.forward_kinematics(qt[1:3001], "tool_frame")
    assert_bitwise(T.cpu().numpy(), Oracle(arm._kin).fk(q[1:3001], "tool_frame"), "fk on a misaligned view")
    for thr in (1e6, -1e6, 5.0, -5.0):
        want = orc.validity(q[:9000], thr, nthreads=8)
        assert np.array_equal(arm.in_collision(q[:9000], thr), want), thr
        assert np.array_equal(arm.in_collision(q[:2000], thr), want[:2000]), thr
        assert want.all() if thr > 0 else not want.any()


def test_jacobian_group_tails_and_unaligned_output(kinova, torch_cuda):
    """k_jacobian_reg stages its rows 32 configurations at a time: batch sizes around the group and wave boundaries, every
    pose mode, and an output slab that is only 8-byte aligned (the 8-byte store path), through the C-ABI."""
    torch = torch_cuda
    arm, chain, _ = kinova
    orc = Oracle(arm._kin)
    dev = arm._kin_device()
    rng = np.random.default_rng(77)
    q = sample_q(chain, 200, seed=78)
    pose = np.tile(np.eye(4)[None], (200, 1, 1))
    pose[:, :3, 3] = rng.normal(size=(200, 3))
    for B in (1, 2, 31, 32, 33, 63, 64, 65, 97, 129, 200):
        for frame in ("tool_frame", "half_arm_2_link"):
            want = orc.jacobian(q[:B], frame)
            assert_bitwise(arm.jacobian(q[:B], frame), want, f"jac B={B} {frame}")
        assert_bitwise(arm.jacobian(q[:B], "tool_frame", global_pose=pose[:B]),
                       orc.jacobian(q[:B], "tool_frame", global_pose=pose[:B]), f"jac global B={B}")
        assert_bitwise(arm.jacobian(q[:B], "tool_frame", local_pose=pose[:B]),
                       orc.jacobian(q[:B], "tool_frame", local_pose=pose[:B]), f"jac local B={B}")
        # unaligned output through the C-ABI
        path, local = dev._frame_args("tool_frame", None)
        qt = torch.from_numpy(q[:B]).cuda()
        slab = torch.full((B * 42 + 1,), -7.0, dtype=torch.float64, device="cuda")
        out = slab[1:]
        assert out.data_ptr() % 16 == 8
        st = dev._lib.nbk_jacobian_batch(dev._h, qt.data_ptr(), B, path.ctypes.data, len(path), local.ctypes.data, 0, None,
                                         out.data_ptr(), dev._stream())
        assert st == 0
        torch.cuda.synchronize()
        assert_bitwise(out.cpu().numpy().reshape(B, 6, 7), orc.jacobian(q[:B], "tool_frame"), f"jac unaligned B={B}")
        assert float(slab[0]) == -7.0


def test_internal_workspace_state_across_thresholds_streams_and_sizes(fresh_world, torch_cuda):
    """nbk_validity_batch keeps its broadphase tables and alternates two counter sets between calls; changing the threshold,
    the batch size (tiles / reallocation), the stream, or interleaving edge batches must not leak state."""
    torch = torch_cuda
    arm, chain, obs = build_scene("c2")
    orc = Oracle(arm.scene_model())
    _, dev = arm._scene_device()
    q = sample_q(chain, 60000, seed=97)
    qt = torch.from_numpy(q).cuda()
    ref = {thr: orc.validity(q, thr, nthreads=8) for thr in (0.0, 1e-6, 0.02)}
    side = torch.cuda.Stream()
    seq = [(0.0, 20000), (0.0, 20000), (0.0, 60000), (1e-6, 9000), (1e-6, 9000), (0.0, 9000), (0.02, 33333), (0.0, 60000), (0.0, 8192)]
    for k, (thr, n) in enumerate(seq):
        if k % 3 == 2:
            with torch.cuda.stream(side):
                got = dev.validity(qt[:n], thr)
            side.synchronize()
        else:
            got = dev.validity(qt[:n], thr)
        assert np.array_equal(got.cpu().numpy(), ref[thr][:n]), (k, thr, n)
        if k == 4:                                   # an edge batch shares the workspace
            ok, _, _ = dev.edge_validity(q[:300], q[300:600], 0.05, 1.5)
            okr, _, _ = orc.edge_validity(q[:300], q[300:600], 0.05, 1.5)
            assert np.array_equal(ok, okr)


def test_descriptor_lifecycle_does_not_leak(fresh_world, torch_cuda):
    """Descriptors own device memory (tables, the validity workspace, edge scratch, frame sets): building, using and
    dropping 60 of them returns it."""
    import gc
    torch = torch_cuda
    from numbotics_amd.engine import DeviceModel
    arm, chain, obs = build_scene("c2")
    sm = arm.scene_model()
    q = torch.from_numpy(sample_q(chain, 20000, seed=99)).cuda()
    names = list(sm.kin.frames.keys())[:4]

    def cycle():
        dev = DeviceModel(sm)
        dev.validity(q, 0.0)
        dev.edge_validity(q[:64], q[64:128], 0.05, 1.0)
        dev.fk_frames(q[:256], names)
        del dev

    for _ in range(3):
        cycle()
    gc.collect(); torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(60):
        cycle()
    gc.collect(); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 * 1024 * 1024, (free0, free1)         # the workspace alone is 40 MB per descriptor


def test_capi_argument_errors_and_graph_capture(fresh_world, torch_cuda):
    """Status codes instead of exceptions across the C boundary; the workspace variant of the validity call is
    capturable into a HIP graph (no allocation, no synchronisation) and replays bit-identically."""
    import ctypes as C
    torch = torch_cuda
    from numbotics_amd import _lib
    arm, chain, obs = build_scene("c2")
    _, dev = arm._scene_device()
    lib = _lib.load()
    q = torch.from_numpy(sample_q(chain, 20000, seed=3)).cuda()
    words = torch.empty((313,), dtype=torch.int64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.nbk_validity_batch(dev._h, q.data_ptr(), 20000, 0.0, None, None, st) == -1          # no output given
    assert lib.nbk_validity_batch(None, q.data_ptr(), 20000, 0.0, words.data_ptr(), None, st) == -1
    assert lib.nbk_validity_batch(dev._h, None, 20000, 0.0, words.data_ptr(), None, st) == -1
    assert lib.nbk_validity_batch(dev._h, q.data_ptr(), -5, 0.0, words.data_ptr(), None, st) == -1
    assert lib.nbk_validity_batch(dev._h, q.data_ptr(), 0, 0.0, words.data_ptr(), None, st) == 0           # empty batch
    bad_path = (C.c_int32 * 2)(0, 99)
    local = (C.c_double * 12)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)
    T = torch.empty((20000, 16), dtype=torch.float64, device="cuda")
    assert lib.nbk_fk_batch(dev._h, q.data_ptr(), 20000, bad_path, 2, local, None, T.data_ptr(), st) == -1   # joint index out of range
    assert lib.nbk_edge_validity_batch(dev._h, q.data_ptr(), q.data_ptr(), None, 10, -0.1, 1.0, 0, 0.0,
                                       words.data_ptr(), None, None, st) == -1                               # bad resolution
    assert lib.nbk_status_string(-1).decode() == "invalid argument"
    # graph capture of the workspace variant
    need = dev.validity_workspace_bytes(20000)
    ws = torch.empty((need,), dtype=torch.uint8, device="cuda")
    ref = dev.validity(q, 0.0, packed=True).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sst = C.c_void_p(side.cuda_stream)
        _lib.check(lib.nbk_validity_batch_ws(dev._h, q.data_ptr(), 20000, 0.0, words.data_ptr(), None, ws.data_ptr(), need, sst), "warm-up")
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cst = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.nbk_validity_batch_ws(dev._h, q.data_ptr(), 20000, 0.0, words.data_ptr(), None, ws.data_ptr(), need, cst), "capture")
    words.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(words, ref)
    q.copy_(torch.from_numpy(sample_q(chain, 20000, seed=4)).cuda())        # new inputs, same graph
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(words, dev.validity(q, 0.0, packed=True))


def test_ten_million_batch_is_tiled_correctly(fresh_world, torch_cuda):
    """BASELINE config 4 size on one GPU: 1e7 q run as several queue-sized tiles; the oracle on a strided slice
    (which crosses every tile boundary) and the concatenation of two half batches give the same bits."""
    torch = torch_cuda
    from numbotics_amd.parallel import unpack_mask
    arm, chain, obs = build_scene("c2")
    _, dev = arm._scene_device()
    orc = Oracle(arm.scene_model())
    B = 10_000_000
    qh = sample_q(chain, B, seed=7)
    q = torch.from_numpy(qh).cuda()
    assert dev.validity_workspace_bytes(B) <= (1 << 30)
    w = dev.validity(q, 0.0, packed=True)
    bits = unpack_mask(w.cpu().numpy(), B)
    sl = np.arange(0, B, 211)
    assert np.array_equal(bits[sl], orc.validity(qh[sl], 0.0, nthreads=16))
    half = 5_000_000 - 5_000_000 % 64
    w2 = torch.cat([dev.validity(q[:half], 0.0, packed=True), dev.validity(q[half:], 0.0, packed=True)])
    assert torch.equal(w, w2)


def test_two_streams_share_one_arm(fresh_world, torch_cuda):
    """Every stream gets its own scratch set inside the descriptor: validity and edge batches issued from two torch
    streams without any synchronisation in between produce the bits of the serial run (no shared plan / queue / mask scratch)."""
    torch = torch_cuda
    arm, chain, obs = build_scene("c3")
    orc = Oracle(arm.scene_model())
    _, dev = arm._scene_device()
    q = sample_q(chain, 120000, seed=71)
    qa, qb = torch.from_numpy(q[:60000]).cuda(), torch.from_numpy(q[60000:]).cuda()
    ea = (torch.from_numpy(q[:3000]).cuda(), torch.from_numpy(q[3000:6000]).cuda())
    eb = (torch.from_numpy(q[6000:9000]).cuda(), torch.from_numpy(q[9000:12000]).cuda())
    ref_a, ref_b = orc.validity(q[:60000], nthreads=8), orc.validity(q[60000:], nthreads=8)
    ref_ea = orc.edge_validity(q[:3000], q[3000:6000], 0.03, 2.0, nthreads=8)[0]
    ref_eb = orc.edge_validity(q[6000:9000], q[9000:12000], 0.03, 2.0, nthreads=8)[0]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for rep in range(4):
        with torch.cuda.stream(s1):
            va = dev.validity(qa, 0.0)
            oka = dev.edge_validity(ea[0], ea[1], 0.03, 2.0)[0]
        with torch.cuda.stream(s2):
            okb = dev.edge_validity(eb[0], eb[1], 0.03, 2.0)[0]
            vb = dev.validity(qb, 0.0)
        outs.append((va, oka, okb, vb))
    torch.cuda.synchronize()
    for va, oka, okb, vb in outs:
        assert np.array_equal(va.cpu().numpy(), ref_a) and np.array_equal(vb.cpu().numpy(), ref_b)
        assert np.array_equal(oka.cpu().numpy(), ref_ea) and np.array_equal(okb.cpu().numpy(), ref_eb)


def test_internal_workspace_calls_are_capturable_once_allocated(fresh_world, torch_cuda):
    """nbk_validity_batch and nbk_edge_validity_batch inside a hipGraph: refused (status, no allocation inside the capture)
    before the stream's scratch exists; after one direct call they are self-contained graph nodes that replay bit-exactly on
    new inputs, and direct calls afterwards still work."""
    import ctypes as C
    torch = torch_cuda
    from numbotics_amd import _lib
    arm, chain, obs = build_scene("c3")
    orc = Oracle(arm.scene_model())
    _, dev = arm._scene_device()
    lib = _lib.load()
    B, E = 30000, 2000
    q = torch.from_numpy(sample_q(chain, B, seed=5)).cuda()
    s = torch.from_numpy(sample_q(chain, E, seed=6)).cuda()
    g = torch.from_numpy(sample_q(chain, E, seed=7)).cuda()
    words = torch.zeros(((B + 63) // 64,), dtype=torch.int64, device="cuda")
    valid = torch.zeros((E,), dtype=torch.uint8, device="cuda")
    end = torch.zeros((E, 7), dtype=torch.float64, device="cuda")
    ns = torch.zeros((E,), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())

    def calls(st):
        r1 = lib.nbk_validity_batch(dev._h, q.data_ptr(), B, 0.0, words.data_ptr(), None, st)
        r2 = lib.nbk_edge_validity_batch(dev._h, s.data_ptr(), g.data_ptr(), None, E, 0.02, 2.5, 1, 0.0, valid.data_ptr(),
                                         end.data_ptr(), ns.data_ptr(), st)
        return r1, r2
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        sst = C.c_void_p(side.cuda_stream)
        # 1. capture on a stream that has no scratch yet: refused with a status code, nothing allocated, capture stays valid
        g0.capture_begin()
        assert calls(sst) == (-4, -4)
        g0.capture_end()
        # 2. one direct call allocates this stream's scratch
        assert calls(sst) == (0, 0)
        side.synchronize()
        ref_words, ref_valid = words.clone(), valid.clone()
        assert np.array_equal(ref_valid.cpu().numpy().astype(bool), orc.edge_validity(s.cpu().numpy(), g.cpu().numpy(), 0.02, 2.5, mode="steer", nthreads=8)[0])
        # 3. capture for real
        g1 = torch.cuda.CUDAGraph()
        g1.capture_begin()
        assert calls(sst) == (0, 0)
        g1.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    words.zero_(); valid.zero_()
    g1.replay()
    torch.cuda.synchronize()
    assert torch.equal(words, ref_words) and torch.equal(valid, ref_valid)
    # new inputs, same graph -- twice, so that a replay also follows a replay
    for seed in (15, 25):
        q.copy_(torch.from_numpy(sample_q(chain, B, seed=seed)).cuda())
        s.copy_(torch.from_numpy(sample_q(chain, E, seed=seed + 1)).cuda())
        g.copy_(torch.from_numpy(sample_q(chain, E, seed=seed + 2)).cuda())
        g1.replay()
        torch.cuda.synchronize()
        assert np.array_equal(words.cpu().numpy(), dev.validity(q, 0.0, packed=True).cpu().numpy())
        okr, endr, nsr = orc.edge_validity(s.cpu().numpy(), g.cpu().numpy(), 0.02, 2.5, mode="steer", nthreads=8)
        assert np.array_equal(valid.cpu().numpy().astype(bool), okr) and np.array_equal(ns.cpu().numpy(), nsr)
        assert_bitwise(end.cpu().numpy(), endr, "captured steer end states")
    # direct calls on the capture stream still work after the capture (its cached state was dropped)
    with torch.cuda.stream(side):
        assert calls(C.c_void_p(side.cuda_stream)) == (0, 0)
    side.synchronize()
    assert np.array_equal(words.cpu().numpy(), dev.validity(q, 0.0, packed=True).cpu().numpy())


def test_graph_replays_interleaved_with_direct_calls(fresh_world, torch_cuda):
    """capture, direct, direct, replay, direct -- with another threshold and a SMALLER batch in the direct calls.  A replayed graph
    rewrites the stream's float32 tables for its own threshold and leaves counter set 0 non-empty behind the host's back; direct
    calls on a stream that has captured therefore never reuse cached tables / counter sets (StreamWs::captured).  Every mask
    against the oracle."""
    import ctypes as C
    torch = torch_cuda
    from numbotics_amd import _lib
    arm, chain, obs = build_scene("c3")
    orc = Oracle(arm.scene_model())
    _, dev = arm._scene_device()
    lib = _lib.load()
    B, Bs = 40000, 9000
    q = torch.from_numpy(sample_q(chain, B, seed=31)).cuda()
    qs = torch.from_numpy(sample_q(chain, Bs, seed=32)).cuda()
    words = torch.zeros(((B + 63) // 64,), dtype=torch.int64, device="cuda")
    words_s = torch.zeros(((Bs + 63) // 64,), dtype=torch.int64, device="cuda")
    ref = {thr: np.packbits(orc.validity(q.cpu().numpy(), thr, nthreads=8), bitorder="little") for thr in (0.0, 0.01)}
    ref_s = {thr: np.packbits(orc.validity(qs.cpu().numpy(), thr, nthreads=8), bitorder="little") for thr in (0.0, 0.01)}

    def bits(t, n):
        return t.cpu().numpy().view(np.uint8)[: (n + 7) // 8]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sst = C.c_void_p(side.cuda_stream)
        assert lib.nbk_validity_batch(dev._h, q.data_ptr(), B, 0.01, words.data_ptr(), None, sst) == 0       # allocates the scratch
        side.synchronize()
        assert np.array_equal(bits(words, B), ref[0.01])
        g1 = torch.cuda.CUDAGraph()
        g1.capture_begin()
        assert lib.nbk_validity_batch(dev._h, q.data_ptr(), B, 0.01, words.data_ptr(), None, sst) == 0
        g1.capture_end()
        for round_ in range(2):
            # two direct calls at another threshold (the second would have taken the cached-table path and counter set 1)
            for _ in range(2):
                words_s.zero_()
                assert lib.nbk_validity_batch(dev._h, qs.data_ptr(), Bs, 0.0, words_s.data_ptr(), None, sst) == 0
                side.synchronize()
                assert np.array_equal(bits(words_s, Bs), ref_s[0.0])
            words.zero_()
            g1.replay()
            side.synchronize()
            assert np.array_equal(bits(words, B), ref[0.01])
            # the direct call after the replay: smaller batch, other threshold (stale items of the replay would index beyond Bs)
            words_s.zero_()
            assert lib.nbk_validity_batch(dev._h, qs.data_ptr(), Bs, 0.0, words_s.data_ptr(), None, sst) == 0
            side.synchronize()
            assert np.array_equal(bits(words_s, Bs), ref_s[0.0])
            # and one at the graph's own threshold
            words_s.zero_()
            assert lib.nbk_validity_batch(dev._h, qs.data_ptr(), Bs, 0.01, words_s.data_ptr(), None, sst) == 0
            side.synchronize()
            assert np.array_equal(bits(words_s, Bs), ref_s[0.01])
    torch.cuda.current_stream().wait_stream(side)


def test_edge_batches_beyond_the_scratch_capacity(fresh_world, torch_cuda):
    """connect() edges far longer than max_distance: the flat batch's capacity (E x (max_distance / resolution + 2) samples)
    overflows, the edges that do not fit are walked one wave each -- same bits; the device reports the true count through
    pinned memory and the next call sizes its scratch from it."""
    arm, chain, obs = build_scene("c2")
    orc = Oracle(arm.scene_model())
    _, dev = arm._scene_device()
    rng = np.random.default_rng(3)
    base = sample_q(chain, 1, seed=9)[0] * 0.2
    s_ = base + rng.uniform(-0.3, 0.3, (400, 7))
    g_ = s_ + rng.uniform(-1.0, 1.0, (400, 7)) * rng.uniform(0.1, 3.0, (400, 1))         # lengths up to ~5 rad
    okr, endr, nsr = orc.edge_validity(s_, g_, 0.01, 0.25, mode="connect", nthreads=8)
    assert nsr.sum() > 3 * 400 * 27 and 0 < okr.sum() < 400                      # well beyond E * (0.25 / 0.01 + 2) samples
    for rep in range(3):                                                          # first call overflows, later ones have grown
        ok, end, ns = dev.edge_validity(s_, g_, 0.01, 0.25, mode="connect")
        assert np.array_equal(ok, okr) and np.array_equal(ns, nsr), rep
        assert_bitwise(end, endr, "overflow edge ends")
    ok, end, ns = dev.edge_validity(s_, g_, 0.01, 0.25, mode="steer")
    okr, endr, nsr = orc.edge_validity(s_, g_, 0.01, 0.25, mode="steer", nthreads=8)
    assert np.array_equal(ok, okr) and np.array_equal(ns, nsr) and nsr.max() <= 27
    assert_bitwise(end, endr, "steer ends")


def test_scalar_host_paths(fresh_world, torch_cuda):
    """Arm.in_collision(q) / DiscreteConnector.connect / steer on single host arrays run through the library's pinned
    staging (nbk_validity_scalar_host / nbk_edge_validity_scalar_host): same answers as the oracle and as the batch path."""
    from numbotics_amd.planning.sampling_based import ConnectorParams, DiscreteConnector
    arm, chain, obs = build_scene("c3")
    orc = Oracle(arm.scene_model())
    q = sample_q(chain, 300, seed=77)
    ref = orc.validity(q)
    got = np.array([arm.in_collision(q[i]) for i in range(300)])
    assert np.array_equal(got, ref) and 0 < ref.sum() < 300
    assert np.array_equal(np.array([arm.in_collision(q[i], 0.05) for i in range(100)]), orc.validity(q[:100], 0.05))
    assert arm.in_collision(np.full(7, np.nan)) is True
    conn = DiscreteConnector(ConnectorParams(resolution=0.02, max_distance=1.0, arm=arm))
    okr, endr, nsr = orc.edge_validity(q[:100], q[100:200], 0.02, 1.0, mode="connect")
    oks, ends, _ = orc.edge_validity(q[:100], q[100:200], 0.02, 1.0, mode="steer")
    for i in range(100):
        c = conn.connect(q[i], q[100 + i])
        assert (c is not None) == bool(okr[i]) and (c is None or np.array_equal(c, q[100 + i]))
        st = conn.steer(q[i], q[100 + i])
        assert (st is not None) == bool(oks[i]) and (st is None or np.array_equal(st, ends[i]))
    assert conn.connect(q[0], q[0]) is None                                   # d <= float32 eps
    _, dev = arm._scene_device()
    ok, end, ns = dev.edge_validity_scalar(q[3], q[150], 0.02, 1.0, mode="steer")
    o1, e1, n1 = orc.edge_validity(q[3:4], q[150:151], 0.02, 1.0, mode="steer")
    assert (ok, ns) == (bool(o1[0]), int(n1[0])) and np.array_equal(end, e1[0])


def test_queue_overflow_is_redecided_without_a_queue(fresh_world, torch_cuda):
    """The item queues are sized for a budget (1 GiB), not for the worst case; a block whose items do not fit marks itself and
    k_validity_redo decides it the queue-less way.  With the budget shrunk to a few KB nearly every block overflows: same
    masks, for plain batches (both workspaces, bit and byte masks) and for edge batches."""
    torch = torch_cuda
    arm, chain, obs = build_scene("c3")
    orc = Oracle(arm.scene_model())
    _, dev = arm._scene_device()
    q = sample_q(chain, 50000, seed=81)
    refs = {thr: orc.validity(q, thr, nthreads=8) for thr in (0.0, 0.02)}
    e_ref = orc.edge_validity(q[:1500], q[1500:3000], 0.03, 2.0, nthreads=8)
    for budget in (1 << 12, 1 << 17, 1 << 21):
        with debug_option("queue_budget", budget):
            for thr, ref in refs.items():
                assert np.array_equal(dev.validity(q, thr), ref), (budget, thr)
                words = dev.validity(q, thr, packed=True)
                from numbotics_amd.parallel import unpack_mask
                assert np.array_equal(unpack_mask(words, 50000), ref), (budget, thr, "packed")
            need = dev.validity_workspace_bytes(50000)
            ws = torch.empty((need,), dtype=torch.uint8, device="cuda")
            assert np.array_equal(dev.validity(q, 0.0, workspace=ws), refs[0.0]), (budget, "caller workspace")
            ok, end, ns = dev.edge_validity(q[:1500], q[1500:3000], 0.03, 2.0)
            assert np.array_equal(ok, e_ref[0]) and np.array_equal(ns, e_ref[2]), (budget, "edges")
    assert np.array_equal(dev.validity(q, 0.0), refs[0.0])                 # and back at the default budget


@pytest.mark.parametrize("margins", [True, False], ids=["bullet", "sharp"])
@pytest.mark.parametrize("scene", ["c2", "c3", "c5m"])
def test_bullet_margin_mode(fresh_world, scene, margins, torch_cuda):
    """``Arm(chain)`` = ``bullet_margins=True`` (the default): every box / cylinder / mesh hull without an explicit margin gets the
    one Bullet applies (min(0.04, a tenth of the smallest half extent); hulls 0.001) -- the scene closest to what the reference's
    getClosestPoints measures; ``bullet_margins=False``: the sharp analytic shapes.  Same bar in both: masks at four thresholds
    (all three validity paths), distances, witnesses and gradient rows bit for bit against the oracle.  (Parity with Bullet itself
    stays unpinned.)"""
    arm, chain, obs = build_scene(scene, bullet_margins=margins)
    sm = arm.scene_model()
    if margins:
        assert (sm.rshape_param[:, 3] > 0).sum() >= 9 and (sm.wshape_param[:, 3] > 0).all()
    else:
        assert (sm.rshape_param[:, 3] == 0).all() and (sm.wshape_param[:, 3] == 0).all()
    orc = Oracle(sm)
    q = sample_q(chain, 20000, seed=2)
    for thr in (0.0, 1e-6, 0.02, -0.005):
        ref = orc.validity(q, thr, nthreads=8)
        assert np.array_equal(arm.in_collision(q, thr), ref), (scene, thr)
        with fused_path():
            assert np.array_equal(arm.in_collision(q[:3000], thr), ref[:3000]), (scene, thr, "fused")
        with debug_option("f64_broad", 1):
            assert np.array_equal(arm.in_collision(q, thr), ref), (scene, thr, "float64 broadphase")
    d, w, rows = arm.proximity_jacobians(q[:1500])
    dr, wr, rr = orc.proximity_jacobian(q[:1500])
    assert_bitwise(d, dr, "bullet-margin distances")
    assert_bitwise(w, wr, "bullet-margin witnesses")
    assert_bitwise(rows, rr, "bullet-margin rows")
    dmin, idx = arm.closest_distance(q[:3000])
    dref, iref = orc.closest(q[:3000])
    assert_bitwise(dmin, dref, "bullet-margin closest")
    assert np.array_equal(idx, iref)
    # switching the mode on an existing arm recompiles the scene
    arm.bullet_margins = not margins
    assert ((arm.scene_model().rshape_param[:, 3] == 0).all()) == margins
    assert np.array_equal(arm.in_collision(q[:5000]), Oracle(arm.scene_model()).validity(q[:5000]))


def test_two_arms_in_one_world(fresh_world, torch_cuda):
    """The links of another chain are obstacles of this arm's scene (robots/arm.py:226-243): masks, distances and closest pair bit
    for bit against the oracle, and the scene follows the other arm's configuration."""
    from numbotics_amd.physics import GraphChain
    from numbotics_amd.robots import Arm
    from numbotics_amd.scenes import KINOVA_URDF, KINOVA_MESH_URDF, apply_rrt_script_removals
    c1 = GraphChain.from_urdf(KINOVA_URDF)
    c2 = GraphChain.from_urdf(KINOVA_MESH_URDF)                              # the neighbour carries mesh links
    T = np.eye(4); T[:3, 3] = [0.55, 0.25, 0.0]
    c2.base_pose = T
    c2.configuration = np.array([0.3, 0.8, -0.5, 1.2, 0.1, -0.7, 0.4])
    arm = Arm(c1)
    apply_rrt_script_removals(arm)
    q = sample_q(c1, 20000, seed=12)
    masks = []
    for conf in (c2.configuration, np.array([-1.0, 1.3, 0.4, 1.8, -0.3, 0.9, 0.0])):
        c2.configuration = conf
        sm = arm.scene_model()
        assert sm.n_wshapes == 11 and (sm.wshape_type == 5).sum() == 10
        orc = Oracle(sm)
        for thr in (0.0, 0.02):
            assert np.array_equal(arm.in_collision(q, thr), orc.validity(q, thr, nthreads=8)), thr
        assert_bitwise(arm.pair_distances(q[:1000]), orc.pair_distances(q[:1000]), "two-arm distances")
        dmin, idx = arm.closest_distance(q[:2000])
        dref, iref = orc.closest(q[:2000])
        assert_bitwise(dmin, dref, "two-arm closest")
        assert np.array_equal(idx, iref)
        masks.append(orc.validity(q, 0.0, nthreads=8))
    assert (masks[0] != masks[1]).any() and 0.02 < masks[0].mean() < 0.9


def test_body_level_distance_to(fresh_world, torch_cuda):
    """``PhysicsObject.distance_to`` / ``Link.distance_to`` / ``Chain.distance_to`` (reference: object.py:325-349, chain.py:352-379,
    :944-969) -- closest-point records at the bodies' current state -- agree with the oracle's shape-pair distances and with the
    arm-level records."""
    from numbotics_amd.physics import GraphChain, Cube, Sphere, Plane, Mesh
    from numbotics_amd.robots import Arm
    from numbotics_amd.scenes import KINOVA_URDF, MESH_DIR
    from oracle.cpu_oracle import shape_distance
    import os
    cube = Cube(0.0, 0.2, position=np.array([0.6, 0.1, 0.3]))
    ball = Sphere(0.0, 0.1, position=np.array([0.6, 0.1, 0.9]))
    floor = Plane(0.0, np.array([0.0, 0.0, 1.0]), position=np.array([0.0, 0.0, -0.05]))
    rock = Mesh(0.0, os.path.join(MESH_DIR, "rock.obj"), position=np.array([-0.5, 0.2, 0.4]))
    # object vs object
    (p,) = cube.distance_to(ball)
    d, wa, wb, n, _ = shape_distance(2, cube.pose, [0.2, 0.2, 0.2, 0.0], 0, ball.pose, [0.1, 0, 0, 0])
    assert p.subject is cube and p.target is ball and p.distance == d
    assert np.array_equal(p.position_on_subject, wa) and np.array_equal(p.position_on_target, wb) and np.array_equal(p.normal_target_to_subject, n)
    assert abs(p.distance - 0.3) < 1e-12
    assert cube.distance_to(ball, max_distance=0.1) == []
    # a plane as the subject: measured the other way round, fields swapped
    (pp,) = floor.distance_to(ball)
    (pq,) = ball.distance_to(floor)
    assert pp.distance == pq.distance and abs(pp.distance - 0.85) < 1e-12
    assert np.array_equal(pp.position_on_subject, pq.position_on_target) and np.array_equal(pp.normal_target_to_subject, -pq.normal_target_to_subject)
    assert len(rock.distance_to(cube)) == 1 and rock.distance_to(cube)[0].distance > 0
    # chain / link vs object at the chain's configuration == the arm-level records at that q
    chain = GraphChain.from_urdf(KINOVA_URDF)
    q = np.array([0.4, 0.9, -0.3, 1.1, 0.2, -0.6, 0.1])
    chain.configuration = q
    arm = Arm(chain)
    prox = chain.distance_to(cube)
    ref = arm.distance_to(q, cube)
    assert len(prox) == len(ref) == 11
    assert sorted(round(p.distance, 12) for p in prox) == sorted(round(p.distance, 12) for p in ref)
    link = next(l for l in chain._links if l._name == "forearm_link")
    (pl,) = link.distance_to(cube)
    assert pl.subject is link and any(abs(pl.distance - r.distance) < 1e-12 and r.subject._name == "forearm_link" for r in ref)
    assert len(chain.distance_to(rock, max_distance=0.5)) <= 11
