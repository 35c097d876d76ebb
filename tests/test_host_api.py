"""Host-side mirror of the reference interface: URDF reader, pair bookkeeping, connector contract,
error behaviour, and the C-ABI library's symbols.  No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, URDF


def test_capi_exports_every_declared_symbol():
    from numbotics_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "nbk.h")).read()
    declared = set(re.findall(r"\b(nbk_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.nbk_abi_version() == 2
    assert lib.nbk_status_string(-2).decode() == "no HIP device available"
    assert lib.nbk_device_count() >= 0


def test_header_is_plain_c(tmp_path):
    """include/nbk.h must be consumable by a C compiler (cgo / JNI / ctypes generators read it as C)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "nbk.h"\nint main(void) { nbk_model_desc d; (void)d; return NBK_ABI_VERSION == 1 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                   check=True)


def test_device_path_fails_loudly_without_gpu(kinova):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from numbotics_amd._lib import NbkError
    arm, chain, _ = kinova
    q = np.zeros((4, 7))
    with pytest.raises(NbkError):
        arm.forward_kinematics(q, "tool_frame")
    with pytest.raises(NbkError):
        arm.in_collision(q)
    # and the C layer itself refuses to create a descriptor
    from numbotics_amd import _lib
    d = _lib.ModelDesc()
    h = ctypes.c_void_p()
    assert _lib.load().nbk_model_create(ctypes.byref(d), ctypes.byref(h)) == -2


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "numbotics_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)
                assert "libnbk_oracle" not in src and "nbk_oracle.h" not in src, os.path.join(dp, f)


def test_urdf_reader(fresh_world):
    from numbotics_amd.physics import GraphChain, Constraint
    from numbotics_amd.utils import Shape
    chain = GraphChain.from_urdf(URDF)
    assert chain.dof == 7 and len(chain._links) == 16
    assert chain.joint_limits.shape == (7, 2) and np.all(chain.joint_limits[:, 0] < chain.joint_limits[:, 1])
    by = {l._name: l for l in chain._links}
    assert by["base_link"]._collision_shape.shape == Shape.CYLINDER
    assert by["robotiq_arg2f_base_link"]._collision_shape.shape == Shape.CUBOID
    assert np.allclose(by["robotiq_arg2f_base_link"]._collision_shape._shape_info["half_extents"], [0.0375, 0.06, 0.045])
    assert by["gripper"]._collision_shape.shape == Shape.SPHERE
    assert by["tool_frame"]._collision_shape.shape == Shape.EMPTY
    assert len(by["bracelet_link"]._collision_shapes) == 2          # compound; upstream reads only the first
    j1 = chain._G.edges[("base_link", "shoulder_link")]["joint"]
    assert j1.type == Constraint.REVOLUTE                              # 'continuous' -> REVOLUTE
    assert np.allclose(j1.offset[:3, 3], [0, 0, 0.15643]) and np.allclose(j1.offset[:3, :3], np.diag([1, -1, -1]))
    assert by["tool_frame"].name == f"{chain.world.name}:{chain._name}:tool_frame"
    # flange height of the Gen3-like kinematics at q = 0
    from oracle.cpu_oracle import Oracle
    from numbotics_amd.robots.model import compile_kinematics
    T = Oracle(compile_kinematics(chain)).fk(np.zeros((1, 7)), "end_effector_link")[0]
    assert T[2, 3] == pytest.approx(1.1873, abs=1e-4) and abs(T[0, 3]) < 1e-12


def test_argument_errors_match_reference(kinova):
    arm, chain, _ = kinova
    q = np.zeros((3, 7))
    with pytest.raises(ValueError, match="not found in chain"):
        arm.forward_kinematics(q, "no_such_frame")
    with pytest.raises(ValueError, match="must have 7 elements"):
        arm.forward_kinematics(np.zeros((3, 6)), "tool_frame")
    with pytest.raises(ValueError, match="4x4"):
        arm.forward_kinematics(q, "tool_frame", local_pose=np.eye(3))
    with pytest.raises(ValueError, match="same batch dimensions"):
        arm.forward_kinematics(q, "tool_frame", local_pose=np.tile(np.eye(4), (2, 1, 1)))
    with pytest.raises(ValueError, match="cannot both"):
        arm.jacobian(q, "tool_frame", local_pose=np.eye(4), global_pose=np.eye(4))
    with pytest.raises(ValueError, match="1D array with 7"):
        arm.collisions(q)
    with pytest.raises(ValueError):
        arm.add_collision_pair("tool_frame", "nope")


def test_pair_bookkeeping(kinova):
    """arm.py:190-366: default rule, string lookup, add/remove, obstacle pairs."""
    arm, chain, obstacles = kinova
    cube = obstacles[0]
    names = lambda pairs: {tuple(sorted((a._name, b._name))) for a, b in pairs}     # noqa: E731
    selfp = names(arm.self_collision_pairs())
    assert ("base_link", "half_arm_1_link") not in selfp                  # removed by the script replay
    assert ("base_link", "forearm_link") in selfp
    assert ("bracelet_link", "gripper") not in selfp
    assert len(selfp) == 28
    allp = arm.collision_pairs()
    assert len(allp) == 28 + 10                                           # + 10 shaped links x 1 cube
    v0 = arm._pairs_version
    arm.remove_collision_pair("forearm_link", cube.name)
    assert len(arm.collision_pairs()) == 37 and arm._pairs_version > v0
    arm.add_collision_pair(cube, "forearm_link")                          # objects, either order
    assert len(arm.collision_pairs()) == 38
    arm.add_collision_pair("base_link", "half_arm_1_link")                # un-void a default pair
    assert ("base_link", "half_arm_1_link") in names(arm.self_collision_pairs())
    arm.add_collision_pair("base_link", "shoulder_link")                  # adjacent links: additional pair
    assert ("base_link", "shoulder_link") in names(arm.self_collision_pairs())
    arm.remove_collision_pair("shoulder_link", "base_link")
    assert ("base_link", "shoulder_link") not in names(arm.self_collision_pairs())
    sm = arm.scene_model()
    assert sm.n_wshapes == 1 and sm.n_pairs == len([1 for a, b in zip(sm.pair_a, sm.pair_b)])
    assert np.all(np.diff(sm.pair_a) >= 0)                                # sorted by subject shape
    # moving the obstacle invalidates the cached scene
    cube.position = np.array([0.5, 0.5, 0.5])
    assert np.allclose(arm.scene_model().wshape_pose[0].reshape(3, 4)[:, 3], [0.5, 0.5, 0.5])


def test_world_registry_is_weak_like_upstream(fresh_world):
    from numbotics_amd.physics import Cube
    c = Cube(half_extent=0.1, mass=0.0)
    assert len(fresh_world.objects()) == 1 and c._static is False and c.name.endswith(c._name)
    del c
    import gc; gc.collect()
    assert len(fresh_world.objects()) == 0


def test_connector_params_validation():
    from numbotics_amd.planning.sampling_based import ConnectorParams
    ok = lambda q: True        # noqa: E731
    for bad in (dict(resolution=0.0), dict(resolution=-1.0), dict(resolution=1.0), dict(max_distance=0.0)):
        with pytest.raises(ValueError):
            ConnectorParams(validity_checker=ok, **bad)
    with pytest.raises(ValueError, match="Validity checker"):
        ConnectorParams()
    with pytest.raises(ValueError, match="Trajectory"):
        ConnectorParams(validity_checker=ok, trajectory_func=None)
    assert ConnectorParams(validity_checker=ok).resolution == 5e-2


def test_discrete_connector_scalar_contract_vs_golden(g5, golden_meta):
    """connect/steer with a Python validity_checker: the exact sample points handed to the checker and the
    return value, as recorded from the reference (connectors.py:57-100)."""
    from numbotics_amd.planning.sampling_based import ConnectorParams, DiscreteConnector
    checked = 0
    for c in golden_meta["g5_cases"]:
        if c["fail_at"] == -1:
            continue
        k = c["id"]
        s, g, ref, ret = g5[f"g5_{k}_start"], g5[f"g5_{k}_goal"], g5[f"g5_{k}_samples"], g5[f"g5_{k}_ret"]
        seen = []

        def checker(qv, _f=c["fail_at"]):
            seen.append(np.array(qv, copy=True))
            return not (_f is not None and len(seen) - 1 == _f)
        conn = DiscreteConnector(ConnectorParams(resolution=c["resolution"], max_distance=c["max_distance"],
                                                 validity_checker=checker))
        out = getattr(conn, c["mode"])(s, g)
        assert (out is None) == c["returned_none"], k
        assert len(seen) == c["n_checked"], k
        if seen:
            assert np.array_equal(np.array(seen), ref), k
        if out is not None:
            assert np.array_equal(out, ret) and out is not g
        checked += 1
    assert checked > 300


def test_links_of_another_chain_are_obstacles(fresh_world):
    """Upstream pairs every link of the arm with every link of the OTHER chains in its world (robots/arm.py:226-243; Bullet holds
    those bodies at their current joint state).  Here they enter the scene as static shapes at that chain's ``configuration``."""
    from numbotics_amd.physics import GraphChain
    from numbotics_amd.robots import Arm
    from numbotics_amd.robots.model import chain_link_poses
    from numbotics_amd.scenes import KINOVA_URDF, apply_rrt_script_removals
    from oracle.cpu_oracle import Oracle
    c1 = GraphChain.from_urdf(KINOVA_URDF)
    c2 = GraphChain.from_urdf(KINOVA_URDF)
    T = np.eye(4); T[:3, 3] = [0.5, 0.3, 0.0]
    c2.base_pose = T
    c2.configuration = np.array([0.3, 0.8, -0.5, 1.2, 0.1, -0.7, 0.4])
    arm = Arm(c1)
    apply_rrt_script_removals(arm)
    n_self = len(arm.self_collision_pairs())
    sm = arm.scene_model()
    assert sm.n_rshapes == 11 and sm.n_wshapes == 11                      # the other arm's 11 primitives
    assert sm.n_pairs == 121 + (sm.n_pairs - 121) and sm.n_pairs > 121    # 11 x 11 cross pairs + the arm's own self pairs
    assert len(arm.collision_pairs()) == n_self + 10 * 16                   # 10 shaped links x all 16 links of the other chain
    # the other chain's link poses = its own FK at its configuration
    poses = chain_link_poses(c2)
    orc2 = Oracle(Arm(c2)._kin)
    for name in ("tool_frame", "forearm_link", "bracelet_link", "base_link"):
        assert np.abs(poses[name] - orc2.fk(c2.configuration[None], name)[0]).max() < 1e-14
    q = np.random.default_rng(3).uniform(-2, 2, (2000, 7))
    m0 = Oracle(sm).validity(q)
    rev = c1.world._revision
    c2.configuration = np.zeros(7)                                         # moving the other arm invalidates the compiled scene
    assert c1.world._revision > rev
    m1 = Oracle(arm.scene_model()).validity(q)
    assert 0 < m0.sum() < 2000 and (m0 != m1).any()
    # pairs against the other chain can be removed by name like any other
    arm.remove_collision_pair('tool_frame', c2._links[3])
