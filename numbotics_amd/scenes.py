"""The benchmark / test scenes of BASELINE.json's configs, built through the public API.

c1 : Kinova-like arm alone (FK plumbing config).
c2 : arm + one Cube(half_extent=0.4) at [1.0, 0.0, 0.2] (reference README.md:96); self pairs = default
     rule minus the hand removals of numbotics/tests/_test_rrt.py:38-61.
c3 : arm + 8 Cube(half_extent=0.25) on a ring of radius 0.9 m, heights alternating 0.25 / 0.75 m, angles
     k*45 deg (build-defined: the reference has no 8-cube scene).
c2m: c2 with the arm's collision primitives replaced by meshes (models/kinova_mesh.urdf: one convex hull per mesh file,
     the bracelet link a compound of two <collision> elements).
c5m: the mesh arm among mesh obstacles -- a rock (one hull), a table (compound: the five objects of one OBJ loaded with
     convex_decomposition=True, i.e. the file taken as its own decomposition; without the flag it would be ONE hull, as in
     the reference, whose trimesh round trip merges the objects of a file), a wedge (binary
     STL, scaled and rotated through the shape kwargs) and one Cube: BASELINE config 5's "compound-mesh collision shapes"
     (build-defined layout: the reference ships no mesh scene, numbotics/tests/_test_manual.py:41 loads a single mesh body).
The Kinova URDF is this build's own asset (numbotics_amd/models/kinova_cyl.urdf).
"""
import os
from itertools import combinations

import numpy as np

KINOVA_URDF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "kinova_cyl.urdf")
KINOVA_MESH_URDF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "kinova_mesh.urdf")
MESH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "meshes")

_WRIST_GROUP = [
    'spherical_wrist_1_link', 'spherical_wrist_2_link', 'bracelet_link', 'end_effector_link', 'camera_link',
    'camera_depth_frame', 'camera_color_frame', 'tool_frame', 'robotiq_arg2f_base_link', 'gripper', 'camera',
]


def apply_rrt_script_removals(arm):
    """The self-pair removals the reference's PRM demo performs by hand (_test_rrt.py:38-61)."""
    from numbotics_amd.utils import logger
    verbose, logger.VERBOSE = logger.VERBOSE, False
    try:
        arm.remove_collision_pair('base_link', 'half_arm_1_link')
        arm.remove_collision_pair('half_arm_2_link', 'spherical_wrist_1_link')
        for a, b in combinations(_WRIST_GROUP, 2):
            arm.remove_collision_pair(a, b)
    finally:
        logger.VERBOSE = verbose


def build_scene(name: str = "c2", urdf: str = None, bullet_margins: bool = True):
    """-> (arm, chain, obstacles).  Keep the returned obstacles alive: the world holds weak references."""
    from numbotics_amd.physics import GraphChain, Cube, Mesh
    from numbotics_amd.robots import Arm
    if urdf is None:
        urdf = KINOVA_MESH_URDF if name in ("c2m", "c5m") else KINOVA_URDF
    chain = GraphChain.from_urdf(urdf)
    arm = Arm(chain, bullet_margins=bullet_margins)
    obstacles = []
    if name == "c1":
        pass
    elif name in ("c2", "c2m"):
        obstacles.append(Cube(half_extent=0.4, mass=0.0, position=np.array([1.0, 0.0, 0.2])))
        apply_rrt_script_removals(arm)
    elif name == "c5m":
        from numbotics_amd.math import rpy_matrix, trans_mat
        obstacles.append(Mesh(0.0, os.path.join(MESH_DIR, "rock.obj"), position=np.array([0.55, 0.25, 0.45])))
        obstacles.append(Mesh(0.0, os.path.join(MESH_DIR, "table.obj"), position=np.array([0.0, -0.75, 0.0]), convex_decomposition=True))
        obstacles.append(Mesh(0.0, os.path.join(MESH_DIR, "wedge.stl"), mesh_scale=np.array([1.5, 1.5, 1.2]),
                              offset=trans_mat(pos=np.array([-0.1, -0.1, 0.0]), orn=rpy_matrix(np.array([0.0, 0.0, 0.6]))),
                              position=np.array([-0.6, 0.35, 0.0])))
        obstacles.append(Cube(half_extent=0.15, mass=0.0, position=np.array([0.45, -0.35, 0.85])))
        apply_rrt_script_removals(arm)
    elif name == "c3":
        for k in range(8):
            ang = np.deg2rad(45.0 * k)
            z = 0.25 if k % 2 == 0 else 0.75
            obstacles.append(Cube(half_extent=0.25, mass=0.0,
                                  position=np.array([0.9 * np.cos(ang), 0.9 * np.sin(ang), z])))
        apply_rrt_script_removals(arm)
    else:
        raise ValueError(f"unknown scene '{name}'")
    return arm, chain, obstacles


def sample_q(chain, B: int, seed: int = 1, margin: float = 0.0):
    """q ~ U(lower + margin, upper - margin), float64 (B, dof), numpy default_rng(seed)."""
    rng = np.random.default_rng(seed)
    lim = chain.joint_limits
    return rng.uniform(lim[:, 0] + margin, lim[:, 1] - margin, (B, chain.dof))


def sample_q_device(chain, B: int, seed: int = 1, margin: float = 0.0, device="cuda", out=None):
    """The same distribution drawn ON the device (torch's generator: not the NumPy stream of ``sample_q``): a float64 (B, dof)
    CUDA tensor, so that a sampling planner never moves q over PCIe -- 56 MB per 1e6 configurations cost 8 x the validity step."""
    import torch
    lim = torch.as_tensor(np.asarray(chain.joint_limits, dtype=np.float64), device=device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    u = torch.rand((B, chain.dof), dtype=torch.float64, device=device, generator=g) if out is None else out.uniform_(0.0, 1.0, generator=g)
    lo = lim[:, 0] + margin
    return u.mul_(lim[:, 1] - margin - lo).add_(lo)
