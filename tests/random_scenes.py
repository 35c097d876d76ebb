"""Random mechanisms + obstacle sets for the fuzz tests (build-authored; nothing here comes from the reference)."""
import numpy as np


def random_hull_obj(rng, path, n_objects=1, scale=0.1):
    """A mesh file of ``n_objects`` random convex polytopes (one OBJ object each)."""
    from numbotics_amd.utils.mesh import write_obj, hull_faces
    parts = []
    for i in range(n_objects):
        n = int(rng.integers(6, 28))
        pts = rng.normal(size=(n, 3))
        pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.5, 1.0, (n, 1)) * rng.uniform(0.3, 1.0, 3) * scale
        V, F = hull_faces(pts + rng.uniform(-scale, scale, 3) * (i > 0))
        parts.append((f"part{i}", V, F))
    return write_obj(path, parts)


def random_urdf(rng, n_links, path, max_back=3, meshes=False):
    """A random tree of n_links links: revolute / continuous / prismatic / fixed joints with random origins and
    axes, every link carrying 0-2 collision primitives (box / sphere / cylinder / capsule; with ``meshes`` also <mesh>
    elements -- random convex polytopes, some files holding two objects -- written next to the URDF)."""
    import os
    mesh_no = [0]

    def geom():
        kind = rng.integers(0, 6 if meshes else 4)
        if kind >= 4:
            fn = f"fuzz_mesh_{mesh_no[0]}.obj"
            mesh_no[0] += 1
            random_hull_obj(rng, os.path.join(os.path.dirname(path), fn), n_objects=int(rng.choice([1, 1, 2])), scale=0.08)
            sc = "" if rng.random() < 0.5 else f' scale="{rng.uniform(0.6, 1.4):.3f} {rng.uniform(0.6, 1.4):.3f} {rng.uniform(0.6, 1.4):.3f}"'
            return f'<mesh filename="{fn}"{sc}/>'
        if kind == 0:
            s = rng.uniform(0.04, 0.16, 3)
            return f'<box size="{s[0]:.4f} {s[1]:.4f} {s[2]:.4f}"/>'
        if kind == 1:
            return f'<sphere radius="{rng.uniform(0.02, 0.07):.4f}"/>'
        if kind == 2:
            return f'<cylinder radius="{rng.uniform(0.015, 0.05):.4f}" length="{rng.uniform(0.05, 0.25):.4f}"/>'
        return f'<capsule radius="{rng.uniform(0.015, 0.04):.4f}" length="{rng.uniform(0.05, 0.2):.4f}"/>'

    def origin(scale):
        xyz = rng.uniform(-scale, scale, 3)
        rpy = rng.uniform(-1.0, 1.0, 3) * (rng.random() < 0.6)
        return f'<origin xyz="{xyz[0]:.4f} {xyz[1]:.4f} {xyz[2]:.4f}" rpy="{rpy[0]:.4f} {rpy[1]:.4f} {rpy[2]:.4f}"/>'

    out = ['<?xml version="1.0"?>', '<robot name="fuzz">']
    for i in range(n_links):
        cols = "".join(f"<collision>{origin(0.05)}<geometry>{geom()}</geometry></collision>"
                       for _ in range(int(rng.choice([0, 1, 1, 1, 2]))))
        out.append(f'<link name="l{i}">{cols}</link>')
    for i in range(1, n_links):
        parent = int(rng.integers(max(0, i - max_back), i))
        jt = str(rng.choice(["revolute", "revolute", "continuous", "prismatic", "fixed"]))
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        lim = ""
        if jt == "revolute":
            lim = f'<limit lower="{-rng.uniform(0.5, 3.0):.3f}" upper="{rng.uniform(0.5, 3.0):.3f}" effort="1" velocity="1"/>'
        elif jt == "prismatic":
            lim = f'<limit lower="{-rng.uniform(0.0, 0.1):.3f}" upper="{rng.uniform(0.05, 0.3):.3f}" effort="1" velocity="1"/>'
        axis = "" if jt == "fixed" else f'<axis xyz="{ax[0]:.5f} {ax[1]:.5f} {ax[2]:.5f}"/>'
        out.append(f'<joint name="j{i}" type="{jt}">{origin(0.4)}<parent link="l{parent}"/><child link="l{i}"/>{axis}{lim}</joint>')
    out.append("</robot>")
    with open(path, "w") as f:
        f.write("\n".join(out))
    return path


def random_obstacles(rng, n, reach=0.9, mesh_dir=None):
    """``mesh_dir``: also Mesh obstacles (random polytope files written there; shape kwargs exercised at random)."""
    from geom_truth import random_pose
    from numbotics_amd.physics import Cube, Cuboid, Sphere, Capsule, Cylinder, Plane, Mesh
    import os
    obs = []
    for i in range(n):
        k = int(rng.integers(0, 9 if mesh_dir is not None else 6))
        if k >= 6:
            fn = random_hull_obj(rng, os.path.join(mesh_dir, f"fuzz_obstacle_{i}.obj"), n_objects=int(rng.choice([1, 2, 3])), scale=0.15)
            kw = {}
            if rng.random() < 0.5:
                kw['mesh_scale'] = rng.uniform(0.5, 1.5, 3)
            if rng.random() < 0.5:
                kw['offset'] = random_pose(rng, 0.1)
            if rng.random() < 0.3:
                kw['auto_center'] = True
            if rng.random() < 0.3:
                kw['collision_margin'] = float(rng.choice([0.005, 0.02]))
            obs.append(Mesh(0.0, fn, pose=random_pose(rng, reach), **kw))
            continue
        pose = random_pose(rng, reach)
        margin = float(rng.choice([0.0, 0.0, 0.01, 0.03]))
        if k == 0:
            obs.append(Cube(0.0, float(rng.uniform(0.04, 0.2)), pose=pose))
        elif k == 1:
            obs.append(Cuboid(0.0, rng.uniform(0.05, 0.2, 3), pose=pose, collision_margin=min(margin, 0.04)))
        elif k == 2:
            obs.append(Sphere(0.0, float(rng.uniform(0.03, 0.15)), pose=pose))
        elif k == 3:
            obs.append(Capsule(0.0, float(rng.uniform(0.02, 0.08)), float(rng.uniform(0.1, 0.4)), pose=pose))
        elif k == 4:
            obs.append(Cylinder(0.0, float(rng.uniform(0.04, 0.12)), float(rng.uniform(0.1, 0.3)), pose=pose,
                                collision_margin=min(margin, 0.03)))
        else:
            n_ = rng.normal(size=3)
            n_ /= np.linalg.norm(n_)
            obs.append(Plane(0.0, n_, position=-n_ * float(rng.uniform(0.7, 1.3))))
    return obs
