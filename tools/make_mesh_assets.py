"""Writes the build-authored mesh assets under numbotics_amd/models/ (run once; outputs are committed):

  meshes/link_<name>.obj   faceted stand-ins for the Kinova fixture's collision cylinders: a 10-gon drum whose end caps are
                           chamfered rings (30 vertices, one convex object), in the SAME local frame as the cylinder it replaces
  meshes/bracelet_*.obj    compound link: drum + camera box, one single-object file per <collision> element (one hull each)
  meshes/gripper_base.obj  one box-like object with bevelled edges (24 vertices)
  meshes/rock.obj          a random convex polytope (obstacle)
  meshes/table.obj         compound obstacle: top + four legs = five objects (loaded with convex_decomposition=True: the file is
                           taken as its own decomposition; without the flag it is ONE hull, as in the reference)
  meshes/wedge.stl         binary STL obstacle (a triangular prism)
  kinova_mesh.urdf         kinova_cyl.urdf with every <cylinder>/<box> collision element replaced by a <mesh> (the sphere of
                           the gripper link stays a sphere), same joint tree

Nothing here comes from the reference (its URDFs / meshes are not in the repository snapshot): these files
exist so that BASELINE config 5's "compound-mesh collision shapes" has a concrete, reproducible scene.
"""
import os
import re
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from numbotics_amd.utils.mesh import write_obj, hull_faces          # noqa: E402

MODELS = os.path.join(ROOT, "numbotics_amd", "models")
MESHES = os.path.join(MODELS, "meshes")


def drum(radius, length, n=10, chamfer=0.18):
    """Vertices of an n-gon prism along z whose end rings are pulled in (a chamfered cylinder stand-in), circumscribed by
    the cylinder (radius, length): the hull lies inside the primitive it replaces."""
    ang = 2.0 * np.pi * (np.arange(n) + 0.5) / n
    ring = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    h = length / 2.0
    pts = []
    for z, r in ((-h, radius * (1 - chamfer)), (-h * (1 - chamfer), radius), (h * (1 - chamfer), radius), (h, radius * (1 - chamfer))):
        # merge the two middle rings into one band for short drums so that vertex counts stay modest
        pts.append(np.concatenate([ring * r, np.full((n, 1), z)], axis=1))
    return np.concatenate([pts[0], pts[1], pts[3]]) if length < 0.08 else np.concatenate(pts[:1] + pts[1:3] + pts[3:])


def bevel_box(size, bevel=0.15):
    hx, hy, hz = np.asarray(size) / 2.0
    b = bevel * min(hx, hy, hz)
    pts = []
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                pts.append([sx * (hx - b), sy * (hy - b), sz * hz])
                pts.append([sx * (hx - b), sy * hy, sz * (hz - b)])
                pts.append([sx * hx, sy * (hy - b), sz * (hz - b)])
    return np.asarray(pts)


def obj_part(name, pts, shift=(0.0, 0.0, 0.0)):
    V, F = hull_faces(np.asarray(pts) + np.asarray(shift))
    return (name, V, F)


def write_binary_stl(path, V, F):
    with open(path, "wb") as fh:
        fh.write(b"numbotics_amd wedge".ljust(80, b" "))
        fh.write(struct.pack("<I", len(F)))
        for f in F:
            a, b, c = V[f[0]], V[f[1]], V[f[2]]
            n = np.cross(b - a, c - a)
            n = n / np.linalg.norm(n)
            fh.write(struct.pack("<12fH", *n, *a, *b, *c, 0))


def main():
    os.makedirs(MESHES, exist_ok=True)
    src = open(os.path.join(MODELS, "kinova_cyl.urdf")).read()
    out = src
    # every link's collision elements, in file order
    link_re = re.compile(r'<link name="([^"]+)">(.*?)</link>', re.S)
    coll_re = re.compile(r'<collision>\s*<origin xyz="([^"]+)" rpy="([^"]+)"/>\s*<geometry>(.*?)</geometry>\s*</collision>', re.S)
    for lm in link_re.finditer(src):
        name, body = lm.group(1), lm.group(2)
        colls = list(coll_re.finditer(body))
        if not colls:
            continue
        new_body = body
        if name == "bracelet_link":
            # compound link: one single-object mesh file per <collision> element, placed in the link frame.  (A file with several
            # objects would NOT stay a compound: the reference hands Bullet trimesh's re-export of the loaded file, and trimesh
            # merges the objects of a single-material OBJ into one mesh -> one hull.  numbotics_amd/utils/mesh.py:mesh_hulls.)
            for i, cm in enumerate(colls):
                xyz = np.array([float(v) for v in cm.group(1).split()])
                g = cm.group(3)
                if "cylinder" in g:
                    r, l = float(re.search(r'radius="([^"]+)"', g).group(1)), float(re.search(r'length="([^"]+)"', g).group(1))
                    fn, part = "meshes/bracelet_drum.obj", obj_part("bracelet_drum", drum(r, l), xyz)
                else:
                    size = [float(v) for v in re.search(r'size="([^"]+)"', g).group(1).split()]
                    fn, part = "meshes/bracelet_camera.obj", obj_part("bracelet_camera", bevel_box(size), xyz)
                write_obj(os.path.join(MODELS, fn), [part])
                rep = ('<collision>\n      <origin xyz="0 0 0" rpy="0 0 0"/>\n      <geometry><mesh filename="' + fn + '"/></geometry>\n    </collision>')
                new_body = new_body.replace(cm.group(0), rep)
        else:
            for cm in colls:
                g = cm.group(3)
                if "sphere" in g:
                    continue
                fn = f"meshes/link_{name}.obj"
                if "cylinder" in g:
                    r, l = float(re.search(r'radius="([^"]+)"', g).group(1)), float(re.search(r'length="([^"]+)"', g).group(1))
                    write_obj(os.path.join(MODELS, fn), [obj_part(name, drum(r, l))])
                else:
                    size = [float(v) for v in re.search(r'size="([^"]+)"', g).group(1).split()]
                    fn = "meshes/gripper_base.obj"
                    write_obj(os.path.join(MODELS, fn), [obj_part(name, bevel_box(size))])
                new_body = new_body.replace(g, f'<mesh filename="{fn}"/>')
        out = out.replace(body, new_body)
    out = out.replace('<robot name="', '<robot name="mesh_', 1) if '<robot name="' in out else out
    out = out.replace("collision primitives (cylinders, one box, one sphere, one two-element compound link) are this",
                      "collision MESHES (tools/make_mesh_assets.py: faceted hulls inside the primitives of kinova_cyl.urdf) are this")
    with open(os.path.join(MODELS, "kinova_mesh.urdf"), "w") as fh:
        fh.write(out)
    # obstacles
    rng = np.random.default_rng(2026)
    rock = rng.normal(size=(60, 3))
    rock = rock / np.linalg.norm(rock, axis=1, keepdims=True) * rng.uniform(0.75, 1.0, (60, 1)) * [0.22, 0.16, 0.12]
    write_obj(os.path.join(MESHES, "rock.obj"), [obj_part("rock", rock)])
    top = bevel_box([0.8, 0.5, 0.04], 0.2)
    leg = drum(0.025, 0.36, n=8, chamfer=0.1)
    parts = [obj_part("top", top, (0, 0, 0.38))]
    for i, (sx, sy) in enumerate(((-1, -1), (-1, 1), (1, -1), (1, 1))):
        parts.append(obj_part(f"leg{i}", leg, (sx * 0.34, sy * 0.2, 0.18)))
    write_obj(os.path.join(MESHES, "table.obj"), parts)
    wedge = np.array([[0, 0, 0], [0.3, 0, 0], [0, 0.2, 0], [0, 0, 0.25], [0.3, 0, 0.25], [0, 0.2, 0.25]], dtype=float)
    V, F = hull_faces(wedge)
    write_binary_stl(os.path.join(MESHES, "wedge.stl"), V, F)
    print("wrote", sorted(os.listdir(MESHES)), "and kinova_mesh.urdf")


if __name__ == "__main__":
    main()
