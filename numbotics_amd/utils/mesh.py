"""Mesh ingestion for MESH collision shapes (reference: numbotics/utils/mesh.py:18-37, utils/shape.py:81-94).

Upstream loads the file with trimesh, optionally recentres it on its centre of mass, scales it, applies the shape
offset, optionally splits it with a convex decomposition (trimesh -> vhacdx), writes the result to a temporary OBJ and
hands that file to ``pybullet.createCollisionShape(GEOM_MESH, fileName=...)``.  Without the concave-trimesh flag (which
the reference never sets) Bullet turns every OBJ object of that file into one convex hull and the body into the compound
of those hulls.  So what the collision path sees of a mesh is: a list of convex hulls.

WHICH objects the exported file has is decided by trimesh, not by the source file: ``trimesh.load`` merges the objects /
groups of a single-material OBJ (its loader's defaults ``split_object=False, group_material=True``) and the solids of an
STL into ONE ``Trimesh``, whose export is one object -- Bullet then builds ONE hull of the whole vertex set, however many
``o`` / ``g`` statements the source had.  Only ``convex_decomposition=True`` produces a multi-object export (the
``trimesh.Scene`` of the V-HACD parts).  (A multi-MATERIAL OBJ loads as a ``Scene``, on which the reference's
``center_mass`` / ``convex_decomposition`` calls do not exist; such files are read here like single-material ones.)

This module produces exactly that list -- no temporary file, no PyBullet:

* own readers for Wavefront OBJ (``v`` / ``f`` / ``o`` / ``g``; one part per object or group) and STL (ASCII and binary;
  one part per ``solid``); trimesh / vhacdx are third-party and absent here;
* ``load_mesh`` applies the reference's transform sequence (auto-centre, scale, offset) to the vertices;
* ``mesh_hulls``: ``convex_decomposition=False`` (the default) -> ONE hull of all the file's vertices;
  ``convex_decomposition=True`` -> one hull per object of the file;
* ``convex_hull`` (scipy.spatial.ConvexHull = Qhull) reduces a vertex set to its hull vertices and merged face planes.

``convex_decomposition=True`` cannot be reproduced (V-HACD is not available): a file that already holds several objects
is taken as its own decomposition (one hull per object -- what Bullet would build from an exported scene with those parts);
a single-object file raises ``NotImplementedError`` naming the missing dependency.

Parity: trimesh's loaders and ``center_mass`` are third-party and absent, so this reader is pinned only by its own
tests (unit cubes, known volumes / centroids); hull geometry is pinned against an independent SLSQP solution
(tests/test_mesh.py).
"""
import os
import struct
from dataclasses import dataclass

import numpy as np


@dataclass
class MeshPart:
    """One object of a mesh file: vertices (n,3) float64 and triangle/polygon faces as index lists into them."""
    name: str
    vertices: np.ndarray
    faces: list


@dataclass
class ConvexPart:
    """Convex hull of one part, centred on the mean of its own vertices.

    ``vertices`` (m,3) are relative to ``center``; ``planes`` (f,4) = unit outward normal and offset d, also relative to
    ``center``: a point x (relative) is inside iff n.x <= d for every face.  ``planes`` is empty for a degenerate (flat or
    collinear) point set: such a hull still has a support function, which is all the distance iteration needs.
    """
    center: np.ndarray
    vertices: np.ndarray
    planes: np.ndarray

    @property
    def radius(self) -> float:
        return float(np.sqrt((self.vertices ** 2).sum(axis=1).max())) if len(self.vertices) else 0.0


# ---- readers -----------------------------------------------------------------------------------------------------------
def _finish_parts(all_v, groups):
    parts = []
    V = np.asarray(all_v, dtype=np.float64).reshape(-1, 3)
    for name, faces in groups:
        if not faces:
            continue
        used = sorted({i for f in faces for i in f})
        remap = {g: l for l, g in enumerate(used)}
        parts.append(MeshPart(name, V[used].copy(), [[remap[i] for i in f] for f in faces]))
    if not parts and len(V):
        parts.append(MeshPart("points", V.copy(), []))       # a bare point cloud: its hull is still well defined
    return parts


def read_obj(path: str):
    """Wavefront OBJ -> [MeshPart]; a new part starts at every ``o`` or ``g`` statement that is followed by faces."""
    all_v, groups = [], []
    cur_name, cur_faces = "default", []
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            if not line or line[0] == '#':
                continue
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'v' and len(tok) >= 4:
                all_v.append([float(tok[1]), float(tok[2]), float(tok[3])])
            elif tok[0] == 'f' and len(tok) >= 4:
                idx = []
                for t in tok[1:]:
                    i = int(t.split('/')[0])
                    idx.append(i - 1 if i > 0 else len(all_v) + i)
                if any(i < 0 or i >= len(all_v) for i in idx):
                    raise ValueError(f"{path}: face references vertex outside the file")
                cur_faces.append(idx)
            elif tok[0] in ('o', 'g'):
                if cur_faces:
                    groups.append((cur_name, cur_faces))
                cur_name, cur_faces = (" ".join(tok[1:]) or tok[0]), []
    if cur_faces:
        groups.append((cur_name, cur_faces))
    return _finish_parts(all_v, groups)


def _read_stl_ascii(text: str):
    parts, cur_v, cur_f, name = [], [], [], "solid"
    for line in text.splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] == 'solid':
            name = " ".join(tok[1:]) or "solid"
            cur_v, cur_f = [], []
        elif tok[0] == 'vertex' and len(tok) >= 4:
            cur_v.append([float(tok[1]), float(tok[2]), float(tok[3])])
        elif tok[0] == 'endfacet':
            n = len(cur_v)
            if n >= 3:
                cur_f.append([n - 3, n - 2, n - 1])
        elif tok[0] == 'endsolid':
            if cur_f:
                parts.append(MeshPart(name, np.asarray(cur_v, dtype=np.float64), cur_f))
            cur_v, cur_f = [], []
    if cur_f:
        parts.append(MeshPart(name, np.asarray(cur_v, dtype=np.float64), cur_f))
    return parts


def read_stl(path: str):
    """STL (binary or ASCII) -> [MeshPart]; vertices are not merged (the hull does not care)."""
    with open(path, "rb") as fh:
        data = fh.read()
    if len(data) >= 84:
        n = struct.unpack_from("<I", data, 80)[0]
        if 84 + 50 * n == len(data):                 # the binary layout is self-describing; ASCII files never match it
            rec = np.frombuffer(data, dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', 9), ('a', '<u2')]), count=n, offset=84)
            V = rec['v'].astype(np.float64).reshape(-1, 3)
            return [MeshPart("solid", V, [[3 * i, 3 * i + 1, 3 * i + 2] for i in range(n)])] if n else []
    parts = _read_stl_ascii(data.decode("ascii", errors="replace"))
    if not parts:
        raise ValueError(f"{path}: neither a binary nor an ASCII STL")
    return parts


def read_mesh(path: str):
    ext = os.path.splitext(path)[1].lower()
    if ext == '.obj':
        parts = read_obj(path)
    elif ext == '.stl':
        parts = read_stl(path)
    else:
        raise ValueError(f"unsupported mesh format '{ext}' (OBJ and STL are read; trimesh is not available)")
    if not parts:
        raise ValueError(f"{path}: no geometry")
    return parts


# ---- mass properties (auto_center) ------------------------------------------------------------------------------------
def _triangles(part: MeshPart):
    tris = []
    for f in part.faces:
        for k in range(1, len(f) - 1):          # fan triangulation of polygons
            tris.append((f[0], f[k], f[k + 1]))
    return np.asarray(tris, dtype=np.int64).reshape(-1, 3)


def center_of_mass(parts) -> np.ndarray:
    """Volume centroid of a closed mesh (signed tetrahedra against the origin); for an open or zero-volume mesh the
    area-weighted centroid of its triangles; for a point cloud the vertex mean.  (trimesh's ``center_mass`` makes the same
    distinction; it is third-party and absent, so this is a restatement of the documented behaviour, not pinned.)"""
    vol, vc = 0.0, np.zeros(3)
    area, ac = 0.0, np.zeros(3)
    pts = []
    closed = True
    for p in parts:
        pts.append(p.vertices)
        T = _triangles(p)
        if len(T) == 0:
            closed = False
            continue
        a, b, c = p.vertices[T[:, 0]], p.vertices[T[:, 1]], p.vertices[T[:, 2]]
        v6 = np.einsum('ij,ij->i', a, np.cross(b, c))
        vol += v6.sum() / 6.0
        vc += ((a + b + c) / 4.0 * (v6 / 6.0)[:, None]).sum(axis=0)
        ar = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
        area += ar.sum()
        ac += ((a + b + c) / 3.0 * ar[:, None]).sum(axis=0)
        # closed <=> every undirected edge is used by exactly two triangles (vertices merged by coordinates)
        key = {}
        ids = []
        for v in p.vertices:
            ids.append(key.setdefault(tuple(np.round(v, 12)), len(key)))
        ids = np.asarray(ids)
        E = np.sort(np.concatenate([ids[T[:, [0, 1]]], ids[T[:, [1, 2]]], ids[T[:, [2, 0]]]]), axis=1)
        _, counts = np.unique(E, axis=0, return_counts=True)
        if not np.all(counts == 2):
            closed = False
    if closed and abs(vol) > 1e-18:
        return vc / vol
    if area > 0.0:
        return ac / area
    return np.concatenate(pts).mean(axis=0)


# ---- the reference's load_mesh, returning geometry instead of a temporary file -----------------------------------------
def load_mesh(filename: str, mesh_scale=np.array([1.0, 1.0, 1.0]), offset=np.eye(4), convex_decomposition: bool = False,
              auto_center: bool = False, **kwargs):
    """-> [MeshPart] with the reference's transform sequence applied to the vertices (numbotics/utils/mesh.py:26-31):
    translate by -center_mass (``auto_center``), scale by ``diag(mesh_scale)``, apply ``offset``."""
    try:
        parts = read_mesh(filename)
    except (OSError, ValueError) as e:
        raise ValueError(f"Invalid mesh file: {filename}") from e
    scale = np.asarray(mesh_scale, dtype=np.float64).reshape(3)
    offset = np.asarray(offset, dtype=np.float64).reshape(4, 4)
    cm = center_of_mass(parts) if auto_center else np.zeros(3)
    out = []
    for p in parts:
        V = (p.vertices - cm) * scale
        V = V @ offset[:3, :3].T + offset[:3, 3]
        out.append(MeshPart(p.name, V, p.faces))
    if convex_decomposition and len(out) < 2:
        raise NotImplementedError(
            "convex_decomposition=True needs V-HACD (trimesh -> vhacdx), which is not available: decompose the mesh "
            "offline into one OBJ object per convex part -- every object of the file becomes its own hull, which is what "
            "Bullet builds from the reference's exported decomposition")
    return out


# ---- convex hull -------------------------------------------------------------------------------------------------------
def convex_hull(points) -> ConvexPart:
    """Hull vertices (input order) and merged unit face planes of a point set, centred on the mean of the hull vertices."""
    from scipy.spatial import ConvexHull, QhullError
    P = np.unique(np.asarray(points, dtype=np.float64).reshape(-1, 3), axis=0)
    if len(P) == 0:
        raise ValueError("a hull needs at least one point")
    planes = np.zeros((0, 4))
    V = P
    if len(P) >= 4:
        try:
            hull = ConvexHull(P)
            V = P[hull.vertices]
            eq = hull.equations                                    # n.x + b <= 0 inside, |n| = 1
            key = np.round(eq / max(1.0, np.abs(P).max()), 9)
            _, first = np.unique(key, axis=0, return_index=True)   # Qhull triangulates: merge coplanar facets
            eq = eq[np.sort(first)]
            planes = np.concatenate([eq[:, :3], -eq[:, 3:4]], axis=1)
        except QhullError:
            pass                                                   # flat / collinear: support function only
    c = V.mean(axis=0)
    Vc = V - c
    if len(planes):
        planes = planes.copy()
        planes[:, 3] -= planes[:, :3] @ c
    return ConvexPart(center=c, vertices=np.ascontiguousarray(Vc), planes=np.ascontiguousarray(planes))


def mesh_hulls(filename: str, **kwargs):
    """[ConvexPart] of a MESH shape.  ``convex_decomposition=False``: one hull of every vertex of the (transformed) file -- the
    reference hands Bullet trimesh's re-export of the loaded file, in which the objects of the source are merged into one
    (module docstring).  ``convex_decomposition=True``: one hull per object of the file (the file is its own decomposition).
    Third-party behaviour (trimesh, Bullet) restated from their documentation: parity unpinned."""
    parts = load_mesh(filename, **kwargs)
    if kwargs.get('convex_decomposition', False):
        return [convex_hull(p.vertices) for p in parts]
    return [convex_hull(np.concatenate([p.vertices for p in parts], axis=0))]


# ---- writers (tests, tools, assets) ---------------------------------------------------------------------------------------
def write_obj(path: str, parts):
    """parts: iterable of (name, vertices (n,3), faces [[i,...]]) -- one ``o`` object each."""
    base = 0
    with open(path, "w") as fh:
        fh.write("# written by numbotics_amd.utils.mesh.write_obj\n")
        for name, V, F in parts:
            fh.write(f"o {name}\n")
            for v in np.asarray(V, dtype=np.float64):
                fh.write(f"v {v[0]:.17g} {v[1]:.17g} {v[2]:.17g}\n")
            for f in F:
                fh.write("f " + " ".join(str(base + int(i) + 1) for i in f) + "\n")
            base += len(V)
    return path


def hull_faces(points):
    """(hull vertices, triangle faces) of a point set: what ``write_obj`` needs to store a convex part."""
    from scipy.spatial import ConvexHull
    P = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    hull = ConvexHull(P)
    remap = {g: l for l, g in enumerate(hull.vertices)}
    c = P[hull.vertices].mean(axis=0)
    faces = []
    for simplex, eq in zip(hull.simplices, hull.equations):
        a, b, cc = P[simplex]
        tri = [remap[i] for i in simplex]
        if np.dot(np.cross(b - a, cc - a), eq[:3]) < 0:           # outward winding
            tri = [tri[0], tri[2], tri[1]]
        faces.append(tri)
    return P[hull.vertices], faces
