"""Host-side scene construction: the POD arrays RVPT uploads to the GPU.

Mirrors the reference's host structs and scene set-up (paths relative to the reference tree):
  * Triangle  — src/rvpt/geometry.h:76-111 (4 x vec4; the face normal rides in the .w lanes)
  * Material  — src/rvpt/material.h:9-26
  * load_model / default scene — src/rvpt/main.cpp:12-62, 102-107
plus the deterministic synthetic scenes BASELINE.json's configs name (Cornell box + subdivided model,
~1M-triangle heightfield).  Pure numpy; no GPU.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

LAMBERT, MIRROR, DIELECTRIC = 0, 1, 2  # Material::Type, material.h:11-16

_ASSETS = Path(__file__).resolve().parent / "assets"


def make_triangles(positions, material_id: int) -> np.ndarray:
    """positions[n,3,3] -> float32[n,16] in the reference Triangle layout (geometry.h:81-91)."""
    p = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3, 3)
    n = p.shape[0]
    out = np.zeros((n, 16), dtype=np.float32)
    nrm = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]).astype(np.float32)
    ln = np.sqrt((nrm * nrm).sum(axis=1, keepdims=True)).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        nrm = (nrm / ln).astype(np.float32)
    out[:, 0:3], out[:, 4:7], out[:, 8:11] = p[:, 0], p[:, 1], p[:, 2]
    out[:, 3], out[:, 7], out[:, 11] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
    out[:, 12] = float(material_id)
    return out


def make_material(albedo, emission, mtype: int) -> np.ndarray:
    """float32[12] = albedo(4), emission(4), data(4) with data.x = type (material.h:17-25).
    albedo[3] doubles as the index of refraction (intersection.glsl:54)."""
    m = np.zeros(12, dtype=np.float32)
    m[0:4] = albedo
    m[4:8] = emission
    m[8] = float(mtype)
    return m


def load_obj_positions(path) -> np.ndarray:
    """Minimal Wavefront OBJ reader: `v` records and `f` records (v, v/vt, v/vt/vn, v//vn, negative
    indices); polygons are fan-triangulated (tinyobjloader's default `triangulate=true`, which is
    what main.cpp:23 uses); normals/uvs/materials are ignored like load_model() (main.cpp:49-59).
    Returns float32[n_tris,3,3]."""
    verts: list[tuple[float, float, float]] = []
    faces: list[tuple[int, int, int]] = []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                a = line.split()
                verts.append((float(a[1]), float(a[2]), float(a[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    v = np.asarray(verts, dtype=np.float32)
    fi = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    return v[fi]


def load_mtl(path) -> dict:
    """Wavefront MTL -> {name: material[12]} in the reference Material layout (material.h:9-26).

    The reference's loader ignores materials (main.cpp:49-59: constant material id); this is the scene-description
    step SURVEY §8(f-2) asks for.  Mapping: `Kd` -> albedo, `Ke` -> emission, `Ni` -> index of refraction
    (albedo.w, intersection.glsl:54); type from `illum`: 3 / 8 (reflection without refraction) -> MIRROR with
    albedo `Ks` when given, 4 / 6 / 7 / 9 (glass / refraction) or `d` < 1 -> DIELECTRIC, anything else -> LAMBERT."""
    mats: dict = {}
    cur = None
    with open(path, "r") as f:
        for line in f:
            a = line.split()
            if not a or a[0].startswith("#"):
                continue
            if a[0] == "newmtl":
                cur = {"Kd": (0.8, 0.8, 0.8), "Ke": (0.0, 0.0, 0.0), "Ks": None, "Ni": 1.5, "illum": 2, "d": 1.0}
                mats[" ".join(a[1:])] = cur
            elif cur is None:
                continue
            elif a[0] in ("Kd", "Ke", "Ks") and len(a) >= 4:
                cur[a[0]] = (float(a[1]), float(a[2]), float(a[3]))
            elif a[0] == "Ni":
                cur["Ni"] = float(a[1])
            elif a[0] == "illum":
                cur["illum"] = int(float(a[1]))
            elif a[0] == "d":
                cur["d"] = float(a[1])
            elif a[0] == "Tr":
                cur["d"] = 1.0 - float(a[1])
    out = {}
    for name, m in mats.items():
        if m["illum"] in (3, 8):
            out[name] = make_material((*(m["Ks"] or m["Kd"]), 0.0), (*m["Ke"], 0.0), MIRROR)
        elif m["illum"] in (4, 6, 7, 9) or m["d"] < 1.0:
            out[name] = make_material((*m["Kd"], m["Ni"]), (*m["Ke"], 0.0), DIELECTRIC)
        else:
            out[name] = make_material((*m["Kd"], 0.0), (*m["Ke"], 0.0), LAMBERT)
    return out


def load_obj_scene(path):
    """OBJ + MTL -> (tris[n,16], mats[m,12], names[m]): `mtllib` files are read relative to the OBJ, `usemtl`
    selects the material of the faces that follow.  Material ids are handed out in order of first use; faces
    before any `usemtl`, or naming a material no library defines, get a white Lambert material called "default"."""
    import os
    verts: list[tuple[float, float, float]] = []
    faces: list[tuple[int, int, int]] = []
    face_mat: list[int] = []
    library: dict = {}
    names: list[str] = []
    mats: list[np.ndarray] = []
    ids: dict = {}

    def material_id(name):
        key = name if name in library else "default"
        if key not in ids:
            ids[key] = len(names)
            names.append(key)
            mats.append(library[key] if key in library else make_material((1, 1, 1, 0), (0, 0, 0, 0), LAMBERT))
        return ids[key]

    cur = None
    base = os.path.dirname(os.path.abspath(path))
    with open(path, "r") as f:
        for line in f:
            a = line.split()
            if not a:
                continue
            if a[0] == "v":
                verts.append((float(a[1]), float(a[2]), float(a[3])))
            elif a[0] == "mtllib":
                for lib in a[1:]:
                    lp = os.path.join(base, lib)
                    if os.path.exists(lp):
                        library.update(load_mtl(lp))
            elif a[0] == "usemtl":
                cur = " ".join(a[1:])
            elif a[0] == "f":
                idx = []
                for tok in a[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                mid = material_id(cur if cur is not None else "default")
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
                    face_mat.append(mid)
    v = np.asarray(verts, dtype=np.float32)
    fi = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    tris = make_triangles(v[fi], 0)
    tris[:, 12] = np.asarray(face_mat, dtype=np.float32)
    return tris, np.stack(mats) if mats else np.zeros((0, 12), np.float32), names


def write_obj(path, positions) -> None:
    """Write de-indexed triangles as OBJ text (used to route generated scenes through the OBJ path)."""
    p = np.asarray(positions, dtype=np.float32).reshape(-1, 3)
    with open(path, "w") as f:
        f.write("# generated by rvpt_amd.scene.write_obj\n")
        for x, y, z in p:
            f.write(f"v {float(x):.9g} {float(y):.9g} {float(z):.9g}\n")
        for t in range(p.shape[0] // 3):
            f.write(f"f {3*t+1} {3*t+2} {3*t+3}\n")


def write_obj_scene(path, tris, mats) -> None:
    """Write (tris[n,16], mats[m,12]) as OBJ + MTL (same stem) so that load_obj_scene() / load_scene() read back the
    same triangles with the same material per triangle (ids are renumbered in order of first use)."""
    import os
    tris = np.asarray(tris, dtype=np.float32).reshape(-1, 16)
    mats = np.asarray(mats, dtype=np.float32).reshape(-1, 12)
    stem = os.path.splitext(str(path))[0]
    with open(stem + ".mtl", "w") as f:
        f.write("# generated by rvpt_amd.scene.write_obj_scene\n")
        for i, m in enumerate(mats):
            t = int(m[8])
            f.write(f"newmtl m{i}\n")
            f.write("Kd {:.9g} {:.9g} {:.9g}\n".format(*map(float, m[0:3])))
            f.write("Ke {:.9g} {:.9g} {:.9g}\n".format(*map(float, m[4:7])))
            if t == MIRROR:
                f.write("Ks {:.9g} {:.9g} {:.9g}\nillum 3\n".format(*map(float, m[0:3])))
            elif t == DIELECTRIC:
                f.write(f"Ni {float(m[3]):.9g}\nillum 7\n")
            else:
                f.write("illum 2\n")
    ids = tris[:, 12].astype(np.int64)
    with open(path, "w") as f:
        f.write("# generated by rvpt_amd.scene.write_obj_scene\n")
        f.write(f"mtllib {os.path.basename(stem)}.mtl\n")
        for t in tris:
            for k in (0, 4, 8):
                f.write(f"v {float(t[k]):.9g} {float(t[k+1]):.9g} {float(t[k+2]):.9g}\n")
        cur = None
        for i in range(tris.shape[0]):
            if ids[i] != cur:
                cur = ids[i]
                f.write(f"usemtl m{cur}\n")
            f.write(f"f {3*i+1} {3*i+2} {3*i+3}\n")


def default_model_positions() -> np.ndarray:
    """De-indexed positions of the default model (143 triangles), see tools/make_default_scene.py."""
    return np.fromfile(_ASSETS / "default_scene_tris.f32", dtype="<f4").reshape(-1, 3, 3).copy()


def default_materials() -> np.ndarray:
    """main.cpp:105-107: id 0 emissive Lambert (unused by the model), id 1 white Lambert."""
    return np.stack([
        make_material((1, 1, 1, 0), (0.1, 0.4, 0.6, 0), LAMBERT),
        make_material((1, 1, 1, 0), (0, 0, 0, 0), LAMBERT),
    ])


def default_scene():
    """(tris[143,16], mats[2,12]) exactly as main() builds them (main.cpp:102-107)."""
    return make_triangles(default_model_positions(), 1), default_materials()


def subdivide(positions, levels: int) -> np.ndarray:
    """Midpoint 1->4 subdivision, `levels` times (deterministic)."""
    p = np.asarray(positions, dtype=np.float32).reshape(-1, 3, 3)
    for _ in range(levels):
        a, b, c = p[:, 0], p[:, 1], p[:, 2]
        ab, bc, ca = ((a + b) * np.float32(0.5)), ((b + c) * np.float32(0.5)), ((c + a) * np.float32(0.5))
        p = np.concatenate([
            np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)
        ]).astype(np.float32)
    return p


def _quad(p0, p1, p2, p3):
    return np.array([[p0, p1, p2], [p0, p2, p3]], dtype=np.float32)


def cornell_scene(subdiv_levels: int = 3):
    """BASELINE config 3: Cornell box (5 walls + ceiling light = 12 triangles) around the default model
    subdivided `subdiv_levels` times (143*4^3 = 9152 triangles).  Materials: 0 white .73, 1 red,
    2 green, 3 light (emission 15), 4 model (Lambert .8).  Box spans x,z in [-2,2], y in [0,4]."""
    lo, hi, y0, y1 = -2.0, 2.0, 0.0, 4.0
    parts, mat_ids = [], []

    def add(q, m):
        parts.append(q)
        mat_ids.extend([m] * q.shape[0])

    add(_quad((lo, y0, lo), (hi, y0, lo), (hi, y0, hi), (lo, y0, hi)), 0)  # floor
    add(_quad((lo, y1, lo), (lo, y1, hi), (hi, y1, hi), (hi, y1, lo)), 0)  # ceiling
    add(_quad((lo, y0, hi), (hi, y0, hi), (hi, y1, hi), (lo, y1, hi)), 0)  # back wall (z = hi)
    add(_quad((lo, y0, lo), (lo, y0, hi), (lo, y1, hi), (lo, y1, lo)), 1)  # left, red
    add(_quad((hi, y0, lo), (hi, y1, lo), (hi, y1, hi), (hi, y0, hi)), 2)  # right, green
    add(_quad((-0.6, y1 - 0.01, -0.6), (-0.6, y1 - 0.01, 0.6), (0.6, y1 - 0.01, 0.6), (0.6, y1 - 0.01, -0.6)), 3)
    model = subdivide(default_model_positions(), subdiv_levels)
    add(model, 4)
    pos = np.concatenate(parts)
    ids = np.asarray(mat_ids)
    tris = make_triangles(pos, 0)
    tris[:, 12] = ids.astype(np.float32)
    mats = np.stack([
        make_material((0.73, 0.73, 0.73, 0), (0, 0, 0, 0), LAMBERT),
        make_material((0.65, 0.05, 0.05, 0), (0, 0, 0, 0), LAMBERT),
        make_material((0.12, 0.45, 0.15, 0), (0, 0, 0, 0), LAMBERT),
        make_material((0.0, 0.0, 0.0, 0), (15, 15, 15, 0), LAMBERT),
        make_material((0.8, 0.8, 0.8, 0), (0, 0, 0, 0), LAMBERT),
    ])
    return tris, mats


def _value_noise(n: int, seed: int, octaves: int = 3) -> np.ndarray:
    rng = np.random.RandomState(seed)
    h = np.zeros((n, n), dtype=np.float64)
    xs = np.linspace(0.0, 1.0, n)
    amp, cells = 1.0, 4
    for _ in range(octaves):
        g = rng.rand(cells + 2, cells + 2)
        fx = xs * cells
        i = np.minimum(fx.astype(np.int64), cells - 1)
        t = fx - i
        t = t * t * (3 - 2 * t)
        rows = g[i][:, i] * (1 - t)[None, :] + g[i][:, i + 1] * t[None, :]
        rows1 = g[i + 1][:, i] * (1 - t)[None, :] + g[i + 1][:, i + 1] * t[None, :]
        h += amp * (rows * (1 - t)[:, None] + rows1 * t[:, None])
        amp *= 0.5
        cells *= 2
    return h


def heightfield_scene(cells: int = 708, seed: int = 1234):
    """BASELINE config 4: cells x cells grid, 2 triangles per cell (708 -> 1 002 528 triangles), height =
    3-octave value noise.  Spans x,z in [-4,4], y in [0,~1.2]; material 0 = Lambert .7 grey."""
    n = cells + 1
    h = (_value_noise(n, seed) * 0.7).astype(np.float32)
    xs = np.linspace(-4.0, 4.0, n, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="xy")
    P = np.stack([X, h, Z], axis=-1)  # [n,n,3], P[j,i] = (x_i, h, z_j)
    p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
    t0 = np.stack([p00, p01, p11], axis=2).reshape(-1, 3, 3)
    t1 = np.stack([p00, p11, p10], axis=2).reshape(-1, 3, 3)
    pos = np.concatenate([t0, t1]).astype(np.float32)
    mats = np.stack([make_material((0.7, 0.7, 0.7, 0), (0, 0, 0, 0), LAMBERT)])
    return make_triangles(pos, 0), mats


def materials_showcase_scene():
    """Small scene exercising all three material types + an emitter (parity stress for the
    mirror / dielectric branches of integrator_Kajiya, integrators.glsl:625-665)."""
    model = default_model_positions()
    parts = [model]
    ids = [2] * model.shape[0]  # glass model
    fl = _quad((-3, 0.0, -3), (3, 0.0, -3), (3, 0.0, 3), (-3, 0.0, 3))
    parts.append(fl); ids += [0, 0]
    mir = _quad((-3, 0.0, 2.0), (3, 0.0, 2.0), (3, 3.0, 2.0), (-3, 3.0, 2.0))
    parts.append(mir); ids += [1, 1]
    light = _quad((-1, 3.5, -1), (-1, 3.5, 1), (1, 3.5, 1), (1, 3.5, -1))
    parts.append(light); ids += [3, 3]
    tris = make_triangles(np.concatenate(parts), 0)
    tris[:, 12] = np.asarray(ids, dtype=np.float32)
    mats = np.stack([
        make_material((0.6, 0.6, 0.6, 0), (0, 0, 0, 0), LAMBERT),
        make_material((0.9, 0.9, 0.9, 0), (0, 0, 0, 0), MIRROR),
        make_material((0.95, 0.95, 0.95, 1.5), (0, 0, 0, 0), DIELECTRIC),
        make_material((0, 0, 0, 0), (8, 8, 8, 0), LAMBERT),
    ])
    return tris, mats
