#!/usr/bin/env python3
"""Fuzz of the packet kernel's two exact culls (the screen rectangles of camera rounds, the bounce cull's half-space table) on scenes nobody
designed (GPU box).  Every case is a random brute-force scene of 1 .. 1024 triangles, a random camera and image, and asserts

  (i)   the device selftests: no (ray, triangle) pair the float test ACCEPTS lies outside the triangle's rectangle / outside the row of the triangle the
        segment leaves (rvpt_hip_selftest_camera_rects, rvpt_hip_selftest_bounce_cull: outside == 0) — the bounce table only where the launch would use it
        (its premise is checked per launch: camera within 64 scene scales of the origin; beyond it the count is recorded, not asserted);
  (ii)  RVPT_HIP_PACKETS_CULL x RVPT_HIP_PACKETS_BOUNCE_CULL in {0, 1}^2, and RVPT_HIP_PACKETS_BOX_CULL = 0 (the leaf boxes of the bounce rounds, round 6), render the
        same bits and trace the same number of segments;
  (iii) the CPU oracle's brute-force variant (oracle/rvpt_oracle.c; the accept rule of intersection.glsl:267-323, the bounce of integrators.glsl:574-671)
        agrees bit for bit wherever it finishes within the case's budget.

Scene families: triangle soups with log-uniform sizes over three decades; soups with exact duplicates, coplanar overlaps, interpenetrating pairs and
zero-area triangles; slivers swept through both thresholds of the culls' premises (sin^2 of the edge angle = 2^-12 and 2^-6, each x 1/4 .. 4); a box of twelve
large triangles around small geometry (the 155-triangle Cornell among them); the default model and the materials showcase; small terrains.  Materials: Lambert,
mirror, glass with ior 1.5 / 1.33 / 0.7 / 1 / 0 / 2.4 / 1e-3 / 40, emitters.  The whole scene (and its camera) scaled by 2^-20 .. 2^20 and translated by
0 .. 128 scene scales (both sides of the 64-scale guard).  Cameras: orbiting, inside the geometry, ON a triangle's plane, looking away, fov 1 .. 179 degrees;
images with partial tiles, up to three frames, one launch per frame or batched, optionally a camera that moves between launches in flight.

    python tools/fuzz_culls.py [n_cases] [seed] [--out FILE] [--first K] [-v]        FUZZ_ONLY=<case> replays one case verbosely
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from rvpt_amd import native, scene  # noqa: E402
from rvpt_amd.camera import Camera  # noqa: E402
from rvpt_amd.renderer import RenderSettings  # noqa: E402

KINDS = ("soup", "soup_dup", "slivers", "box", "model", "terrain")
KIND_P = (0.28, 0.16, 0.16, 0.22, 0.10, 0.08)
IORS = (1.5, 1.33, 0.7, 1.0, 0.0, 2.4, 1e-3, 40.0)
TRANSLATE_SCALES = (0.5, 8.0, 48.0, 63.0, 65.0, 80.0, 128.0)
FOVS = (1.0, 5.0, 30.0, 60.0, 90.0, 120.0, 150.0, 170.0, 179.0)


def _unit(v):
    return v / np.linalg.norm(v)


def _soup(rng, n, lo_decade=-3.0):
    centers = rng.uniform(-1.0, 1.0, (n, 1, 3))
    size = 10.0 ** rng.uniform(lo_decade, 0.0, (n, 1, 1))
    return centers + size * rng.normal(size=(n, 3, 3))


def _sliver(rng, kappa):
    """a triangle whose edges e0, e1 make an angle with sin^2 = kappa (what both culls call well shaped or not)."""
    e0 = _unit(rng.normal(size=3)) * 10.0 ** rng.uniform(-2.0, 0.0)
    perp = _unit(np.cross(e0, rng.normal(size=3)))
    th = math.asin(math.sqrt(min(1.0, kappa)))
    length = np.linalg.norm(e0) * 10.0 ** rng.uniform(-0.5, 0.5)
    e1 = length * (math.cos(th) * _unit(e0) + math.sin(th) * perp)
    v0 = rng.uniform(-1.0, 1.0, 3)
    return np.stack([v0, v0 + e0, v0 + e1])


def _box(half, center):
    """twelve triangles: the six faces of an axis-aligned box (orientation is irrelevant: the shader turns the normal against the ray)"""
    c = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], float) * half + center
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    out = []
    for a, b, cc, d in quads:
        out += [[c[a], c[b], c[cc]], [c[a], c[cc], c[d]]]
    return np.array(out)


def make_scene(rng):
    kind = str(rng.choice(KINDS, p=KIND_P))
    if kind == "soup":
        pos = _soup(rng, int(2 ** rng.uniform(0, 10)))
    elif kind == "soup_dup":
        base = _soup(rng, int(2 ** rng.uniform(1, 9)))
        extra = []
        for _ in range(int(rng.integers(1, max(2, len(base) // 2 + 1)))):
            t = base[int(rng.integers(len(base)))]
            how = int(rng.integers(4))
            if how == 0:  # an exact duplicate: ties on (t, index)
                extra.append(t.copy())
            elif how == 1:  # coplanar, overlapping: shifted inside its own plane
                e0, e1 = t[1] - t[0], t[2] - t[0]
                extra.append(t + rng.uniform(-0.6, 0.6) * e0 + rng.uniform(-0.6, 0.6) * e1)
            elif how == 2:  # interpenetrating: a triangle through the first one's centroid
                c = t.mean(axis=0)
                s = np.linalg.norm(t[1] - t[0]) + np.linalg.norm(t[2] - t[0])
                extra.append(c + 0.5 * s * rng.normal(size=(3, 3)) * np.array([[1.0], [1.0], [0.0]]) - 0.5 * s * rng.normal(size=(1, 3)) * np.array([[0.0], [0.0], [1.0]]))
            else:  # zero area (two equal vertices): a record with n = 0
                d = t.copy()
                d[2] = d[1]
                extra.append(d)
        pos = np.concatenate([base, np.array(extra)])[:1024]
    elif kind == "slivers":
        n = int(2 ** rng.uniform(2, 8))
        tris = []
        for _ in range(n):
            if rng.random() < 0.7:
                thr = 2.0 ** (-12 if rng.random() < 0.5 else -6)
                tris.append(_sliver(rng, thr * float(rng.choice([0.25, 0.5, 0.99, 1.0, 1.01, 2.0, 4.0]))))
            else:
                tris.append(_soup(rng, 1, -1.5)[0])
        pos = np.array(tris)
    elif kind == "box":
        inner = str(rng.choice(["model", "soup", "one"]))
        if inner == "model":
            inside = scene.default_model_positions().astype(np.float64)
        elif inner == "soup":
            inside = _soup(rng, int(2 ** rng.uniform(0, 9)), -2.0)
        else:
            inside = _soup(rng, 1, -1.0)
        lo, hi = inside.reshape(-1, 3).min(0), inside.reshape(-1, 3).max(0)
        half = 0.5 * (hi - lo).max() * float(rng.choice([1.05, 2.0, 8.0, 30.0]))
        pos = np.concatenate([_box(half, 0.5 * (lo + hi)), inside])[:1024]
    elif kind == "model":
        if rng.random() < 0.5:
            pos = scene.default_model_positions().astype(np.float64)
        else:
            t, _ = scene.materials_showcase_scene()
            pos = np.stack([t[:, 0:3], t[:, 4:7], t[:, 8:11]], 1).astype(np.float64)
    else:
        t, _ = scene.heightfield_scene(cells=int(rng.integers(2, 23)), seed=int(rng.integers(1 << 30)))
        pos = np.stack([t[:, 0:3], t[:, 4:7], t[:, 8:11]], 1).astype(np.float64)
    # materials
    n_mats = int(rng.integers(1, 9))
    mats = []
    for _ in range(n_mats):
        mtype = int(rng.choice([0, 1, 2], p=[0.5, 0.2, 0.3]))
        ior = float(rng.choice(IORS)) if mtype == 2 else float(rng.choice([0.0, 1.5]))
        emis = rng.uniform(0.0, 4.0, 3) if rng.random() < 0.25 else np.zeros(3)
        mats.append(scene.make_material((*rng.uniform(0.2, 1.0, 3), ior), (*emis, 0.0), mtype))
    mats = np.stack(mats)
    ids = rng.integers(0, n_mats, len(pos)) if rng.random() < 0.6 else np.full(len(pos), int(rng.integers(n_mats)))
    return kind, pos, mats, ids


def look_rotation(forward):
    """Euler angles (degrees) of rvpt_amd.camera whose +Z is `forward`: M = T R_up(rx) R_right(ry) R_fwd(rz), forward = (sin rx cos ry, -sin ry, cos rx cos ry)."""
    f = _unit(np.asarray(forward, float))
    return np.array([math.degrees(math.atan2(f[0], f[2])), math.degrees(-math.asin(max(-1.0, min(1.0, f[1])))), 0.0])


def make_camera(rng, pos, aspect):
    pts = pos.reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0)
    center, extent = 0.5 * (lo + hi), max(float((hi - lo).max()), 1e-6)
    kind = str(rng.choice(["orbit", "orbit", "inside", "onplane", "behind", "far"]))
    c = Camera(aspect)
    if kind in ("orbit", "far"):
        r = extent * (float(rng.uniform(0.8, 4.0)) if kind == "orbit" else float(rng.uniform(20.0, 200.0)))
        p = center + r * _unit(rng.normal(size=3))
        target = center + 0.3 * extent * rng.normal(size=3)
        c.translation, c.rotation = p, look_rotation(target - p)
    elif kind == "inside":
        c.translation = lo + rng.uniform(0.2, 0.8, 3) * (hi - lo)
        c.rotation = rng.uniform(-180, 180, 3)
    elif kind == "onplane":
        t = pos[int(rng.integers(len(pos)))]
        a, b = (rng.uniform(0.1, 0.4, 2) if rng.random() < 0.5 else rng.uniform(-1.0, 2.0, 2))
        c.translation = t[0] + a * (t[1] - t[0]) + b * (t[2] - t[0])
        c.rotation = rng.uniform(-180, 180, 3)
    else:
        p = center + extent * float(rng.uniform(1.0, 3.0)) * _unit(rng.normal(size=3))
        c.translation, c.rotation = p, look_rotation(p - center)
    c.rotation[2] = float(rng.uniform(-30, 30)) if rng.random() < 0.3 else 0.0
    c.set_fov(float(np.clip(float(rng.choice(FOVS)) * float(rng.uniform(0.9, 1.1)), 1.0, 179.0)))
    return kind, c


def make_case(seed, idx):
    rng = np.random.default_rng([seed, idx])
    kind, pos, mats, ids = make_scene(rng)
    big = rng.random() < 0.06
    if big:
        W, H = 640, 360
    else:
        while True:
            W, H = int(rng.integers(17, 501)), int(rng.integers(5, 281))
            if W * H <= 130000:
                break
    cam_kind, cam = make_camera(rng, pos, W / H)
    moving = rng.random() < 0.2
    cam2 = None
    if moving:
        _, cam2 = make_camera(rng, pos, W / H)
    # the similarity transform: scale 2^k, then a translation of 0 .. 128 scene scales (scene scale as upload_scene computes it: largest |coordinate| + largest extent)
    k = 0 if rng.random() < 0.4 else int(rng.integers(-20, 21))
    s = 2.0 ** k
    pts = pos.reshape(-1, 3) * s
    scale0 = float(np.abs(pts).max() + (pts.max(0) - pts.min(0)).max())
    T = np.zeros(3)
    far = 0.0
    if rng.random() < 0.5:
        far = float(rng.choice(TRANSLATE_SCALES))
        T = _unit(rng.normal(size=3)) * far * scale0
    tris = scene.make_triangles((pos * s + T).astype(np.float32), 0)
    tris[:, 12] = ids.astype(np.float32)
    cams = []
    for c in (cam, cam2):
        if c is None:
            continue
        c.translation = np.asarray(c.translation, float) * s + T
        cams.append(np.array(c.get_data(), np.float32))
    frames = 4 if big else int(rng.integers(1, 4))
    batched = big or (not moving and rng.random() < 0.5)
    return dict(idx=idx, kind=kind, cam_kind=cam_kind, tris=tris, mats=mats.astype(np.float32), cams=cams, W=W, H=H, k=k, far=far, moving=moving,
                frames=frames, batched=batched, max_bounces=int(rng.integers(1, 9)), aa=int(rng.choice([1, 1, 1, 2, 3])), big=big)


def launches(case):
    """[(first frame, frames in the launch, camera)]: one launch per frame, or everything as one launch; a moving camera takes the second pose from frame 1 on"""
    if case["batched"]:
        return [(0, case["frames"], case["cams"][0])]
    return [(f, 1, case["cams"][1 if (case["moving"] and f >= 1) else 0]) for f in range(case["frames"])]


def render_gpu(case, cull, bounce_cull, box_cull=True, interleave=None):
    if interleave is None:
        os.environ.pop("RVPT_HIP_PACKETS_INTERLEAVE", None)
    else:  # the claim order of the packet kernel (round 6): 0 = tile-linear, g = groups of g blocks dealt from all over the frame (launches below four frames), -g = every launch
        os.environ["RVPT_HIP_PACKETS_INTERLEAVE"] = str(interleave)
    os.environ["RVPT_HIP_PACKETS_CULL"] = "1" if cull else "0"
    os.environ["RVPT_HIP_PACKETS_BOUNCE_CULL"] = "1" if bounce_cull else "0"
    os.environ["RVPT_HIP_PACKETS_BOX_CULL"] = "1" if box_cull else "0"
    ctx = native.Context(case["W"], case["H"], 0, 0, 1, native.COUNT_SEGMENTS, lab=False)  # the SHIPPED kernels (the release library reads these two knobs for exactly this A/B)
    try:
        ctx.upload_scene(None, case["tris"], case["mats"])
        info = 0
        for f0, n, cam in launches(case):  # no wait in between: the launches are in flight together
            rs = RenderSettings(max_bounces=case["max_bounces"], aa=case["aa"], current_frame=f0)
            ctx.set_frame(rs.pack(), cam)
            ctx.dispatch() if n == 1 else ctx.dispatch_frames(n)
            info |= ctx.cull_info() | (8 if ctx.launch_info()[2] == 6 else 0)
        img = ctx.read()
        seg = ctx.stats()[0]
    finally:
        ctx.close()
    return img, seg, info


def selftests(case):
    """device selftests on the first camera (and the second, if the camera moves): (rect counts, bounce counts) summed"""
    os.environ["RVPT_HIP_PACKETS_CULL"] = "1"
    os.environ["RVPT_HIP_PACKETS_BOUNCE_CULL"] = "1"
    os.environ["RVPT_HIP_PACKETS_BOX_CULL"] = "1"
    rect = np.zeros(4, np.int64)
    bounce = np.zeros(8, np.int64)
    ctx = native.Context(case["W"], case["H"], 0, 0, 1, 0, lab=True)  # the selftests live in the laboratory build; its tables and rectangles are the release build's
    try:
        ctx.upload_scene(None, case["tris"], case["mats"])
        for cam in case["cams"]:
            rs = RenderSettings(max_bounces=case["max_bounces"], aa=1, current_frame=0)
            ctx.set_frame(rs.pack(), cam)
            rect += np.array(ctx.selftest_camera_rects(2)[0], np.int64)
            bounce += np.array(ctx.selftest_bounce_cull(2), np.int64)
    finally:
        ctx.close()
    return rect, bounce


def render_oracle(case):
    from oracle import oracle
    prev, seg = None, 0
    for f0, n, cam in launches(case):
        for f in range(f0, f0 + n):
            s = oracle.settings_bytes(max_bounces=case["max_bounces"], aa=case["aa"], current_frame=f)
            prev, st = oracle.render(s, cam, None, case["tris"], case["mats"], case["W"], case["H"], oracle.TRAVERSAL_BRUTE, prev=prev)
            seg += int(st[0])
    return prev, seg


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def same_values(a, b):  # NaN payloads may differ between the CPU and the GPU; NaN positions and everything else may not
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=-7.0).view(np.uint32), np.nan_to_num(b, nan=-7.0).view(np.uint32))


def run_case(case, oracle_budget):
    problems = []
    img, seg, info = render_gpu(case, True, True)
    for cull, bc, box in ((False, True, True), (True, False, True), (False, False, True), (True, True, False)):
        img2, seg2, _ = render_gpu(case, cull, bc, box)
        if not same_bits(img, img2) or seg != seg2:
            problems.append(f"cull={int(cull)} bounce_cull={int(bc)} box_cull={int(box)}: {int((img.view(np.uint32) != img2.view(np.uint32)).any(axis=2).sum())} pixels differ, segments {seg2} vs {seg}")
    for order in (0, -1 if case["idx"] % 2 else -4):  # the claim order never shows in the image either
        img2, seg2, _ = render_gpu(case, True, True, True, interleave=order)
        if not same_bits(img, img2) or seg != seg2:
            problems.append(f"interleave={order}: {int((img.view(np.uint32) != img2.view(np.uint32)).any(axis=2).sum())} pixels differ, segments {seg2} vs {seg}")
    os.environ.pop("RVPT_HIP_PACKETS_INTERLEAVE", None)
    rect, bounce = selftests(case)
    if rect[1] != 0:
        problems.append(f"camera rectangles exclude {int(rect[1])} accepted pairs of {int(rect[0])}")
    guard_on = bool(info & 2)
    if bounce[1] != 0 and guard_on:
        problems.append(f"bounce table excludes {int(bounce[1])} accepted pairs of {int(bounce[0])}")
    if bounce[4] != 0 and guard_on:
        problems.append(f"leaf boxes exclude {int(bounce[4])} accepted pairs of {int(bounce[0])}")
    cost = case["W"] * case["H"] * len(case["tris"]) * case["aa"] * case["frames"]
    checked = False
    if cost <= oracle_budget:
        ref, seg_ref = render_oracle(case)
        checked = True
        if not same_values(img, ref) or seg != seg_ref:
            problems.append(f"oracle: {int((np.nan_to_num(img, nan=-7.0) != np.nan_to_num(ref, nan=-7.0)).any(axis=2).sum())} pixels differ, segments {seg} vs {seg_ref}")
    return dict(problems=problems, rect=rect, bounce=bounce, info=info, oracle=checked, seg=seg)


def describe(case):
    return (f"case {case['idx']}: {case['kind']} n={len(case['tris'])} mats={len(case['mats'])} {case['W']}x{case['H']} cam={case['cam_kind']} scale=2^{case['k']} "
            f"translate={case['far']:g} scales moving={case['moving']} frames={case['frames']} batched={case['batched']} bounces={case['max_bounces']} aa={case['aa']}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n_cases", nargs="?", type=int, default=200)
    ap.add_argument("seed", nargs="?", type=int, default=1)
    ap.add_argument("--first", type=int, default=0, help="index of the first case (a run can be split over several calls)")
    ap.add_argument("--out", default=None, help="append the summary to this file")
    ap.add_argument("--oracle-budget", type=float, default=1.2e8, help="largest pixels x triangles x aa x frames the CPU oracle is asked for")
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    only = int(os.environ["FUZZ_ONLY"]) if os.environ.get("FUZZ_ONLY") else None
    t0 = time.time()
    bad, n_oracle, n_guard_off, n_packets, n_aligned = [], 0, 0, 0, 0
    rect_tot, bounce_tot, beyond_guard = np.zeros(4, np.int64), np.zeros(8, np.int64), np.zeros(2, np.int64)
    by_kind = {}
    seg_tot = 0
    for idx in range(a.first, a.first + a.n_cases):
        if only is not None and idx != only:
            continue
        case = make_case(a.seed, idx)
        if a.v or only is not None:
            print(describe(case), flush=True)
        try:
            r = run_case(case, a.oracle_budget)
        except Exception as e:  # a case that cannot run is a failure of the fuzz, not a pass
            r = dict(problems=[f"exception: {e!r}"], rect=np.zeros(4, np.int64), bounce=np.zeros(8, np.int64), info=0, oracle=False, seg=0)
        k = by_kind.setdefault(case["kind"], [0, 0])
        k[0] += 1
        n_oracle += int(r["oracle"])
        n_packets += int(bool(r["info"] & 8))
        n_aligned += int(bool(r["info"] & 4))
        rect_tot += r["rect"]
        seg_tot += r["seg"]
        if r["info"] & 2:
            bounce_tot += r["bounce"]
        else:
            n_guard_off += 1
            beyond_guard += r["bounce"][:2]
        if r["problems"]:
            k[1] += 1
            bad.append(idx)
            print("FAIL " + describe(case))
            for p in r["problems"]:
                print("     " + p)
    n = a.n_cases if only is None else 1
    lines = [
        f"fuzz_culls seed {a.seed} cases {a.first}..{a.first + a.n_cases - 1}: {n - len(bad)}/{n} pass, {len(bad)} fail {bad[:20]}  ({time.time() - t0:.0f} s)",
        f"  packet kernel in {n_packets} cases, camera rounds block-aligned in {n_aligned}; oracle (brute, CPU) compared in {n_oracle}; segments traced {seg_tot}",
        f"  camera rectangles: accepted pairs {int(rect_tot[0])}, outside their rectangle {int(rect_tot[1])}; rectangles hold {int(rect_tot[2])} of {int(rect_tot[3])} (block, triangle) pairs"
        f" = {rect_tot[2] / max(1, rect_tot[3]):.3f}",
        f"  bounce table (launches that use it): accepted pairs {int(bounce_tot[0])}, outside the row {int(bounce_tot[1])}; rows hold {int(bounce_tot[2])} of {int(bounce_tot[3])} bits"
        f" = {bounce_tot[2] / max(1, bounce_tot[3]):.3f}",
        f"  leaf boxes (the same launches): accepted pairs whose ray fails its triangle's box {int(bounce_tot[4])}; a ray passes {int(bounce_tot[6])} of {int(bounce_tot[5])} boxes"
        f" = {bounce_tot[6] / max(1, bounce_tot[5]):.3f}",
        f"  launches beyond the table's premise (camera > 64 scene scales out, or no table): {n_guard_off} cases; the table would have excluded {int(beyond_guard[1])} of {int(beyond_guard[0])} accepted pairs there",
        "  by family: " + ", ".join(f"{k} {v[0]} ({v[1]} fail)" for k, v in sorted(by_kind.items())),
    ]
    print("\n".join(lines))
    if a.out:
        with open(a.out, "a") as f:
            f.write("\n".join(lines) + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
