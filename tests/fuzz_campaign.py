#!/usr/bin/env python3
"""A differential campaign of the HIP path against the oracle, longer and wider than the suite's fixed cases -- run by hand on
a GPU box, its summary committed under profiles/:

    python tests/fuzz_campaign.py --seconds 600 --seed 1 > gpurun_out/fuzz_seed1.log

Every case draws a volume (shape with ragged / whole-word / one-voxel-thick rows, one of the ten pixel types, a noise, blob,
plane or shell field, blanked slices for quirk Q1, non-finite voxels now and then), a geometry (spacing, origin, a rotation or a
shear as direction matrix), the filter's eight parameters, a projection branch, and a route through the C ABI -- host upload,
resident volume, streamed upload, Z-slabs stitched by hand, count + emit, a held gradient (quirk Q3), a development switch
that forces a fallback kernel -- and compares ids, cell order and the float bits of every coordinate with the oracle's mesh
of the same volume.  TEST INFRASTRUCTURE: the oracle is the checker here, never the thing measured or shipped.
Prints one progress line every ~20 s; exit code 1 with the failing case's recipe if a mesh ever differs."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
from conftest import assert_same_mesh  # noqa: E402

DTYPES = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64, np.int64, np.uint64]
XS = [1, 2, 5, 31, 63, 64, 65, 100, 127, 128, 129, 192, 200, 256, 257, 320]


def draw_field(rng, shape, dt):
    """(voxels, iso): the inside set is what matters to the topology, the values to the walk."""
    nz, ny, nx = shape
    kind = rng.choice(["noise", "blobs", "plane", "shell", "smooth_noise"])
    z, y, x = np.meshgrid(np.arange(nz, dtype=np.float64), np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="ij")
    if kind == "noise":
        f = rng.random(shape) - rng.choice([0.05, 0.3, 0.5, 0.8])
    elif kind == "smooth_noise":
        f = rng.random(shape)
        for ax in range(3):
            if shape[ax] > 2:
                f = (f + np.roll(f, 1, ax) + np.roll(f, -1, ax)) / 3.0
        f = f - np.quantile(f, rng.choice([0.2, 0.5, 0.85]))
    elif kind == "blobs":
        f = np.full(shape, -1.0)
        for _ in range(int(rng.integers(1, 5))):
            c = [rng.uniform(0, n) for n in (nx, ny, nz)]
            r = rng.uniform(1.0, 0.6 * max(2.0, min(max(nx, 2), 24)))
            s = rng.uniform(0.5, 2.0, size=3)
            d = np.sqrt(((x - c[0]) / s[0]) ** 2 + ((y - c[1]) / s[1]) ** 2 + ((z - c[2]) / s[2]) ** 2)
            f = np.maximum(f, (r - d) / max(r, 1.0))
    elif kind == "plane":
        n = rng.normal(size=3)
        n /= np.linalg.norm(n) + 1e-9
        f = (x - nx / 2.0) * n[0] + (y - ny / 2.0) * n[1] + (z - nz / 2.0) * n[2] + rng.uniform(-1, 1)
        f = f / (np.abs(f).max() + 1e-9)
    else:
        c = [nx / 2.0 + rng.uniform(-2, 2), ny / 2.0 + rng.uniform(-2, 2), nz / 2.0 + rng.uniform(-2, 2)]
        d = np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2)
        r = rng.uniform(1.5, 0.5 * max(3.0, min(nx, ny * 3, nz * 3)))
        f = (1.0 - np.abs(d - r) / max(r, 1.0)) - 0.8
    dt = np.dtype(dt)
    if dt.kind == "f":
        vox = (f * rng.choice([1.0, 37.5, 1e-3])).astype(dt)
        iso = 0.0
        if rng.random() < 0.06 and vox.size > 8:              # non-finite voxels (quirk Q4's relatives)
            idx = tuple(rng.integers(0, n) for n in shape)
            vox[idx] = rng.choice([np.inf, -np.inf, np.nan])
    else:
        info = np.iinfo(dt)
        lo, hi = (0, min(info.max, 240)) if info.min == 0 else (max(info.min, -120), min(info.max, 120))
        mid = (lo + hi) / 2.0
        g = np.clip(mid + f * (hi - lo) * 0.5 * rng.choice([1.0, 0.3]), lo, hi)
        vox = np.rint(g).astype(dt)
        iso = int(np.rint(mid)) + int(rng.integers(0, 2))
        if dt.itemsize == 8 and rng.random() < 0.5:               # beyond 2^53: a double cannot hold these
            big = (1 << 60) if info.min == 0 else -(1 << 60)
            vox = vox + dt.type(big)
            iso = int(iso) + big
    if rng.random() < 0.25 and nz > 2:
        for _ in range(int(rng.integers(1, 3))):
            vox[int(rng.integers(0, nz))] = vox.min()                 # an empty slice (or a full one when min is inside)
    return np.ascontiguousarray(vox), iso


def to_device(torch, a):
    """A host array on the GPU as a tensor of the same bytes (torch has no arithmetic for some unsigned types: signed views)."""
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "u" and a.dtype.itemsize > 1:
        a = a.view({2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize])
    return torch.from_numpy(a).cuda()


def draw_geometry(rng):
    spacing = tuple(float(v) for v in rng.choice([0.25, 0.5, 1.0, 1.0, 1.7, 3.0], size=3))
    origin = tuple(float(v) for v in rng.normal(0, 7, size=3).round(3)) if rng.random() < 0.7 else (0.0, 0.0, 0.0)
    u = rng.random()
    if u < 0.55:
        d = np.eye(3)
    elif u < 0.85:
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        d = q
    else:
        d = np.eye(3)[rng.permutation(3)] * rng.choice([-1.0, 1.0], size=3)[:, None]
    return spacing, origin, np.ascontiguousarray(d, dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-voxels", type=int, default=900000)
    ap.add_argument("--recursive-gaussian", action="store_true", help="every case that can takes the recursive-Gaussian gradient (h:21)")
    ap.add_argument("--big", action="store_true", help="up to 120 rows and slices (fewer, larger cases: set --max-voxels too)")
    args = ap.parse_args()
    import torch
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    oracle.build()
    rng = np.random.default_rng(args.seed)
    ex = pkg.Extractor(0)
    held = pkg.Extractor(0)
    t0 = last = time.time()
    stats = {"cases": 0, "points": 0, "cells": 0, "routes": {}, "dtypes": {}, "refused_slab_alias": 0, "nan_points": 0}
    case = -1
    try:
        while time.time() - t0 < args.seconds:
            case += 1
            nx = int(rng.choice(XS))
            ny = int(rng.integers(1, 121 if args.big else 41))
            nz = int(rng.integers(1, 121 if args.big else 41))
            while nx * ny * nz > args.max_voxels:
                ny = max(1, ny // 2)
                nz = max(1, nz // 2)
            dt = DTYPES[int(rng.integers(0, len(DTYPES)))]
            vox, iso = draw_field(rng, (nz, ny, nx), dt)
            spacing, origin, direction = draw_geometry(rng)
            # the buffered region's start index (a cropped image keeps the index it was cut at)
            start = tuple(int(v) for v in rng.integers(-3000, 3000, size=3)) if rng.random() < 0.3 else (0, 0, 0)
            project = bool(rng.random() < 0.75)
            variant = int(rng.choice([0, 0, 0, 1, 2])) if project else 0
            kw = dict(triangles=bool(rng.integers(0, 2)), project=project,
                      threshold=float(rng.choice([0.0, 0.01, 0.2, 5.0])) * (1.0 if np.dtype(dt).kind != "f" else 0.05),
                      step=float(rng.choice([0.1, 0.25, 0.6, 1.3])) * min(spacing),
                      relax=float(rng.choice([0.5, 0.9, 0.95, 1.0])), max_steps=int(rng.choice([0, 1, 4, 25, 50])), variant=variant)
            route = str(rng.choice(["host", "device", "stream", "slabs", "thin_slabs", "count_emit", "held", "switch"]))
            if args.recursive_gaussian and project and min(nx, ny, nz) >= 4 and route in ("host", "device", "stream", "count_emit", "switch"):
                kw["gradient"] = 1                         # USE_GRADIENT_RECURSIVE_GAUSSIAN (whole volumes, four voxels along every axis)
            vol = pkg.Volume(vox, spacing=spacing, origin=origin, direction=direction, index_start=start)
            okw = dict(kw, spacing=spacing, origin=origin, direction=direction, index_start=start)
            recipe = dict(case=case, seed=args.seed, shape=[nz, ny, nx], dtype=np.dtype(dt).name, iso=iso, route=route, kw=kw,
                          spacing=spacing, origin=origin, direction=direction.tolist(), index_start=start)
            prm = pkg.make_params(iso, **kw)
            desc = pkg.make_desc(vox.dtype, (nx, ny, nz), spacing, origin, direction, start)
            try:
                ref = None
                if route == "host":
                    ex.extract_host(vol, prm)
                    mesh = ex.download()
                elif route == "device":
                    dev = to_device(torch, vox)
                    ex.extract_device(dev.data_ptr(), desc, prm)
                    mesh = ex.mesh_host()
                    mesh = pkg.Mesh(mesh.points.copy(), mesh.cells.copy())
                elif route == "stream":
                    def source(dst, z0, z1):
                        dst[...] = vox[z0:z1]
                    ex.extract_stream(desc, source, prm)
                    mesh = ex.download()
                elif route == "count_emit":
                    dev = to_device(torch, vox)
                    n_p, n_c = ex.count(dev.data_ptr(), desc, prm)
                    ex.emit(0)
                    mesh = ex.download()
                    assert (n_p, n_c) == (mesh.points.shape[0], mesh.cells.shape[0])
                elif route == "switch":
                    name, val = [("no_cmap", 1), ("no_heads", 1), ("no_vqueue", 1), ("count_variant", 1), ("count_variant", 2),
                                 ("count_variant", 3), ("count_variant", 0), ("points_variant", 2), ("points_variant", 1),
                                 ("points_variant", 0), ("classify_variant", 1), ("proj_literal", 1), ("cmap_linear", 1),
                                 ("no_stream_classify", 1), ("proj_short", 1), ("proj_short", 0), ("count_no_fold", 1)][int(rng.integers(0, 17))]
                    recipe["switch"] = [name, val]
                    ex.debug_option(name, val)
                    try:
                        ex.extract_host(vol, prm)
                        mesh = ex.download()
                    finally:
                        ex.debug_option("defaults", 0)
                elif route == "held":
                    # quirk Q3: a first volume of its own (same pixel type), then this one along the first one's gradient
                    fshape = (int(rng.integers(2, 20)), int(rng.integers(2, 20)), int(rng.choice([3, 17, 64, 70])))
                    fvox, _ = draw_field(rng, fshape, dt)
                    fs, fo, fd = draw_geometry(rng)
                    fstart = tuple(int(v) for v in rng.integers(-100, 100, size=3)) if rng.random() < 0.3 else (0, 0, 0)
                    held.hold_gradient(False)
                    held.hold_gradient(True)
                    if not project:
                        kw["project"] = okw["project"] = True
                        prm = pkg.make_params(iso, **kw)
                    held.extract_host(pkg.Volume(fvox, spacing=fs, origin=fo, direction=fd, index_start=fstart), prm)
                    held.extract_host(vol, prm)
                    mesh = held.download()
                    recipe["first"] = dict(shape=list(fshape), spacing=fs, origin=fo, direction=fd.tolist(), index_start=fstart)
                    ref = oracle.run(vox, iso, first=(fvox, fs, fo, fd, fstart), **okw)
                else:   # slabs stitched by hand: counts, id offsets, concatenation; thin: 3 + 3 halo slices, escaped walks again
                    thin = route == "thin_slabs"
                    occupied = True
                    ins = vox >= (np.dtype(dt).type(iso) if np.dtype(dt).kind != "f" else iso)
                    if np.dtype(dt).kind == "f":
                        ins = ins & ~np.isnan(vox)
                    occupied = bool(ins.reshape(nz, -1).any(axis=1).all())
                    if nz < 3 or not occupied or variant != 0 and (thin or rng.random() < 0.5):
                        route = recipe["route"] = "host"        # (quirk Q1 across a cut needs the ranks' protocol: tests/test_gpu_slabs.py)
                        ex.extract_host(vol, prm)
                        mesh = ex.download()
                    else:
                        below, above = pkg.required_halo(desc, prm)
                        ncut = int(rng.integers(1, min(4, nz - 1) + 1))
                        cuts = [0] + sorted(set(int(v) for v in rng.integers(1, nz, size=ncut))) + [nz]
                        recipe["cuts"] = cuts
                        pts, cells, poff = [], [], 0
                        tb, ta = pkg.cuberille.minimum_halo(desc, prm)
                        for a, b in zip(cuts[:-1], cuts[1:]):
                            lo, hi = (max(a - tb - 1, 0), min(b + ta + 1, nz)) if thin else (max(a - below, 0), min(b + above, nz))
                            dev = to_device(torch, vox[lo:hi])
                            sdesc = pkg.make_desc(vox.dtype, (nx, ny, hi - lo), spacing, origin, direction, start)
                            n_p, n_c = ex.count(dev.data_ptr(), sdesc, prm, pkg._abi.Slab(nz, lo, a, b, 0, pkg._abi.SLAB_THIN_HALO if thin else 0))
                            if thin:
                                ex.emit_points()
                                n_esc = ex.escaped_count()
                                stats["escaped_walks"] = stats.get("escaped_walks", 0) + int(n_esc)
                                if n_esc:
                                    dlo, dhi = max(a - below, 0), min(b + above, nz)
                                    deep = to_device(torch, vox[dlo:dhi])
                                    ex.reproject_escaped(deep.data_ptr(), dlo, dhi - dlo)
                            ex.emit(poff)
                            m = ex.download()
                            pts.append(m.points)
                            cells.append(m.cells)
                            poff += n_p
                        mesh = pkg.Mesh(np.concatenate(pts), np.concatenate(cells))
                if ref is None:
                    ref = oracle.run(vox, iso, **okw)
                assert_same_mesh(mesh, ref)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"FAILED": recipe, "error": "%s: %s" % (type(e).__name__, str(e)[:400])}), flush=True)
                return 1
            stats["cases"] += 1
            stats["points"] += int(mesh.points.shape[0])
            stats["cells"] += int(mesh.cells.shape[0])
            stats["nan_points"] += int(np.isnan(mesh.points).any(axis=1).sum()) if mesh.points.size else 0
            stats["routes"][route] = stats["routes"].get(route, 0) + 1
            stats["dtypes"][np.dtype(dt).name] = stats["dtypes"].get(np.dtype(dt).name, 0) + 1
            if time.time() - last > 20:
                last = time.time()
                print("t %.0f s: %d cases, %d points, %d cells, all identical" % (last - t0, stats["cases"], stats["points"], stats["cells"]), flush=True)
    finally:
        ex.close()
        held.close()
    stats["seconds"] = round(time.time() - t0, 1)
    stats["seed"] = args.seed
    stats["identical"] = True
    print(json.dumps(stats), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
