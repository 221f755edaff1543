"""Helpers shared by the -m gpu test files."""
import os

import numpy as np


def run_gpu(pkg, extractor, vol, iso, **kw):
    prm = pkg.make_params(iso, **kw)
    extractor.extract_host(vol, prm)
    return extractor.download()


def _host_threads():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _read_vtk_polydata(path):
    tok = open(path).read().split()
    i = tok.index("POINTS")
    n = int(tok[i + 1])
    pts = np.array(tok[i + 3:i + 3 + 3 * n], dtype=np.float64).reshape(n, 3)
    j = tok.index("POLYGONS")
    nc, total = int(tok[j + 1]), int(tok[j + 2])
    flat = np.array(tok[j + 3:j + 3 + total], dtype=np.int64)
    k = int(flat[0]) if nc else 0
    cells = flat.reshape(nc, k + 1)[:, 1:] if nc else np.zeros((0, 3), dtype=np.int64)
    return pts, cells


def _closed_form_counts_torch(ins):
    """(#points, #quads) of the closed form on a bool tensor [z,y,x] on the GPU."""
    import torch
    quads = 0
    for z0 in range(0, ins.shape[0], 64):
        a = ins[z0:z0 + 65]                  # one plane of overlap for the z-faces between chunks
        quads += int((a[1:] != a[:-1]).sum()) + int((a[:64, 1:] != a[:64, :-1]).sum()) \
            + int((a[:64, :, 1:] != a[:64, :, :-1]).sum())
    nz, ny, nx = ins.shape
    points = 0
    for c0 in range(0, nz + 1, 32):          # corner planes c0..c1-1, from voxel planes clamp(c-1), clamp(c); chunks
        c1 = min(c0 + 32, nz + 1)            # keep every tensor far below 2^31 elements
        zi = torch.arange(c0 - 1, c1, device=ins.device).clamp_(0, nz - 1)
        p = ins[zi]
        p = torch.cat([p[:, :1], p, p[:, -1:]], 1)
        p = torch.cat([p[:, :, :1], p, p[:, :, -1:]], 2)
        all_in = torch.ones((c1 - c0, ny + 1, nx + 1), dtype=torch.bool, device=ins.device)
        any_in = torch.zeros_like(all_in)
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    s_ = p[dz:dz + c1 - c0, dy:dy + ny + 1, dx:dx + nx + 1]
                    all_in &= s_
                    any_in |= s_
        points += int((any_in & ~all_in).sum())
    return points, quads


def _bench_field(pkg, field, n):
    if field == "sphere_sdf":
        return pkg.volumes.sphere_sdf(n)
    if field == "gradient_noise":
        return pkg.volumes.gradient_noise(n, n, n * 1000000, 0, n)
    raise ValueError(field)
