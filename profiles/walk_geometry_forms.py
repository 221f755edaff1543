#!/usr/bin/env python3
"""The three forms of the walk kernel by geometry (k_project<T, MODE, GEOM>) on one resident volume: identity (GEOM 2), identity
direction with anisotropic spacing (GEOM 1: a diagonal PhysicalPointToIndex), a rotated direction (GEOM 0) -- each against the
general form forced with proj_ident=0.   python profiles/walk_geometry_forms.py [--size 768]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    import torch
    pkg = graft.load_package()
    n = args.size
    dtype, iso, thr = bench.WORKLOADS["marschner_lobb"]
    vol = bench.generate_block(pkg, torch, "marschner_lobb", n, 0, n, None, torch.device("cuda", 0))
    torch.cuda.synchronize()
    ex = pkg.Extractor(0)
    c, s = np.cos(0.3), np.sin(0.3)
    rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    for name, spacing, direction in (("identity", (1.0, 1.0, 1.0), None), ("anisotropic", (0.7, 0.7, 2.5), None), ("rotated", (0.7, 0.7, 2.5), rot)):
        desc = pkg.make_desc(dtype, (n, n, n), spacing, (0.0, 0.0, 0.0), direction)
        prm = pkg.make_params(iso, triangles=True, project=True, threshold=thr, step=0.25 * min(spacing), relax=0.95, max_steps=50)
        for ident in (0, 1, 0, 1):
            ex.debug_option("defaults", 0)
            ex.debug_option("stage_timing", 1)
            ex.debug_option("proj_ident", ident)
            t = []
            for _ in range(args.reps + 2):
                r = ex.extract_device(vol.data_ptr(), desc, prm)
                t.append(r.ms_project)
            print("%-12s proj_ident=%d  walk %.4f ms  (%d passes, %d points)" % (name, ident, float(np.median(t[2:])), r.proj_iterations, r.n_points), flush=True)


if __name__ == "__main__":
    main()
