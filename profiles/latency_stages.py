#!/usr/bin/env python3
"""Per-stage device times of the reference's small volumes (stage_timing on: an event pair per stage, so the sum is
larger than the un-instrumented extraction of profiles/latency_small.py):  python profiles/latency_stages.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch
    pkg = graft.load_package()
    ex = pkg.Extractor(0)
    ex.debug_option("stage_timing", 1)
    data = os.path.join(ROOT, "tests", "golden", "data")
    for name, iso in [("blob0.mha", 200), ("nucleon.mha", 128), ("fuel.mha", 15), ("silicium.mha", 85), ("hydrogenAtom.mha", 15)]:
        vol = pkg.read_mha(os.path.join(data, name))
        dev = torch.from_numpy(vol.voxels).cuda()
        desc = pkg.make_desc(vol.voxels.dtype, vol.dims)
        prm = pkg.make_params(iso, triangles=True, project=True, threshold=0.2, step=0.24, relax=0.95, max_steps=100)
        rows = []
        for i in range(60):
            r = ex.extract_device(dev.data_ptr(), desc, prm)
            rows.append([r.ms_classify, r.ms_count, r.ms_emit_points, r.ms_project, r.ms_emit_cells, r.ms_total])
        m = np.median(np.array(rows[10:]), axis=0)
        print("%-18s classify %.4f count+scan %.4f heads+points %.4f project %.4f cells %.4f total %.4f  passes %d" %
              (name, *m, r.proj_iterations), flush=True)


if __name__ == "__main__":
    main()
