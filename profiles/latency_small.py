#!/usr/bin/env python3
"""Wall time of one whole extraction (count + host round trip + emit) on the reference's own small volumes, voxels
resident on the device:  python profiles/latency_small.py [name=value ...]  -> one line per volume (median of 200
calls); name=value: development switches of the context (cuberille_debug_set_option)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch
    pkg = graft.load_package()
    ex = pkg.Extractor(0)
    for kv in sys.argv[1:]:
        name, value = kv.split("=")
        ex.debug_option(name, int(value))
    data = os.path.join(ROOT, "tests", "golden", "data")
    for name, iso in [("blob0.mha", 200), ("nucleon.mha", 128), ("fuel.mha", 15), ("silicium.mha", 85), ("hydrogenAtom.mha", 15),
                      ("engine.mha", 100)]:
        path = os.path.join(data, name)
        if not os.path.exists(path):
            continue
        vol = pkg.read_mha(path)
        dev = torch.from_numpy(vol.voxels).cuda()
        desc = pkg.make_desc(vol.voxels.dtype, vol.dims)
        prm = pkg.make_params(iso, triangles=True, project=True, threshold=0.2, step=0.24, relax=0.95, max_steps=100)
        torch.cuda.synchronize()
        walls, devs = [], []
        for i in range(220):
            t0 = time.perf_counter()
            r = ex.extract_device(dev.data_ptr(), desc, prm)
            walls.append(time.perf_counter() - t0)
            devs.append(r.ms_total)
        w = np.median(walls[20:]) * 1e3
        print("%-18s %4dx%4dx%4d  points %8d cells %8d  wall %.3f ms  device stages %.3f ms" %
              (name, *vol.dims, r.n_points, r.n_cells, w, np.median(devs[20:])), flush=True)


if __name__ == "__main__":
    main()
