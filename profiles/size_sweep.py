#!/usr/bin/env python3
"""Per-stage device times over volume sizes and workloads (per-stage events on), with the rates that show where a launch shape
stops fitting: GB/s of the sweep, ps per vertex / per pass of the walk, ps per quad of the cell pass.
  python profiles/size_sweep.py [name=value ...]   (development switches of the context)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
import bench
pkg = g.load_package()
dev = torch.device("cuda", 0)
cases = [("sphere", n) for n in (128, 256, 384, 500, 512, 640, 768, 1000)] + [("noise", n) for n in (128, 256, 384, 512, 768)] + [("marschner_lobb", n) for n in (256, 512, 768, 1024)]
opts = [kv for kv in sys.argv[1:]]
print("options", opts, flush=True)
for wl, n in cases:
    dtype, iso, thr = bench.WORKLOADS[wl]
    vol = bench.generate_block(pkg, torch, wl, n, 0, n, None, dev)
    torch.cuda.synchronize()
    desc = pkg.make_desc(dtype, (n, n, n))
    prm = pkg.make_params(iso, triangles=True, project=True, threshold=thr, step=0.25, relax=0.95, max_steps=50)
    ex = pkg.Extractor(0)
    for kv in opts:
        ex.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    ex.debug_option("stage_timing", 1)
    acc = None
    for i in range(10):
        res = ex.extract_device(vol.data_ptr(), desc, prm)
        if i >= 4:
            d = res.as_dict()
            acc = d if acc is None else {k: acc[k] + d[k] for k in d}
    k = 6.0
    nb = n ** 3 * np.dtype(dtype).itemsize
    V, Q = int(res.n_points), int(res.n_cells) // 2
    c, cn, p, pr, ce, t = (acc[x] / k for x in ("ms_classify", "ms_count", "ms_emit_points", "ms_project", "ms_emit_cells", "ms_total"))
    print("%-15s %5d^3 %-7s V %9d passes/V %5.2f | classify %7.4f ms %5.0f GB/s | count %7.4f | points %7.4f %5.1f ps/V | project %7.4f %6.1f ps/V %5.1f ps/pass | cells %7.4f %5.1f ps/Q | total %7.4f" % (
        wl, n, np.dtype(dtype).name, V, res.proj_iterations / max(V, 1), c, nb / c / 1e6, cn, p, p * 1e9 / max(V, 1), pr, pr * 1e9 / max(V, 1),
        pr * 1e9 / max(int(res.proj_iterations), 1), ce, ce * 1e9 / max(Q, 1), t), flush=True)
    ex.close()
    del vol
