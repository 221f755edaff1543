#!/usr/bin/env python3
"""What ONE rank of a multi-GPU run does per step, stage by stage, measured on this GPU: slabs of the 1024^3 Marschner-Lobb volume
(THIN_HALO, through cuberille_step_begin / _end with itself as the only rank) beside the whole volume, under a few development
switches.  python profiles/slab_stages.py [z0:z1 ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
import bench
pkg = g.load_package()
from midas_journal_740_amd.cuberille import minimum_halo
n = 1024
vol = bench.generate_block(pkg, torch, "marschner_lobb", n, 0, n, None, torch.device("cuda", 0))
torch.cuda.synchronize()
prm = pkg.make_params(0.5, triangles=True, project=True, threshold=0.002, step=0.25, relax=0.95, max_steps=50)
whole = pkg.make_desc(np.float32, (n, n, n))
below, above = minimum_halo(whole, prm)
ranges = [tuple(int(v) for v in r.split(":")) for r in sys.argv[1:]] or [(384, 512), (448, 512), (512, 640), (0, 1024)]
for (a, b) in ranges:
    for opts in ((), ("classify_keep_tail=1",), ("count_variant=1",), ("proj_chunk=128",)) if not sys.argv[1:] else ((),):
        ex = pkg.Extractor(0)
        for kv in opts:
            ex.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
        if (a, b) == (0, 1024):
            lo, hi, slab = 0, n, None
        else:
            lo, hi = max(a - below - 1, 0), min(b + above + 1, n)
            slab = pkg._abi.Slab(n, lo, a, b, 0, pkg._abi.SLAB_THIN_HALO, None, None)
        desc = pkg.make_desc(np.float32, (n, n, hi - lo))
        sub = vol[lo:hi]
        ex.debug_option("stage_timing", 1)
        acc = None
        for i in range(8):
            ptr, _ = ex.step_begin(sub.data_ptr(), desc, prm, slab)
            res, done = ex.step_end(ptr, 1, 0)
            assert done
            if i >= 3:
                d = res.as_dict()
                acc = d if acc is None else {k: acc[k] + d[k] for k in d}
        k = 5.0
        print("slices [%d,%d) %-34s points %8d  classify %.3f count %.3f points %.3f project %.3f cells %.3f total %.3f" % (
            a, b, " ".join(opts), res.n_points, acc["ms_classify"] / k, acc["ms_count"] / k, acc["ms_emit_points"] / k,
            acc["ms_project"] / k, acc["ms_emit_cells"] / k, acc["ms_total"] / k), flush=True)
        ex.close()
