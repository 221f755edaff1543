#!/usr/bin/env python3
"""Same-box A/B of per-context development switches (cuberille_debug_set_option) on a bench workload:
   python profiles/ablate.py [--workload marschner_lobb --size 1024] name=v1,v2,... [name2=...]
Every combination is run `--reps` times on the resident volume; prints the median stage times."""
import argparse
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="marschner_lobb")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--quads", action="store_true", help="quadrilateral cells (no split, no point gathers in the cell pass)")
    ap.add_argument("--pass-only", action="store_true", help="no events between the stages (the production timing): "
                    "ms_pass and ms_total only, the combinations interleaved round by round")
    ap.add_argument("sweeps", nargs="*")
    args = ap.parse_args()
    import torch
    pkg = graft.load_package()
    dev = torch.device("cuda", 0)
    n = args.size
    dtype, iso, thr = bench.WORKLOADS[args.workload]
    vol = bench.generate_block(pkg, torch, args.workload, n, 0, n, None, dev)
    torch.cuda.synchronize()
    ex = pkg.Extractor(0)
    desc = pkg.make_desc(dtype, (n, n, n))
    prm = pkg.make_params(iso, triangles=not args.quads, project=True, threshold=thr, step=0.25, relax=0.95, max_steps=50)
    names, values = [], []
    for sw in args.sweeps:
        k, v = sw.split("=")
        names.append(k)
        values.append([int(x) for x in v.split(",")])
    keys = ["ms_classify", "ms_count", "ms_emit_points", "ms_project", "ms_emit_cells", "ms_total"]
    if args.pass_only:
        combos = list(itertools.product(*values)) if names else [()]
        rows = {cb: [] for cb in combos}
        for rnd in range(args.reps + 1):
            for cb in combos:
                ex.debug_option("defaults", 0)
                for k, v in zip(names, cb):
                    ex.debug_option(k, v)
                for i in range(4):
                    r = ex.extract_device(vol.data_ptr(), desc, prm)
                    if rnd and i:
                        rows[cb].append([r.ms_pass, r.ms_total])
        for cb in combos:
            a = np.array(rows[cb])
            print(" ".join("%s=%d" % kv for kv in zip(names, cb)) or "defaults",
                  "pass %.4f (min %.4f) total %.4f (min %.4f)" % (np.median(a[:, 0]), a[:, 0].min(), np.median(a[:, 1]), a[:, 1].min()),
                  "points", r.n_points, flush=True)
        return
    for combo in itertools.product(*values) if names else [()]:
        ex.debug_option("defaults", 0)
        ex.debug_option("stage_timing", 1)
        for k, v in zip(names, combo):
            ex.debug_option(k, v)
        rows = []
        for _ in range(args.reps + 2):
            r = ex.extract_device(vol.data_ptr(), desc, prm)
            rows.append([getattr(r, k) for k in keys])
        med = np.median(np.array(rows[2:]), axis=0)
        print(" ".join("%s=%d" % kv for kv in zip(names, combo)) or "defaults",
              " ".join("%s %.4f" % (k[3:], m) for k, m in zip(keys, med)),
              "iters", r.proj_iterations, flush=True)


if __name__ == "__main__":
    main()
