#!/usr/bin/env python3
"""Bounded experiment (round-3 review, item 2): can the HBM-bound sweep of one extraction run beside the f64-bound walk of
another?  Two contexts (two streams) on the same resident volume; the one-wait step of each is opened back to back
(cuberille_step_begin returns without waiting) and closed in order.  What the hardware's own scheduling makes of two
independent extractions in flight is the ceiling of any hand-made overlap of sweep and walk INSIDE one extraction (which
adds slab seams, a split launch sequence and a dependency on the previous extraction's surface distribution on top).

  python profiles/overlap_two_contexts.py [--size 1024] [--reps 20]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--workload", default="marschner_lobb")
    ap.add_argument("--opt", action="append", default=[], help="name=value for cuberille_debug_set_option on every context")
    args = ap.parse_args()
    import torch
    import bench
    pkg = graft.load_package()
    n = args.size
    dtype, iso, thr = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    vol = bench.generate_block(pkg, torch, args.workload, n, 0, n, None, dev)
    torch.cuda.synchronize()
    desc = pkg.make_desc(dtype, (n, n, n))
    prm = pkg.make_params(iso, triangles=True, project=True, threshold=thr, step=0.25, relax=0.95, max_steps=50)
    ctx = [pkg.Extractor(0) for _ in range(3)]
    for ex in ctx:
        for kv in args.opt:
            ex.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
        for _ in range(3):
            res = ex.extract_device(vol.data_ptr(), desc, prm)
    print("options %s" % args.opt)
    print("one extraction: %d points, %d cells, device %.4f ms" % (res.n_points, res.n_cells, res.ms_total), flush=True)

    def one_by_one(k):
        t0 = time.perf_counter()
        for i in range(args.reps * k):
            p, _ = ctx[0].step_begin(vol.data_ptr(), desc, prm)
            ctx[0].step_end(p, 1, 0)
        return (time.perf_counter() - t0) / (args.reps * k) * 1e3

    def in_flight(k):
        """k extractions open at once, each on its own context and stream"""
        t0 = time.perf_counter()
        for i in range(args.reps):
            rows = [ctx[j].step_begin(vol.data_ptr(), desc, prm)[0] for j in range(k)]
            for j in range(k):
                _, done = ctx[j].step_end(rows[j], 1, 0)
                assert done
        return (time.perf_counter() - t0) / (args.reps * k) * 1e3

    def staggered():
        """two contexts, one always a step ahead: begin(i+1) is issued before end(i) is waited for"""
        t0 = time.perf_counter()
        row = ctx[0].step_begin(vol.data_ptr(), desc, prm)[0]
        for i in range(args.reps * 2):
            nxt = ctx[(i + 1) & 1].step_begin(vol.data_ptr(), desc, prm)[0]
            _, done = ctx[i & 1].step_end(row, 1, 0)
            assert done
            row = nxt
        ctx[(args.reps * 2) & 1].step_end(row, 1, 0)
        return (time.perf_counter() - t0) / (args.reps * 2 + 1) * 1e3

    for rnd in range(2):
        print("round %d: one by one %.4f ms per extraction | 2 in flight %.4f | 3 in flight %.4f | staggered pair %.4f" % (
            rnd, one_by_one(2), in_flight(2), in_flight(3), staggered()), flush=True)
    # the meshes of overlapped extractions are the same bytes
    a = ctx[0].download()
    b = ctx[1].download()
    assert np.array_equal(a.cells, b.cells) and np.array_equal(a.points.view(np.uint32), b.points.view(np.uint32))
    print("meshes of the overlapped extractions: identical")


if __name__ == "__main__":
    main()
