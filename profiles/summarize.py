#!/usr/bin/env python3
"""Boil the rocprofv3 output of profiles/collect.sh down to the small CSVs kept in this directory.
usage: summarize.py <dir with stats/ fetch/ write/ bench.json> <output prefix>"""
import collections
import csv
import glob
import os
import sys

OURS = ("cuberille::", "rocprim", "hipcub")


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    # per-kernel time: the --stats summary, our kernels only (torch's generator kernels dropped)
    rows = []
    for f in find(os.path.join(src, "stats"), "*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if any(t in r["Name"] for t in OURS):
                rows.append(r)
    if rows:
        with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(sorted(rows, key=lambda r: -float(r["TotalDurationNs"])))
    # HBM counters: mean per dispatch, KiB (rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB)
    acc = collections.defaultdict(list)
    for leg in ("fetch", "write"):
        for f in find(os.path.join(src, leg), "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
                if not any(t in name for t in OURS[:1]):
                    continue
                acc[(name.split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    if acc:
        with open(prefix + "_pmc_hbm.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "dispatches", "mean_value_KB"])
            for (k, c), v in sorted(acc.items()):
                w.writerow([k, c, len(v), sum(v) / len(v)])
    b = os.path.join(src, "bench.json")
    if os.path.exists(b):
        lines = [l for l in open(b) if l.startswith("{")]
        if lines:
            open(prefix + "_bench_under_rocprof.json", "w").write(lines[-1])
    print("wrote", prefix + "_*")


if __name__ == "__main__":
    main()
