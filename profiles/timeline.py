#!/usr/bin/env python3
"""Timeline of the cuberille kernels of ONE extraction in a rocprofv3 kernel trace (start / end relative to the
extraction's first kernel, queue): shows which launches of the pass overlap.
  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-slab-probe
  python3 profiles/timeline.py DIR [which extraction, default -4]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "cuberille" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    # extractions: each starts with a sweep kernel that follows a cell kernel; bench.py ends with three stage-timed ones
    # (serial by construction), so the default is the fourth from the end: the last of the timed region
    starts = [i for i, r in enumerate(rows) if "k_classify" in r[2] and (i == 0 or "k_emit_cells" in rows[i - 1][2])]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -4
    a = starts[which]
    b = starts[which + 1] if which + 1 < len(starts) and which != -1 else len(rows)
    rows = rows[a:b]
    t0 = rows[0][0]
    print("# start_us end_us dur_us queue kernel")
    for s, e, name, q in rows:
        short = name.split("(")[0].replace("void cuberille::", "").replace("cuberille::", "")
        print("%9.1f %9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short))


if __name__ == "__main__":
    main()
