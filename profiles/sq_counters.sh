#!/bin/bash
# SQ counters per kernel of one bench configuration, two rocprofv3 --pmc passes (VALU counters; wave / wait counters):
#   bash profiles/sq_counters.sh <tag> [bench args...]   ->  gpurun_out/<tag>_sq_counters.txt
# SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = mean active lanes per vector instruction; SQ_WAIT_ANY / SQ_WAVE_CYCLES = share of
# the wave-cycles parked on a wait; SQ_BUSY_CYCLES in units of the whole chip.
set -e
TAG=${1:-r3}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --kernel-include-regex cuberille --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d "$OUT/a" -o a -- python3 "$R/bench.py" --steps 1 --warmup 1 --cpu-sample 0 --no-slab-probe --no-warm-up "$@" > /dev/null 2> "$OUT/a.err"
echo "pass 1 done"
rocprofv3 --kernel-trace --kernel-include-regex cuberille --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/b" -o b -- python3 "$R/bench.py" --steps 1 --warmup 1 --cpu-sample 0 --no-slab-probe --no-warm-up "$@" > /dev/null 2> "$OUT/b.err"
echo "pass 2 done"
python3 - "$OUT" "$R/gpurun_out/${TAG}_sq_counters.txt" "$*" <<'PY'
import collections, csv, glob, os, sys
src, dst, args = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = (r.get("Kernel_Name") or "").split("(")[0]
        if "cuberille::" in name:
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(dst, "w") as o:
    o.write("# rocprofv3 --kernel-trace --kernel-include-regex cuberille --pmc <counters> -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 %s  (two passes)\n" % args)
    for (k, c), v in sorted(acc.items()):
        o.write("%-60s %-24s launches %d  mean %.4g\n" % (k, c, len(v), sum(v) / len(v)))
print("wrote", dst)
PY
rm -rf "$OUT/a" "$OUT/b"
