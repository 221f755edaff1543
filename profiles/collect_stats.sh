#!/bin/bash
# The kernel-trace + stats pass of collect.sh alone (for workloads on which rocprofv3's counter passes die):
#   bash profiles/collect_stats.sh <tag> [bench args...]
# -> gpurun_out/<tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json (summarize.py); the raw trace is removed.
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$R/bench.py" --steps 5 --warmup 2 --cpu-sample 0 --no-slab-probe --no-warm-up "$@" > "$OUT/bench.json" 2> "$OUT/stats.err" \
  || { echo "stats pass FAILED (see $OUT/stats.err)"; exit 1; }
python3 "$R/profiles/summarize.py" "$OUT" "$R/gpurun_out/$TAG" || exit 1
rm -rf "$OUT/stats"
