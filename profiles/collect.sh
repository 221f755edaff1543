#!/bin/bash
# Regenerates the profile artefacts of this directory on a GPU box:
#   bash profiles/collect.sh <tag> [bench args...]      e.g.  bash profiles/collect.sh r1
# Three separate rocprofv3 runs of the same bench command (kernel trace + stats; FETCH_SIZE; WRITE_SIZE --
# counters never share a run with the stats pass), then profiles/summarize.py boils them down to
#   gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_hbm.csv, <tag>_bench_under_rocprof.json
# which are copied into profiles/ by hand once looked at.
set -e
set +e   # (a counter pass that dies -- rocprofv3 has done so on large volumes -- must not cost the others)
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$R/bench.py" --steps 5 --warmup 2 --cpu-sample 0 --no-slab-probe "$@" > "$OUT/bench.json" 2> "$OUT/stats.err"
echo "stats pass done"
rocprofv3 --kernel-trace --kernel-include-regex cuberille --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$R/bench.py" --steps 1 --warmup 2 --cpu-sample 0 --no-slab-probe "$@" > /dev/null 2> "$OUT/fetch.err"
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --kernel-include-regex cuberille --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 "$R/bench.py" --steps 1 --warmup 2 --cpu-sample 0 --no-slab-probe "$@" > /dev/null 2> "$OUT/write.err"
echo "WRITE_SIZE pass done"
python3 "$R/profiles/summarize.py" "$OUT" "$R/gpurun_out/$TAG"
# the raw traces are tens of MiB per pass (gpurun brings back 64 MiB at most): the summaries are what is kept
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
