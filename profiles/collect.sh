#!/bin/bash
# Regenerates the profile artefacts of this directory on a GPU box:
#   bash profiles/collect.sh <tag> [bench args...]      e.g.  bash profiles/collect.sh r4
# Three separate rocprofv3 runs of the same bench command (kernel trace + stats; FETCH_SIZE; WRITE_SIZE --
# counters never share a run with the stats pass), then profiles/summarize.py boils them down to
#   gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_hbm.csv, <tag>_bench_under_rocprof.json
# which are copied into profiles/ by hand once looked at.
# A pass that dies (rocprofv3's counter passes have done so on large volumes) does not cost the others, but it is
# said out loud, its raw directory and log are kept, and the script ends non-zero: a summary made of the remaining
# passes is partial and must not be taken for a complete one.
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FAIL=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$R/bench.py" --steps 5 --warmup 2 --cpu-sample 0 --no-slab-probe --no-warm-up "$@" > "$OUT/bench.json" 2> "$OUT/stats.err" \
  && echo "stats pass done" || { echo "stats pass FAILED (see $OUT/stats.err)"; FAIL=1; }
rocprofv3 --kernel-trace --kernel-include-regex cuberille --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$R/bench.py" --steps 1 --warmup 2 --cpu-sample 0 --no-slab-probe --no-warm-up "$@" > /dev/null 2> "$OUT/fetch.err" \
  && echo "FETCH_SIZE pass done" || { echo "FETCH_SIZE pass FAILED (see $OUT/fetch.err)"; FAIL=1; }
rocprofv3 --kernel-trace --kernel-include-regex cuberille --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 "$R/bench.py" --steps 1 --warmup 2 --cpu-sample 0 --no-slab-probe --no-warm-up "$@" > /dev/null 2> "$OUT/write.err" \
  && echo "WRITE_SIZE pass done" || { echo "WRITE_SIZE pass FAILED (see $OUT/write.err)"; FAIL=1; }
python3 "$R/profiles/summarize.py" "$OUT" "$R/gpurun_out/$TAG" || { echo "summarize.py FAILED"; FAIL=1; }
# the raw traces are tens of MiB per pass (gpurun brings back 64 MiB at most): the summaries are what is kept -- of a
# run whose passes all came through
if [ "$FAIL" = 0 ]; then
  rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
else
  echo "collect.sh: at least one pass failed -- raw output kept under $OUT, summaries (if any) are PARTIAL"
fi
exit $FAIL
