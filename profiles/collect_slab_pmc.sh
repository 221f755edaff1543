#!/bin/bash
# HBM counters of configs[4]'s field on a slab rocprofv3's counter passes survive (its --pmc passes die on the 8.6 GB
# volume itself: profiles/README.md): 2048 x 2048 x 256 of the same uint8 noise, FETCH_SIZE and WRITE_SIZE in separate passes,
# every kernel of the extraction.  bench.py scales the pass kernels' figures by the slice ratio (roofline.traffic of
# --workload noise --size 2048, with a note saying so).
#   bash profiles/collect_slab_pmc.sh <tag>   ->  gpurun_out/<tag>_pmc_hbm.csv
TAG=${1:-r5_config5_slab2048x2048x256}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=/tmp/slabpmc_$TAG
mkdir -p "$OUT" "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
FAIL=0
for leg in fetch write; do
  C=FETCH_SIZE; [ $leg = write ] && C=WRITE_SIZE
  timeout -k 10 300 rocprofv3 --kernel-trace --kernel-include-regex cuberille --pmc $C --output-format csv -d "$OUT/$leg" -o $leg -- python3 "$R/profiles/walk_row_width.py" 2048 2048 256 1 > "$OUT/$leg.log" 2> "$OUT/$leg.err" \
    && echo "$C pass done" || { echo "$C pass FAILED or timed out"; tail -5 "$OUT/$leg.err"; FAIL=1; }
done
python3 "$R/profiles/summarize.py" "$OUT" "$R/gpurun_out/$TAG" || FAIL=1
grep -h "noise u8" "$OUT"/*.log | head -2
exit $FAIL
