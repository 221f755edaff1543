#!/bin/bash
# Why the dense-field walk follows the row width (DESIGN: 47 / 53 / 70 / 75 / 81 ps per vertex at 512 ... 8192 voxels per row):
# the memory-side counters of k_project<unsigned char, 0> on two volumes of the same field, the same slice size (4 MiB) and the
# same vertex count, 512 x 8192 x 256 against 2048 x 2048 x 256 (1.07 GB each: rocprofv3's counter passes survive that size).
#   bash profiles/walk_counters.sh <tag>   ->  gpurun_out/<tag>_walk_row_width_counters.txt
# One rocprofv3 --pmc pass per counter group and shape (counters never share a run with --stats); only counters this
# rocprofv3 lists for the device are asked for.
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# raw rocprofv3 output stays in /tmp (tens of MiB per pass; gpurun_out/ carries 64 MiB at most): the summary is what is kept
OUT=/tmp/walk_$TAG
LOGS=$R/gpurun_out/walk_$TAG
mkdir -p "$OUT" "$LOGS"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 -L > "$LOGS/counters_available.txt" 2>&1 || echo "rocprofv3 -L failed or timed out"
cp "$LOGS/counters_available.txt" "$OUT/counters_available.txt"
echo "counters listed: $(wc -l < "$LOGS/counters_available.txt") lines"
have() { grep -qw "$1" "$OUT/counters_available.txt"; }
GROUPS_=(
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
  "TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_READ_sum"
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum"
  "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum"
  "GRBM_GUI_ACTIVE GRBM_COUNT"
)
FAIL=0
i=0
for G in "${GROUPS_[@]}"; do
  ASK=""
  for c in $G; do if have "$c"; then ASK="$ASK $c"; else echo "not listed on this device: $c" >> "$OUT/skipped.txt"; fi; done
  i=$((i+1))
  [ -z "$ASK" ] && continue
  for SHAPE in "512 8192 256" "2048 2048 256"; do
    D="$OUT/g${i}_$(echo $SHAPE | tr ' ' 'x')"
    timeout -k 10 240 rocprofv3 --kernel-trace --kernel-include-regex k_project --pmc $ASK --output-format csv -d "$D" -o p -- python3 "$R/profiles/walk_row_width.py" $SHAPE 1 > "$D.log" 2> "$D.err" \
      && echo "group $i ($ASK) on $SHAPE done" || { echo "group $i ($ASK) on $SHAPE FAILED or timed out"; tail -5 "$D.err"; cp "$D.err" "$LOGS/"; FAIL=1; }
  done
done
python3 - "$OUT" "$R/gpurun_out/${TAG}_walk_row_width_counters.txt" <<'PY'
import collections, csv, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "g*", "**", "*counter_collection.csv"), recursive=True):
    shape = f[len(src):].split(os.sep)[1].split("_", 1)[1]
    for r in csv.DictReader(open(f)):
        name = (r.get("Kernel_Name") or "").split("(")[0]
        if "k_project" in name:
            acc[(r["Counter_Name"], shape)].append(float(r["Counter_Value"]))
with open(dst, "w") as o:
    o.write("# k_project<unsigned char, 0>, uint8 gradient noise, iso 128: rocprofv3 --pmc, one pass per group and shape, mean over the launches of a run (3)\n")
    o.write("# %-38s %18s %18s %8s\n" % ("counter", "512x8192x256", "2048x2048x256", "ratio"))
    names = sorted({k[0] for k in acc})
    for c in names:
        a, b = acc.get((c, "512x8192x256"), []), acc.get((c, "2048x2048x256"), [])
        ma, mb = (sum(a) / len(a) if a else float("nan")), (sum(b) / len(b) if b else float("nan"))
        o.write("%-40s %18.6g %18.6g %8.3f\n" % (c, ma, mb, mb / ma if ma else float("nan")))
    for f in sorted(glob.glob(os.path.join(src, "g1_*.log"))):
        o.write("# " + open(f).read().strip().replace("\n", "\n# ") + "\n")
print(open(dst).read())
PY
exit $FAIL
