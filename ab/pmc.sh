R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc2/$tag -o p -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 > /dev/null 2> $R/gpurun_out/pmc2/$tag.err
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$R/gpurun_out/pmc2/$tag/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_project' in k or 'k_emit' in k or 'k_count' in k or 'k_classify' in k:
            acc[(k[:40],r['Counter_Name'])].append(float(r['Counter_Value']))
for (k,c),v in sorted(acc.items()): print(k,c,'%.4g'%(sum(v)/len(v)))
PY
done
