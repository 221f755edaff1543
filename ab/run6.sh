for l in 0 20000 40000 80000 0; do
  CUBERILLE_CELLS_LDS=$l python bench.py --cpu-sample 0 --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('lds=$l', 'cells %.3f'%s['ms_emit_cells'])"
done
