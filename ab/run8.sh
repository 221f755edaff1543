for d in 0 1 3 0; do
  CUBERILLE_DBG=$d python bench.py --cpu-sample 0 --steps 10 --warmup 2 --no-project 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('dbg=$d', 'count %.3f'%s['ms_count'])"
done
