for x in 0 64 256 1024 4096 0; do
  CUBERILLE_PROJ_XCD=$x python bench.py --cpu-sample 0 --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('xcd=$x', j['value'], j['config']['points'], 'project %.3f'%s['ms_project'])"
done
