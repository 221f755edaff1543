for gcap in 2048 1024 768 512 256 2048; do
  CUBERILLE_CLASSIFY_GRID=$gcap python bench.py --cpu-sample 0 --steps 10 --warmup 2 --no-project 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('grid=$gcap', 'classify %.3f'%s['ms_classify'])"
done
