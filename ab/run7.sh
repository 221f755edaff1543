for i in 1 2; do for v in 2 3; do
  CUBERILLE_POINTS_VARIANT=$v python bench.py --cpu-sample 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('variant=$v', j['value'], ' '.join('%s %.3f'%(k[3:],v) for k,v in s.items()))"
done; done
for v in 2 3; do
  CUBERILLE_POINTS_VARIANT=$v python bench.py --cpu-sample 0 --workload noise --size 1024 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('variant=$v noise1024', j['value'], ' '.join('%s %.3f'%(k[3:],v) for k,v in s.items()))"
  CUBERILLE_POINTS_VARIANT=$v python bench.py --cpu-sample 0 --workload sphere --size 512 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('variant=$v sphere512', j['value'], ' '.join('%s %.3f'%(k[3:],v) for k,v in s.items()))"
done
