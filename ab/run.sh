for i in 1 2; do
for lib in ab/lib_old.so midas-journal-740_amd/csrc/libcuberille_hip.so; do
  CUBERILLE_LIB=$PWD/$lib python bench.py --cpu-sample 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('$lib', j['value'], ' '.join('%s %.3f'%(k[3:],v) for k,v in s.items()))"
done; done
CUBERILLE_LIB=$PWD/ab/lib_old.so python bench.py --cpu-sample 0 --workload noise --size 1000 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('old noise1000', j['value'], ' '.join('%s %.3f'%(k[3:],v) for k,v in s.items()))"
python bench.py --cpu-sample 0 --workload noise --size 1000 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('new noise1000', j['value'], ' '.join('%s %.3f'%(k[3:],v) for k,v in s.items()))"
