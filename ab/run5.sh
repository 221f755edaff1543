for i in 1 2; do
for lib in ab/lib_old.so midas-journal-740_amd/csrc/libcuberille_hip.so; do
  CUBERILLE_LIB=$PWD/$lib python bench.py --cpu-sample 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('$lib', j['value'], j['ms_per_step'], ' '.join('%s %.3f'%(k[3:],v) for k,v in s.items()))"
done; done
