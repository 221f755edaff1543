for wl in "noise 1000" "marschner_lobb 1000" "sphere 700"; do set -- $wl
for v in 1 0; do
  if [ $v = 1 ]; then export CUBERILLE_NO_STREAM_CLASSIFY=1; else unset CUBERILLE_NO_STREAM_CLASSIFY; fi
  python bench.py --cpu-sample 0 --workload $1 --size $2 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('$1 $2 old_rows=$v', j['value'], j['config']['points'], ' '.join('%s %.3f'%(k[3:],v) for k,v in s.items()))"
done; done
