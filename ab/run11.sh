run() { env "$@" python bench.py --cpu-sample 0 --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('$*', j['value'], 'classify %.3f count %.3f scan %.3f pass %.3f total %.3f pts %d'%(s['ms_classify'],s['ms_count'],s['ms_scan'],s['ms_classify']+s['ms_count']+s['ms_scan'],s['ms_total'], j['config']['points']))"; }
run A=0
run CUBERILLE_OVERLAP=4
run CUBERILLE_OVERLAP=8
run CUBERILLE_OVERLAP=16
run CUBERILLE_OVERLAP=8 CUBERILLE_OVERLAP_GRID=1536
run CUBERILLE_OVERLAP=8 CUBERILLE_OVERLAP_GRID=768
run CUBERILLE_OVERLAP=8 CUBERILLE_OVERLAP_GRID=2048
run CUBERILLE_OVERLAP=32
run A=0
