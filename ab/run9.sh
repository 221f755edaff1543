run() { env "$@" python bench.py --cpu-sample 0 --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); s=j['stages_ms']; print('$*', 'project %.3f'%s['ms_project'])"; }
run A=0
run CUBERILLE_PROJ_REFILL=8
run CUBERILLE_PROJ_REFILL=12
run CUBERILLE_PROJ_REFILL=24
run CUBERILLE_PROJ_REFILL=32
run CUBERILLE_PROJ_CHUNK=64
run CUBERILLE_PROJ_CHUNK=256
run CUBERILLE_PROJ_WAVES=8192
run CUBERILLE_PROJ_WAVES=32768
run CUBERILLE_PROJ_WAVES=12288
run CUBERILLE_PROJ_WAVES=24576
run A=0
