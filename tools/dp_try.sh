for K in "" 100000 32; do MMQG_SKINNY_KS8_FROM=$K python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ks8_from [$K]', d['ms_per_step'],'ms', d['value'], 'q/s')"; done
