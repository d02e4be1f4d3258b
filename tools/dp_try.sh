for W in 0 1; do for i in 1 2; do MMQG_SIDE_FIRST=$W python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('side_first $W:', d['ms_per_step'],'ms', d['value'], 'q/s')"; done; done
