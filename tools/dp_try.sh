timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -k "attention" 2>&1 | tail -2
timeout -k 10 600 python -m pytest tests/test_hip_model.py -x -q 2>&1 | tail -2
for i in 1 2; do python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms', d['value'], 'q/s')"; done
