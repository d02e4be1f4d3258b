timeout -k 10 300 python -m pytest tests/test_hip_model.py tests/test_hip_kernels.py -x -q -k "zero_padded or attn or attention" 2>&1 | tail -3
for F in "" "--skip-zero-rows"; do python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$F', d['ms_per_step'],'ms', d['value'], 'q/s | attn', r['us_per_launch'],'us', r['frac'])"; done
