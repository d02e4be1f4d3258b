timeout -k 10 300 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -x -q -k "attn or attention or zero_padded or dropin or golden or ragged" 2>&1 | tail -2
for W in config2 config4 config5; do python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$W', d['ms_per_step'],'ms | attn', r['us_per_launch'],'us', r['frac'], '| cold', r['us_per_launch_beyond_mall'], r['frac_beyond_mall'])"; done
