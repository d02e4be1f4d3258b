timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -k "attention" 2>&1 | tail -1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-iters 5 > gpurun_out/st.log 2>&1
f=$(find gpurun_out/st -name '*kernel_stats.csv' | head -1); grep "attn_dweights\|Name" $f | cut -c1-140; rm -rf gpurun_out/st
for i in 1 2; do python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms', d['value'], 'q/s')"; done
