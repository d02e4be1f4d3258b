#!/bin/bash
# Round profile set (run on the GPU box from the repo root):  bash tools/profile_round.sh r02
# kernel-trace stats + bench lines for configs 2, 4, 5, and the two PMC passes of the attention kernel.
set -o pipefail
R=${1:-r02}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
for W in config2 config4 config5; do
  python3 bench.py --workload $W --steps 20 --warmup 5 > $OUT/bench_$W.json 2> $OUT/bench_$W.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats_$W.log 2>&1 || exit 1
done
for W in config2 config4 config5; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$W -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --kernel-iters 20 > $OUT/pmc_fetch_$W.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$W -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --kernel-iters 20 > $OUT/pmc_write_$W.log 2>&1 || exit 1
  python3 tools/pmc_traffic.py $OUT/pmc_fetch_$W $OUT/pmc_write_$W attn_softmax_context_fwd_kernel $W $OUT/attn_traffic.json || exit 1
done
# keep only the summaries (the raw traces are large)
for W in config2 config4 config5; do
  f=$(find $OUT/stats_$W -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/kernel_stats_$W.csv
  for k in fetch write; do
    f=$(find $OUT/pmc_${k}_$W -name '*counter_collection.csv' | head -1)
    (head -1 "$f"; grep attn_softmax_context_fwd "$f" | head -60) > $OUT/pmc_${k}_attn_$W.csv
  done
  rm -rf $OUT/stats_$W $OUT/pmc_fetch_$W $OUT/pmc_write_$W
done
echo profile set done
