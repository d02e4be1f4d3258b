#!/bin/bash
# Round profile set (run on the GPU box from the repo root):  bash tools/profile_round.sh r04 [workloads...]
# For every workload: the bench line, rocprofv3 kernel-trace stats of (a) the step launches alone (--kernel-iters 0:
# the CSV the in-step attention figure is recomputed from) and (b) the default command, the two PMC passes of the
# attention kernel (FETCH_SIZE / WRITE_SIZE, collected separately), and the per-phase times.
set -o pipefail
R=${1:-r03}
shift
WL=${@:-config2 config4 config5}
OUT=gpurun_out/$R
mkdir -p $OUT profiles
export TMPDIR=/tmp
for W in $WL; do
  echo "== $W: step-only kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_$W -- python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --kernel-iters 0 > $OUT/step_$W.log 2>&1 || exit 1
  f=$(find $OUT/step_$W -name '*kernel_trace.csv' | head -1)
  python3 tools/attn_in_step.py "$f" $W profiles/attn_in_step.json 60 || exit 1      # skip the warm-up pass's launches
  python3 tools/step_kernels.py "$f" > $OUT/step_timeline_$W.txt 2>/dev/null
  f=$(find $OUT/step_$W -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/kernel_stats_${W}_step_only.csv
  rm -rf $OUT/step_$W
  echo "== $W: PMC passes"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$W -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --kernel-iters 20 > $OUT/pmc_fetch_$W.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$W -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --kernel-iters 20 > $OUT/pmc_write_$W.log 2>&1 || exit 1
  python3 tools/pmc_traffic.py $OUT/pmc_fetch_$W $OUT/pmc_write_$W attn_softmax_context_fwd_kernel $W profiles/attn_traffic.json || exit 1
  # (where the decoder's forward loop is one persistent launch: that launch's traffic too; absent at config 5)
  python3 tools/pmc_traffic.py $OUT/pmc_fetch_$W $OUT/pmc_write_$W decoder_persist_fwd_kernel $W profiles/attn_traffic.json persist_dec.hip $W:decoder_persist_fwd || true
  for k in fetch write; do
    f=$(find $OUT/pmc_${k}_$W -name '*counter_collection.csv' | head -1)
    (head -1 "$f"; grep attn_softmax_context_fwd "$f" | head -60; grep decoder_persist_fwd "$f" | head -10) > $OUT/pmc_${k}_attn_$W.csv
  done
  rm -rf $OUT/pmc_fetch_$W $OUT/pmc_write_$W
  echo "== $W: bench line and the kernel stats of the same command"
  python3 bench.py --workload $W --steps 20 --warmup 5 > $OUT/bench_$W.json 2> $OUT/bench_$W.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats_$W.log 2>&1 || exit 1
  f=$(find $OUT/stats_$W -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/kernel_stats_$W.csv
  rm -rf $OUT/stats_$W
  python3 tools/phase_times.py $W > $OUT/phase_times_$W.txt 2>/dev/null
  python3 tools/persist_dec_trace.py $W > $OUT/persist_dec_trace_$W.txt 2>/dev/null || rm -f $OUT/persist_dec_trace_$W.txt
  python3 tools/persist_dec_bwd_trace.py $W > $OUT/persist_dec_bwd_trace_$W.txt 2>/dev/null || rm -f $OUT/persist_dec_bwd_trace_$W.txt
  python3 tools/host_batch_rate.py $W 50 2>/dev/null | grep "^$W " >> $OUT/host_batch_rate.txt
done
# one rank through the data-parallel schedule over RCCL (the self-launcher), with and without the CUs reserved for RCCL, and
# the plain step on the same box
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --kernel-iters 0 > $OUT/bench_config2_same_box.json 2>/dev/null
MMQG_FORCE_DP=1 MMQG_BENCH_FORCE_LAUNCH=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --kernel-iters 0 > $OUT/bench_config2_force_dp.json 2>/dev/null
MMQG_DP_RESERVE_CUS=0 MMQG_FORCE_DP=1 MMQG_BENCH_FORCE_LAUNCH=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --kernel-iters 0 > $OUT/bench_config2_force_dp_noreserve.json 2>/dev/null
MMQG_X3_MAX_SPLIT=1 python3 tools/x3_fixed_cost.py > $OUT/x3_fixed_cost.txt 2>/dev/null
MMQG_X3_MAX_SPLIT=1 MMQG_X3_DWORD_EPILOGUE=1 python3 tools/x3_fixed_cost.py >> $OUT/x3_fixed_cost.txt 2>/dev/null
MMQG_X3_MAX_SPLIT=1 MMQG_X3_BAL=1 python3 tools/x3_fixed_cost.py >> $OUT/x3_fixed_cost.txt 2>/dev/null
cp profiles/attn_in_step.json profiles/attn_traffic.json $OUT/
echo profile set done
