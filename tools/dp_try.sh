export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
MMQG_FORCE_DP=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/dp.out 2> gpurun_out/dp.err; echo "rc=$?"; tail -c 600 gpurun_out/dp.out; tail -20 gpurun_out/dp.err
