for T in 0 1 2 3; do echo "TILE=$T"; MMQG_GEMM_TILE=$T python tools/bench_gemm.py 2>&1 | grep -v amdgpu | sed -n 2,4p; done
