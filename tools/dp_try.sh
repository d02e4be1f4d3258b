python tools/bench_gemm.py 2>&1 | grep -v amdgpu | sed -n 2,3p
