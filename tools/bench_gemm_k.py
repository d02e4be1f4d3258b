#!/usr/bin/env python3
"""Fixed cost vs per-k cost of the big GEMM tile: vocabulary-projection shape at several K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmqg_amd  # noqa
from mmqg_amd import ops

def t(al, bl, M, N, K, iters=20):
    A = torch.randn(M * K, device="cuda"); B = torch.randn(N * K, device="cuda"); C = torch.zeros(M, N, device="cuda")
    lda = K if al == 0 else M
    ldb = K if bl == 0 else N
    f = lambda: ops.gemm(al, bl, M, N, K, A, lda, B, ldb, C, N, beta=0)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"al={al} bl={bl} M={M:5d} N={N:5d} K={K:5d}  {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TFLOP/s", flush=True)

print("env", {k: v for k, v in os.environ.items() if k.startswith("MMQG_")})
for K in (128, 256, 512, 1024, 2048):
    t(0, 0, 1280, 10000, K)
for N in (2048, 4096, 8192, 10000, 10240, 16384):
    t(0, 0, 1280, N, 512)
for M in (256, 512, 1024, 1280, 2560):
    t(0, 0, M, 10000, 512)
