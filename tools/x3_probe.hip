// Where the split-bf16 GEMM's time goes: the kernel file compiled with MMQG_X3_ABLATE (bit 0: no MFMA phase, bit 1: no
// split/LDS-write, bit 2: no global loads after the prologue) and timed on 4096^3 and the vocabulary projection.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Imulti-modal-qg_amd/csrc [-DMMQG_X3_ABLATE=n] tools/x3_probe.hip -o tools/x3_probe
#include <stdarg.h>
#include <stdio.h>

#include "../multi-modal-qg_amd/csrc/gemm_x3.hip"

namespace mmqg {
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}
}  // namespace mmqg

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static void run(const char* name, int al, int bl, int M, int N, int K, int iters) {
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMemset(A, 0x3c, (size_t)M * K * 4)); CK(hipMemset(B, 0x3c, (size_t)N * K * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mmqg::GemmProblem q{M, N, K, A, al == 0 ? K : M, B, bl == 0 ? K : N, C, N, 0};
    for (int i = 0; i < 2; ++i) if (mmqg::gemm_x3_grouped(al, bl, &q, nullptr, nullptr, 1, s)) exit(3);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) mmqg::gemm_x3_grouped(al, bl, &q, nullptr, nullptr, 1, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-22s M=%5d N=%5d K=%5d  %8.1f us  %6.1f TFLOP/s (fp32-equivalent)\n", name, M, N, K, ms * 1e3 / iters,
           2.0 * M * N * K / (ms * 1e-3 / iters) / 1e12);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
}

int main() {
#ifdef MMQG_X3_ABLATE
    printf("ablation mask %d (1: no MFMA phase, 2: no split/LDS write, 4: no global loads in the loop)\n", MMQG_X3_ABLATE);
#else
    printf("full kernel\n");
#endif
    run("square 4096 (NT)", 0, 0, 4096, 4096, 4096, 5);
    run("square 4096 (TN)", 1, 1, 4096, 4096, 4096, 5);
    run("vocab fwd (NT)", 0, 0, 1280, 10000, 512, 20);
    run("vocab dgrad (NN)", 0, 1, 1280, 512, 10000, 20);
    run("vocab wgrad (TN)", 1, 1, 10000, 512, 1280, 20);
    run("config5 vocab fwd (NT)", 0, 0, 2560, 50000, 1024, 3);
    run("config5 vocab dgrad (NN)", 0, 1, 2560, 1024, 50000, 3);
    run("config5 vocab wgrad (TN)", 1, 1, 50000, 1024, 2560, 3);
    return 0;
}
