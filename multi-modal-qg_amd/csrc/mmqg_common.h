// Internal helpers shared by the gfx950 kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MMQG_WAVE 64

namespace mmqg {

// thread-local last-error text, surfaced through mmqg_last_error()
void set_error(const char* fmt, ...);
const char* get_error();

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

#define MMQG_REQUIRE(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            mmqg::set_error(__VA_ARGS__);      \
            return -1;                         \
        }                                      \
    } while (0)

#define MMQG_TRY(expr)            \
    do {                          \
        int _rc = (expr);         \
        if (_rc != 0) return _rc; \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mmqg

// ---- device-side helpers -------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Counter-based RNG for dropout: Philox-4x32-10.  One call yields four 32-bit words for a
// (seed, offset) pair, so a mask element is a pure function of its coordinates and can be
// regenerated in the backward pass instead of being stored.
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                                       uint32_t c2, uint32_t c3, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Dropout keep-scale for element `idx` of stream `stream_id`: 0 or 1/(1-p).
__host__ __device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t stream_id, uint64_t idx, float p) {
    uint32_t r[4];
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(idx >> 2), (uint32_t)(idx >> 34),
                  (uint32_t)stream_id, (uint32_t)(stream_id >> 32), r);
    const uint32_t sel = (uint32_t)idx & 3u;   // select chain: a runtime-indexed array would live in scratch
    const uint32_t w = sel == 0 ? r[0] : sel == 1 ? r[1] : sel == 2 ? r[2] : r[3];
    const float u = (float)(w >> 8) * (1.0f / 16777216.0f);   // [0,1)
    return u < p ? 0.0f : 1.0f / (1.0f - p);
}
