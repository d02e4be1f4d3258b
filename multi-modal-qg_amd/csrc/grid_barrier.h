// Device-wide barrier for persistent kernels on MI355X (8 XCDs with private, mutually non-coherent L2s).
//
// XCD-hierarchical and fence-free (MI355X_MICROARCH.md price list, row barrier-xcd; measured here by
// tools/barrier_probe.hip: 2.3 us per barrier at 256 workgroups, 4.0 us with release/acquire fences, against
// 8.7 us for a flat counter with fences): every workgroup adds to its XCC's counter; the XCC's last arriver adds
// to the top counter, waits for all XCCs and then bumps its XCC's generation word, which the others poll.
// All counters are monotonic (barrier k waits for k * n arrivals), live on 128-byte lines of their own and
// are polled with relaxed agent-scope (sc1) loads + s_sleep.
//
// Fence-free means: the barrier orders nothing but its own counters.  Data handed from one workgroup to
// another across it must therefore be STORED write-through (sc1 stores: raw_buffer_store ... aux 16, or relaxed
// agent-scope atomic stores), every storing wave must have drained its stores (s_waitcnt vmcnt(0), done in
// sync() before the workgroup barrier), and every LOAD of such data must be an sc1 load (raw_buffer_load ...
// aux 16) — cdna_hip_programming.md Guideline 16, recipe R1 with sc1 loads in place of the acquire.
//
// Every spin is bounded (Ctx::max_spins polls); on a timeout the fail word is set, every workgroup leaves at its
// next poll and sync() returns false; a leader whose wait failed does NOT release its XCC (its peers leave through
// the fail word), so nobody runs on past a failed barrier.  The XBar block must be zero when the kernel starts,
// and the whole grid must be resident (grid <= number of CUs, one workgroup fitting a CU).  report_failure() makes
// the failure visible to the host: a sticky counter in device memory and a word in pinned host memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmqg {
namespace gb {

typedef unsigned u32;
constexpr int kLine = 32;   // u32 per 128-byte line

struct XBar {
    u32 xcc_count[8 * kLine];
    u32 top[kLine];
    u32 xcc_gen[8 * kLine];
    u32 flat[kLine];
    u32 census[8 * kLine];
    u32 fail[kLine];
};

__device__ __forceinline__ u32 xcc_id() {
    u32 v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ u32 ld_rlx(u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rlx(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 add_rlx(u32* p, u32 v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr u32 kDefaultSpins = 1u << 22;     // polls (each an s_sleep + an L2 round trip: seconds in all)

__device__ __forceinline__ bool spin_until_ge(u32* p, u32 target, u32* fail, u32 max_spins) {
    u32 spins = 0;
    while ((int)(ld_rlx(p) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > max_spins || ((spins & 255u) == 0 && ld_rlx(fail))) { st_rlx(fail, 1u); return false; }
    }
    return true;
}

struct Ctx {
    XBar* b;
    u32 x, n_x, n_groups, k, leader, max_spins;
};

// census of the workgroups per XCC (placement is whatever the dispatcher chose) + one flat barrier
__device__ __forceinline__ bool init(Ctx& c, XBar* b, u32 n_workgroups, u32 max_spins = kDefaultSpins) {
    c.b = b;
    c.x = xcc_id();
    c.k = 0;
    c.max_spins = max_spins;
    bool ok = true;
    if (threadIdx.x == 0) {
        // the census add must be PERFORMED before this workgroup counts in the flat barrier: whoever sees
        // flat == n_workgroups reads the census next.  The returned value is made a register operand of an (empty)
        // asm statement, so hipcc waits for the atomic's return before it issues the flat add.
        const u32 seen = add_rlx(&b->census[c.x * kLine], 1u);
        asm volatile("; census add returned" ::"v"(seen) : "memory");
        add_rlx(&b->flat[0], 1u);
        ok = spin_until_ge(&b->flat[0], n_workgroups, b->fail, c.max_spins);
    }
    ok = __syncthreads_and(ok);
    c.n_x = ld_rlx(&b->census[c.x * kLine]);
    u32 g = 0;
    for (int i = 0; i < 8; ++i) g += ld_rlx(&b->census[i * kLine]) ? 1u : 0u;
    c.n_groups = g;
    return ok;
}

// init() in two halves: a kernel whose prologue is long (tens of KB of weights into LDS) announces itself first and waits
// for the others afterwards, so the flat barrier turns while the prologue's loads fly.
__device__ __forceinline__ void init_arrive(Ctx& c, XBar* b, u32 max_spins = kDefaultSpins) {
    c.b = b;
    c.x = xcc_id();
    c.k = 0;
    c.max_spins = max_spins;
    if (threadIdx.x == 0) {
        const u32 seen = add_rlx(&b->census[c.x * kLine], 1u);      // (performed before the flat add: see init())
        asm volatile("; census add returned" ::"v"(seen) : "memory");
        add_rlx(&b->flat[0], 1u);
    }
}
__device__ __forceinline__ bool init_wait(Ctx& c, u32 n_workgroups) {
    XBar* b = c.b;
    bool ok = true;
    if (threadIdx.x == 0) ok = spin_until_ge(&b->flat[0], n_workgroups, b->fail, c.max_spins);
    ok = __syncthreads_and(ok);
    c.n_x = ld_rlx(&b->census[c.x * kLine]);
    u32 g = 0;
    for (int i = 0; i < 8; ++i) g += ld_rlx(&b->census[i * kLine]) ? 1u : 0u;
    c.n_groups = g;
    return ok;
}

// The barrier in two halves, so that work which nobody waits for (stores to buffers that only later kernels
// read, prefetches) can sit between a workgroup's arrival and its wait.
//   arrive(): every wave drains its (sc1) stores, the workgroup's lane 0 adds to its XCC's counter; the XCC's
//             last arriver also adds to the top counter.
//   wait():   lane 0 polls (the leader: the top counter, then it releases its XCC; the others: their XCC's
//             generation word); false after a timeout anywhere on the chip.
__device__ __forceinline__ void arrive(Ctx& c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its sc1 stores have been acknowledged
    __syncthreads();
    c.k += 1;
    c.leader = 0;
    if (threadIdx.x == 0) {
        XBar* b = c.b;
        const u32 v = add_rlx(&b->xcc_count[c.x * kLine], 1u);
        if (v == c.k * c.n_x - 1) {
            add_rlx(&b->top[0], 1u);
            c.leader = 1;
        }
    }
}

__device__ __forceinline__ bool wait(Ctx& c) {
    bool ok = true;
    if (threadIdx.x == 0) {
        XBar* b = c.b;
        if (c.leader) {
            ok = spin_until_ge(&b->top[0], c.k * c.n_groups, b->fail, c.max_spins);
            if (ok) st_rlx(&b->xcc_gen[c.x * kLine], c.k);     // a failed wait releases nobody: peers leave via the fail word
        } else {
            ok = spin_until_ge(&b->xcc_gen[c.x * kLine], c.k, b->fail, c.max_spins);
        }
    }
    return __syncthreads_and(ok);
}

__device__ __forceinline__ bool sync(Ctx& c) {
    arrive(c);
    return wait(c);
}

// A workgroup that left a barrier with ok == false tells the host (one lane): `sticky` is a device counter that
// nothing resets between launches, `host` (nullable) a word in pinned host memory the library polls without
// synchronising.
__device__ __forceinline__ void report_failure(u32* sticky, u32* host) {
    if (threadIdx.x == 0) {
        add_rlx(sticky, 1u);
        if (host) __hip_atomic_store(host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace gb
}  // namespace mmqg
