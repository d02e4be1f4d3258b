// Host-side runtime shared by the persistent kernels (persist.hip: forward time loop, persist_bwd.hip: backward
// time loop): who may launch one, and how a failed one reaches the caller.
//
// A persistent kernel is only correct while ALL its workgroups are resident (they meet at device-wide barriers).
// Two such launches in flight on one device can each end up half resident and wait for workgroups the other keeps
// off the CUs.  The rules enforced here, per device:
//   * eager launches: a request on stream s is granted when the previous persistent launch was issued on s too
//     (stream order) or has completed (hipEventQuery); otherwise the caller takes its launch-per-diagonal path;
//   * during stream capture: the nodes of two streams of one capture are parallel branches of the graph, so only
//     ONE stream of a capture may hold persistent launches; a request from another stream of the same capture is
//     declined.
// What the host cannot see (two captured graphs replayed side by side, another PROCESS on the same device) is
// covered on the device: every barrier spin is bounded, a timed-out launch poisons its output with NaN, bumps the
// sticky counter of its workspace and writes a word in pinned host memory, which persist_failures() reads without
// synchronising; the sequence executors refuse to run once it is set (mmqg_persist_clear_failures() re-arms them).
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "mmqg_common.h"
#include "mmqg_kernels.h"

namespace mmqg {

namespace {

constexpr int kMaxDev = 16;

struct DevState {
    hipEvent_t ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    unsigned long long cap_id = 0;
    hipStream_t cap_stream = nullptr;
    bool cap_valid = false;
    int cus = 0;
};

std::mutex g_mu;
DevState g_dev[kMaxDev];
unsigned* g_fail_host = nullptr;        // pinned host word (host address)
unsigned* g_fail_dev = nullptr;         // the same word as the device sees it
bool g_fail_tried = false;
int g_declined = 0;
std::atomic<int> g_extra_wg{0};
std::atomic<unsigned> g_max_spins{0};
std::atomic<int> g_reserved{0};         // CUs a persistent grid that can shrink leaves to collectives (data parallel)

int cur_dev() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return d < 0 || d >= kMaxDev ? 0 : d;
}

bool capturing(hipStream_t s, unsigned long long* id) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    if (hipStreamGetCaptureInfo(s, &st, &cid) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (id) *id = cid;
    return st == hipStreamCaptureStatusActive;
}

}  // namespace

// one-time setup that must not happen inside a stream capture: the pinned failure word and the device's event.
// Called from the *_ws_bytes queries (descriptor-build time) and again, harmlessly, before every launch.
void persist_runtime_prepare() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_fail_tried) {
        g_fail_tried = true;
        void* h = nullptr;
        // portable: one word every device of the process can address (the kernels of any device report through it)
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && h) {
            *reinterpret_cast<volatile unsigned*>(h) = 0u;
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, h, 0) == hipSuccess && dp) {
                g_fail_host = reinterpret_cast<unsigned*>(h);
                g_fail_dev = reinterpret_cast<unsigned*>(dp);
            } else {
                (void)hipGetLastError();
                (void)hipHostFree(h);
            }
        } else {
            (void)hipGetLastError();
        }
    }
    DevState& d = g_dev[cur_dev()];
    if (!d.ev && hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); d.ev = nullptr; }
    if (!d.cus) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) d.cus = p.multiProcessorCount;
        else (void)hipGetLastError();
    }
}

int persist_device_cus() {
    persist_runtime_prepare();
    return g_dev[cur_dev()].cus;
}

// CUs a persistent grid launched on s can count on: the device's, or fewer when the stream (hipExtStreamCreateWithCUMask)
// or the process (ROC_GLOBAL_CU_MASK / HSA_CU_MASK) is confined to a CU mask — a grid sized from multiProcessorCount
// would then never become fully resident and stall to its spin bound.  `can_shrink`: the kernel's plan works for any
// grid size, so it also leaves out the CUs reserved for collectives (persist_set_reserved_cus).  Every launch still
// checks its workgroup against hipOccupancyMaxActiveBlocksPerMultiprocessor; a kernel that needs more CUs than this
// returns "not taken" and its caller runs the launch-per-step path.
int persist_usable_cus(hipStream_t s, bool can_shrink) {
    int cus = persist_device_cus();
    uint32_t mask[32] = {};
    const int words = std::min(32, (cus + 31) / 32);
    if (words > 0 && hipExtStreamGetCUMask(s, (uint32_t)words, mask) == hipSuccess) {
        int n = 0;
        for (int i = 0; i < words; ++i) n += __builtin_popcount(mask[i]);
        if (n > 0 && n < cus) cus = n;
    } else {
        (void)hipGetLastError();
    }
    if (can_shrink) cus -= std::min(std::max(g_reserved.load(), 0), cus / 2);
    return cus;
}

void persist_set_reserved_cus(int n) { g_reserved.store(n < 0 ? 0 : n); }
int persist_reserved_cus() { return g_reserved.load(); }

unsigned* persist_host_fail_word() { return g_fail_dev; }

// 0 = a persistent kernel may be launched on s now; 1 = declined (see the rules on top of this file)
int persist_begin(hipStream_t s) {
    persist_runtime_prepare();
    std::lock_guard<std::mutex> lk(g_mu);
    DevState& d = g_dev[cur_dev()];
    unsigned long long id = 0;
    if (capturing(s, &id)) {
        if (d.cap_valid && d.cap_id == id && d.cap_stream != s) { g_declined += 1; return 1; }
        return 0;
    }
    if (!d.have_last || d.last_stream == s) return 0;
    if (!d.ev) { g_declined += 1; return 1; }
    const hipError_t e = hipEventQuery(d.ev);
    if (e == hipSuccess) return 0;
    (void)hipGetLastError();            // hipErrorNotReady: the previous persistent launch is still in flight elsewhere
    g_declined += 1;
    return 1;
}

// after a granted launch has been enqueued on s
void persist_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    DevState& d = g_dev[cur_dev()];
    unsigned long long id = 0;
    if (capturing(s, &id)) {
        d.cap_valid = true; d.cap_id = id; d.cap_stream = s;
        return;
    }
    if (d.ev && hipEventRecord(d.ev, s) == hipSuccess) { d.last_stream = s; d.have_last = true; }
    else (void)hipGetLastError();
}

int persist_declined_count() { return g_declined; }

int persist_failures() {
    if (!g_fail_host) return 0;
    return (int)*reinterpret_cast<volatile unsigned*>(g_fail_host);
}

void persist_clear_failures() {
    if (g_fail_host) *reinterpret_cast<volatile unsigned*>(g_fail_host) = 0u;
}

int persist_check_healthy(const char* who) {
    if (persist_failures() > 0) {
        set_error("%s: an earlier persistent time loop timed out at a device-wide barrier (its workgroups were not all "
                  "resident: another persistent launch or another process on this device?); its outputs were poisoned "
                  "with NaN and every result since then is invalid — mmqg_persist_clear_failures() re-arms the library",
                  who);
        return -1;
    }
    return 0;
}

// test hook: the following persistent launches wait for `extra_workgroups` more arrivals than their grid has and
// give up after `max_spins` polls (0 = the default bound)
void persist_set_test_fault(int extra_workgroups, unsigned max_spins) {
    g_extra_wg.store(extra_workgroups);
    g_max_spins.store(max_spins);
}
int persist_test_extra_wg() { return g_extra_wg.load(); }
unsigned persist_test_max_spins() { return g_max_spins.load(); }

}  // namespace mmqg
