// Stage timeline of the fused layer-step kernel (csrc/skinny.hip) at config-2 shapes: the kernel is
// compiled here with MMQG_SKINNY_TRACE, launched as a dependent chain inside a hipGraph (like the
// trainer's time loop), and the last launch's per-workgroup stamps are summarised.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Imulti-modal-qg_amd/csrc tools/skinny_probe.hip -o tools/skinny_probe
#define MMQG_SKINNY_TRACE 1
#include <stdarg.h>
#include <stdio.h>

#include "../multi-modal-qg_amd/csrc/skinny.hip"

namespace mmqg {
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}
}  // namespace mmqg

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float* dalloc(size_t n, float val) {
    float* p; CK(hipMalloc(&p, n * 4));
    std::vector<float> h(n, val);
    for (size_t i = 0; i < n; ++i) h[i] = val * (float)((i * 2654435761u >> 8) % 1000) / 1000.f;
    CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
    return p;
}

static void report(const char* name, const std::vector<unsigned long long>& tr, int nwg, float us_per_launch) {
    unsigned long long t0min = ~0ull;
    for (int w = 0; w < nwg; ++w) t0min = std::min(t0min, tr[(size_t)w * 8]);
    double mean[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
    int have4 = 0;
    for (int w = 0; w < nwg; ++w)
        for (int s = 0; s < 5; ++s) {
            const unsigned long long v = tr[(size_t)w * 8 + s];
            if (!v) continue;
            const double us = (double)(v - t0min) * 0.01;    // wall_clock64: 100 MHz
            mean[s] += us / nwg; mx[s] = std::max(mx[s], us);
            if (s == 4) ++have4;
        }
    printf("%-28s %6.2f us/launch | stage mean (max) us since first workgroup start: entry %.2f (%.2f)  prefetch issued %.2f (%.2f)  "
           "operands+MFMA done %.2f (%.2f)  partials exchanged %.2f (%.2f)  epilogue stores issued %.2f (%.2f)\n",
           name, us_per_launch, mean[0], mx[0], mean[1], mx[1], mean[2], mx[2], mean[3], mx[3], mean[4] * nwg / std::max(have4, 1), mx[4]);
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int B = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 512, chain = 60;
    constexpr int L = 3;
    printf("B = %d, H = %d\n", B, H);
    float* W_ih[L]; float* W_hh[L]; float* WT_ih[L]; float* WT_hh[L]; float* b1[L]; float* b2[L];
    for (int l = 0; l < L; ++l) {
        W_ih[l] = dalloc((size_t)4 * H * H, 0.02f); W_hh[l] = dalloc((size_t)4 * H * H, 0.02f);
        WT_ih[l] = dalloc((size_t)4 * H * H, 0.02f); WT_hh[l] = dalloc((size_t)4 * H * H, 0.02f);
        b1[l] = dalloc(4 * H, 0.1f); b2[l] = dalloc(4 * H, 0.1f);
    }
    float* x = dalloc((size_t)B * H, 1.f);
    float* hs = dalloc((size_t)L * (chain + 2) * B * H, 0.5f);
    float* cs = dalloc((size_t)L * (chain + 2) * B * H, 0.5f);
    float* gates = dalloc((size_t)L * B * 4 * H, 0.5f);
    float* dgates = dalloc((size_t)L * 2 * B * 4 * H, 0.01f);
    float* dh = dalloc((size_t)L * B * H, 0.01f);
    float* dc = dalloc((size_t)L * B * H, 0.01f);
    const size_t BH = (size_t)B * H;
    unsigned long long* trace; const int max_wg = 8192;
    CK(hipMalloc(&trace, (size_t)max_wg * 8 * 8));
    unsigned long long* null_trace = nullptr;

    auto fwd_job = [&](int l, int t) {
        mmqg::SkinnyFwdJob j{};
        j.pairs[0] = mmqg::SkinnyPair{l == 0 ? x : hs + (size_t)(l - 1) * (chain + 2) * BH + (size_t)(t + 1) * BH, H, W_ih[l], H, H, 0};
        j.pairs[1] = mmqg::SkinnyPair{hs + (size_t)l * (chain + 2) * BH + (size_t)t * BH, H, W_hh[l], H, H, 0};
        j.npairs = 2; j.gates_has_pre = 0; j.bias1 = b1[l]; j.bias2 = b2[l];
        mmqg::CellFwd& c = j.cell;
        c.B = B; c.H = H; c.gates = gates + (size_t)l * B * 4 * H; c.ld_g = 4 * H;
        c.h_prev = hs + (size_t)l * (chain + 2) * BH + (size_t)t * BH; c.c_prev = cs + (size_t)l * (chain + 2) * BH + (size_t)t * BH;
        c.h_out = hs + (size_t)l * (chain + 2) * BH + (size_t)(t + 1) * BH; c.c_out = cs + (size_t)l * (chain + 2) * BH + (size_t)(t + 1) * BH;
        c.t = t;
        return j;
    };
    auto bwd_job = [&](int l, int t) {
        mmqg::SkinnyBwdJob j{};
        const size_t G = (size_t)B * 4 * H;
        j.pairs[0] = mmqg::SkinnyPair{dgates + (size_t)l * 2 * G + (size_t)((t + 1) & 1) * G, 4 * H, WT_hh[l], 4 * H, 4 * H, 0};
        j.npairs = 1;
        if (l < L - 1) { j.pairs[1] = mmqg::SkinnyPair{dgates + (size_t)(l + 1) * 2 * G + (size_t)(t & 1) * G, 4 * H, WT_ih[l + 1], 4 * H, 4 * H, 1}; j.npairs = 2; }
        mmqg::CellBwd& c = j.cell;
        c.B = B; c.H = H; c.gates_act = gates + (size_t)l * G; c.c_prev = cs + (size_t)l * (chain + 2) * BH; c.c_new = cs + (size_t)l * (chain + 2) * BH + BH;
        c.dh_rec = dh + (size_t)l * BH; c.dc = dc + (size_t)l * BH; c.dgates = dgates + (size_t)l * 2 * G + (size_t)(t & 1) * G; c.ld_dg = 4 * H;
        c.t = t;
        return j;
    };

    struct Case { const char* name; int njobs; bool bwd; int nwg; int layer; };
    const int ft = (H / 4) * ((B + 15) / 16), bt = (H / 16) * ((B + 15) / 16);
    const Case cases[] = {{"FWD_CELL 1 job (K=2xH)", 1, false, ft, 1}, {"FWD_CELL 3 jobs", 3, false, ft * 3, 0},
                          {"BWD_CELL 1 job (K=2x4H)", 1, true, bt, 1}, {"BWD_CELL 1 job (K=1x4H)", 1, true, bt, 2},
                          {"BWD_CELL 3 jobs", 3, true, bt * 3, 0}};
    for (const Case& cs_ : cases) {
        auto enqueue = [&](int t) {
            if (!cs_.bwd) {
                mmqg::SkinnyFwdJob jobs[3];
                for (int i = 0; i < cs_.njobs; ++i) jobs[i] = fwd_job(cs_.njobs == 1 ? cs_.layer : i, t);
                if (mmqg::skinny_cell_fwd_multi(jobs, cs_.njobs, s)) exit(3);
            } else {
                mmqg::SkinnyBwdJob jobs[3];
                for (int i = 0; i < cs_.njobs; ++i) jobs[i] = bwd_job(cs_.njobs == 1 ? cs_.layer : i, t);
                if (mmqg::skinny_cell_bwd_multi(jobs, cs_.njobs, s)) exit(3);
            }
        };
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_skinny_trace), &null_trace, sizeof(null_trace)));
        for (int t = 0; t < 3; ++t) enqueue(t);
        CK(hipStreamSynchronize(s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < chain; ++t) enqueue(t);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        // traced replay: every launch stamps, the last one's stamps survive
        CK(hipMemset(trace, 0, (size_t)max_wg * 64));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_skinny_trace), &trace, sizeof(trace)));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> tr((size_t)max_wg * 8);
        CK(hipMemcpy(tr.data(), trace, tr.size() * 8, hipMemcpyDeviceToHost));
        report(cs_.name, tr, cs_.nwg, ms * 1e3f / chain);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // two independent chains of single-job forward layer-steps on two streams: do latency-bound kernels overlap?
    {
        hipStream_t s2; CK(hipStreamCreate(&s2));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_skinny_trace), &null_trace, sizeof(null_trace)));
        float* hs2 = dalloc((size_t)L * (chain + 2) * B * H, 0.5f);
        float* cs2 = dalloc((size_t)L * (chain + 2) * B * H, 0.5f);
        float* gates2 = dalloc((size_t)L * B * 4 * H, 0.5f);
        auto run = [&](hipStream_t st, float* hsb, float* csb, float* gb, int t) {
            mmqg::SkinnyFwdJob j = fwd_job(1, t);
            j.pairs[0].A = hsb + (size_t)0 * (chain + 2) * BH + (size_t)(t + 1) * BH;
            j.pairs[1].A = hsb + (size_t)1 * (chain + 2) * BH + (size_t)t * BH;
            j.cell.gates = gb + (size_t)B * 4 * H;
            j.cell.h_prev = j.pairs[1].A; j.cell.c_prev = csb + (size_t)1 * (chain + 2) * BH + (size_t)t * BH;
            j.cell.h_out = hsb + (size_t)1 * (chain + 2) * BH + (size_t)(t + 1) * BH;
            j.cell.c_out = csb + (size_t)1 * (chain + 2) * BH + (size_t)(t + 1) * BH;
            if (mmqg::skinny_cell_fwd_multi(&j, 1, st)) exit(3);
        };
        for (int rep = 0; rep < 2; ++rep) {
            for (int mode = 0; mode < 2; ++mode) {       // 0: one chain, 1: two chains side by side
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, s));
                hipEvent_t f; CK(hipEventCreate(&f)); CK(hipEventRecord(f, s)); CK(hipStreamWaitEvent(s2, f, 0));
                for (int t = 0; t < chain; ++t) { run(s, hs, cs, gates, t); if (mode) run(s2, hs2, cs2, gates2, t); }
                hipEvent_t jn; CK(hipEventCreate(&jn)); CK(hipEventRecord(jn, s2)); CK(hipStreamWaitEvent(s, jn, 0));
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("%s of %d forward layer-steps: %.2f us per step\n", mode ? "TWO chains side by side" : "one chain", chain, ms * 1e3f / chain);
            }
        }
    }
    return 0;
}
