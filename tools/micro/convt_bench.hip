// Standalone timing harness of convt_stream_kernel's five variants (csrc/convt.hip; the other templates of the file are timed in the step: tools/_tl-style traces): synthetic DCCRN-shaped operands, hipEvent time per launch and the
// core-clock cycles wave 0 of workgroup 0 spends in each phase of the frame loop.  Results are not checked here (tests/ do that).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/micro/convt_bench.hip -o tools/micro/_convt_bench
//   SEHIP_CT_ABL=<bits> SEHIP_CT_CHUNKS=<n> tools/micro/_convt_bench [variant 0..4] [B] [T]
#define SEHIP_TOOLS_BUILD
#define CT_PHASE_TIMERS
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../speech-enhancement-pytorch_amd/sehip/csrc/convt.hip"

void sehip_note_kernel(const char*, ...) {}
int sehip_deterministic(void) { return 0; }
float* sehip_wgrad_scratch(hipStream_t, size_t bytes) {        // (the library's per-stream pool: one buffer is enough here)
    static float* p = nullptr; static size_t have = 0;
    if (bytes > have) { if (p) (void)hipFree(p); if (hipMalloc(&p, bytes) != hipSuccess) return nullptr; have = bytes; }
    return p;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const int B = argc > 2 ? atoi(argv[2]) : 16, T = argc > 3 ? atoi(argv[3]) : 641;
    const int Cs[5] = {64, 32, 64, 32, 16}, NSs[5] = {2, 2, 1, 1, 2}, COs[5] = {32, 16, 32, 16, 2}, Js[5] = {32, 64, 32, 64, 128};
    const int C = Cs[variant], NS = NSs[variant], CO = COs[variant], J = Js[variant];
    const bool stats = variant < 2, res = variant == 2 || variant == 3, mask = variant == 4;
    const int npad = mask ? 16 : CO, osz = mask ? 4 : 2;
    const size_t in_elems = (size_t)B * T * J * C, out_elems = (size_t)B * T * 2 * J * CO;
    void *src[2] = {nullptr, nullptr}, *out, *resp = nullptr, *W0, *W1;
    float *bias, *st;
    for (int s = 0; s < NS; ++s) { CK(hipMalloc(&src[s], in_elems * 2)); CK(hipMemset(src[s], 0x3c, in_elems * 2)); }
    CK(hipMalloc(&out, out_elems * osz));
    if (res) { CK(hipMalloc(&resp, out_elems * 2)); CK(hipMemset(resp, 0x3c, out_elems * 2)); }
    const int K0 = 2 * 3 * NS * C, K1 = 2 * 2 * NS * C;
    CK(hipMalloc(&W0, (size_t)npad * K0 * 2)); CK(hipMemset(W0, 0x3c, (size_t)npad * K0 * 2));
    CK(hipMalloc(&W1, (size_t)npad * K1 * 2)); CK(hipMemset(W1, 0x3c, (size_t)npad * K1 * 2));
    CK(hipMalloc(&bias, 64 * 4)); CK(hipMemset(bias, 0, 64 * 4));
    CK(hipMalloc(&st, 8 * 5 * 64 * 4)); CK(hipMemset(st, 0, 8 * 5 * 64 * 4));
    sehip_gemm_desc d[2];
    memset(d, 0, sizeof(d));
    for (int p = 0; p < 2; ++p) {
        sehip_gemm_desc& x = d[p];
        for (int s = 0; s < NS; ++s) {
            x.src[s].ptr = src[s]; x.src[s].T = T; x.src[s].tlo = 0; x.src[s].thi = T; x.src[s].F = J; x.src[s].C = C;
            x.cv_toff[s][0] = -1; x.cv_toff[s][1] = 0;
        }
        x.dst[0].ptr = out; x.dst[0].T = T; x.dst[0].F = 2 * J; x.dst[0].C = CO; x.dst[0].fmul = 2; x.dst[0].fadd = p; x.dst[0].tmul = 1; x.dst[0].is_f32 = mask;
        x.W = p ? W1 : W0; x.K = p ? K1 : K0; x.M = B * T * J; x.N = CO; x.Npad = npad; x.TT = T; x.J = J; x.fmul = 1; x.tmul = 1;
        x.cv_nf = p ? 2 : 3; x.cv_fadd = p ? 0 : -1;
        if (stats) { x.stats = st; x.stats_cr = CO / 2; x.bias = bias; } else if (res) x.res = resp; else x.bias = bias;
    }
    hipStream_t q; CK(hipStreamCreate(&q));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int r = 0; r < 3; ++r) if (!sehip_try_convt_stream(d[0], d[1], q, false)) { printf("not taken\n"); return 1; }
    CK(hipStreamSynchronize(q));
    unsigned long long z[8] = {0}, ph[8];
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ct_phase), z, sizeof(z)));
    CK(hipEventRecord(e0, q));
    for (int r = 0; r < reps; ++r) sehip_try_convt_stream(d[0], d[1], q, false);
    CK(hipEventRecord(e1, q)); CK(hipStreamSynchronize(q));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(ct_phase), sizeof(ph)));
    const double bytes = (double)NS * in_elems * 2 + out_elems * osz * (res ? 2 : 1);
    printf("variant %d (C %d NS %d CO %d J %d) B %d T %d abl %s chunks %s: %.1f us/launch, %.2f TB/s algorithmic\n", variant, C, NS, CO, J, B, T,
           getenv("SEHIP_CT_ABL") ? getenv("SEHIP_CT_ABL") : "0", getenv("SEHIP_CT_CHUNKS") ? getenv("SEHIP_CT_CHUNKS") : "16", 1e3 * ms / reps,
           bytes / (1e-3 * ms / reps) * 1e-12);
    const char* nm[8] = {"pre-loop", "vmcnt wait", "barrier", "store phase", "compute", "dma issue", "drain", "epilogue"};
    for (int k = 0; k < 8; ++k) printf("   %-12s %9.0f cycles per launch\n", nm[k], (double)ph[k] / reps);
    return 0;
}
