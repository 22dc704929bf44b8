// dense_rows_gemm_kernel: out[m][n] = sum_k A[m][k] W[n][k] (+ res[m][n]) for plain dense products whose whole reduction fits one
// pass -- the 1x1 convolutions of ConvTasNet (src/model/conv_tasnet.py:307-402: bottleneck 128 -> 128, block input 128 -> 256, block
// output 256 -> 128 with the residual, mask 128 -> 256, and their input gradients), M = 51 168 rows at the C4 shape, N, K in
// {128, 256}.  Such a product moves 40-51 MB and does 3-7 GFLOP: HBM-bound by a wide margin, yet the table-gathered gemm_kernel ran
// it in 17-20 us (2.1-3.1 TB/s; 60 launches = 1.09 ms of the 3.4-ms C4 step): operands staged through registers, a K loop of 2-4
// short steps whose prologue / epilogue dominate.
//
// Here (round 6): a 256-thread workgroup takes 64 rows at a time.  W never touches LDS: wave w owns N / 4 output columns and keeps
// their W fragments for ALL of K in registers for the whole launch (16 fragments = 64 VGPRs), loaded once straight from the packed
// [N][K] weights in the MFMA operand layout.  The 64 x K tile of A reaches LDS by LDS-DMA as whole rows (coalesced: a row's 16-byte
// pieces are permuted inside the row, piece p of row r at slot p ^ (r & 15), so that the 16 lanes of every bank group of a
// ds_read_b128 fragment read -- rows r .. r + 15 at piece p -- hit 16 different 16-byte slots).  The product is formed TRANSPOSED
// (a = W fragment, b = A fragment: D[n][m]), which leaves a lane 4 consecutive output columns of one row: 8-byte pieces into a
// row-major staging tile (same XOR by row), from where the tile leaves as 16-byte coalesced stores with the residual added on the way.
// No software pipelining inside a workgroup (3 workgroups per CU overlap each other's DMA / MFMA / store phases): every wait is a
// plain vmcnt(0), so hipcc's conservative waits around LDS-DMA cost nothing.  Rows past M are out-of-range DMA offsets: zeros.
#include <stdlib.h>
#include "common.h"
#include "../../../include/sehip.h"

typedef __attribute__((address_space(3))) void dg_lds_void;
#define DG_OOB 0x7ffffff0u

// RES: the product adds a residual tensor (ConvTasNet: the block output x + W u, the input gradient through the residual connection):
// the tile is then staged in fp32, so that product + residual is rounded ONCE (a bf16 staging tile would round twice: up to 1.5 ulp)
// gstats (non-RES products only): += per utterance (TT rows) the sum and sum of squares of PReLU(stored bf16 value; *gslope) -- the gLN
// statistics of the tensor this product writes (sehip_gemm_desc.gln_stats), taken from the staged tile on its way out
template <int N, int K, bool RES>
__global__ __launch_bounds__(256, 2) void dense_rows_gemm_kernel(const bf16_raw* __restrict__ A, const bf16_raw* __restrict__ W,
                                                                 const bf16_raw* __restrict__ res, bf16_raw* __restrict__ out, int M,
                                                                 int stages_per_wg, double* __restrict__ gstats,
                                                                 const float* __restrict__ gslope, int TT) {
    constexpr int ROWS = 64, PPR = K / 8, ROWB = K * 2;            // rows per stage, 16-byte pieces per row of A, bytes per row of A
    constexpr int ABYTES = ROWS * ROWB, OPR = N / 8, OROWB = N * (RES ? 4 : 2); // 8-column groups / bytes per row of the output tile
    constexpr int NW = N / 4, NT = NW / 16, KS = K / 32;           // columns per wave, its 16-column tiles, MFMA k steps
    constexpr int AI = ROWS * PPR / 256, OI = ROWS * OPR / 256;    // DMA instructions / output pieces per thread and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* abuf = smem;
    unsigned char* obuf = smem + ABYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int nstages = (M + ROWS - 1) / ROWS;
    const int s_begin = blockIdx.x * stages_per_wg, s_end = min(nstages, s_begin + stages_per_wg);
    if (s_begin >= s_end) return;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(A), 0, (unsigned)((size_t)M * ROWB), 0x00020000);
    // this wave's W fragments: column n0w + 16 nt + c, k = 32 ks + 8 g .. + 7 (the MFMA operand layout as it lies in [N][K])
    const int n0w = wave * NW;
    bf16x8 wf[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[nt][ks] = *reinterpret_cast<const bf16x8*>(W + (size_t)(n0w + nt * 16 + c) * K + ks * 32 + 8 * g);
    // DMA pieces of this thread: piece i = u * 256 + tid of the tile = row i / PPR, slot i % PPR <- source piece slot ^ (row & 15)
    unsigned aoff[AI];
#pragma unroll
    for (int u = 0; u < AI; ++u) {
        const int i = u * 256 + tid, r = i / PPR, sl = i % PPR;
        aoff[u] = (unsigned)(r * ROWB + ((sl ^ (r & 15)) * 16));
    }
    // gLN statistics: the rows of this workgroup (at most stages_per_wg x 64 < TT) touch at most two utterances, ua and ua + 1
    const bool gl = !RES && gstats != nullptr;
    const float gsl = gl ? gslope[0] : 0.f;
    const int ua = gl ? (s_begin * ROWS) / TT : 0;
    const int ub_row = (ua + 1) * TT;                 // first row of utterance ua + 1
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
    for (int s = s_begin; s < s_end; ++s) {
        const int m0 = s * ROWS;
        // (the previous stage's store phase has read obuf, its MFMA phase abuf: every thread is past both when it arrives here; the
        //  barrier below the DMA wait orders them against this stage's writes)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < AI; ++u) {
            const int r = (u * 256 + tid) / PPR;
            const unsigned vo = m0 + r < M ? (unsigned)m0 * (unsigned)ROWB + aoff[u] : DG_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (dg_lds_void*)(abuf + (u * 256 + wave * 64) * 16), 16, vo, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 acc[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(abuf + (mt * 16 + c) * ROWB + (((ks * 4 + g) ^ c) * 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], af, acc[mt][nt], 0, 0, 0);   // D[n][m]
            }
        }
        // D rows n = n0w + 16 nt + 4 g + u, column m = 16 mt + c: 4 consecutive output columns of one row per lane
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int row = mt * 16 + c, n = n0w + nt * 16 + 4 * g;
                const f32x4 v = acc[mt][nt];
                if (RES)       // fp32 tile: 16-byte pieces of 4 columns, piece n / 4 of row r at slot (n / 4) ^ (r & 15)
                    *reinterpret_cast<float4*>(obuf + row * OROWB + (((n >> 2) ^ (row & 15)) * 16)) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    *reinterpret_cast<uint2*>(obuf + row * OROWB + (((n >> 3) ^ (row & 15)) * 16) + (n & 4) * 2) =
                        make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < OI; ++u) {
            const int j = u * 256 + tid, row = j / OPR, pc = j % OPR;
            if (m0 + row < M) {
                const size_t o = (size_t)(m0 + row) * N + pc * 8;
                uint4 v;
                if (RES) {
                    const float4 a0 = *reinterpret_cast<const float4*>(obuf + row * OROWB + (((2 * pc) ^ (row & 15)) * 16));
                    const float4 a1 = *reinterpret_cast<const float4*>(obuf + row * OROWB + (((2 * pc + 1) ^ (row & 15)) * 16));
                    const uint4 r = *reinterpret_cast<const uint4*>(res + o);
                    v = make_uint4(pack_bf2(a0.x + __uint_as_float(r.x << 16), a0.y + __uint_as_float(r.x & 0xffff0000u)),
                                   pack_bf2(a0.z + __uint_as_float(r.y << 16), a0.w + __uint_as_float(r.y & 0xffff0000u)),
                                   pack_bf2(a1.x + __uint_as_float(r.z << 16), a1.y + __uint_as_float(r.z & 0xffff0000u)),
                                   pack_bf2(a1.z + __uint_as_float(r.w << 16), a1.w + __uint_as_float(r.w & 0xffff0000u)));
                } else {
                    v = *reinterpret_cast<const uint4*>(obuf + row * OROWB + ((pc ^ (row & 15)) * 16));
                    if (gl) {
                        const unsigned w4[4] = {v.x, v.y, v.z, v.w};
                        float ps = 0.f, pq = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float x0 = __uint_as_float(w4[i] << 16), x1 = __uint_as_float(w4[i] & 0xffff0000u);
                            x0 = x0 > 0.f ? x0 : gsl * x0; x1 = x1 > 0.f ? x1 : gsl * x1;
                            ps += x0 + x1; pq += x0 * x0 + x1 * x1;
                        }
                        if (m0 + row < ub_row) { gs0 += ps; gq0 += pq; } else { gs1 += ps; gq1 += pq; }
                    }
                }
                *reinterpret_cast<uint4*>(out + o) = v;
            }
        }
    }
    if (gl) {
        __shared__ float gred[4][4];
        gs0 = wave_sum(gs0); gq0 = wave_sum(gq0); gs1 = wave_sum(gs1); gq1 = wave_sum(gq1);
        if (lane == 0) { gred[wave][0] = gs0; gred[wave][1] = gq0; gred[wave][2] = gs1; gred[wave][3] = gq1; }
        __syncthreads();
        if (tid < 4) {
            const double t = (double)gred[0][tid] + gred[1][tid] + gred[2][tid] + gred[3][tid];
            const int u = ua + (tid >> 1);
            if ((long)u * TT < M && (tid < 2 || (long)s_end * ROWS > ub_row)) atomicAdd(&gstats[2 * u + (tid & 1)], t);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (no DMA is in flight here; the stores drain before the LDS is given back anyway)
}

template <int N, int K, bool RES>
static int dg_launch(const sehip_gemm_desc& d, hipStream_t st) {
    constexpr size_t lds = (size_t)64 * K * 2 + (size_t)64 * N * (RES ? 4 : 2);
    static unsigned char state[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
    if (state[dev] == 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_rows_gemm_kernel<N, K, RES>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) (void)hipGetLastError();
        state[dev] = e == hipSuccess ? 1 : 2;
    }
    if (state[dev] != 1) return 0;
    // 64-row stages per workgroup (C4 step, ms: 1: 3.29, 2: 3.18, 3: 3.29, 4: 3.31, 6: 3.61; the generic kernel: 3.41)
    static const int spw = getenv("SEHIP_DG_STAGES") ? atoi(getenv("SEHIP_DG_STAGES")) : 2;
    const int nstages = (d.M + 63) / 64;
    int per = spw < 1 ? 1 : spw;
    // (deterministic schedule: the statistics' double atomics are not part of it -- sehip_gemm_takes_gln_stats answers 0 and the
    //  caller runs sehip_ctn_gln_stats, whose workgroups add in a fixed order)
    const bool gl = !RES && d.gln_stats != nullptr && d.gln_slope != nullptr && !sehip_deterministic();
    while (gl && per > 1 && per * 64 >= d.TT) --per;       // (a workgroup's rows may touch two utterances, not three)
    sehip_note_kernel("dense_rows_gemm_kernel<%d, %d, %d>", N, K, RES ? 1 : 0);
    dense_rows_gemm_kernel<N, K, RES><<<(nstages + per - 1) / per, 256, lds, st>>>(
        reinterpret_cast<const bf16_raw*>(d.src[0].ptr), reinterpret_cast<const bf16_raw*>(d.W), reinterpret_cast<const bf16_raw*>(d.res),
        reinterpret_cast<bf16_raw*>(d.dst[0].ptr), d.M, per, gl ? d.gln_stats : nullptr, gl ? d.gln_slope : nullptr, d.TT);
    return 1;
}

// returns 1 if the kernel was launched, 0 if the descriptor does not qualify (the caller goes on to gemm_kernel).  The caller vouches
// for dense rows through sehip_gemm_desc.dense_rows (ONE source whose row m is the K contiguous elements at m K, ONE bf16 destination
// whose row m is the N contiguous elements at m N, every frame valid, trivial tables); the geometry is checked here all the same.
static bool dg_qualifies(const sehip_gemm_desc& d);
extern "C" int sehip_gemm_takes_gln_stats(const sehip_gemm_desc* d) {
    return d && !d->res && d->TT > 64 && !sehip_deterministic() && dg_qualifies(*d) ? 1 : 0;
}
int sehip_try_dense_rows_gemm(const sehip_gemm_desc& d, hipStream_t st) {
    if (!dg_qualifies(d)) return 0;
    if (d.res) {                  // (fp32 staging tile: 64 x 128 x 4 B beside the 64 x K tile of A -- the 128-output products only)
        if (d.Npad == 128 && d.K == 256) return dg_launch<128, 256, true>(d, st);
        if (d.Npad == 128 && d.K == 128) return dg_launch<128, 128, true>(d, st);
        return 0;
    }
    if (d.Npad == 256 && d.K == 128) return dg_launch<256, 128, false>(d, st);
    if (d.Npad == 128 && d.K == 256) return dg_launch<128, 256, false>(d, st);
    if (d.Npad == 128 && d.K == 128) return dg_launch<128, 128, false>(d, st);
    return 0;
}
static bool dg_qualifies(const sehip_gemm_desc& d) {
    static const bool off = getenv("SEHIP_NO_DENSE_GEMM") != nullptr;
    if (off || !d.dense_rows || d.cv_nf > 0 || d.cv2_nkt > 0 || d.J != 1 || d.tmul > 1 || d.bias || d.stats || d.w_tiled) return false;
    if (d.src[1].ptr || d.dst[1].ptr || d.dst[0].is_f32 || d.N != d.Npad) return false;
    if (d.src[0].C != d.K || d.src[0].F != 1 || d.dst[0].C != d.Npad || d.dst[0].F != 1) return false;
    if (d.src[0].T != d.TT || d.src[0].tlo != 0 || d.src[0].thi != d.TT || d.dst[0].T != d.TT || d.dst[0].toff || d.dst[0].fadd ||
        d.dst[0].tmul > 1 || d.M % d.TT) return false;
    if ((size_t)d.M * 256 * 2 >= (1ull << 31) - (1u << 20)) return false;
    if (d.res) return d.Npad == 128 && (d.K == 256 || d.K == 128);
    return (d.Npad == 256 && d.K == 128) || (d.Npad == 128 && d.K == 256) || (d.Npad == 128 && d.K == 128);
}
