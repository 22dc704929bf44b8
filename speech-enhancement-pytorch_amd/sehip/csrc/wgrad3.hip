// conv_wgrad_v3_kernel: weight gradient of the deep DCCRN convolutions (ComplexConv2d / ComplexConvTranspose2d, src/model/dccrn.py:316-450;
// what loss.backward() computes for their weights), the descriptors with the regular-convolution description, >= 64 input channels
// and a multiple of 128 output columns.
//
// What bounded conv_wgrad_kernel (0.11-0.25 of the dense bf16 MFMA peak): both operands staged through registers + ds_write,
// two barriers per 128-row tile, wave tiles of 32 n x (taps x 16 channels) = 24 transposed LDS reads per 20 MFMAs.  Here, per
// 512-thread workgroup (one per CU): dW[128 n][2 NF taps][64 channels] in registers, waves as 2 (n) x 4 (16-channel planes),
// wave tile 64 n x taps x 16 channels = 28 transposed reads per 40 MFMAs; the reduction runs over 64-row stages:
//   * BOTH operands reach LDS by LDS-DMA (buffer_load ... lds), ring of NB stage buffers, one barrier per stage, counted vmcnt;
//   * every image is a stack of [rows][32 B] planes (16 columns of dOut / 16 channels of the input patch): the 8 consecutive
//     rows that a 32-lane half reads in one ds_read_b64_tr_b16 are 256 contiguous bytes -- conflict free without padding or
//     swizzle.  The patch plane is frames x [parity planes of a stride-2 layer] x rows as in conv_gemm_v3, so that consecutive
//     output rows of one tap are consecutive physical rows; J = 4: frame stride 12 rows (the two 4-row runs of neighbouring
//     frames are 384 B apart: disjoint banks);
//   * padding (frames outside the source, rows outside the frequency range, rows of dOut past the last frame) is a buffer offset
//     beyond num_records: the hardware writes zeros.
// The MFMA's k index <-> tile row mapping is conv_wgrad_kernel's (lane group g, element j: row 32 ks + 16 (j >> 2) + 4 g + (j & 3)).
#include <stdlib.h>
#include "common.h"
#include "../../../include/sehip.h"

typedef __attribute__((address_space(3))) void w3_lds_void;
typedef __attribute__((address_space(3))) s16x4 w3_lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short w3_s16x8;

#define W3_OOB 0x7ffffff0u
#define W3_ONES __builtin_bit_cast(bf16x8, (w3_s16x8){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80})

template <int NF, int FM, int J, int NPL = 4>
struct W3Geo {
    static constexpr int NWV = 2 * NPL, NTH = 64 * NWV;                // waves = 2 (64-column halves of the 128 n) x NPL (16-channel planes)
    static constexpr int GI = 1024 / NTH;                              // dOut DMA instructions per thread and stage
    static constexpr int FR = (J - 1) * FM + NF;                       // patch rows per frame
    static constexpr int S = J == 4 ? 12 : FR;                         // frame stride in rows
    static constexpr int P1 = FM == 2 ? (FR + 1) / 2 : 0;              // first physical row of the odd-row plane
    static constexpr int TB = 64 / J;                                  // frames per stage
    static constexpr int NPP = (TB + 1) * S * 2;                       // 16-byte pieces per 16-channel plane
    static constexpr int PLB = NPP * 16;
    static constexpr int NPIECE = NPL * NPP;
    static constexpr int MAXP = (NPIECE + NTH - 1) / NTH;              // patch DMA instructions per thread and stage
    static constexpr int NI = GI + MAXP;                               // DMA instructions per thread and stage
    static constexpr int STAGE = (16384 + NPL * PLB + 1023) / 1024 * 1024;
    static_assert(S >= FR && (FM == 1 || P1 + FR / 2 <= S), "frame stride");
    static_assert(J == 4 || J == 8 || J == 16 || J == 32 || J == 64, "rows per frame");
};

template <int N>
__device__ __forceinline__ void w3_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ABL: timing ablations for tools/ (SEHIP_W3_ABL, wrong results): 1 = no DMA inside the loop, 2 = no MFMA, 4 = no fragment reads
// NPL: 16-channel planes of the patch per workgroup.  4 (round 3): 512 threads, dW[128 n][taps][64 channels], one workgroup takes a
// CU's whole register file (8 waves x 234 VGPRs for the 5-tap layers).  2 (round 6, the "half-CU" form the round-5 verdict asked
// for): 256 threads = ONE wave per SIMD with the same wave tile, dW[128 n][taps][32 channels], <= 240 of a SIMD's 512 VGPRs and
// 59 KB of LDS -- a conv_gemm_v3 workgroup (4 waves x 256 VGPRs) or a BatchNorm pass fits beside it on the same CU, which is what
// the weight-gradient queue needs next to the step's dependent chain (finding 1).
template <int NF, int FM, int J, int NB, int NPL = 4, int ABL = 0>
__global__ __launch_bounds__(128 * NPL, 2) void conv_wgrad_v3_kernel(const sehip_gemm_desc d, int tiles_per_wg, int nsplit, float* scratch) {
    using G = W3Geo<NF, FM, J, NPL>;
    constexpr int NIT = 2 * NF, S = G::S, P1 = G::P1, FR = G::FR, TB = G::TB, NPP = G::NPP, PLB = G::PLB, NPIECE = G::NPIECE;
    constexpr int MAXP = G::MAXP, NI = G::NI, STAGE = G::STAGE, NWV = G::NWV, GI = G::GI, CW = 16 * NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* dump = smem + NB * STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv % NPL, nh = wv / NPL;           // wave -> 16-channel plane w of the patch, 64-wide half nh of the 128 output columns
    // All (n-tile, channel-chunk) workgroups of one m-split read the same dOut rows and patches at the same time: one XCD per
    // group of splits (blocks are dealt round-robin over the 8 XCDs), so the re-reads hit its L2
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int Ctot = C0 + C1;
    const int ntn = d.Npad >> 7;
    const int gx = ntn * (Ctot / CW);
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int xi = rr % gx, split = (rr / gx) * 8 + xcd;       // nsplit is a multiple of 8
    (void)nsplit;
    const int nt = xi % ntn, cc = xi / ntn;
    const int n0 = nt * 128;
    const int tblocks = (d.TT + TB - 1) / TB;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * tblocks;
    const int mt_begin = split * tiles_per_wg, mt_end = min(MT, mt_begin + tiles_per_wg);
    const int ns = mt_end - mt_begin;
    if (ns <= 0) return;

    const bool second = cc * CW >= C0;
    const sehip_src& Sr = second ? d.src[1] : d.src[0];
    const int sT = Sr.T, sF = Sr.F, sC = Sr.C, tlo = Sr.tlo, thi = Sr.thi;
    const int cbase = cc * CW - (second ? C0 : 0);
    const int tmin = second ? min(d.cv_toff[1][0], d.cv_toff[1][1]) : min(d.cv_toff[0][0], d.cv_toff[0][1]);
    const int dt0 = (second ? d.cv_toff[1][0] : d.cv_toff[0][0]) - tmin, dt1 = (second ? d.cv_toff[1][1] : d.cv_toff[0][1]) - tmin;
    const int f0 = d.cv_fadd;
    const sehip_dst& dd = d.dst[0];
    // num_records = the tensor's own size (round 6; it was a fixed 0x7fff0000): an offset that leaves the tensor reads zeros like the
    // padding marker does, not a neighbour's bytes.  (Below 2^30 bytes: sehip_try_conv_wgrad_v3 checks.)
    const unsigned recx = 2u * (unsigned)B * (unsigned)sT * (unsigned)sF * (unsigned)sC;
    const unsigned recg = 2u * (unsigned)B * (unsigned)dd.T * (unsigned)dd.F * (unsigned)dd.C;
    const __amdgpu_buffer_rsrc_t rsx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(reinterpret_cast<const bf16_raw*>(Sr.ptr)), 0, recx, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsg =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(reinterpret_cast<const bf16_raw*>(dd.ptr)), 0, recg, 0x00020000);
    const unsigned sframe = 2u * (unsigned)(sF * sC), gframe = 2u * (unsigned)(dd.F * dd.C);     // bytes per time frame

    // ---- DMA pieces of this thread.  dOut image: 8 planes (16 columns) x 64 rows x 32 B; piece Q = (NWV u + wave) * 64 + lane =
    // plane Q >> 7, row (Q & 127) >> 1, half Q & 1.  Patch image: NPL planes (16 channels) x (TB + 1) frames x S rows x 32 B.
    unsigned gconst[GI], pconst[MAXP];                // byte offset for stage frame 0 of batch item 0 (W3_OOB: always zero)
    int gtl[GI], pfr[MAXP];                           // frame inside the stage
#pragma unroll
    for (int u = 0; u < GI; ++u) {
        const int Q = (u * NWV + wv) * 64 + lane;
        const int plane = Q >> 7, m = (Q & 127) >> 1, half = Q & 1;
        const int tl = m / J, jl = m - tl * J;
        const int n = n0 + plane * 16 + half * 8;
        const sehip_nchunk c0 = d.ntab[n >> 2], c1 = d.ntab[(n >> 2) + 1];
        const bool ok = c0.nvalid == 4 && c1.nvalid == 4 && c0.dst == 0 && c1.dst == 0 && c1.coff == c0.coff + 4;
        gtl[u] = tl;
        gconst[u] = ok ? 2u * (unsigned)((dd.toff * dd.F + jl * dd.fmul + dd.fadd) * dd.C + c0.coff) + (unsigned)tl * gframe : W3_OOB;
    }
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
        const int P = (u * NWV + wv) * 64 + lane;
        const int pl = P / NPP, rem = P - pl * NPP;
        const int prow = rem >> 1, half = rem & 1;
        const int p = prow / S, rs_ = prow - p * S;
        int r;
        if (FM == 2) { if (rs_ < P1) r = 2 * rs_; else r = 2 * (rs_ - P1) + 1; } else r = rs_;
        const int f = f0 + r;
        const bool ok = P < NPIECE && r < FR && (unsigned)f < (unsigned)sF;
        pfr[u] = p;
        pconst[u] = ok ? 2u * (unsigned)(f * sC + cbase + pl * 16 + half * 8) + (unsigned)p * sframe : W3_OOB;
    }
    auto issue = [&](int mt, int buf) {               // mt is uniform; past the end everything is padding (constant DMA count)
        const bool live = mt < mt_end;
        const int b_ = mt / tblocks, t0_ = (mt - b_ * tblocks) * TB;
        unsigned char* base = smem + buf * STAGE + wv * 1024;
        const unsigned gb = (unsigned)(b_ * dd.T + t0_) * gframe;
#pragma unroll
        for (int u = 0; u < GI; ++u) {
            const unsigned gc = gconst[u];
            const unsigned vo = (live && gc != W3_OOB && t0_ + gtl[u] < d.TT) ? gc + gb : W3_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (w3_lds_void*)(base + u * (NWV * 1024)), 16, vo, 0, 0, 0);
        }
        const int ts0 = t0_ + tmin;
        const unsigned pb = (unsigned)(b_ * sT + ts0) * sframe;      // (may wrap for ts0 = -1: added to a piece of frame >= 1 only)
#pragma unroll
        for (int u = 0; u < MAXP; ++u) {
            const unsigned pc = pconst[u];
            const int ts = ts0 + pfr[u];
            const unsigned vo = (live && pc != W3_OOB && ts >= tlo && ts < thi) ? pc + pb : W3_OOB;
            unsigned char* dst = ((u * NWV + wv) * 64 < NPIECE) ? base + 16384 + u * (NWV * 1024) : dump;      // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (w3_lds_void*)dst, 16, vo, 0, 0, 0);
        }
    };

    // ---- transposed-read addresses: lane supplies row 4 g + q (+ 16 h + 32 ks) of a 32-row k step, columns 4 p .. 4 p + 3
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    const int ml = 4 * g + q;
    const int ga = (nh * 4) * 2048 + ml * 32 + 8 * p4;                                   // + ni * 2048 + ks * 1024 + h * 512
    const int tl_l = J <= 16 ? ml / J : 0, jl_l = J <= 16 ? ml % J : ml;
    const int pa = 16384 + w * PLB + (tl_l * S + jl_l) * 32 + 8 * p4;
    const int paA = pa + dt0 * (S * 32), paB = pa + dt1 * (S * 32);
    auto imm_of = [](int ks, int h) constexpr {
        const int m = 32 * ks + 16 * h;
        return J <= 16 ? (m / J) * S * 32 : ((m / J) * S + m % J) * 32;
    };
    auto tap_off = [](int tap) constexpr { return (FM == 2 ? ((tap & 1) * P1 + (tap >> 1)) : tap) * 32; };

    f32x4 acc[4][NIT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NIT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = d.dbias != nullptr && cc == 0 && w == 0;          // column sums of dOut by MFMA against ones
    f32x4 accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NB - 1; ++s) issue(mt_begin + s, s);
    int buf = 0;
    for (int s = 0; s < ns; ++s) {
        // stage s has landed (the NB - 2 younger stages may still be in flight); behind the barrier every wave has also finished
        // reading stage s - 1, whose buffer the DMA of stage s + NB - 1 overwrites
        if (ABL & 1) w3_wait_vm<0>(); else w3_wait_vm<(NB - 2) * NI>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int nb = buf == 0 ? NB - 1 : buf - 1;
            if (!(ABL & 1)) issue(mt_begin + s + NB - 1, nb);
        }
        const unsigned char* sb = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 gf[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                s16x4 lo, hi;
                if (ABL & 4) { lo = (s16x4){(short)(0x3f80 + s), 0x3f80, 0x3f80, (short)lane}; hi = lo; }
                else {
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)(sb + ga + ni * 2048 + ks * 1024));
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)(sb + ga + ni * 2048 + ks * 1024 + 512));
                }
                gf[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            if (do_bias) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) accb[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], W3_ONES, accb[ni], 0, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int o = (it < NF ? paA : paB) + tap_off(it < NF ? it : it - NF);
                s16x4 lo, hi;
                if (ABL & 4) { lo = (s16x4){(short)(0x3f80 + it), 0x3f80, (short)s, (short)lane}; hi = lo; }
                else {
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)(sb + o + imm_of(ks, 0)));
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)(sb + o + imm_of(ks, 1)));
                }
                const bf16x8 xf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                if (ABL & 2) { asm volatile("" ::"v"(xf)); if (it == 0) { asm volatile("" ::"v"(gf[0]), "v"(gf[1]), "v"(gf[2]), "v"(gf[3])); } }
                else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[ni][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], xf, acc[ni][it], 0, 0, 0);
                }
            }
        }
        buf = buf == NB - 1 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may land after the workgroup has given its LDS back

    // D rows = n (4 (lane >> 4) + u), columns = channel (lane & 15).  The flush: plain stores into this split's own [Npad][K]
    // array (w3_reduce_kernel adds the arrays into dW), or atomics when there is no scratch.  Measured (tools/micro/atomic_bench.hip,
    // 256 workgroups x 327 KB): fp32 atomics 65-68 us whatever the scope, stores 14 us + 14 us for the reduction.
    float* out = scratch ? scratch + (size_t)split * d.Npad * d.K : d.dW;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = n0 + 64 * nh + ni * 16 + 4 * (lane >> 4);
            const int k = it * Ctot + cc * CW + 16 * w + (lane & 15);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (scratch) out[(size_t)(n + u) * d.K + k] = acc[ni][it][u];
                else atomicAdd(&out[(size_t)(n + u) * d.K + k], acc[ni][it][u]);
            }
        }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int u = 0; u < 4; ++u) atomicAdd(&d.dbias[n0 + 64 * nh + ni * 16 + 4 * (lane >> 4) + u], accb[ni][u]);
    }
}

// dW[i] += sum over the splits' arrays (n4 = Npad K / 4 float4 each)
__global__ __launch_bounds__(256) void w3_reduce_kernel(const float* parts, int nparts, size_t n4, float* dW) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    // gridDim.y slices of the partial arrays (their sums meet with atomics).  One slice by default: a 32-K-entry dW is then 32
    // workgroups of columns (ConvTasNet's 1x1 products, 160 arrays: 16.5 us per launch at 1.2 TB/s), and that is what the step
    // wants -- the reduction runs on the weight-gradient stream beside an HBM-bound chain, and 16 slices (512 workgroups, a third
    // of the time per launch) made the C4 step 3.75 ms instead of 3.51 (DCCRN: no difference).  SEHIP_W3_REDUCE_WGS=<workgroups>
    const int per = (nparts + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = p0 + per < nparts ? p0 + per : nparts;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int p = p0; p < p1; ++p) {
        const float4 v = reinterpret_cast<const float4*>(parts)[p * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (gridDim.y == 1) {
        float4 o = reinterpret_cast<const float4*>(dW)[i];
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        reinterpret_cast<float4*>(dW)[i] = o;
    } else if (p0 < p1) {
        float* o = dW + 4 * i;
        atomicAdd(o, s.x); atomicAdd(o + 1, s.y); atomicAdd(o + 2, s.z); atomicAdd(o + 3, s.w);
    }
}
static void w3_reduce_launch(const float* parts, int nparts, size_t n, float* dW, hipStream_t st) {
    const unsigned gx = (unsigned)((n / 4 + 255) / 256);
    static const int target = getenv("SEHIP_W3_REDUCE_WGS") ? atoi(getenv("SEHIP_W3_REDUCE_WGS")) : 1;
    int gy = (int)(target / gx);
    if (gy > nparts / 8) gy = nparts / 8;              // at least eight arrays per slice
    if (gy < 1) gy = 1;
    w3_reduce_kernel<<<dim3(gx, (unsigned)gy), 256, 0, st>>>(parts, nparts, n / 4, dW);
}

// Scratch for the splits' partial arrays: one per stream that launches weight gradients (launches on one stream are ordered; two
// streams must not share).  Allocated on first use, never inside a stream capture (the caller then takes the atomic flush).
struct W3Scratch { hipStream_t st; float* p; size_t bytes; bool pinned; };
static W3Scratch w3_pool[16];
// A slot handed out while its stream is capturing is baked into a hipGraph: it is PINNED from then on -- never freed, never grown
// (a recycle or a grow would leave the replays writing partial sums into freed memory).  A request a pinned slot cannot
// serve gets nullptr, i.e. the caller's atomic flush, as any request inside a capture that finds no array does.
static float* w3_scratch_for(hipStream_t st, size_t bytes) {
    static const bool off = getenv("SEHIP_W3_ATOMIC_FLUSH") != nullptr;
    if (off) return nullptr;
    W3Scratch* e = nullptr;
    for (auto& q : w3_pool)
        if (q.p && q.st == st) { e = &q; break; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    if (e && e->bytes >= bytes) {
        if (capturing) e->pinned = true;
        return e->p;
    }
    if (capturing || (e && e->pinned)) return nullptr;
    if (!e)
        for (auto& q : w3_pool)
            if (!q.p) { e = &q; break; }
    if (!e) {
        // every slot belongs to some stream (workspaces come and go, each with a weight-gradient stream of its own): give the
        // unpinned ones back once the device is idle and start over -- rare, outside any capture
        if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        for (auto& q : w3_pool) {
            if (q.pinned) continue;
            if (q.p) (void)hipFree(q.p);
            q.p = nullptr; q.bytes = 0; q.st = nullptr;
            if (!e) e = &q;
        }
        if (!e) return nullptr;         // sixteen captured streams: atomic flush from here on
    }
    if (e->p) {                       // grow: the old array may still be in use on the stream
        if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
        (void)hipFree(e->p);
        e->p = nullptr;
    }
    float* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    e->st = st; e->p = p; e->bytes = bytes; e->pinned = false;
    return p;
}

// (the same pool for the other weight-gradient kernels that keep partial arrays: csrc/gemm.hip's narrow_wgrad_mfma_kernel)
float* sehip_wgrad_scratch(hipStream_t st, size_t bytes) { return w3_scratch_for(st, bytes); }

template <int NF, int FM, int J, int NB, int NPL = 4>
static size_t w3_lds_bytes() {
    return (size_t)NB * W3Geo<NF, FM, J, NPL>::STAGE + 1024;
}
template <int NF, int FM, int J, int NB, int NPL = 4>
static void w3_set_attr() {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_v3_kernel<NF, FM, J, NB, NPL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
}
template <int NF, int FM, int J>
static void w3_launch(const sehip_gemm_desc& d, int grid, int tiles_per_wg, int splits, int nb, int npl, float* scratch, hipStream_t st) {
    sehip_note_kernel("conv_wgrad_v3_kernel<%d, %d, %d, %d, %d>", NF, FM, J, nb, npl);
    if (npl == 2) {                                    // the half-CU form: one wave per SIMD
        if (nb == 2) {
            w3_set_attr<NF, FM, J, 2, 2>();
            conv_wgrad_v3_kernel<NF, FM, J, 2, 2><<<grid, 256, w3_lds_bytes<NF, FM, J, 2, 2>(), st>>>(d, tiles_per_wg, splits, scratch);
        } else {
            w3_set_attr<NF, FM, J, 3, 2>();
            conv_wgrad_v3_kernel<NF, FM, J, 3, 2><<<grid, 256, w3_lds_bytes<NF, FM, J, 3, 2>(), st>>>(d, tiles_per_wg, splits, scratch);
        }
        return;
    }
#ifdef SEHIP_TOOLS_BUILD      // timing ablations (wrong results): tools builds only
    static const int abl = getenv("SEHIP_W3_ABL") ? atoi(getenv("SEHIP_W3_ABL")) : 0;
    if (abl && NF == 5 && J <= 8) {
        const size_t lds = w3_lds_bytes<NF, FM, J, 3>();
#define W3_ABL(A_) case A_: (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_v3_kernel<NF, FM, J, 3, 4, A_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                            conv_wgrad_v3_kernel<NF, FM, J, 3, 4, A_><<<grid, 512, lds, st>>>(d, tiles_per_wg, splits, scratch); return;
        switch (abl) { W3_ABL(1) W3_ABL(2) W3_ABL(3) W3_ABL(4) W3_ABL(5) W3_ABL(6) W3_ABL(7) default: break; }
#undef W3_ABL
    }
#endif
    if (nb == 2) {
        w3_set_attr<NF, FM, J, 2>();
        conv_wgrad_v3_kernel<NF, FM, J, 2><<<grid, 512, w3_lds_bytes<NF, FM, J, 2>(), st>>>(d, tiles_per_wg, splits, scratch);
    } else {
        w3_set_attr<NF, FM, J, 3>();
        conv_wgrad_v3_kernel<NF, FM, J, 3><<<grid, 512, w3_lds_bytes<NF, FM, J, 3>(), st>>>(d, tiles_per_wg, splits, scratch);
    }
}
// ------------------------------------------------------------------------------------------------------------------------------
// dense_wgrad_kernel: dW[N][K] += dOut[M][N]^T A[M][K] for plain dense products (J = 1, one source whose row m is K contiguous
// elements, one bf16 dOut whose row m is N contiguous elements): the 1x1 convolutions of ConvTasNet (src/model/conv_tasnet.py:307-402),
// N, K in {128, 256}, M = 51 168 rows.  The table-gathered wgrad_kernel walks 64-row slabs with two barriers and a register-staged
// tile each: 52 us per launch for 39 MB of operands.  Here ONE workgroup holds the whole dW (N K = 32 768 accumulators = 64 per
// lane of 8 waves) and streams its share of the rows in 64-row stages: both operands by LDS-DMA into [16 columns][64 rows][32 B]
// planes (conflict-free transposed reads, as conv_wgrad_v3_kernel), one barrier per stage; the flush is the store + reduction of
// conv_wgrad_v3.  Rows past M are beyond num_records: zeros.
template <int N, int K, int NB>
__global__ __launch_bounds__(512, 2) void dense_wgrad_kernel(const sehip_gemm_desc d, int stages_per_wg, float* scratch) {
    constexpr int WN = N >= 256 ? 4 : 2, WK = 8 / WN;               // waves along n / along k
    constexpr int TN = N / WN / 16, TK = K / WK / 16;               // MFMA tiles per wave
    constexpr int GB = N * 64 * 2, XB = K * 64 * 2, STAGE = GB + XB;
    constexpr int GI = GB / 16 / 512, XI = XB / 16 / 512, NI = GI + XI;
    static_assert(GB % (16 * 512) == 0 && XB % (16 * 512) == 0, "whole DMA instructions");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv % WN, wk = wv / WN;
    const int nstages = (d.M + 63) / 64;
    const int s_begin = blockIdx.x * stages_per_wg, s_end = min(nstages, s_begin + stages_per_wg);
    const int ns = s_end - s_begin;
    if (ns <= 0) return;
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_raw*>(reinterpret_cast<const bf16_raw*>(d.dst[0].ptr)), 0, (unsigned)((size_t)d.M * N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_raw*>(reinterpret_cast<const bf16_raw*>(d.src[0].ptr)), 0, (unsigned)((size_t)d.M * K * 2), 0x00020000);
    // piece i = (8 u + wave) * 64 + lane of an image: plane i >> 7, row (i >> 1) & 63, half i & 1  <-  X[m0 + row][16 plane + 8 half]
    unsigned goff[GI], xoff[XI];
#pragma unroll
    for (int u = 0; u < GI; ++u) {
        const int i = (u * 8 + wv) * 64 + lane;
        goff[u] = 2u * (unsigned)(((i >> 1) & 63) * N + (i >> 7) * 16 + (i & 1) * 8);
    }
#pragma unroll
    for (int u = 0; u < XI; ++u) {
        const int i = (u * 8 + wv) * 64 + lane;
        xoff[u] = 2u * (unsigned)(((i >> 1) & 63) * K + (i >> 7) * 16 + (i & 1) * 8);
    }
    auto issue = [&](int stage, int buf) {            // past the end: offsets beyond num_records, zeros (constant DMA count)
        unsigned char* base = smem + buf * STAGE + wv * 1024;
        const unsigned gs = (unsigned)stage * (64u * N * 2u), xs = (unsigned)stage * (64u * K * 2u);
#pragma unroll
        for (int u = 0; u < GI; ++u) {
            const unsigned vo = goff[u] + gs;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (w3_lds_void*)(base + u * 8192), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < XI; ++u) {
            const unsigned vo = xoff[u] + xs;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (w3_lds_void*)(base + GB + u * 8192), 16, vo, 0, 0, 0);
        }
    };
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    const int ga = (wn * TN) * 2048 + (4 * g + q) * 32 + 8 * p4;              // + tn * 2048 + ks * 1024 + h * 512
    const int xa = GB + (wk * TK) * 2048 + (4 * g + q) * 32 + 8 * p4;
    f32x4 acc[TN][TK];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NB - 1; ++s) issue(s_begin + s < s_end ? s_begin + s : nstages, s);
    int buf = 0;
    for (int s = 0; s < ns; ++s) {
        w3_wait_vm<(NB - 2) * NI>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int nx = s_begin + s + NB - 1;
            issue(nx < s_end ? nx : nstages, buf == 0 ? NB - 1 : buf - 1);
        }
        const unsigned char* sb = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 gf[TN], xf[TK];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)(sb + ga + tn * 2048 + ks * 1024));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)(sb + ga + tn * 2048 + ks * 1024 + 512));
                gf[tn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)(sb + xa + tk * 2048 + ks * 1024));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)(sb + xa + tk * 2048 + ks * 1024 + 512));
                xf[tk] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tk = 0; tk < TK; ++tk)
                    acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[tn], xf[tk], acc[tn][tk], 0, 0, 0);
        }
        buf = buf == NB - 1 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* out = scratch ? scratch + (size_t)blockIdx.x * N * K : d.dW;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            const int n = (wn * TN + tn) * 16 + 4 * (lane >> 4);
            const int k = (wk * TK + tk) * 16 + (lane & 15);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (scratch) out[(size_t)(n + u) * K + k] = acc[tn][tk][u];
                else atomicAdd(&out[(size_t)(n + u) * K + k], acc[tn][tk][u]);
            }
        }
}

template <int N, int K>
static int dw_launch(const sehip_gemm_desc& d, hipStream_t st) {
    constexpr int NB = 2;
    constexpr size_t lds = (size_t)NB * (N + K) * 64 * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_wgrad_kernel<N, K, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    static const int env_wgs = getenv("SEHIP_DW_WGS") ? atoi(getenv("SEHIP_DW_WGS")) : 0;
    const int want = env_wgs ? env_wgs : (d.wg_hint > 0 ? d.wg_hint : 128);
    const int nstages = (d.M + 63) / 64;
    int spw = (nstages + want - 1) / want;
    if (spw < 2) spw = 2;
    const int grid = (nstages + spw - 1) / spw;
    const size_t n = (size_t)N * K;
    float* scratch = w3_scratch_for(st, (size_t)grid * n * sizeof(float));
    sehip_note_kernel("dense_wgrad_kernel<%d, %d>", N, K);
    dense_wgrad_kernel<N, K, NB><<<grid, 512, lds, st>>>(d, spw, scratch);
    if (scratch) w3_reduce_launch(scratch, grid, n, d.dW, st);
    return 1;
}

// returns 1 if the kernel was launched, 0 if the descriptor does not qualify (the caller goes on to wgrad_kernel)
int sehip_try_dense_wgrad(const sehip_gemm_desc& d, hipStream_t st) {
    static const bool disabled = getenv("SEHIP_NO_DENSE_WGRAD") != nullptr;
    if (disabled || d.cv_nf > 0 || d.cv2_nkt > 0 || d.J != 1 || d.tmul > 1 || d.dbias) return 0;
    if (d.src[1].ptr || d.dst[1].ptr || d.dst[0].is_f32) return 0;
    if (d.N != d.Npad || d.src[0].C != d.K || d.src[0].F != 1 || d.dst[0].C != d.Npad || d.dst[0].F != 1) return 0;
    if (d.src[0].T != d.TT || d.src[0].tlo != 0 || d.src[0].thi != d.TT || d.dst[0].T != d.TT || d.dst[0].toff || d.dst[0].fadd ||
        d.dst[0].tmul > 1) return 0;
    if ((size_t)d.M * 256 * 2 >= (1ull << 32) - (1u << 20)) return 0;
    if (d.Npad == 256 && d.K == 128) return dw_launch<256, 128>(d, st);
    if (d.Npad == 128 && d.K == 256) return dw_launch<128, 256>(d, st);
    if (d.Npad == 128 && d.K == 128) return dw_launch<128, 128>(d, st);
    return 0;
}

template <int NF, int FM>
static int w3_launch_j(const sehip_gemm_desc& d, hipStream_t st) {
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int TB = 64 / d.J;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * ((d.TT + TB - 1) / TB);
    // SEHIP_W3_NPL: 16-channel planes per workgroup (4: 512 threads, a whole CU's registers; 2: 256 threads, half of them)
    static const int npl = getenv("SEHIP_W3_NPL") ? (atoi(getenv("SEHIP_W3_NPL")) == 2 ? 2 : 4) : 4;
    const int gx = (d.Npad >> 7) * ((C0 + C1) / (16 * npl));
    // m-splits: groups of 8 (one split of each group per XCD), at least four stages each.  A workgroup takes a whole CU (all its
    // VGPRs), and inside the train step the launch runs on the weight-gradient stream beside the dependent chain: alone on the GPU
    // 256 workgroups are fastest (enc4 / enc5: 83 / 80 us against 157 / 161 for conv_wgrad_kernel), in the step about 96 are
    // (B = 32 step, ms: 64: 4.30, 80: 4.23, 96: 4.22-4.25, 112: 4.24-4.27, 128: 4.24, 192: 4.29-4.33, conv_wgrad_kernel: 4.25-4.27)
    // -- more of them take the CUs from the chain's kernels, whose tiles then run in more rounds.  Two stage buffers or three: the same.
    static const int want = getenv("SEHIP_W3_WGS") ? atoi(getenv("SEHIP_W3_WGS")) : (npl == 2 ? 256 : 96);
    static const int nb = getenv("SEHIP_W3_NB") ? atoi(getenv("SEHIP_W3_NB")) : 2;
    int splits = (want / gx + 7) / 8 * 8;
    if (splits < 8) splits = 8;
    while (splits > 8 && MT / splits < 4) splits -= 8;
    const int tiles_per_wg = (MT + splits - 1) / splits;
    const int grid = gx * splits;
    const int used = (MT + tiles_per_wg - 1) / tiles_per_wg;                    // splits with a non-empty range: they store
    const size_t n = (size_t)d.Npad * d.K;
    float* scratch = w3_scratch_for(st, (size_t)used * n * sizeof(float));
    switch (d.J) {
        case 4: w3_launch<NF, FM, 4>(d, grid, tiles_per_wg, splits, nb, npl, scratch, st); break;
        case 8: w3_launch<NF, FM, 8>(d, grid, tiles_per_wg, splits, nb, npl, scratch, st); break;
        case 16: w3_launch<NF, FM, 16>(d, grid, tiles_per_wg, splits, nb, npl, scratch, st); break;
        default: return 0;
    }
    if (scratch) w3_reduce_launch(scratch, used, n, d.dW, st);
    return 1;
}

// returns 1 if the kernel was launched, 0 if the descriptor does not qualify (the caller falls back to conv_wgrad_kernel)
int sehip_try_conv_wgrad_v3(const sehip_gemm_desc& d, hipStream_t st) {
    static const bool disabled = getenv("SEHIP_NO_WGRAD_V3") != nullptr || getenv("SEHIP_NO_PATCH") != nullptr;
    if (disabled || d.cv_nf <= 0 || d.tmul > 1) return 0;
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    if ((C0 & 63) || (C1 & 63) || (d.Npad & 127)) return 0;
    if (d.J != 4 && d.J != 8 && d.J != 16) return 0;
    if (d.K != 2 * d.cv_nf * (C0 + C1)) return 0;
    if (d.dst[1].ptr || d.dst[0].is_f32 || d.dst[0].tmul > 1 || (d.dst[0].C & 7)) return 0;
    if (d.M % (d.TT * d.J)) return 0;
    const int B = d.M / (d.TT * d.J);
    for (int s = 0; s < 2; ++s) {
        if (!d.src[s].ptr) continue;
        for (int kt = 0; kt < 2; ++kt)
            if (d.cv_toff[s][kt] < -1 || d.cv_toff[s][kt] > 1) return 0;
        if (abs(d.cv_toff[s][0] - d.cv_toff[s][1]) > 1) return 0;
        if ((long)B * d.src[s].T * d.src[s].F * d.src[s].C >= (1L << 30) - (1L << 20)) return 0;       // byte offsets below W3_RECORDS
    }
    if ((long)B * d.dst[0].T * d.dst[0].F * d.dst[0].C >= (1L << 30) - (1L << 20)) return 0;
    // Which layers: bit 0 the 5-tap stride-2 encoder layers, bit 1 / 2 the 3- / 2-tap decoder products.  Default: the encoder only.
    // The decoder products are faster alone as well (dec1: 85 / 72 us against 156 / 131) but their launches run beside the decoder
    // and LSTM part of the chain, where conv_wgrad_kernel's half-CU workgroups share CUs with the chain's kernels: the step was
    // 4.30-4.31 ms with them against 4.25 (SEHIP_W3_CLASSES=7 to measure again)
    static const int classes = getenv("SEHIP_W3_CLASSES") ? atoi(getenv("SEHIP_W3_CLASSES")) : 1;
    // SEHIP_W3_DEC_J=<sum of J>: the decoder products (3- / 2-tap) of the layers with these rows per frame only (16: decoder 2)
    static const int dec_j = getenv("SEHIP_W3_DEC_J") ? atoi(getenv("SEHIP_W3_DEC_J")) : 0;
    const bool dec_ok = dec_j == 0 || (dec_j & d.J);
    if (d.cv_nf == 5 && d.fmul == 2 && (classes & 1)) return w3_launch_j<5, 2>(d, st);
    if (d.cv_nf == 3 && d.fmul == 1 && (classes & 2) && dec_ok) return w3_launch_j<3, 1>(d, st);
    if (d.cv_nf == 2 && d.fmul == 1 && (classes & 4) && dec_ok) return w3_launch_j<2, 1>(d, st);
    return 0;
}

template <int NF, int FM>
static void w3_init_nf() {
    w3_set_attr<NF, FM, 4, 2>(); w3_set_attr<NF, FM, 8, 2>(); w3_set_attr<NF, FM, 16, 2>();
    w3_set_attr<NF, FM, 4, 3>(); w3_set_attr<NF, FM, 8, 3>(); w3_set_attr<NF, FM, 16, 3>();
    w3_set_attr<NF, FM, 4, 2, 2>(); w3_set_attr<NF, FM, 8, 2, 2>(); w3_set_attr<NF, FM, 16, 2, 2>();
    w3_set_attr<NF, FM, 4, 3, 2>(); w3_set_attr<NF, FM, 8, 3, 2>(); w3_set_attr<NF, FM, 16, 3, 2>();
}
void sehip_wgrad3_init(void) {
    w3_init_nf<5, 2>(); w3_init_nf<3, 1>(); w3_init_nf<2, 1>();
}
