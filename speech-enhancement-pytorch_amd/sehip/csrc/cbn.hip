// ComplexBatchNorm (2x2 whitening) + single-slope PReLU, forward and backward, on channels-last bf16
// activations [rows][C], C = 2*Cr (real half | imag half).
// Reference: src/model/dccrn.py:457-634 (ComplexBatchNorm; training branch :549-611, whitening :593-602,
// running-stat lerp :555-556,577-579) and nn.PReLU() at :79,122.  The reference spends ~15 elementwise passes and
// 10 chained .mean() reductions per call; here the layer is
//   forward : stats (1 read)  -> per-channel finalize -> apply+PReLU (1 read, 1 write)
//   backward: reduce (2 reads) -> per-channel finalize -> apply (2 reads, 1 write)
// all HBM-bound with 16-byte loads: one thread owns 8 complex channels (a real chunk and its imag chunk).
//
// Backward derivation (c = y - M, xh = U c, o = Wm xh + B, z = prelu(o)):
//   d = dz * (o > 0 ? 1 : a);  da = sum dz*o*[o<=0];  dB = sum d;  dWm from  P = Q U  (Q = sum d c^T)
//   dxh = Wm d;  dU from H = Wm Q;  dV by reverse mode through the closed-form inverse square root;
//   dy = A d + E c - A mean(d),  A = U Wm,  E = (1/N) [[2 dVrr, dVri],[dVri, 2 dVii]].
#include <stdlib.h>
#include "common.h"

#define COEF_STRIDE 16  // floats per channel in the coefficient records

// fwd coef record: 0 Zrr 1 Zri 2 Zir 3 Zii 4 Mr 5 Mi 6 Br 7 Bi 8 Urr 9 Uri 10 Uii 11 vrr(+eps) 12 vri 13 vii(+eps)
// bwd coef record: 0 Arr 1 Ari 2 Air 3 Aii 4 Err 5 Eri 6 Eii 7 kr 8 ki

// A thread's slice of a row: CH complex channels = one 2*CH-byte load from the real half and one from the imaginary half.
// CH = 8 (16-byte loads) keeps 8 channels' coefficient records in registers, CH = 4 (8-byte loads) half of them: the
// backward passes hold 17 floats per channel, and at 8 channels that is one wave per SIMD -- too few loads in flight.
template <int CH> struct Raw;
template <> struct Raw<8> { typedef uint4 type; };
template <> struct Raw<4> { typedef uint2 type; };
template <int CH> struct Chunk { float v[CH]; };
__device__ __forceinline__ Chunk<8> unpack(uint4 u) {
    Chunk<8> c;
    c.v[0] = bf2f((bf16_raw)(u.x & 0xffff)); c.v[1] = bf2f((bf16_raw)(u.x >> 16));
    c.v[2] = bf2f((bf16_raw)(u.y & 0xffff)); c.v[3] = bf2f((bf16_raw)(u.y >> 16));
    c.v[4] = bf2f((bf16_raw)(u.z & 0xffff)); c.v[5] = bf2f((bf16_raw)(u.z >> 16));
    c.v[6] = bf2f((bf16_raw)(u.w & 0xffff)); c.v[7] = bf2f((bf16_raw)(u.w >> 16));
    return c;
}
__device__ __forceinline__ Chunk<4> unpack(uint2 u) {
    Chunk<4> c;
    c.v[0] = bf2f((bf16_raw)(u.x & 0xffff)); c.v[1] = bf2f((bf16_raw)(u.x >> 16));
    c.v[2] = bf2f((bf16_raw)(u.y & 0xffff)); c.v[3] = bf2f((bf16_raw)(u.y >> 16));
    return c;
}
__device__ __forceinline__ void pack_store(bf16_raw* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}
__device__ __forceinline__ void pack_store(bf16_raw* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
}

// Reduce NS per-thread sums (for 8 channels each) over the threads of the block that share a chunk column q and
// store the block's partial sums at part[blockIdx.x][a*Cr + channel] (no atomics: a later per-channel wave adds the
// partials of all blocks in double precision).
// nrep > 0: `part` holds nrep zeroed rows instead and the block ADDS its sums to row blockIdx.x % nrep with atomics (so few rows
// that the pass that needs the totals can add them up itself: cbn_bwd_apply_fin_kernel).
template <int NS, int CH>
__device__ __forceinline__ void block_partials(float (&s)[NS][CH], int nq, int Cr, float* __restrict__ part, int stride,
                                               float* lds /* [4][NS*CH][nq] floats */, int nrep = 0) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* out = part + (size_t)(nrep > 0 ? blockIdx.x % nrep : blockIdx.x) * stride;
    // fold the lanes of a wave that share a chunk column (nq divides 64), park the wave partials in LDS, ONE barrier,
    // then NS*CH*nq threads add the four waves
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float v = s[a][j];
            for (int o = 32; o >= nq; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane < nq) lds[(w * NS * CH + a * CH + j) * nq + lane] = v;
        }
    __syncthreads();
    const int n = NS * CH * nq;
    for (int i = tid; i < n; i += 256) {
        const float t = lds[i] + lds[n + i] + lds[2 * n + i] + lds[3 * n + i];
        const int aj = i / nq, q = i - aj * nq;
        if (nrep > 0) atomicAdd(&out[(size_t)(aj / CH) * Cr + q * CH + (aj % CH)], t);
        else out[(size_t)(aj / CH) * Cr + q * CH + (aj % CH)] = t;
    }
}

// sums of part[b][idx0 + k*Cr] (k < NS) over the blocks, by one wave.  The partials were written by other CUs moments
// ago, so every load is a round trip to memory: ALL of a lane's loads (8 blocks x NS values; nblk <= 512) are issued before
// the first add -- a trip-by-trip loop took 8 serial round trips, 14 of this kernel's 19 us, on the dependent chain.
#define CBN_MAX_BLOCKS 512
template <int NS, int MAXB = CBN_MAX_BLOCKS>
__device__ __forceinline__ void wave_reduce_partials(const float* __restrict__ part, int nblk, int stride, int idx0, int Cr,
                                                     double (&out)[NS]) {
    float v[MAXB / 64][NS];
#pragma unroll
    for (int t = 0; t < MAXB / 64; ++t) {
        const int b = (threadIdx.x & 63) + 64 * t;
        const float* p = part + (size_t)(b < nblk ? b : 0) * stride + idx0;
#pragma unroll
        for (int k = 0; k < NS; ++k) v[t][k] = p[(size_t)k * Cr];  // unconditional (block 0 for b >= nblk): a predicated
    }                                                                 // load is followed by vmcnt(0) and serialises them all
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
#pragma unroll
    for (int t = 0; t < MAXB / 64; ++t) {
        const bool live = (int)(threadIdx.x & 63) + 64 * t < nblk;
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] += live ? (double)v[t][k] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] = wave_sum_d(acc[k]);
}

// ---------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(256) void cbn_stats_kernel(const bf16_raw* __restrict__ y, long rows, int Cr,
                                                        float* __restrict__ part /* [nblk][5*Cr] */) {
    typedef typename Raw<CH>::type raw_t;
    __shared__ float lds[4 * 5 * 8 * 32];
    const int nq = Cr / CH;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    float s[5][CH];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int j = 0; j < CH; ++j) s[a][j] = 0.f;
    const int C = 2 * Cr;
    auto add_row = [&](const raw_t& ua, const raw_t& ub) {
        const Chunk<CH> a = unpack(ua), b = unpack(ub);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            s[0][j] += a.v[j]; s[1][j] += b.v[j];
            s[2][j] += a.v[j] * a.v[j]; s[3][j] += a.v[j] * b.v[j]; s[4][j] += b.v[j] * b.v[j];
        }
    };
    // four rows per trip, no predicate inside the trip: 8 independent loads in flight per thread
    const long stride = (long)gridDim.x * rpb;
    const bf16_raw* p = y + q * CH;
    long r = (long)blockIdx.x * rpb + rl;
    for (; r + 3 * stride < rows; r += 4 * stride) {
        raw_t ua[4], ub[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ua[u] = *reinterpret_cast<const raw_t*>(p + (r + u * stride) * C);
            ub[u] = *reinterpret_cast<const raw_t*>(p + (r + u * stride) * C + Cr);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add_row(ua[u], ub[u]);
    }
    for (; r < rows; r += stride)
        add_row(*reinterpret_cast<const raw_t*>(p + r * C), *reinterpret_cast<const raw_t*>(p + r * C + Cr));
    block_partials<5, CH>(s, nq, Cr, part, 5 * Cr, lds);
}

// the forward coefficient record of one channel from its moments (covariance WITHOUT eps) and affine parameters.
// eps < 0 selects the REAL BatchNorm2d of DCCRN(use_cbn=False) (src/model/dccrn.py:110-113, :130-133: nn.BatchNorm2d over the
// [real half | imaginary half] channels) with |eps|: the two halves of a complex channel are normalised independently -- exactly this
// record with the cross covariance taken as zero (then U = diag(Vrr + eps, Vii + eps)^(-1/2)), weights (Wrr, 0, Wii) = the layer's
// weight halves and biases (Br, Bi) = its bias halves.  Field 14 tells the backward pass (cbn_bwd_record) which kind it is.
__device__ __forceinline__ void cbn_fwd_record(float mr, float mi, float vrr, float vri, float vii, float eps, float wrr, float wri,
                                               float wii, float br_, float bi_, float (&o)[15]) {
    const bool real = eps < 0.f;
    if (real) { eps = -eps; vri = 0.f; wri = 0.f; }
    vrr += eps; vii += eps;
    const float tau = vrr + vii;
    const float delta = vrr * vii - vri * vri;
    const float s = sqrtf(delta);
    const float t = sqrtf(tau + 2.f * s);
    const float rst = 1.f / (s * t);
    const float urr = (s + vii) * rst, uii = (s + vrr) * rst, uri = -vri * rst;
    o[0] = wrr * urr + wri * uri;
    o[1] = wrr * uri + wri * uii;
    o[2] = wri * urr + wii * uri;
    o[3] = wri * uri + wii * uii;
    o[4] = mr; o[5] = mi; o[6] = br_; o[7] = bi_;
    o[8] = urr; o[9] = uri; o[10] = uii; o[11] = vrr; o[12] = vri; o[13] = vii;
    o[14] = real ? 1.f : 0.f;
}
// running-variance factor: nn.BatchNorm2d keeps the UNBIASED batch variance (n / (n - 1)), the reference's ComplexBatchNorm the biased one
__device__ __forceinline__ float cbn_unbias(float eps, long rows) { return eps < 0.f && rows > 1 ? (float)((double)rows / (double)(rows - 1)) : 1.f; }

// one wave per complex channel
__global__ void cbn_finalize_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ Wrr, const float* __restrict__ Wri,
                                    const float* __restrict__ Wii, const float* __restrict__ Br, const float* __restrict__ Bi,
                                    float* __restrict__ RMr, float* __restrict__ RMi, float* __restrict__ RVrr,
                                    float* __restrict__ RVri, float* __restrict__ RVii, long* __restrict__ nbt, long rows,
                                    int Cr, float eps, float momentum, int training, float* __restrict__ coef) {
    const int c = blockIdx.x;
    // requested before the reduction (one memory round trip less on the chain)
    const float wrr = Wrr[c], wri = Wri[c], wii = Wii[c], br_ = Br[c], bi_ = Bi[c];
    const float rmr = RMr[c], rmi = RMi[c], rvrr = RVrr[c], rvri = RVri[c], rvii = RVii[c];
    float mr, mi, vrr, vri, vii;
    if (training) {
        const double n = (double)rows;
        double a[5];
        wave_reduce_partials<5>(part, nblk, 5 * Cr, c, Cr, a);
        const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4];
        if (threadIdx.x != 0) return;
        const double dmr = a0 / n, dmi = a1 / n;
        mr = (float)dmr; mi = (float)dmi;
        vrr = (float)(a2 / n - dmr * dmr);
        vri = (float)(a3 / n - dmr * dmi);
        vii = (float)(a4 / n - dmi * dmi);
        RMr[c] = rmr + momentum * (mr - rmr);
        RMi[c] = rmi + momentum * (mi - rmi);
        const float unb = cbn_unbias(eps, rows);
        RVrr[c] = rvrr + momentum * (vrr * unb - rvrr);
        RVri[c] = rvri + momentum * (vri - rvri);
        RVii[c] = rvii + momentum * (vii * unb - rvii);
        if (c == 0 && nbt) nbt[0] += 1;
    } else {
        if (threadIdx.x != 0) return;
        mr = rmr; mi = rmi; vrr = rvrr; vri = rvri; vii = rvii;
    }
    float rec[15];
    cbn_fwd_record(mr, mi, vrr, vri, vii, eps, wrr, wri, wii, br_, bi_, rec);
    float* o = coef + (size_t)c * COEF_STRIDE;
#pragma unroll
    for (int i = 0; i < 15; ++i) o[i] = rec[i];
}

// The per-channel coefficient records reach the threads through LDS: every thread needs the records of its CH complex
// channels, and fetching them straight from memory (16-40 scattered 16-byte loads per thread, 32 cache lines per wave
// instruction on the wide layers) cost more address traffic than the rows the thread then streams.  One coalesced pass
// puts field f of channel q*CH+j at slot [f][j*nq + q]: the lanes of a wave read consecutive slots (no bank conflict).
template <int CH>
__device__ __forceinline__ void stage_coef(const float* __restrict__ rec, int Cr, int nfield, float4* __restrict__ slot) {
    const int nq = Cr / CH;
    for (int i = threadIdx.x; i < Cr * nfield; i += 256) {
        const int c = i / nfield, f = i - c * nfield;
        slot[f * Cr + (c % CH) * nq + (c / CH)] = reinterpret_cast<const float4*>(rec)[c * (COEF_STRIDE / 4) + f];
    }
}

template <int U, int CH>
__global__ __launch_bounds__(256, 2) void cbn_apply_kernel(const bf16_raw* __restrict__ y, const float* __restrict__ coef,
                                                           const float* __restrict__ slope, long rows, int Cr,
                                                           bf16_raw* __restrict__ z) {
    typedef typename Raw<CH>::type raw_t;
    __shared__ float4 cl[2 * 256];
    // a thread owns ONE chunk of CH complex channels for all its rows: their coefficient floats stay in registers
    const int nq = Cr / CH;
    const int C = 2 * Cr;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const float a = slope[0];
    stage_coef<CH>(coef, Cr, 2, cl);
    __syncthreads();
    float4 zc[CH], mb[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { zc[j] = cl[j * nq + q]; mb[j] = cl[Cr + j * nq + q]; }
    auto row = [&](const raw_t& ur, const raw_t& ui, long r) {
        const Chunk<CH> xr = unpack(ur), xi = unpack(ui);
        float orr[CH], oii[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float cr = xr.v[j] - mb[j].x, ci = xi.v[j] - mb[j].y;
            const float vr = zc[j].x * cr + zc[j].y * ci + mb[j].z;
            const float vi = zc[j].z * cr + zc[j].w * ci + mb[j].w;
            orr[j] = vr > 0.f ? vr : a * vr;
            oii[j] = vi > 0.f ? vi : a * vi;
        }
        pack_store(z + r * C + q * CH, orr);
        pack_store(z + r * C + Cr + q * CH, oii);
    };
    const long stride = (long)gridDim.x * rpb;
    const bf16_raw* p = y + q * CH;
    long r = (long)blockIdx.x * rpb + rl;
    for (; r + (U - 1) * stride < rows; r += U * stride) {
        raw_t ur[U], ui[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ur[u] = *reinterpret_cast<const raw_t*>(p + (r + u * stride) * C);
            ui[u] = *reinterpret_cast<const raw_t*>(p + (r + u * stride) * C + Cr);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) row(ur[u], ui[u], r + u * stride);
    }
    if (U > 1)
        for (; r < rows; r += stride)
            row(*reinterpret_cast<const raw_t*>(p + r * C), *reinterpret_cast<const raw_t*>(p + r * C + Cr), r);
}

// cbn_finalize_kernel + cbn_apply_kernel in one launch, for the layers whose sums arrive as a few replica rows (the convolution
// epilogues' `stats`: nblk = 8): every workgroup derives the coefficient records of all Cr channels itself (thread c: 5 nblk loads
// that hit L2, the moments in double precision, the closed-form inverse square root) straight into the LDS slots the streaming
// loop reads them from; workgroup 0 also writes the records for the backward pass and moves the running statistics.  Saves the
// 5-6 us finalize launch and a kernel boundary per layer on the dependent chain for ~1 us more per apply workgroup.
template <int U, int CH>
__global__ __launch_bounds__(256, 2) void cbn_apply_fin_kernel(const bf16_raw* __restrict__ y, const float* __restrict__ part, int nblk,
                                                               const float* __restrict__ Wrr, const float* __restrict__ Wri,
                                                               const float* __restrict__ Wii, const float* __restrict__ Br,
                                                               const float* __restrict__ Bi, float* __restrict__ RMr, float* __restrict__ RMi,
                                                               float* __restrict__ RVrr, float* __restrict__ RVri, float* __restrict__ RVii,
                                                               long* __restrict__ nbt, long rows, int Cr, float eps, float momentum,
                                                               int training, float* __restrict__ coef, const float* __restrict__ slope,
                                                               bf16_raw* __restrict__ z) {
    typedef typename Raw<CH>::type raw_t;
    __shared__ float4 cl[2 * 256];
    const int nq = Cr / CH;
    const int C = 2 * Cr;
    for (int c = threadIdx.x; c < Cr; c += 256) {
        const float wrr = Wrr[c], wri = Wri[c], wii = Wii[c], br_ = Br[c], bi_ = Bi[c];
        const float rmr = RMr[c], rmi = RMi[c], rvrr = RVrr[c], rvri = RVri[c], rvii = RVii[c];
        float mr = rmr, mi = rmi, vrr = rvrr, vri = rvri, vii = rvii;
        if (training) {
            double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            for (int r0 = 0; r0 < nblk; r0 += 8) {               // 40 independent loads per trip
                float v[8][5];
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int k = 0; k < 5; ++k) v[r][k] = part[(size_t)(r0 + r < nblk ? r0 + r : 0) * 5 * Cr + k * Cr + c];
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int k = 0; k < 5; ++k) a[k] += r0 + r < nblk ? (double)v[r][k] : 0.0;
            }
            const double n = (double)rows;
            const double dmr = a[0] / n, dmi = a[1] / n;
            mr = (float)dmr; mi = (float)dmi;
            vrr = (float)(a[2] / n - dmr * dmr);
            vri = (float)(a[3] / n - dmr * dmi);
            vii = (float)(a[4] / n - dmi * dmi);
            if (blockIdx.x == 0) {
                const float unb = cbn_unbias(eps, rows);
                RMr[c] = rmr + momentum * (mr - rmr);
                RMi[c] = rmi + momentum * (mi - rmi);
                RVrr[c] = rvrr + momentum * (vrr * unb - rvrr);
                RVri[c] = rvri + momentum * (vri - rvri);
                RVii[c] = rvii + momentum * (vii * unb - rvii);
                if (c == 0 && nbt) nbt[0] += 1;
            }
        }
        float rec[15];
        cbn_fwd_record(mr, mi, vrr, vri, vii, eps, wrr, wri, wii, br_, bi_, rec);
        const int slot = (c % CH) * nq + (c / CH);               // the layout stage_coef gives the streaming loop
        cl[slot] = make_float4(rec[0], rec[1], rec[2], rec[3]);
        cl[Cr + slot] = make_float4(rec[4], rec[5], rec[6], rec[7]);
        if (blockIdx.x == 0) {
            float* o = coef + (size_t)c * COEF_STRIDE;
#pragma unroll
            for (int i = 0; i < 15; ++i) o[i] = rec[i];
        }
    }
    __syncthreads();
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const float a = slope[0];
    float4 zc[CH], mb[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { zc[j] = cl[j * nq + q]; mb[j] = cl[Cr + j * nq + q]; }
    auto row = [&](const raw_t& ur, const raw_t& ui, long r) {
        const Chunk<CH> xr = unpack(ur), xi = unpack(ui);
        float orr[CH], oii[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float cr = xr.v[j] - mb[j].x, ci = xi.v[j] - mb[j].y;
            const float vr = zc[j].x * cr + zc[j].y * ci + mb[j].z;
            const float vi = zc[j].z * cr + zc[j].w * ci + mb[j].w;
            orr[j] = vr > 0.f ? vr : a * vr;
            oii[j] = vi > 0.f ? vi : a * vi;
        }
        pack_store(z + r * C + q * CH, orr);
        pack_store(z + r * C + Cr + q * CH, oii);
    };
    const long stride = (long)gridDim.x * rpb;
    const bf16_raw* p = y + q * CH;
    long r = (long)blockIdx.x * rpb + rl;
    for (; r + (U - 1) * stride < rows; r += U * stride) {
        raw_t ur[U], ui[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ur[u] = *reinterpret_cast<const raw_t*>(p + (r + u * stride) * C);
            ui[u] = *reinterpret_cast<const raw_t*>(p + (r + u * stride) * C + Cr);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) row(ur[u], ui[u], r + u * stride);
    }
    if (U > 1)
        for (; r < rows; r += stride)
            row(*reinterpret_cast<const raw_t*>(p + r * C), *reinterpret_cast<const raw_t*>(p + r * C + Cr), r);
}

// ---------------------------------------------------------------------------------------------
// backward pass 1: per-channel sums  0 sum d_r  1 sum d_i  2 Qrr  3 Qri  4 Qir  5 Qii ; slope grad -> acc[6*Cr]
// rows whose stored frame index (row / F) % Tst is < tfirst carry dz == 0 (dropped decoder frame).
// ---------------------------------------------------------------------------------------------
template <int CH, bool HAS2> struct RowIn { typename Raw<CH>::type yr, yi, gr, gi, hr, hi; };

template <int CH, bool HAS2>
__device__ __forceinline__ RowIn<CH, HAS2> load_row(const bf16_raw* __restrict__ y, const bf16_raw* __restrict__ dz,
                                                    const bf16_raw* __restrict__ dz2, long o, int Cr) {
    typedef typename Raw<CH>::type raw_t;
    RowIn<CH, HAS2> v;
    v.yr = *reinterpret_cast<const raw_t*>(y + o);
    v.yi = *reinterpret_cast<const raw_t*>(y + o + Cr);
    v.gr = *reinterpret_cast<const raw_t*>(dz + o);
    v.gi = *reinterpret_cast<const raw_t*>(dz + o + Cr);
    if (HAS2) {
        v.hr = *reinterpret_cast<const raw_t*>(dz2 + o);
        v.hi = *reinterpret_cast<const raw_t*>(dz2 + o + Cr);
    }
    return v;
}

// The finalize step inside the reduce launch (round 6, sehip_cbn_bwd_reduce_fin): with nrep > 0 (block sums added to nrep replica
// rows) and fin.ticket != NULL, the LAST workgroup to finish -- a ticket counter behind a device-scope release of its adds -- derives
// every channel's parameter gradients and apply-pass record from the rows (what cbn_bwd_finalize_kernel does in a launch of its
// own: 11 launches, 0.10 ms of the DCCRN chain for microseconds of work), writes them where the plain apply pass reads them, and
// leaves the rows and the counter zero for the next call.  ONE workgroup finalizes and the coefficients travel through global
// memory as in the three-launch form: the per-workgroup finalize of cbn_bwd_apply_fin_kernel is what produced non-finite rows beside
// the weight-gradient stream.
struct CbnFin {
    const float *Wrr, *Wri, *Wii;
    float *gWrr, *gWri, *gWii, *gBr, *gBi, *gslope, *bcoef;
    unsigned* ticket;
};
__device__ __forceinline__ void cbn_bwd_record(float sdr, float sdi, float qrr, float qri, float qir, float qii, float urr, float uri,
                                               float uii, float vrr, float vri, float vii, float wrr, float wri, float wii, float n,
                                               float (&g5)[5], float (&o)[9], float real = 0.f);

template <int U, int CH, bool HAS2>
__global__ __launch_bounds__(256, 2) void cbn_bwd_reduce_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ dz2,
                                                                const bf16_raw* __restrict__ y, const float* __restrict__ coef,
                                                                const float* __restrict__ slope, long rows, int Cr, int F,
                                                                int Tst, int tfirst, float* __restrict__ part, int nrep, const CbnFin fin) {
    __shared__ float lds[4 * 6 * 8 * 32];
    const int nq = Cr / CH;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const int C = 2 * Cr;
    const float a = slope[0];
    float s[6][CH];
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int j = 0; j < CH; ++j) s[u][j] = 0.f;
    float da = 0.f;
    float4 zc[CH], mb[CH];
    {
        float4* cl = reinterpret_cast<float4*>(lds);   // 2 * Cr float4 <= 8 KB of the 24 KB partial-sum area
        stage_coef<CH>(coef, Cr, 2, cl);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CH; ++j) { zc[j] = cl[j * nq + q]; mb[j] = cl[Cr + j * nq + q]; }
        __syncthreads();
    }
    // rows of a dropped decoder frame are loaded like the others (the buffer is there) and then skipped
    auto row = [&](const RowIn<CH, HAS2>& v, long r) {
        if (tfirst > 0 && (int)(((unsigned)r / (unsigned)F) % (unsigned)Tst) < tfirst) return;
        const Chunk<CH> xr = unpack(v.yr), xi = unpack(v.yi);
        Chunk<CH> gr = unpack(v.gr), gi = unpack(v.gi);
        if (HAS2) {
            const Chunk<CH> hr = unpack(v.hr), hi = unpack(v.hi);
#pragma unroll
            for (int j = 0; j < CH; ++j) { gr.v[j] += hr.v[j]; gi.v[j] += hi.v[j]; }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float cr = xr.v[j] - mb[j].x, ci = xi.v[j] - mb[j].y;
            const float vr = zc[j].x * cr + zc[j].y * ci + mb[j].z;
            const float vi = zc[j].z * cr + zc[j].w * ci + mb[j].w;
            float dr = gr.v[j], di = gi.v[j];
            if (!(vr > 0.f)) { da += dr * vr; dr *= a; }
            if (!(vi > 0.f)) { da += di * vi; di *= a; }
            s[0][j] += dr; s[1][j] += di;
            s[2][j] += dr * cr; s[3][j] += dr * ci; s[4][j] += di * cr; s[5][j] += di * ci;
        }
    };
    const long stride = (long)gridDim.x * rpb;
    long r = (long)blockIdx.x * rpb + rl;
    for (; r + (U - 1) * stride < rows; r += U * stride) {
        RowIn<CH, HAS2> v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load_row<CH, HAS2>(y, dz, dz2, (r + u * stride) * C + q * CH, Cr);
#pragma unroll
        for (int u = 0; u < U; ++u) row(v[u], r + u * stride);
    }
    if (U > 1)
        for (; r < rows; r += stride) row(load_row<CH, HAS2>(y, dz, dz2, r * C + q * CH, Cr), r);
    block_partials<6, CH>(s, nq, Cr, part, 6 * Cr + 1, lds, nrep);
    da = wave_sum(da);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = da;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = lds[0] + lds[1] + lds[2] + lds[3];
        if (nrep > 0) atomicAdd(&part[(size_t)(blockIdx.x % nrep) * (6 * Cr + 1) + 6 * Cr], t);
        else part[(size_t)blockIdx.x * (6 * Cr + 1) + 6 * Cr] = t;
    }
    if (nrep <= 0 || fin.ticket == nullptr) return;
    // ---- last workgroup: finalize.  Release (every thread's adds are performed device-wide before the ticket is drawn), ticket,
    // acquire; the rows are read with device-scope loads (they were written by atomics of every XCD)
    // (cdna_hip_programming.md, slab-reducer recipe: every wave drains its own memory operations, ONE lane per workgroup releases and
    //  draws the ticket -- a device-scope fence per thread is an L2 write-back per wave: +0.58 ms per step when every thread did it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned is_last;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int st = 6 * Cr + 1;
    for (int c = threadIdx.x; c < Cr; c += 256) {
        const float* k = coef + (size_t)c * COEF_STRIDE;
        const float urr = k[8], uri = k[9], uii = k[10], vrr = k[11], vri = k[12], vii = k[13];
        const float wrr = fin.Wrr[c], wri = fin.Wri[c], wii = fin.Wii[c];
        double a6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // (plain loads behind the acquire fence, 48 in flight per trip: one device-scope atomic load per value was 48 dependent round
        //  trips in the one workgroup everybody waits for -- 3.80 ms per step instead of 3.22)
        for (int r0 = 0; r0 < nrep; r0 += 8) {
            float v[8][6];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) v[r][kk] = part[(size_t)(r0 + r < nrep ? r0 + r : 0) * st + kk * Cr + c];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) a6[kk] += r0 + r < nrep ? (double)v[r][kk] : 0.0;
        }
        float g5[5], rec[9];
        cbn_bwd_record((float)a6[0], (float)a6[1], (float)a6[2], (float)a6[3], (float)a6[4], (float)a6[5], urr, uri, uii, vrr, vri, vii,
                       wrr, wri, wii, (float)rows, g5, rec, k[14]);
        fin.gWrr[c] = g5[0]; fin.gWri[c] = g5[1]; fin.gWii[c] = g5[2]; fin.gBr[c] = g5[3]; fin.gBi[c] = g5[4];
        float* o = fin.bcoef + (size_t)c * COEF_STRIDE;
#pragma unroll
        for (int i = 0; i < 9; ++i) o[i] = rec[i];
    }
    if (threadIdx.x == 255) {
        double ds = 0.0;
        for (int r = 0; r < nrep; ++r) ds += (double)part[(size_t)r * st + 6 * Cr];
        fin.gslope[0] = (float)ds;
    }
    __syncthreads();                 // every row has been read: clear them and the counter for the layer's next call
    for (int i = threadIdx.x; i < nrep * st; i += 256) part[i] = 0.f;
    if (threadIdx.x == 0) __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// parameter gradients (g5: dWrr dWri dWii dBr dBi) and the apply pass's record of one channel from its six sums
// real != 0 (field 14 of the forward record): the REAL BatchNorm2d of DCCRN(use_cbn=False) -- the cross covariance is not part of the
// computation (no gradient flows through it: o[5] = 0), the cross weight does not exist (its gradient slot receives 0)
__device__ __forceinline__ void cbn_bwd_record(float sdr, float sdi, float qrr, float qri, float qir, float qii, float urr, float uri,
                                               float uii, float vrr, float vri, float vii, float wrr, float wri, float wii, float n,
                                               float (&g5)[5], float (&o)[9], float real) {
    if (real != 0.f) wri = 0.f;
    // P = Q U  (sum d xh^T)
    const float prr = qrr * urr + qri * uri, pri = qrr * uri + qri * uii;
    const float pir = qir * urr + qii * uri, pii = qir * uri + qii * uii;
    g5[0] = prr; g5[1] = pri + pir; g5[2] = pii; g5[3] = sdr; g5[4] = sdi;
    // H = Wm Q  (sum dxh c^T)
    const float hrr = wrr * qrr + wri * qir, hri = wrr * qri + wri * qii;
    const float hir = wri * qrr + wii * qir, hii = wri * qri + wii * qii;
    const float dUrr = hrr, dUri = hri + hir, dUii = hii;
    // reverse mode through the closed-form V^(-1/2)
    const float tau = vrr + vii;
    const float delta = vrr * vii - vri * vri;
    const float s = sqrtf(delta);
    const float t = sqrtf(tau + 2.f * s);
    const float rst = 1.f / (s * t);
    const float d_rst = dUrr * (s + vii) - dUri * vri + dUii * (s + vrr);
    float d_s = (dUrr + dUii) * rst;
    float d_vii = dUrr * rst, d_vrr = dUii * rst, d_vri = -dUri * rst;
    d_s += d_rst * (-rst / s);
    const float d_t = d_rst * (-rst / t);
    const float d_tau = d_t / (2.f * t);
    d_s += d_t / t;
    const float d_delta = d_s / (2.f * s);
    d_vrr += d_delta * vii + d_tau;
    d_vii += d_delta * vrr + d_tau;
    d_vri += d_delta * (-2.f * vri);
    // A = U Wm
    const float arr = urr * wrr + uri * wri, ari = urr * wri + uri * wii;
    const float air = uri * wrr + uii * wri, aii = uri * wri + uii * wii;
    o[0] = arr; o[1] = ari; o[2] = air; o[3] = aii;
    o[4] = 2.f * d_vrr / n; o[5] = d_vri / n; o[6] = 2.f * d_vii / n;
    o[7] = -(arr * sdr + ari * sdi) / n;
    o[8] = -(air * sdr + aii * sdi) / n;
    if (real != 0.f) { o[5] = 0.f; g5[1] = 0.f; }
}

// one wave per channel: parameter gradients + coefficients of the apply pass
template <int MAXB>
__global__ void cbn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ coef,
                                        const float* __restrict__ Wrr, const float* __restrict__ Wri,
                                        const float* __restrict__ Wii, long rows, int Cr, float* __restrict__ gWrr,
                                        float* __restrict__ gWri, float* __restrict__ gWii, float* __restrict__ gBr,
                                        float* __restrict__ gBi, float* __restrict__ gslope, float* __restrict__ bcoef) {
    const int c = blockIdx.x;
    const int st = 6 * Cr + 1;
    if (c == Cr) {  // extra block: the PReLU slope gradient (its own block, so that channel 0 is not a straggler)
        double ds[1];
        wave_reduce_partials<1, MAXB>(part, nblk, st, 6 * Cr, Cr, ds);
        if (threadIdx.x == 0) gslope[0] = (float)ds[0];
        return;
    }
    // the per-channel constants are requested before the reduction, not after it (one memory round trip less on the chain)
    const float* k = coef + (size_t)c * COEF_STRIDE;
    const float urr = k[8], uri = k[9], uii = k[10], vrr = k[11], vri = k[12], vii = k[13];
    const float wrr = Wrr[c], wri = Wri[c], wii = Wii[c];
    double a[6];
    wave_reduce_partials<6, MAXB>(part, nblk, st, c, Cr, a);
    const float sdr = (float)a[0], sdi = (float)a[1], qrr = (float)a[2], qri = (float)a[3], qir = (float)a[4], qii = (float)a[5];
    if (threadIdx.x != 0) return;
    float g5[5], rec[9];
    cbn_bwd_record(sdr, sdi, qrr, qri, qir, qii, urr, uri, uii, vrr, vri, vii, wrr, wri, wii, (float)rows, g5, rec, k[14]);
    gWrr[c] = g5[0]; gWri[c] = g5[1]; gWii[c] = g5[2]; gBr[c] = g5[3]; gBi[c] = g5[4];
    float* o = bcoef + (size_t)c * COEF_STRIDE;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = rec[i];
}

template <int U, int CH, bool HAS2>
__global__ __launch_bounds__(256, 2) void cbn_bwd_apply_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ dz2,
                                                               const bf16_raw* __restrict__ y, const float* __restrict__ coef,
                                                               const float* __restrict__ bcoef, const float* __restrict__ slope,
                                                               long rows, int Cr, int F, int Tst, int tfirst,
                                                               bf16_raw* __restrict__ dy) {
    __shared__ float4 cl[5 * 256];
    const int nq = Cr / CH;
    const int C = 2 * Cr;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const float a = slope[0];
    // per-channel coefficients of this thread's CH complex channels, in registers for the whole pass
    stage_coef<CH>(coef, Cr, 2, cl);
    stage_coef<CH>(bcoef, Cr, 3, cl + 2 * Cr);
    __syncthreads();
    float4 zc[CH], mb[CH], A[CH], E[CH];
    float ki[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        zc[j] = cl[j * nq + q];
        mb[j] = cl[Cr + j * nq + q];
        A[j] = cl[2 * Cr + j * nq + q];
        E[j] = cl[3 * Cr + j * nq + q];  // Err Eri Eii kr
        ki[j] = cl[4 * Cr + j * nq + q].x;
    }
    auto row = [&](const RowIn<CH, HAS2>& v, long r) {
        const bool dropped = tfirst > 0 && (int)(((unsigned)r / (unsigned)F) % (unsigned)Tst) < tfirst;
        const Chunk<CH> xr = unpack(v.yr), xi = unpack(v.yi);
        Chunk<CH> gr = unpack(v.gr), gi = unpack(v.gi);
        if (HAS2) {
            const Chunk<CH> hr = unpack(v.hr), hi = unpack(v.hi);
#pragma unroll
            for (int j = 0; j < CH; ++j) { gr.v[j] += hr.v[j]; gi.v[j] += hi.v[j]; }
        }
        float orr[CH], oii[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float cr = xr.v[j] - mb[j].x, ci = xi.v[j] - mb[j].y;
            const float vr = zc[j].x * cr + zc[j].y * ci + mb[j].z;
            const float vi = zc[j].z * cr + zc[j].w * ci + mb[j].w;
            float dr = dropped ? 0.f : gr.v[j], di = dropped ? 0.f : gi.v[j];
            if (!(vr > 0.f)) dr *= a;
            if (!(vi > 0.f)) di *= a;
            orr[j] = A[j].x * dr + A[j].y * di + E[j].x * cr + E[j].y * ci + E[j].w;
            oii[j] = A[j].z * dr + A[j].w * di + E[j].y * cr + E[j].z * ci + ki[j];
        }
        pack_store(dy + r * C + q * CH, orr);
        pack_store(dy + r * C + Cr + q * CH, oii);
    };
    const long stride = (long)gridDim.x * rpb;
    long r = (long)blockIdx.x * rpb + rl;
    for (; r + (U - 1) * stride < rows; r += U * stride) {
        RowIn<CH, HAS2> v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load_row<CH, HAS2>(y, dz, dz2, (r + u * stride) * C + q * CH, Cr);
#pragma unroll
        for (int u = 0; u < U; ++u) row(v[u], r + u * stride);
    }
    if (U > 1)
        for (; r < rows; r += stride) row(load_row<CH, HAS2>(y, dz, dz2, r * C + q * CH, Cr), r);
}

// cbn_bwd_finalize_kernel + cbn_bwd_apply_kernel in one launch, for sums that the reduce pass left in a few rows (nrep <= 64, fp32
// atomics: cbn_bwd_reduce_kernel with nrep > 0).  As cbn_apply_fin_kernel does for the forward pass.
template <int U, int CH, bool HAS2>
__global__ __launch_bounds__(256, 2) void cbn_bwd_apply_fin_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ dz2,
                                                               const bf16_raw* __restrict__ y, const float* __restrict__ coef,
                                                               const float* __restrict__ rep /* [nrep][6 Cr + 1] */, int nrep,
                                                               float* __restrict__ rep_next /* the other set of rows: cleared here */,
                                                               const float* __restrict__ Wrr, const float* __restrict__ Wri,
                                                               const float* __restrict__ Wii, float* __restrict__ gWrr,
                                                               float* __restrict__ gWri, float* __restrict__ gWii, float* __restrict__ gBr,
                                                               float* __restrict__ gBi, float* __restrict__ gslope,
                                                               const float* __restrict__ slope, long rows, int Cr, int F, int Tst,
                                                               int tfirst, bf16_raw* __restrict__ dy, int dbg_mode) {
    __shared__ float4 cl[5 * 256];
    const int nq = Cr / CH;
    const int C = 2 * Cr;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const float a = slope[0];
    // per-channel coefficients of this thread's CH complex channels, in registers for the whole pass
    // (the rows the NEXT backward pass of this layer will add to: nobody reads them now, so no memset on the chain)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nrep * (6 * Cr + 1); i += gridDim.x * 256) rep_next[i] = 0.f;
    stage_coef<CH>(coef, Cr, 2, cl);
    // cbn_bwd_finalize_kernel's work, by every workgroup for itself: thread c adds the nrep rows of channel c's six sums (L2 hits) and
    // derives the channel's record straight into the slots the streaming loop reads; workgroup 0 stores the parameter gradients
    for (int c = threadIdx.x; c < Cr; c += 256) {
        const float* k = coef + (size_t)c * COEF_STRIDE;
        auto ldg = [&](const float* p_) { return (dbg_mode & 2) ? __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p_; };
        const float urr = ldg(k + 8), uri = ldg(k + 9), uii = ldg(k + 10), vrr = ldg(k + 11), vri = ldg(k + 12), vii = ldg(k + 13);
        const float wrr = ldg(Wrr + c), wri = ldg(Wri + c), wii = ldg(Wii + c);
        double a6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int r0 = 0; r0 < nrep; r0 += 8) {                   // 48 independent loads per trip
            float v[8][6];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) {
                    const float* pr = &rep[(size_t)(r0 + r < nrep ? r0 + r : 0) * (6 * Cr + 1) + kk * Cr + c];
                    v[r][kk] = (dbg_mode & 1) ? __hip_atomic_load(pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *pr;
                }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) a6[kk] += r0 + r < nrep ? (double)v[r][kk] : 0.0;
        }
        float g5[5], rec[9];
        cbn_bwd_record((float)a6[0], (float)a6[1], (float)a6[2], (float)a6[3], (float)a6[4], (float)a6[5], urr, uri, uii, vrr, vri, vii,
                       wrr, wri, wii, (float)rows, g5, rec, k[14]);
        const int slot = (c % CH) * nq + (c / CH);
        cl[2 * Cr + slot] = make_float4(rec[0], rec[1], rec[2], rec[3]);
        cl[3 * Cr + slot] = make_float4(rec[4], rec[5], rec[6], rec[7]);
        cl[4 * Cr + slot] = make_float4(rec[8], 0.f, 0.f, 0.f);
        if (blockIdx.x == 0) { gWrr[c] = g5[0]; gWri[c] = g5[1]; gWii[c] = g5[2]; gBr[c] = g5[3]; gBi[c] = g5[4]; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 255) {                 // the PReLU slope gradient
        double ds = 0.0;
        for (int r = 0; r < nrep; ++r) ds += (double)rep[(size_t)r * (6 * Cr + 1) + 6 * Cr];
        gslope[0] = (float)ds;
    }
    __syncthreads();
    float4 zc[CH], mb[CH], A[CH], E[CH];
    float ki[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        zc[j] = cl[j * nq + q];
        mb[j] = cl[Cr + j * nq + q];
        A[j] = cl[2 * Cr + j * nq + q];
        E[j] = cl[3 * Cr + j * nq + q];  // Err Eri Eii kr
        ki[j] = cl[4 * Cr + j * nq + q].x;
    }
    auto row = [&](const RowIn<CH, HAS2>& v, long r) {
        const bool dropped = tfirst > 0 && (int)(((unsigned)r / (unsigned)F) % (unsigned)Tst) < tfirst;
        const Chunk<CH> xr = unpack(v.yr), xi = unpack(v.yi);
        Chunk<CH> gr = unpack(v.gr), gi = unpack(v.gi);
        if (HAS2) {
            const Chunk<CH> hr = unpack(v.hr), hi = unpack(v.hi);
#pragma unroll
            for (int j = 0; j < CH; ++j) { gr.v[j] += hr.v[j]; gi.v[j] += hi.v[j]; }
        }
        float orr[CH], oii[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float cr = xr.v[j] - mb[j].x, ci = xi.v[j] - mb[j].y;
            const float vr = zc[j].x * cr + zc[j].y * ci + mb[j].z;
            const float vi = zc[j].z * cr + zc[j].w * ci + mb[j].w;
            float dr = dropped ? 0.f : gr.v[j], di = dropped ? 0.f : gi.v[j];
            if (!(vr > 0.f)) dr *= a;
            if (!(vi > 0.f)) di *= a;
            orr[j] = A[j].x * dr + A[j].y * di + E[j].x * cr + E[j].y * ci + E[j].w;
            oii[j] = A[j].z * dr + A[j].w * di + E[j].y * cr + E[j].z * ci + ki[j];
        }
        pack_store(dy + r * C + q * CH, orr);
        pack_store(dy + r * C + Cr + q * CH, oii);
    };
    const long stride = (long)gridDim.x * rpb;
    long r = (long)blockIdx.x * rpb + rl;
    for (; r + (U - 1) * stride < rows; r += U * stride) {
        RowIn<CH, HAS2> v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load_row<CH, HAS2>(y, dz, dz2, (r + u * stride) * C + q * CH, Cr);
#pragma unroll
        for (int u = 0; u < U; ++u) row(v[u], r + u * stride);
    }
    if (U > 1)
        for (; r < rows; r += stride) row(load_row<CH, HAS2>(y, dz, dz2, r * C + q * CH, Cr), r);
}

// ---------------------------------------------------------------------------------------------
static int check_cbn(const char* who, long rows, int Cr) {
    SEHIP_REQUIRE(rows > 0, "%s: empty input", who);
    SEHIP_REQUIRE(Cr >= 8 && Cr <= 256 && (Cr & (Cr - 1)) == 0,
                  "%s: complex channels Cr=%d must be a power of two in [8, 256]", who, Cr);
    return 0;
}
static int grid_for(long work_items) {
    long g = (work_items + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

static int env_int(const char* key, int dflt) {
    const char* v = getenv(key);
    return v ? atoi(v) : dflt;
}

// streaming passes: rows-per-thread target -> grid (the coefficient preamble is paid per workgroup)
// experiment knobs (defaults = the measured best, tools/bench_cbn.py): channels per thread, rows per trip, rows per thread of
// the streaming passes.  Sum over the six C1 layer shapes (us), 8 channels x 1 row (the first version) -> 4 channels x 2 rows:
// stats 110 -> 77, apply 108 -> 108, bwd_reduce 203 -> 123, bwd_apply 174 -> 160 (4.4 TB/s read+write).
static int cbn_ch(int Cr) {
    static const int ch = env_int("SEHIP_CBN_CH", 4);
    return (ch == 8 && Cr >= 8) ? 8 : 4;
}
static int unroll_rows() {
    static const int u = env_int("SEHIP_CBN_U", 2);
    return u == 1 || u == 4 ? u : 2;
}
// The backward passes run beside the weight-gradient stream, whose workgroups (conv_wgrad_kernel: 171-204 registers x 8 waves) stay
// resident for 100-200 us: a backward pass with two rows per trip needs 146 / 194 registers and finds no room on those CUs -- it waits
// for them (round 5: one apply pass took 84 us beside conv_wgrad_kernel<2>, 12 us alone).  One row per trip: 92 / 118 registers, the
// passes share the CUs (B = 32 step 3.236 -> 3.216 ms, same box; the forward passes, which run alone, keep two rows).
static int unroll_rows_bwd() {
    static const int u = env_int("SEHIP_CBN_BU", 1);
    return u == 2 || u == 4 ? u : 1;
}
// streaming passes: rows-per-thread target -> grid (the coefficient preamble is paid per workgroup)
// Cap: 768 since the forward apply pass also finalizes the sums (every workgroup pays ~1 us for the records of all channels): B = 32
// step, ms, two runs each: 2048: 4.157 / 4.184, 1536: 4.167 / 4.178, 1024: 4.146 / 4.158, 768: 4.138 / 4.149, 512: 4.142 / 4.164,
// 384: 4.175 / 4.178 (2048 was the optimum of the three-launch version)
static int apply_blocks(long rows, int Cr, bool bwd = false) {
    static const int rpt = env_int("SEHIP_CBN_APPLY_ROWS", 8);
    static const int capf = env_int("SEHIP_CBN_APPLY_BLOCKS", 768), capb = env_int("SEHIP_CBN_BAPPLY_BLOCKS", 768);
    const int cap = bwd ? capb : capf;
    const int rpb = 256 / (Cr / cbn_ch(Cr));
    long g = (rows + (long)rpb * rpt - 1) / ((long)rpb * rpt);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

static int stat_blocks(long rows, int Cr) {
    const int rpb = 256 / (Cr / cbn_ch(Cr));
    long g = (rows + (long)rpb * 8 - 1) / ((long)rpb * 8);
    // two workgroups per CU: with 4 channels per thread the reductions run 3-5 waves per SIMD, and the second workgroup's
    // loads cover the first one's arithmetic (tools/bench_cbn.py, sum over the six C1 shapes: bwd_reduce 158 us at 256
    // blocks, 139 at 384, 123 at 512); the finalize kernels read twice the partials for it (+1 us each)
    static const int cap = env_int("SEHIP_CBN_BLOCKS", 512);
    if (g > cap) g = cap;
    if (g > CBN_MAX_BLOCKS) g = CBN_MAX_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

// number of floats the partial-sum scratch must hold (forward stats and backward reduce share it)
extern "C" long sehip_cbn_scratch_floats(long rows, int Cr) { return (long)stat_blocks(rows, Cr) * (6L * Cr + 1); }

extern "C" int sehip_cbn_stats(const void* y, long rows, int Cr, float* part, void* stream) {
    if (int e = check_cbn("cbn_stats", rows, Cr)) return e;
    if (cbn_ch(Cr) == 8)
        cbn_stats_kernel<8><<<stat_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, rows, Cr, part);
    else
        cbn_stats_kernel<4><<<stat_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, rows, Cr, part);
    SEHIP_CHECK_LAUNCH("cbn_stats");
    return 0;
}

// params: Wrr,Wri,Wii,Br,Bi [Cr] fp32; buffers RMr,RMi,RVrr,RVri,RVii [Cr] fp32 (updated in training), nbt int64[1]
static int cbn_finalize_launch(const float* part, int nblk, const float* Wrr, const float* Wri, const float* Wii, const float* Br,
                               const float* Bi, float* RMr, float* RMi, float* RVrr, float* RVri, float* RVii, long* nbt, long rows,
                               int Cr, float eps, float momentum, int training, float* coef, void* stream) {
    if (int e = check_cbn("cbn_finalize", rows, Cr)) return e;
    SEHIP_REQUIRE(nblk >= 1 && nblk <= CBN_MAX_BLOCKS, "cbn_finalize: %d partial rows (1..%d)", nblk, CBN_MAX_BLOCKS);
    cbn_finalize_kernel<<<Cr, 64, 0, (hipStream_t)stream>>>(part, nblk, Wrr, Wri, Wii, Br, Bi, RMr, RMi, RVrr, RVri, RVii, nbt, rows,
                                                            Cr, eps, momentum, training, coef);
    SEHIP_CHECK_LAUNCH("cbn_finalize");
    return 0;
}

extern "C" int sehip_cbn_finalize(const float* part, const float* Wrr, const float* Wri, const float* Wii, const float* Br,
                                  const float* Bi, float* RMr, float* RMi, float* RVrr, float* RVri, float* RVii, long* nbt,
                                  long rows, int Cr, float eps, float momentum, int training, float* coef, void* stream) {
    return cbn_finalize_launch(part, stat_blocks(rows, Cr), Wrr, Wri, Wii, Br, Bi, RMr, RMi, RVrr, RVri, RVii, nbt, rows, Cr, eps,
                               momentum, training, coef, stream);
}

// the same from nblk rows of sums [nblk][5][Cr] that somebody else accumulated (the convolution kernel's `stats` field: 8 replicas)
extern "C" int sehip_cbn_finalize_n(const float* part, int nblk, const float* Wrr, const float* Wri, const float* Wii, const float* Br,
                                    const float* Bi, float* RMr, float* RMi, float* RVrr, float* RVri, float* RVii, long* nbt,
                                    long rows, int Cr, float eps, float momentum, int training, float* coef, void* stream) {
    return cbn_finalize_launch(part, nblk, Wrr, Wri, Wii, Br, Bi, RMr, RMi, RVrr, RVri, RVii, nbt, rows, Cr, eps, momentum, training,
                               coef, stream);
}

extern "C" int sehip_cbn_apply(const void* y, const float* coef, const float* slope, long rows, int Cr, void* z, void* stream) {
    if (int e = check_cbn("cbn_apply", rows, Cr)) return e;
#define CBN_APPLY(U, CH) cbn_apply_kernel<U, CH><<<apply_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, coef, slope, rows, Cr, (bf16_raw*)z)
    switch (unroll_rows() * 16 + cbn_ch(Cr)) {
        case 1 * 16 + 8: CBN_APPLY(1, 8); break;
        case 2 * 16 + 8: CBN_APPLY(2, 8); break;
        case 4 * 16 + 8: CBN_APPLY(4, 8); break;
        case 1 * 16 + 4: CBN_APPLY(1, 4); break;
        case 4 * 16 + 4: CBN_APPLY(4, 4); break;
        default: CBN_APPLY(2, 4); break;
    }
#undef CBN_APPLY
    SEHIP_CHECK_LAUNCH("cbn_apply");
    return 0;
}

// sehip_cbn_finalize_n + sehip_cbn_apply in one launch (cbn_apply_fin_kernel): for sums that arrive as a few rows (nblk <= 64)
extern "C" int sehip_cbn_finalize_apply_n(const void* y, const float* part, int nblk, const float* Wrr, const float* Wri, const float* Wii,
                                          const float* Br, const float* Bi, float* RMr, float* RMi, float* RVrr, float* RVri, float* RVii,
                                          long* nbt, long rows, int Cr, float eps, float momentum, int training, float* coef,
                                          const float* slope, void* z, void* stream) {
    if (int e = check_cbn("cbn_finalize_apply_n", rows, Cr)) return e;
    SEHIP_REQUIRE(nblk >= 1 && nblk <= 64, "cbn_finalize_apply_n: %d rows of sums (1..64)", nblk);
#define CBN_FAPP(U, CH) cbn_apply_fin_kernel<U, CH><<<apply_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>(                 \
        (const bf16_raw*)y, part, nblk, Wrr, Wri, Wii, Br, Bi, RMr, RMi, RVrr, RVri, RVii, nbt, rows, Cr, eps, momentum, training, coef, \
        slope, (bf16_raw*)z)
    switch (unroll_rows() * 16 + cbn_ch(Cr)) {
        case 1 * 16 + 8: CBN_FAPP(1, 8); break;
        case 2 * 16 + 8: CBN_FAPP(2, 8); break;
        case 4 * 16 + 8: CBN_FAPP(4, 8); break;
        case 1 * 16 + 4: CBN_FAPP(1, 4); break;
        case 4 * 16 + 4: CBN_FAPP(4, 4); break;
        default: CBN_FAPP(2, 4); break;
    }
#undef CBN_FAPP
    SEHIP_CHECK_LAUNCH("cbn_finalize_apply_n");
    return 0;
}

static int cbn_bwd_reduce_launch(const void* dz, const void* dz2, const void* y, const float* coef, const float* slope,
                                 long rows, int Cr, int F, int Tst, int tfirst, float* part, int nrep, void* stream,
                                 const CbnFin fin = CbnFin{}) {
    if (int e = check_cbn("cbn_bwd_reduce", rows, Cr)) return e;
    SEHIP_REQUIRE(rows < (1L << 31), "cbn_bwd_reduce: %ld rows exceed the 32-bit frame arithmetic", rows);
#define CBN_RED(U, CH, H2) cbn_bwd_reduce_kernel<U, CH, H2><<<stat_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>( \
        (const bf16_raw*)dz, (const bf16_raw*)dz2, (const bf16_raw*)y, coef, slope, rows, Cr, F, Tst, tfirst, part, nrep, fin)
#define CBN_RED2(U, CH) do { if (dz2) CBN_RED(U, CH, true); else CBN_RED(U, CH, false); } while (0)
    switch (unroll_rows_bwd() * 16 + cbn_ch(Cr)) {
        case 1 * 16 + 8: CBN_RED2(1, 8); break;
        case 2 * 16 + 8: CBN_RED2(2, 8); break;
        case 4 * 16 + 8: CBN_RED2(4, 8); break;
        case 1 * 16 + 4: CBN_RED2(1, 4); break;
        case 4 * 16 + 4: CBN_RED2(4, 4); break;
        default: CBN_RED2(2, 4); break;
    }
#undef CBN_RED2
#undef CBN_RED
    SEHIP_CHECK_LAUNCH("cbn_bwd_reduce");
    return 0;
}
extern "C" int sehip_cbn_bwd_reduce(const void* dz, const void* dz2, const void* y, const float* coef, const float* slope,
                                    long rows, int Cr, int F, int Tst, int tfirst, float* part, void* stream) {
    return cbn_bwd_reduce_launch(dz, dz2, y, coef, slope, rows, Cr, F, Tst, tfirst, part, 0, stream);
}

extern "C" int sehip_cbn_bwd_reduce_fin(const void* dz, const void* dz2, const void* y, const float* coef, const float* Wrr,
                                        const float* Wri, const float* Wii, const float* slope, long rows, int Cr, int F, int Tst,
                                        int tfirst, float* rep, int nrep, unsigned* ticket, float* gWrr, float* gWri, float* gWii,
                                        float* gBr, float* gBi, float* gslope, float* bcoef, void* stream) {
    SEHIP_REQUIRE(nrep >= 1 && nrep <= 64 && rep && ticket && bcoef, "cbn_bwd_reduce_fin: %d rows of sums (1..64), rows / ticket / bcoef", nrep);
    SEHIP_REQUIRE(!sehip_deterministic(), "cbn_bwd_reduce_fin: not part of the deterministic schedule (fp32 atomics): use "
                                          "sehip_cbn_bwd_reduce + sehip_cbn_bwd_finalize");
    CbnFin fin{Wrr, Wri, Wii, gWrr, gWri, gWii, gBr, gBi, gslope, bcoef, ticket};
    return cbn_bwd_reduce_launch(dz, dz2, y, coef, slope, rows, Cr, F, Tst, tfirst, rep, nrep, stream, fin);
}

extern "C" int sehip_cbn_bwd_finalize(const float* part, const float* coef, const float* Wrr, const float* Wri,
                                      const float* Wii, long rows, int Cr, float* gWrr, float* gWri, float* gWii, float* gBr,
                                      float* gBi, float* gslope, float* bcoef, void* stream) {
    if (int e = check_cbn("cbn_bwd_finalize", rows, Cr)) return e;
    cbn_bwd_finalize_kernel<CBN_MAX_BLOCKS><<<Cr + 1, 64, 0, (hipStream_t)stream>>>(part, stat_blocks(rows, Cr), coef, Wrr, Wri, Wii, rows, Cr,
                                                                                gWrr, gWri, gWii, gBr, gBi, gslope, bcoef);
    SEHIP_CHECK_LAUNCH("cbn_bwd_finalize");
    return 0;
}

extern "C" int sehip_cbn_bwd_finalize_n(const float* part, int nblk, const float* coef, const float* Wrr, const float* Wri,
                                        const float* Wii, long rows, int Cr, float* gWrr, float* gWri, float* gWii, float* gBr,
                                        float* gBi, float* gslope, float* bcoef, void* stream) {
    if (int e = check_cbn("cbn_bwd_finalize_n", rows, Cr)) return e;
    SEHIP_REQUIRE(nblk >= 1 && nblk <= 1024, "cbn_bwd_finalize_n: %d rows of sums (1..1024)", nblk);
    if (nblk <= CBN_MAX_BLOCKS)
        cbn_bwd_finalize_kernel<CBN_MAX_BLOCKS><<<Cr + 1, 64, 0, (hipStream_t)stream>>>(part, nblk, coef, Wrr, Wri, Wii, rows, Cr, gWrr, gWri,
                                                                                    gWii, gBr, gBi, gslope, bcoef);
    else
        cbn_bwd_finalize_kernel<1024><<<Cr + 1, 64, 0, (hipStream_t)stream>>>(part, nblk, coef, Wrr, Wri, Wii, rows, Cr, gWrr, gWri, gWii, gBr,
                                                                          gBi, gslope, bcoef);
    SEHIP_CHECK_LAUNCH("cbn_bwd_finalize_n");
    return 0;
}

extern "C" int sehip_cbn_bwd_apply(const void* dz, const void* dz2, const void* y, const float* coef, const float* bcoef,
                                   const float* slope, long rows, int Cr, int F, int Tst, int tfirst, void* dy, void* stream) {
    if (int e = check_cbn("cbn_bwd_apply", rows, Cr)) return e;
    SEHIP_REQUIRE(rows < (1L << 31), "cbn_bwd_apply: %ld rows exceed the 32-bit frame arithmetic", rows);
#define CBN_BAPP(U, CH, H2) cbn_bwd_apply_kernel<U, CH, H2><<<apply_blocks(rows, Cr, true), 256, 0, (hipStream_t)stream>>>( \
        (const bf16_raw*)dz, (const bf16_raw*)dz2, (const bf16_raw*)y, coef, bcoef, slope, rows, Cr, F, Tst, tfirst, (bf16_raw*)dy)
#define CBN_BAPP2(U, CH) do { if (dz2) CBN_BAPP(U, CH, true); else CBN_BAPP(U, CH, false); } while (0)
    switch (unroll_rows_bwd() * 16 + cbn_ch(Cr)) {
        case 1 * 16 + 8: CBN_BAPP2(1, 8); break;
        case 2 * 16 + 8: CBN_BAPP2(2, 8); break;
        case 4 * 16 + 8: CBN_BAPP2(4, 8); break;
        case 1 * 16 + 4: CBN_BAPP2(1, 4); break;
        case 4 * 16 + 4: CBN_BAPP2(4, 4); break;
        default: CBN_BAPP2(2, 4); break;
    }
#undef CBN_BAPP2
#undef CBN_BAPP
    SEHIP_CHECK_LAUNCH("cbn_bwd_apply");
    return 0;
}

// backward pass of the layer in two launches instead of three: the reduce pass adds its block sums to `rep` ([nrep][6 Cr + 1] fp32,
// zero on entry, nrep <= 64), the apply pass finalizes them itself (cbn_bwd_apply_fin_kernel) and clears `rep_next`, the rows the
// caller hands in as `rep` next time (two sets used alternately: no memset on the dependent chain; both zero before the first call)
extern "C" int sehip_cbn_bwd_fused(const void* dz, const void* dz2, const void* y, const float* coef, const float* Wrr, const float* Wri,
                                   const float* Wii, const float* slope, long rows, int Cr, int F, int Tst, int tfirst, float* rep,
                                   float* rep_next, int nrep, float* gWrr, float* gWri, float* gWii, float* gBr, float* gBi, float* gslope, void* dy, void* stream) {
    if (int e = check_cbn("cbn_bwd_fused", rows, Cr)) return e;
    SEHIP_REQUIRE(rows < (1L << 31), "cbn_bwd_fused: %ld rows exceed the 32-bit frame arithmetic", rows);
    SEHIP_REQUIRE(nrep >= 1 && nrep <= 64 && rep != nullptr && rep_next != nullptr && rep_next != rep,
                  "cbn_bwd_fused: %d rows of sums (1..64), two distinct sets", nrep);
    if (int e = cbn_bwd_reduce_launch(dz, dz2, y, coef, slope, rows, Cr, F, Tst, tfirst, rep, nrep, stream)) return e;
    static const int dbg_mode = env_int("SEHIP_BWD_FIN_DBG", 0);
#define CBN_BFIN(U, CH, H2) cbn_bwd_apply_fin_kernel<U, CH, H2><<<apply_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>(              \
        (const bf16_raw*)dz, (const bf16_raw*)dz2, (const bf16_raw*)y, coef, rep, nrep, rep_next, Wrr, Wri, Wii, gWrr, gWri, gWii, gBr, gBi, gslope, \
        slope, rows, Cr, F, Tst, tfirst, (bf16_raw*)dy, dbg_mode)
#define CBN_BFIN2(U, CH) do { if (dz2) CBN_BFIN(U, CH, true); else CBN_BFIN(U, CH, false); } while (0)
    switch (unroll_rows_bwd() * 16 + cbn_ch(Cr)) {
        case 1 * 16 + 8: CBN_BFIN2(1, 8); break;
        case 2 * 16 + 8: CBN_BFIN2(2, 8); break;
        case 4 * 16 + 8: CBN_BFIN2(4, 8); break;
        case 1 * 16 + 4: CBN_BFIN2(1, 4); break;
        case 4 * 16 + 4: CBN_BFIN2(4, 4); break;
        default: CBN_BFIN2(2, 4); break;
    }
#undef CBN_BFIN2
#undef CBN_BFIN
    SEHIP_CHECK_LAUNCH("cbn_bwd_fused");
    return 0;
}
