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

struct Chunk8 { float v[8]; };
__device__ __forceinline__ Chunk8 unpack8(uint4 u) {
    Chunk8 c;
    c.v[0] = bf2f((bf16_raw)(u.x & 0xffff)); c.v[1] = bf2f((bf16_raw)(u.x >> 16));
    c.v[2] = bf2f((bf16_raw)(u.y & 0xffff)); c.v[3] = bf2f((bf16_raw)(u.y >> 16));
    c.v[4] = bf2f((bf16_raw)(u.z & 0xffff)); c.v[5] = bf2f((bf16_raw)(u.z >> 16));
    c.v[6] = bf2f((bf16_raw)(u.w & 0xffff)); c.v[7] = bf2f((bf16_raw)(u.w >> 16));
    return c;
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    return make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}

// Reduce NS per-thread sums (for 8 channels each) over the threads of the block that share a chunk column q and
// store the block's partial sums at part[blockIdx.x][a*Cr + channel] (no atomics: a later per-channel wave adds the
// partials of all blocks in double precision).
template <int NS>
__device__ __forceinline__ void block_partials(float (&s)[NS][8], int nq, int Cr, float* __restrict__ part, int stride,
                                               float* lds /* [4][NS*8][nq] floats */) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* out = part + (size_t)blockIdx.x * stride;
    // fold the lanes of a wave that share a chunk column (nq divides 64), park the wave partials in LDS, ONE barrier,
    // then NS*8*nq threads add the four waves
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = s[a][j];
            for (int o = 32; o >= nq; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane < nq) lds[(w * NS * 8 + a * 8 + j) * nq + lane] = v;
        }
    __syncthreads();
    for (int i = tid; i < NS * 8 * nq; i += 256) {
        const float t = lds[i] + lds[NS * 8 * nq + i] + lds[2 * NS * 8 * nq + i] + lds[3 * NS * 8 * nq + i];
        const int aj = i / nq, q = i - aj * nq;
        out[(size_t)(aj >> 3) * Cr + q * 8 + (aj & 7)] = t;
    }
}

// sums of part[b][idx0 + k*Cr] (k < NS) over the blocks, by one wave.  The partials were written by other CUs moments
// ago, so every load is a round trip to memory: ALL of a lane's loads (8 blocks x NS values; nblk <= 512) are issued before
// the first add -- a trip-by-trip loop took 8 serial round trips, 14 of this kernel's 19 us, on the dependent chain.
#define CBN_MAX_BLOCKS 512
template <int NS>
__device__ __forceinline__ void wave_reduce_partials(const float* __restrict__ part, int nblk, int stride, int idx0, int Cr,
                                                     double (&out)[NS]) {
    float v[CBN_MAX_BLOCKS / 64][NS];
#pragma unroll
    for (int t = 0; t < CBN_MAX_BLOCKS / 64; ++t) {
        const int b = (threadIdx.x & 63) + 64 * t;
        const float* p = part + (size_t)(b < nblk ? b : 0) * stride + idx0;
#pragma unroll
        for (int k = 0; k < NS; ++k) v[t][k] = p[(size_t)k * Cr];  // unconditional (block 0 for b >= nblk): a predicated
    }                                                                 // load is followed by vmcnt(0) and serialises them all
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
#pragma unroll
    for (int t = 0; t < CBN_MAX_BLOCKS / 64; ++t) {
        const bool live = (int)(threadIdx.x & 63) + 64 * t < nblk;
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] += live ? (double)v[t][k] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] = wave_sum_d(acc[k]);
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cbn_stats_kernel(const bf16_raw* __restrict__ y, long rows, int Cr,
                                                        float* __restrict__ part /* [nblk][5*Cr] */) {
    __shared__ float lds[4 * 5 * 8 * 32];
    const int nq = Cr >> 3;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    float s[5][8];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[a][j] = 0.f;
    const int C = 2 * Cr;
    if (rl < rpb) {
        // four rows per trip: 8 independent 16-byte loads in flight per thread (one workgroup per CU must cover HBM latency)
        const long stride = (long)gridDim.x * rpb;
        for (long r0 = (long)blockIdx.x * rpb + rl; r0 < rows; r0 += 4 * stride) {
            uint4 ua[4], ub[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long r = r0 + u * stride;
                ua[u] = make_uint4(0u, 0u, 0u, 0u); ub[u] = ua[u];
                if (r < rows) {
                    ua[u] = *reinterpret_cast<const uint4*>(y + r * C + q * 8);
                    ub[u] = *reinterpret_cast<const uint4*>(y + r * C + Cr + q * 8);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const Chunk8 a = unpack8(ua[u]), b = unpack8(ub[u]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s[0][j] += a.v[j]; s[1][j] += b.v[j];
                    s[2][j] += a.v[j] * a.v[j]; s[3][j] += a.v[j] * b.v[j]; s[4][j] += b.v[j] * b.v[j];
                }
            }
        }
    }
    block_partials<5>(s, nq, Cr, part, 5 * Cr, lds);
}

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
        RVrr[c] = rvrr + momentum * (vrr - rvrr);
        RVri[c] = rvri + momentum * (vri - rvri);
        RVii[c] = rvii + momentum * (vii - rvii);
        if (c == 0 && nbt) nbt[0] += 1;
    } else {
        if (threadIdx.x != 0) return;
        mr = rmr; mi = rmi; vrr = rvrr; vri = rvri; vii = rvii;
    }
    vrr += eps; vii += eps;
    const float tau = vrr + vii;
    const float delta = vrr * vii - vri * vri;
    const float s = sqrtf(delta);
    const float t = sqrtf(tau + 2.f * s);
    const float rst = 1.f / (s * t);
    const float urr = (s + vii) * rst, uii = (s + vrr) * rst, uri = -vri * rst;
    float* o = coef + (size_t)c * COEF_STRIDE;
    o[0] = wrr * urr + wri * uri;
    o[1] = wrr * uri + wri * uii;
    o[2] = wri * urr + wii * uri;
    o[3] = wri * uri + wii * uii;
    o[4] = mr; o[5] = mi; o[6] = br_; o[7] = bi_;
    o[8] = urr; o[9] = uri; o[10] = uii; o[11] = vrr; o[12] = vri; o[13] = vii;
}

__global__ __launch_bounds__(256) void cbn_apply_kernel(const bf16_raw* __restrict__ y, const float* __restrict__ coef,
                                                        const float* __restrict__ slope, long rows, int Cr,
                                                        bf16_raw* __restrict__ z) {
    // a thread owns ONE chunk of 8 complex channels for all its rows: the 64 coefficient floats stay in registers
    // (re-loading them per row made the kernel TA-issue bound: 16 coefficient loads per 2 data loads)
    const int nq = Cr >> 3;
    const int C = 2 * Cr;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const float a = slope[0];
    float4 zc[8], mb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* k = coef + (size_t)(q * 8 + j) * COEF_STRIDE;
        zc[j] = *reinterpret_cast<const float4*>(k);
        mb[j] = *reinterpret_cast<const float4*>(k + 4);
    }
    for (long r = (long)blockIdx.x * rpb + rl; r < rows; r += (long)gridDim.x * rpb) {
        const Chunk8 xr = unpack8(*reinterpret_cast<const uint4*>(y + r * C + q * 8));
        const Chunk8 xi = unpack8(*reinterpret_cast<const uint4*>(y + r * C + Cr + q * 8));
        float orr[8], oii[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float cr = xr.v[j] - mb[j].x, ci = xi.v[j] - mb[j].y;
            const float vr = zc[j].x * cr + zc[j].y * ci + mb[j].z;
            const float vi = zc[j].z * cr + zc[j].w * ci + mb[j].w;
            orr[j] = vr > 0.f ? vr : a * vr;
            oii[j] = vi > 0.f ? vi : a * vi;
        }
        *reinterpret_cast<uint4*>(z + r * C + q * 8) = pack8(orr);
        *reinterpret_cast<uint4*>(z + r * C + Cr + q * 8) = pack8(oii);
    }
}

// ---------------------------------------------------------------------------------------------
// backward pass 1: per-channel sums  0 sum d_r  1 sum d_i  2 Qrr  3 Qri  4 Qir  5 Qii ; slope grad -> acc[6*Cr]
// rows whose stored frame index (row / F) % Tst is < tfirst carry dz == 0 (dropped decoder frame).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cbn_bwd_reduce_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ dz2,
                                                             const bf16_raw* __restrict__ y, const float* __restrict__ coef,
                                                             const float* __restrict__ slope, long rows, int Cr, int F,
                                                             int Tst, int tfirst, float* __restrict__ part) {
    __shared__ float lds[4 * 6 * 8 * 32];
    const int nq = Cr >> 3;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const int C = 2 * Cr;
    const float a = slope[0];
    float s[6][8];
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[u][j] = 0.f;
    float da = 0.f;
    float4 zc[8], mb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* k = coef + (size_t)(q * 8 + j) * COEF_STRIDE;
        zc[j] = *reinterpret_cast<const float4*>(k);
        mb[j] = *reinterpret_cast<const float4*>(k + 4);
    }
    if (rl < rpb)
        for (long r = (long)blockIdx.x * rpb + rl; r < rows; r += (long)gridDim.x * rpb) {
            if (tfirst > 0 && (int)((r / F) % Tst) < tfirst) continue;
            const Chunk8 xr = unpack8(*reinterpret_cast<const uint4*>(y + r * C + q * 8));
            const Chunk8 xi = unpack8(*reinterpret_cast<const uint4*>(y + r * C + Cr + q * 8));
            Chunk8 gr = unpack8(*reinterpret_cast<const uint4*>(dz + r * C + q * 8));
            Chunk8 gi = unpack8(*reinterpret_cast<const uint4*>(dz + r * C + Cr + q * 8));
            if (dz2) {
                const Chunk8 hr = unpack8(*reinterpret_cast<const uint4*>(dz2 + r * C + q * 8));
                const Chunk8 hi = unpack8(*reinterpret_cast<const uint4*>(dz2 + r * C + Cr + q * 8));
#pragma unroll
                for (int j = 0; j < 8; ++j) { gr.v[j] += hr.v[j]; gi.v[j] += hi.v[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float cr = xr.v[j] - mb[j].x, ci = xi.v[j] - mb[j].y;
                const float vr = zc[j].x * cr + zc[j].y * ci + mb[j].z;
                const float vi = zc[j].z * cr + zc[j].w * ci + mb[j].w;
                float dr = gr.v[j], di = gi.v[j];
                if (!(vr > 0.f)) { da += dr * vr; dr *= a; }
                if (!(vi > 0.f)) { da += di * vi; di *= a; }
                s[0][j] += dr; s[1][j] += di;
                s[2][j] += dr * cr; s[3][j] += dr * ci; s[4][j] += di * cr; s[5][j] += di * ci;
            }
        }
    block_partials<6>(s, nq, Cr, part, 6 * Cr + 1, lds);
    da = wave_sum(da);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = da;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * (6 * Cr + 1) + 6 * Cr] = lds[0] + lds[1] + lds[2] + lds[3];
}

// one wave per channel: parameter gradients + coefficients of the apply pass
__global__ void cbn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ coef,
                                        const float* __restrict__ Wrr, const float* __restrict__ Wri,
                                        const float* __restrict__ Wii, long rows, int Cr, float* __restrict__ gWrr,
                                        float* __restrict__ gWri, float* __restrict__ gWii, float* __restrict__ gBr,
                                        float* __restrict__ gBi, float* __restrict__ gslope, float* __restrict__ bcoef) {
    const int c = blockIdx.x;
    const int st = 6 * Cr + 1;
    if (c == Cr) {  // extra block: the PReLU slope gradient (its own block, so that channel 0 is not a straggler)
        double ds[1];
        wave_reduce_partials<1>(part, nblk, st, 6 * Cr, Cr, ds);
        if (threadIdx.x == 0) gslope[0] = (float)ds[0];
        return;
    }
    // the per-channel constants are requested before the reduction, not after it (one memory round trip less on the chain)
    const float* k = coef + (size_t)c * COEF_STRIDE;
    const float urr = k[8], uri = k[9], uii = k[10], vrr = k[11], vri = k[12], vii = k[13];
    const float wrr = Wrr[c], wri = Wri[c], wii = Wii[c];
    double a[6];
    wave_reduce_partials<6>(part, nblk, st, c, Cr, a);
    const float sdr = (float)a[0], sdi = (float)a[1], qrr = (float)a[2], qri = (float)a[3], qir = (float)a[4], qii = (float)a[5];
    if (threadIdx.x != 0) return;
    const float n = (float)rows;
    // P = Q U  (sum d xh^T)
    const float prr = qrr * urr + qri * uri, pri = qrr * uri + qri * uii;
    const float pir = qir * urr + qii * uri, pii = qir * uri + qii * uii;
    gWrr[c] = prr;
    gWri[c] = pri + pir;
    gWii[c] = pii;
    gBr[c] = sdr;
    gBi[c] = sdi;
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
    float* o = bcoef + (size_t)c * COEF_STRIDE;
    o[0] = arr; o[1] = ari; o[2] = air; o[3] = aii;
    o[4] = 2.f * d_vrr / n; o[5] = d_vri / n; o[6] = 2.f * d_vii / n;
    o[7] = -(arr * sdr + ari * sdi) / n;
    o[8] = -(air * sdr + aii * sdi) / n;
}

__global__ __launch_bounds__(256) void cbn_bwd_apply_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ dz2,
                                                            const bf16_raw* __restrict__ y, const float* __restrict__ coef,
                                                            const float* __restrict__ bcoef, const float* __restrict__ slope,
                                                            long rows, int Cr, int F, int Tst, int tfirst,
                                                            bf16_raw* __restrict__ dy) {
    const int nq = Cr >> 3;
    const int C = 2 * Cr;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const float a = slope[0];
    // per-channel coefficients of this thread's 8 complex channels, in registers for the whole pass
    float4 zc[8], mb[8], A[8], E[8];
    float ki[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* k = coef + (size_t)(q * 8 + j) * COEF_STRIDE;
        const float* kb = bcoef + (size_t)(q * 8 + j) * COEF_STRIDE;
        zc[j] = *reinterpret_cast<const float4*>(k);
        mb[j] = *reinterpret_cast<const float4*>(k + 4);
        A[j] = *reinterpret_cast<const float4*>(kb);
        E[j] = *reinterpret_cast<const float4*>(kb + 4);  // Err Eri Eii kr
        ki[j] = kb[8];
    }
    for (long r = (long)blockIdx.x * rpb + rl; r < rows; r += (long)gridDim.x * rpb) {
        const bool dropped = tfirst > 0 && (int)((r / F) % Tst) < tfirst;
        const Chunk8 xr = unpack8(*reinterpret_cast<const uint4*>(y + r * C + q * 8));
        const Chunk8 xi = unpack8(*reinterpret_cast<const uint4*>(y + r * C + Cr + q * 8));
        Chunk8 gr, gi;
        if (dropped) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { gr.v[j] = 0.f; gi.v[j] = 0.f; }
        } else {
            gr = unpack8(*reinterpret_cast<const uint4*>(dz + r * C + q * 8));
            gi = unpack8(*reinterpret_cast<const uint4*>(dz + r * C + Cr + q * 8));
            if (dz2) {
                const Chunk8 hr = unpack8(*reinterpret_cast<const uint4*>(dz2 + r * C + q * 8));
                const Chunk8 hi = unpack8(*reinterpret_cast<const uint4*>(dz2 + r * C + Cr + q * 8));
#pragma unroll
                for (int j = 0; j < 8; ++j) { gr.v[j] += hr.v[j]; gi.v[j] += hi.v[j]; }
            }
        }
        float orr[8], oii[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float cr = xr.v[j] - mb[j].x, ci = xi.v[j] - mb[j].y;
            const float vr = zc[j].x * cr + zc[j].y * ci + mb[j].z;
            const float vi = zc[j].z * cr + zc[j].w * ci + mb[j].w;
            float dr = gr.v[j], di = gi.v[j];
            if (!(vr > 0.f)) dr *= a;
            if (!(vi > 0.f)) di *= a;
            orr[j] = A[j].x * dr + A[j].y * di + E[j].x * cr + E[j].y * ci + E[j].w;
            oii[j] = A[j].z * dr + A[j].w * di + E[j].y * cr + E[j].z * ci + ki[j];
        }
        *reinterpret_cast<uint4*>(dy + r * C + q * 8) = pack8(orr);
        *reinterpret_cast<uint4*>(dy + r * C + Cr + q * 8) = pack8(oii);
    }
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

static int apply_blocks(long rows, int Cr) {
    const int rpb = 256 / (Cr >> 3);
    long g = (rows + (long)rpb * 4 - 1) / ((long)rpb * 4);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

static int stat_blocks(long rows, int Cr) {
    const int rpb = 256 / (Cr >> 3);
    long g = (rows + (long)rpb * 8 - 1) / ((long)rpb * 8);
    // one block per CU: the finalize kernels (on the dependent chain) read half the partials of the 512-block version;
    // measured 6.24 vs 6.27 ms per step (128 blocks: 6.40)
    static const int cap = getenv("SEHIP_CBN_BLOCKS") ? atoi(getenv("SEHIP_CBN_BLOCKS")) : 256;
    if (g > cap) g = cap;
    if (g > CBN_MAX_BLOCKS) g = CBN_MAX_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

// number of floats the partial-sum scratch must hold (forward stats and backward reduce share it)
extern "C" long sehip_cbn_scratch_floats(long rows, int Cr) { return (long)stat_blocks(rows, Cr) * (6L * Cr + 1); }

extern "C" int sehip_cbn_stats(const void* y, long rows, int Cr, float* part, void* stream) {
    if (int e = check_cbn("cbn_stats", rows, Cr)) return e;
    cbn_stats_kernel<<<stat_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, rows, Cr, part);
    SEHIP_CHECK_LAUNCH("cbn_stats");
    return 0;
}

// params: Wrr,Wri,Wii,Br,Bi [Cr] fp32; buffers RMr,RMi,RVrr,RVri,RVii [Cr] fp32 (updated in training), nbt int64[1]
extern "C" int sehip_cbn_finalize(const float* part, const float* Wrr, const float* Wri, const float* Wii, const float* Br,
                                  const float* Bi, float* RMr, float* RMi, float* RVrr, float* RVri, float* RVii, long* nbt,
                                  long rows, int Cr, float eps, float momentum, int training, float* coef, void* stream) {
    if (int e = check_cbn("cbn_finalize", rows, Cr)) return e;
    cbn_finalize_kernel<<<Cr, 64, 0, (hipStream_t)stream>>>(part, stat_blocks(rows, Cr), Wrr, Wri, Wii, Br, Bi, RMr, RMi, RVrr,
                                                            RVri, RVii, nbt, rows, Cr, eps, momentum, training, coef);
    SEHIP_CHECK_LAUNCH("cbn_finalize");
    return 0;
}

extern "C" int sehip_cbn_apply(const void* y, const float* coef, const float* slope, long rows, int Cr, void* z, void* stream) {
    if (int e = check_cbn("cbn_apply", rows, Cr)) return e;
    cbn_apply_kernel<<<apply_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, coef, slope, rows, Cr,
                                                                              (bf16_raw*)z);
    SEHIP_CHECK_LAUNCH("cbn_apply");
    return 0;
}

extern "C" int sehip_cbn_bwd_reduce(const void* dz, const void* dz2, const void* y, const float* coef, const float* slope,
                                    long rows, int Cr, int F, int Tst, int tfirst, float* part, void* stream) {
    if (int e = check_cbn("cbn_bwd_reduce", rows, Cr)) return e;
    cbn_bwd_reduce_kernel<<<stat_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>(
        (const bf16_raw*)dz, (const bf16_raw*)dz2, (const bf16_raw*)y, coef, slope, rows, Cr, F, Tst, tfirst, part);
    SEHIP_CHECK_LAUNCH("cbn_bwd_reduce");
    return 0;
}

extern "C" int sehip_cbn_bwd_finalize(const float* part, const float* coef, const float* Wrr, const float* Wri,
                                      const float* Wii, long rows, int Cr, float* gWrr, float* gWri, float* gWii, float* gBr,
                                      float* gBi, float* gslope, float* bcoef, void* stream) {
    if (int e = check_cbn("cbn_bwd_finalize", rows, Cr)) return e;
    cbn_bwd_finalize_kernel<<<Cr + 1, 64, 0, (hipStream_t)stream>>>(part, stat_blocks(rows, Cr), coef, Wrr, Wri, Wii, rows, Cr, gWrr,
                                                                gWri, gWii, gBr, gBi, gslope, bcoef);
    SEHIP_CHECK_LAUNCH("cbn_bwd_finalize");
    return 0;
}

extern "C" int sehip_cbn_bwd_apply(const void* dz, const void* dz2, const void* y, const float* coef, const float* bcoef,
                                   const float* slope, long rows, int Cr, int F, int Tst, int tfirst, void* dy, void* stream) {
    if (int e = check_cbn("cbn_bwd_apply", rows, Cr)) return e;
    cbn_bwd_apply_kernel<<<apply_blocks(rows, Cr), 256, 0, (hipStream_t)stream>>>(
        (const bf16_raw*)dz, (const bf16_raw*)dz2, (const bf16_raw*)y, coef, bcoef, slope, rows, Cr, F, Tst, tfirst,
        (bf16_raw*)dy);
    SEHIP_CHECK_LAUNCH("cbn_bwd_apply");
    return 0;
}
