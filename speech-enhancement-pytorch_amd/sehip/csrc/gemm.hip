// Implicit-GEMM engine on bf16 MFMA (v_mfma_f32_16x16x32_bf16, fp32 accumulate) -- see include/sehip.h.
//
// Replaces the 4-real-conv formulation of ComplexConv2d / ComplexConvTranspose2d (src/model/dccrn.py:316-450),
// the channel chunk/cat copies around them (complex_cat :304-314) and the dense products of NavieComplexLSTM
// (:264-302).  A complex conv is ONE real GEMM over K = taps x (real|imag) input channels with the packed block
// weight [[Wr,-Wi],[Wi,Wr]]; the im2col matrix is never built: 16-byte chunks (8 consecutive k) are gathered from
// the channels-last activations through a per-chunk table, so conv, parity-split transposed conv, both dgrads, the
// two-source skip concatenation and plain linear layers all run through the same two kernels.
//
// gemm_kernel   D[n][m] = sum_k W[n][k] A[m][k]     (weights are the MFMA "A" operand, activations the "B"
//               operand, so a lane ends up with 4 consecutive output channels of one row -> 8/16-byte stores)
//   tile BN x BM x 64, 4 waves, LDS rows of 128 B with a 16-byte XOR swizzle (conflict-free b128 reads and
//   writes), register-staged prefetch of the next K tile while the MFMAs of the current one run.
// wgrad_kernel  dW[n][k] += sum_m dOut[m][n] A[m][k]  (split over m, fp32 atomics)
//   both operands are staged row-major in m and fed to the MFMA through ds_read_b64_tr_b16 transposed reads.
#include "common.h"
#include "../../../include/sehip.h"

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct RowPos { int b, t, jf; bool valid; };

__device__ __forceinline__ RowPos row_pos(int m, int M, int TT, int J, int fmul) {
    RowPos r;
    r.valid = m < M;
    const int mm = r.valid ? m : 0;
    const int bt = mm / J;
    r.jf = (mm - bt * J) * fmul;
    r.b = bt / TT;
    r.t = bt - r.b * TT;
    return r;
}

// element offset of (b, t, jf) in source s, before the per-chunk delta
__device__ __forceinline__ long row_base(const sehip_src& s, const RowPos r) {
    return (((long)r.b * s.T + r.t) * s.F + r.jf) * s.C;
}

// One 16-byte chunk (8 consecutive k) of the implicit A matrix.  The chunk table carries the precomputed element
// delta (toff*F + fadd)*C + coff, so the address is row_base + delta; toff/fadd are only needed for the bounds.
__device__ __forceinline__ uint4 gather_chunk(const sehip_src& s0, const sehip_src& s1, const sehip_kchunk e, const RowPos r,
                                              long rb0, long rb1) {
    uint4 z = make_uint4(0u, 0u, 0u, 0u);
    if (!r.valid || e.src < 0) return z;
    const bool second = e.src != 0;
    const int toff = e.toff >> 16, fadd = (int)(short)(e.toff & 0xffff);
    const int ts = r.t + toff;
    const int f = r.jf + fadd;
    const int tlo = second ? s1.tlo : s0.tlo, thi = second ? s1.thi : s0.thi, F = second ? s1.F : s0.F;
    if (ts < tlo || ts >= thi) return z;
    const bf16_raw* base = reinterpret_cast<const bf16_raw*>(second ? s1.ptr : s0.ptr);
    const long off = (second ? rb1 : rb0) + e.fadd;  // e.fadd holds the element delta
    const int C = second ? s1.C : s0.C;
    if (C == 2) {  // narrow source: up to 4 consecutive rows x (re, im); e.coff = number of valid rows
        const unsigned* p = reinterpret_cast<const unsigned*>(base + off);
        unsigned v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int fr = f + q;
            v[q] = (q < e.coff && fr >= 0 && fr < F) ? p[q] : 0u;
        }
        return make_uint4(v[0], v[1], v[2], v[3]);
    }
    if ((unsigned)f >= (unsigned)F) return z;
    return *reinterpret_cast<const uint4*>(base + off);
}

__device__ __forceinline__ size_t dst_row_offset(const sehip_dst& d, const RowPos r, int fmul_row) {
    // r.jf = j*fmul_row ; destination uses its own multiplier
    const int j = r.jf / fmul_row;
    return (((size_t)r.b * d.T + r.t + d.toff) * d.F + (size_t)j * d.fmul + d.fadd) * d.C;
}

// ------------------------------------------------------------------------------------------------
template <int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(256) void gemm_kernel(const sehip_gemm_desc d) {
    constexpr int TN = BN / WN / 16, TM = BM / WM / 16;
    constexpr int NRA = BM / 32;
    constexpr int NRW = (BN + 31) / 32;
    __shared__ uint4 sW[BN * 8];
    __shared__ uint4 sA[BM * 8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    const int kc = tid & 7, r0 = tid >> 3;
    RowPos rp[NRA];
    long rb0[NRA], rb1[NRA];
#pragma unroll
    for (int i = 0; i < NRA; ++i) {
        rp[i] = row_pos(m0 + r0 + 32 * i, d.M, d.TT, d.J, d.fmul);
        rb0[i] = row_base(d.src[0], rp[i]);
        rb1[i] = row_base(d.src[1], rp[i]);
    }

    const int nk = d.K >> 6;
    const bf16_raw* Wb = reinterpret_cast<const bf16_raw*>(d.W);
    uint4 ra[NRA], rw[NRW];

    auto issue = [&](int kt) {
        const sehip_kchunk e = d.ktab[kt * 8 + kc];
#pragma unroll
        for (int i = 0; i < NRA; ++i) ra[i] = gather_chunk(d.src[0], d.src[1], e, rp[i], rb0[i], rb1[i]);
#pragma unroll
        for (int i = 0; i < NRW; ++i) {
            const int rw_row = r0 + 32 * i;
            if (rw_row < BN)
                rw[i] = *reinterpret_cast<const uint4*>(Wb + (size_t)(n0 + rw_row) * d.K + kt * 64 + kc * 8);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int r = r0 + 32 * i;
            sA[r * 8 + (kc ^ (r & 7))] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NRW; ++i) {
            const int r = r0 + 32 * i;
            if (r < BN) sW[r * 8 + (kc ^ (r & 7))] = rw[i];
        }
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + (lane >> 4);
            bf16x8 wf[TN], af[TM];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int r = wn * (BN / WN) + ni * 16 + (lane & 15);
                wf[ni] = __builtin_bit_cast(bf16x8, sW[r * 8 + (c ^ (r & 7))]);
            }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int r = wm * (BM / WM) + mi * 16 + (lane & 15);
                af[mi] = __builtin_bit_cast(bf16x8, sA[r * 8 + (c ^ (r & 7))]);
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: lane holds n = nb + 4*(lane>>4) + {0..3}, m = mb + (lane&15)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + wm * (BM / WM) + mi * 16 + (lane & 15);
        const RowPos r = row_pos(m, d.M, d.TT, d.J, d.fmul);
        if (!r.valid) continue;
        const size_t ro0 = dst_row_offset(d.dst[0], r, d.fmul);
        const size_t ro1 = d.dst[1].ptr ? dst_row_offset(d.dst[1], r, d.fmul) : 0;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * (BN / WN) + ni * 16 + 4 * (lane >> 4);
            const sehip_nchunk nc = d.ntab[n >> 2];
            if (nc.nvalid <= 0) continue;
            f32x4 v = acc[ni][mi];
            if (d.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(d.bias + n);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            const sehip_dst& ds = nc.dst ? d.dst[1] : d.dst[0];
            const size_t off = (nc.dst ? ro1 : ro0) + nc.coff;
            if (ds.is_f32) {
                float* p = reinterpret_cast<float*>(ds.ptr) + off;
                if (nc.nvalid == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int q = 0; q < nc.nvalid; ++q) p[q] = v[q];
            } else {
                bf16_raw* p = reinterpret_cast<bf16_raw*>(ds.ptr) + off;
                if (nc.nvalid == 4) *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                else
                    for (int q = 0; q < nc.nvalid; ++q) p[q] = f2bf(v[q]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad: tile BNW (n) x 64 (k), m consumed 64 rows per step
// ------------------------------------------------------------------------------------------------
template <int BNW, int WNN, int WNK>
__global__ __launch_bounds__(256) void wgrad_kernel(const sehip_gemm_desc d, int m_per_block) {
    constexpr int TN = BNW / WNN / 16, TK = 64 / WNK / 16;
    constexpr int PG = BNW + 8;  // pitch in bf16 elements (16 B pad)
    constexpr int PX = 64 + 8;
    constexpr int GCH = BNW / 8;         // 16-byte chunks per dOut row
    constexpr int GPT = (64 * GCH + 255) / 256;  // dOut chunks per thread
    __shared__ __attribute__((aligned(16))) bf16_raw sG[64 * PG];
    __shared__ __attribute__((aligned(16))) bf16_raw sX[64 * PX];
    __shared__ sehip_dst sdst[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WNK, wk = wave % WNK;
    const int n0 = blockIdx.x * BNW, k0 = blockIdx.y * 64;
    const int m_begin = blockIdx.z * m_per_block;
    const int m_end = min(d.M, m_begin + m_per_block);
    if (tid < 2) sdst[tid] = d.dst[tid];
    __syncthreads();

    const int kc = tid & 7, r0 = tid >> 3;
    const sehip_kchunk e = d.ktab[(k0 >> 3) + kc];

    f32x4 acc[TN][TK];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float colsum = 0.f;
    const bool do_bias = d.dbias != nullptr && blockIdx.y == 0;

    for (int mb = m_begin; mb < m_end; mb += 64) {
        // ---- stage A chunks: 64 rows x 8 chunks
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = r0 + 32 * i;
            const int m = mb + r;
            RowPos rp = row_pos(m, m_end, d.TT, d.J, d.fmul);
            const uint4 v = gather_chunk(d.src[0], d.src[1], e, rp, row_base(d.src[0], rp), row_base(d.src[1], rp));
            *reinterpret_cast<uint4*>(&sX[r * PX + kc * 8]) = v;
        }
        // ---- stage dOut chunks: 64 rows x GCH chunks
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int id = tid + 256 * i;
            if (id < 64 * GCH) {
                const int r = id / GCH, gc = id - r * GCH;
                const int m = mb + r;
                RowPos rp = row_pos(m, m_end, d.TT, d.J, d.fmul);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (rp.valid) {
                    const int n = n0 + gc * 8;
                    const sehip_nchunk c0 = d.ntab[n >> 2], c1 = d.ntab[(n >> 2) + 1];
                    if (c0.nvalid == 4 && c1.nvalid == 4 && c1.dst == c0.dst && c1.coff == c0.coff + 4) {
                        const sehip_dst& ds = sdst[c0.dst];
                        const bf16_raw* p = reinterpret_cast<const bf16_raw*>(ds.ptr) + dst_row_offset(ds, rp, d.fmul) + c0.coff;
                        v = *reinterpret_cast<const uint4*>(p);
                    } else {
                        bf16_raw tmp[8];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const sehip_nchunk c = h ? c1 : c0;
                            const sehip_dst& ds = sdst[c.dst > 0 ? 1 : 0];
                            const bf16_raw* p = reinterpret_cast<const bf16_raw*>(ds.ptr) + dst_row_offset(ds, rp, d.fmul) + c.coff;
#pragma unroll
                            for (int q = 0; q < 4; ++q) tmp[h * 4 + q] = (q < c.nvalid) ? p[q] : (bf16_raw)0;
                        }
                        v = make_uint4(tmp[0] | ((unsigned)tmp[1] << 16), tmp[2] | ((unsigned)tmp[3] << 16),
                                       tmp[4] | ((unsigned)tmp[5] << 16), tmp[6] | ((unsigned)tmp[7] << 16));
                    }
                }
                *reinterpret_cast<uint4*>(&sG[r * PG + gc * 8]) = v;
            }
        }
        __syncthreads();
        if (do_bias && tid < BNW) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < 64; ++r) s += bf2f(sG[r * PG + tid]);
            colsum += s;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int g = lane >> 4, i16 = lane & 15;
            const int mrow = sub * 32 + 8 * g + (i16 >> 2);
            bf16x8 gf[TN], xf[TK];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = wn * (BNW / WNN) + ni * 16 + 4 * (i16 & 3);
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[mrow * PG + col]);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[(mrow + 4) * PG + col]);
                gf[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int ki = 0; ki < TK; ++ki) {
                const int col = wk * (64 / WNK) + ki * 16 + 4 * (i16 & 3);
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sX[mrow * PX + col]);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sX[(mrow + 4) * PX + col]);
                xf[ki] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int ki = 0; ki < TK; ++ki)
                    acc[ni][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], xf[ki], acc[ni][ki], 0, 0, 0);
        }
        __syncthreads();
    }

    // D rows = n (4*(lane>>4)+q), cols = k (lane&15)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int ki = 0; ki < TK; ++ki) {
            const int n = n0 + wn * (BNW / WNN) + ni * 16 + 4 * (lane >> 4);
            const int k = k0 + wk * (64 / WNK) + ki * 16 + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(&d.dW[(size_t)(n + q) * d.K + k], acc[ni][ki][q]);
        }
    if (do_bias && tid < BNW) atomicAdd(&d.dbias[n0 + tid], colsum);
}

// ------------------------------------------------------------------------------------------------
static int check_desc(const char* who, const sehip_gemm_desc* d) {
    SEHIP_REQUIRE(d != nullptr, "%s: null descriptor", who);
    SEHIP_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "%s: empty problem (M=%d N=%d K=%d)", who, d->M, d->N, d->K);
    SEHIP_REQUIRE((d->K & 63) == 0, "%s: K=%d must be a multiple of 64", who, d->K);
    SEHIP_REQUIRE((d->Npad & 15) == 0 && d->Npad >= d->N, "%s: Npad=%d must be a multiple of 16 and >= N", who, d->Npad);
    SEHIP_REQUIRE(d->TT > 0 && d->J > 0 && d->fmul > 0, "%s: bad row decomposition", who);
    SEHIP_REQUIRE(d->ktab && d->ntab && d->dst[0].ptr, "%s: missing table / destination", who);
    SEHIP_REQUIRE(d->M % d->J == 0 && (d->M / d->J) % d->TT == 0, "%s: M=%d is not B*TT*J", who, d->M);
    SEHIP_REQUIRE(d->src[2].ptr == nullptr && d->src[3].ptr == nullptr, "%s: at most two sources are built", who);
    for (int s = 0; s < 4; ++s)
        if (d->src[s].ptr) {
            SEHIP_REQUIRE(d->src[s].C == 2 || (d->src[s].C & 7) == 0, "%s: source %d has C=%d (need 2 or a multiple of 8)", who, s, d->src[s].C);
            SEHIP_REQUIRE((((uintptr_t)d->src[s].ptr) & 15) == 0, "%s: source %d is not 16-byte aligned", who, s);
        }
    return 0;
}

extern "C" int sehip_gemm_desc_size(void) { return (int)sizeof(sehip_gemm_desc); }

extern "C" int sehip_gemm(const sehip_gemm_desc* d, void* stream) {
    if (int e = check_desc("gemm", d)) return e;
    SEHIP_REQUIRE(d->W != nullptr, "gemm: missing weights");
    hipStream_t st = (hipStream_t)stream;
    if (d->Npad == 16) {
        gemm_kernel<16, 256, 1, 4><<<dim3(cdiv(d->M, 256), 1), 256, 0, st>>>(*d);
    } else if (d->Npad == 32) {
        gemm_kernel<32, 256, 1, 4><<<dim3(cdiv(d->M, 256), 1), 256, 0, st>>>(*d);
    } else if (d->Npad == 64) {
        gemm_kernel<64, 256, 1, 4><<<dim3(cdiv(d->M, 256), 1), 256, 0, st>>>(*d);
    } else {
        SEHIP_REQUIRE(d->Npad % 128 == 0, "gemm: Npad=%d must be 16, 32, 64 or a multiple of 128", d->Npad);
        gemm_kernel<128, 128, 2, 2><<<dim3(cdiv(d->M, 128), d->Npad / 128), 256, 0, st>>>(*d);
    }
    SEHIP_CHECK_LAUNCH("gemm");
    return 0;
}

extern "C" int sehip_wgrad(const sehip_gemm_desc* d, void* stream) {
    if (int e = check_desc("wgrad", d)) return e;
    SEHIP_REQUIRE(d->dW != nullptr, "wgrad: missing dW");
    SEHIP_REQUIRE(!d->dst[0].is_f32 && !(d->dst[1].ptr && d->dst[1].is_f32), "wgrad: dOut must be bf16");
    hipStream_t st = (hipStream_t)stream;
    const int ktiles = d->K / 64;
    int ntiles, bnw;
    if (d->Npad == 16) bnw = 16; else if (d->Npad == 32) bnw = 32; else if (d->Npad == 64) bnw = 64; else bnw = 128;
    SEHIP_REQUIRE(d->Npad % bnw == 0, "wgrad: Npad=%d must be 16, 32, 64 or a multiple of 128", d->Npad);
    ntiles = d->Npad / bnw;
    // split m so that the grid has ~2048 workgroups, at least 256 rows each
    long want = 2048 / ((long)ntiles * ktiles);
    if (want < 1) want = 1;
    long mpb = ((d->M + want - 1) / want + 63) / 64 * 64;
    if (mpb < 256) mpb = 256;
    const int splits = cdiv(d->M, mpb);
    dim3 grid(ntiles, ktiles, splits);
    if (bnw == 16) wgrad_kernel<16, 1, 4><<<grid, 256, 0, st>>>(*d, (int)mpb);
    else if (bnw == 32) wgrad_kernel<32, 1, 4><<<grid, 256, 0, st>>>(*d, (int)mpb);
    else if (bnw == 64) wgrad_kernel<64, 2, 2><<<grid, 256, 0, st>>>(*d, (int)mpb);
    else wgrad_kernel<128, 2, 2><<<grid, 256, 0, st>>>(*d, (int)mpb);
    SEHIP_CHECK_LAUNCH("wgrad");
    return 0;
}
