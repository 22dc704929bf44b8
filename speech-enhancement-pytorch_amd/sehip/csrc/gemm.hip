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
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "../../../include/sehip.h"

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct RowPos { int b, t, jf; bool valid; int ts; int j; };  // ts = source frame of the row (t * tmul); t = frame of the row space; jf = j * fmul

__device__ __forceinline__ RowPos row_pos(int m, int M, int TT, int J, int fmul, int tmul = 1) {
    RowPos r;
    r.valid = m < M;
    const int mm = r.valid ? m : 0;
    const int bt = mm / J;
    r.j = mm - bt * J;
    r.jf = r.j * fmul;
    r.b = bt / TT;
    r.t = bt - r.b * TT;
    r.ts = r.t * tmul;
    return r;
}
__device__ __forceinline__ int src_tmul(const sehip_gemm_desc& d) { return d.tmul > 1 ? d.tmul : 1; }

// element offset of (b, t*tmul, jf) in source s, before the per-chunk delta
__device__ __forceinline__ long row_base(const sehip_src& s, const RowPos r) {
    return (((long)r.b * s.T + r.ts) * s.F + r.jf) * s.C;
}

// One 16-byte chunk (8 consecutive k) of the implicit A matrix.  The chunk table carries the precomputed element
// delta (toff*F + fadd)*C + coff, so the address is row_base + delta; toff/fadd are only needed for the bounds.
__device__ __forceinline__ uint4 gather_chunk(const sehip_src& s0, const sehip_src& s1, const sehip_kchunk e, const RowPos r,
                                              long rb0, long rb1) {
    uint4 z = make_uint4(0u, 0u, 0u, 0u);
    if (!r.valid || e.src < 0) return z;
    const bool second = e.src != 0;
    const int toff = e.toff >> 16, fadd = (int)(short)(e.toff & 0xffff);
    const int ts = r.ts + toff;
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
    // the destination uses its own row multiplier (fmul_row is the row space's: r.jf = r.j * fmul_row)
    (void)fmul_row;
    const int j = r.j;
    return (((size_t)r.b * d.T + r.t * (d.tmul > 1 ? d.tmul : 1) + d.toff) * d.F + (size_t)j * d.fmul + d.fadd) * d.C;
}

// ------------------------------------------------------------------------------------------------
// 16 zero bytes in device memory: staging loads of padded / out-of-range pieces read this instead of branching
__device__ uint4 sehip_zero16 = {0u, 0u, 0u, 0u};

#define BF16_ONES __builtin_bit_cast(bf16x8, (s16x8){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80})

template <int N>
struct RegTile { uint4 v[N]; };

// Staging fetch shared by the small-channel kernels: piece u of this thread lives at p_ptr[u] (address for batch 0, tile
// frame 0) + a per-source tile offset; p_pp[u] = patch frame | source << 16, or -1 for a piece that is always zero.  A
// piece outside the source's frame range reads sehip_zero16 instead of branching (one wave per SIMD: every branch and
// every dependent instruction is exposed).  The loads are explicit global-address-space loads: a select between
// pointers of unknown provenance becomes a flat load, which may alias private memory and forced the prefetch
// registers into scratch.
template <int NPC>
__device__ __forceinline__ RegTile<NPC> sw_fetch_patch(const bf16_raw* const (&p_ptr)[NPC], const int (&p_pp)[NPC], long off0, long off1,
                                                       int lo0, int span0, int lo1, int span1, const bf16_raw* zero_page) {
    RegTile<NPC> t;
#pragma unroll
    for (int u = 0; u < NPC; ++u) {
        const int e = p_pp[u];
        const int sec = (e >> 16) & 1;
        const unsigned rel = (unsigned)((e & 0xffff) - (sec ? lo1 : lo0));
        const int ok = (int)(e >= 0) & (int)(rel < (unsigned)(sec ? span1 : span0));
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) u32x4_t* gvec_ptr;
        const bf16_raw* q = p_ptr[u] + (sec ? off1 : off0);
        q = ok ? q : zero_page;
        const u32x4_t x = *(gvec_ptr)(q);
        t.v[u] = make_uint4(x[0], x[1], x[2], x[3]);
    }
    return t;
}


// residual (descriptor field res): out = product + res, bf16 in, fp32 add
__device__ __forceinline__ uint4 add_bf16x8(uint4 a, uint4 r) {
    const unsigned av[4] = {a.x, a.y, a.z, a.w}, rv[4] = {r.x, r.y, r.z, r.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = pack_bf2(__uint_as_float(av[i] << 16) + __uint_as_float(rv[i] << 16),
                        __uint_as_float(av[i] & 0xffff0000u) + __uint_as_float(rv[i] & 0xffff0000u));
    return make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void add_res4(f32x4& v, const void* res, size_t off) {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_raw*>(res) + off);
    v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
    v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
}

// Scatter of one lane's 4 consecutive output channels (shared by both product kernels).
__device__ __forceinline__ void store_out4(const sehip_gemm_desc& d, const sehip_nchunk nc, f32x4 v, size_t ro0, size_t ro1,
                                           int n) {
    if (d.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(d.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    }
    const size_t off = (nc.dst ? ro1 : ro0) + nc.coff;
    void* dptr = nc.dst ? d.dst[1].ptr : d.dst[0].ptr;
    const int is_f32 = nc.dst ? d.dst[1].is_f32 : d.dst[0].is_f32;
    if (d.res && nc.dst == 0 && nc.nvalid == 4) add_res4(v, d.res, off);
    if (is_f32) {
        float* p = reinterpret_cast<float*>(dptr) + off;
        if (nc.nvalid == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int q = 0; q < nc.nvalid; ++q) p[q] = v[q];
    } else {
        bf16_raw* p = reinterpret_cast<bf16_raw*>(dptr) + off;
        if (nc.nvalid == 4) *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        else
            for (int q = 0; q < nc.nvalid; ++q) p[q] = f2bf(v[q]);
    }
}

template <int NRA>
__device__ __forceinline__ RegTile<NRA> fetch_a_tile(const sehip_gemm_desc& d, int kt, int kc, const RowPos (&rp)[NRA],
                                                     const long (&rb0)[NRA], const long (&rb1)[NRA]) {
    RegTile<NRA> t;
    const sehip_kchunk e = d.ktab[kt * 8 + kc];
#pragma unroll
    for (int i = 0; i < NRA; ++i) t.v[i] = gather_chunk(d.src[0], d.src[1], e, rp[i], rb0[i], rb1[i]);
    return t;
}

template <int NRW, int BN>
__device__ __forceinline__ RegTile<NRW> fetch_w_tile(const bf16_raw* Wb, int K, int n0, int r0, int kcol) {
    RegTile<NRW> t;
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
        const int row = r0 + 32 * i;
        t.v[i] = make_uint4(0u, 0u, 0u, 0u);
        if (row < BN) t.v[i] = *reinterpret_cast<const uint4*>(Wb + (size_t)(n0 + row) * K + kcol);
    }
    return t;
}

// kt0 / kt1: the K steps (64 columns each) this workgroup sums, kt1 < 0 = all; part != NULL: the accumulators go to part[m][Npad]
// (fp32, no bias / residual / scatter: gemm_splitk_finish_kernel adds the splits and stores) -- the split-K launches of sehip_gemm
template <int BN, int BM, int WN, int WM>
__device__ __forceinline__ void gemm_body(const sehip_gemm_desc& d, const int kt0 = 0, const int kt1_ = -1, float* __restrict__ part = nullptr) {
    constexpr int TN = BN / WN / 16, TM = BM / WM / 16;
    constexpr int NRA = BM / 32;
    constexpr int NRW = (BN + 31) / 32;
    // one LDS block: the two operand tiles during the K loop, the waves' bf16 output images afterwards (dense epilogue)
    constexpr int WR = BM / WM, WC = BN / WN, TP = WC + 8;
    constexpr int OPER_U4 = (BN + BM) * 8, IMG_U4 = (4 * WR * TP * 2 + 15) / 16;
    __shared__ uint4 sbuf[OPER_U4 > IMG_U4 ? OPER_U4 : IMG_U4];
    uint4* sW = sbuf;
    uint4* sA = sbuf + BN * 8;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Give every XCD a contiguous range of
    // the (m-tile, n-tile) space with the n-tiles of one m-tile adjacent, so the gathered activation rows (shared by
    // all n-tiles and, through the conv halo, by neighbouring m-tiles) are re-read from that XCD's L2.
    const int ntn = d.Npad / BN;
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int m0 = (logical / ntn) * BM, n0 = (logical % ntn) * BN;

    const int kc = tid & 7, r0 = tid >> 3;
    // rows r0 + 32 i of the tile: ONE decomposition into (utterance, frame, row), the others by stepping 32 rows with carries
    // (a generic division is ~17 vector instructions; products with a short K spend their time in this prologue / the epilogue)
    RowPos rp[NRA];
    long rb0[NRA], rb1[NRA];
    {
        const int tm = src_tmul(d);
        RowPos cur = row_pos(m0 + r0, d.M, d.TT, d.J, d.fmul, tm);
        const int st = 32 / d.J, sj = 32 - st * d.J;
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            rp[i] = cur;
            rp[i].valid = m0 + r0 + 32 * i < d.M;
            rb0[i] = row_base(d.src[0], rp[i]);
            rb1[i] = row_base(d.src[1], rp[i]);
            cur.j += sj; cur.t += st;
            if (cur.j >= d.J) { cur.j -= d.J; ++cur.t; }
            for (; cur.t >= d.TT; cur.t -= d.TT) ++cur.b;
            cur.jf = cur.j * d.fmul; cur.ts = cur.t * tm;
        }
    }

    const int nk = kt1_ < 0 ? (d.K >> 6) : kt1_;
    const bf16_raw* Wb = reinterpret_cast<const bf16_raw*>(d.W);

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A-tile gather.  The chunk-table entry of K step kt+2 is requested while step kt runs (it used to be loaded and then used
    // for the gather addresses inside the same step: two dependent memory round trips per 64 k-columns), out-of-range pieces
    // read a zero page instead of branching, and the weight rows are a pointer that moves 64 columns per step.
    const bool narrow = d.src[0].C == 2 || (d.src[1].ptr && d.src[1].C == 2);
    const bf16_raw* zero_page = reinterpret_cast<const bf16_raw*>(&sehip_zero16);
    const bf16_raw* sp0 = reinterpret_cast<const bf16_raw*>(d.src[0].ptr);
    const bf16_raw* sp1 = reinterpret_cast<const bf16_raw*>(d.src[1].ptr);
    auto gather = [&](const sehip_kchunk e) {
        RegTile<NRA> t;
        if (narrow) {
#pragma unroll
            for (int i = 0; i < NRA; ++i) t.v[i] = gather_chunk(d.src[0], d.src[1], e, rp[i], rb0[i], rb1[i]);
            return t;
        }
        const bool second = e.src > 0;
        const int toff = e.toff >> 16, fadd = (int)(short)(e.toff & 0xffff);
        const int tlo = second ? d.src[1].tlo : d.src[0].tlo, thi = second ? d.src[1].thi : d.src[0].thi;
        const unsigned F = (unsigned)(second ? d.src[1].F : d.src[0].F);
        const bf16_raw* base = (second ? sp1 : sp0) + e.fadd;        // e.fadd = element delta of the chunk
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int ts = rp[i].ts + toff;
            const bool ok = rp[i].valid && e.src >= 0 && ts >= tlo && ts < thi && (unsigned)(rp[i].jf + fadd) < F;
            const bf16_raw* q = ok ? base + (second ? rb1[i] : rb0[i]) : zero_page;
            t.v[i] = *reinterpret_cast<const uint4*>(q);
        }
        return t;
    };
    const bf16_raw* wrow[NRW];
#pragma unroll
    for (int i = 0; i < NRW; ++i) wrow[i] = Wb + (size_t)(n0 + min(r0 + 32 * i, BN - 1)) * d.K + kc * 8;
    // (two tiles of at most four pieces: returned by value as ONE six-piece tile, the weights of the 192-column instantiation went
    //  through scratch memory inside the K loop and the kernel ran three times slower)
    constexpr int NRW0 = NRW < 4 ? NRW : 4, NRW1 = NRW > 4 ? NRW - 4 : 0;
    auto fetch_w = [&](int kt) {
        RegTile<NRW0> t;
#pragma unroll
        for (int i = 0; i < NRW0; ++i) t.v[i] = *reinterpret_cast<const uint4*>(wrow[i] + kt * 64);   // rows >= BN: a duplicate, never stored
        return t;
    };
    auto fetch_w1 = [&](int kt) {
        RegTile<(NRW1 > 0 ? NRW1 : 1)> t;
#pragma unroll
        for (int i = 0; i < NRW1; ++i) t.v[i] = *reinterpret_cast<const uint4*>(wrow[4 + i] + kt * 64);
        return t;
    };

    sehip_kchunk e1 = d.ktab[min((kt0 + 1) * 8 + kc, (d.K >> 3) - 1)];          // entry of the second K step
    RegTile<NRA> ra = gather(d.ktab[kt0 * 8 + kc]);
    RegTile<NRW0> rw = fetch_w(kt0);
    RegTile<(NRW1 > 0 ? NRW1 : 1)> rw1;
    if (NRW1 > 0) rw1 = fetch_w1(kt0);
    for (int kt = kt0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int r = r0 + 32 * i;
            sA[r * 8 + (kc ^ (r & 7))] = ra.v[i];
        }
#pragma unroll
        for (int i = 0; i < NRW0; ++i) {
            const int r = r0 + 32 * i;
            if (r < BN) sW[r * 8 + (kc ^ (r & 7))] = rw.v[i];
        }
#pragma unroll
        for (int i = 0; i < NRW1; ++i) {
            const int r = r0 + 32 * (4 + i);
            if (r < BN) sW[r * 8 + (kc ^ (r & 7))] = rw1.v[i];
        }
        __syncthreads();
        if (kt + 1 < nk) {  // next K tile in flight behind this tile's MFMAs
            ra = gather(e1);
            rw = fetch_w(kt + 1);
            if (NRW1 > 0) rw1 = fetch_w1(kt + 1);
            e1 = d.ktab[min((kt + 2) * 8 + kc, (d.K >> 3) - 1)];
        }
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

    // epilogue: lane holds n = nb + 4*(lane>>4) + {0..3}, m = mb + (lane&15).
    if (part) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m0 + wm * (BM / WM) + mi * 16 + (lane & 15);
            if (m >= d.M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wn * (BN / WN) + ni * 16 + 4 * (lane >> 4);
                const f32x4 v = acc[ni][mi];
                *reinterpret_cast<float4*>(part + (size_t)m * d.Npad + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        return;
    }
    // Dense case (the wave's WC columns are one contiguous run of a bf16 destination): the tile leaves through a wave-private
    // LDS image as 16-byte pieces, a row's WC columns in consecutive lanes, rows stepped with carries -- instead of 8-byte
    // pieces scattered over 16 rows per store instruction and two divisions per row.
    {
        const int nw0 = n0 + wn * WC;
        const sehip_nchunk first = d.ntab[nw0 >> 2];
        const sehip_nchunk mine = d.ntab[(nw0 >> 2) + (lane % (WC / 4))];
        const bool ok = mine.nvalid == 4 && mine.dst == first.dst && mine.coff == first.coff + 4 * (lane % (WC / 4));
        const sehip_dst& dd = first.dst ? d.dst[1] : d.dst[0];
        const bool dense = WC >= 8 && __all(ok) && !dd.is_f32 && (first.coff & 7) == 0 && (dd.C & 7) == 0;
        if (dense) {
            if (d.res && first.dst == 0) {
                // residual: added to the fp32 accumulators BEFORE the one rounding to bf16 (the residual stream of ConvTasNet
                // passes 14 of these in a row); rows mi*16 + (lane & 15) of the wave, stepped 16 at a time
                RowPos cur = row_pos(m0 + wm * WR + (lane & 15), d.M, d.TT, d.J, d.fmul);
                const int st = 16 / d.J, sj = 16 - st * d.J;
                const bf16_raw* rp0 = reinterpret_cast<const bf16_raw*>(d.res) + first.coff + 4 * (lane >> 4);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    if (m0 + wm * WR + mi * 16 + (lane & 15) < d.M) {
                        const size_t off = dst_row_offset(dd, cur, d.fmul);
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni) add_res4(acc[ni][mi], rp0, off + ni * 16);
                    }
                    cur.j += sj; cur.t += st;
                    if (cur.j >= d.J) { cur.j -= d.J; ++cur.t; }
                    for (; cur.t >= d.TT; cur.t -= d.TT) ++cur.b;
                }
            }
            bf16_raw* img = reinterpret_cast<bf16_raw*>(sbuf) + wave * (WR * TP);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (d.bias) bv = *reinterpret_cast<const float4*>(d.bias + nw0 + ni * 16 + 4 * (lane >> 4));
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const f32x4 v = acc[ni][mi];
                    *reinterpret_cast<uint2*>(&img[(mi * 16 + (lane & 15)) * TP + ni * 16 + 4 * (lane >> 4)]) =
                        make_uint2(pack_bf2(v[0] + bv.x, v[1] + bv.y), pack_bf2(v[2] + bv.z, v[3] + bv.w));
                }
            }
            constexpr int PPR = WC / 8, RPI = 64 / PPR;      // 16-byte pieces per row, rows per pass of the wave
            const int c8 = lane % PPR, rl = lane / PPR;
            bf16_raw* dptr = reinterpret_cast<bf16_raw*>(dd.ptr) + first.coff + c8 * 8;
            const int mrow0 = m0 + wm * WR + rl;
            RowPos cur = row_pos(mrow0, d.M, d.TT, d.J, d.fmul);
            const int st = RPI / d.J, sj = RPI - st * d.J;
#pragma unroll
            for (int it = 0; it < (WR + RPI - 1) / RPI; ++it) {
                const int row = rl + RPI * it;
                if (row < WR && mrow0 + RPI * it < d.M) {
                    const uint4 v = *reinterpret_cast<const uint4*>(&img[row * TP + c8 * 8]);
                    *reinterpret_cast<uint4*>(dptr + dst_row_offset(dd, cur, d.fmul)) = v;
                }
                cur.j += sj; cur.t += st;
                if (cur.j >= d.J) { cur.j -= d.J; ++cur.t; }
                for (; cur.t >= d.TT; cur.t -= d.TT) ++cur.b;
            }
            return;
        }
    }
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
            store_out4(d, nc, acc[ni][mi], ro0, ro1, n);
        }
    }
}

template <int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(256) void gemm_kernel(const sehip_gemm_desc d) {
    gemm_body<BN, BM, WN, WM>(d);
}

// split-K: blockIdx.y sums K steps [y * per, (y + 1) * per) into its own fp32 image of the output; the finish kernel adds the images in
// order and stores through the ordinary epilogue (bias, residual, destination table).  For very short row spaces (Demucs' deepest
// levels, src/model/demucs.py:386-413 at 736 rows: 2048-4096 x 6144-12288 weights streamed by under one workgroup per CU, each a
// chain of 100-200 dependent K steps: 0.3 TB/s of weights).
template <int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(256) void gemm_splitk_kernel(const sehip_gemm_desc d, int per, float* __restrict__ part) {
    const int nk = d.K >> 6;
    const int k0 = (int)blockIdx.y * per, k1 = min(nk, k0 + per);
    gemm_body<BN, BM, WN, WM>(d, k0, k1, part + (size_t)blockIdx.y * d.M * d.Npad);
}
__global__ __launch_bounds__(256) void gemm_splitk_finish_kernel(const sehip_gemm_desc d, const float* __restrict__ part, int nsplit) {
    const int n4 = d.Npad >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)d.M * n4) return;
    const int m = (int)(idx / n4), n = 4 * (int)(idx - (long)m * n4);
    const sehip_nchunk nc = d.ntab[n >> 2];
    if (nc.nvalid <= 0) return;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(part + ((size_t)s * d.M + m) * d.Npad + n);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    }
    const RowPos r = row_pos(m, d.M, d.TT, d.J, d.fmul);
    const size_t ro0 = dst_row_offset(d.dst[0], r, d.fmul);
    const size_t ro1 = d.dst[1].ptr ? dst_row_offset(d.dst[1], r, d.fmul) : 0;
    store_out4(d, nc, v, ro0, ro1, n);
}

// two products of the same shape in one launch (blockIdx.y picks the descriptor): the real / imaginary halves of the LSTM
// input, projection and input-gradient products are 10-20 us kernels that are mostly launch and pipeline fill
struct GemmPair { sehip_gemm_desc d[2]; };
template <int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(256) void gemm_pair_kernel(const GemmPair p) {
    // indexed in the kernel-argument segment (a select between two by-value descriptors was copied to scratch)
    gemm_body<BN, BM, WN, WM>(p.d[blockIdx.y]);
}

// ------------------------------------------------------------------------------------------------
// conv_gemm_kernel: same product for descriptors that carry the regular-convolution description (cv_*).
// The generic kernel re-gathers every input element once per tap from L2/HBM (10x for a 5x2 kernel) and that
// stream, not the MFMA, bounds it.  Here a workgroup owns 128 output rows = TB frames x JB rows of ONE utterance
// and, per 64-channel chunk, stages the input PATCH ((TB+1) frames x ((JB-1)*fmul+NF) rows x 64 ch) in LDS once;
// all 2*NF taps read their MFMA operand fragments straight from that patch (row pitch 144 B).  Only the weight
// tile is staged per K step.  The next patch chunk is prefetched into registers a few 16-byte pieces per K step.
// ------------------------------------------------------------------------------------------------
#define CV_PITCH 72   // bf16 elements per patch row (64 + 8 pad)
#define CV_MAXP 12    // 16-byte patch pieces per thread (patch <= 48 KB)

template <int BN, int WN, int WM, int NF>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const sehip_gemm_desc d, int TB, int JB, int FR) {
    constexpr int BM = 128;
    constexpr int TN = BN / WN / 16, TM = BM / WM / 16;
    constexpr int NRW = BN / 32;
    constexpr int NIT = 2 * NF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* sW = reinterpret_cast<uint4*>(smem);
    bf16_raw* patch = reinterpret_cast<bf16_raw*>(sW + BN * 8);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    const int ntn = d.Npad / BN;
    const int tblocks = (d.TT + TB - 1) / TB, jblocks = d.J / JB;
    // XCD-contiguous order, n-tile fastest (see gemm_kernel)
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int nt = logical % ntn;
    int rest = logical / ntn;
    const int jb = rest % jblocks; rest /= jblocks;
    const int tb = rest % tblocks;
    const int b = rest / tblocks;
    const int t0 = tb * TB, j0 = jb * JB, n0 = nt * BN;
    const int f0 = j0 * d.fmul + d.cv_fadd;

    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int Ctot = C0 + C1;
    const int ncc = Ctot >> 6;
    const int tmin0 = min(d.cv_toff[0][0], d.cv_toff[0][1]), tmin1 = min(d.cv_toff[1][0], d.cv_toff[1][1]);
    const int NP = (TB + 1) * FR * 8;

    auto fetch_piece = [&](int cc, int i) -> uint4 {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const int idx = tid + 256 * i;
        if (idx < NP) {
            const int p = idx / (FR * 8), rem = idx - p * (FR * 8);
            const int r = rem >> 3, c8 = rem & 7;
            const bool second = cc * 64 >= C0;
            const int sT = second ? d.src[1].T : d.src[0].T, sF = second ? d.src[1].F : d.src[0].F;
            const int sC = second ? C1 : C0;
            const int tlo = second ? d.src[1].tlo : d.src[0].tlo, thi = second ? d.src[1].thi : d.src[0].thi;
            const bf16_raw* base = reinterpret_cast<const bf16_raw*>(second ? d.src[1].ptr : d.src[0].ptr);
            const int ts = t0 + p + (second ? tmin1 : tmin0);
            const int f = f0 + r;
            if (ts >= tlo && ts < thi && f >= 0 && f < sF) {
                const bf16_raw* g = base + ((((long)b * sT + ts) * sF + f) * sC + (cc * 64 - (second ? C0 : 0)) + c8 * 8);
                v = *reinterpret_cast<const uint4*>(g);
            }
        }
        return v;
    };
    const bf16_raw* Wb = reinterpret_cast<const bf16_raw*>(d.W);
    const int kc = tid & 7, r0 = tid >> 3;

    // per-lane patch offsets of its TM activation rows (element units, k-chunk of the lane included)
    int abase[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int r = wm * (BM / WM) + mi * 16 + (lane & 15);
        const int tl = r / JB, jl = r - tl * JB;
        abase[mi] = (tl * FR + jl * d.fmul) * CV_PITCH + 8 * (lane >> 4);
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int bb = 0; bb < TM; ++bb) acc[a][bb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    RegTile<NRW> rw = fetch_w_tile<NRW, BN>(Wb, d.K, n0, r0, kc * 8);

    for (int cc = 0; cc < ncc; ++cc) {
        // stage the patch of this channel chunk (all reads of the previous one ended at the last barrier); the second
        // workgroup resident on the CU computes meanwhile
#pragma unroll
        for (int i0 = 0; i0 < CV_MAXP; i0 += 4) {
            uint4 pr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) pr[q] = fetch_piece(cc, i0 + q);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + 256 * (i0 + q);
                if (idx < NP) {
                    const int p = idx / (FR * 8), rem = idx - p * (FR * 8);
                    *reinterpret_cast<uint4*>(&patch[(p * FR + (rem >> 3)) * CV_PITCH + (rem & 7) * 8]) = pr[q];
                }
            }
        }
        const bool second = cc * 64 >= C0;
        const int dt0 = (second ? d.cv_toff[1][0] - tmin1 : d.cv_toff[0][0] - tmin0);
        const int dt1 = (second ? d.cv_toff[1][1] - tmin1 : d.cv_toff[0][1] - tmin0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int i = 0; i < NRW; ++i) {
                const int r = r0 + 32 * i;
                sW[r * 8 + (kc ^ (r & 7))] = rw.v[i];
            }
            __syncthreads();
            // next weight tile goes in flight behind this step's MFMAs
            if (it + 1 < NIT) rw = fetch_w_tile<NRW, BN>(Wb, d.K, n0, r0, (it + 1) * Ctot + cc * 64 + kc * 8);
            else if (cc + 1 < ncc) rw = fetch_w_tile<NRW, BN>(Wb, d.K, n0, r0, (cc + 1) * 64 + kc * 8);
            const int kt = it / NF, tap = it - kt * NF;
            const int toff_e = ((kt ? dt1 : dt0) * FR + tap) * CV_PITCH;
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
                for (int mi = 0; mi < TM; ++mi)
                    af[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&patch[abase[mi] + toff_e + 32 * ks]));
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    // ---- epilogue.  A lane holds 4 consecutive channels of a row, so the direct scatter writes 8-byte pieces (32 bytes per
    // row and instruction).  When the wave's 64 output channels are one dense run of a bf16 destination, the tile goes
    // through a wave-private LDS image instead and leaves as 16-byte pieces, 128 contiguous bytes per row.
    constexpr int WROWS = BM / WM, WCOLS = BN / WN, TP = WCOLS + 8;
    const int nw0 = n0 + wn * WCOLS;
    bool dense = WCOLS == 64;
    sehip_nchunk first = d.ntab[nw0 >> 2];
    {
        const sehip_nchunk mine = d.ntab[(nw0 >> 2) + (lane & 15)];
        const bool ok = mine.nvalid == 4 && mine.dst == first.dst && mine.coff == first.coff + 4 * (lane & 15);
        dense = dense && __all(ok) && !(first.dst ? d.dst[1].is_f32 : d.dst[0].is_f32) && ((first.coff & 7) == 0) &&
                (((first.dst ? d.dst[1].C : d.dst[0].C) & 7) == 0);
    }
    __syncthreads();  // every wave has finished reading the weight tile and the patch: the LDS is free (unconditional: the
                      // waves of a workgroup may disagree about `dense`)
    if (dense) {
        bf16_raw* tb_ = reinterpret_cast<bf16_raw*>(smem) + wave * (WROWS * TP);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (d.bias) bv = *reinterpret_cast<const float4*>(d.bias + nw0 + ni * 16 + 4 * (lane >> 4));
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const f32x4 v = acc[ni][mi];
                *reinterpret_cast<uint2*>(&tb_[(mi * 16 + (lane & 15)) * TP + ni * 16 + 4 * (lane >> 4)]) =
                    make_uint2(pack_bf2(v[0] + bv.x, v[1] + bv.y), pack_bf2(v[2] + bv.z, v[3] + bv.w));
            }
        }
        // wave-private image: the LDS queue of a wave is in order, no barrier needed between its writes and reads
        const sehip_dst& dd = first.dst ? d.dst[1] : d.dst[0];
        bf16_raw* dptr = reinterpret_cast<bf16_raw*>(dd.ptr) + first.coff;
#pragma unroll
        for (int it = 0; it < WROWS / 8; ++it) {
            const int row = it * 8 + (lane >> 3), c8 = lane & 7;
            const int rr = wm * WROWS + row;
            const int tl = rr / JB, jl = rr - tl * JB;
            RowPos r;
            r.b = b; r.t = t0 + tl; r.j = j0 + jl; r.jf = r.j * d.fmul; r.valid = r.t < d.TT;
            uint4 v = *reinterpret_cast<const uint4*>(&tb_[row * TP + c8 * 8]);
            if (r.valid) {
                const size_t off = dst_row_offset(dd, r, d.fmul) + c8 * 8;
                if (d.res && first.dst == 0)
                    v = add_bf16x8(v, *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_raw*>(d.res) + first.coff + off));
                *reinterpret_cast<uint4*>(dptr + off) = v;
            }
        }
        return;
    }
    // direct scatter (same as gemm_kernel)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int rr = wm * (BM / WM) + mi * 16 + (lane & 15);
        const int tl = rr / JB, jl = rr - tl * JB;
        RowPos r;
        r.b = b; r.t = t0 + tl; r.j = j0 + jl; r.jf = r.j * d.fmul; r.valid = r.t < d.TT;
        if (!r.valid) continue;
        const size_t ro0 = dst_row_offset(d.dst[0], r, d.fmul);
        const size_t ro1 = d.dst[1].ptr ? dst_row_offset(d.dst[1], r, d.fmul) : 0;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * (BN / WN) + ni * 16 + 4 * (lane >> 4);
            const sehip_nchunk nc = d.ntab[n >> 2];
            if (nc.nvalid <= 0) continue;
            store_out4(d, nc, acc[ni][mi], ro0, ro1, n);
        }
    }
}

template <int BN, int WN, int WM>
static int launch_conv(const sehip_gemm_desc& d, int TB, int JB, int FR, int grid, size_t lds, hipStream_t st) {
#define CV_CASE(NF_)                                                                                                  \
    case NF_: {                                                                                                       \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<BN, WN, WM, NF_>),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                               \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        sehip_note_kernel("conv_gemm_kernel<%d, %d, %d, %d>", BN, WN, WM, NF_);                                        \
        conv_gemm_kernel<BN, WN, WM, NF_><<<grid, 256, lds, st>>>(d, TB, JB, FR);                                     \
        return 1;                                                                                                     \
    }
    switch (d.cv_nf) {
        CV_CASE(2)
        CV_CASE(3)
        CV_CASE(5)
        default: return 0;
    }
#undef CV_CASE
}

int sehip_try_conv_gemm_v2(const sehip_gemm_desc& d, hipStream_t st);   // conv2.hip
void sehip_conv2_init(void);
int sehip_try_conv_gemm_v3(const sehip_gemm_desc& d, hipStream_t st);   // conv3.hip
int sehip_try_conv_gemm_v3_pair(const sehip_gemm_desc& a, const sehip_gemm_desc& b, hipStream_t st);
void sehip_conv3_init(void);
int sehip_try_conv_wgrad_v3(const sehip_gemm_desc& d, hipStream_t st);   // wgrad3.hip
int sehip_try_convs_stream(const sehip_gemm_desc& a, hipStream_t st, bool dry);   // convt.hip
int sehip_try_wgrads_stream(const sehip_gemm_desc& a, hipStream_t st);            // convt.hip
int sehip_try_wgradt_stream(const sehip_gemm_desc& a, const sehip_gemm_desc& b, hipStream_t st);   // convt.hip
void sehip_wgrad3_init(void);
int sehip_try_dense_wgrad(const sehip_gemm_desc& d, hipStream_t st);   // wgrad3.hip
int sehip_try_dense_rows_gemm(const sehip_gemm_desc& d, hipStream_t st);   // dgemm.hip

// returns 1 if the LDS-patch kernel was launched, 0 if the descriptor does not qualify
static int try_conv_gemm(const sehip_gemm_desc& d, hipStream_t st) {
    static const bool disabled = getenv("SEHIP_NO_PATCH") != nullptr;
    if (disabled || d.cv_nf <= 0) return 0;
    if (sehip_try_conv_gemm_v3(d, st)) return 1;
    if (d.w_tiled) return 0;                     // nobody else reads the tile order (sehip_gemm reports it)
    if (sehip_try_conv_gemm_v2(d, st)) return 1;
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    if ((C0 & 63) || (C1 & 63) || (d.Npad & 63) || d.J > 64 || (128 % d.J)) return 0;
    if (d.K != 2 * d.cv_nf * (C0 + C1)) return 0;
    const int JB = d.J, TB = 128 / JB;
    const int FR = (JB - 1) * d.fmul + d.cv_nf;
    const size_t patch_bytes = (size_t)(TB + 1) * FR * CV_PITCH * 2;
    if ((TB + 1) * FR * 8 > CV_MAXP * 256) return 0;
    const int B = d.M / (d.TT * d.J);
    const int tblocks = (d.TT + TB - 1) / TB;
    if ((d.Npad & 127) == 0) {
        const int grid = B * tblocks * (d.Npad / 128);
        return launch_conv<128, 2, 2>(d, TB, JB, FR, grid, 128 * 128 + patch_bytes, st);
    }
    const int grid = B * tblocks * (d.Npad / 64);
    return launch_conv<64, 1, 4>(d, TB, JB, FR, grid, 64 * 128 + patch_bytes, st);
}

// ------------------------------------------------------------------------------------------------
// conv_small2_kernel: small-channel layers (<= 128 output channels, <= 128 concatenated input channels: the outer
// encoder/decoder layers, HBM-bound).  The whole packed weight [BN][K] lives in LDS for the lifetime of the workgroup,
// which walks over tiles of 64 MI rows: stage the input patch (every element read once), run all K/32 MFMA steps
// straight from LDS, store.  A first version ran the phases of a tile one after the other with divergent staging code;
// 30-45 % of its time was that instruction stream.  Here the next tile's patch is fetched into registers
// (branch-free, see sw_fetch_patch) while the current one is multiplied and stored, the store addressing (row
// offsets, n-chunk table entries, bias) is computed once per workgroup, and the K loop walks taps with scalar offsets.
// ------------------------------------------------------------------------------------------------
// Fused ComplexBatchNorm sums of the small-channel kernels (sehip_gemm_desc.stats; conv_small2_kernel, conv_narrow_kernel), from
// the accumulators, of the values as they are stored.  A lane holds D[row (lane & 15) of tile mi][columns 16 ni + 4 g ..].
// 32 outputs (TN = 2): column tile 0 = real parts, tile 1 = imaginary parts of complex channels 4 g .. 4 g + 3.  16 outputs
// (TN = 1): k groups 0-1 hold the real parts of channels 4 g .., groups 2-3 the imaginary parts of channels 4 (g - 2) ..: the
// lower half fetches its partner's values (lane + 32, same row); the upper half accumulates garbage that is never flushed.
// st[5 q + k]: sum re, sum im, sum re^2, sum re im, sum im^2 of channel 4 g + q.
template <int TN, int MI>
__device__ __forceinline__ void small_stats_add(const f32x4 (&acc)[TN][MI], const float4 (&bias4)[TN], const int (&e_tl)[MI], int t0,
                                                int TT, float (&st)[20]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const bool ok = t0 + e_tl[mi] < TT;
        const f32x4 vr = acc[0][mi], vi = acc[TN - 1][mi];
        const float br[4] = {bias4[0].x, bias4[0].y, bias4[0].z, bias4[0].w};
        const float bi[4] = {bias4[TN - 1].x, bias4[TN - 1].y, bias4[TN - 1].z, bias4[TN - 1].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float yr = bf2f(f2bf(vr[q] + br[q]));
            float yi = bf2f(f2bf(vi[q] + bi[q]));
            if (TN == 1) yi = __shfl_xor(yi, 32, 64);
            if (ok) {
                st[5 * q] += yr; st[5 * q + 1] += yi;
                st[5 * q + 2] += yr * yr; st[5 * q + 3] += yr * yi; st[5 * q + 4] += yi * yi;
            }
        }
    }
}
// end of the workgroup (after a barrier; sred: 320 floats of LDS nobody uses any more): the 16 lanes of a k group hold different rows
// -> xor-shuffles; the four waves meet in LDS; 20 atomics per k group that holds sums (4 for 32 outputs, the lower 2 for 16)
template <int TN>
__device__ __forceinline__ void small_stats_flush(float (&st)[20], float* sred, float* stats, int Cr, int tid) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int i = 0; i < 20; ++i)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) st[i] += __shfl_xor(st[i], o, 64);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 20; ++i) sred[(w * 4 + (lane >> 4)) * 20 + i] = st[i];
    }
    __syncthreads();
    if (tid < 40 * TN) {
        const int gg = tid / 20, i = tid - gg * 20, q = i / 5, k = i - 5 * q;
        const float v = sred[gg * 20 + i] + sred[(4 + gg) * 20 + i] + sred[(8 + gg) * 20 + i] + sred[(12 + gg) * 20 + i];
        atomicAdd(stats + (size_t)(blockIdx.x & 7) * 5 * Cr + k * Cr + 4 * gg + q, v);
    }
}

// Everything of one product over a staged patch that depends on the descriptor: weights in LDS, tap geometry, store
// addressing.  A launch carries one side, or two (the two output-row parities of a transposed convolution read the same
// input: staged once, multiplied by both weight sets -- see sehip_gemm_pair).
template <int TN, int MI>
struct Cs2Side {
    const bf16_raw* sW;
    int KP, NIT, nf, KR;
    int dt00, dt01, dt10, dt11;  // patch frame of (source, kt); scalars: a dynamically indexed member array lives in scratch
    int fshift;            // rows between the patch's first row and this side's first tap
    int e_off0[MI], e_off1[MI];
    long bs0, bs1;
    int ts0, ts1;
    sehip_nchunk nck[TN];
    float4 bias4[TN];
    // dense bf16 destination (all TN*16 channels one contiguous run): the tile leaves through a wave-private LDS image as
    // 16-byte pieces; o_off[it] = destination offset (b 0, tile frame 0) of the row of piece lane + 64 it, o_tl its tile frame
    bool dense;
    int o_off[(MI * TN + 1) / 2], o_tl[(MI * TN + 1) / 2];
};

template <int TN, int MI>
__device__ __forceinline__ Cs2Side<TN, MI> cs2_side_init(const sehip_gemm_desc& d, const bf16_raw* sW, int tmin0, int tmin1,
                                                          int fadd_u, int JB, int w, int lane) {
    Cs2Side<TN, MI> sd;  // returned by value: filled through a reference it stayed in scratch
    sd.sW = sW;
    sd.KP = d.K + 8;
    sd.nf = d.cv_nf;
    sd.NIT = 2 * d.cv_nf;
    sd.KR = sd.NIT * (d.src[0].C + (d.src[1].ptr ? d.src[1].C : 0));
    sd.dt00 = d.cv_toff[0][0] - tmin0; sd.dt01 = d.cv_toff[0][1] - tmin0;
    sd.dt10 = d.cv_toff[1][0] - tmin1; sd.dt11 = d.cv_toff[1][1] - tmin1;
    sd.fshift = d.cv_fadd - fadd_u;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int r = 16 * MI * w + mi * 16 + (lane & 15);
        const int tl = r / JB, jl = r - tl * JB;
        RowPos rp;
        rp.b = 0; rp.t = tl; rp.j = jl; rp.jf = jl * d.fmul; rp.valid = true;
        sd.e_off0[mi] = (int)dst_row_offset(d.dst[0], rp, d.fmul);
        sd.e_off1[mi] = d.dst[1].ptr ? (int)dst_row_offset(d.dst[1], rp, d.fmul) : 0;
    }
    sd.bs0 = (long)d.dst[0].T * d.dst[0].F * d.dst[0].C; sd.bs1 = (long)d.dst[1].T * d.dst[1].F * d.dst[1].C;
    sd.ts0 = d.dst[0].F * d.dst[0].C; sd.ts1 = d.dst[1].F * d.dst[1].C;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = ni * 16 + 4 * (lane >> 4);
        sd.nck[ni] = d.ntab[n >> 2];
        sd.bias4[ni] = d.bias ? *reinterpret_cast<const float4*>(d.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        constexpr int CH = TN * 4;                      // 4-channel chunks of a row
        const sehip_nchunk first = d.ntab[0];
        const sehip_nchunk mine = d.ntab[lane % CH];
        const bool ok = mine.nvalid == 4 && mine.dst == first.dst && mine.coff == first.coff + 4 * (lane % CH);
        const sehip_dst& dd = first.dst ? d.dst[1] : d.dst[0];
        sd.dense = TN >= 2 && TN <= 4 && __all(ok) &&   // 128 outputs: the staging cost more registers than it saved (82 -> 100 us)
                   !dd.is_f32 && (first.coff & 7) == 0 && (dd.C & 7) == 0;
        constexpr int PPR = TN * 2;                     // 16-byte pieces per row
#pragma unroll
        for (int it = 0; it < (MI * TN + 1) / 2; ++it) {
            const int row = (lane + 64 * it) / PPR, c8 = (lane + 64 * it) - row * PPR;
            const int r = 16 * MI * w + row;
            const int tl = r / JB, jl = r - tl * JB;
            RowPos rp;
            rp.b = 0; rp.t = tl; rp.j = jl; rp.jf = jl * d.fmul; rp.valid = true;
            sd.o_off[it] = (int)dst_row_offset(dd, rp, d.fmul) + first.coff + c8 * 8;
            sd.o_tl[it] = tl;
        }
    }
    return sd;
}

// weights -> LDS (once per workgroup)
template <int BN>
__device__ __forceinline__ void cs2_load_weights(const sehip_gemm_desc& d, bf16_raw* sW, int tid) {
    const bf16_raw* Wb = reinterpret_cast<const bf16_raw*>(d.W);
    const int cpr = d.K >> 3, KP = d.K + 8;
    for (int idx = tid; idx < BN * cpr; idx += 256) {
        const int r = idx / cpr, c = idx - r * cpr;
        *reinterpret_cast<uint4*>(&sW[r * KP + c * 8]) = *reinterpret_cast<const uint4*>(Wb + (size_t)r * d.K + c * 8);
    }
}

// all K steps of one side over the staged patch, then its stores
template <int TN, int MI>
__device__ __forceinline__ void cs2_multiply_store(const Cs2Side<TN, MI> sd, const sehip_gemm_desc& d, const bf16_raw* patch,
                                                   bf16_raw* obuf /* wave-private [16 MI][16 TN + 8] */, const int (&abase)[MI],
                                                   const int (&e_tl)[MI], int FR, int PP, int CT, int C0, int b, int t0, int lane,
                                                   float (&st)[20], bool do_stats) {
    const int g = lane >> 4;
    const int wrow = (lane & 15) * sd.KP + 8 * g;  // this lane's weight fragment: row (lane & 15) of each 16-row tile, k chunk g
    f32x4 acc[TN][MI];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[a][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (CT >= 32) {
        // software pipeline over the K steps (tap it, 32-channel slice cs): the fragments of step s+1 are requested before
        // the MFMAs of step s (the loop is too long to unroll and the compiler does not rotate it by itself)
        const int cs_n = CT >> 5, lgcs = 31 - __clz(cs_n);
        const int steps = sd.NIT << lgcs;
        bf16x8 af[MI], wf[TN], afn[MI], wfn[TN];
#define CS2_LOAD(s_, af_, wf_)                                                                                         \
        {                                                                                                              \
            const int it_ = (s_) >> lgcs, cs_ = (s_) & (cs_n - 1);                                                     \
            const int kt_ = it_ >= sd.nf ? 1 : 0, tap_ = it_ - kt_ * sd.nf;                                            \
            const int c_ = 32 * cs_ + 8 * g;                                                                           \
            const int dt_ = c_ >= C0 ? (kt_ ? sd.dt11 : sd.dt10) : (kt_ ? sd.dt01 : sd.dt00);                          \
            const int poff_ = (dt_ * FR + tap_ + sd.fshift) * PP + c_;                                                 \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                          \
                af_[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&patch[abase[mi] + poff_]));      \
            _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)                                                          \
                wf_[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&sd.sW[ni * 16 * sd.KP + wrow + it_ * CT + 32 * cs_])); \
        }
        CS2_LOAD(0, af, wf)
        for (int s = 0; s < steps; ++s) {
            const int sn = s + 1 < steps ? s + 1 : s;
            CS2_LOAD(sn, afn, wfn)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = afn[mi];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) wf[ni] = wfn[ni];
        }
#undef CS2_LOAD
    } else {
        const int ksteps = sd.KR >> 5;
        const int lgct = 31 - __clz(CT);
        for (int s = 0; s < ksteps; ++s) {
            const int k = 32 * s + 8 * g;
            const int it = k >> lgct, c = k & (CT - 1);
            const int kt = it >= sd.nf ? 1 : 0, tap = it - kt * sd.nf;
            const int dtv = c >= C0 ? (kt ? sd.dt11 : sd.dt10) : (kt ? sd.dt01 : sd.dt00);
            const int poff = (dtv * FR + tap + sd.fshift) * PP + c;
            bf16x8 af[MI], wf[TN];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&patch[abase[mi] + poff]));
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                wf[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&sd.sW[ni * 16 * sd.KP + wrow + 32 * s]));
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    if constexpr (TN <= 2) {
        if (do_stats) small_stats_add<TN, MI>(acc, sd.bias4, e_tl, t0, d.TT, st);
    }
    if (sd.dense) {
        constexpr int TP = 16 * TN + 8;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const f32x4 v = acc[ni][mi];
                *reinterpret_cast<uint2*>(&obuf[(mi * 16 + (lane & 15)) * TP + ni * 16 + 4 * g]) =
                    make_uint2(pack_bf2(v[0] + sd.bias4[ni].x, v[1] + sd.bias4[ni].y), pack_bf2(v[2] + sd.bias4[ni].z, v[3] + sd.bias4[ni].w));
            }
        const sehip_nchunk first = sd.nck[0];
        bf16_raw* dptr = reinterpret_cast<bf16_raw*>(first.dst ? d.dst[1].ptr : d.dst[0].ptr);
        const long tile_off = first.dst ? b * sd.bs1 + (long)t0 * sd.ts1 : b * sd.bs0 + (long)t0 * sd.ts0;
        constexpr int PPR = TN * 2;
#pragma unroll
        for (int it = 0; it < (MI * TN + 1) / 2; ++it) {
            const int idx = lane + 64 * it;
            const int row = idx / PPR, c8 = idx - row * PPR;
            if (idx < 16 * MI * PPR && t0 + sd.o_tl[it] < d.TT) {
                uint4 v = *reinterpret_cast<const uint4*>(&obuf[row * TP + c8 * 8]);
                if (d.res && first.dst == 0)
                    v = add_bf16x8(v, *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_raw*>(d.res) + tile_off + sd.o_off[it]));
                *reinterpret_cast<uint4*>(dptr + tile_off + sd.o_off[it]) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if (t0 + e_tl[mi] >= d.TT) continue;
        const long ro0 = sd.e_off0[mi] + b * sd.bs0 + (long)t0 * sd.ts0;
        const long ro1 = sd.e_off1[mi] + b * sd.bs1 + (long)t0 * sd.ts1;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const sehip_nchunk nc = sd.nck[ni];
            if (nc.nvalid <= 0) continue;
            f32x4 v = acc[ni][mi];
            v[0] += sd.bias4[ni].x; v[1] += sd.bias4[ni].y; v[2] += sd.bias4[ni].z; v[3] += sd.bias4[ni].w;
            const long off = (nc.dst ? ro1 : ro0) + nc.coff;
            void* dptr = nc.dst ? d.dst[1].ptr : d.dst[0].ptr;
            const int is_f32 = nc.dst ? d.dst[1].is_f32 : d.dst[0].is_f32;
            if (d.res && nc.dst == 0 && nc.nvalid == 4) add_res4(v, d.res, (size_t)off);
            if (is_f32) {
                float* q = reinterpret_cast<float*>(dptr) + off;
                if (nc.nvalid == 4) *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int e = 0; e < nc.nvalid; ++e) q[e] = v[e];
            } else {
                bf16_raw* q = reinterpret_cast<bf16_raw*>(dptr) + off;
                if (nc.nvalid == 4) *reinterpret_cast<uint2*>(q) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                else
                    for (int e = 0; e < nc.nvalid; ++e) q[e] = f2bf(v[e]);
            }
        }
    }
}

// d2 is read only when PAIR; fadd_u = first input row of the patch relative to row j*fmul (min of the sides' cv_fadd)
template <int BN, int NPC, int MI, bool PAIR>
__global__ __launch_bounds__(256) void conv_small2_kernel(const sehip_gemm_desc d, const sehip_gemm_desc d2, int TB, int JB, int FR,
                                                          int fadd_u, int tiles_per_wg) {
    constexpr int TN = BN / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int CT = C0 + C1;
    const int PP = CT + 8;            // LDS pitch of a patch row
    bf16_raw* sW = reinterpret_cast<bf16_raw*>(smem);
    bf16_raw* sW2 = sW + BN * (d.K + 8);
    bf16_raw* patch = PAIR ? sW2 + BN * (d2.K + 8) : sW2;
    // output staging image of this wave, behind the patch and its 16-byte dump slot
    bf16_raw* obuf = patch + (TB + 1) * FR * PP + 8 + (threadIdx.x >> 6) * (16 * MI * (BN + 8));

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    cs2_load_weights<BN>(d, sW, tid);
    if (PAIR) cs2_load_weights<BN>(d2, sW2, tid);
    const int tblocks = (d.TT + TB - 1) / TB;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * tblocks;       // JB == J
    int tmin0 = min(d.cv_toff[0][0], d.cv_toff[0][1]), tmin1 = min(d.cv_toff[1][0], d.cv_toff[1][1]);
    if (PAIR) {
        tmin0 = min(tmin0, min(d2.cv_toff[0][0], d2.cv_toff[0][1]));
        tmin1 = min(tmin1, min(d2.cv_toff[1][0], d2.cv_toff[1][1]));
    }
    const int cp8 = CT >> 3;
    const int NP = (TB + 1) * FR * cp8;

    // ---- staging slots (see sw_fetch_patch)
    const bf16_raw* p_ptr[NPC];
    int p_lds[NPC], p_pp[NPC];
    const int lds_dump = (TB + 1) * FR * PP;
#pragma unroll
    for (int u = 0; u < NPC; ++u) {
        const int idx = tid + 256 * u;
        p_pp[u] = -1; p_lds[u] = lds_dump; p_ptr[u] = nullptr;
        if (idx < NP) {
            const int pp = idx / (FR * cp8), rem = idx - pp * (FR * cp8);
            const int r = rem / cp8, c8 = rem - r * cp8;
            const bool second = c8 * 8 >= C0;
            const int sF = second ? d.src[1].F : d.src[0].F, sC = second ? C1 : C0;
            const int f = fadd_u + r;
            p_lds[u] = (pp * FR + r) * PP + c8 * 8;
            if (f >= 0 && f < sF) {
                p_pp[u] = pp | (second ? 0x10000 : 0);
                p_ptr[u] = reinterpret_cast<const bf16_raw*>(second ? d.src[1].ptr : d.src[0].ptr) +
                           ((long)((second ? tmin1 : tmin0) + pp) * sF + f) * sC + (c8 * 8 - (second ? C0 : 0));
            }
        }
    }
    const bf16_raw* zero_page = reinterpret_cast<const bf16_raw*>(&sehip_zero16);

    // ---- MFMA operand rows of this lane (wave w owns rows [16 MI w, 16 MI (w + 1)) of the 64 MI-row tile) and the sides
    int abase[MI], e_tl[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int r = 16 * MI * w + mi * 16 + (lane & 15);
        const int tl = r / JB, jl = r - tl * JB;
        abase[mi] = (tl * FR + jl * d.fmul) * PP;
        e_tl[mi] = tl;
    }
    const Cs2Side<TN, MI> sa = cs2_side_init<TN, MI>(d, sW, tmin0, tmin1, fadd_u, JB, w, lane);
    const Cs2Side<TN, MI> sb = cs2_side_init<TN, MI>(PAIR ? d2 : d, sW2, tmin0, tmin1, fadd_u, JB, w, lane);

    const int mt_begin = blockIdx.x * tiles_per_wg, mt_end = min(MT, mt_begin + tiles_per_wg);
    float st[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) st[i] = 0.f;
    const bool do_stats = TN <= 2 && d.stats != nullptr;
    RegTile<NPC> pr;
#define CS_FETCH(mt_)                                                                                             \
    {                                                                                                             \
        const int b_ = (mt_) / tblocks, t0_ = ((mt_) - b_ * tblocks) * TB;                                        \
        const long off0_ = ((long)b_ * d.src[0].T + t0_) * d.src[0].F * C0;                                       \
        const long off1_ = C1 ? ((long)b_ * d.src[1].T + t0_) * d.src[1].F * C1 : 0;                              \
        pr = sw_fetch_patch<NPC>(p_ptr, p_pp, off0_, off1_, d.src[0].tlo - t0_ - tmin0, d.src[0].thi - d.src[0].tlo, \
                                 d.src[1].tlo - t0_ - tmin1, d.src[1].thi - d.src[1].tlo, zero_page);             \
    }
    if (mt_begin < mt_end) CS_FETCH(mt_begin)
    for (int mt = mt_begin; mt < mt_end; ++mt) {
        const int b = mt / tblocks, t0 = (mt - b * tblocks) * TB;
#pragma unroll
        for (int u = 0; u < NPC; ++u) *reinterpret_cast<uint4*>(&patch[p_lds[u]]) = pr.v[u];
        __syncthreads();
        if (mt + 1 < mt_end) CS_FETCH(mt + 1)   // in flight while this tile is multiplied and stored
        cs2_multiply_store<TN, MI>(sa, d, patch, obuf, abase, e_tl, FR, PP, CT, C0, b, t0, lane, st, do_stats);
        if (PAIR) cs2_multiply_store<TN, MI>(sb, d2, patch, obuf, abase, e_tl, FR, PP, CT, C0, b, t0, lane, st, do_stats);
        __syncthreads();  // every read of this tile's patch is done before the next one is written
    }
#undef CS_FETCH
    if constexpr (TN <= 2) {
        // (the patch area is free: the tile loop ends with a barrier; the pair build has no LDS left for a static array)
        if (do_stats) small_stats_flush<TN>(st, reinterpret_cast<float*>(patch), d.stats, d.stats_cr, tid);
    }
}

// d2 == nullptr: one product.  d2 != nullptr: the pair (d, *d2) over one staged patch; returns 0 when the pair does not
// qualify (the caller then launches the two products separately).
// dry: only say whether the kernel would take the product(s)
static int try_conv_small(const sehip_gemm_desc& d, const sehip_gemm_desc* d2, hipStream_t st, bool dry = false) {
    static const bool disabled = getenv("SEHIP_NO_PATCH") != nullptr || getenv("SEHIP_NO_SMALL") != nullptr;
    if (disabled || d.cv_nf <= 0) return 0;
    // fused BatchNorm sums: the 32-output build only ([16 re | 16 im] in natural column order, one bf16 destination)
    // (and the 16-output build: [8 re | 8 im])
    if (d.stats && ((d.Npad != 32 && d.Npad != 16) || d.N != d.Npad || d.stats_cr * 2 != d.Npad || d.dst[1].ptr || d.dst[0].is_f32 ||
                    d.dst[0].C != d.Npad)) return 0;
    if (d2 && (d2->stats != d.stats || (d.stats && (d2->stats_cr != d.stats_cr || d2->dst[1].ptr || d2->dst[0].is_f32 || d2->dst[0].C != d.Npad))))
        return 0;
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int CT = C0 + C1;
    if ((C0 & 7) || (C1 & 7) || CT > 128 || (CT & (CT - 1)) || d.Npad > 128 || CT < 16) return 0;
    if (d.J > 128 || (128 % d.J)) return 0;
    const int KR = 2 * d.cv_nf * CT;
    if ((KR & 31) || KR > d.K) return 0;
    int fadd_u = d.cv_fadd, fend = d.cv_fadd + d.cv_nf;
    size_t wbytes = (size_t)d.Npad * (d.K + 8) * 2;
    if (d2) {
        static const bool nopair = getenv("SEHIP_NO_PAIR") != nullptr;
        const sehip_gemm_desc& e = *d2;
        if (nopair || e.cv_nf <= 0 || d.Npad > 32 || e.Npad != d.Npad || e.M != d.M || e.TT != d.TT || e.J != d.J || e.fmul != d.fmul) return 0;
        if (memcmp(e.src, d.src, sizeof(d.src)) != 0) return 0;  // same sources, same frame windows
        const int KR2 = 2 * e.cv_nf * CT;
        if ((KR2 & 31) || KR2 > e.K) return 0;
        fadd_u = min(fadd_u, e.cv_fadd);
        fend = max(fend, e.cv_fadd + e.cv_nf);
        wbytes += (size_t)e.Npad * (e.K + 8) * 2;
    }
    const int JB = d.J;
    const int FR = (JB - 1) * d.fmul + (fend - fadd_u);
    // rows per tile = 64 MI.  Larger tiles amortise the per-tile instruction stream and the weight-fragment reads (every wave
    // reads all of W per K step) but cost registers and LDS, i.e. resident workgroups; measured per layer (B=32, us, MI 2 / 4):
    // enc1.fwd 35/31, enc1.dg 27/27, enc2.fwd 56/48, enc2.dg0 29/33, dec4.fwd 42/37, dec4.dg 46/63, dec5.fwd 42/53.
    // Rule that reproduces the winners: 4 when the pieces-per-thread class (6 / 12) does not grow, for 64 outputs when K >= 256.
    static const int mi_force = getenv("SEHIP_SMALL_MI") ? atoi(getenv("SEHIP_SMALL_MI")) : 0;
    const size_t lds_cap = d2 ? 160 * 1024 : 120 * 1024;
    auto pieces = [&](int mi) { return (64 * mi / JB + 1) * FR * (CT >> 3); };
    auto lds_of = [&](int mi) {  // weights + patch + dump slot + the four waves' output staging images
        return wbytes + (size_t)(64 * mi / JB + 1) * FR * (CT + 8) * 2 + 16 + (d.Npad >= 32 && d.Npad <= 64 ? (size_t)4 * 16 * mi * (d.Npad + 8) * 2 : 0);
    };
    auto fits = [&](int mi) { return pieces(mi) <= 12 * 256 && lds_of(mi) <= lds_cap; };
    int MI = 2;
    if (d.Npad <= 32 && fits(4) && (pieces(2) <= 6 * 256) == (pieces(4) <= 6 * 256)) MI = 4;
    if (d.Npad == 64 && fits(4) && d.K >= 256) MI = 4;
    if (mi_force == 2 || (mi_force == 4 && fits(4) && d.Npad <= 64)) MI = mi_force;
    if (!fits(MI)) return 0;
    const int TB = 64 * MI / JB;
    const size_t lds = lds_of(MI);  // includes the 16-byte dump slot
    const int B = d.M / (d.TT * d.J);
    const int MT = B * ((d.TT + TB - 1) / TB);
    int wgs = lds > 64 * 1024 ? 256 : (lds > 40 * 1024 ? 512 : 1024);
    if (wgs > MT) wgs = MT;
    const int tiles_per_wg = (MT + wgs - 1) / wgs;
    const int grid = (MT + tiles_per_wg - 1) / tiles_per_wg;
    const bool few = pieces(MI) <= 6 * 256;
#define CS2_CASE(BN_, NPC_, MI_, PAIR_)                                                                             \
    if (d.Npad == BN_ && MI == MI_ && few == (NPC_ == 6) && (d2 != nullptr) == PAIR_) {                              \
        if (dry) return 1;                                                                                          \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small2_kernel<BN_, NPC_, MI_, PAIR_>),    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_ ? 160 * 1024 : 120 * 1024); \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        sehip_note_kernel(PAIR_ ? "conv_small2_kernel<%d, %d, %d, pair>" : "conv_small2_kernel<%d, %d, %d>", BN_, NPC_, MI_); \
        conv_small2_kernel<BN_, NPC_, MI_, PAIR_><<<grid, 256, lds, st>>>(d, d2 ? *d2 : d, TB, JB, FR, fadd_u, tiles_per_wg); \
        return 1;                                                                                                   \
    }
#define CS2_BOTH(BN_, MI_) CS2_CASE(BN_, 6, MI_, false) CS2_CASE(BN_, 12, MI_, false)
#define CS2_PAIR(BN_, MI_) CS2_CASE(BN_, 6, MI_, true) CS2_CASE(BN_, 12, MI_, true)
    CS2_BOTH(16, 4) CS2_BOTH(16, 2)
    CS2_BOTH(32, 4) CS2_BOTH(32, 2)
    CS2_BOTH(64, 4) CS2_BOTH(64, 2)
    CS2_BOTH(128, 2)
    CS2_PAIR(16, 4) CS2_PAIR(16, 2) CS2_PAIR(32, 4) CS2_PAIR(32, 2)
#undef CS2_PAIR
#undef CS2_BOTH
#undef CS2_CASE
    return 0;
}


// ------------------------------------------------------------------------------------------------
// wgrad: tile BNW (n) x 64*KQ (k), m consumed 64 rows per step.  KQ = 1: the LSTM / projection products (K <= 512).  KQ = 4:
// products with a long K (the DCUnet convolutions, K up to 4480 = 35 taps x 128 channels): every 64-row slab of dOut that a
// workgroup stages is multiplied with 256 k-columns instead of 64, i.e. dOut is re-read K/256 instead of K/64 times (it was
// 2/3 of that kernel's traffic).
// ------------------------------------------------------------------------------------------------
template <int BNW, int WNN, int WNK, int KQ>
__device__ __forceinline__ void wgrad_body(const sehip_gemm_desc& d, const int bx, const int by, const int bz, const int m_per_block) {
    constexpr int TN = BNW / WNN / 16, TK = 64 / WNK / 16;
    constexpr int PG = BNW + 8;  // pitch in bf16 elements (16 B pad)
    constexpr int PX = 64 * KQ + 8;
    constexpr int GCH = BNW / 8;         // 16-byte chunks per dOut row
    constexpr int GPT = (64 * GCH + 255) / 256;  // dOut chunks per thread
    constexpr int NCH = 8 * KQ;          // 16-byte chunks per A row of the tile
    constexpr int RPP = 256 / NCH;       // rows per staging pass
    constexpr int NXA = 64 / RPP;        // A chunks per thread
    __shared__ __attribute__((aligned(16))) bf16_raw sG[64 * PG];
    __shared__ __attribute__((aligned(16))) bf16_raw sX[64 * PX];
    __shared__ sehip_dst sdst[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WNK, wk = wave % WNK;
    const int n0 = bx * BNW, k0 = by * (64 * KQ);
    const int m_begin = bz * m_per_block;
    const int m_end = min(d.M, m_begin + m_per_block);
    if (tid < 2) sdst[tid] = d.dst[tid];
    __syncthreads();

    const int kc = tid % NCH, r0 = tid / NCH;
    sehip_kchunk e = d.ktab[min((k0 >> 3) + kc, (d.K >> 3) - 1)];
    if (k0 + 8 * kc >= d.K) e.src = -1;  // the last k tile of a K that is not a multiple of 64*KQ: zero chunks

    f32x4 acc[KQ][TN][TK];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TK; ++b) acc[kq][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of dOut: one extra MFMA against an all-ones operand in the waves that own k-subtile 0
    const bool do_bias = d.dbias != nullptr && by == 0 && wk == 0;
    f32x4 accb[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) accb[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging addresses.  Everything that does not depend on the slab is fixed per thread: its k chunk (source, tap
    // offsets, bounds) and, for dOut, its 8-column group (destination, column offset).  The rows it stages advance by 64 per
    // slab, so (utterance, frame, row) are stepped with carries instead of two integer divisions per row and slab -- the
    // first version spent ~630 vector instructions per slab on this against 16-24 MFMAs (the DCUnet weight gradients ran at
    // 95 TFLOP/s on address arithmetic).
    struct RowState { int b, t, j; };
    auto decompose = [&](int m) {
        RowState r;
        const int mm = m < d.M ? m : 0;
        const int bt = mm / d.J;
        r.j = mm - bt * d.J;
        r.b = bt / d.TT;
        r.t = bt - r.b * d.TT;
        return r;
    };
    const int adv_t = 64 / d.J, adv_j = 64 - adv_t * d.J;
    auto advance = [&](RowState& r) {
        r.j += adv_j; r.t += adv_t;
        if (r.j >= d.J) { r.j -= d.J; ++r.t; }
        for (; r.t >= d.TT; r.t -= d.TT) ++r.b;
    };
    // A operand: this thread's chunk of the rows r0 + RPP*i
    const bool a_second = e.src > 0;
    const sehip_src& AS = a_second ? d.src[1] : d.src[0];
    const bool a_narrow = e.src >= 0 && AS.C == 2;
    const int a_toff = e.toff >> 16, a_fadd = (int)(short)(e.toff & 0xffff);
    const int a_tlo = AS.tlo, a_thi = AS.thi, a_F = AS.F, a_C = AS.C, a_T = AS.T, a_tm = src_tmul(d);
    const bf16_raw* a_base = reinterpret_cast<const bf16_raw*>(AS.ptr) + e.fadd;      // e.fadd = element delta of the chunk
    RowState ra[NXA];
#pragma unroll
    for (int i = 0; i < NXA; ++i) ra[i] = decompose(m_begin + r0 + RPP * i);
    // dOut: piece id = tid + 256 i is row id / GCH, column group id % GCH; 256 is a multiple of GCH, so the column group (hence
    // destination, column offset, density) is the same for all of a thread's pieces and only the row differs
    static_assert(256 % GCH == 0, "column group of a thread must not depend on the piece");
    const int g_gc = tid % GCH, g_r0 = tid / GCH;
    const bool g_have = tid < 64 * GCH;
    const sehip_nchunk g_c0 = d.ntab[(n0 + g_gc * 8) >> 2], g_c1 = d.ntab[((n0 + g_gc * 8) >> 2) + 1];
    const bool g_dense = g_c0.nvalid == 4 && g_c1.nvalid == 4 && g_c1.dst == g_c0.dst && g_c1.coff == g_c0.coff + 4;
    const sehip_dst& GD = sdst[g_c0.dst > 0 ? 1 : 0];
    const bf16_raw* g_base = reinterpret_cast<const bf16_raw*>(GD.ptr) + g_c0.coff;
    const int g_T = GD.T, g_F = GD.F, g_C = GD.C, g_tm = GD.tmul > 1 ? GD.tmul : 1, g_toff = GD.toff, g_fmul = GD.fmul, g_fadd = GD.fadd;
    RowState rg[GPT];
#pragma unroll
    for (int i = 0; i < GPT; ++i) rg[i] = decompose(m_begin + g_r0 + (256 / GCH) * i);
    const bf16_raw* zero_page = reinterpret_cast<const bf16_raw*>(&sehip_zero16);

    uint4 xa[NXA], ga[GPT];
    auto fetch = [&](int mb) {
#pragma unroll
        for (int i = 0; i < NXA; ++i) {
            const RowState r = ra[i];
            const bool valid = mb + r0 + RPP * i < m_end;
            if (a_narrow) {    // 2-channel source: the generic gather (4 rows x (re, im) per chunk)
                RowPos rp;
                rp.valid = valid; rp.b = r.b; rp.t = r.t; rp.j = r.j; rp.jf = r.j * d.fmul; rp.ts = r.t * a_tm;
                xa[i] = gather_chunk(d.src[0], d.src[1], e, rp, row_base(d.src[0], rp), row_base(d.src[1], rp));
            } else {
                const int ts = r.t * a_tm + a_toff, f = r.j * d.fmul + a_fadd;
                const bool ok = valid && e.src >= 0 && ts >= a_tlo && ts < a_thi && (unsigned)f < (unsigned)a_F;
                const long off = ((long)(r.b * a_T + r.t * a_tm) * a_F + r.j * d.fmul) * a_C;
                const bf16_raw* q = ok ? a_base + off : zero_page;        // unconditional load (zeros for padding): no branch per piece
                xa[i] = *reinterpret_cast<const uint4*>(q);
            }
            advance(ra[i]);
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int row = g_r0 + (256 / GCH) * i;
            if (g_have && row < 64) {
                const RowState r = rg[i];
                const bool valid = mb + row < m_end;
                if (g_dense) {
                    const long off = ((long)(r.b * g_T + r.t * g_tm + g_toff) * g_F + r.j * g_fmul + g_fadd) * g_C;
                    const bf16_raw* q = valid ? g_base + off : zero_page;
                    v = *reinterpret_cast<const uint4*>(q);
                } else if (valid) {
                    bf16_raw tmp[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const sehip_nchunk c = h ? g_c1 : g_c0;
                        const sehip_dst& ds = sdst[c.dst > 0 ? 1 : 0];
                        const long o2 = ((long)(r.b * ds.T + r.t * (ds.tmul > 1 ? ds.tmul : 1) + ds.toff) * ds.F + r.j * ds.fmul + ds.fadd) * ds.C;
                        const bf16_raw* p = reinterpret_cast<const bf16_raw*>(ds.ptr) + o2 + c.coff;
#pragma unroll
                        for (int q = 0; q < 4; ++q) tmp[h * 4 + q] = (q < c.nvalid) ? p[q] : (bf16_raw)0;
                    }
                    v = make_uint4(tmp[0] | ((unsigned)tmp[1] << 16), tmp[2] | ((unsigned)tmp[3] << 16),
                                   tmp[4] | ((unsigned)tmp[5] << 16), tmp[6] | ((unsigned)tmp[7] << 16));
                }
                advance(rg[i]);
            }
            ga[i] = v;
        }
    };
    fetch(m_begin);
    for (int mb = m_begin; mb < m_end; mb += 64) {
#pragma unroll
        for (int i = 0; i < NXA; ++i) *reinterpret_cast<uint4*>(&sX[(r0 + RPP * i) * PX + kc * 8]) = xa[i];
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int id = tid + 256 * i;
            if (id < 64 * GCH) {
                const int r = id / GCH, gc = id - r * GCH;
                *reinterpret_cast<uint4*>(&sG[r * PG + gc * 8]) = ga[i];
            }
        }
        __syncthreads();
        if (mb + 64 < m_end) fetch(mb + 64);  // next slab in flight while this one is multiplied
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int g = lane >> 4, i16 = lane & 15;
            const int mrow = sub * 32 + 8 * g + (i16 >> 2);
            bf16x8 gf[TN];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = wn * (BNW / WNN) + ni * 16 + 4 * (i16 & 3);
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[mrow * PG + col]);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[(mrow + 4) * PG + col]);
                gf[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                bf16x8 xf[TK];
#pragma unroll
                for (int ki = 0; ki < TK; ++ki) {
                    const int col = kq * 64 + wk * (64 / WNK) + ki * 16 + 4 * (i16 & 3);
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sX[mrow * PX + col]);
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sX[(mrow + 4) * PX + col]);
                    xf[ki] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int ki = 0; ki < TK; ++ki)
                        acc[kq][ni][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], xf[ki], acc[kq][ni][ki], 0, 0, 0);
            }
            if (do_bias) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) accb[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], BF16_ONES, accb[ni], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // D rows = n (4*(lane>>4)+q), cols = k (lane&15).  Deterministic schedule: m-split bz adds to its own zeroed array (every entry
    // then has ONE contributor: the waves of a workgroup own disjoint (n, k) ranges), det_finish adds the arrays in split order
    float* dWs = d.dW + (size_t)bz * d.dw_split_stride;
    float* dbs = d.dbias ? d.dbias + (size_t)bz * d.dw_split_stride : nullptr;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int ki = 0; ki < TK; ++ki) {
                const int n = n0 + wn * (BNW / WNN) + ni * 16 + 4 * (lane >> 4);
                const int k = k0 + kq * 64 + wk * (64 / WNK) + ki * 16 + (lane & 15);
                if (k < d.K) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) atomicAdd(&dWs[(size_t)(n + q) * d.K + k], acc[kq][ni][ki][q]);
                }
            }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                atomicAdd(&dbs[n0 + wn * (BNW / WNN) + ni * 16 + 4 * (lane >> 4) + q], accb[ni][q]);
    }
}

template <int BNW, int WNN, int WNK, int KQ>
__global__ __launch_bounds__(256) void wgrad_kernel(const sehip_gemm_desc d, int m_per_block) {
    wgrad_body<BNW, WNN, WNK, KQ>(d, blockIdx.x, blockIdx.y, blockIdx.z, m_per_block);
}

// Several weight-gradient products in ONE launch (sehip_wgrad_group): the LSTM input / recurrent products are 82-1300
// workgroups of 256 rows each and take ~30 us apiece back to back on the side stream, almost all of it launch ramp and
// the dependent slab chain; side by side they take about as long as the largest one.  The descriptors live in device
// memory (copied once, sehip_wgrad_group_prepare); gtab[g] = {first block, n tiles, k tiles, rows per block}.
struct WgradGroupEntry { int first, ntiles, ktiles, mpb; };
#define SEHIP_WGRAD_GROUP_MAX 16
template <int BNW, int WNN, int WNK>
__global__ __launch_bounds__(256) void wgrad_group_kernel(const sehip_gemm_desc* __restrict__ descs,
                                                          const WgradGroupEntry* __restrict__ gtab, int ngroups) {
    int g = 0;
    for (int i = 1; i < ngroups; ++i)
        if ((int)blockIdx.x >= gtab[i].first) g = i;
    const WgradGroupEntry e = gtab[g];
    const int local = blockIdx.x - e.first;
    const int bx = local % e.ntiles, rest = local / e.ntiles;
    const int by = rest % e.ktiles, bz = rest / e.ktiles;
    wgrad_body<BNW, WNN, WNK, 1>(descs[g], bx, by, bz, e.mpb);   // g is workgroup-uniform: the fields arrive by scalar loads
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

int sehip_convs_bnr_rows(const sehip_gemm_desc& a);   // convt.hip
extern "C" int sehip_bnr_rows(const sehip_gemm_desc* a, const sehip_gemm_desc* b) {
    if (!a || b) return 0;                            // (pairs: not built)
    return sehip_convs_bnr_rows(*a);
}

// ------------------------------------------------------------------------------------------------
// conv_narrow_kernel: forward / dgrad product whose SOURCE has 2 channels (the first encoder layer reads the
// spectrogram, the last decoder layer's input gradient reads d(mask)): K = 2 frames x 5 taps x 2 channels in the
// layout k = kt*16 + tap*2 + c (K = 32 with the padding), so the whole reduction is ONE MFMA step and the layer is a
// pure stream: 5 MB in, 42-85 MB out.  The table-gathered generic kernel spent its time on 4-byte gathers (60-71 us).
// Here the weights live in registers, the input frames of a tile are staged in LDS as they lie in memory (4 bytes per
// row), and a lane's 8 consecutive k are 16 contiguous bytes of that image (4 taps x 2 channels; the taps beyond the
// fifth meet zero weights).  Tile loop and store addressing as in conv_small2_kernel.
// ------------------------------------------------------------------------------------------------
template <int BN, int MI>
__global__ __launch_bounds__(256) void conv_narrow_kernel(const sehip_gemm_desc d, int TB, int FRA, int fa, int tiles_per_wg) {
    constexpr int TN = BN / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* sX = reinterpret_cast<unsigned*>(smem);  // [(TB + 1) frames][FRA rows] dwords (re | im)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4;
    const int JB = d.J;
    const int tblocks = (d.TT + TB - 1) / TB;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * tblocks;
    const int tmin = min(d.cv_toff[0][0], d.cv_toff[0][1]);
    const int sT = d.src[0].T, sF = d.src[0].F;
    const unsigned* xsrc = reinterpret_cast<const unsigned*>(d.src[0].ptr);

    // weights: row (lane & 15) of each 16-row tile, k chunk g
    bf16x8 wf[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
        wf[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_raw*>(d.W) +
                                                                              (size_t)(ni * 16 + (lane & 15)) * d.K + 8 * g));
    // staging: piece = 4 rows of one frame; (TB + 1) * FRA / 4 pieces, NPL per thread
    constexpr int NPL = 4;
    const int ppf = FRA >> 2;                 // pieces per frame
    const int NP = (TB + 1) * ppf;
    int p_fr[NPL], p_row[NPL];
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
        const int idx = tid + 256 * u;
        p_fr[u] = -1; p_row[u] = 0;
        if (idx < NP) { p_fr[u] = idx / ppf; p_row[u] = (idx - p_fr[u] * ppf) * 4; }
    }
    // operand rows / store addressing of this lane
    const int kt = g >> 1, hf = g & 1;
    int abase[MI], e_tl[MI], e_off0[MI], e_off1[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int r = 16 * MI * w + mi * 16 + (lane & 15);
        const int tl = r / JB, jl = r - tl * JB;
        abase[mi] = (tl + d.cv_toff[0][kt] - tmin) * FRA + jl * d.fmul + 4 * hf + (d.cv_fadd - fa);
        e_tl[mi] = tl;
        RowPos rp;
        rp.b = 0; rp.t = tl; rp.j = jl; rp.jf = jl * d.fmul; rp.valid = true;
        e_off0[mi] = (int)dst_row_offset(d.dst[0], rp, d.fmul);
        e_off1[mi] = d.dst[1].ptr ? (int)dst_row_offset(d.dst[1], rp, d.fmul) : 0;
    }
    const long bs0 = (long)d.dst[0].T * d.dst[0].F * d.dst[0].C, bs1 = (long)d.dst[1].T * d.dst[1].F * d.dst[1].C;
    const int ts0 = d.dst[0].F * d.dst[0].C, ts1 = d.dst[1].F * d.dst[1].C;
    sehip_nchunk nck[TN];
    float4 bias4[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = ni * 16 + 4 * g;
        nck[ni] = d.ntab[n >> 2];
        bias4[ni] = d.bias ? *reinterpret_cast<const float4*>(d.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const int mt_begin = blockIdx.x * tiles_per_wg, mt_end = min(MT, mt_begin + tiles_per_wg);
    float st[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) st[i] = 0.f;
    const bool do_stats = d.stats != nullptr;          // (fused BatchNorm sums: small_stats_add)
    uint4 pr[NPL];
#define CN_FETCH(mt_)                                                                                              \
    {                                                                                                              \
        const int b_ = (mt_) / tblocks, t0_ = ((mt_) - b_ * tblocks) * TB;                                         \
        _Pragma("unroll") for (int u = 0; u < NPL; ++u) {                                                          \
            pr[u] = make_uint4(0u, 0u, 0u, 0u);                                                                    \
            const int ts = t0_ + tmin + p_fr[u], f = fa + p_row[u];                                                \
            if (p_fr[u] >= 0 && ts >= d.src[0].tlo && ts < d.src[0].thi) {                                         \
                const unsigned* q = xsrc + ((long)b_ * sT + ts) * sF + f;                                          \
                if (f >= 0 && f + 3 < sF) pr[u] = *reinterpret_cast<const uint4*>(q);                              \
                else {                                                                                             \
                    unsigned v[4];                                                                                 \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = (f + r >= 0 && f + r < sF) ? q[r] : 0u;   \
                    pr[u] = make_uint4(v[0], v[1], v[2], v[3]);                                                    \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }
    if (mt_begin < mt_end) CN_FETCH(mt_begin)
    for (int mt = mt_begin; mt < mt_end; ++mt) {
        const int b = mt / tblocks, t0 = (mt - b * tblocks) * TB;
#pragma unroll
        for (int u = 0; u < NPL; ++u)
            if (p_fr[u] >= 0) *reinterpret_cast<uint4*>(&sX[p_fr[u] * FRA + p_row[u]]) = pr[u];
        __syncthreads();
        if (mt + 1 < mt_end) CN_FETCH(mt + 1)

        f32x4 acc[TN][MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            // 16 contiguous bytes at an 8-byte aligned address: two ds_read_b64
            const uint2 lo = *reinterpret_cast<const uint2*>(&sX[abase[mi]]), hi = *reinterpret_cast<const uint2*>(&sX[abase[mi] + 2]);
            const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        if (do_stats) small_stats_add<TN, MI>(acc, bias4, e_tl, t0, d.TT, st);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (t0 + e_tl[mi] >= d.TT) continue;
            const long ro0 = e_off0[mi] + b * bs0 + (long)t0 * ts0;
            const long ro1 = e_off1[mi] + b * bs1 + (long)t0 * ts1;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const sehip_nchunk nc = nck[ni];
                if (nc.nvalid <= 0) continue;
                f32x4 v = acc[ni][mi];
                v[0] += bias4[ni].x; v[1] += bias4[ni].y; v[2] += bias4[ni].z; v[3] += bias4[ni].w;
                const long off = (nc.dst ? ro1 : ro0) + nc.coff;
                void* dptr = nc.dst ? d.dst[1].ptr : d.dst[0].ptr;
                const int is_f32 = nc.dst ? d.dst[1].is_f32 : d.dst[0].is_f32;
                if (is_f32) {
                    float* q = reinterpret_cast<float*>(dptr) + off;
                    if (nc.nvalid == 4) *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
                    else
                        for (int e = 0; e < nc.nvalid; ++e) q[e] = v[e];
                } else {
                    bf16_raw* q = reinterpret_cast<bf16_raw*>(dptr) + off;
                    if (nc.nvalid == 4) *reinterpret_cast<uint2*>(q) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    else
                        for (int e = 0; e < nc.nvalid; ++e) q[e] = f2bf(v[e]);
                }
            }
        }
        __syncthreads();
    }
#undef CN_FETCH
    if (do_stats) small_stats_flush<TN>(st, reinterpret_cast<float*>(smem), d.stats, d.stats_cr, tid);
}

static int try_conv_narrow(const sehip_gemm_desc& d, hipStream_t st, bool dry = false) {
    static const bool disabled = getenv("SEHIP_NO_NARROW") != nullptr;
    if (disabled || d.cv_nf <= 0 || d.cv_nf > 5 || d.src[0].C != 2 || d.src[1].ptr || d.res) return 0;
    if (d.stats && (d.N != d.Npad || d.stats_cr * 2 != d.Npad || d.dst[1].ptr || d.dst[0].is_f32 || d.dst[0].C != d.Npad)) return 0;
    if ((d.Npad != 16 && d.Npad != 32) || d.K < 32 || d.J > 128 || (128 % d.J)) return 0;
    const int fa = (d.cv_fadd >= 0 ? d.cv_fadd / 4 : -((-d.cv_fadd + 3) / 4)) * 4;
    if ((d.cv_fadd - fa) & 1) return 0;  // 8-byte aligned operand reads
    constexpr int MI = 4;
    const int TB = 64 * MI / d.J;
    if (TB < 1) return 0;
    // staged rows per frame: the lane of the last row reads taps 4..7 as well (zero weights): (J-1)*fmul + 8 rows
    const int FRA = ((d.cv_fadd - fa) + (d.J - 1) * d.fmul + 8 + 3) / 4 * 4;
    if ((TB + 1) * (FRA >> 2) > 4 * 256 || (d.src[0].F & 3)) return 0;
    size_t lds = (size_t)(TB + 1) * FRA * 4;
    if (lds < 320 * sizeof(float)) lds = 320 * sizeof(float);      // small_stats_flush
    if (dry) return 1;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * ((d.TT + TB - 1) / TB);
    int wgs = 1024;
    if (wgs > MT) wgs = MT;
    const int tiles_per_wg = (MT + wgs - 1) / wgs;
    const int grid = (MT + tiles_per_wg - 1) / tiles_per_wg;
    sehip_note_kernel("conv_narrow_kernel<%d, %d>", d.Npad, MI);
    if (d.Npad == 16) conv_narrow_kernel<16, MI><<<grid, 256, lds, st>>>(d, TB, FRA, fa, tiles_per_wg);
    else conv_narrow_kernel<32, MI><<<grid, 256, lds, st>>>(d, TB, FRA, fa, tiles_per_wg);
    return 1;
}

float* sehip_wgrad_scratch(hipStream_t st, size_t bytes);   // csrc/wgrad3.hip: per-stream pool of partial arrays

extern "C" int sehip_gemm(const sehip_gemm_desc* d, void* stream) {
    if (int e = check_desc("gemm", d)) return e;
    SEHIP_REQUIRE(d->W != nullptr, "gemm: missing weights");
    hipStream_t st = (hipStream_t)stream;
    if (sehip_try_convs_stream(*d, st, false)) {
        SEHIP_CHECK_LAUNCH("gemm(convs-stream)");
        return 0;
    }
    if (d->stats) {   // only the LDS-DMA convolution kernel accumulates the BatchNorm statistics (sehip.h): no silent omission
        if (sehip_try_conv_gemm_v3(*d, st) || (!d->w_tiled && sehip_try_conv_gemm_v2(*d, st)) || (!d->w_tiled && (try_conv_small(*d, nullptr, st) || try_conv_narrow(*d, st)))) {
            SEHIP_CHECK_LAUNCH("gemm(conv+stats)");
            return 0;
        }
        return sehip_set_error(-1, "gemm: this product cannot accumulate BatchNorm statistics (field stats): it does not qualify "
                                   "for conv_gemm_v2 (Npad=%d, stats_cr=%d, cv_nf=%d)", d->Npad, d->stats_cr, d->cv_nf);
    }
    if (try_conv_gemm(*d, st)) {
        SEHIP_CHECK_LAUNCH("gemm(conv)");
        return 0;
    }
    SEHIP_REQUIRE(!d->w_tiled, "gemm: W is in conv_gemm_v3's tile order (w_tiled) but that kernel did not take the product: it does not "
                               "qualify (Npad=%d, J=%d, cv_nf=%d, fmul=%d), or the kernel's dynamic-LDS limit could not be raised on "
                               "the current device", d->Npad, d->J, d->cv_nf, d->fmul);
    if (try_conv_small(*d, nullptr, st)) {
        SEHIP_CHECK_LAUNCH("gemm(conv-small)");
        return 0;
    }
    if (try_conv_narrow(*d, st)) {
        SEHIP_CHECK_LAUNCH("gemm(conv-narrow)");
        return 0;
    }
    if (sehip_try_dense_rows_gemm(*d, st)) {      // plain dense rows, the whole reduction in one pass (ConvTasNet's 1x1 convolutions)
        SEHIP_CHECK_LAUNCH("gemm(dense-rows)");
        return 0;
    }
    sehip_note_kernel("gemm_kernel<%d, %d, %d, %d>", d->Npad <= 64 ? d->Npad : 128, d->Npad <= 64 ? 256 : 128,
                      d->Npad <= 64 ? 1 : 2, d->Npad <= 64 ? 4 : 2);
    if (d->Npad == 16) {
        gemm_kernel<16, 256, 1, 4><<<cdiv(d->M, 256), 256, 0, st>>>(*d);
    } else if (d->Npad == 32) {
        gemm_kernel<32, 256, 1, 4><<<cdiv(d->M, 256), 256, 0, st>>>(*d);
    } else if (d->Npad == 64 && d->M < 256 * 192) {  // too few 256-row tiles to fill the GPU (LSTM input gradients)
        sehip_note_kernel("gemm_kernel<64, 64, 2, 2>");
        gemm_kernel<64, 64, 2, 2><<<cdiv(d->M, 64), 256, 0, st>>>(*d);
    } else if (d->Npad == 64) {
        gemm_kernel<64, 256, 1, 4><<<cdiv(d->M, 256), 256, 0, st>>>(*d);
    } else if (d->Npad == 192) {   // 128 + 64 output columns (DCUnet's last decoder input gradient: decoder input | skip): one 192-wide tile
        sehip_note_kernel("gemm_kernel<192, 128, 2, 2>");
        gemm_kernel<192, 128, 2, 2><<<cdiv(d->M, 128), 256, 0, st>>>(*d);
    } else {
        SEHIP_REQUIRE(d->Npad % 128 == 0, "gemm: Npad=%d must be 16, 32, 64, 192 or a multiple of 128", d->Npad);
        static const bool no64 = getenv("SEHIP_NO_BM64") != nullptr;
        // very short row spaces (Demucs' deepest levels: 736 rows x 2048-4096 columns x 6144-12288 k): even 64-row tiles leave
        // under one workgroup per CU, each a chain of dependent K steps: 64 x 64 tiles double the workgroups in flight
        // ... and long K: split-K (4 splits of >= 8 steps; the partial images are S x M x Npad floats from the per-stream pool)
        static const int splitk = getenv("SEHIP_GEMM_SPLITK") ? atoi(getenv("SEHIP_GEMM_SPLITK")) : 4;
        if (splitk > 1 && !no64 && (long)cdiv(d->M, 64) * (d->Npad / 128) < 512 && (d->K >> 6) >= 8 * splitk) {
            const int nk = d->K >> 6, per = cdiv(nk, splitk), S = cdiv(nk, per);
            float* part = sehip_wgrad_scratch(st, (size_t)S * d->M * d->Npad * sizeof(float));
            if (part) {
                sehip_note_kernel("gemm_splitk_kernel<128, 64, 2, 2>");
                gemm_splitk_kernel<128, 64, 2, 2><<<dim3(cdiv(d->M, 64) * (d->Npad / 128), S), 256, 0, st>>>(*d, per, part);
                gemm_splitk_finish_kernel<<<cdiv((long)d->M * (d->Npad >> 2), 256), 256, 0, st>>>(*d, part, S);
                SEHIP_CHECK_LAUNCH("gemm(split-K)");
                return 0;
            }
        }
        static const int small64 = getenv("SEHIP_GEMM_SMALL64") ? atoi(getenv("SEHIP_GEMM_SMALL64")) : 384;
        if (!no64 && (long)cdiv(d->M, 64) * (d->Npad / 128) < small64) {
            sehip_note_kernel("gemm_kernel<64, 64, 2, 2>");
            gemm_kernel<64, 64, 2, 2><<<cdiv(d->M, 64) * (d->Npad / 64), 256, 0, st>>>(*d);
            SEHIP_CHECK_LAUNCH("gemm");
            return 0;
        }
        if (!no64 && (long)cdiv(d->M, 128) * (d->Npad / 128) < 512) {  // under one round of 2 workgroups per CU: halve the tile
            sehip_note_kernel("gemm_kernel<128, 64, 2, 2>");
            gemm_kernel<128, 64, 2, 2><<<cdiv(d->M, 64) * (d->Npad / 128), 256, 0, st>>>(*d);
        } else
            gemm_kernel<128, 128, 2, 2><<<cdiv(d->M, 128) * (d->Npad / 128), 256, 0, st>>>(*d);
    }
    SEHIP_CHECK_LAUNCH("gemm");
    return 0;
}

// Two products over the SAME sources (the two output-row parities of a transposed convolution, src/model/dccrn.py:387-450):
// one launch that stages the input once when the small-channel kernel takes the pair, otherwise the two launches.
// 1 when conv_small2_kernel takes the product (b == NULL) or the pair as described, fused BatchNorm sums (field stats) included:
// the plan asks before it relies on them (workspace-dependent: the kernel must fit its patch into LDS)
int sehip_try_convt_stream(const sehip_gemm_desc& a, const sehip_gemm_desc& b, hipStream_t st, bool dry);   // convt.hip

extern "C" int sehip_conv_small_takes(const sehip_gemm_desc* a, const sehip_gemm_desc* b) {
    if (!a) return 0;
    if (b && sehip_try_convt_stream(*a, *b, nullptr, true)) return 1;
    if (!b && sehip_try_convs_stream(*a, nullptr, true)) return 1;
    if (try_conv_small(*a, b, nullptr, true)) return 1;
    return b ? 0 : try_conv_narrow(*a, nullptr, true);          // (2-channel input: conv_narrow_kernel)
}

extern "C" int sehip_gemm_pair(const sehip_gemm_desc* a, const sehip_gemm_desc* b, void* stream) {
    if (int e = check_desc("gemm_pair", a)) return e;
    if (int e = check_desc("gemm_pair", b)) return e;
    SEHIP_REQUIRE(a->W != nullptr && b->W != nullptr, "gemm_pair: missing weights");
    if (sehip_try_convt_stream(*a, *b, (hipStream_t)stream, false)) {
        SEHIP_CHECK_LAUNCH("gemm_pair(convt-stream)");
        return 0;
    }
    if (try_conv_small(*a, b, (hipStream_t)stream)) {
        SEHIP_CHECK_LAUNCH("gemm_pair(conv-small)");
        return 0;
    }
    if (sehip_try_conv_gemm_v3_pair(*a, *b, (hipStream_t)stream)) {       // both parities' tiles in one conv_gemm_v3 launch
        SEHIP_CHECK_LAUNCH("gemm_pair(conv-v3-pair)");
        return 0;
    }
    // same-shape dense products that the generic 64-row kernels take: one launch
    static const bool nopair = getenv("SEHIP_NO_PAIR") != nullptr;
    if (!nopair && a->cv_nf <= 0 && b->cv_nf <= 0 && a->M == b->M && a->Npad == b->Npad && a->K == b->K && a->J == b->J &&
        a->TT == b->TT && a->fmul == b->fmul) {
        hipStream_t st = (hipStream_t)stream;
        if (a->Npad == 64 && a->M < 256 * 192) {
            sehip_note_kernel("gemm_pair_kernel<64, 64, 2, 2>");
            GemmPair pr; pr.d[0] = *a; pr.d[1] = *b;
            gemm_pair_kernel<64, 64, 2, 2><<<dim3(cdiv(a->M, 64), 2), 256, 0, st>>>(pr);
            SEHIP_CHECK_LAUNCH("gemm_pair");
            return 0;
        }
        if (a->Npad % 128 == 0 && (long)cdiv(a->M, 128) * (a->Npad / 128) < 512) {
            sehip_note_kernel("gemm_pair_kernel<128, 64, 2, 2>");
            GemmPair pr; pr.d[0] = *a; pr.d[1] = *b;
            gemm_pair_kernel<128, 64, 2, 2><<<dim3(cdiv(a->M, 64) * (a->Npad / 128), 2), 256, 0, st>>>(pr);
            SEHIP_CHECK_LAUNCH("gemm_pair");
            return 0;
        }
    }
    if (int e = sehip_gemm(a, stream)) return e;
    return sehip_gemm(b, stream);
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad_kernel: weight gradient for descriptors with the regular-convolution description.
// A workgroup owns 64 output channels (n) x ONE 64-channel input chunk x ALL 2*NF taps and walks over 128-row
// m-tiles: per tile it stages the input patch and the dOut tile [128 m][64 n] in LDS once and every tap reuses them
// (the generic wgrad_kernel re-reads dOut once per 64 k-columns and the inputs once per tap).  Wave w accumulates
// dW[64 n][taps][16 channels (16w..)] in registers: 4 x 2NF MFMA tiles; both operands come from transposed LDS reads.
// ------------------------------------------------------------------------------------------------
#ifndef CW_MINW
#define CW_MINW 2      // minimum waves per SIMD the register allocation must allow (4: two workgroups, or one + a chain workgroup, per CU)
#endif
template <int NF>
__global__ __launch_bounds__(512, CW_MINW) void conv_wgrad_kernel(const sehip_gemm_desc d, int TB, int JB, int FR, int tiles_per_wg, int nsplit) {
    constexpr int NIT = 2 * NF;
    constexpr int GP = 80;  // pitch of the dOut tile: 160 B, so 8 consecutive rows sit on 8 disjoint 32-byte bank slots
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_raw* sG = reinterpret_cast<bf16_raw*>(smem);  // [128][GP]
    bf16_raw* patch = sG + 128 * GP;
    // patch pitch: consecutive m are consecutive patch rows (fmul 1) or every other row (fmul 2); 160 B / 144 B make
    // the 8 rows a 32-lane half reads in one transposed read conflict-free in both cases
    const int PPW = d.fmul == 2 ? 72 : 80;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w = wv & 3, nh = wv >> 2;  // wave -> 16-channel subtile w, 32-wide half nh of the 64 output channels
    // All (n-tile, channel-chunk) workgroups of one m-split read the same dOut tiles and patches at the same time:
    // put them on ONE XCD (blocks are dealt round-robin over the 8 XCDs) so that those re-reads hit its L2.
    const int ntn = d.Npad >> 6;
    const int gx = ntn * ((d.src[0].C + (d.src[1].ptr ? d.src[1].C : 0)) >> 6);
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    int xi, split;
    if (nsplit >= 8) { xi = rr % gx; split = (rr / gx) * 8 + xcd; }
    else { split = xcd % nsplit; xi = (xcd / nsplit) * (gx * nsplit >> 3) + rr; }  // 8/nsplit XCDs share one split
    const int nt = xi % ntn, cc = xi / ntn;
    const int n0 = nt * 64;
    const int tblocks = (d.TT + TB - 1) / TB;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * tblocks;
    const int mt_begin = split * tiles_per_wg, mt_end = min(MT, mt_begin + tiles_per_wg);

    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int Ctot = C0 + C1;
    const bool second = cc * 64 >= C0;
    const int sT = second ? d.src[1].T : d.src[0].T, sF = second ? d.src[1].F : d.src[0].F, sC = second ? C1 : C0;
    const int tlo = second ? d.src[1].tlo : d.src[0].tlo, thi = second ? d.src[1].thi : d.src[0].thi;
    const bf16_raw* sbase = reinterpret_cast<const bf16_raw*>(second ? d.src[1].ptr : d.src[0].ptr) + (cc * 64 - (second ? C0 : 0));
    const int tmin = second ? min(d.cv_toff[1][0], d.cv_toff[1][1]) : min(d.cv_toff[0][0], d.cv_toff[0][1]);
    const int dt0 = (second ? d.cv_toff[1][0] : d.cv_toff[0][0]) - tmin, dt1 = (second ? d.cv_toff[1][1] : d.cv_toff[0][1]) - tmin;
    const int NP = (TB + 1) * FR * 8;
    const int f0 = d.cv_fadd;  // JB == J: the tile starts at row 0

    // transposed-read addresses: lane supplies row (8g + q [+4]) of each 32-row k-step, columns 4p..4p+3
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = 4 * (i16 & 3);
    int pbase[4][2], gbase[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // MFMA k index j of lane group g  <->  tile row m = 32 ks + 16 (j>>2) + 4 g + (j&3): the two groups of a
            // 32-lane half then read 8 CONSECUTIVE rows (the reduction order is free as long as both operands agree)
            const int m = ks * 32 + 16 * h + 4 * g + q;
            const int tl = m / JB, jl = m - tl * JB;
            pbase[ks][h] = (tl * FR + jl * d.fmul) * PPW + 16 * w + p4;
            gbase[ks][h] = m * GP + p4 + 32 * nh;
        }

    f32x4 acc[2][NIT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NIT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = d.dbias != nullptr && cc == 0 && w == 0;  // column sums of dOut by MFMA against ones
    f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

    // Everything about a thread's staging slots that does not depend on the tile is computed once: the tile loop
    // then only adds the tile base and checks the frame range (no integer divisions inside the loop).
    constexpr int NPS = CV_MAXP / 2;
    int p_goff[NPS], p_lds[NPS];  // element offset from the tile's first patch frame (or INT_MIN), LDS slot | frame << 16 (or -1)
#pragma unroll
    for (int i = 0; i < NPS; ++i) {
        const int idx = tid + 512 * i;
        p_goff[i] = INT_MIN; p_lds[i] = -1;
        if (idx < NP) {
            const int pp = idx / (FR * 8), rem = idx - pp * (FR * 8);
            const int f = f0 + (rem >> 3);
            p_lds[i] = ((pp * FR + (rem >> 3)) * PPW + (rem & 7) * 8) | (pp << 16);
            if (f >= 0 && f < sF) p_goff[i] = (pp * sF + f) * sC + (rem & 7) * 8;
        }
    }
    const bf16_raw* g_ptr[2];
    int g_tl[2];
    long g_bstride[2];
    int g_tstride[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + 512 * u;
        const int r = idx >> 3, gc = idx & 7;
        const int tl = r / JB, jl = r - tl * JB;
        g_tl[u] = tl; g_ptr[u] = nullptr; g_bstride[u] = 0; g_tstride[u] = 0;
        const int n = n0 + gc * 8;
        const sehip_nchunk c0 = d.ntab[n >> 2], c1 = d.ntab[(n >> 2) + 1];
        if (c0.nvalid == 4 && c1.nvalid == 4 && c1.dst == c0.dst && c1.coff == c0.coff + 4) {
            const sehip_dst& dd = c0.dst ? d.dst[1] : d.dst[0];
            RowPos rp;
            rp.b = 0; rp.t = tl; rp.j = jl; rp.jf = jl * d.fmul; rp.valid = true;
            g_ptr[u] = reinterpret_cast<const bf16_raw*>(dd.ptr) + dst_row_offset(dd, rp, d.fmul) + c0.coff;
            g_tstride[u] = dd.F * dd.C;
            g_bstride[u] = (long)dd.T * dd.F * dd.C;
        }
    }

    uint4 pr[NPS], gr[2];
#define CW_FETCH(mt_)                                                                                              \
    {                                                                                                              \
        const int b_ = (mt_) / tblocks, t0_ = ((mt_) - b_ * tblocks) * TB;                                         \
        const bf16_raw* tb_ = sbase + ((long)b_ * sT + t0_ + tmin) * sF * sC;                                      \
        _Pragma("unroll") for (int i = 0; i < NPS; ++i) {                                                          \
            pr[i] = make_uint4(0u, 0u, 0u, 0u);                                                                    \
            const int ts_ = t0_ + tmin + (p_lds[i] >> 16);                                                         \
            if (p_goff[i] != INT_MIN && ts_ >= tlo && ts_ < thi) pr[i] = *reinterpret_cast<const uint4*>(tb_ + p_goff[i]); \
        }                                                                                                          \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                            \
            gr[u] = make_uint4(0u, 0u, 0u, 0u);                                                                    \
            if (g_ptr[u] && t0_ + g_tl[u] < d.TT)                                                                  \
                gr[u] = *reinterpret_cast<const uint4*>(g_ptr[u] + b_ * g_bstride[u] + (long)t0_ * g_tstride[u]);  \
        }                                                                                                          \
    }

    if (mt_begin < mt_end) CW_FETCH(mt_begin)
    for (int mt = mt_begin; mt < mt_end; ++mt) {
#pragma unroll
        for (int i = 0; i < NPS; ++i)
            if (p_lds[i] >= 0) *reinterpret_cast<uint4*>(&patch[p_lds[i] & 0xffff]) = pr[i];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + 512 * u;
            *reinterpret_cast<uint4*>(&sG[(idx >> 3) * GP + (idx & 7) * 8]) = gr[u];
        }
        __syncthreads();
        if (mt + 1 < mt_end) CW_FETCH(mt + 1)   // in flight while this tile is consumed
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 gf[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[gbase[ks][0] + ni * 16]);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[gbase[ks][1] + ni * 16]);
                gf[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            if (do_bias) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) accb[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], BF16_ONES, accb[ni], 0, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int toff_e = ((it < NF ? dt0 : dt1) * FR + (it < NF ? it : it - NF)) * PPW;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&patch[pbase[ks][0] + toff_e]);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&patch[pbase[ks][1] + toff_e]);
                const bf16x8 xf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[ni][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], xf, acc[ni][it], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    float* dWs = d.dW + (size_t)split * d.dw_split_stride;        // (deterministic schedule: one array per m-split, see wgrad_body)
    float* dbs = d.dbias ? d.dbias + (size_t)split * d.dw_split_stride : nullptr;
    // D rows = n (4*(lane>>4)+u), cols = channel (lane&15)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = n0 + 32 * nh + ni * 16 + 4 * (lane >> 4);
            const int k = it * Ctot + cc * 64 + 16 * w + (lane & 15);
#pragma unroll
            for (int u = 0; u < 4; ++u) atomicAdd(&dWs[(size_t)(n + u) * d.K + k], acc[ni][it][u]);
        }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int u = 0; u < 4; ++u) atomicAdd(&dbs[n0 + 32 * nh + ni * 16 + 4 * (lane >> 4) + u], accb[ni][u]);
    }
#undef CW_FETCH
}

// ---- deterministic schedule of the atomic-flush weight-gradient kernels (sehip_set_deterministic) -------------------------------
// det_begin: a private zeroed [Npad K + Npad] array per m-split (the kernel adds to d.dW + split * dw_split_stride); det_finish: the
// arrays are added into the caller's dW / dbias in split order by ONE thread per entry.  Returns false (error set) when no scratch
// can be had (inside a stream capture).
float* sehip_wgrad_scratch(hipStream_t st, size_t bytes);   // csrc/wgrad3.hip
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ parts, int nparts, size_t stride, size_t nw, int nb,
                                                         float* __restrict__ dW, float* __restrict__ dbias) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nw + (size_t)nb) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += parts[(size_t)p * stride + i];
    if (i < nw) dW[i] += s; else dbias[i - nw] += s;
}
static bool det_begin(const sehip_gemm_desc& d, int splits, hipStream_t st, sehip_gemm_desc& out) {
    out = d;
    if (!sehip_deterministic()) return true;
    const size_t stride = (size_t)d.Npad * d.K + d.Npad;
    float* sc = sehip_wgrad_scratch(st, (size_t)splits * stride * sizeof(float));
    if (!sc) {
        sehip_set_error(-2, "wgrad: the deterministic schedule could not get its partial arrays (allocation failed, or inside a stream capture)");
        return false;
    }
    if (hipMemsetAsync(sc, 0, (size_t)splits * stride * sizeof(float), st) != hipSuccess) {
        sehip_set_error(-2, "wgrad: clearing the deterministic partial arrays failed");
        return false;
    }
    out.dW = sc;
    out.dbias = d.dbias ? sc + (size_t)d.Npad * d.K : nullptr;
    out.dw_split_stride = (int64_t)stride;
    return true;
}
static void det_finish(const sehip_gemm_desc& d, const sehip_gemm_desc& used, int splits, hipStream_t st) {
    if (!used.dw_split_stride) return;
    const size_t nw = (size_t)d.Npad * d.K;
    const int nb = d.dbias ? d.Npad : 0;
    det_reduce_kernel<<<(unsigned)((nw + nb + 255) / 256), 256, 0, st>>>(used.dW, splits, (size_t)used.dw_split_stride, nw, nb, d.dW, d.dbias);
}

static int try_conv_wgrad(const sehip_gemm_desc& d_in, hipStream_t st) {
    const sehip_gemm_desc& d = d_in;
    static const bool disabled = getenv("SEHIP_NO_PATCH") != nullptr;
    if (disabled || d.cv_nf <= 0) return 0;
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    if ((C0 & 63) || (C1 & 63) || (d.Npad & 63) || d.J > 64 || (128 % d.J)) return 0;
    if (d.K != 2 * d.cv_nf * (C0 + C1)) return 0;
    // the dOut tile loader handles dense 8-column groups only
    const int JB = d.J, TB = 128 / JB;
    const int FR = (JB - 1) * d.fmul + d.cv_nf;
    if ((TB + 1) * FR * 8 > CV_MAXP * 256) return 0;
    const size_t lds = (size_t)128 * 80 * 2 + (size_t)(TB + 1) * FR * 80 * 2;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * ((d.TT + TB - 1) / TB);
    const int gx = (d.Npad >> 6) * ((C0 + C1) >> 6);
    // m-splits: a multiple of 8 (one group of splits per XCD) or 1, 2, 4 (8, 4, 2 XCDs share a split).  Every split flushes
    // its dW with atomics, and the weight gradients run beside the dependent chain on a second stream, so fewer, longer-lived
    // workgroups than "fill the GPU twice" pay.  Measured per step (ms), splits 4 / 8 / 16 / 32 with at least 128 workgroups
    // per launch: 5.95 / 5.97 / 6.08 / 6.17 (earlier in the round, with a slower chain, 16 was the optimum).
    static const int cw_splits = getenv("SEHIP_CW_SPLITS") ? atoi(getenv("SEHIP_CW_SPLITS")) : 8;
    int splits = cw_splits >= 8 ? cw_splits / 8 * 8 : (cw_splits >= 4 ? 4 : (cw_splits >= 2 ? 2 : 1));
    static const int cw_minwg = getenv("SEHIP_CW_MINWG") ? atoi(getenv("SEHIP_CW_MINWG")) : 128;
    if (gx * splits < cw_minwg) splits = (cw_minwg / gx + 7) / 8 * 8;  // few (n, channel) tiles (enc3, dec2): 16 splits would leave 32-64 workgroups
    while (splits < 8 && (gx * splits) % 8) splits <<= 1;
    const int tiles_per_wg = (MT + splits - 1) / splits;
    const int grid = gx * splits;  // splits beyond the data simply find an empty m range
#define CW_CASE(NF_)                                                                                              \
    case NF_: {                                                                                                   \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<NF_>),                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                     \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        sehip_note_kernel("conv_wgrad_kernel<%d>", NF_);                                                         \
        sehip_gemm_desc dd;                                                                                       \
        if (!det_begin(d, splits, st, dd)) return -1;                                                             \
        conv_wgrad_kernel<NF_><<<grid, 512, lds, st>>>(dd, TB, JB, FR, tiles_per_wg, splits);                             \
        det_finish(d, dd, splits, st);                                                                            \
        return 1;                                                                                                 \
    }
    switch (d.cv_nf) {
        CW_CASE(2)
        CW_CASE(3)
        CW_CASE(5)
        default: return 0;
    }
#undef CW_CASE
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad2_kernel: the same idea for the second convolution description (cv2_*: NKT consecutive time taps x NF consecutive
// row taps, stride 1 -- the output-parity classes of DCUnet's transposed convolutions, 6-12 taps over 128-192 input
// channels at up to 129 x 129 positions per clip).  The table-gathered wgrad_kernel re-reads dOut once per 64 k-columns
// (36 times for 12 taps x 192 channels) and gathers every input element once per tap from L2: 14.7 GB of L2 traffic for
// one class of the last decoder.  Here a workgroup owns 64 output channels x ONE 64-channel input chunk x ALL taps and walks
// over tiles of TB whole frames (TB*J <= 32 NKS rows, padded to a multiple of 32; J = 129 takes NKS = 5): the dOut tile and
// the input patch ((TB + NKT - 1) frames x (J + NF - 1) rows) are staged in LDS once per tile and every tap reuses them.
// Waves as in conv_wgrad_kernel: wave w accumulates dW[32 n][taps][16 channels] in registers (2 x NIT MFMA tiles).
// ------------------------------------------------------------------------------------------------
#define CW2_NPS 9   // patch pieces per thread (16 bytes each, 512 threads)
template <int NKT, int NF, int NKS>
__global__ __launch_bounds__(512) void conv_wgrad2_kernel(const sehip_gemm_desc d, int TB, int FR, int tiles_per_wg, int nsplit) {
    constexpr int NIT = NKT * NF;
    constexpr int TR = 32 * NKS;         // rows of a tile incl. the padding rows behind TB*J
    constexpr int GP = 80, PPW = 80;     // pitches (bf16 elements): 160 B, 8 consecutive rows on 8 disjoint 32-byte bank slots
    constexpr int GPT = (TR * 8 + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_raw* sG = reinterpret_cast<bf16_raw*>(smem);  // [TR][GP]
    bf16_raw* patch = sG + TR * GP;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w = wv & 3, nh = wv >> 2;
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int Ctot = C0 + C1;
    const int ntn = d.Npad >> 6;
    const int gx = ntn * (Ctot >> 6);
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    int xi, split;
    if (nsplit >= 8) { xi = rr % gx; split = (rr / gx) * 8 + xcd; }
    else { split = xcd % nsplit; xi = (xcd / nsplit) * (gx * nsplit >> 3) + rr; }
    const int nt = xi % ntn, cc = xi / ntn;
    const int n0 = nt * 64;
    const int J = d.J, rows_valid = TB * J;
    const int tblocks = (d.TT + TB - 1) / TB;
    const int B = d.M / (d.TT * J);
    const int MT = B * tblocks;
    const int mt_begin = split * tiles_per_wg, mt_end = min(MT, mt_begin + tiles_per_wg);

    const bool second = cc * 64 >= C0;
    const int sT = second ? d.src[1].T : d.src[0].T, sF = second ? d.src[1].F : d.src[0].F, sC = second ? C1 : C0;
    const int tlo = second ? d.src[1].tlo : d.src[0].tlo, thi = second ? d.src[1].thi : d.src[0].thi;
    const bf16_raw* sbase = reinterpret_cast<const bf16_raw*>(second ? d.src[1].ptr : d.src[0].ptr) + (cc * 64 - (second ? C0 : 0));
    const int tfirst = d.cv2_t0, f0 = d.cv2_fadd;
    // lattice steps of the source (1, or 2 for the tap-parity classes of DCUnet's strided encoder convolutions, src/model/dcunet.py:
    // 165-212): row (t, j) and tap (a', b') read source frame TS (t + a') + cv2_t0, source row FS (j + b') + cv2_fadd -- the patch is
    // the same (TB + NKT - 1) x (J + NF - 1) positions, taken every TS-th frame / FS-th row
    const int TS = d.tmul > 1 ? d.tmul : 1, FS = d.fmul > 1 ? d.fmul : 1;
    const int NPF = TB + NKT - 1;
    const int NP = NPF * FR * 8;

    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = 4 * (i16 & 3);
    int pbase[NKS][2], gbase[NKS][2];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = ks * 32 + 16 * h + 4 * g + q;          // (see conv_wgrad_kernel: 8 consecutive rows per 32-lane half)
            const int mc = m < rows_valid ? m : 0;               // padding rows carry dOut == 0: any valid patch row will do
            const int tl = mc / J, jl = mc - tl * J;
            pbase[ks][h] = (tl * FR + jl) * PPW + 16 * w + p4;
            gbase[ks][h] = m * GP + p4 + 32 * nh;
        }

    f32x4 acc[2][NIT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NIT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = d.dbias != nullptr && cc == 0 && w == 0;
    f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

    int p_goff[CW2_NPS], p_lds[CW2_NPS];  // element offset from the tile's first patch frame (or INT_MIN), LDS slot | frame << 16 (or -1)
#pragma unroll
    for (int i = 0; i < CW2_NPS; ++i) {
        const int idx = tid + 512 * i;
        p_goff[i] = INT_MIN; p_lds[i] = -1;
        if (idx < NP) {
            const int pp = idx / (FR * 8), rem = idx - pp * (FR * 8);
            const int f = f0 + FS * (rem >> 3);
            p_lds[i] = ((pp * FR + (rem >> 3)) * PPW + (rem & 7) * 8) | (pp << 16);
            if (f >= 0 && f < sF) p_goff[i] = (pp * TS * sF + f) * sC + (rem & 7) * 8;
        }
    }
    const bf16_raw* g_ptr[GPT];
    int g_tl[GPT];
    long g_bstride = 0;
    int g_tstride = 0;
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
        const int idx = tid + 512 * u;
        const int r = idx >> 3, gc = idx & 7;
        g_ptr[u] = nullptr; g_tl[u] = 0;
        if (r < rows_valid) {
            const int tl = r / J, jl = r - tl * J;
            g_tl[u] = tl;
            const int n = n0 + gc * 8;
            const sehip_nchunk c0 = d.ntab[n >> 2], c1 = d.ntab[(n >> 2) + 1];
            if (c0.nvalid == 4 && c1.nvalid == 4 && c1.dst == c0.dst && c1.coff == c0.coff + 4) {
                const sehip_dst& dd = c0.dst ? d.dst[1] : d.dst[0];
                RowPos rp;
                rp.b = 0; rp.t = tl; rp.j = jl; rp.jf = jl * d.fmul; rp.valid = true;
                g_ptr[u] = reinterpret_cast<const bf16_raw*>(dd.ptr) + dst_row_offset(dd, rp, d.fmul) + c0.coff;
                g_tstride = dd.F * dd.C * (dd.tmul > 1 ? dd.tmul : 1);
                g_bstride = (long)dd.T * dd.F * dd.C;
            }
        }
    }

    uint4 pr[CW2_NPS], gr[GPT];
#define CW2_FETCH(mt_)                                                                                             \
    {                                                                                                              \
        const int b_ = (mt_) / tblocks, t0_ = ((mt_) - b_ * tblocks) * TB;                                         \
        const bf16_raw* tb_ = sbase + ((long)b_ * sT + t0_ * TS + tfirst) * sF * sC;                               \
        _Pragma("unroll") for (int i = 0; i < CW2_NPS; ++i) {                                                      \
            pr[i] = make_uint4(0u, 0u, 0u, 0u);                                                                    \
            const int ts_ = t0_ * TS + tfirst + TS * (p_lds[i] >> 16);                                             \
            if (p_goff[i] != INT_MIN && ts_ >= tlo && ts_ < thi) pr[i] = *reinterpret_cast<const uint4*>(tb_ + p_goff[i]); \
        }                                                                                                          \
        _Pragma("unroll") for (int u = 0; u < GPT; ++u) {                                                          \
            gr[u] = make_uint4(0u, 0u, 0u, 0u);                                                                    \
            if (g_ptr[u] && t0_ + g_tl[u] < d.TT)                                                                  \
                gr[u] = *reinterpret_cast<const uint4*>(g_ptr[u] + b_ * g_bstride + (long)t0_ * g_tstride);        \
        }                                                                                                          \
    }

    if (mt_begin < mt_end) CW2_FETCH(mt_begin)
    for (int mt = mt_begin; mt < mt_end; ++mt) {
#pragma unroll
        for (int i = 0; i < CW2_NPS; ++i)
            if (p_lds[i] >= 0) *reinterpret_cast<uint4*>(&patch[p_lds[i] & 0xffff]) = pr[i];
#pragma unroll
        for (int u = 0; u < GPT; ++u) {
            const int idx = tid + 512 * u;
            if (idx < TR * 8) *reinterpret_cast<uint4*>(&sG[(idx >> 3) * GP + (idx & 7) * 8]) = gr[u];
        }
        __syncthreads();
        if (mt + 1 < mt_end) CW2_FETCH(mt + 1)   // in flight while this tile is consumed
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            bf16x8 gf[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[gbase[ks][0] + ni * 16]);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[gbase[ks][1] + ni * 16]);
                gf[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            if (do_bias) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) accb[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], BF16_ONES, accb[ni], 0, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int toff_e = ((it / NF) * FR + (it % NF)) * PPW;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&patch[pbase[ks][0] + toff_e]);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&patch[pbase[ks][1] + toff_e]);
                const bf16x8 xf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[ni][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], xf, acc[ni][it], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#undef CW2_FETCH

#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = n0 + 32 * nh + ni * 16 + 4 * (lane >> 4);
            const int k = it * Ctot + cc * 64 + 16 * w + (lane & 15);
#pragma unroll
            for (int u = 0; u < 4; ++u) atomicAdd(&d.dW[(size_t)(n + u) * d.K + k], acc[ni][it][u]);
        }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int u = 0; u < 4; ++u) atomicAdd(&d.dbias[n0 + 32 * nh + ni * 16 + 4 * (lane >> 4) + u], accb[ni][u]);
    }
}

static int try_conv_wgrad2(const sehip_gemm_desc& d, hipStream_t st) {
    static const bool disabled = getenv("SEHIP_NO_PATCH") != nullptr || getenv("SEHIP_NO_WGRAD2") != nullptr;
    if (disabled || d.cv2_nkt <= 0) return 0;
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    if ((C0 & 63) || (C1 & 63) || (d.Npad & 63) || d.fmul < 1 || d.fmul > 2 || d.tmul > 2) return 0;
    const int TSh = d.tmul > 1 ? d.tmul : 1;
    if (d.K != d.cv2_nkt * d.cv2_nf * (C0 + C1)) return 0;
    if (d.src[1].ptr && (d.src[1].T != d.src[0].T || d.src[1].F != d.src[0].F)) return 0;      // one grid for both sources
    const int J = d.J;
    int TB;
    if (J >= 128) TB = 1;
    else if (128 % J == 0) TB = 128 / J;
    else return 0;
    const int rows = TB * J;
    if (rows > 160) return 0;
    const int NKS = (rows + 31) / 32;
    const int FR = J + d.cv2_nf - 1;
    const int NPF = TB + d.cv2_nkt - 1;
    if (NPF * FR * 8 > CW2_NPS * 512 || NPF * FR * 80 > 65535) return 0;
    for (int s = 0; s < 2; ++s)
        if (d.src[s].ptr && (long)d.src[s].T * d.src[s].F * d.src[s].C * (NPF * TSh + 1) >= (1L << 31)) return 0;   // 32-bit piece offsets
    const size_t lds = (size_t)NKS * 32 * 80 * 2 + (size_t)NPF * FR * 80 * 2;
    if (lds > 160 * 1024) return 0;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * ((d.TT + TB - 1) / TB);
    const int gx = (d.Npad >> 6) * ((C0 + C1) >> 6);
    // one resident workgroup per CU (512 threads, >100 KB of LDS): about one round of workgroups, splits a multiple of 8
    int splits = (256 / gx) / 8 * 8;
    if (splits < 8) splits = 8;
    const int tiles_per_wg = (MT + splits - 1) / splits;
    const int grid = gx * splits;
#define CW2_CASE(NKT_, NF_, NKS_)                                                                                  \
    if (d.cv2_nkt == NKT_ && d.cv2_nf == NF_ && NKS == NKS_) {                                                     \
        static bool attr_set = false;                                                                              \
        if (!attr_set) {                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad2_kernel<NKT_, NF_, NKS_>),         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                     \
            attr_set = true;                                                                                       \
        }                                                                                                          \
        sehip_note_kernel("conv_wgrad2_kernel<%d, %d, %d>", NKT_, NF_, NKS_);                                      \
        conv_wgrad2_kernel<NKT_, NF_, NKS_><<<grid, 512, lds, st>>>(d, TB, FR, tiles_per_wg, splits);              \
        return 1;                                                                                                  \
    }
    CW2_CASE(4, 3, 5) CW2_CASE(3, 3, 5) CW2_CASE(4, 3, 4) CW2_CASE(3, 3, 4)
    CW2_CASE(4, 2, 4) CW2_CASE(3, 2, 4) CW2_CASE(4, 2, 5) CW2_CASE(3, 2, 5)
    CW2_CASE(2, 2, 4) CW2_CASE(2, 3, 4) CW2_CASE(3, 1, 4) CW2_CASE(2, 1, 4)
#undef CW2_CASE
    return 0;
}

// dOut pieces: mode 1 = eight dense channels, 2 = the two channels of a narrow layer as one dword, 3 = element by element
template <int GPT>
__device__ __forceinline__ RegTile<GPT> sw_fetch_dout(const sehip_gemm_desc& d, const bf16_raw* const (&g_ptr)[GPT], const int (&g_mode)[GPT],
                                                      const int (&g_tl)[GPT], long tile_off, int t0) {
    RegTile<GPT> t;
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (g_mode[u] && t0 + g_tl[u] < d.TT) {
            const bf16_raw* gp = g_ptr[u] + tile_off;
            const int mode = g_mode[u] & 0xff;
            if (mode == 1) v = *reinterpret_cast<const uint4*>(gp);
            else if (mode == 2) v.x = *reinterpret_cast<const unsigned*>(gp);
            else {
                const int gc = g_mode[u] >> 8;
                const sehip_nchunk c0 = d.ntab[(gc * 8) >> 2], c1 = d.ntab[((gc * 8) >> 2) + 1];
                unsigned h[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = e < c0.nvalid ? (unsigned)gp[e] : 0u;
                    h[4 + e] = e < c1.nvalid ? (unsigned)gp[c1.coff - c0.coff + e] : 0u;
                }
                v = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            }
        }
        t.v[u] = v;
    }
    return t;
}

// ------------------------------------------------------------------------------------------------
// conv_small_wgrad_kernel: weight gradient of the small-channel layers.  The WHOLE dW[BN][K] of the layer lives in
// the registers of one workgroup (wave w owns the 16-column k-tiles w, w+4, ...), which streams over 128-row m-tiles:
// dOut and the input patch are each read from HBM exactly once per workgroup pass; one atomic flush at the end.
// ------------------------------------------------------------------------------------------------
template <int BN, int KPW, int NPC>
__global__ __launch_bounds__(256) void conv_small_wgrad_kernel(const sehip_gemm_desc d, int TB, int JB, int FR, int tiles_per_wg) {
    constexpr int TN = BN / 16;
    constexpr int GP = BN == 16 ? 16 : BN + 16;  // row offsets = distinct multiples of 32 B for 8 consecutive rows
    constexpr int GCH = BN / 8;
    constexpr int GPT = (128 * GCH + 255) / 256;
    constexpr int MAXPC = NPC;  // 16-byte patch pieces per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int CT = C0 + C1;
    const int KR = 2 * d.cv_nf * CT;
    // patch pitch in elements: 8 consecutive (fmul 1) / alternate (fmul 2) rows on disjoint 32-byte bank slots
    const int PP = d.fmul == 2 ? (CT == 64 ? 72 : CT + 8) : (CT == 16 ? 16 : CT + 16);
    bf16_raw* sG = reinterpret_cast<bf16_raw*>(smem);
    bf16_raw* patch = sG + 128 * GP;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tblocks = (d.TT + TB - 1) / TB;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * tblocks;
    const int tmin0 = min(d.cv_toff[0][0], d.cv_toff[0][1]), tmin1 = min(d.cv_toff[1][0], d.cv_toff[1][1]);
    const int cp8 = CT >> 3;
    const int NP = (TB + 1) * FR * cp8;
    const int lgct = 31 - __clz(CT);
    const int nf = d.cv_nf;

    // Staging slots of this thread.  p_ptr: address of the piece for (b 0, first patch frame at t 0); p_pp: patch frame |
    // source << 16, or -1 for a piece that is always zero (frequency padding, beyond NP).  The tile loop adds one
    // per-source offset, checks the frame range with one unsigned compare and reads sehip_zero16 for anything invalid,
    // so it contains no divergent branch (with one wave per SIMD every branch and dependent instruction is exposed).
    const bf16_raw* p_ptr[MAXPC];
    int p_lds[MAXPC], p_pp[MAXPC];
    const int lds_dump = (TB + 1) * FR * PP;  // 16 spare bytes behind the patch take the stores of idx >= NP
#pragma unroll
    for (int u = 0; u < MAXPC; ++u) {
        const int idx = tid + 256 * u;
        p_pp[u] = -1; p_lds[u] = lds_dump; p_ptr[u] = nullptr;
        if (idx < NP) {
            const int pp = idx / (FR * cp8), rem = idx - pp * (FR * cp8);
            const int r = rem / cp8, c8 = rem - r * cp8;
            const bool second = c8 * 8 >= C0;
            const int sF = second ? d.src[1].F : d.src[0].F, sC = second ? C1 : C0;
            const int f = d.cv_fadd + r;
            p_lds[u] = (pp * FR + r) * PP + c8 * 8;
            if (f >= 0 && f < sF) {
                p_pp[u] = pp | (second ? 0x10000 : 0);
                p_ptr[u] = reinterpret_cast<const bf16_raw*>(second ? d.src[1].ptr : d.src[0].ptr) +
                           ((long)((second ? tmin1 : tmin0) + pp) * sF + f) * sC + (c8 * 8 - (second ? C0 : 0));
            }
        }
    }
    // transposed-read rows of this lane, and the patch offset of each owned k-tile
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = 4 * (i16 & 3);
    int pbase[4][2], gbase[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = ks * 32 + 16 * h + 4 * g + q;  // see conv_wgrad_kernel
            const int tl = m / JB, jl = m - tl * JB;
            pbase[ks][h] = (tl * FR + jl * d.fmul) * PP + p4;
            gbase[ks][h] = m * GP + p4;
        }
    int koff[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int k0 = 16 * (w + 4 * i);
        koff[i] = -1;
        if (k0 < KR) {
            const int it = k0 >> lgct, c = k0 & (CT - 1);
            const bool second = c >= C0;
            const int kt = it >= nf ? 1 : 0, tap = it - kt * nf;
            const int dt = second ? d.cv_toff[1][kt] - tmin1 : d.cv_toff[0][kt] - tmin0;
            koff[i] = (dt * FR + tap) * PP + c;
        }
    }

    f32x4 acc[TN][KPW];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < KPW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = d.dbias != nullptr && w == 0;  // column sums of dOut by MFMA against ones (every wave holds all n)
    f32x4 accb[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) accb[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // dOut staging slots of this thread: row / column chunk, source pointer of (b 0, t tl) and how to read it
    // (1: eight dense channels, 2: the two channels of a narrow layer as one dword, 3: element by element)
    const bf16_raw* g_ptr[GPT];
    int g_tl[GPT], g_mode[GPT], g_lds[GPT];
    const long g_bstride = (long)d.dst[0].T * d.dst[0].F * d.dst[0].C;
    const int g_tstride = d.dst[0].F * d.dst[0].C;
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
        const int idx = tid + 256 * u;
        g_ptr[u] = nullptr; g_tl[u] = 0; g_mode[u] = 0; g_lds[u] = -1;
        if (idx < 128 * GCH) {
            const int r = idx / GCH, gc = idx - r * GCH;
            const int tl = r / JB, jl = r - tl * JB;
            RowPos rp;
            rp.b = 0; rp.t = tl; rp.j = jl; rp.jf = jl * d.fmul; rp.valid = true;
            const sehip_nchunk c0 = d.ntab[(gc * 8) >> 2], c1 = d.ntab[((gc * 8) >> 2) + 1];
            g_ptr[u] = reinterpret_cast<const bf16_raw*>(d.dst[0].ptr) + dst_row_offset(d.dst[0], rp, d.fmul) + c0.coff;
            g_tl[u] = tl; g_lds[u] = r * GP + gc * 8;
            if (c0.nvalid == 4 && c1.nvalid == 4 && c1.coff == c0.coff + 4) g_mode[u] = 1;
            else if (c0.nvalid == 2 && c1.nvalid <= 0 && ((dst_row_offset(d.dst[0], rp, d.fmul) + c0.coff) & 1) == 0 &&
                     (g_tstride & 1) == 0 && (g_bstride & 1) == 0) g_mode[u] = 2;
            else if (c0.nvalid > 0 || c1.nvalid > 0) g_mode[u] = 3 | (gc << 8);
        }
    }

    const bf16_raw* zero_page = reinterpret_cast<const bf16_raw*>(&sehip_zero16);
    RegTile<NPC> pr;
    RegTile<GPT> gr;
#define SW_FETCH(mt_)                                                                                             \
    {                                                                                                             \
        const int b_ = (mt_) / tblocks, t0_ = ((mt_) - b_ * tblocks) * TB;                                        \
        const long off0_ = ((long)b_ * d.src[0].T + t0_) * d.src[0].F * C0;                                       \
        const long off1_ = C1 ? ((long)b_ * d.src[1].T + t0_) * d.src[1].F * C1 : 0;                              \
        pr = sw_fetch_patch<NPC>(p_ptr, p_pp, off0_, off1_, d.src[0].tlo - t0_ - tmin0, d.src[0].thi - d.src[0].tlo, \
                                 d.src[1].tlo - t0_ - tmin1, d.src[1].thi - d.src[1].tlo, zero_page);             \
        gr = sw_fetch_dout<GPT>(d, g_ptr, g_mode, g_tl, b_ * g_bstride + (long)t0_ * g_tstride, t0_);             \
    }

    const int mt_begin = blockIdx.x * tiles_per_wg, mt_end = min(MT, mt_begin + tiles_per_wg);
    if (mt_begin < mt_end) SW_FETCH(mt_begin)
    for (int mt = mt_begin; mt < mt_end; ++mt) {
#pragma unroll
        for (int u = 0; u < NPC; ++u) *reinterpret_cast<uint4*>(&patch[p_lds[u]]) = pr.v[u];
#pragma unroll
        for (int u = 0; u < GPT; ++u)
            if (g_lds[u] >= 0) *reinterpret_cast<uint4*>(&sG[g_lds[u]]) = gr.v[u];
        __syncthreads();
        if (mt + 1 < mt_end) SW_FETCH(mt + 1)   // next tile in flight while this one is multiplied
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 gf[TN];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[gbase[ks][0] + ni * 16]);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[gbase[ks][1] + ni * 16]);
                gf[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            if (do_bias) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) accb[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], BF16_ONES, accb[ni], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                const int ko = koff[i] < 0 ? 0 : koff[i];  // padded k-tiles read a valid address and are discarded
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&patch[pbase[ks][0] + ko]);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&patch[pbase[ks][1] + ko]);
                const bf16x8 xf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[ni][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ni], xf, acc[ni][i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* dWs = d.dW + (size_t)blockIdx.x * d.dw_split_stride;   // (deterministic schedule: one array per workgroup = m-split)
    float* dbs = d.dbias ? d.dbias + (size_t)blockIdx.x * d.dw_split_stride : nullptr;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            if (koff[i] < 0) continue;
            const int n = ni * 16 + 4 * (lane >> 4);
            const int k = 16 * (w + 4 * i) + (lane & 15);
#pragma unroll
            for (int u = 0; u < 4; ++u) atomicAdd(&dWs[(size_t)(n + u) * d.K + k], acc[ni][i][u]);
        }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int u = 0; u < 4; ++u) atomicAdd(&dbs[ni * 16 + 4 * (lane >> 4) + u], accb[ni][u]);
    }
#undef SW_FETCH
}

template <int BN, int KPW, int NPC>
static int launch_small_wgrad(const sehip_gemm_desc& d, int TB, int JB, int FR, int tiles_per_wg, int grid, size_t lds,
                              hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_wgrad_kernel<BN, KPW, NPC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        attr_set = true;
    }
    sehip_note_kernel("conv_small_wgrad_kernel<%d, %d, %d>", BN, KPW, NPC);
    sehip_gemm_desc dd;
    if (!det_begin(d, grid, st, dd)) return -1;
    conv_small_wgrad_kernel<BN, KPW, NPC><<<grid, 256, lds, st>>>(dd, TB, JB, FR, tiles_per_wg);
    det_finish(d, dd, grid, st);
    return 1;
}

static int try_conv_small_wgrad(const sehip_gemm_desc& d, hipStream_t st) {
    static const bool disabled = getenv("SEHIP_NO_PATCH") != nullptr || getenv("SEHIP_NO_SMALL") != nullptr;
    if (disabled || d.cv_nf <= 0 || d.dst[1].ptr) return 0;
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int CT = C0 + C1;
    if ((C0 & 15) || (C1 & 15) || CT > 128 || (CT & (CT - 1)) || d.Npad > 64 || CT < 16) return 0;
    if (d.J > 128 || (128 % d.J)) return 0;
    const int KR = 2 * d.cv_nf * CT;
    if (KR > d.K) return 0;
    const int JB = d.J, TB = 128 / JB;
    const int FR = (JB - 1) * d.fmul + d.cv_nf;
    if ((TB + 1) * FR * (CT >> 3) > 12 * 256) return 0;
    const size_t lds = (size_t)128 * (d.Npad + 16) * 2 + (size_t)(TB + 1) * FR * (CT + 16) * 2 + 16;  // + dump slot
    if (lds > 120 * 1024) return 0;
    const int kpw = d.K / 64;
    const int B = d.M / (d.TT * d.J);
    const int MT = B * ((d.TT + TB - 1) / TB);
    static const int sw_wgs = getenv("SEHIP_SW_WGS") ? atoi(getenv("SEHIP_SW_WGS")) : 256;
    int wgs = sw_wgs;
    if (wgs > MT) wgs = MT;
    const int tiles_per_wg = (MT + wgs - 1) / wgs;
    const int grid = (MT + tiles_per_wg - 1) / tiles_per_wg;
    const bool few = (TB + 1) * FR * (CT >> 3) <= 6 * 256;  // 16-byte patch pieces per thread: 6 or 12
#define SW(BN_, KPW_)                                                                                           \
    if (d.Npad == BN_ && kpw == KPW_)                                                                           \
        return few ? launch_small_wgrad<BN_, KPW_, 6>(d, TB, JB, FR, tiles_per_wg, grid, lds, st)               \
                   : launch_small_wgrad<BN_, KPW_, 12>(d, TB, JB, FR, tiles_per_wg, grid, lds, st);
    SW(16, 1) SW(16, 2) SW(16, 3) SW(16, 4) SW(16, 5) SW(16, 6) SW(16, 8) SW(16, 12)
    SW(32, 1) SW(32, 2) SW(32, 3) SW(32, 4) SW(32, 5) SW(32, 6) SW(32, 8) SW(32, 12)
    SW(64, 1) SW(64, 2) SW(64, 3) SW(64, 4) SW(64, 5)
#undef SW
    return 0;
}

// ------------------------------------------------------------------------------------------------
// narrow_wgrad_kernel: weight gradient of the FIRST encoder layer (2 input channels = re | im of the spectrogram, 16
// output channels): dW[16][2 x NF x 2].  320 numbers reduced over 1.3 M rows is VALU work, not MFMA work, and it is the
// tail of every step: it can only start when the last kernel of the backward chain has produced dOut and nothing is
// left to overlap it with (the table-gathered generic kernel took 110-150 us there).  A wave stages one frame (its 128
// dOut rows and the two input frames it touches) in a private LDS region -- no workgroup barrier in the frame loop --
// and lane (n, kt, c) accumulates the NF taps of dW[n][kt][.][c] over the frame's rows with a sliding window over the
// input column; the next frame's loads are in flight meanwhile.  Workgroups reduce over their waves through LDS before
// the atomics.
// ------------------------------------------------------------------------------------------------
#define NW_WAVES 8
template <int NF, int FMUL>
__global__ __launch_bounds__(64 * NW_WAVES) void narrow_wgrad_kernel(const sehip_gemm_desc d, int FRA /* staged input rows, multiple of 4 */,
                                                                      int fa /* first staged row (multiple of 4, <= cv_fadd) */,
                                                                      int frames_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int J = d.J;
    const int gbytes = J * 16 * 2, xbytes = 2 * FRA * 2 * 2;
    unsigned char* wbase = smem + (size_t)w * (gbytes + xbytes);
    bf16_raw* sG = reinterpret_cast<bf16_raw*>(wbase);             // [J][16]
    bf16_raw* sX = reinterpret_cast<bf16_raw*>(wbase + gbytes);    // [2 (kt)][FRA][2]
    // lane = n + 16 kt + 32 half: both input channels of (n, kt), rows [half J/2, (half + 1) J/2) of the frame
    const int n = lane & 15, kt = (lane >> 4) & 1, half = lane >> 5;
    const int gch = d.ntab[n >> 2].coff + (n & 3);  // channel of dOut that packed row n of dW belongs to

    const int sT = d.src[0].T, sF = d.src[0].F;
    const bf16_raw* xsrc = reinterpret_cast<const bf16_raw*>(d.src[0].ptr);
    const bf16_raw* gsrc = reinterpret_cast<const bf16_raw*>(d.dst[0].ptr);
    const int gpieces = J * 2;              // 16-byte pieces of a dOut frame (8 channels each)
    const int xpieces = FRA >> 2;           // 16-byte pieces (4 rows x 2 ch) of one input frame
    constexpr int GPL = 4, XPL = 2;         // pieces per lane: J <= 128, FRA <= 512

    float acc[NF][2], accb = 0.f;
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i][0] = acc[i][1] = 0.f;

    const int wave_id = blockIdx.x * NW_WAVES + w, nwaves = gridDim.x * NW_WAVES;
    uint4 gr[GPL], xr[2][XPL];
#define NW_FETCH(fr_)                                                                                              \
    {                                                                                                              \
        const int b_ = (fr_) / d.TT, t_ = (fr_) - b_ * d.TT;                                                       \
        _Pragma("unroll") for (int u = 0; u < GPL; ++u) {                                                          \
            const int idx = lane + 64 * u;                                                                         \
            gr[u] = make_uint4(0u, 0u, 0u, 0u);                                                                    \
            if (idx < gpieces) {                                                                                   \
                const int j = idx >> 1, h = idx & 1;                                                               \
                const long off = (((long)b_ * d.dst[0].T + t_ + d.dst[0].toff) * d.dst[0].F + (long)j * d.dst[0].fmul + d.dst[0].fadd) * d.dst[0].C; \
                gr[u] = *reinterpret_cast<const uint4*>(gsrc + off + 8 * h);                                       \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                            \
            const int ts = t_ + d.cv_toff[0][k];                                                                   \
            const bool tok = ts >= d.src[0].tlo && ts < d.src[0].thi;                                              \
            _Pragma("unroll") for (int u = 0; u < XPL; ++u) {                                                      \
                const int idx = lane + 64 * u;                                                                     \
                const int f = fa + 4 * idx;                                                                        \
                xr[k][u] = make_uint4(0u, 0u, 0u, 0u);                                                             \
                if (tok && idx < xpieces && f >= 0 && f + 3 < sF)                                                  \
                    xr[k][u] = *reinterpret_cast<const uint4*>(xsrc + (((long)b_ * sT + ts) * sF + f) * 2);        \
                else if (tok && idx < xpieces && f + 3 >= 0 && f < sF) {  /* piece straddles the edge: row by row */ \
                    unsigned q[4];                                                                                 \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                  \
                        q[r] = (f + r >= 0 && f + r < sF) ? *reinterpret_cast<const unsigned*>(xsrc + (((long)b_ * sT + ts) * sF + f + r) * 2) : 0u; \
                    xr[k][u] = make_uint4(q[0], q[1], q[2], q[3]);                                                 \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }

    int fr = wave_id;
    if (fr < frames_total) NW_FETCH(fr)
    for (; fr < frames_total; fr += nwaves) {
        // registers -> this wave's LDS region (only this wave reads it: no barrier, the LDS pipe is in order per wave)
#pragma unroll
        for (int u = 0; u < GPL; ++u) {
            const int idx = lane + 64 * u;
            if (idx < gpieces) *reinterpret_cast<uint4*>(&sG[(idx >> 1) * 16 + 8 * (idx & 1)]) = gr[u];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int u = 0; u < XPL; ++u) {
                const int idx = lane + 64 * u;
                if (idx < xpieces) *reinterpret_cast<uint4*>(&sX[(k * FRA + 4 * idx) * 2]) = xr[k][u];
            }
        if (fr + nwaves < frames_total) NW_FETCH(fr + nwaves)

        // row j of the frame reads input rows j*FMUL + tap + (cv_fadd - fa), tap < NF; one dword = both channels of a row.
        // Consecutive rows of a lane shift the tap window by FMUL rows: only FMUL new dwords per row.
        const unsigned* xcol = reinterpret_cast<const unsigned*>(sX) + kt * FRA + (d.cv_fadd - fa);
        const bf16_raw* gcol = sG + gch;
        const int j0 = half * (J >> 1), j1 = j0 + (J >> 1);
        float xw[NF][2];
#pragma unroll
        for (int tap = 0; tap < NF; ++tap) {
            const unsigned q = xcol[j0 * FMUL + tap];
            xw[tap][0] = __uint_as_float(q << 16); xw[tap][1] = __uint_as_float(q & 0xffff0000u);
        }
#pragma unroll 8
        for (int j = j0; j < j1; ++j) {
            const float g = bf2f(gcol[j * 16]);
            accb += g;
#pragma unroll
            for (int tap = 0; tap < NF; ++tap) { acc[tap][0] += g * xw[tap][0]; acc[tap][1] += g * xw[tap][1]; }
#pragma unroll
            for (int tap = 0; tap + FMUL < NF; ++tap) { xw[tap][0] = xw[tap + FMUL][0]; xw[tap][1] = xw[tap + FMUL][1]; }
#pragma unroll
            for (int tap = (NF > FMUL ? NF - FMUL : 0); tap < NF; ++tap) {
                const unsigned q = xcol[(j + 1) * FMUL + tap];   // the row after the last one reads staged padding
                xw[tap][0] = __uint_as_float(q << 16); xw[tap][1] = __uint_as_float(q & 0xffff0000u);
            }
        }
    }
#undef NW_FETCH

    // reduce the waves (and the two row halves) of the workgroup through LDS, then one atomic per dW entry and workgroup
    __syncthreads();
    constexpr int NR = 2 * NF + 1;
    float* red = reinterpret_cast<float*>(smem);  // [NW_WAVES][64][NR]
#pragma unroll
    for (int i = 0; i < NF; ++i) { red[(w * 64 + lane) * NR + 2 * i] = acc[i][0]; red[(w * 64 + lane) * NR + 2 * i + 1] = acc[i][1]; }
    red[(w * 64 + lane) * NR + 2 * NF] = accb;
    __syncthreads();
    if (w == 0 && lane < 32) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW_WAVES; ++ww) v += red[(ww * 64 + lane) * NR + i] + red[(ww * 64 + lane + 32) * NR + i];
            if (i < 2 * NF) atomicAdd(&d.dW[(size_t)n * d.K + kt * 16 + i], v);  // K order of a 2-channel source: kt*16 + tap*2 + c
            else if (kt == 0 && d.dbias) atomicAdd(&d.dbias[n], v);
        }
    }
}


// sum e (0-3: kt 0, 4-7: kt 1, 8-11: channel sums; q = e & 3) of lane (g = lane >> 4, column i16 = lane & 15) of the MFMA kernel
// below = D[dOut channel 4 g + q][column]: added to the packed row of dW that the column table maps to that channel
template <int NF>
__device__ __forceinline__ void narrow_wgrad_add(const sehip_gemm_desc& d, int e, int lane, float v) {
    const int g = lane >> 4, i16 = lane & 15, q = e & 3;
    int n = -1;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (d.ntab[m].coff == 4 * g) n = 4 * m + q;
    if (n < 0) return;
    if (e < 8) {
        if (i16 < 2 * NF) atomicAdd(&d.dW[(size_t)n * d.K + (e >> 2) * 16 + i16], v);   // K order of a 2-channel source: kt*16 + tap*2 + c
    } else if (i16 == 0 && d.dbias) atomicAdd(&d.dbias[n], v);
}

// narrow_wgrad_mfma_kernel: the same product on the matrix cores.  dW[16 n][(kt, tap, c)] = sum over rows of dOut[row][n] *
// x[frame + kt offset][FMUL j + tap][c] is a 16 x (2 x 16) x rows GEMM: per 32 rows of the frame one transposed LDS read pair
// gives the dOut operand ([n][32 rows]), and for each kt the lane of column (tap, c) gathers its 8 rows of the input column
// (dwords FMUL rows apart, the channel's half picked with v_perm) -- 12 MFMAs and ~80 LDS / VALU instructions per frame
// instead of ~1 300 VALU instructions, so the launch is bound by reading dOut (the VALU kernel above: 68 us at the headline
// shape, 5x its HBM time; it stays for frames that are not a multiple of 32 rows).
// BN (sehip_gemm_desc.bn_dz ...): dOut is the ComplexBatchNorm + PReLU backward pass's output (cbn_bwd_apply_kernel's arithmetic, in
// its order, rounded to bf16 as that kernel stores it) computed while the frame is staged: a lane fetches the same 16-byte pieces of
// bn_dz and bn_y it would have fetched of dOut, trades them with the lane that holds the row's other part (real | imaginary: lane ^ 1)
// and writes its half of the row's eight complex channels.  The 8 channels' forward records are wave-uniform, the five backward
// coefficients of a lane's part live in 40 registers.
template <int NF, int FMUL, bool BN>
__global__ __launch_bounds__(64 * NW_WAVES) void narrow_wgrad_mfma_kernel(const sehip_gemm_desc d, int FRA, int fa, int frames_total,
                                                                           float* __restrict__ parts /* [workgroups][12][64] or NULL */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int J = d.J;
    const int gbytes = J * 16 * 2, xbytes = 2 * FRA * 2 * 2;
    unsigned char* wbase = smem + (size_t)w * (gbytes + xbytes);
    bf16_raw* sG = reinterpret_cast<bf16_raw*>(wbase);             // [J][16]
    bf16_raw* sX = reinterpret_cast<bf16_raw*>(wbase + gbytes);    // [2 (kt)][FRA][2]
    const int g = lane >> 4, i16 = lane & 15;
    const int tap = (i16 >> 1) < NF ? (i16 >> 1) : NF - 1;         // columns >= 2 NF are not stored
    const unsigned sel = (i16 & 1) ? 0x07060302u : 0x05040100u;    // the channel's half of two consecutive rows' dwords

    const int sT = d.src[0].T, sF = d.src[0].F;
    const bf16_raw* xsrc = reinterpret_cast<const bf16_raw*>(d.src[0].ptr);
    const bf16_raw* gsrc = reinterpret_cast<const bf16_raw*>(BN ? d.bn_dz : d.dst[0].ptr);
    const bf16_raw* ysrc = reinterpret_cast<const bf16_raw*>(d.bn_y);
    const int gpieces = J * 2;              // 16-byte pieces of a dOut frame (8 channels each)
    float4 bn_zc[BN ? 8 : 1], bn_mb[BN ? 8 : 1];   // forward records of the 8 complex channels (COEF_STRIDE = 16 floats per channel)
    float bn_p[BN ? 8 : 1][5];                     // this lane's part of the backward records: out = p0 dr + p1 di + p2 cr + p3 ci + p4
    float bn_a = 0.f;
    if (BN) {
        bn_a = d.bn_slope[0];
        const bool im = lane & 1;                   // piece index = lane + 64 u: its part is the lane's parity
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            bn_zc[c] = reinterpret_cast<const float4*>(d.bn_coef)[4 * c];
            bn_mb[c] = reinterpret_cast<const float4*>(d.bn_coef)[4 * c + 1];
            const float4 A = reinterpret_cast<const float4*>(d.bn_bcoef)[4 * c], E = reinterpret_cast<const float4*>(d.bn_bcoef)[4 * c + 1];
            const float ki = d.bn_bcoef[16 * c + 8];
            bn_p[c][0] = im ? A.z : A.x; bn_p[c][1] = im ? A.w : A.y; bn_p[c][2] = im ? E.y : E.x; bn_p[c][3] = im ? E.z : E.y;
            bn_p[c][4] = im ? ki : E.w;
        }
    }
    const int xpieces = FRA >> 2;           // 16-byte pieces (4 rows x 2 ch) of one input frame
    constexpr int GPL = 4, XPL = 2;         // pieces per lane: J <= 128, FRA <= 512

    f32x4 acc[2], accb = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[0] = acc[1] = accb;

    const int wave_id = blockIdx.x * NW_WAVES + w, nwaves = gridDim.x * NW_WAVES;
    uint4 grA[GPL], xrA[2][XPL], grB[GPL], xrB[2][XPL];     // two frames in flight per wave
    uint4 yrA[BN ? GPL : 1], yrB[BN ? GPL : 1];
#define NWM_FETCH(fr_, gr, xr, yr)                                                                                 \
    {                                                                                                              \
        const int b_ = (fr_) / d.TT, t_ = (fr_) - b_ * d.TT;                                                       \
        _Pragma("unroll") for (int u = 0; u < GPL; ++u) {                                                          \
            const int idx = lane + 64 * u;                                                                         \
            gr[u] = make_uint4(0u, 0u, 0u, 0u);                                                                    \
            if (idx < gpieces) {                                                                                   \
                const int j = idx >> 1, h = idx & 1;                                                               \
                const long off = (((long)b_ * d.dst[0].T + t_ + d.dst[0].toff) * d.dst[0].F + (long)j * d.dst[0].fmul + d.dst[0].fadd) * d.dst[0].C; \
                gr[u] = *reinterpret_cast<const uint4*>(gsrc + off + 8 * h);                                       \
                if (BN) yr[u] = *reinterpret_cast<const uint4*>(ysrc + off + 8 * h);                               \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                            \
            const int ts = t_ + d.cv_toff[0][k];                                                                   \
            const bool tok = ts >= d.src[0].tlo && ts < d.src[0].thi;                                              \
            _Pragma("unroll") for (int u = 0; u < XPL; ++u) {                                                      \
                const int idx = lane + 64 * u;                                                                     \
                const int f = fa + 4 * idx;                                                                        \
                xr[k][u] = make_uint4(0u, 0u, 0u, 0u);                                                             \
                if (tok && idx < xpieces && f >= 0 && f + 3 < sF)                                                  \
                    xr[k][u] = *reinterpret_cast<const uint4*>(xsrc + (((long)b_ * sT + ts) * sF + f) * 2);        \
                else if (tok && idx < xpieces && f + 3 >= 0 && f < sF) {  /* piece straddles the edge: row by row */ \
                    unsigned q[4];                                                                                 \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                  \
                        q[r] = (f + r >= 0 && f + r < sF) ? *reinterpret_cast<const unsigned*>(xsrc + (((long)b_ * sT + ts) * sF + f + r) * 2) : 0u; \
                    xr[k][u] = make_uint4(q[0], q[1], q[2], q[3]);                                                 \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }

    auto stage = [&](const uint4 (&gr)[GPL], const uint4 (&xr)[2][XPL], const uint4 (&yr)[BN ? GPL : 1]) {
        // registers -> this wave's LDS region (only this wave reads it: no barrier, the LDS pipe is in order per wave)
#pragma unroll
        for (int u = 0; u < GPL; ++u) {
            const int idx = lane + 64 * u;
            uint4 go = gr[u];
            if (BN) {
                // (every lane takes part in the exchange: gpieces is even, so a lane and its partner are inside or outside together)
                const unsigned gm[4] = {gr[u].x, gr[u].y, gr[u].z, gr[u].w}, ym[4] = {yr[u].x, yr[u].y, yr[u].z, yr[u].w};
                unsigned o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned gp = (unsigned)__shfl_xor((int)gm[e], 1, 64), yp = (unsigned)__shfl_xor((int)ym[e], 1, 64);
                    const bool im = lane & 1;
                    const unsigned g_re = im ? gp : gm[e], g_im = im ? gm[e] : gp, y_re = im ? yp : ym[e], y_im = im ? ym[e] : yp;
                    float ov[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int c = 2 * e + hh;
                        const float xr_ = hh ? __uint_as_float(y_re & 0xffff0000u) : __uint_as_float(y_re << 16);
                        const float xi_ = hh ? __uint_as_float(y_im & 0xffff0000u) : __uint_as_float(y_im << 16);
                        float dr = hh ? __uint_as_float(g_re & 0xffff0000u) : __uint_as_float(g_re << 16);
                        float di = hh ? __uint_as_float(g_im & 0xffff0000u) : __uint_as_float(g_im << 16);
                        const float cr = xr_ - bn_mb[c].x, ci = xi_ - bn_mb[c].y;
                        const float vr = bn_zc[c].x * cr + bn_zc[c].y * ci + bn_mb[c].z;
                        const float vi = bn_zc[c].z * cr + bn_zc[c].w * ci + bn_mb[c].w;
                        if (!(vr > 0.f)) dr *= bn_a;
                        if (!(vi > 0.f)) di *= bn_a;
                        ov[hh] = bn_p[c][0] * dr + bn_p[c][1] * di + bn_p[c][2] * cr + bn_p[c][3] * ci + bn_p[c][4];
                    }
                    o[e] = pack_bf2(ov[0], ov[1]);
                }
                go = make_uint4(o[0], o[1], o[2], o[3]);
            }
            if (idx < gpieces) *reinterpret_cast<uint4*>(&sG[(idx >> 1) * 16 + 8 * (idx & 1)]) = go;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int u = 0; u < XPL; ++u) {
                const int idx = lane + 64 * u;
                if (idx < xpieces) *reinterpret_cast<uint4*>(&sX[(k * FRA + 4 * idx) * 2]) = xr[k][u];
            }
        asm volatile("" ::: "memory");
    };
    auto multiply = [&]() {
        const unsigned* xcol = reinterpret_cast<const unsigned*>(sX) + (d.cv_fadd - fa) + tap;
        for (int ks = 0; ks < (J >> 5); ++ks) {
            const int m0 = 32 * ks + 8 * g;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[(m0 + (i16 >> 2)) * 16 + 4 * (i16 & 3)]);
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)&sG[(m0 + 4 + (i16 >> 2)) * 16 + 4 * (i16 & 3)]);
            const bf16x8 gf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, BF16_ONES, accb, 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const unsigned* xc = xcol + kt * FRA + FMUL * m0;
                unsigned q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = xc[FMUL * i];
                const uint4 xv = make_uint4(__builtin_amdgcn_perm(q[1], q[0], sel), __builtin_amdgcn_perm(q[3], q[2], sel),
                                            __builtin_amdgcn_perm(q[5], q[4], sel), __builtin_amdgcn_perm(q[7], q[6], sel));
                acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, __builtin_bit_cast(bf16x8, xv), acc[kt], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
    };
    int fr = wave_id;
    if (fr < frames_total) NWM_FETCH(fr, grA, xrA, yrA)
    if (fr + nwaves < frames_total) NWM_FETCH(fr + nwaves, grB, xrB, yrB)
    for (; fr < frames_total; fr += 2 * nwaves) {
        stage(grA, xrA, yrA);
        if (fr + 2 * nwaves < frames_total) NWM_FETCH(fr + 2 * nwaves, grA, xrA, yrA)
        multiply();
        if (fr + nwaves >= frames_total) break;
        stage(grB, xrB, yrB);
        if (fr + 3 * nwaves < frames_total) NWM_FETCH(fr + 3 * nwaves, grB, xrB, yrB)
        multiply();
    }
#undef NWM_FETCH

    // a lane holds D[channel 4 g + q][column i16] of both kt and the channel sums: reduce the workgroup's waves through LDS; then
    // the workgroup's 12 x 64 sums go to its row of the partial array (narrow_wgrad_reduce_kernel adds the rows into dW).  Without
    // a partial array: one atomic per dW entry and workgroup -- 256 workgroups x 16 entries per cache line, and atomics on one
    // line are served one after the other: 28 of this kernel's 33 us were that queue.
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [NW_WAVES][12][64]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        red[(w * 12 + q) * 64 + lane] = acc[0][q];
        red[(w * 12 + 4 + q) * 64 + lane] = acc[1][q];
        red[(w * 12 + 8 + q) * 64 + lane] = accb[q];
    }
    __syncthreads();
    for (int e = w; e < 12; e += NW_WAVES) {
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW_WAVES; ++ww) v += red[(ww * 12 + e) * 64 + lane];
        if (parts) parts[((size_t)blockIdx.x * 12 + e) * 64 + lane] = v;
        else narrow_wgrad_add<NF>(d, e, lane, v);
    }
}

// rows of the partial array -> dW / dbias: workgroup = 16 sums x 16 groups of rows
template <int NF>
__global__ __launch_bounds__(256) void narrow_wgrad_reduce_kernel(const sehip_gemm_desc d, const float* __restrict__ parts, int nparts) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int out = blockIdx.x * 16 + o;                 // (e, lane) = (out >> 6, out & 63)
    float v = 0.f;
    for (int p0 = grp; p0 < nparts; p0 += 256) {          // 16 independent loads in flight per thread
        float t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = p0 + 16 * i < nparts ? parts[(size_t)(p0 + 16 * i) * 768 + out] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += t[i];
    }
    red[grp][o] = v;
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int i = 1; i < 16; ++i) v += red[i][o];
        narrow_wgrad_add<NF>(d, out >> 6, out & 63, v);
    }
}

float* sehip_wgrad_scratch(hipStream_t st, size_t bytes);   // csrc/wgrad3.hip: per-stream pool of partial arrays

static int try_narrow_wgrad(const sehip_gemm_desc& d, hipStream_t st) {
    static const bool disabled = getenv("SEHIP_NO_NARROW") != nullptr;
    if (disabled || d.cv_nf <= 0 || d.src[0].C != 2 || d.src[1].ptr || d.dst[1].ptr) return 0;
    if (d.N != 16 || d.Npad != 16 || d.J > 128 || d.K < 32 || d.cv_nf > 8) return 0;
    const bool bn = d.bn_dz != nullptr;
    if (bn && (!d.bn_y || !d.bn_coef || !d.bn_bcoef || !d.bn_slope || d.dst[0].C != 16 || d.dst[0].tmul > 1)) return 0;
    if (d.dst[0].is_f32 || (d.dst[0].C & 7)) return 0;
    const int FR = (d.J - 1) * d.fmul + d.cv_nf;
    const int fa = (d.cv_fadd >= 0 ? d.cv_fadd / 4 : -((-d.cv_fadd + 3) / 4)) * 4;
    const int FRA = ((d.cv_fadd - fa) + FR + d.fmul + 3) / 4 * 4;  // + fmul: the window refill after a lane's last row stays inside
    if (FRA > 512 || (d.src[0].F & 3)) return 0;
    const int B = d.M / (d.TT * d.J);
    const int frames = B * d.TT;
    const size_t per_wave = (size_t)d.J * 16 * 2 + (size_t)2 * FRA * 2 * 2;
    size_t lds = per_wave * NW_WAVES;
    const size_t red = (size_t)NW_WAVES * 64 * (2 * 8 + 1) * 4;
    if (lds < red) lds = red;
    if (lds > 64 * 1024) return 0;
    int grid = (frames + NW_WAVES - 1) / NW_WAVES;
    static const int cap = getenv("SEHIP_NARROW_WGS") ? atoi(getenv("SEHIP_NARROW_WGS")) : 256;
    if (grid > cap) grid = cap;
    if (d.cv_nf != 5 || d.fmul != 2 || (d.J & 1)) return 0;
    static const bool no_mfma = getenv("SEHIP_NO_NARROW_MFMA") != nullptr;
    // (the MFMA build gathers 8 rows FMUL apart above a 32-row block's first tap: inside the staged FRA rows for J % 32 == 0)
    if (!no_mfma && (d.J & 31) == 0) {
        sehip_note_kernel("narrow_wgrad_mfma_kernel<%d>", d.cv_nf);
        static const bool atomic_flush = getenv("SEHIP_NARROW_ATOMIC_FLUSH") != nullptr;
        float* parts = atomic_flush ? nullptr : sehip_wgrad_scratch(st, (size_t)grid * 768 * sizeof(float));
        if (bn) narrow_wgrad_mfma_kernel<5, 2, true><<<grid, 64 * NW_WAVES, lds, st>>>(d, FRA, fa, frames, parts);
        else narrow_wgrad_mfma_kernel<5, 2, false><<<grid, 64 * NW_WAVES, lds, st>>>(d, FRA, fa, frames, parts);
        if (parts) narrow_wgrad_reduce_kernel<5><<<48, 256, 0, st>>>(d, parts, grid);
        return 1;
    }
    if (bn) return 0;                                   // (only the MFMA build computes dOut on the way in: sehip_wgrad reports it)
    sehip_note_kernel("narrow_wgrad_kernel<%d>", d.cv_nf);
    narrow_wgrad_kernel<5, 2><<<grid, 64 * NW_WAVES, lds, st>>>(d, FRA, fa, frames);
    return 1;
}

extern "C" int sehip_wgrad(const sehip_gemm_desc* d, void* stream);

// The weight gradients of two products over the SAME sources and the same dOut tensor (the two output-row parities of a transposed
// convolution, src/model/dccrn.py:387-450): one streaming launch where the library has one (csrc/convt.hip), otherwise the two calls.
extern "C" int sehip_wgrad_pair(const sehip_gemm_desc* a, const sehip_gemm_desc* b, void* stream) {
    if (int e = check_desc("wgrad_pair", a)) return e;
    if (int e = check_desc("wgrad_pair", b)) return e;
    SEHIP_REQUIRE(a->dW != nullptr && b->dW != nullptr, "wgrad_pair: missing dW");
    if (sehip_try_wgradt_stream(*a, *b, (hipStream_t)stream)) {
        SEHIP_CHECK_LAUNCH("wgrad_pair(stream)");
        return 0;
    }
    if (int e = sehip_wgrad(a, stream)) return e;
    return sehip_wgrad(b, stream);
}

extern "C" int sehip_wgrad(const sehip_gemm_desc* d, void* stream) {
    if (int e = check_desc("wgrad", d)) return e;
    SEHIP_REQUIRE(d->dW != nullptr, "wgrad: missing dW");
    SEHIP_REQUIRE(!d->dst[0].is_f32 && !(d->dst[1].ptr && d->dst[1].is_f32), "wgrad: dOut must be bf16");
    hipStream_t st = (hipStream_t)stream;
    if (try_narrow_wgrad(*d, st)) {
        SEHIP_CHECK_LAUNCH("wgrad(narrow)");
        return 0;
    }
    SEHIP_REQUIRE(d->bn_dz == nullptr, "wgrad: dOut computed from bn_dz / bn_y (sehip.h) is built into narrow_wgrad_mfma_kernel only, and this "
                                        "product does not qualify for it (2-channel source, 16 outputs, 5 row taps at stride 2, J a multiple of 32)");
    const bool det = sehip_deterministic() != 0;
    // (deterministic schedule: conv_wgrad_v3's bias sums and conv_wgrad2 / dense_wgrad flush with plain atomics: those products take
    //  the kernels below, which keep one array per m-split)
    if (!det && sehip_try_conv_wgrad_v3(*d, st)) {
        SEHIP_CHECK_LAUNCH("wgrad(conv v3)");
        return 0;
    }
    if (int r = try_conv_wgrad(*d, st)) {
        if (r < 0) return -2;
        SEHIP_CHECK_LAUNCH("wgrad(conv)");
        return 0;
    }
    if (!det && try_conv_wgrad2(*d, st)) {
        SEHIP_CHECK_LAUNCH("wgrad(conv2)");
        return 0;
    }
    if (sehip_try_wgrads_stream(*d, st)) {
        SEHIP_CHECK_LAUNCH("wgrad(stream)");
        return 0;
    }
    if (int r = try_conv_small_wgrad(*d, st)) {
        if (r < 0) return -2;
        SEHIP_CHECK_LAUNCH("wgrad(conv-small)");
        return 0;
    }
    if (!det && d->dense_rows && sehip_try_dense_wgrad(*d, st)) {
        SEHIP_CHECK_LAUNCH("wgrad(dense)");
        return 0;
    }
    int ntiles, bnw;
    if (d->Npad == 16) bnw = 16; else if (d->Npad == 32) bnw = 32; else if (d->Npad == 64) bnw = 64; else bnw = 128;
    SEHIP_REQUIRE(d->Npad % bnw == 0, "wgrad: Npad=%d must be 16, 32, 64 or a multiple of 128", d->Npad);
    ntiles = d->Npad / bnw;
    // 256 k-columns per staged dOut slab (KQ = 4) quarter the dOut re-reads of long-K products, but measured SLOWER on the DCUnet
    // weight gradients (B=64: 1124 vs 914 us per launch): what bounds them is the per-tap gather of the input, not dOut, and
    // the wide tile has a quarter of the workgroups.  Opt-in for experiments only.
    static const int wide = getenv("SEHIP_WIDE_WGRAD") ? atoi(getenv("SEHIP_WIDE_WGRAD")) : 0;   // 2 or 4: k-columns per slab / 64
    const int kq = (wide >= 4 && d->K >= 1024) ? 4 : ((wide >= 2 && d->K >= 1024) ? 2 : 1);
    const int ktiles = cdiv(d->K, 64 * kq);
    // split m so that the grid has ~2048 workgroups, at least 256 rows each
    static const int gw_env = getenv("SEHIP_GW_WGS") ? atoi(getenv("SEHIP_GW_WGS")) : 0;
    // every m-split adds its whole dW tile with fp32 atomics, and the weight gradients run beside the step's dependent chain on the
    // second stream: for the dense row spaces of the 1-D models (J == 1: ConvTasNet's 51 168 rows, Demucs' 736 .. 775 488) ~256
    // workgroups with 1 500 - 3 000 rows each beat 2048 small ones -- Demucs 25.35 -> 24.03 ms per step (128: 26.65), ConvTasNet
    // 5.19 -> 4.88 (128: 4.79, 96: 5.28), e0.rw of Demucs alone 275 -> 161 us; DCUnet's 2-D products keep the 2048 (512: +1 %)
    // (round 3, with ConvTasNet's chain 20 % shorter, its small products -- four (n, k) tiles -- are best at ~160: 96: 4.28, 128: 3.68,
    //  160: 3.59, 192: 3.79, 256: 3.75 ms per step; Demucs keeps 256: 128: 23.5, 192: 21.1, 256: 20.7)
    //  -- the plan says so through the descriptor's wg_hint)
    const int gw_wgs = gw_env ? gw_env : (d->wg_hint > 0 ? d->wg_hint : ((d->J == 1 && d->cv_nf == 0 && d->cv2_nkt == 0) ? 256 : 2048));
    long want = gw_wgs / ((long)ntiles * ktiles);
    if (want < 1) want = 1;
    long mpb = ((d->M + want - 1) / want + 63) / 64 * 64;
    if (mpb < 256) mpb = 256;
    const int splits = cdiv(d->M, mpb);
    dim3 grid(ntiles, ktiles, splits);
    sehip_note_kernel("wgrad_kernel<%d, %d, %d, %d>", bnw, bnw >= 64 ? 2 : 1, bnw >= 64 ? 2 : 4, kq);
    sehip_gemm_desc dd_;
    if (!det_begin(*d, splits, st, dd_)) return -2;
    const sehip_gemm_desc* dorig = d;
    d = &dd_;
    if (kq == 2) {
        if (bnw == 64) wgrad_kernel<64, 2, 2, 2><<<grid, 256, 0, st>>>(*d, (int)mpb);
        else if (bnw == 128) wgrad_kernel<128, 2, 2, 2><<<grid, 256, 0, st>>>(*d, (int)mpb);
        else return sehip_set_error(-1, "wgrad: the 128-column slab variant is built for 64 / 128 output columns only");
    } else if (kq == 4) {
        if (bnw == 16) wgrad_kernel<16, 1, 4, 4><<<grid, 256, 0, st>>>(*d, (int)mpb);
        else if (bnw == 32) wgrad_kernel<32, 1, 4, 4><<<grid, 256, 0, st>>>(*d, (int)mpb);
        else if (bnw == 64) wgrad_kernel<64, 2, 2, 4><<<grid, 256, 0, st>>>(*d, (int)mpb);
        else wgrad_kernel<128, 2, 2, 4><<<grid, 256, 0, st>>>(*d, (int)mpb);
    } else {
        if (bnw == 16) wgrad_kernel<16, 1, 4, 1><<<grid, 256, 0, st>>>(*d, (int)mpb);
        else if (bnw == 32) wgrad_kernel<32, 1, 4, 1><<<grid, 256, 0, st>>>(*d, (int)mpb);
        else if (bnw == 64) wgrad_kernel<64, 2, 2, 1><<<grid, 256, 0, st>>>(*d, (int)mpb);
        else wgrad_kernel<128, 2, 2, 1><<<grid, 256, 0, st>>>(*d, (int)mpb);
    }
    det_finish(*dorig, dd_, splits, st);
    SEHIP_CHECK_LAUNCH("wgrad");
    return 0;
}

// ---- grouped weight gradients -----------------------------------------------------------------------------
// Device image: [SEHIP_WGRAD_GROUP_MAX] WgradGroupEntry, then the descriptors.
static size_t wgrad_group_desc_off() { return SEHIP_WGRAD_GROUP_MAX * sizeof(WgradGroupEntry); }
extern "C" long sehip_wgrad_group_bytes(int n) { return (long)(wgrad_group_desc_off() + (size_t)n * sizeof(sehip_gemm_desc)); }

// Copies the n descriptors and their block table into dev_buf (synchronous: call at bind time, not per step).
// Only products that sehip_wgrad would give to wgrad_kernel<128, 2, 2, 1> can be grouped (Npad a multiple of 128, no
// convolution description): anything else is an error, the caller then launches them one by one.
extern "C" int sehip_wgrad_group_prepare(const sehip_gemm_desc* descs, int n, void* dev_buf, int* total_blocks) {
    SEHIP_REQUIRE(descs && dev_buf && total_blocks, "wgrad_group_prepare: null argument");
    SEHIP_REQUIRE(n >= 1 && n <= SEHIP_WGRAD_GROUP_MAX, "wgrad_group_prepare: %d products (1..%d)", n, SEHIP_WGRAD_GROUP_MAX);
    WgradGroupEntry tab[SEHIP_WGRAD_GROUP_MAX] = {};
    int first = 0;
    for (int g = 0; g < n; ++g) {
        const sehip_gemm_desc* d = descs + g;
        if (int e = check_desc("wgrad_group", d)) return e;
        SEHIP_REQUIRE(d->dW != nullptr, "wgrad_group: product %d has no dW", g);
        SEHIP_REQUIRE(d->cv_nf <= 0 && (d->Npad & 127) == 0, "wgrad_group: product %d is not a plain 128-column-tile product", g);
        const int ntiles = d->Npad / 128, ktiles = cdiv(d->K, 64);
        long want = 2048 / ((long)ntiles * ktiles);
        if (want < 1) want = 1;
        long mpb = ((d->M + want - 1) / want + 63) / 64 * 64;
        if (mpb < 256) mpb = 256;
        const int splits = cdiv(d->M, mpb);
        tab[g] = {first, ntiles, ktiles, (int)mpb};
        first += ntiles * ktiles * splits;
    }
    *total_blocks = first;
    hipError_t e = hipMemcpy(dev_buf, tab, sizeof(tab), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy((char*)dev_buf + wgrad_group_desc_off(), descs, (size_t)n * sizeof(sehip_gemm_desc), hipMemcpyHostToDevice);
    if (e != hipSuccess) return sehip_set_error(-2, "wgrad_group_prepare: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int sehip_wgrad_group(const void* dev_buf, int n, int total_blocks, void* stream) {
    SEHIP_REQUIRE(!sehip_deterministic(), "wgrad_group: not part of the deterministic schedule (launch the products one by one: sehip_wgrad)");
    SEHIP_REQUIRE(dev_buf && n >= 1 && n <= SEHIP_WGRAD_GROUP_MAX && total_blocks > 0, "wgrad_group: bad arguments");
    sehip_note_kernel("wgrad_group_kernel<128, 2, 2>");
    wgrad_group_kernel<128, 2, 2><<<total_blocks, 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const sehip_gemm_desc*>((const char*)dev_buf + wgrad_group_desc_off()),
        reinterpret_cast<const WgradGroupEntry*>(dev_buf), n);
    SEHIP_CHECK_LAUNCH("wgrad_group");
    return 0;
}

template <typename K>
static void set_lds(K kernel, int bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

extern "C" int sehip_init(void) {
#define INIT_CONV(NF_)                                           \
    set_lds(&conv_gemm_kernel<128, 2, 2, NF_>, 96 * 1024);      \
    set_lds(&conv_gemm_kernel<64, 1, 4, NF_>, 96 * 1024);       \
    set_lds(&conv_wgrad_kernel<NF_>, 96 * 1024);
    INIT_CONV(2) INIT_CONV(3) INIT_CONV(5)
#undef INIT_CONV
#define INIT_CS2(BN_, MI_) set_lds(&conv_small2_kernel<BN_, 6, MI_, false>, 120 * 1024); set_lds(&conv_small2_kernel<BN_, 12, MI_, false>, 120 * 1024);
    INIT_CS2(16, 4) INIT_CS2(16, 2) INIT_CS2(32, 4) INIT_CS2(32, 2) INIT_CS2(64, 4) INIT_CS2(64, 2) INIT_CS2(128, 2)
#undef INIT_CS2
#define INIT_CS2P(BN_, MI_) set_lds(&conv_small2_kernel<BN_, 6, MI_, true>, 160 * 1024); set_lds(&conv_small2_kernel<BN_, 12, MI_, true>, 160 * 1024);
    INIT_CS2P(16, 4) INIT_CS2P(16, 2) INIT_CS2P(32, 4) INIT_CS2P(32, 2)
#undef INIT_CS2P
#define INIT_SW(BN_, KPW_) set_lds(&conv_small_wgrad_kernel<BN_, KPW_, 6>, 120 * 1024); set_lds(&conv_small_wgrad_kernel<BN_, KPW_, 12>, 120 * 1024);
    INIT_SW(16, 1) INIT_SW(16, 2) INIT_SW(16, 3) INIT_SW(16, 4) INIT_SW(16, 5) INIT_SW(16, 6) INIT_SW(16, 8) INIT_SW(16, 12)
    INIT_SW(32, 1) INIT_SW(32, 2) INIT_SW(32, 3) INIT_SW(32, 4) INIT_SW(32, 5) INIT_SW(32, 6) INIT_SW(32, 8) INIT_SW(32, 12)
    INIT_SW(64, 1) INIT_SW(64, 2) INIT_SW(64, 3) INIT_SW(64, 4) INIT_SW(64, 5)
#undef INIT_SW
    sehip_conv2_init();
    sehip_conv3_init();
    sehip_wgrad3_init();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sehip_set_error(-2, "init: %s", hipGetErrorString(e));
    return 0;
}

