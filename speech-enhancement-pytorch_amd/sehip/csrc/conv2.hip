// conv_gemm_v2_kernel: the LDS-patch implicit-GEMM of gemm.hip (conv_gemm_kernel) rebuilt around what bounded it.
//
// Round-1 kernel (128 x 128 tile, 4 waves, 2 workgroups per CU): per 64-deep K step every thread wrote 64 bytes of the weight
// tile from registers into LDS (ds_write_b128: ~79 B/clk per CU, as costly as the 16 fragment reads of the step), with TWO
// workgroup barriers around it, and tiles never crossed an utterance (T = 323 frames against 32-frame blocks: 8 % of the
// workgroups' rows were padding).  27 % of the dense bf16 MFMA peak.
//
// Here (one 512-thread workgroup per CU, 8 waves as 4 (m) x 2 (n), tile 256 rows x 128 output channels):
//   * the weight tile [128 n][64 k] goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, two instructions per wave and K step:
//     no VGPR round trip, no ds_write), through a ring of THREE 16 KB slots with a counted s_waitcnt vmcnt: tile s+2 is issued
//     while tile s is multiplied, one tile stays in flight across the barrier.  The 16-byte XOR swizzle of the tile (conflict
//     free b128 fragment reads) sits on the per-lane SOURCE address, the LDS image is lane-linear (what the DMA writes);
//   * ONE raw s_barrier per K step (it publishes tile s and retires the reads of tile s-1 whose slot tile s+2 overwrites);
//   * the weight tile is shared by 256 rows instead of 128: half the DMA bytes and half the barriers per FLOP;
//   * rows are tiled over a VIRTUAL flat frame index: every utterance owns TT + 2 frames, the last two are padding, so a tile
//     runs across utterance boundaries while "frame -1" / "frame TT" of an utterance still read zeros (the causal time padding
//     of ComplexConv2d, src/model/dccrn.py:359-360, and the frame the reference drops after the transposed convolution,
//     :193-196): 0.6 % padding rows instead of 8-16 %;
//   * the input patch ((TB + 1) frames x FR rows x 64 channels, pitch 144 B) is still staged through registers (its image is
//     padded, an LDS-DMA image cannot be), prefetched one channel chunk ahead; per-piece addresses and validity are computed
//     once per workgroup.
// Same operand / fragment / epilogue conventions as conv_gemm_kernel (weights are the MFMA A operand, a lane ends up with 4
// consecutive output channels of a row; dense 64-channel runs leave through a wave-private LDS image as 16-byte pieces).
#include <stdlib.h>
#include "common.h"
#include "../../../include/sehip.h"

#define C2_MAXP 7          // 16-byte patch pieces per thread and 32-channel chunk
#define C2_WSLOT 16384     // bytes per weight-tile slot: 128 rows x 128 B
typedef __attribute__((address_space(3))) void c2_lds_void;
typedef __attribute__((address_space(1))) const void c2_gvoid;
typedef unsigned c2_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) c2_u32x4* c2_gvec_ptr;

// 16 zero bytes: pieces outside a source (padding frames / rows) and the unused piece slots of a thread read this instead of
// skipping the load, so that EVERY thread issues exactly C2_MAXP loads per prefetch: the counted s_waitcnt vmcnt below relies on it
__device__ uint4 c2_zero16 = {0u, 0u, 0u, 0u};

__device__ __forceinline__ uint4 c2_add_bf16x8(uint4 a, uint4 r) {
    const unsigned av[4] = {a.x, a.y, a.z, a.w}, rv[4] = {r.x, r.y, r.z, r.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = pack_bf2(__uint_as_float(av[i] << 16) + __uint_as_float(rv[i] << 16),
                        __uint_as_float(av[i] & 0xffff0000u) + __uint_as_float(rv[i] & 0xffff0000u));
    return make_uint4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ size_t c2_dst_off(const sehip_dst& d, int b, int t, int j) {
    return (((size_t)b * d.T + t * (d.tmul > 1 ? d.tmul : 1) + d.toff) * d.F + (size_t)j * d.fmul + d.fadd) * d.C;
}

// WMW = waves along m (64 rows each): 2 -> 128-row tile, 256 threads, two workgroups per CU; 4 -> 256-row tile, 512 threads.
// PP = patch row pitch in elements (40 / 48 / 56: 32 channels + padding), a template parameter so that the tap offset of every
// fragment read is an instruction immediate.
//
// Instruction diet (PMC, round 2: 3.3 non-MFMA vector instructions per MFMA in the first version, with an MFMA occupying the
// SIMD's vector issue for half of its 16 cycles that is an issue-bound loop): every LDS fragment address is ONE register per
// (row group, frame offset) / (k half) + an immediate; the weight DMA takes a wave-uniform base (SGPR) + a constant per-lane
// offset; the patch pieces keep their LDS address and row offset in registers and only look the frame up per channel chunk.
template <int NF, int WMW, int PP>
__global__ __launch_bounds__(128 * WMW, 2) void conv_gemm_v2_kernel(const sehip_gemm_desc d, int TB, int JB, int FR, int B,
                                                                 int FS /* patch rows per frame (>= FR) */) {
    constexpr int BM = 64 * WMW, BN = 128, NTHR = 128 * WMW;
    constexpr int TN = 4, TM = 4;          // 16 x 16 MFMA tiles per wave: 64 output channels x 64 rows
    constexpr int H = NF;                  // K steps per 32-channel chunk: taps (2j, 2j + 1), j < H
    constexpr int DW = 1024 / NTHR;        // weight-tile DMA instructions per thread and K step
    static_assert(BM == 128 || BM == 256, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_raw* patch = reinterpret_cast<bf16_raw*>(smem + 3 * C2_WSLOT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                 // wave-uniform: LDS-DMA destinations stay scalar math
    const int wm = wave % WMW, wn = wave / WMW;
    const int ntn = d.Npad / BN;
    const int TV = d.TT + 2;                                                   // virtual frames per utterance (2 padding frames)
    // XCD-contiguous order, n-tile fastest (see gemm_kernel)
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int nt = logical % ntn, mt = logical / ntn;
    const int g0 = mt * TB, n0 = nt * BN;                                      // first virtual frame / output channel of the tile
    const int f0 = d.cv_fadd;                                                  // patch row 0 <-> source row j*fmul + cv_fadd at j = 0

    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    const int Ctot = C0 + C1;
    const int nch = Ctot >> 5;                                                 // 32-channel chunks
    const int tmin0 = min(d.cv_toff[0][0], d.cv_toff[0][1]), tmin1 = min(d.cv_toff[1][0], d.cv_toff[1][1]);
    const int NP = (TB + 1) * FR * 4;

    // ---- patch staging.  Piece u of this thread = 16 bytes at (patch frame p, row r, channel piece c4).  Kept per piece: its
    // LDS element address, its patch frame and its offset inside a source frame ((f0 + r) * C_s + 8 c4, per source; -1 when the
    // row lies outside the source).  What depends on the frame -- which utterance / source frame a patch frame is, or that it is
    // padding -- is the same for every piece of that frame: a table in LDS, filled once per workgroup (ftab[s][p] = element
    // offset of frame b*T_s + x, -1 if invalid).
    const bf16_raw* s0p = reinterpret_cast<const bf16_raw*>(d.src[0].ptr);
    const bf16_raw* s1p = reinterpret_cast<const bf16_raw*>(d.src[1].ptr);
    const bf16_raw* zero_page = reinterpret_cast<const bf16_raw*>(&c2_zero16);
    int* ftab = reinterpret_cast<int*>(smem + 3 * C2_WSLOT + (size_t)(TB + 1) * FS * PP * 2);   // [2][TB + 1], behind the patch
    const int dump = (TB + 1) * FS * PP + 4 * (TB + 1);                        // 16 bytes behind the table for the unused piece slots
    if (tid < 2 * (TB + 1)) {
        const int s = tid / (TB + 1), p = tid - s * (TB + 1);
        const sehip_src& S = s ? d.src[1] : d.src[0];
        const int sv = g0 + p + (s ? tmin1 : tmin0);
        int v = -1;
        if (sv >= 0 && (s == 0 || C1)) {
            const int b = sv / TV, x = sv - b * TV;
            if (b < B && x >= S.tlo && x < S.thi) v = (b * S.T + x) * S.F * S.C;
        }
        ftab[tid] = v;
    }
    int p_at[C2_MAXP], p_fr[C2_MAXP], p_r0[C2_MAXP], p_r1[C2_MAXP];
    {
        // idx = tid + NTHR*u -> (frame p, row r, channel piece c4): ONE division (u = 0), then steps of NTHR with a carry
        // (a generic 32-bit division is ~17 vector instructions; seven of them were a fifth of the workgroup's prologue)
        const unsigned fr4 = (unsigned)FR * 4u;
        const unsigned dp = (unsigned)NTHR / fr4, dr = (unsigned)NTHR - dp * fr4;
        unsigned p = (unsigned)tid / fr4, rem = (unsigned)tid - p * fr4;
#pragma unroll
        for (int u = 0; u < C2_MAXP; ++u) {
            const int idx = tid + NTHR * u;
            const int r = (int)(rem >> 2), c4 = (int)(rem & 3), f = f0 + r;
            const bool have = idx < NP;
            p_at[u] = have ? ((int)p * FS + r) * PP + c4 * 8 : dump;
            p_fr[u] = have ? (int)p : 0;
            p_r0[u] = (have && (unsigned)f < (unsigned)d.src[0].F) ? f * C0 + c4 * 8 : -1;
            p_r1[u] = (have && C1 && (unsigned)f < (unsigned)d.src[1].F) ? f * C1 + c4 * 8 : -1;
            rem += dr; p += dp;
            if (rem >= fr4) { rem -= fr4; ++p; }
        }
    }
    __syncthreads();
    uint4 pr[C2_MAXP];
    auto fetch_patch = [&](int ch) {
        const int second = ch * 32 >= C0 ? 1 : 0;
        const bf16_raw* base = (second ? s1p : s0p) + (ch * 32 - (second ? C0 : 0));
        const int* ft = ftab + second * (TB + 1);
        int fb[C2_MAXP];
#pragma unroll
        for (int u = 0; u < C2_MAXP; ++u) fb[u] = ft[p_fr[u]];                 // all table reads in flight together
#pragma unroll
        for (int u = 0; u < C2_MAXP; ++u) {
            const int ro = second ? p_r1[u] : p_r0[u];
            const bf16_raw* q = ((fb[u] | ro) >= 0) ? base + (fb[u] + ro) : zero_page;   // either negative: padding -> zeros
            const c2_u32x4 x4 = *(c2_gvec_ptr)(q);       // explicit global-address-space load (a select of generic pointers is a flat load)
            pr[u] = make_uint4(x4[0], x4[1], x4[2], x4[3]);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int u = 0; u < C2_MAXP; ++u) *reinterpret_cast<uint4*>(&patch[p_at[u]]) = pr[u];   // unused slots hit the dump
    };

    // ---- weight-tile DMA: piece q = tid + NTHR*i -> row r = q >> 3; LDS chunk c' = q & 7 holds tile chunk c = c' ^ (r & 7);
    //      tile chunk c = tap (2j + (c >> 2)), channels 8*(c & 3) .. +7 of the 32-channel chunk.  Address = wave-uniform pointer
    //      (W + step offset, SGPR) + this lane's constant element offset.
    const bf16_raw* Wb = reinterpret_cast<const bf16_raw*>(d.W) + (size_t)n0 * d.K;
    unsigned woff[DW];
#pragma unroll
    for (int i = 0; i < DW; ++i) {
        const int q = tid + NTHR * i;
        const int r = q >> 3, c = (q & 7) ^ (r & 7);
        woff[i] = (unsigned)(r * d.K + (c >> 2) * Ctot + (c & 3) * 8);
    }
    auto issue_w = [&](int ch, int j, int slot) {        // K step (ch, j) reads W columns (2j + {0, 1}) * Ctot + ch*32 ..
        const bf16_raw* wb = Wb + (2 * j * Ctot + ch * 32);
        unsigned char* dst = smem + slot * C2_WSLOT + wave * 1024;
#pragma unroll
        for (int i = 0; i < DW; ++i)
            __builtin_amdgcn_global_load_lds((c2_gvoid*)(wb + woff[i]), (c2_lds_void*)(dst + i * (NTHR * 16)), 16, 0, 0);
    };

    // ---- fragment addresses.  Weights: byte offset of this lane's row / k piece inside a slot for the two k halves of a step;
    // the 16-row tiles of a wave are immediates (ni * 16 rows * 128 B).  Activations: element offset of the lane's row in the
    // patch for each of its TM row groups; frame offset of the tap pair and the tap itself are added per chunk / as immediates.
    int wrd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int r = wn * 64 + (lane & 15);
        wrd[ks] = (r * 8 + ((ks * 4 + (lane >> 4)) ^ (r & 7))) * 16;
    }
    int abase[TM];
    const int lgJ = 31 - __clz(JB);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int r = wm * 64 + mi * 16 + (lane & 15);
        const int tl = r >> lgJ, jl = r & (JB - 1);                            // JB is a power of two (it divides 128)
        abase[mi] = (tl * FS + jl * d.fmul) * PP + 8 * (lane >> 4);
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int bb = 0; bb < TM; ++bb) acc[a][bb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int S = nch * H;
    // prologue: patch of chunk 0 through registers, weight tiles 0 and 1 in flight
    fetch_patch(0);
    issue_w(0, 0, 0);
    if (S > 1) issue_w(H > 1 ? 0 : 1, H > 1 ? 1 : 0, 1);
    store_patch();
    if (nch > 1) fetch_patch(1);           // next chunk's patch rides behind the first K steps

    int slot = 0;                          // slot of tile s; tile s + 2 goes to (slot + 2) % 3
    for (int ch = 0; ch < nch; ++ch) {
        const bool second = ch * 32 >= C0;
        const int dt0 = (second ? d.cv_toff[1][0] - tmin1 : d.cv_toff[0][0] - tmin0);
        const int dt1 = (second ? d.cv_toff[1][1] - tmin1 : d.cv_toff[0][1] - tmin0);
        int afr[2][TM];                    // row-group address + frame offset of kt = 0 / 1
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) { afr[0][mi] = abase[mi] + dt0 * FS * PP; afr[1][mi] = abase[mi] + dt1 * FS * PP; }
        if (ch > 0) {
            // every wave has finished reading the previous chunk's patch once it has passed this barrier
            __builtin_amdgcn_s_barrier();
            store_patch();                                    // (the compiler waits for the prefetched registers here)
            if (ch + 1 < nch) fetch_patch(ch + 1);
        }
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int s = ch * H + j;
            // weight tile s has landed for this wave's own pieces; the barrier makes everybody's pieces (and the patch stores)
            // visible and retires the reads of tile s-1.  Outstanding behind tile s: tile s+1 (DW DMAs) and, in the first two
            // steps after a chunk boundary, the patch prefetch of the next chunk (C2_MAXP loads) issued behind it.
            if (s + 1 >= S) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (j <= 1 && ch + 1 < nch) {
                if (DW == 4) asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            } else {
                if (DW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < S) {                                   // tile s + 2 = (ch, j + 2) or the next chunk's (j + 2 - H)
                const int slot2 = slot == 0 ? 2 : slot - 1;    // (slot + 2) % 3
                if (j + 2 < H) issue_w(ch, j + 2, slot2); else issue_w(ch + 1, j + 2 - H, slot2);
            }
            const unsigned char* wslot = smem + slot * C2_WSLOT;
            // All 16 fragment reads of the step are issued first, then the 32 MFMAs (scheduling groups: left alone the compiler
            // interleaved pairs of reads with short MFMA runs and a full `s_waitcnt lgkmcnt(0)` each time -- six exposed LDS
            // latencies per K step).  The LDS returns in order, so the waits in front of the MFMAs are counted ones.
            bf16x8 wf[2][TN], af[2][TM];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int it = 2 * j + ks;
                const int kt = it / NF, tap = it - kt * NF;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    wf[ks][ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wslot + wrd[ks] + ni * (16 * 128)));
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    af[ks][mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&patch[afr[kt][mi] + tap * PP]));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][ni], af[ks][mi], acc[ni][mi], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TN + TM), 0);   // DS reads
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN * TM, 0);     // MFMAs
            slot = slot == 2 ? 0 : slot + 1;
        }
    }

    // ---- epilogue (as conv_gemm_kernel): dense 64-channel runs of a bf16 destination leave through a wave-private LDS image
    constexpr int WROWS = 64, WCOLS = 64, TP = WCOLS + 8;
    const int nw0 = n0 + wn * WCOLS;
    sehip_nchunk first = d.ntab[nw0 >> 2];
    bool dense;
    {
        const sehip_nchunk mine = d.ntab[(nw0 >> 2) + (lane & 15)];
        const bool ok = mine.nvalid == 4 && mine.dst == first.dst && mine.coff == first.coff + 4 * (lane & 15);
        dense = __all(ok) && !(first.dst ? d.dst[1].is_f32 : d.dst[0].is_f32) && ((first.coff & 7) == 0) &&
                (((first.dst ? d.dst[1].C : d.dst[0].C) & 7) == 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // every wave has finished reading the weight tiles and the patch: the LDS is free
    // statistics need every wave of the workgroup on the dense path (they meet at a barrier below): decided workgroup-wide
    // (flags in the dynamic LDS behind the waves' images: __syncthreads_and would add static LDS on top of the 160 KB request)
    bool with_stats = false;
    if (d.stats) {
        int* flag = reinterpret_cast<int*>(smem + (size_t)(NTHR / 64) * (64 * 72 * 2));
        if (lane == 0) flag[wave] = dense ? 1 : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        with_stats = true;
#pragma unroll
        for (int i = 0; i < NTHR / 64; ++i) with_stats = with_stats && flag[i] != 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // everybody has read the flags before the images are written
    }
    if (dense) {
        bf16_raw* tb_ = reinterpret_cast<bf16_raw*>(smem) + wave * (WROWS * TP);   // 9 KB per wave: inside the (now idle) weight ring + patch
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
        const sehip_dst& dd = first.dst ? d.dst[1] : d.dst[0];
        if (with_stats) {
            // Batch statistics of the ComplexBatchNorm that follows (descriptor field `stats`), from the bf16 values just parked
            // in the waves' LDS images: wave (wm, 0) holds the real halves and wave (wm, 1) the imaginary halves of the tile's
            // 64 complex channels for the same 64 rows.  Wave (wm, wn) takes channels 32 wn .. 32 wn + 31: lane = channel pair
            // (lane & 15) x row group (lane >> 4, 16 rows each); padding rows (virtual frames) are masked out.
            bool rv;
            {
                const int rr = wm * WROWS + lane;
                const int tl = rr >> lgJ;
                int b = g0 / TV, t = g0 - b * TV + tl;
                for (; t >= TV; t -= TV) ++b;
                rv = b < B && t < d.TT;
            }
            const unsigned long long rmask = __ballot(rv);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // both images of this row block are complete
            const bf16_raw* imr = reinterpret_cast<const bf16_raw*>(smem) + (wm + WMW * 0) * (WROWS * TP);
            const bf16_raw* imi = reinterpret_cast<const bf16_raw*>(smem) + (wm + WMW * 1) * (WROWS * TP);
            const int cp = 32 * wn + 2 * (lane & 15), rg = lane >> 4;
            float sr[2] = {0.f, 0.f}, si[2] = {0.f, 0.f}, srr[2] = {0.f, 0.f}, sri[2] = {0.f, 0.f}, sii[2] = {0.f, 0.f};
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int r = 16 * rg + it;
                const unsigned ur = *reinterpret_cast<const unsigned*>(&imr[r * TP + cp]);
                const unsigned ui = *reinterpret_cast<const unsigned*>(&imi[r * TP + cp]);
                const float ok = (rmask >> r) & 1ull ? 1.f : 0.f;
                const float yr[2] = {__uint_as_float(ur << 16) * ok, __uint_as_float(ur & 0xffff0000u) * ok};
                const float yi[2] = {__uint_as_float(ui << 16) * ok, __uint_as_float(ui & 0xffff0000u) * ok};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    sr[e] += yr[e]; si[e] += yi[e];
                    srr[e] += yr[e] * yr[e]; sri[e] += yr[e] * yi[e]; sii[e] += yi[e] * yi[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    sr[e] += __shfl_xor(sr[e], o, 64); si[e] += __shfl_xor(si[e], o, 64);
                    srr[e] += __shfl_xor(srr[e], o, 64); sri[e] += __shfl_xor(sri[e], o, 64); sii[e] += __shfl_xor(sii[e], o, 64);
                }
            }
            if (lane < 16) {
                const int Cr = d.stats_cr;
                float* sp = d.stats + (size_t)(blockIdx.x & 7) * 5 * Cr + (n0 >> 1) + cp;   // tile n0 <-> complex channels n0/2 ..
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    atomicAdd(sp + e, sr[e]); atomicAdd(sp + Cr + e, si[e]);
                    atomicAdd(sp + 2 * Cr + e, srr[e]); atomicAdd(sp + 3 * Cr + e, sri[e]); atomicAdd(sp + 4 * Cr + e, sii[e]);
                }
            }
        }
        bf16_raw* dptr = reinterpret_cast<bf16_raw*>(dd.ptr) + first.coff;
        const bf16_raw* rptr = (d.res && first.dst == 0) ? reinterpret_cast<const bf16_raw*>(d.res) + first.coff : nullptr;
        // destination offsets in 32 bits (the dispatcher checked the sizes) from per-workgroup constants; the tile's first
        // virtual frame is split into (utterance, frame) once, the rows step from there with a wrap
        const int b0 = g0 / TV, t0 = g0 - b0 * TV;
        const int tsz = dd.F * dd.C * (dd.tmul > 1 ? dd.tmul : 1), bsz = dd.T * dd.F * dd.C, jsz = dd.fmul * dd.C;
        const int base0 = (dd.toff * dd.F + dd.fadd) * dd.C + (lane & 7) * 8;
#pragma unroll
        for (int itr = 0; itr < WROWS / 8; ++itr) {
            const int row = itr * 8 + (lane >> 3);
            const int rr = wm * WROWS + row;
            const int tl = rr >> lgJ, jl = rr & (JB - 1);
            int b = b0, t = t0 + tl;
            for (; t >= TV; t -= TV) ++b;                     // at most once unless the utterances are shorter than a tile
            uint4 v = *reinterpret_cast<const uint4*>(&tb_[row * TP + (lane & 7) * 8]);
            if (b < B && t < d.TT) {
                const int off = b * bsz + t * tsz + jl * jsz + base0;
                if (rptr) v = c2_add_bf16x8(v, *reinterpret_cast<const uint4*>(rptr + off));
                *reinterpret_cast<uint4*>(dptr + off) = v;
            }
        }
        return;
    }
    // direct scatter, 4 consecutive channels per lane (fp32 destinations, narrow or split column groups)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int rr = wm * 64 + mi * 16 + (lane & 15);
        const int tl = rr / JB, jl = rr - tl * JB;
        const int gv = g0 + tl;
        const int b = gv / TV, t = gv - b * TV;
        if (b >= B || t >= d.TT) continue;
        const size_t ro0 = c2_dst_off(d.dst[0], b, t, jl);
        const size_t ro1 = d.dst[1].ptr ? c2_dst_off(d.dst[1], b, t, jl) : 0;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = nw0 + ni * 16 + 4 * (lane >> 4);
            const sehip_nchunk nc = d.ntab[n >> 2];
            if (nc.nvalid <= 0) continue;
            f32x4 v = acc[ni][mi];
            if (d.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(d.bias + n);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            const size_t off = (nc.dst ? ro1 : ro0) + nc.coff;
            void* dptr = nc.dst ? d.dst[1].ptr : d.dst[0].ptr;
            const int is_f32 = nc.dst ? d.dst[1].is_f32 : d.dst[0].is_f32;
            if (d.res && nc.dst == 0 && nc.nvalid == 4) {
                const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_raw*>(d.res) + off);
                v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
            }
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

template <int NF, int WMW, int PP>
static void c2_launch_pp(const sehip_gemm_desc& d, int TB, int JB, int FR, int B, int FS, int grid, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_v2_kernel<NF, WMW, PP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    conv_gemm_v2_kernel<NF, WMW, PP><<<grid, 128 * WMW, lds, st>>>(d, TB, JB, FR, B, FS);
}
template <int NF, int WMW>
static void c2_launch(const sehip_gemm_desc& d, int TB, int JB, int FR, int B, int PP, int FS, int grid, size_t lds, hipStream_t st) {
    sehip_note_kernel("conv_gemm_v2_kernel<%d, %d, %d>", NF, WMW, PP == 40 || PP == 48 ? PP : 56);   // as rocprofv3 prints the symbol
    if (PP == 40) c2_launch_pp<NF, WMW, 40>(d, TB, JB, FR, B, FS, grid, lds, st);
    else if (PP == 48) c2_launch_pp<NF, WMW, 48>(d, TB, JB, FR, B, FS, grid, lds, st);
    else c2_launch_pp<NF, WMW, 56>(d, TB, JB, FR, B, FS, grid, lds, st);
}

// LDS cycles of one A-fragment ds_read_b128 wave instruction (4 = conflict free) for a patch with row pitch `pitch` bytes and
// `fs` rows per frame, averaged over the taps / frame offsets / wave rows of the kernel: 16-byte reads are served in four
// 16-lane groups, bank = (address / 4) % 64 (MI355X_MICROARCH.md, LDS).  The rows a group touches (8 rows at one 8-channel piece
// and 8 others at the next) depend on rows per frame, row stride and taps, so no single pitch is conflict free for every
// layer: the dispatcher takes a census over a few (pitch, frame stride) candidates once per geometry and keeps the best.
static double c2_read_cycles(int JB, int fmul, int NF, int pitch, int fs) {
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    long tot = 0, n = 0;
    for (int row0 = 0; row0 < 128; row0 += 16)
        for (int tap = 0; tap < NF; ++tap)
            for (int dt = 0; dt < 2; ++dt) {
                for (int g = 0; g < 4; ++g) {
                    int cnt[64] = {0}, worst = 0;
                    for (int q = 0; q < 16; ++q) {
                        const int l = groups[g][q];
                        const int r = row0 + (l & 15), tl = r / JB, jl = r - tl * JB;
                        const int addr = ((tl + dt) * fs + jl * fmul + tap) * pitch + 16 * (l >> 4);
                        for (int k = 0; k < 4; ++k) {
                            int& c = cnt[((addr >> 2) + k) & 63];
                            if (++c > worst) worst = c;
                        }
                    }
                    tot += worst;
                }
                ++n;
            }
    return (double)tot / n;
}

static_assert(C2_MAXP == 7, "the counted waits (vmcnt(DW + C2_MAXP)) are written for 7 prefetch loads per thread");

// returns 1 if the kernel was launched, 0 if the descriptor does not qualify (the caller falls back to conv_gemm_kernel)
int sehip_try_conv_gemm_v2(const sehip_gemm_desc& d, hipStream_t st) {
    static const bool disabled = getenv("SEHIP_NO_CONV_V2") != nullptr || getenv("SEHIP_NO_PATCH") != nullptr;
    static const int bm_force = getenv("SEHIP_CONV_V2_BM") ? atoi(getenv("SEHIP_CONV_V2_BM")) : 0;
    static const bool census = getenv("SEHIP_CONV_V2_CENSUS") != nullptr;
    if (disabled || d.cv_nf <= 0 || d.tmul > 1) return 0;
    if (d.stats && (d.dst[1].ptr || d.dst[0].is_f32 || (d.dst[0].C & 7) || d.stats_cr * 2 != d.Npad)) return 0;   // see sehip.h
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    if ((C0 & 31) || (C1 & 31) || (d.Npad & 127) || d.J > 64 || (128 % d.J)) return 0;   // (32-channel patch chunks, 128-column tiles)
    if (d.K != 2 * d.cv_nf * (C0 + C1)) return 0;
    if ((d.dst[0].tmul > 1) || (d.dst[1].ptr && d.dst[1].tmul > 1)) return 0;
    for (int s = 0; s < 2; ++s) {
        if (!d.src[s].ptr) continue;
        // frame offsets reach at most one frame outside [0, TT): what the two padding frames per utterance absorb
        for (int kt = 0; kt < 2; ++kt)
            if (d.cv_toff[s][kt] < -1 || d.cv_toff[s][kt] > 1) return 0;
        if (d.src[s].thi > d.TT + 1) return 0;
        if ((long)d.M / d.J / d.TT * d.src[s].T * d.src[s].F * d.src[s].C >= (1L << 31)) return 0;   // 32-bit piece offsets
    }
    const int B = d.M / (d.TT * d.J);
    for (int s = 0; s < 2; ++s)                          // 32-bit destination offsets in the epilogue
        if (d.dst[s].ptr && (long)B * d.dst[s].T * d.dst[s].F * d.dst[s].C >= (1L << 31)) return 0;
    const long vframes = (long)B * (d.TT + 2);
    int BM = bm_force == 256 ? 256 : 128;
    const int JB = d.J;
    const int FR = (JB - 1) * d.fmul + d.cv_nf;
    for (;;) {
        const int TB = BM / JB;
        const int nthr = BM * 2;
        // bank-conflict census: row pitch 80 / 96 / 112 bytes x up to 3 padding rows per frame
        // (opt-in: with the register budget pinned to two waves per SIMD the plain 80-byte pitch measures the same or better --
        //  5 taps 104 vs 108 us, 3 taps 96 vs 111, 2 taps 83 vs 81; an earlier "145 -> 110 us" of the census was an occupancy
        //  artefact: without __launch_bounds__(.., 2) that build spilled into AGPRs and ran ONE wave per SIMD)
        int PP = 40, FS = FR;
        if (census) {
            double best = 1e9;
            for (int pitch = 80; pitch <= 112; pitch += 16)
                for (int fs = FR; fs <= FR + 3; ++fs) {
                    const size_t bytes = (size_t)(TB + 1) * fs * pitch + 3 * C2_WSLOT + 2 * (TB + 1) * sizeof(int) + 16;
                    if (bytes > (size_t)(BM == 128 ? 80 : 160) * 1024) continue;
                    const double c = c2_read_cycles(JB, d.fmul, d.cv_nf, pitch, fs) + 1e-3 * (pitch - 80) + 1e-4 * (fs - FR);
                    if (c < best) { best = c; PP = pitch / 2; FS = fs; }
                }
        }
        const size_t patch_bytes = (size_t)(TB + 1) * FS * PP * 2;
        size_t lds = 3 * C2_WSLOT + patch_bytes + 2 * (TB + 1) * sizeof(int) + 16;
        const size_t epi = (size_t)(nthr / 64) * 64 * 72 * 2 + 64;    // the waves' output staging images reuse the same LDS (+ flags)
        if (lds < epi) lds = epi;
        const bool fits = (TB + 1) * FR * 4 <= C2_MAXP * nthr && TB + 1 <= 255 && lds <= (BM == 128 ? 80 : 160) * 1024;
        if (!fits) {
            if (BM == 256) return 0;
            BM = 256;                                                 // one workgroup per CU with all its LDS
            continue;
        }
        const int mtiles = (int)((vframes + TB - 1) / TB);
        const int grid = mtiles * (d.Npad / 128);
#define C2_CASE(NF_)                                                                          \
        case NF_:                                                                             \
            if (BM == 128) c2_launch<NF_, 2>(d, TB, JB, FR, B, PP, FS, grid, lds, st);        \
            else c2_launch<NF_, 4>(d, TB, JB, FR, B, PP, FS, grid, lds, st);                  \
            return 1;
        switch (d.cv_nf) {
            C2_CASE(2) C2_CASE(3) C2_CASE(5)
            default: return 0;
        }
#undef C2_CASE
    }
}

template <int NF, int WMW, int PP>
static void c2_init_one() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_v2_kernel<NF, WMW, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int NF>
static void c2_init_nf() {
    c2_init_one<NF, 2, 40>(); c2_init_one<NF, 2, 48>(); c2_init_one<NF, 2, 56>();
    c2_init_one<NF, 4, 40>(); c2_init_one<NF, 4, 48>(); c2_init_one<NF, 4, 56>();
}
void sehip_conv2_init(void) {
    c2_init_nf<2>(); c2_init_nf<3>(); c2_init_nf<5>();
}
