// conv_gemm_v3_kernel: the implicit-GEMM convolution of the deep DCCRN layers (ComplexConv2d / ComplexConvTranspose2d forward
// and input gradients, src/model/dccrn.py:316-450, C >= 16 channels per source, >= 128 output columns), third design.
//
// What bounded conv_gemm_v2 (round 2: 0.26 of the dense bf16 MFMA peak): 64 x 64 wave tiles (16 fragment reads per 32 MFMAs),
// a 32-channel input patch staged through registers with ds_write_b128 and a ~3 200-cycle stall at every chunk boundary,
// 16 KB of weights per 2.1 MFLOP.  Here, per 256-thread workgroup (two per CU), tile = 256 rows x 128 output channels:
//   * 4 waves as 2 (m) x 2 (n), wave tile 128 rows x 64 channels = 8 x 4 MFMA tiles (16x16x32 bf16; 192-row and 64-column
//     variants for under-filled grids / 64-output layers): 12 fragment reads per 32
//     MFMAs, one barrier per 32 MFMAs, 8 KB of weights + ~4 KB of patch per 2.1 MFLOP;
//   * K runs over 16-channel chunks; an MFMA's k = 32 is (tap pair) x (16 channels): k group g = lane >> 4 is tap 2j + (g >> 1),
//     8-channel piece g & 1, for the weight AND the activation operand -- every LDS row is 32 bytes, and 16 rows x 32 B is
//     exactly what one 16-lane group of ds_read_b128 serves without bank conflicts;
//   * BOTH operands reach LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write).  The DMA writes lane-linear
//     images but reads per-lane sources, so the patch image is a free permutation of 16-byte pieces: frames x [parity planes of
//     a stride-2 layer] x rows, frame stride S and the frame order inside a 16-column MFMA tile chosen per geometry so that every
//     fragment read is conflict free (tools/c3_census.py); padding frames / rows come from a zero page;
//   * the patch is double buffered by 16-channel chunk: chunk c + 1 lands while chunk c is multiplied (no chunk-boundary
//     stall); weights: ring of FOUR 8 KB tiles, tile s + 3 issued at step s; counted s_waitcnt vmcnt, ONE raw s_barrier per
//     K step.  The DMA stream is strictly periodic (it runs past the end into unused slots) so that the counts are constants.
// Operand / epilogue conventions as conv_gemm_v2 (weights are the MFMA A operand, a lane ends up with 4 consecutive output
// channels of a row, dense 64-channel runs leave through a wave-private LDS image, optional fused ComplexBatchNorm sums).
#include <stdlib.h>
#include "common.h"
#include "../../../include/sehip.h"

typedef __attribute__((address_space(3))) void c3_lds_void;
typedef __attribute__((address_space(1))) const void c3_gvoid;
typedef __attribute__((address_space(3))) s16x4 c3_lds_s16x4;

// Both operands are fetched with buffer_load_dwordx4 ... lds through raw buffer descriptors: the per-lane byte offset is a
// register computed once per workgroup, what changes per K step / channel chunk is the scalar offset (no vector arithmetic in
// the loop), and padding pieces carry an offset beyond num_records -- the hardware range check returns zeros without a read.
#define C3_OOB 0x7ffffff0u

#define C3_NSLOT 4
#ifdef C3_STAMPS      // tools/build_variant.py ... -DC3_STAMPS: per-wave cycle stamps (tools/c3_stamps.py reads them); never in the product build
__device__ unsigned* c3_stamp_ptr;
extern "C" int sehip_c3_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(c3_stamp_ptr), &p, sizeof(p)); }
#define C3_T() ((unsigned)__builtin_amdgcn_s_memtime())
#endif
#ifndef C3_PATCH_AUX
#define C3_PATCH_AUX 0        // cache policy of the patch DMAs (2 = nt: measured)
#endif

// geometry of the patch image (tools/c3_census.py: conflict-free frame stride per (row stride, taps, rows per frame))
template <int NF, int FM, int J, int TM, int NWN = 2>
struct C3Geo {
    static constexpr int FR = (J - 1) * FM + NF;                       // patch rows per frame
    static constexpr int S = FM == 2 ? (J == 4 ? 11 : J == 8 ? 20 : J == 16 ? 35 : 67)
                             : NF == 3 ? (J == 4 ? 7 : J == 8 ? 12 : J == 16 ? 18 : 34)
                                       : (J == 4 ? 5 : J == 8 ? 12 : J == 16 ? 17 : 33);
    static constexpr int P1 = FM == 2 ? (FR + 1) / 2 : 0;              // first physical row of the odd-row plane
    static constexpr int TB = 32 * TM / J;                             // frames per tile (2 M-waves x TM x 16 rows)
    static constexpr int NPIECE = (TB + 1) * S * 2;                    // 16-byte pieces per buffer
    static constexpr int NTH = 128 * NWN;                              // threads: 2 (m) x NWN (n) waves
    static constexpr int MAXP = (NPIECE + NTH - 1) / NTH;              // DMA instructions per thread and chunk
    static constexpr int PBYTES = ((TB + 1) * S * 32 + 1023) / 1024 * 1024;
    static constexpr int RING = C3_NSLOT * 1024 * 2 * NWN * 2;         // four weight tiles of 32 TN NWN rows x 64 B (TN = 4)
    static constexpr int TABLES = (2 * (TB + 1) + TB + 8) * 4;         // source frame table [2][TB + 1], destination frame table [TB], 8 wave flags
    static constexpr int LDS_MAIN = RING + 2 * PBYTES + 1024 + TABLES;
    static_assert(S >= FR && (FM == 1 || P1 + FR / 2 <= S), "frame stride");
    static_assert(J == 4 || J == 8 || J == 16 || J == 32, "rows per frame");
    static_assert(TM == 8 || TM == 6 || TM == 4, "MFMA row tiles per wave (even: the frame order of J = 4 / 8 pairs them)");
};

// row rw (= 16 mi + column c) of M-wave wm -> frame inside the tile and row inside the frame
template <int J, int TM>
__device__ __forceinline__ void c3_row(int wm, int rw, int& tl, int& jl) {
    const int mi = rw >> 4, c = rw & 15;
    constexpr int FW = 16 * TM / J;                                    // frames per M-wave
    if (J >= 16) {
        constexpr int per = J >= 16 ? J / 16 : 1;
        tl = wm * FW + mi / per;
        jl = (mi % per) * 16 + c;
    } else if (J == 8) {
        tl = wm * FW + (mi >> 1) * 4 + (mi & 1) + 2 * (c >> 3);
        jl = c & 7;
    } else {
        const int q = c >> 2;
        tl = wm * FW + (mi >> 1) * 8 + (mi & 1) * 2 + ((q & 2) ? 4 : 0) + ((q ^ (q >> 1)) & 1);
        jl = c & 3;
    }
}

// scheduling groups: ND DMA instructions, each behind G MFMAs, then the remaining MFMAs (NM in all)
template <int ND, int NM, int G = 3>
__device__ __forceinline__ void c3_interleave() {
    if constexpr (ND == 0 || NM < G) {
        if constexpr (ND > 0) __builtin_amdgcn_sched_group_barrier(0x020, ND, 0);
        if constexpr (NM > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    } else {
        __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        c3_interleave<ND - 1, NM - G, G>();
    }
}
template <int N>
__device__ __forceinline__ void c3_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// DMA instructions issued behind the youngest item step (.., j) needs (its weight tile, and the patch buffer when j == 0):
// per step [weight tile s + 3: DW], then at j == 0 [next chunk's patch: MAXP]
template <int H, int MAXP, int DW>
constexpr int c3_count(int j) {
    if (j == 0 && H <= 3) return DW * (H - 1);
    int n = 2 * DW;
    for (int back = 1; back <= 3; ++back)
        if (((j - back) % H + H) % H == 0) n += MAXP;
    return n;
}

template <int H, int MAXP, int DW>
__device__ __forceinline__ void c3_wait_step(int j) {          // j is a constant after unrolling: the switch folds
    switch (j) {
        case 0: c3_wait_vm<c3_count<H, MAXP, DW>(0)>(); break;
        case 1: c3_wait_vm<c3_count<H, MAXP, DW>(1)>(); break;
        case 2: c3_wait_vm<c3_count<H, MAXP, DW>(2)>(); break;
        case 3: c3_wait_vm<c3_count<H, MAXP, DW>(3)>(); break;
        default: c3_wait_vm<c3_count<H, MAXP, DW>(4)>(); break;
    }
}

// ABL: ablation builds for tools/ (never launched by the product path unless SEHIP_C3_ABL is set): 1 = no DMA inside the loop,
// 2 = no MFMA, 4 = no fragment reads
// TN = 16-column MFMA tiles per wave: 4 -> 128 output channels per workgroup, 2 -> 64 (the layers with 64 outputs)
// NWN = waves along n: 2 -> 256 threads, two workgroups per CU; 4 -> 512 threads, 256 output channels, one workgroup per CU
// The kernel body: workgroup `bid` of `nwg` of the product d (a function, so that one launch can run the tiles of TWO products -- the
// output-row parities of a transposed convolution -- as conv_gemm_v3_pair_kernel does below).
template <int NF, int FM, int J, int TM, int TN, int NWN = 2, int ABL = 0>
// `first_logical`, `halves`: the tail form (conv_gemm_v3_tail_kernel): this workgroup's position `bid` of `nwg` counts HALF tiles of
// the tiles from `first_logical` on -- tile height TM is then half of the launch's full tiles, two consecutive positions share a tile.
__device__ __forceinline__ void c3_body(const sehip_gemm_desc& d, const int B, const int order, const int bid, const int nwg,
                                        const int first_logical = 0, const bool halves = false) {
    using G = C3Geo<NF, FM, J, TM, NWN>;
    constexpr int NWV = 2 * NWN, NTH = 64 * NWV;     // waves, threads
    constexpr int WSLOT = 16 * TN * NWN * 64, RING = G::RING;
    static_assert(NWN == 2 || TN == 4, "the 8-wave build is for 256-column tiles");
    constexpr int TB = G::TB, S = G::S, P1 = G::P1, FR = G::FR, H = NF, MAXP = G::MAXP, NPIECE = G::NPIECE, PBYTES = G::PBYTES;
    constexpr int BN = 16 * TN * NWN;                 // output channels per workgroup (NWN N-waves x TN x 16)
    constexpr int DW = TN / 2;                        // weight-tile DMA instructions per wave and K step: BN rows x 2 taps x 32 B
    constexpr int WPL = BN * 32;                      // bytes of one tap plane of a weight tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* pbuf = smem + RING;
    unsigned char* dump = pbuf + 2 * PBYTES;
    // The frame tables live above BOTH the main-loop image and the epilogue's staging image (which reuses the memory from 0): the
    // epilogue reads the destination table, so nothing in it divides or touches the descriptor again.
    constexpr int EPI_BYTES = NWV * (16 * TM * (16 * TN + 8) * 2) + 64;
    constexpr int TBL = RING + 2 * PBYTES + 1024 > EPI_BYTES ? RING + 2 * PBYTES + 1024 : EPI_BYTES;
    int* ftab = reinterpret_cast<int*>(smem + TBL);                            // [2][TB + 1] source frame -> element offset, -1 = padding
    int* otab = ftab + 2 * (TB + 1);                                           // [TB] tile frame -> (utterance << 16) | frame, -1 = outside

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
#ifdef C3_STAMPS
    const unsigned st_rt0 = (unsigned)__builtin_amdgcn_s_memrealtime(), st_t0 = C3_T();
    unsigned st_issue = 0, st_dma = 0, st_bar = 0, st_tail = 0, st_n = 0, st_prev = 0, st_imax = 0, st_bmax = 0;
#endif
    // The whole descriptor in ONE batch of scalar loads: left to itself the compiler fetches the kernel arguments where they are first
    // used, i.e. in four to five dependent groups, each a round trip to memory (~600 cycles: the kernarg segment is in no cache when a
    // launch starts) before the first DMA can be issued, and again in the epilogue.
#define C3_KEEP(x) asm volatile("" ::"s"(x))
    C3_KEEP(d.Npad); C3_KEEP(d.TT); C3_KEEP(d.K); C3_KEEP(d.cv_fadd); C3_KEEP(d.w_tiled); C3_KEEP(B); C3_KEEP(order);
    C3_KEEP(d.cv_toff[0][0]); C3_KEEP(d.cv_toff[0][1]); C3_KEEP(d.cv_toff[1][0]); C3_KEEP(d.cv_toff[1][1]);
    C3_KEEP(d.src[0].ptr); C3_KEEP(d.src[0].T); C3_KEEP(d.src[0].tlo); C3_KEEP(d.src[0].thi); C3_KEEP(d.src[0].F); C3_KEEP(d.src[0].C);
    C3_KEEP(d.src[1].ptr); C3_KEEP(d.src[1].T); C3_KEEP(d.src[1].tlo); C3_KEEP(d.src[1].thi); C3_KEEP(d.src[1].F); C3_KEEP(d.src[1].C);
    C3_KEEP(d.ntab); C3_KEEP(d.W); C3_KEEP(d.stats); C3_KEEP(d.bias); C3_KEEP(d.res); C3_KEEP(d.stats_cr);
    C3_KEEP(d.dst[0].ptr); C3_KEEP(d.dst[0].T); C3_KEEP(d.dst[0].F); C3_KEEP(d.dst[0].C); C3_KEEP(d.dst[0].toff); C3_KEEP(d.dst[0].fmul);
    C3_KEEP(d.dst[0].fadd); C3_KEEP(d.dst[0].is_f32);
    C3_KEEP(d.dst[1].ptr); C3_KEEP(d.dst[1].T); C3_KEEP(d.dst[1].F); C3_KEEP(d.dst[1].C); C3_KEEP(d.dst[1].toff); C3_KEEP(d.dst[1].fmul);
    C3_KEEP(d.dst[1].fadd); C3_KEEP(d.dst[1].is_f32);
#undef C3_KEEP
    const int ntn = d.Npad / BN;
    const int TV = d.TT + 2;
    const int xcd = bid & 7, within = bid >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    // tile order.  0 (default): n-tile fastest.  1: the n-tiles of one m-tile 32 logical places apart -- under round-robin dispatch
    // (blocks b, b + 8, ... on one XCD, its 32 CUs filled in order) the two workgroups of a CU then stream the SAME patch
    int nt = logical % ntn, mt = logical / ntn;
    if (order == 1 && ntn > 1) {
        const int span = 32 * ntn, grp = logical / span, r = logical - grp * span;
        const int mtiles = nwg / ntn;
        const int m_ = grp * 32 + (r & 31);
        if (grp * span + span <= nwg && m_ < mtiles) { nt = r >> 5; mt = m_; }     // (the ragged tail keeps the default order)
    }
    if (halves) {                 // position -> (tile, half): the tile grid is the FULL tiles' (2 TB frames each)
        const int L = first_logical + (logical >> 1);
        nt = L % ntn; mt = 2 * (L / ntn) + (logical & 1);
    }
    const int g0 = mt * TB, n0 = nt * BN;
    const int f0 = d.cv_fadd;

    // (the 5-tap products -- encoder forward, decoder input gradient -- have ONE source: the launcher checks; no second offset set)
    constexpr bool TWO = NF != 5;
    const int C0 = d.src[0].C, C1 = (TWO && d.src[1].ptr) ? d.src[1].C : 0;
    const int Ctot = C0 + C1;
    const int nch = Ctot >> 4;
    const int tmin0 = min(d.cv_toff[0][0], d.cv_toff[0][1]), tmin1 = min(d.cv_toff[1][0], d.cv_toff[1][1]);
    const bf16_raw* s0p = reinterpret_cast<const bf16_raw*>(d.src[0].ptr);
    const bf16_raw* s1p = reinterpret_cast<const bf16_raw*>(d.src[1].ptr);
    // num_records = the tensor's own size ([B][T][F][C] bf16, below 2^30 bytes: c3_qualifies): an offset that leaves the tensor
    // reads zeros like the padding marker does, not a neighbour's bytes
    const unsigned rec0 = 2u * (unsigned)B * (unsigned)d.src[0].T * (unsigned)d.src[0].F * (unsigned)C0;
    const unsigned rec1 = C1 ? 2u * (unsigned)B * (unsigned)d.src[1].T * (unsigned)d.src[1].F * (unsigned)C1 : rec0;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(s0p), 0, rec0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(C1 ? s1p : s0p), 0, rec1, 0x00020000);

    // ---- this wave's 64 output columns: one destination? (ntab entries fetched now, used behind the barrier below: the epilogue
    // does not wait for anything but its own stores)
    constexpr int WCOLS_ = 16 * TN;
    const int nw0 = n0 + wn * WCOLS_;
    const sehip_nchunk nc_first = d.ntab[nw0 >> 2];
    const sehip_nchunk nc_mine = d.ntab[(nw0 >> 2) + (lane & 15) % (WCOLS_ / 4)];

    // ---- frame tables.  Every descriptor field is selected from SCALAR loads of both candidates: indexing the by-value descriptor
    // with a per-lane (or even wave-uniform runtime) index becomes per-lane global loads from the kernarg segment -- three dependent
    // round trips in front of the first DMA, more in the epilogue (round 5: 3 000 of the prologue's 9 000 cycles).
    if (tid < 2 * (TB + 1)) {
        const bool s = tid >= TB + 1;
        const int p = s ? tid - (TB + 1) : tid;
        const int sT = s ? d.src[1].T : d.src[0].T, sF = s ? d.src[1].F : d.src[0].F, sC = s ? d.src[1].C : d.src[0].C;
        const int slo = s ? d.src[1].tlo : d.src[0].tlo, shi = s ? d.src[1].thi : d.src[0].thi;
        const int sv = g0 + p + (s ? tmin1 : tmin0);
        int v = -1;
        if (sv >= 0 && (!s || C1)) {
            const int b = sv / TV, x = sv - b * TV;
            if (b < B && x >= slo && x < shi) v = (b * sT + x) * sF * sC;
        }
        ftab[tid] = v;
    } else if (tid < 2 * (TB + 1) + TB) {
        const int gv = g0 + (tid - 2 * (TB + 1));
        const int b = gv / TV, t = gv - b * TV;
        otab[tid - 2 * (TB + 1)] = (b < B && t < d.TT) ? ((b << 16) | t) : -1;
    }
    __syncthreads();
#ifdef C3_STAMPS
    const unsigned st_p1 = C3_T();
#endif
    // ---- patch pieces of this thread: piece P = (4 u + wave) * 64 + lane of a buffer = physical row P >> 1, half P & 1
    unsigned off0[MAXP], off1[TWO ? MAXP : 1];                                   // byte offsets; C3_OOB = padding
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
        const int P = (u * NWV + wave) * 64 + lane;
        const int prow = P >> 1, half = P & 1;
        const int p = prow / S, rr = prow - p * S;
        int r;
        bool ok = P < NPIECE;
        if (FM == 2) {
            if (rr < P1) { r = 2 * rr; } else { r = 2 * (rr - P1) + 1; }
        } else r = rr;
        ok = ok && r < FR;
        const int f = f0 + r;
        const int fa = ok ? ftab[p] : -1, fb = (ok && C1) ? ftab[(TB + 1) + p] : -1;
        off0[u] = (fa >= 0 && (unsigned)f < (unsigned)d.src[0].F) ? 2u * (unsigned)(fa + f * C0 + half * 8) : C3_OOB;
        if (TWO) off1[u] = (fb >= 0 && (unsigned)f < (unsigned)d.src[1].F) ? 2u * (unsigned)(fb + f * C1 + half * 8) : C3_OOB;
    }
    auto issue_p = [&](int ch, int buf, bool past = false) {      // past: the periodic stream's pieces beyond the last chunk: zeros, no read
        const int second = (TWO && ch * 16 >= C0) ? 1 : 0;
        const int soff = 2 * (second ? ch * 16 - C0 : ch * 16);                 // bytes, scalar
        unsigned char* dst = pbuf + buf * PBYTES + wave * 1024;
#pragma unroll
        for (int u = 0; u < MAXP; ++u) {
            unsigned char* dd = ((u * NWV + wave) * 64 < NPIECE) ? dst + u * (NWV * 1024) : dump;     // wave-uniform
            unsigned vo = past ? C3_OOB : (TWO && second) ? off1[TWO ? u : 0] : off0[u];
            if (ABL & 8) vo = (unsigned)(((bid & 1023) * 16384 + ((u * NWV + wave) * 64 + lane) * 8 + (ch & 15) * 1024) * 2);   // contiguous (wrong) source
            if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (c3_lds_void*)dd, 16, vo, soff, 0, C3_PATCH_AUX);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (c3_lds_void*)dd, 16, vo, soff, 0, C3_PATCH_AUX);
        }
    };
    // ---- weight tile of step (ch, j): [tap 2j + u][128 n][16 channels]; instruction u of wave w: rows 32 w .. 32 w + 31.
    // W in tile order (w_tiled): the step's 8 KB are contiguous, an instruction reads 1 KB of whole lines; otherwise [Npad][K]
    // (32-byte pieces of 5 KB rows: four times the L2 -> L1 line traffic)
    const bool tiled = d.w_tiled != 0;
    const bf16_raw* Wb = reinterpret_cast<const bf16_raw*>(d.W) + (size_t)n0 * d.K;     // tile order: the n-tile's K * 128 elements
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(Wb), 0, 2u * (unsigned)(d.Npad - n0) * (unsigned)d.K, 0x00020000);
    // piece i = 256 u + 64 wave + lane of a tile: tap plane i / (2 BN), row (i / 2) % BN, half i & 1
    // (two named registers, not an array: hipcc's host pass silently dropped the whole kernel when a lambda passed an element of a
    //  captured array to the buffer-load builtin -- no diagnostic, undefined kernel symbol at load time)
    auto woff_of = [&](int u) {
        const int i = NTH * u + 64 * wave + lane;
        const int pl = i / (2 * BN), n = (i >> 1) % BN;
        if (NWN == 4)       // the tile order is by 128 columns: a 256-column tile is two of them, d.K * 128 elements apart
            return 2u * (tiled ? (unsigned)((n >> 7) * 128 * d.K + pl * 2048 + (n & 127) * 16 + (i & 1) * 8)
                               : (unsigned)(n * d.K + pl * Ctot + (i & 1) * 8));
        return 2u * (tiled ? (unsigned)(i * 8) : (unsigned)(n * d.K + pl * Ctot + (i & 1) * 8));
    };
    const unsigned woff0 = woff_of(0), woff1 = woff_of(DW - 1);
    auto issue_w = [&](int ch, int j, int slot, bool past = false) {
        const int soff = past ? 0 : 2 * (tiled ? (ch * H + j) * (2 * (NWN == 4 ? 128 : BN) * 16) : 2 * j * Ctot + ch * 16);
        unsigned char* dst = smem + slot * WSLOT + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (c3_lds_void*)dst, 16, past ? C3_OOB : woff0, soff, 0, 0);
        if (DW > 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (c3_lds_void*)(dst + NWV * 1024), 16, past ? C3_OOB : woff1, soff, 0, 0);
    };

    // ---- fragment addresses
    const int g = lane >> 4, c = lane & 15;
    const int wrd = (g >> 1) * WPL + (wn * (16 * TN) + c) * 32 + (g & 1) * 16;     // + ni * 512
    int vbase;
    {
        int tl, jl;
        c3_row<J, TM>(wm, c, tl, jl);                                               // mi = 0
        vbase = (tl * S + jl) * 32 + (g & 1) * 16;
    }
    auto imm_of = [](int mi) constexpr {
        if (J >= 16) { constexpr int per = J >= 16 ? J / 16 : 1; return ((mi / per) * S + (mi % per) * 16) * 32; }
        if (J == 8) return ((mi >> 1) * 4 + (mi & 1)) * S * 32;
        return ((mi >> 1) * 8 + (mi & 1) * 2) * S * 32;
    };
    auto tap_off = [](int tap) constexpr { return (FM == 2 ? ((tap & 1) * P1 + (tap >> 1)) : tap) * 32; };

    // prologue: patch of chunk 0, weight tiles 0..2 (the periodic stream from here on: step s issues tile s + 3, a chunk's first
    // step also the next chunk's patch), then the first step's wait + barrier
    // the bias of this lane's columns, requested in front of the first DMAs: the accumulators START from it (no bias in the epilogue)
    float4 bv[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
        bv[ni] = d.bias ? *reinterpret_cast<const float4*>(d.bias + nw0 + ni * 16 + 4 * (lane >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef C3_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    const unsigned st_p2 = C3_T();
    __builtin_amdgcn_sched_barrier(0);
#endif
    issue_p(0, 0);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int cc = s / H, jj = s % H;
        issue_w(cc < nch ? cc : 0, jj, s, cc >= nch);
    }
#ifdef C3_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    const unsigned st_p3 = C3_T();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- this wave's destination, from the ntab entries requested at the top (by now they have landed: no round trip is waited for
    // here, and none in the epilogue).  The waves' `dense` flags meet over the loop's first barrier.
    int fdst, fcoff;
    bool dense;
    {
        const int qc = (lane & 15) % (WCOLS_ / 4);
        fdst = __builtin_amdgcn_readfirstlane(nc_first.dst);
        fcoff = __builtin_amdgcn_readfirstlane(nc_first.coff);
        const bool ok = nc_mine.nvalid == 4 && nc_mine.dst == fdst && nc_mine.coff == fcoff + 4 * qc;
        const int f32 = fdst ? d.dst[1].is_f32 : d.dst[0].is_f32, ddC_ = fdst ? d.dst[1].C : d.dst[0].C;
        dense = __all(ok) && !f32 && ((fcoff & 7) == 0) && ((ddC_ & 7) == 0);
    }
    // destination geometry of this wave's columns (scalar selects now: the epilogue touches no kernel argument)
    const int ddF = fdst ? d.dst[1].F : d.dst[0].F, ddC = fdst ? d.dst[1].C : d.dst[0].C, ddT = fdst ? d.dst[1].T : d.dst[0].T;
    const int ddtoff = fdst ? d.dst[1].toff : d.dst[0].toff, ddfmul = fdst ? d.dst[1].fmul : d.dst[0].fmul;
    const int ddfadd = fdst ? d.dst[1].fadd : d.dst[0].fadd;
    void* const ddptr = fdst ? d.dst[1].ptr : d.dst[0].ptr;
    if (d.stats && lane == 0) otab[TB + wave] = dense ? 1 : 0;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int bb = 0; bb < TM; ++bb) acc[a][bb] = (f32x4){bv[a].x, bv[a].y, bv[a].z, bv[a].w};     // the bias is the accumulators' initial value
    c3_wait_step<H, MAXP, DW>(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool with_stats = false;
    if (d.stats) {
        with_stats = true;
#pragma unroll
        for (int i = 0; i < NWV; ++i) with_stats = with_stats && otab[TB + i] != 0;
    }
#ifdef C3_STAMPS
    const unsigned st_t1 = C3_T();
    st_prev = st_t1;
#endif
    int slot = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const bool second = TWO && ch * 16 >= C0;
        const int dtA = second ? d.cv_toff[1][0] - tmin1 : d.cv_toff[0][0] - tmin0;
        const int dtB = second ? d.cv_toff[1][1] - tmin1 : d.cv_toff[0][1] - tmin0;
        const int bufoff = RING + (ch & 1) * PBYTES;
        int aoff[H];                         // per K step: this lane's patch address (its tap of the pair, frame offset, buffer)
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int itA = 2 * j, itB = 2 * j + 1;
            const int oA = (itA / NF ? dtB : dtA) * (S * 32) + tap_off(itA % NF);
            const int oB = (itB / NF ? dtB : dtA) * (S * 32) + tap_off(itB % NF);
            aoff[j] = vbase + bufoff + (g >= 2 ? oB : oA);
        }
#pragma unroll
        for (int j = 0; j < H; ++j) {
            // Step s = (ch, j); its weight tile and patch are visible (the wait + barrier sit in front of the LAST four MFMAs of
            // the previous step: by then every fragment read of that step has returned, so the barrier also frees its slot).
            // Order: 12 fragment reads, the DMA of tile s + 3 (and the next chunk's patch) behind them, 28 MFMAs with the
            // compiler's counted lgkmcnt waits, then wait for step s + 1's operands + barrier, then the last 4 MFMAs.
            __builtin_amdgcn_sched_barrier(0);
#ifdef C3_STAMPS
            const unsigned st_a = C3_T();
            st_tail += st_a - st_prev;
            __builtin_amdgcn_sched_barrier(0);
#endif
            const unsigned char* wslot = smem + slot * WSLOT + wrd;
            const unsigned char* ap = smem + aoff[j];
            bf16x8 wf[TN], af[TM];
            if (ABL & 4) {
                const uint4 cst = make_uint4(0x3f803f80u + j, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + lane);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) wf[ni] = __builtin_bit_cast(bf16x8, cst);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) af[mi] = __builtin_bit_cast(bf16x8, cst);
            } else {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) wf[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wslot + ni * 512));
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) af[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ap + imm_of(mi)));
            }
            if (!(ABL & 1)) {   // tile s + 3 -> the slot tile s - 1 has left; past the end: re-load something harmless (constant counts)
                const int jj = (j + 3) % H, dc = (j + 3) / H;
                const int cc = ch + dc < nch ? ch + dc : 0;
                issue_w(cc, jj, (slot + 3) & 3, ch + dc >= nch);
                if (j == 0) issue_p(ch + 1 < nch ? ch + 1 : 0, (ch + 1) & 1, ch + 1 >= nch);
            }
            if (ABL & 2) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) asm volatile("" ::"v"(wf[ni]));
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) asm volatile("" ::"v"(af[mi]));
            } else {
#pragma unroll
                for (int mi = 0; mi < TM - 1; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
            }
            if (!(ABL & 6)) {
                __builtin_amdgcn_sched_group_barrier(0x100, TN + TM, 0);                       // DS reads
                // Round 5: in the 5-tap instantiations the DMAs are spread between the MFMAs (three MFMAs in front of each) instead of
                // issued in one block in front of them: a DMA instruction holds the wave's issue for 60-180 cycles, and a wave that
                // has no partner on its SIMD (the last round of a launch) then idles its matrix pipe per DMA, not for all of them
                // (dec0.dg 113 -> 104 us, dec1.dg 112 -> 106, same box; the 3- / 2-tap pair launches measured the same or slower).
                if constexpr (NF == 5) {
                    if (j == 0) c3_interleave<DW + MAXP, TN * (TM - 1)>();
                    else c3_interleave<DW, TN * (TM - 1)>();
                } else {
                    if (j == 0) __builtin_amdgcn_sched_group_barrier(0x020, DW + MAXP, 0);     // the DMAs
                    else __builtin_amdgcn_sched_group_barrier(0x020, DW, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, TN * (TM - 1), 0);             // MFMAs
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef C3_STAMPS
            const unsigned st_b = C3_T();
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else c3_wait_step<H, MAXP, DW>((j + 1) % H);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef C3_STAMPS
            __builtin_amdgcn_sched_barrier(0);
            const unsigned st_c = C3_T();
            __builtin_amdgcn_sched_barrier(0);
#endif
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#ifdef C3_STAMPS
            const unsigned st_d = C3_T();
            st_issue += st_b - st_a; st_dma += st_c - st_b; st_bar += st_d - st_c; st_prev = st_d; ++st_n;
            st_imax = st_b - st_a > st_imax ? st_b - st_a : st_imax; st_bmax = st_d - st_c > st_bmax ? st_d - st_c : st_bmax;
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (!(ABL & 2)) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[ni][TM - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[TM - 1], acc[ni][TM - 1], 0, 0, 0);
            }
            slot = (slot + 1) & 3;
        }
    }

    // ---- epilogue.  Everything it needs was fetched or tabulated in the prologue (destination choice `fdst / fcoff / dense`, the frame
    // table `otab`): no division, no descriptor-indexed load and no table look-up in global memory sits between the last MFMA and the
    // stores (round 5: the old epilogue spent 5 000 cycles waiting for such loads and 3 300-8 800 in a store loop that divided per row).
#ifdef C3_STAMPS
    const unsigned st_t2 = C3_T();
    unsigned st_e1 = st_t2;
    auto st_flush = [&]() {
        const unsigned e2 = C3_T();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the output stores have left
        const unsigned t3 = C3_T(), rt1 = (unsigned)__builtin_amdgcn_s_memrealtime();
        if (lane == 0 && c3_stamp_ptr) {
            unsigned* o = c3_stamp_ptr + ((size_t)blockIdx.x * NWV + wave) * 16;
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[0] = blockIdx.x; o[1] = hw; o[2] = xcc; o[3] = st_rt0; o[4] = rt1; o[5] = st_t1 - st_t0; o[6] = st_t2 - st_t1; o[7] = t3 - st_t2;
            o[8] = st_issue; o[9] = st_dma; o[10] = st_bar; o[11] = st_tail; o[12] = st_n; o[13] = st_imax; o[14] = st_bmax; o[15] = 0x5eed;
            unsigned* o2 = o + (size_t)8192 * 8 * 16;      // second record: prologue / epilogue split
            o2[0] = st_p1 - st_t0; o2[1] = st_p2 - st_p1; o2[2] = st_p3 - st_p2; o2[3] = st_t1 - st_p3;
            o2[4] = st_e1 - st_t2; o2[5] = e2 - st_e1; o2[6] = t3 - e2;
        }
    };
#endif
    constexpr int WROWS = 16 * TM, WCOLS = 16 * TN, TP = WCOLS + 8, PPR = WCOLS / 8, RPI = 64 / PPR;   // 16-byte pieces per row, rows per store trip
    // dense path: where the product adds a residual tensor (the skip connection's gradient), the first pieces of it are requested
    // before the drain; the store loop below keeps RD of them in flight (destination offsets come from the frame table: no division)
    constexpr int NIT = WROWS / RPI, RD = NIT < 4 ? NIT : 4;
    const bf16_raw* rptr = (d.res && fdst == 0) ? reinterpret_cast<const bf16_raw*>(d.res) + fcoff : nullptr;
    const int tsz = ddF * ddC, bsz = ddT * ddF * ddC, jsz = ddfmul * ddC;
    if (dense && rptr && with_stats) {
        // A product that carries the layer's fused BatchNorm sums AND a residual (round 6: the main-source half of a decoder's forward
        // product, whose skip-connection half ran earlier, under the LSTM, into `res`): the sums are of product + residual, so the
        // residual joins the accumulators BEFORE the tile is staged -- 8-byte loads in the accumulator layout (row 16 mi + lane % 16,
        // columns 16 ni + 4 (lane / 16) ...), all requested before the first use; the store loop then adds nothing.
        const int basea = (ddtoff * ddF + ddfadd) * ddC + 4 * (lane >> 4);
        uint2 rv[TM][TN];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            int tl, jl;
            c3_row<J, TM>(wm, mi * 16 + (lane & 15), tl, jl);
            const int bt = otab[tl];
            const int ro = bt >= 0 ? (bt >> 16) * bsz + (bt & 0xffff) * tsz + jl * jsz + basea : -1;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                rv[mi][ni] = ro >= 0 ? *reinterpret_cast<const uint2*>(rptr + ro + ni * 16) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                acc[ni][mi][0] += __uint_as_float(rv[mi][ni].x << 16); acc[ni][mi][1] += __uint_as_float(rv[mi][ni].x & 0xffff0000u);
                acc[ni][mi][2] += __uint_as_float(rv[mi][ni].y << 16); acc[ni][mi][3] += __uint_as_float(rv[mi][ni].y & 0xffff0000u);
            }
        rptr = nullptr;
    }
    const int base0 = (ddtoff * ddF + ddfadd) * ddC + (lane % PPR) * 8;
    auto row_off = [&](int itr) {
        int tl, jl;
        c3_row<J, TM>(wm, itr * RPI + lane / PPR, tl, jl);
        const int bt = otab[tl];                               // (utterance << 16) | frame, -1 = outside the tensor
        return bt >= 0 ? (bt >> 16) * bsz + (bt & 0xffff) * tsz + jl * jsz + base0 : -1;
    };
    uint4 r4[RD];
    if (dense && rptr) {
#pragma unroll
        for (int itr = 0; itr < RD; ++itr) {
            const int o = row_off(itr);
            r4[itr] = o >= 0 ? *reinterpret_cast<const uint4*>(rptr + o) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // every DMA has landed (the run-ahead ones past the end are out-of-range pieces: zeros, no memory read) and every wave has
    // finished reading before the LDS is reused
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (dense) {
        bf16_raw* tb_ = reinterpret_cast<bf16_raw*>(smem) + wave * (WROWS * TP);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            bool live = true;                                      // (fused sums only: rows outside the tensor are staged as zeros)
            if (with_stats) {
                int tl, jl;
                c3_row<J, TM>(wm, mi * 16 + (lane & 15), tl, jl);
                live = otab[tl] >= 0;
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const f32x4 v = acc[ni][mi];
                *reinterpret_cast<uint2*>(&tb_[(mi * 16 + (lane & 15)) * TP + ni * 16 + 4 * (lane >> 4)]) =
                    live ? make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])) : make_uint2(0u, 0u);
            }
        }
        if (with_stats) {
            // Batch statistics of the ComplexBatchNorm that follows, from the values as stored, ON THE MATRIX CORES (round 5; the VALU
            // form was ~700 instructions per wave with all eight waves of the CU in it at once: 6 000 of the epilogue's cycles).
            // Wave (wm, 0) holds the real halves and wave (wm, 1) the imaginary halves of the tile's complex channels for the same
            // rows; wave (wm, wn) takes the channels WCOLS / 2 * wn .. of both images.  With Yr, Yi = [rows][16 channels] slabs read
            // as transposed fragments (ds_read_b64_tr_b16: a lane gets 8 consecutive ROWS of one channel -- the A and the B operand
            // layout alike), the sums over rows are the diagonals of Yr^T Yr, Yr^T Yi, Yi^T Yi and any row of 1^T Yr, 1^T Yi: five
            // MFMAs per 32 rows and 16 channels.  Rows outside the tensor were staged as zeros (`live` above).
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bf16_raw* imr = reinterpret_cast<const bf16_raw*>(smem) + (wm + 2 * (wn & ~1)) * (WROWS * TP);
            const bf16_raw* imi = reinterpret_cast<const bf16_raw*>(smem) + (wm + 2 * (wn | 1)) * (WROWS * TP);
            constexpr int NCG = WCOLS / 32;                        // 16-channel groups per wave: 2 (128-column tiles) or 1
            const int i16 = lane & 15, gq = lane >> 4;
            const int trow = 8 * gq + (i16 >> 2), tcol = (WCOLS / 2) * (wn & 1) + 4 * (i16 & 3);
            const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
            const bool diag = gq == (i16 >> 2);                    // this lane holds D[c][c] of its column c = i16, in component i16 & 3
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                f32x4 a_r = {0.f, 0.f, 0.f, 0.f}, a_i = a_r, a_rr = a_r, a_ri = a_r, a_ii = a_r;
#pragma unroll
                for (int ks = 0; ks < WROWS / 32; ++ks) {
                    const int o = (ks * 32 + trow) * TP + tcol + cg * 16;
                    const s16x4 rl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_s16x4*)&imr[o]);
                    const s16x4 rh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_s16x4*)&imr[o + 4 * TP]);
                    const s16x4 il = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_s16x4*)&imi[o]);
                    const s16x4 ih = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_s16x4*)&imi[o + 4 * TP]);
                    const bf16x8 yr = __builtin_bit_cast(bf16x8, __builtin_shufflevector(rl, rh, 0, 1, 2, 3, 4, 5, 6, 7));
                    const bf16x8 yi = __builtin_bit_cast(bf16x8, __builtin_shufflevector(il, ih, 0, 1, 2, 3, 4, 5, 6, 7));
                    a_r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yr, a_r, 0, 0, 0);
                    a_i = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yi, a_i, 0, 0, 0);
                    a_rr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yr, yr, a_rr, 0, 0, 0);
                    a_ri = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yr, yi, a_ri, 0, 0, 0);
                    a_ii = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yi, yi, a_ii, 0, 0, 0);
                }
                if (diag) {
                    const int e = i16 & 3;
                    auto pick = [&](const f32x4& v) { return e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : v[3]; };
                    const int Cr = d.stats_cr;
                    float* sp = d.stats + (size_t)(blockIdx.x & 7) * 5 * Cr + (n0 >> 1) + 64 * (wn >> 1) + (WCOLS / 2) * (wn & 1) + cg * 16 + i16;
                    atomicAdd(sp, a_r[0]); atomicAdd(sp + Cr, a_i[0]);
                    atomicAdd(sp + 2 * Cr, pick(a_rr)); atomicAdd(sp + 3 * Cr, pick(a_ri)); atomicAdd(sp + 4 * Cr, pick(a_ii));
                }
            }
        }
#ifdef C3_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        st_e1 = C3_T();
        __builtin_amdgcn_sched_barrier(0);
#endif
        bf16_raw* dptr = reinterpret_cast<bf16_raw*>(ddptr) + fcoff;
#pragma unroll
        for (int itr = 0; itr < NIT; ++itr) {
            const int row = itr * RPI + lane / PPR;
            uint4 v = *reinterpret_cast<const uint4*>(&tb_[row * TP + (lane % PPR) * 8]);
            const int off = row_off(itr);
            if (rptr) {
                const uint4 rr = r4[itr % RD];
                if (itr + RD < NIT) {                              // the piece RD rows ahead takes the register this one leaves
                    const int o2 = row_off(itr + RD);
                    r4[itr % RD] = o2 >= 0 ? *reinterpret_cast<const uint4*>(rptr + o2) : make_uint4(0u, 0u, 0u, 0u);
                }
                const unsigned av[4] = {v.x, v.y, v.z, v.w}, rv[4] = {rr.x, rr.y, rr.z, rr.w};
                unsigned o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    o[i] = pack_bf2(__uint_as_float(av[i] << 16) + __uint_as_float(rv[i] << 16),
                                    __uint_as_float(av[i] & 0xffff0000u) + __uint_as_float(rv[i] & 0xffff0000u));
                v = make_uint4(o[0], o[1], o[2], o[3]);
            }
            if (off >= 0) *reinterpret_cast<uint4*>(dptr + off) = v;
        }
#ifdef C3_STAMPS
        st_flush();
#endif
        return;
    }
    // direct scatter, 4 consecutive channels per lane (fp32 destinations, narrow or split column groups)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        int tl, jl;
        c3_row<J, TM>(wm, mi * 16 + (lane & 15), tl, jl);
        const int bt = otab[tl];
        if (bt < 0) continue;
        const int b = bt >> 16, t = bt & 0xffff;
        size_t ro[2];
        ro[0] = d.dst[0].ptr ? (((size_t)b * d.dst[0].T + t + d.dst[0].toff) * d.dst[0].F + (size_t)jl * d.dst[0].fmul + d.dst[0].fadd) * d.dst[0].C : 0;
        ro[1] = d.dst[1].ptr ? (((size_t)b * d.dst[1].T + t + d.dst[1].toff) * d.dst[1].F + (size_t)jl * d.dst[1].fmul + d.dst[1].fadd) * d.dst[1].C : 0;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = nw0 + ni * 16 + 4 * (lane >> 4);
            const sehip_nchunk nc = d.ntab[n >> 2];
            if (nc.nvalid <= 0) continue;
            f32x4 v = acc[ni][mi];
            const size_t off = (nc.dst ? ro[1] : ro[0]) + nc.coff;
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
#ifdef C3_STAMPS
    st_flush();
#endif
}

template <int NF, int FM, int J, int TM, int TN, int NWN = 2, int ABL = 0>
__global__ __launch_bounds__(128 * NWN, NWN == 2 ? 2 : 1) void conv_gemm_v3_kernel(const sehip_gemm_desc d, int B, int order) {
    c3_body<NF, FM, J, TM, TN, NWN, ABL>(d, B, order, (int)blockIdx.x, (int)gridDim.x);
}
// The last, partial round of a launch as HALF tiles: a launch of 650 tiles on 512 slots runs 138 tiles alone on their CUs for a whole
// tile life while 118 CUs idle; as 276 tiles of half the height the same rows occupy every CU for about half as long.  Workgroups
// [0, nfull) run full tiles (TM = 8), the rest half tiles (TM = 4) of the remaining tiles.
template <int NF, int FM, int J, int TN>
__global__ __launch_bounds__(256, 2) void conv_gemm_v3_tail_kernel(const sehip_gemm_desc d, int B, int order, int nfull) {
    if ((int)blockIdx.x < nfull) c3_body<NF, FM, J, 8, TN, 2, 0>(d, B, order, (int)blockIdx.x, nfull);
    else c3_body<NF, FM, J, 4, TN, 2, 0>(d, B, order, (int)blockIdx.x - nfull, (int)gridDim.x - nfull, nfull, true);
}
// Both output-row parities of a transposed convolution (3 and 2 row taps over the same sources: sehip_gemm_pair) in ONE launch: the
// first `na` workgroups run product a's tiles, the rest product b's.  Two launches of ~1.3 rounds of tiles each leave their last
// rounds half empty and pay two launch ramps; together the tiles of the second product fill the first one's tail.
template <int J, int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_gemm_v3_pair_kernel(const sehip_gemm_desc da, const sehip_gemm_desc db, int B, int order, int na) {
    if ((int)blockIdx.x < na) c3_body<3, 1, J, TM, TN, 2, 0>(da, B, order, (int)blockIdx.x, na);
    else c3_body<2, 1, J, TM, TN, 2, 0>(db, B, order, (int)blockIdx.x - na, (int)gridDim.x - na);
}

static int c3_order() {
    static const int o = getenv("SEHIP_C3_ORDER") ? atoi(getenv("SEHIP_C3_ORDER")) : 0;
    return o;
}
template <int NF, int FM, int J, int TM, int TN, int NWN = 2>
static size_t c3_lds_bytes() {
    using G = C3Geo<NF, FM, J, TM, NWN>;
    const size_t epi = 2 * NWN * (16 * TM * (16 * TN + 8) * 2) + 64, main_ = (size_t)G::LDS_MAIN - G::TABLES;
    return (main_ > epi ? main_ : epi) + G::TABLES;      // the tables sit above both images (kernel: TBL)
}
// The dynamic-LDS limit of a kernel is a PER-DEVICE attribute: set once per (instantiation, device), the result checked (a
// process-wide flag set it on whichever device was current first, and a failure surfaced as an unexplained launch error).
template <int NF, int FM, int J, int TM, int TN, int NWN = 2>
static bool c3_set_attr() {
    static unsigned char state[64] = {};      // per device: 0 = not tried, 1 = set, 2 = failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return false; }
    if (state[dev] == 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_v3_kernel<NF, FM, J, TM, TN, NWN>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (NWN == 2 ? 80 : 160) * 1024);
        if (e != hipSuccess) (void)hipGetLastError();
        state[dev] = e == hipSuccess ? 1 : 2;
    }
    return state[dev] == 1;
}
template <int NF, int FM, int J, int TM, int TN, int NWN = 2>
static int c3_launch(const sehip_gemm_desc& d, int B, int grid, hipStream_t st) {
    if (!c3_set_attr<NF, FM, J, TM, TN, NWN>()) return 0;      // sehip_gemm then reports that the tile-ordered weights found no kernel
    sehip_note_kernel("conv_gemm_v3_kernel<%d, %d, %d, %d, %d, %d, 0>", NF, FM, J, TM, TN, NWN);
    conv_gemm_v3_kernel<NF, FM, J, TM, TN, NWN><<<grid, 128 * NWN, c3_lds_bytes<NF, FM, J, TM, TN, NWN>(), st>>>(d, B, c3_order());
    return 1;
}
template <int NF, int FM, int J, int TN>
static int c3_launch_tail(const sehip_gemm_desc& d, int B, int nfull, int nhalf, hipStream_t st) {
    static unsigned char state[64] = {};      // per device: 0 = not tried, 1 = set, 2 = failed (as c3_set_attr)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
    if (state[dev] == 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_v3_tail_kernel<NF, FM, J, TN>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) (void)hipGetLastError();
        state[dev] = e == hipSuccess ? 1 : 2;
    }
    if (state[dev] != 1) return 0;
    const size_t l8 = c3_lds_bytes<NF, FM, J, 8, TN>(), l4 = c3_lds_bytes<NF, FM, J, 4, TN>();
    sehip_note_kernel("conv_gemm_v3_tail_kernel<%d, %d, %d, %d>", NF, FM, J, TN);
    conv_gemm_v3_tail_kernel<NF, FM, J, TN><<<nfull + nhalf, 256, l8 > l4 ? l8 : l4, st>>>(d, B, c3_order(), nfull);
    return 1;
}
#ifdef SEHIP_TOOLS_BUILD
template <int J, int ABL>
static void c3_launch_abl(const sehip_gemm_desc& d, int B, int grid, hipStream_t st) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_v3_kernel<5, 2, J, 8, 4, 2, ABL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    conv_gemm_v3_kernel<5, 2, J, 8, 4, 2, ABL><<<grid, 256, c3_lds_bytes<5, 2, J, 8, 4>(), st>>>(d, B, c3_order());
}
#endif
// Rows per tile: 256 (TM 8) or 192 (TM 6).  Tiles run two per CU (512 slots).  Measured at the headline shapes (B (T + 2) = 10400
// frames): where 256-row tiles do not even fill one round (326 tiles) 192-row tiles (434) take 10-15 % less time; from 650 tiles
// up the launch is bound by the LDS-DMA rate of the CUs that hold two workgroups and the smaller tile (more weight bytes per
// FLOP) is the same or slower although its last round is fuller.
static int c3_pick_tm(long vframes, int J, int ntn) {
    static const int force = getenv("SEHIP_C3_TM") ? atoi(getenv("SEHIP_C3_TM")) : 0;
    if (force == 8 || force == 6) return force;
    const int TB = 256 / J;
    return (vframes + TB - 1) / TB * ntn < 512 ? 6 : 8;
}
template <int NF, int FM>
static int c3_launch_j(const sehip_gemm_desc& d, int B, hipStream_t st) {
    const long vframes = (long)B * (d.TT + 2);
    const int BN = (d.Npad & 127) ? 64 : 128;             // 64-column tiles only for the layers whose width is not a multiple of 128
    const int ntn = d.Npad / BN;
#ifdef SEHIP_TOOLS_BUILD      // timing ablations (wrong results) and the 8-wave 256 x 256 experiment: tools builds only
      // (python speech-enhancement-pytorch_amd/sehip/build.py --tools), never in the product library
    static const int abl = getenv("SEHIP_C3_ABL") ? atoi(getenv("SEHIP_C3_ABL")) : 0;     // tools/ only: timing ablations, wrong results
    if (abl && NF == 5 && BN == 128 && (d.J == 4 || d.J == 8)) {
        const int TB = 256 / d.J, grid = (int)((vframes + TB - 1) / TB) * ntn;
#define C3_ABL(A_) case A_: if (d.J == 4) c3_launch_abl<4, A_>(d, B, grid, st); else c3_launch_abl<8, A_>(d, B, grid, st); return 1;
        switch (abl) { C3_ABL(1) C3_ABL(2) C3_ABL(3) C3_ABL(4) C3_ABL(5) C3_ABL(6) C3_ABL(8) C3_ABL(14) default: break; }
#undef C3_ABL
    }
    // 8-wave build (tools/ experiment, SEHIP_C3_W8): 256 x 256 tiles, one workgroup per CU
    static const int w8 = getenv("SEHIP_C3_W8") ? atoi(getenv("SEHIP_C3_W8")) : 0;
    if (w8 && NF == 5 && (d.Npad & 255) == 0 && (d.J == 4 || d.J == 8)) {
        const int TB = 256 / d.J, grid = (int)((vframes + TB - 1) / TB) * (d.Npad / 256);
        return d.J == 4 ? c3_launch<5, 2, 4, 8, 4, 4>(d, B, grid, st) : c3_launch<5, 2, 8, 8, 4, 4>(d, B, grid, st);
    }
#endif
    const int tm = c3_pick_tm(vframes, d.J, ntn);
    // tail form: full rounds of 512 tiles as they are, the partial last round (at most 300 tiles: they would run alone on their CUs)
    // as half tiles
    // (opt-in, SEHIP_C3_TAIL=1: built, tested, measured neutral -- a workgroup alone on its CU takes as long per K step with a
    //  half tile as with a full one)
    static const bool tail = getenv("SEHIP_C3_TAIL") != nullptr;
    if (tail && tm == 8) {
        const int TB8 = 256 / d.J, nt8 = (int)((vframes + TB8 - 1) / TB8) * ntn, nfull = nt8 / 512 * 512, rem = nt8 - nfull;
        if (nfull > 0 && rem > 0 && rem <= 300) {
#define C3T_CASE(J_)                                                                                                 \
            case J_: return BN == 128 ? c3_launch_tail<NF, FM, J_, 4>(d, B, nfull, 2 * rem, st) : c3_launch_tail<NF, FM, J_, 2>(d, B, nfull, 2 * rem, st);
            switch (d.J) {
                C3T_CASE(4) C3T_CASE(8) C3T_CASE(16) C3T_CASE(32)
                default: break;
            }
#undef C3T_CASE
        }
    }
#define C3_CASE(J_)                                                                                       \
    case J_: {                                                                                            \
        const int TB = 32 * tm / J_, grid = (int)((vframes + TB - 1) / TB) * ntn;                         \
        if (BN == 128)                                                                                    \
            return tm == 8 ? c3_launch<NF, FM, J_, 8, 4>(d, B, grid, st) : c3_launch<NF, FM, J_, 6, 4>(d, B, grid, st); \
        return tm == 8 ? c3_launch<NF, FM, J_, 8, 2>(d, B, grid, st) : c3_launch<NF, FM, J_, 6, 2>(d, B, grid, st);     \
    }
    switch (d.J) {
        C3_CASE(4) C3_CASE(8) C3_CASE(16) C3_CASE(32)
        default: return 0;
    }
#undef C3_CASE
}

// does the product qualify for the kernel?  (B = utterances)
static bool c3_qualifies(const sehip_gemm_desc& d, int* Bout) {
    static const bool disabled = getenv("SEHIP_NO_CONV_V3") != nullptr || getenv("SEHIP_NO_PATCH") != nullptr;
    if (disabled || d.cv_nf <= 0 || d.tmul > 1) return false;
    if (d.stats && (d.dst[1].ptr || d.dst[0].is_f32 || (d.dst[0].C & 7) || d.stats_cr * 2 != d.Npad)) return false;
    const int C0 = d.src[0].C, C1 = d.src[1].ptr ? d.src[1].C : 0;
    if ((C0 & 15) || (C1 & 15) || (d.Npad & 63)) return false;
    if (d.J != 4 && d.J != 8 && d.J != 16 && d.J != 32) return false;
    if (d.K != 2 * d.cv_nf * (C0 + C1)) return false;
    if (d.cv_nf == 5 && C1) return false;                                    // (the 5-tap instantiations carry one source's offsets)
    if ((d.dst[0].tmul > 1) || (d.dst[1].ptr && d.dst[1].tmul > 1)) return false;
    if (d.TT >= 65535 || d.M % (d.TT * d.J)) return false;                  // (the destination frame table packs (utterance << 16) | frame)
    const int B = d.M / (d.TT * d.J);
    if (B >= 32768) return false;
    for (int s = 0; s < 2; ++s) {
        if (!d.src[s].ptr) continue;
        for (int kt = 0; kt < 2; ++kt)
            if (d.cv_toff[s][kt] < -1 || d.cv_toff[s][kt] > 1) return false;
        if (abs(d.cv_toff[s][0] - d.cv_toff[s][1]) > 1) return false;
        if (d.src[s].thi > d.TT + 1) return false;
        if ((long)B * d.src[s].T * d.src[s].F * d.src[s].C >= (1L << 30) - (1L << 20)) return false;      // byte offsets (and num_records) below the padding marker
    }
    for (int s = 0; s < 2; ++s)
        if (d.dst[s].ptr && (long)B * d.dst[s].T * d.dst[s].F * d.dst[s].C >= (1L << 31)) return false;
    if ((long)d.Npad * d.K >= (1L << 30) - (1L << 20)) return false;
    if (!((d.cv_nf == 5 && d.fmul == 2) || (d.cv_nf == 3 && d.fmul == 1) || (d.cv_nf == 2 && d.fmul == 1))) return false;
    *Bout = B;
    return true;
}

// returns 1 if the kernel was launched, 0 if the descriptor does not qualify (the caller falls back to conv_gemm_v2 / v1)
int sehip_try_conv_gemm_v3(const sehip_gemm_desc& d, hipStream_t st) {
    int B = 0;
    if (!c3_qualifies(d, &B)) return 0;
    if (d.cv_nf == 5 && d.fmul == 2) return c3_launch_j<5, 2>(d, B, st);
    if (d.cv_nf == 3 && d.fmul == 1) return c3_launch_j<3, 1>(d, B, st);
    if (d.cv_nf == 2 && d.fmul == 1) return c3_launch_j<2, 1>(d, B, st);
    return 0;
}

// ---- the two output-row parities of a transposed convolution (or of a stride-2 convolution's input gradient) in one launch
template <int J, int TM, int TN>
static int c3_launch_pair(const sehip_gemm_desc& a, const sehip_gemm_desc& b, int B, int na, int nb, hipStream_t st) {
    static unsigned char state[64] = {};      // per device: 0 = not tried, 1 = set, 2 = failed (as c3_set_attr)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
    if (state[dev] == 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_v3_pair_kernel<J, TM, TN>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) (void)hipGetLastError();
        state[dev] = e == hipSuccess ? 1 : 2;
    }
    if (state[dev] != 1) return 0;
    const size_t la = c3_lds_bytes<3, 1, J, TM, TN>(), lb = c3_lds_bytes<2, 1, J, TM, TN>();
    sehip_note_kernel("conv_gemm_v3_pair_kernel<%d, %d, %d>", J, TM, TN);
    conv_gemm_v3_pair_kernel<J, TM, TN><<<na + nb, 256, la > lb ? la : lb, st>>>(a, b, B, c3_order(), na);
    return 1;
}
// a: the 3-tap product, b: the 2-tap product over the same sources (any order is accepted).  1 = launched.
int sehip_try_conv_gemm_v3_pair(const sehip_gemm_desc& a_, const sehip_gemm_desc& b_, hipStream_t st) {
    static const bool off = getenv("SEHIP_NO_C3_PAIR") != nullptr;
    if (off) return 0;
    const sehip_gemm_desc& a = a_.cv_nf == 3 ? a_ : b_;
    const sehip_gemm_desc& b = a_.cv_nf == 3 ? b_ : a_;
    int Ba = 0, Bb = 0;
    if (!c3_qualifies(a, &Ba) || !c3_qualifies(b, &Bb)) return 0;
    if (a.cv_nf != 3 || b.cv_nf != 2 || a.fmul != 1 || b.fmul != 1) return 0;
    if (Ba != Bb || a.J != b.J || a.TT != b.TT || a.Npad != b.Npad) return 0;
    const long vframes = (long)Ba * (a.TT + 2);
    const int BN = (a.Npad & 127) ? 64 : 128, ntn = a.Npad / BN;
    // rows per tile: the candidate (256 or 192) with the smaller (rounds of 512 tiles) x (tile height) for BOTH products together
    static const int force = getenv("SEHIP_C3_PAIR_TM") ? atoi(getenv("SEHIP_C3_PAIR_TM")) : 0;
    auto tiles = [&](int tm) { const int TB = 32 * tm / a.J; return (int)((vframes + TB - 1) / TB) * ntn; };
    int tm = ((2 * tiles(8) + 511) / 512) * 8 <= ((2 * tiles(6) + 511) / 512) * 6 ? 8 : 6;
    if (force == 8 || force == 6) tm = force;
    const int n1 = tiles(tm);
#define C3P_CASE(J_)                                                                                                        \
    case J_:                                                                                                                \
        if (BN == 128) return tm == 8 ? c3_launch_pair<J_, 8, 4>(a, b, Ba, n1, n1, st) : c3_launch_pair<J_, 6, 4>(a, b, Ba, n1, n1, st); \
        return tm == 8 ? c3_launch_pair<J_, 8, 2>(a, b, Ba, n1, n1, st) : c3_launch_pair<J_, 6, 2>(a, b, Ba, n1, n1, st);
    switch (a.J) {
        C3P_CASE(4) C3P_CASE(8) C3P_CASE(16) C3P_CASE(32)
        default: return 0;
    }
#undef C3P_CASE
}

template <int NF, int FM, int J, int TM, int TN>
static void c3_init_k() {
    c3_set_attr<NF, FM, J, TM, TN>();
}
template <int NF, int FM, int J>
static void c3_init_one() {
    c3_init_k<NF, FM, J, 8, 4>(); c3_init_k<NF, FM, J, 6, 4>(); c3_init_k<NF, FM, J, 8, 2>(); c3_init_k<NF, FM, J, 6, 2>();
}
template <int NF, int FM>
static void c3_init_nf() {
    c3_init_one<NF, FM, 4>(); c3_init_one<NF, FM, 8>(); c3_init_one<NF, FM, 16>(); c3_init_one<NF, FM, 32>();
}
void sehip_conv3_init(void) {
    c3_init_nf<5, 2>(); c3_init_nf<3, 1>(); c3_init_nf<2, 1>();
}
