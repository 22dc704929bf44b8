// dense_tile_wgrad_kernel: the weight gradients of a GROUP of plain dense-row products in one streaming launch --
//     dW_p[n][k] += sum_m dOut_p[m][n] * A_p[m][k],   m = (utterance, frame), up to 16 products p
// built for the twelve LSTM products of NavieComplexLSTM (src/model/dccrn.py:264-302: W_ih of both layers for the real and the
// imaginary input, W_hh of both layers for the four (lstm, input part) combinations).  They are 10 336-deep GEMMs with 512 x 512,
// 512 x 128 and 256 x 64 outputs; the table-gathered wgrad_group_kernel ran them at 0.045 of the MFMA peak and read 381 MB for
// ~60 MB of operands (round 4: 207 us at the end of the weight-gradient queue, which is what the step waits for since round 5).
//
// Construction (csrc/wgrad3.hip dense_wgrad_kernel, generalised): a workgroup of 8 waves owns ONE tile of one product's dW -- 128 x 256,
// 256 x 128 or 256 x 64 accumulators in registers -- and a range of 64-row stages; both operands reach LDS by LDS-DMA as
// [16 columns][64 rows][32 B] planes (transposed fragment reads without bank conflicts), NB stage buffers, one barrier per stage.
// What is general here is the ADDRESSING, resolved on the host once per binding (sehip_wgrad_dense_group_prepare reads each
// product's chunk table): every 16-column plane of A is {source tensor, element offset inside the row, frame shift}, so the planes
// of one tile may come from two tensors (layer 2's input = two h1 parts), from strided pieces of a wider row (layer 1's input =
// 4 x 128 channels of the encoder output) or from the PREVIOUS frame (the recurrent products: A = h[t - 1], zero at t = 0 -- a
// per-thread frame counter, no division in the loop).  Tiles are cut so that every workgroup streams about the same number of
// rows; a tile's partial sums go to `scratch` by plain stores and dtw_reduce_kernel adds the splits in a fixed order into dW
// (deterministic by construction: no atomics).
#include <stdlib.h>
#include <vector>
#include "common.h"
#include "../../../include/sehip.h"

typedef __attribute__((address_space(3))) void dtw_lds_void;
typedef __attribute__((address_space(3))) s16x4 dtw_lds_s16x4;
#define DTW_OOB 0x7ffffff0u
#define DTW_MAXP 16

struct DtwProd {                 // one product (device copy)
    const bf16_raw* g;           // dOut tensor, row (b, t) at g + ((b * Tg + t + gtoff) * gstride)
    const bf16_raw* x[2];        // source tensors, row (b, t) at x[s] + ((b * Tx[s] + t) * xstride[s])
    float* dW;                   // [N][K] fp32
    float* dbias;                // [N] fp32 column sums of dOut, or NULL
    int gstride, gcol0, xstride[2];
    int T, M, N, K;              // T: frames per utterance of the ROW space (M = B * T)
    int plane0;                  // first entry of this product in the plane table
    int B, Tg, gtoff;            // utterances; frames per utterance stored in dOut, frame offset of row t in it
    int Tx[2], tlo[2], thi[2];   // frames per utterance stored in the sources, their valid frame ranges
    unsigned gbytes, xbytes[2];  // tensor sizes (num_records)
    int pad_[1];
};
struct DtwPlane { int src, delta, shift, live; };      // 16 columns of A: tensor, element offset inside the row, source frame = t - shift; live 0: zero columns (K padding)
struct DtwTile { int prod, n0, k0, s_begin, s_end, out_off, cls, bias; };   // out_off: floats into scratch, -1: the only split of its tile, adds to dW directly; bias: also the column sums of dOut
struct DtwRed { int first, prod, n0, k0, N, K, parts_off, nparts, part_stride, bias, liveN, liveK; };   // one (product, tile): blocks [first, ...); bias: one more block; live*: the part of the tile inside [Npad][K]

template <int N>
__device__ __forceinline__ void dtw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int N, int K, int NB>
__device__ __forceinline__ void dtw_body(const DtwProd& P, const DtwPlane* __restrict__ planes, const DtwTile& t, float* __restrict__ scratch,
                                         unsigned char* smem) {
    constexpr int WN = N >= 256 ? 4 : 2, WK = 8 / WN;               // waves along n / along k
    constexpr int TN = N / WN / 16, TK = K / WK / 16;               // MFMA tiles per wave
    constexpr int GB = N * 64 * 2, XB = K * 64 * 2, STAGE = GB + XB;
    constexpr int GI = GB / 16 / 512, XI = XB / 16 / 512 > 0 ? XB / 16 / 512 : 1, NI = GI + XI;
    static_assert(GB % (16 * 512) == 0 && XB % (16 * 512) == 0, "whole DMA instructions");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv % WN, wk = wv / WN;
    const int ns = t.s_end - t.s_begin;
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(P.g), 0, P.gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(P.x[0]), 0, P.xbytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(P.x[1] ? P.x[1] : P.x[0]), 0,
                                                                          P.x[1] ? P.xbytes[1] : P.xbytes[0], 0x00020000);
    // piece i = (8 u + wave) * 64 + lane of an image: plane i >> 7, row (i >> 1) & 63, half i & 1.  The row and the half depend on
    // (lane, wave parity) only; the plane is wave-uniform per instruction.
    const int row = (lane >> 1) + 32 * (wv & 1), half = lane & 1;
    const int r0 = t.s_begin * 64 + row;
    // per instruction: the column part of the offset (elements) and whether the plane exists (the tile may reach past Npad / K)
    int gcol[GI];
    bool glive[GI];
#pragma unroll
    for (int u = 0; u < GI; ++u) {
        const int pl = (u * 8 + wv) >> 1;
        glive[u] = t.n0 + pl * 16 < P.N;
        gcol[u] = P.gcol0 + t.n0 + pl * 16 + half * 8;
    }
    // A planes: per instruction the tensor (wave-uniform), the column part, the frame shift
    int xcol[XI], xsrc[XI], xshift[XI];
    bool xlive[XI];
#pragma unroll
    for (int u = 0; u < XI; ++u) {
        const int pl = (u * 8 + wv) >> 1;
        const bool in = pl < K / 16 && t.k0 + pl * 16 < P.K;       // (K = 64: four planes, waves 0..7 cover them with one instruction)
        const DtwPlane e = planes[P.plane0 + (in ? (t.k0 >> 4) + pl : 0)];
        xlive[u] = in && __builtin_amdgcn_readfirstlane(e.live) != 0;
        xsrc[u] = __builtin_amdgcn_readfirstlane(e.src);
        xshift[u] = __builtin_amdgcn_readfirstlane(e.shift);
        xcol[u] = e.delta + half * 8;
    }
    // this thread's row as (utterance, frame); every issue advances it by the 64 rows of a stage (no division in the loop)
    int ub = r0 / P.T, tm = r0 - ub * P.T;
    auto issue = [&](int buf, bool past) {             // stages are issued in order; past the range: offsets beyond num_records, zeros (constant DMA count)
        unsigned char* base = smem + buf * STAGE + wv * 1024;
        const bool rowlive = !past && ub < P.B;
        const int grow = ((ub * P.Tg + tm + P.gtoff) * P.gstride);
#pragma unroll
        for (int u = 0; u < GI; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, (dtw_lds_void*)(base + u * 8192), 16,
                                                     (rowlive && glive[u]) ? 2u * (unsigned)(grow + gcol[u]) : DTW_OOB, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < XI; ++u) {
            const int sx = xsrc[u];
            const int ts = tm - xshift[u];
            const bool ok = rowlive && xlive[u] && ts >= (sx ? P.tlo[1] : P.tlo[0]) && ts < (sx ? P.thi[1] : P.thi[0]);
            const int xrow = (ub * (sx ? P.Tx[1] : P.Tx[0]) + ts) * (sx ? P.xstride[1] : P.xstride[0]);
            const unsigned vo = ok ? 2u * (unsigned)(xrow + xcol[u]) : DTW_OOB;
            unsigned char* dd = base + GB + u * 8192;
            if (sx) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx1, (dtw_lds_void*)dd, 16, vo, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx0, (dtw_lds_void*)dd, 16, vo, 0, 0, 0);
        }
        tm += 64;                                      // the next stage's row
        while (tm >= P.T) { tm -= P.T; ++ub; }
    };
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    const int ga = (wn * TN) * 2048 + (4 * g + q) * 32 + 8 * p4;              // + tn * 2048 + ks * 1024 + h * 512
    const int xa = GB + (wk * TK) * 2048 + (4 * g + q) * 32 + 8 * p4;
    f32x4 acc[TN][TK], accb[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < TK; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = t.bias != 0 && wk == 0;      // the column sums of dOut (dbias): one more MFMA per row tile against ones
    const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
#pragma unroll
    for (int s = 0; s < NB - 1; ++s) issue(s, s >= ns);
    int buf = 0;
    for (int s = 0; s < ns; ++s) {
        dtw_wait_vm<(NB - 2) * NI>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(buf == 0 ? NB - 1 : buf - 1, s + NB - 1 >= ns);
        const unsigned char* sb = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 gf[TN], xf[TK];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dtw_lds_s16x4*)(sb + ga + tn * 2048 + ks * 1024));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dtw_lds_s16x4*)(sb + ga + tn * 2048 + ks * 1024 + 512));
                gf[tn] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dtw_lds_s16x4*)(sb + xa + tk * 2048 + ks * 1024));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dtw_lds_s16x4*)(sb + xa + tk * 2048 + ks * 1024 + 512));
                xf[tk] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tk = 0; tk < TK; ++tk)
                    acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[tn], xf[tk], acc[tn][tk], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) accb[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[tn], ones, accb[tn], 0, 0, 0);
            }
        }
        buf = buf == NB - 1 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t.out_off < 0) {                              // the tile's only split: straight into dW (one writer per element)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tk = 0; tk < TK; ++tk) {
                const int n = t.n0 + (wn * TN + tn) * 16 + 4 * (lane >> 4);
                const int k = t.k0 + (wk * TK + tk) * 16 + (lane & 15);
                if (k < P.K) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (n + u < P.N) P.dW[(size_t)(n + u) * P.K + k] += acc[tn][tk][u];
                }
            }
        if (do_bias && (lane & 15) == 0) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int n = t.n0 + (wn * TN + tn) * 16 + 4 * (lane >> 4) + u;
                    if (n < P.N) P.dbias[n] += accb[tn][u];
                }
        }
        return;
    }
    float* out = scratch + t.out_off;                 // this split's partial tile [N][K]
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tk = 0; tk < TK; ++tk) {
            const int n = (wn * TN + tn) * 16 + 4 * (lane >> 4);
            const int k = (wk * TK + tk) * 16 + (lane & 15);
#pragma unroll
            for (int u = 0; u < 4; ++u) out[(size_t)(n + u) * K + k] = acc[tn][tk][u];
        }
    if (do_bias && (lane & 15) == 0) {                // every column of accb holds the row sums: column 0's lanes store them
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int u = 0; u < 4; ++u) out[(size_t)N * K + (wn * TN + tn) * 16 + 4 * (lane >> 4) + u] = accb[tn][u];
    }
}

constexpr int DTW_NB = 2;
__global__ __launch_bounds__(512, 1) void dense_tile_wgrad_kernel(const DtwProd* __restrict__ prods, const DtwPlane* __restrict__ planes,
                                                                  const DtwTile* __restrict__ tiles, float* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DtwTile t = tiles[blockIdx.x];              // block-uniform: scalar loads
    const DtwProd P = prods[t.prod];
    switch (t.cls) {
        case 0: dtw_body<128, 256, DTW_NB>(P, planes, t, scratch, smem); break;
        case 1: dtw_body<256, 128, DTW_NB>(P, planes, t, scratch, smem); break;
        default: dtw_body<256, 64, DTW_NB>(P, planes, t, scratch, smem); break;
    }
}

// dW[n0 + n][k0 + k] += sum over the tile's splits, in split order; 256 threads x 4 floats per block
__global__ __launch_bounds__(256) void dtw_reduce_kernel(const DtwProd* __restrict__ prods, const DtwRed* __restrict__ red, int nred,
                                                         const float* __restrict__ scratch) {
    int e = 0;
    for (int i = 1; i < nred; ++i)
        if ((int)blockIdx.x >= red[i].first) e = i;
    const DtwRed r = red[e];
    const int bi = (int)blockIdx.x - r.first, nmain = (r.N * r.K + 1023) / 1024;
    if (bi >= nmain) {                                // the tile's column sums of dOut -> dbias
        const int n = (int)threadIdx.x;
        if (!r.bias || n >= r.liveN) return;
        float s = 0.f;
        for (int i = 0; i < r.nparts; ++i) s += scratch[r.parts_off + (size_t)i * r.part_stride + r.N * r.K + n];
        prods[r.prod].dbias[r.n0 + n] += s;
        return;
    }
    const int idx = bi * 1024 + (int)threadIdx.x * 4;      // element of the [N][K] tile
    if (idx >= r.N * r.K) return;
    const float* p = scratch + r.parts_off + idx;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < r.nparts; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)i * r.part_stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int n = idx / r.K, k = idx - n * r.K;
    if (n >= r.liveN || k >= r.liveK) return;         // (the tile reaches past [Npad][K]: whole 16-column planes, so whole float4s)
    float* o = prods[r.prod].dW + (size_t)(r.n0 + n) * prods[r.prod].K + r.k0 + k;
    float4 c = *reinterpret_cast<float4*>(o);
    c.x += s.x; c.y += s.y; c.z += s.z; c.w += s.w;
    *reinterpret_cast<float4*>(o) = c;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
#define DTW_MAXPLANES 1024       // 16-column planes of A over the whole group (K <= 16384 for a single product)
#define DTW_MAXTILES 4096
#define DTW_MAXRED 1024
namespace {
struct DtwLayout { size_t prods, planes, tiles, red, total; };
DtwLayout dtw_layout(int n) {
    DtwLayout L;
    L.prods = 0;
    L.planes = L.prods + (size_t)DTW_MAXP * sizeof(DtwProd);
    L.tiles = L.planes + (size_t)DTW_MAXPLANES * sizeof(DtwPlane);
    L.red = L.tiles + (size_t)DTW_MAXTILES * sizeof(DtwTile);
    L.total = L.red + (size_t)DTW_MAXRED * sizeof(DtwRed);
    (void)n;
    return L;
}
}  // namespace

extern "C" long sehip_wgrad_dense_group_bytes(int n) { return (long)dtw_layout(n).total; }

// info[0] = 1: the group qualifies and dev_buf holds its tables (info[1] = workgroups, info[2] = reduction workgroups, info[3] =
// (product, tile) entries of the reduction, info[4..5] = scratch floats (low / high 31 bits)); info[0] = 0: it does not (no error: the
// caller keeps its other path).  Synchronous (reads the products' chunk tables back): once per binding, never inside a capture.
// What qualifies: dense rows (J = 1, one row per frame), up to two bf16 sources whose 16-column planes are contiguous pieces of a
// source row at ANY frame offset (zero outside the source's valid frame range: the taps of Demucs' 1-D convolutions on their
// [B][T][C] / quad-view tensors, src/model/demucs.py:386-413, :191), sources and dOut with their own frames per utterance, one dense
// run of destination columns.  The tile is 128 x 256 (256 x 128 / 256 x 64 for the shapes that divide so); a tile may reach past
// [Npad][K] (planes that do not exist are not loaded).
extern "C" int sehip_wgrad_dense_group_prepare(const sehip_gemm_desc* descs, int n, void* dev_buf, long dev_bytes, int* info) {
    SEHIP_REQUIRE(descs && dev_buf && info, "wgrad_dense_group_prepare: null argument");
    info[0] = 0;
    static const bool off = getenv("SEHIP_NO_DENSE_GROUP") != nullptr;
    if (off || n < 1 || n > DTW_MAXP) return 0;
    const DtwLayout L = dtw_layout(n);
    SEHIP_REQUIRE(dev_bytes >= (long)L.total, "wgrad_dense_group_prepare: dev_buf has %ld bytes, needs %ld", dev_bytes, (long)L.total);
    std::vector<DtwProd> prods(n);
    std::vector<DtwPlane> planes;
    struct Shape { int cls, N, K; };
    std::vector<Shape> shp(n);
    const long lim = (1L << 31) - (1L << 20);          // byte offsets stay below the out-of-range marker
    for (int p = 0; p < n; ++p) {
        const sehip_gemm_desc& d = descs[p];
        if (!d.dW || d.cv_nf > 0 || d.cv2_nkt > 0 || d.J != 1 || d.tmul > 1 || (d.N & 3) || (d.Npad & 15) || (d.K & 15)) return 0;
        if (d.dst[1].ptr || d.dst[0].is_f32 || d.dst[0].F != 1 || d.dst[0].fadd || d.dst[0].tmul > 1) return 0;
        if (d.TT < 1 || d.M % d.TT) return 0;
        const int B = d.M / d.TT;
        if ((long)B * d.dst[0].T * d.dst[0].C * 2 >= lim) return 0;
        if (d.dst[0].toff < 0 || d.dst[0].toff + d.TT > d.dst[0].T) return 0;
        int cls;
        if (d.Npad % 128 == 0 && d.K % 256 == 0) cls = 0;
        else if (d.Npad % 256 == 0 && d.K % 128 == 0) cls = 1;
        else if (d.Npad % 256 == 0 && d.K == 64) cls = 2;
        else cls = 0;                                  // 128 x 256 tiles that reach past the edges
        for (int s = 0; s < 2; ++s)
            if (d.src[s].ptr && (long)B * d.src[s].T * d.src[s].F * d.src[s].C * 2 >= lim) return 0;
        // destination columns: one dense run (then padding)
        std::vector<sehip_nchunk> nt(d.Npad / 4);
        std::vector<sehip_kchunk> kt(d.K / 8);
        if (hipMemcpy(nt.data(), d.ntab, nt.size() * sizeof(sehip_nchunk), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(kt.data(), d.ktab, kt.size() * sizeof(sehip_kchunk), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        for (size_t i = 0; i < nt.size(); ++i) {
            if ((int)i < d.N / 4) {
                if (nt[i].dst != 0 || nt[i].nvalid != 4 || nt[i].coff != nt[0].coff + 4 * (int)i) return 0;
            } else if (nt[i].nvalid != 0) return 0;
        }
        // (the planes are read 16 columns at a time: the padded width must exist in the row)
        if ((nt[0].coff & 7) || (d.dst[0].C & 7) || nt[0].coff + d.Npad > d.dst[0].C) return 0;
        DtwProd& P = prods[p];
        P = DtwProd{};
        P.g = reinterpret_cast<const bf16_raw*>(d.dst[0].ptr);
        P.gstride = d.dst[0].C; P.gcol0 = nt[0].coff;
        P.B = B; P.Tg = d.dst[0].T; P.gtoff = d.dst[0].toff;
        P.gbytes = (unsigned)((long)B * d.dst[0].T * d.dst[0].C * 2);
        for (int s = 0; s < 2; ++s) {
            P.x[s] = reinterpret_cast<const bf16_raw*>(d.src[s].ptr);
            P.xstride[s] = d.src[s].ptr ? d.src[s].F * d.src[s].C : 0;
            P.Tx[s] = d.src[s].T; P.tlo[s] = d.src[s].tlo; P.thi[s] = d.src[s].thi;
            P.xbytes[s] = d.src[s].ptr ? (unsigned)((long)B * d.src[s].T * d.src[s].F * d.src[s].C * 2) : 0u;
        }
        P.dW = d.dW; P.dbias = d.dbias; P.T = d.TT; P.M = d.M; P.N = d.Npad; P.K = d.K;
        P.plane0 = (int)planes.size();
        for (int pl = 0; pl < d.K / 16; ++pl) {
            const sehip_kchunk a = kt[2 * pl], b = kt[2 * pl + 1];
            if (a.src < 0 && b.src < 0) {              // K padding: zero columns
                planes.push_back(DtwPlane{0, 0, 0, 0});
                continue;
            }
            if (a.src < 0 || a.src > 1 || b.src != a.src || b.toff != a.toff || b.fadd != a.fadd + 8 || !d.src[a.src].ptr) return 0;
            const int foff = a.toff >> 16, roff = (int)(short)(a.toff & 0xffff);
            if (foff < -64 || foff > 64 || roff < 0 || roff >= d.src[a.src].F) return 0;
            const int stride = d.src[a.src].F * d.src[a.src].C;
            const int delta = a.fadd - foff * stride;                     // the column part of the element delta
            if (delta < 0 || delta + 16 > stride || (delta & 7)) return 0;
            planes.push_back(DtwPlane{a.src, delta, -foff, 1});
        }
        shp[p] = {cls, cls == 0 ? 128 : 256, cls == 0 ? 256 : cls == 1 ? 128 : 64};
    }
    if (planes.size() > DTW_MAXPLANES) return 0;
    // tiles: every workgroup streams about the same number of rows
    long row_tiles = 0;
    for (int p = 0; p < n; ++p)
        row_tiles += (long)((prods[p].N + shp[p].N - 1) / shp[p].N) * ((prods[p].K + shp[p].K - 1) / shp[p].K) * ((prods[p].M + 63) / 64);
    static const int want = getenv("SEHIP_DTW_WGS") ? atoi(getenv("SEHIP_DTW_WGS")) : 248;
    long per = (row_tiles + want - 1) / want;                             // stages per workgroup
    if (per < 4) per = 4;
    std::vector<DtwTile> tiles;
    std::vector<DtwRed> red;
    long out_off = 0;
    int red_first = 0;
    for (int p = 0; p < n; ++p) {
        const int nst = (prods[p].M + 63) / 64, splits = (int)((nst + per - 1) / per), spw = (nst + splits - 1) / splits;
        const int tn = shp[p].N, tk = shp[p].K;
        for (int n0 = 0; n0 < prods[p].N; n0 += tn)
            for (int k0 = 0; k0 < prods[p].K; k0 += tk) {
                const int bias = (k0 == 0 && prods[p].dbias) ? 1 : 0;      // the first k-tile of every n-tile also sums dOut's columns
                if (splits == 1) {                     // one workgroup streams all rows of this tile: it adds to dW itself
                    tiles.push_back(DtwTile{p, n0, k0, 0, nst, -1, shp[p].cls, bias});
                    continue;
                }
                DtwRed r{};
                r.first = red_first; r.prod = p; r.n0 = n0; r.k0 = k0; r.N = tn; r.K = tk; r.parts_off = (int)out_off; r.part_stride = tn * tk + tn;
                r.bias = bias;
                r.liveN = prods[p].N - n0 < tn ? prods[p].N - n0 : tn;
                r.liveK = prods[p].K - k0 < tk ? prods[p].K - k0 : tk;
                int np = 0;
                for (int s = 0; s < splits; ++s) {
                    const int sb = s * spw, se = sb + spw < nst ? sb + spw : nst;
                    if (sb >= se) break;
                    tiles.push_back(DtwTile{p, n0, k0, sb, se, (int)out_off, shp[p].cls, bias});
                    out_off += (long)tn * tk + tn;
                    ++np;
                }
                r.nparts = np;
                red.push_back(r);
                red_first += (tn * tk + 1023) / 1024 + bias;
            }
    }
    if (tiles.size() > DTW_MAXTILES || red.size() > DTW_MAXRED || out_off >= (1L << 31)) return 0;
    char* db = reinterpret_cast<char*>(dev_buf);
    hipError_t e = hipMemcpy(db + L.prods, prods.data(), prods.size() * sizeof(DtwProd), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db + L.planes, planes.data(), planes.size() * sizeof(DtwPlane), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db + L.tiles, tiles.data(), tiles.size() * sizeof(DtwTile), hipMemcpyHostToDevice);
    if (e == hipSuccess && !red.empty()) e = hipMemcpy(db + L.red, red.data(), red.size() * sizeof(DtwRed), hipMemcpyHostToDevice);
    if (e != hipSuccess) return sehip_set_error(-2, "wgrad_dense_group_prepare: %s", hipGetErrorString(e));
    info[0] = 1; info[1] = (int)tiles.size(); info[2] = red_first; info[3] = (int)red.size();
    info[4] = (int)(out_off & 0x7fffffff); info[5] = (int)(out_off >> 31);
    return 0;
}

extern "C" int sehip_wgrad_dense_group(const void* dev_buf, int n, const int* info, float* scratch, void* stream) {
    SEHIP_REQUIRE(dev_buf && info && info[0] == 1 && (scratch || info[2] == 0), "wgrad_dense_group: not prepared (info[0] != 1) or no scratch");
    const DtwLayout L = dtw_layout(n);
    const char* db = reinterpret_cast<const char*>(dev_buf);
    hipStream_t st = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {

        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_tile_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    // SEHIP_DTW_LDS_ALL=1 (tools/dev/det_diff.py): the CU's whole LDS, i.e. no LDS-using workgroup of another kernel beside this one --
    // the setting under which the Demucs step's deterministic schedule was bit-stable WITH the second stream
    static const bool lds_all = getenv("SEHIP_DTW_LDS_ALL") != nullptr;
    const size_t lds = lds_all ? (size_t)160 * 1024 : (size_t)DTW_NB * (128 + 256) * 64 * 2;
    sehip_note_kernel("dense_tile_wgrad_kernel");
    dense_tile_wgrad_kernel<<<info[1], 512, lds, st>>>(reinterpret_cast<const DtwProd*>(db + L.prods), reinterpret_cast<const DtwPlane*>(db + L.planes),
                                                       reinterpret_cast<const DtwTile*>(db + L.tiles), scratch);
    if (info[2] > 0)
        dtw_reduce_kernel<<<info[2], 256, 0, st>>>(reinterpret_cast<const DtwProd*>(db + L.prods), reinterpret_cast<const DtwRed*>(db + L.red), info[3], scratch);
    SEHIP_CHECK_LAUNCH("wgrad_dense_group");
    return 0;
}
