// Recurrent part of NavieComplexLSTM (src/model/dccrn.py:264-302): the four nn.LSTM passes of one complex layer
// (real_lstm / imag_lstm applied to the real and the imaginary input) as ONE persistent launch.
//
// The input products x @ W_ih^T + b_ih + b_hh are done beforehand by the implicit-GEMM engine ("pre" gates);
// this kernel only walks the T sequential steps:   gates = pre[t] + h[t-1] @ W_hh^T ; (i,f,g,o) ; c ; h.
// combo = part*2 + lstm  (part 0 = real input, 1 = imag input; lstm 0 = real_lstm, 1 = imag_lstm).
// One workgroup = one combo x 4 batch rows; wave w owns hidden units [16w, 16w+16) of all four gates.  The step is a
// chain of dependent latencies (LDS read, MFMA, ten transcendentals per (unit, batch) pair, LDS write, barrier), so the
// kernel is laid out for the shortest chain, not for MFMA efficiency: the 16 columns of the 16x16x32 bf16 MFMAs
// (D[unit][column], W_hh fragments resident in registers for the whole sequence) carry the four batch rows four times
// over, and lane (m, ug) keeps only row 4 ug + (m >> 2) of its result quad -- ONE (unit, batch) pair per lane, all
// four gates lane-local, a quarter of the exp/rcp work per wave of the 16-row tile this replaced (0.9 us per step).
// h[t] goes back through a double-buffered LDS tile: one barrier per step.  Latency-bound by construction (hidden 64).
#include "common.h"

// The kernels are written for a hidden size H that is a multiple of 32 (H / 16 waves of 64 lanes, 4 H threads: one (unit, batch row)
// pair per lane); instantiated for 32, 64 (rnn_units = 128, the reference default, src/model/dccrn.py:13), 96 and 128 (rnn_units = 256,
// the DCCRN paper's complex LSTM).
#define NBT 4    // batch rows per workgroup
#define PD 8     // prefetch distance of the per-step inputs in time steps (2: 4.74 ms per step, 4: 4.68, 8: 4.63, 16: 4.65)

// Workgroup barrier that only waits for this wave's LDS traffic: the per-step global stores / prefetch loads stay in
// flight across it (__syncthreads() would also drain vmcnt and expose a full HBM round trip on every time step).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// v_exp_f32 + v_rcp_f32 (1 ulp) instead of the IEEE division sequence: the cell update sits on the serial path
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

// The lane's own element of an MFMA result quad (rs is lane-constant: three v_cndmask per call)
__device__ __forceinline__ float quad_pick(const f32x4& a, int rs) {
    return rs == 0 ? a[0] : rs == 1 ? a[1] : rs == 2 ? a[2] : a[3];
}

// Steps [t0, t1) of the sequence.  A chunk that does not start at 0 resumes from the h / c records the previous chunk wrote
// (the two stacked layers are pipelined chunk by chunk on two streams, sehip/plan.py).
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_fwd_kernel(const float* __restrict__ pre0, const float* __restrict__ pre1,
                                                       const bf16_raw* __restrict__ whh, int B, int T, int t0, int t1,
                                                       bf16_raw* __restrict__ hout, bf16_raw* __restrict__ gates,
                                                       float* __restrict__ cout, int real) {
    constexpr int G4 = 4 * H, NT = 4 * H, HP = H + 8, KS = H / 32;     // gates per LSTM, threads, LDS pitch of the h tile, k-steps
    __shared__ __attribute__((aligned(16))) bf16_raw hbuf[2][NBT * HP];
    // real != 0: ONE plain nn.LSTM (DCCRN(use_clstm=False), src/model/dccrn.py:98-106): a single "combo", pre rows of 4 H gates
    const int combo = real ? 0 : (blockIdx.x & 3), tile = real ? blockIdx.x : (blockIdx.x >> 2);
    const int PS = real ? G4 : 2 * G4;          // row pitch of the pre-gates
    const int part = combo >> 1, lstm = combo & 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, ug = lane >> 4;
    const int bl = m & (NBT - 1), rs = m >> 2;   // MFMA column m carries batch row bl; this lane owns row 4 ug + rs of the quad
    const int unit = 16 * w + 4 * ug + rs;
    const int b = tile * NBT + bl;
    const bool bvalid = b < B;
    const int bc = bvalid ? b : B - 1;
    const float* pre = (part ? pre1 : pre0) + ((size_t)bc * T) * PS + lstm * G4 + unit;
    const size_t obase = ((size_t)combo * B + bc) * T;
    const int ntiles = real ? gridDim.x : (gridDim.x >> 2);
    const size_t rbase = (size_t)(combo * ntiles + tile) * T;  // records private to the backward kernel: [combo][tile][t][thread]

    // W_hh fragments: gate g, k-step s: rows g*H + 16w + (lane&15), cols 32 s + 8 (lane>>4) ..
    bf16x8 wf[4][KS];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            wf[g][s] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const uint4*>(whh + ((size_t)lstm * G4 + g * H + 16 * w + m) * H + 32 * s + 8 * ug));

    for (int i = threadIdx.x; i < NBT * HP; i += NT) hbuf[0][i] = 0;
    float c = 0.f;
    // The pre-gates of step t + PD are requested while step t runs (PD rotating register sets, the loop is unrolled by PD so
    // that they stay in registers): one step of dependent work (0.4 us) does not cover a memory round trip, and beside the
    // weight gradients of the second stream, which saturate HBM, a round trip takes several steps.
    float pf[PD][4];
#pragma unroll
    for (int k = 0; k < PD; ++k)
        if (t0 + k < t1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) pf[k][g] = pre[(size_t)(t0 + k) * PS + g * H];
        }
    __syncthreads();
    if (t0 > 0) {  // resume: h(t0-1) from the output, c(t0-1) from the cell-state record
        hbuf[0][bl * HP + unit] = hout[(obase + t0 - 1) * H + unit];
        c = cout[(rbase + t0 - 1) * NT + threadIdx.x];
        __syncthreads();
    }
    int cur = 0;
    auto step = [&](int t, float (&pq)[4]) {
        const float p0 = pq[0], p1 = pq[1], p2 = pq[2], p3 = pq[3];
        if (t + PD < t1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) pq[g] = pre[(size_t)(t + PD) * PS + g * H];
        }
        bf16x8 hf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s)
            hf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&hbuf[cur][bl * HP + 32 * s + 8 * ug]));
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][0], hf[0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < KS; ++s) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][s], hf[s], acc[g], 0, 0, 0);
        }
        const float gi = sigmoidf_(quad_pick(acc[0], rs) + p0);
        const float gf = sigmoidf_(quad_pick(acc[1], rs) + p1);
        const float gg = tanhf_(quad_pick(acc[2], rs) + p2);
        const float go = sigmoidf_(quad_pick(acc[3], rs) + p3);
        c = gf * c + gi * gg;
        const bf16_raw hb = f2bf(go * tanhf_(c));
        hbuf[cur ^ 1][bl * HP + unit] = hb;
        if (bvalid) hout[(obase + t) * H + unit] = hb;
        const size_t rec = (rbase + t) * NT + threadIdx.x;    // one contiguous 1 KB / 2 KB run per workgroup and step (H = 64)
        cout[rec] = c;
        *reinterpret_cast<uint2*>(gates + rec * 4) = make_uint2(pack_bf2(gi, gf), pack_bf2(gg, go));
        lds_barrier();
        cur ^= 1;
    };
    for (int t = t0; t < t1; t += PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k)
            if (t + k < t1) step(t + k, pf[k]);
    }
}

// Backward through time.  dh_a / dh_b are the gradients w.r.t. the layer's two outputs
//   out_r = h[r,real] - h[i,imag]   (dh_a),    out_i = h[i,real] + h[r,imag]   (dh_b)
// combo 0 (r,real): +dh_a   combo 1 (r,imag): +dh_b   combo 2 (i,real): +dh_b   combo 3 (i,imag): -dh_a
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_bwd_kernel(const bf16_raw* __restrict__ dh_a, const bf16_raw* __restrict__ dh_b,
                                                       const bf16_raw* __restrict__ whhT, const bf16_raw* __restrict__ gates,
                                                       const float* __restrict__ cst, int B, int T, int t0, int t1,
                                                       float* __restrict__ state,
                                                       bf16_raw* __restrict__ dpre0, bf16_raw* __restrict__ dpre1, int real) {
    constexpr int G4 = 4 * H, NT = 4 * H, DGP = 4 * H + 8, KS = G4 / 32;   // LDS pitch of the dgate tile; k-steps over the 4 H gate gradients
    __shared__ __attribute__((aligned(16))) bf16_raw dgbuf[2][NBT * DGP];
    const int combo = real ? 0 : (blockIdx.x & 3), tile = real ? blockIdx.x : (blockIdx.x >> 2);     // (real: see lstm_fwd_kernel)
    const int PS = real ? G4 : 2 * G4;
    const int part = combo >> 1, lstm = combo & 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, ug = lane >> 4;
    const int bl = m & (NBT - 1), rs = m >> 2;
    const int unit = 16 * w + 4 * ug + rs;
    const int b = tile * NBT + bl;
    const int bc = b < B ? b : B - 1;
    const bf16_raw* dh = (combo == 0 || combo == 3) ? dh_a : dh_b;
    const float sign = combo == 3 ? -1.f : 1.f;
    const bf16_raw* dhp = dh + ((size_t)bc * T) * H + unit;
    // the step's gate gradients leave through LDS: H threads (wave w for H = 64) store one batch row of the tile as one 8 H-byte run
    const int sr = threadIdx.x / H, sc = (threadIdx.x % H) * 4;
    const int srow = tile * NBT + sr;
    bf16_raw* dpre = (part ? dpre1 : dpre0) + ((size_t)(srow < B ? srow : 0) * T) * PS + lstm * G4 + sc;
    const bool svalid = srow < B;

    // W_hh^T fragments: rows k = 16w + (lane&15), reduction index n = 32 s + 8 (lane>>4) ..
    bf16x8 wf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        wf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(whhT + ((size_t)lstm * H + 16 * w + m) * G4 + 32 * s + 8 * ug));

    // steps t1-1 ... t0; a chunk that does not end at T resumes from the (dc, recurrent dh) the later chunk left in `state`
    float dc = 0.f, dhrec = 0.f;
    float* stp = state ? state + ((size_t)blockIdx.x * NT + threadIdx.x) * 2 : nullptr;
    if (t1 < T) {
        const float2 a = *reinterpret_cast<const float2*>(stp);
        dc = a.x; dhrec = a.y;
    }
    // software pipeline: the inputs of step t - PD (gates, dh, c[t], c[t-1]) are requested while step t runs; PD register
    // sets, the loop is unrolled by PD (see lstm_fwd_kernel)
    struct StepIn { uint2 g; bf16_raw dh; float c, cp; };
    const int ntiles = real ? gridDim.x : (gridDim.x >> 2);
    const size_t rbase = (size_t)(combo * ntiles + tile) * T;  // records written by lstm_fwd_kernel
    const float* cbase = cst + rbase * NT + threadIdx.x;
    auto load_step = [&](int t) {
        StepIn v;
        v.g = *reinterpret_cast<const uint2*>(gates + ((rbase + t) * NT + threadIdx.x) * 4);
        v.dh = dhp[(size_t)t * H];
        v.c = cbase[(size_t)t * NT];
        v.cp = t > 0 ? cbase[(size_t)(t - 1) * NT] : 0.f;
        return v;
    };
    StepIn ring[PD];
#pragma unroll
    for (int k = 0; k < PD; ++k)
        if (t1 - 1 - k >= t0) ring[k] = load_step(t1 - 1 - k);
    int cur = 0;
    auto step = [&](int t, StepIn& slot) {
        const StepIn in = slot;
        if (t - PD >= t0) slot = load_step(t - PD);
        const float gi = bf2f(in.g.x & 0xffff), gf = bf2f(in.g.x >> 16), gg = bf2f(in.g.y & 0xffff), go = bf2f(in.g.y >> 16);
        const float dhv = sign * bf2f(in.dh) + dhrec;
        const float tc = tanhf_(in.c);
        const float d_o = dhv * tc;
        const float dcv = dc + dhv * go * (1.f - tc * tc);
        const bf16_raw di = f2bf(dcv * gg * gi * (1.f - gi));
        const bf16_raw df = f2bf(dcv * in.cp * gf * (1.f - gf));
        const bf16_raw dg = f2bf(dcv * gi * (1.f - gg * gg));
        const bf16_raw dob = f2bf(d_o * go * (1.f - go));
        dc = dcv * gf;
        bf16_raw* lb = &dgbuf[cur][bl * DGP + unit];
        lb[0] = di; lb[H] = df; lb[2 * H] = dg; lb[3 * H] = dob;
        lds_barrier();
        f32x4 r0 = (f32x4){0.f, 0.f, 0.f, 0.f}, r1 = r0;   // two accumulation chains of four
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            const bf16x8 g0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&dgbuf[cur][bl * DGP + 32 * s + 8 * ug]));
            const bf16x8 g1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&dgbuf[cur][bl * DGP + 32 * s + 32 + 8 * ug]));
            r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], g0, r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s + 1], g1, r1, 0, 0, 0);
        }
        {
            const uint2 row = *reinterpret_cast<const uint2*>(&dgbuf[cur][sr * DGP + sc]);
            if (svalid) *reinterpret_cast<uint2*>(dpre + (size_t)t * PS) = row;
        }
        dhrec = quad_pick(r0, rs) + quad_pick(r1, rs);
        cur ^= 1;
    };
    for (int t = t1 - 1; t >= t0; t -= PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k)
            if (t - k >= t0) step(t - k, ring[k]);
    }
    if (t0 > 0 && stp) *reinterpret_cast<float2*>(stp) = make_float2(dc, dhrec);
}

static int lstm_fwd_launch(const float* pre0, const float* pre1, const void* whh, int B, int T, int hidden, int t0, int t1, void* h,
                           void* gates, float* c, int real, void* stream) {
    SEHIP_REQUIRE(hidden == 32 || hidden == 64 || hidden == 96 || hidden == 128, "lstm_fwd: hidden size 32, 64, 96 or 128, got %d", hidden);
    SEHIP_REQUIRE(B > 0 && T > 0, "lstm_fwd: empty input");
    SEHIP_REQUIRE(0 <= t0 && t0 < t1 && t1 <= T, "lstm_fwd: bad step range [%d, %d) of %d", t0, t1, T);
    const int grid = (real ? 1 : 4) * cdiv(B, NBT);
#define LSTM_FWD(HH)                                                                                                              \
    lstm_fwd_kernel<HH><<<grid, 4 * HH, 0, (hipStream_t)stream>>>(pre0, pre1, (const bf16_raw*)whh, B, T, t0, t1, (bf16_raw*)h, \
                                                                 (bf16_raw*)gates, c, real)
    if (hidden == 32) LSTM_FWD(32);
    else if (hidden == 64) LSTM_FWD(64);
    else if (hidden == 96) LSTM_FWD(96);
    else LSTM_FWD(128);
#undef LSTM_FWD
    SEHIP_CHECK_LAUNCH("lstm_fwd");
    return 0;
}

static int lstm_bwd_launch(const void* dh_a, const void* dh_b, const void* whhT, const void* gates, const float* c, int B, int T,
                           int hidden, int t0, int t1, float* state, void* dpre0, void* dpre1, int real, void* stream) {
    SEHIP_REQUIRE(hidden == 32 || hidden == 64 || hidden == 96 || hidden == 128, "lstm_bwd: hidden size 32, 64, 96 or 128, got %d", hidden);
    SEHIP_REQUIRE(B > 0 && T > 0, "lstm_bwd: empty input");
    SEHIP_REQUIRE(0 <= t0 && t0 < t1 && t1 <= T, "lstm_bwd: bad step range [%d, %d) of %d", t0, t1, T);
    SEHIP_REQUIRE(state != nullptr || (t0 == 0 && t1 == T), "lstm_bwd: a partial step range needs the state buffer");
    const int grid = (real ? 1 : 4) * cdiv(B, NBT);
#define LSTM_BWD(HH)                                                                                                                    \
    lstm_bwd_kernel<HH><<<grid, 4 * HH, 0, (hipStream_t)stream>>>((const bf16_raw*)dh_a, (const bf16_raw*)dh_b, (const bf16_raw*)whhT, \
                                                                 (const bf16_raw*)gates, c, B, T, t0, t1, state, (bf16_raw*)dpre0,    \
                                                                 (bf16_raw*)dpre1, real)
    if (hidden == 32) LSTM_BWD(32);
    else if (hidden == 64) LSTM_BWD(64);
    else if (hidden == 96) LSTM_BWD(96);
    else LSTM_BWD(128);
#undef LSTM_BWD
    SEHIP_CHECK_LAUNCH("lstm_bwd");
    return 0;
}

extern "C" int sehip_lstm_fwd_chunk(const float* pre0, const float* pre1, const void* whh, int B, int T, int hidden, int t0,
                                    int t1, void* h, void* gates, float* c, void* stream) {
    return lstm_fwd_launch(pre0, pre1, whh, B, T, hidden, t0, t1, h, gates, c, 0, stream);
}

extern "C" int sehip_lstm_fwd(const float* pre0, const float* pre1, const void* whh, int B, int T, int hidden, void* h,
                              void* gates, float* c, void* stream) {
    return lstm_fwd_launch(pre0, pre1, whh, B, T, hidden, 0, T, h, gates, c, 0, stream);
}

// state: 4 * ceil(B/4) * 4 * hidden * 2 floats carried between chunks (needed unless the chunk is the whole sequence)
extern "C" int sehip_lstm_bwd_chunk(const void* dh_a, const void* dh_b, const void* whhT, const void* gates, const float* c, int B,
                                    int T, int hidden, int t0, int t1, float* state, void* dpre0, void* dpre1, void* stream) {
    return lstm_bwd_launch(dh_a, dh_b, whhT, gates, c, B, T, hidden, t0, t1, state, dpre0, dpre1, 0, stream);
}

extern "C" int sehip_lstm_bwd(const void* dh_a, const void* dh_b, const void* whhT, const void* gates, const float* c, int B,
                              int T, int hidden, void* dpre0, void* dpre1, void* stream) {
    return lstm_bwd_launch(dh_a, dh_b, whhT, gates, c, B, T, hidden, 0, T, nullptr, dpre0, dpre1, 0, stream);
}

// One plain nn.LSTM layer (unidirectional, zero initial state): DCCRN(use_clstm=False), src/model/dccrn.py:98-106 / :184-189.
//   pre [B][T][4 H] fp32 = x @ W_ih^T + b_ih + b_hh;  whh bf16 [4 H][H];  h bf16 [B][T][H];  gates / c: records for sehip_rlstm_bwd,
//   [ceil(B/4)*4][T][4 H] bf16 / [ceil(B/4)*4][T][H] fp32.  Backward: dh bf16 [B][T][H] -> dpre bf16 [B][T][4 H]; whhT bf16 [H][4 H].
extern "C" int sehip_rlstm_fwd(const float* pre, const void* whh, int B, int T, int hidden, void* h, void* gates, float* c, void* stream) {
    return lstm_fwd_launch(pre, pre, whh, B, T, hidden, 0, T, h, gates, c, 1, stream);
}

extern "C" int sehip_rlstm_bwd(const void* dh, const void* whhT, const void* gates, const float* c, int B, int T, int hidden, void* dpre,
                               void* stream) {
    return lstm_bwd_launch(dh, dh, whhT, gates, c, B, T, hidden, 0, T, nullptr, dpre, dpre, 1, stream);
}
