// Recurrent part of NavieComplexLSTM (src/model/dccrn.py:264-302): the four nn.LSTM passes of one complex layer
// (real_lstm / imag_lstm applied to the real and the imaginary input) as ONE persistent launch.
//
// The input products x @ W_ih^T + b_ih + b_hh are done beforehand by the implicit-GEMM engine ("pre" gates);
// this kernel only walks the T sequential steps:   gates = pre[t] + h[t-1] @ W_hh^T ; (i,f,g,o) ; c ; h.
// combo = part*2 + lstm  (part 0 = real input, 1 = imag input; lstm 0 = real_lstm, 1 = imag_lstm).
// One workgroup = one combo x 16 batch rows; wave w owns hidden units [16w, 16w+16) of all four gates, so after
// the 16x16x32 bf16 MFMAs (D[unit][batch], W_hh fragments resident in registers for the whole sequence) a lane
// holds i,f,g,o of the same (batch, unit) and the cell update is lane-local.  h[t] goes back through a
// double-buffered 2 KB LDS tile: one barrier per step.  Latency-bound by construction (hidden size 64).
#include "common.h"

#define H 64
#define G4 256
#define HP 72    // LDS pitch of the h tile (bf16 elements)
#define DGP 264  // LDS pitch of the dgate tile

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

// Steps [t0, t1) of the sequence.  A chunk that does not start at 0 resumes from the h / c records the previous chunk wrote
// (the two stacked layers are pipelined chunk by chunk on two streams, sehip/plan.py).
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* __restrict__ pre0, const float* __restrict__ pre1,
                                                       const bf16_raw* __restrict__ whh, int B, int T, int t0, int t1,
                                                       bf16_raw* __restrict__ hout, bf16_raw* __restrict__ gates,
                                                       float* __restrict__ cout) {
    __shared__ __attribute__((aligned(16))) bf16_raw hbuf[2][16 * HP];
    const int combo = blockIdx.x & 3, tile = blockIdx.x >> 2;
    const int part = combo >> 1, lstm = combo & 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, ug = lane >> 4;
    const int b = tile * 16 + m;
    const bool bvalid = b < B;
    const int bc = bvalid ? b : B - 1;
    const float* pre = (part ? pre1 : pre0) + ((size_t)bc * T) * (2 * G4) + lstm * G4 + 16 * w + 4 * ug;
    const size_t obase = ((size_t)combo * B + bc) * T;
    const int ntiles = gridDim.x >> 2;

    // W_hh fragments: gate g, k-step s: rows g*64 + 16w + (lane&15), cols 32 s + 8 (lane>>4) ..
    bf16x8 wf[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            wf[g][s] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const uint4*>(whh + ((size_t)lstm * G4 + g * H + 16 * w + m) * H + 32 * s + 8 * ug));

    for (int i = threadIdx.x; i < 16 * HP; i += 256) hbuf[0][i] = 0;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    float4 pn[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) pn[g] = *reinterpret_cast<const float4*>(pre + (size_t)t0 * (2 * G4) + g * H);
    __syncthreads();
    if (t0 > 0) {  // resume: h(t0-1) of this lane's four units from the output, c(t0-1) from the cell-state record
        if (bvalid) *reinterpret_cast<uint2*>(&hbuf[0][m * HP + 16 * w + 4 * ug]) =
            *reinterpret_cast<const uint2*>(hout + (obase + t0 - 1) * H + 16 * w + 4 * ug);
        const size_t rec = ((size_t)(combo * ntiles + tile) * T + t0 - 1);
        const float4 cv = *reinterpret_cast<const float4*>(cout + rec * 1024 + w * 256 + lane * 4);
        c[0] = cv.x; c[1] = cv.y; c[2] = cv.z; c[3] = cv.w;
        __syncthreads();
    }

    // The pre-gates of step t + 2 are requested while step t runs (two rotating register sets, the loop is unrolled by two so
    // that they stay in registers): one step (~0.5 us of dependent work) does not cover an HBM round trip under load, and the
    // eight workgroups of this kernel have nothing else to hide it with.
    float4 pm[4];
    if (t0 + 1 < t1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) pm[g] = *reinterpret_cast<const float4*>(pre + (size_t)(t0 + 1) * (2 * G4) + g * H);
    }
    int cur = 0;
    auto step = [&](int t, float4 (&pq)[4]) {
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){pq[g].x, pq[g].y, pq[g].z, pq[g].w};
        if (t + 2 < t1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) pq[g] = *reinterpret_cast<const float4*>(pre + (size_t)(t + 2) * (2 * G4) + g * H);
        }
        bf16x8 hf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
            hf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&hbuf[cur][m * HP + 32 * s + 8 * ug]));
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][s], hf[s], acc[g], 0, 0, 0);
        float hv[4], gi[4], gf[4], gg[4], go[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gi[r] = sigmoidf_(acc[0][r]);
            gf[r] = sigmoidf_(acc[1][r]);
            gg[r] = tanhf_(acc[2][r]);
            go[r] = sigmoidf_(acc[3][r]);
            c[r] = gf[r] * c[r] + gi[r] * gg[r];
            hv[r] = go[r] * tanhf_(c[r]);
        }
        const uint2 hp = make_uint2(pack_bf2(hv[0], hv[1]), pack_bf2(hv[2], hv[3]));
        *reinterpret_cast<uint2*>(&hbuf[cur ^ 1][m * HP + 16 * w + 4 * ug]) = hp;
        if (bvalid) {
            const size_t o = obase + t;
            *reinterpret_cast<uint2*>(hout + o * H + 16 * w + 4 * ug) = hp;
        }
        {   // gates / cell state are private to lstm_bwd_kernel: stored tile-major so that every store instruction of a
            // wave is one contiguous 512 B / 1 KB run ([combo][tile][t][wave][gate][lane]) instead of 16 partial lines
            const size_t rec = ((size_t)(combo * ntiles + tile) * T + t);
            *reinterpret_cast<float4*>(cout + rec * 1024 + w * 256 + lane * 4) = make_float4(c[0], c[1], c[2], c[3]);
            bf16_raw* gp = gates + rec * 4096 + w * 1024 + lane * 4;
            *reinterpret_cast<uint2*>(gp) = make_uint2(pack_bf2(gi[0], gi[1]), pack_bf2(gi[2], gi[3]));
            *reinterpret_cast<uint2*>(gp + 256) = make_uint2(pack_bf2(gf[0], gf[1]), pack_bf2(gf[2], gf[3]));
            *reinterpret_cast<uint2*>(gp + 512) = make_uint2(pack_bf2(gg[0], gg[1]), pack_bf2(gg[2], gg[3]));
            *reinterpret_cast<uint2*>(gp + 768) = make_uint2(pack_bf2(go[0], go[1]), pack_bf2(go[2], go[3]));
        }
        lds_barrier();
        cur ^= 1;
    };
    for (int t = t0; t < t1; t += 2) {
        step(t, pn);
        if (t + 1 < t1) step(t + 1, pm);
    }
}

// Backward through time.  dh_a / dh_b are the gradients w.r.t. the layer's two outputs
//   out_r = h[r,real] - h[i,imag]   (dh_a),    out_i = h[i,real] + h[r,imag]   (dh_b)
// combo 0 (r,real): +dh_a   combo 1 (r,imag): +dh_b   combo 2 (i,real): +dh_b   combo 3 (i,imag): -dh_a
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const bf16_raw* __restrict__ dh_a, const bf16_raw* __restrict__ dh_b,
                                                       const bf16_raw* __restrict__ whhT, const bf16_raw* __restrict__ gates,
                                                       const float* __restrict__ cst, int B, int T, int t0, int t1,
                                                       float* __restrict__ state,
                                                       bf16_raw* __restrict__ dpre0, bf16_raw* __restrict__ dpre1) {
    __shared__ __attribute__((aligned(16))) bf16_raw dgbuf[2][16 * DGP];
    const int combo = blockIdx.x & 3, tile = blockIdx.x >> 2;
    const int part = combo >> 1, lstm = combo & 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, ug = lane >> 4;
    const int b = tile * 16 + m;
    const bool bvalid = b < B;
    const int bc = bvalid ? b : B - 1;
    const bf16_raw* dh = (combo == 0 || combo == 3) ? dh_a : dh_b;
    const float sign = combo == 3 ? -1.f : 1.f;
    const size_t sbase = ((size_t)combo * B + bc) * T;
    const int uo = 16 * w + 4 * ug;
    bf16_raw* dpre = (part ? dpre1 : dpre0) + ((size_t)bc * T) * (2 * G4) + lstm * G4 + uo;
    const bf16_raw* dhp = dh + ((size_t)bc * T) * H + uo;

    // W_hh^T fragments: rows k = 16w + (lane&15), reduction index n = 32 s + 8 (lane>>4) ..
    bf16x8 wf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
        wf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(whhT + ((size_t)lstm * H + 16 * w + m) * G4 + 32 * s + 8 * ug));

    // steps t1-1 ... t0; a chunk that does not end at T resumes from the (dc, recurrent dh) the later chunk left in `state`
    float dc[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 dhrec = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* stp = state ? state + ((size_t)blockIdx.x * 256 + threadIdx.x) * 8 : nullptr;
    if (t1 < T) {
        const float4 a = *reinterpret_cast<const float4*>(stp), bq = *reinterpret_cast<const float4*>(stp + 4);
        dc[0] = a.x; dc[1] = a.y; dc[2] = a.z; dc[3] = a.w;
        dhrec = (f32x4){bq.x, bq.y, bq.z, bq.w};
    }
    // software pipeline: the (gates, dh) and the cell state of step t-2 are requested while step t runs
    struct StepIn { uint2 gi, gf, gg, go, dh; };
    const int ntiles = gridDim.x >> 2;
    const size_t rbase = (size_t)(combo * ntiles + tile) * T;  // tile-major records written by lstm_fwd_kernel
    const float* cbase = cst + w * 256 + lane * 4;
    auto load_step = [&](int t) {
        StepIn v;
        const bf16_raw* gp = gates + (rbase + t) * 4096 + w * 1024 + lane * 4;
        v.gi = *reinterpret_cast<const uint2*>(gp);
        v.gf = *reinterpret_cast<const uint2*>(gp + 256);
        v.gg = *reinterpret_cast<const uint2*>(gp + 512);
        v.go = *reinterpret_cast<const uint2*>(gp + 768);
        v.dh = *reinterpret_cast<const uint2*>(dhp + (size_t)t * H);
        return v;
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 c_t = *reinterpret_cast<const float4*>(cbase + (rbase + t1 - 1) * 1024);
    float4 c_m1 = t1 > 1 ? *reinterpret_cast<const float4*>(cbase + (rbase + t1 - 2) * 1024) : zero4;
    StepIn in = load_step(t1 - 1);
    StepIn nx1 = in;                       // two steps ahead (one step of dependent work does not cover an HBM round trip)
    if (t1 - 2 >= t0) nx1 = load_step(t1 - 2);
    int cur = 0;
    for (int t = t1 - 1; t >= t0; --t) {
        StepIn nx2 = nx1;
        if (t - 2 >= t0) nx2 = load_step(t - 2);
        float4 c_m2 = zero4;
        if (t > 1 && t > t0) c_m2 = *reinterpret_cast<const float4*>(cbase + (rbase + t - 2) * 1024);
        const float gi[4] = {bf2f(in.gi.x & 0xffff), bf2f(in.gi.x >> 16), bf2f(in.gi.y & 0xffff), bf2f(in.gi.y >> 16)};
        const float gf[4] = {bf2f(in.gf.x & 0xffff), bf2f(in.gf.x >> 16), bf2f(in.gf.y & 0xffff), bf2f(in.gf.y >> 16)};
        const float gg[4] = {bf2f(in.gg.x & 0xffff), bf2f(in.gg.x >> 16), bf2f(in.gg.y & 0xffff), bf2f(in.gg.y >> 16)};
        const float go[4] = {bf2f(in.go.x & 0xffff), bf2f(in.go.x >> 16), bf2f(in.go.y & 0xffff), bf2f(in.go.y >> 16)};
        const float dho[4] = {bf2f(in.dh.x & 0xffff), bf2f(in.dh.x >> 16), bf2f(in.dh.y & 0xffff), bf2f(in.dh.y >> 16)};
        const float cc[4] = {c_t.x, c_t.y, c_t.z, c_t.w};
        const float cp[4] = {c_m1.x, c_m1.y, c_m1.z, c_m1.w};
        float di[4], df[4], dg[4], dob[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dhv = sign * dho[r] + dhrec[r];
            const float tc = tanhf_(cc[r]);
            const float d_o = dhv * tc;
            const float dcv = dc[r] + dhv * go[r] * (1.f - tc * tc);
            di[r] = dcv * gg[r] * gi[r] * (1.f - gi[r]);
            df[r] = dcv * cp[r] * gf[r] * (1.f - gf[r]);
            dg[r] = dcv * gi[r] * (1.f - gg[r] * gg[r]);
            dob[r] = d_o * go[r] * (1.f - go[r]);
            dc[r] = dcv * gf[r];
        }
        const uint2 pi = make_uint2(pack_bf2(di[0], di[1]), pack_bf2(di[2], di[3]));
        const uint2 pf = make_uint2(pack_bf2(df[0], df[1]), pack_bf2(df[2], df[3]));
        const uint2 pg = make_uint2(pack_bf2(dg[0], dg[1]), pack_bf2(dg[2], dg[3]));
        const uint2 po = make_uint2(pack_bf2(dob[0], dob[1]), pack_bf2(dob[2], dob[3]));
        bf16_raw* lb = &dgbuf[cur][m * DGP + uo];
        *reinterpret_cast<uint2*>(lb) = pi;
        *reinterpret_cast<uint2*>(lb + H) = pf;
        *reinterpret_cast<uint2*>(lb + 2 * H) = pg;
        *reinterpret_cast<uint2*>(lb + 3 * H) = po;
        if (bvalid) {
            bf16_raw* dp = dpre + (size_t)t * (2 * G4);
            *reinterpret_cast<uint2*>(dp) = pi;
            *reinterpret_cast<uint2*>(dp + H) = pf;
            *reinterpret_cast<uint2*>(dp + 2 * H) = pg;
            *reinterpret_cast<uint2*>(dp + 3 * H) = po;
        }
        lds_barrier();
        dhrec = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf16x8 gfrag = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&dgbuf[cur][m * DGP + 32 * s + 8 * ug]));
            dhrec = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], gfrag, dhrec, 0, 0, 0);
        }
        cur ^= 1;
        in = nx1;
        nx1 = nx2;
        c_t = c_m1;
        c_m1 = c_m2;
    }
    if (t0 > 0 && stp) {
        *reinterpret_cast<float4*>(stp) = make_float4(dc[0], dc[1], dc[2], dc[3]);
        *reinterpret_cast<float4*>(stp + 4) = make_float4(dhrec[0], dhrec[1], dhrec[2], dhrec[3]);
    }
}

extern "C" int sehip_lstm_fwd_chunk(const float* pre0, const float* pre1, const void* whh, int B, int T, int hidden, int t0,
                                    int t1, void* h, void* gates, float* c, void* stream) {
    SEHIP_REQUIRE(hidden == H, "lstm_fwd: only hidden size 64 (rnn_units=128) is built, got %d", hidden);
    SEHIP_REQUIRE(B > 0 && T > 0, "lstm_fwd: empty input");
    SEHIP_REQUIRE(0 <= t0 && t0 < t1 && t1 <= T, "lstm_fwd: bad step range [%d, %d) of %d", t0, t1, T);
    lstm_fwd_kernel<<<4 * cdiv(B, 16), 256, 0, (hipStream_t)stream>>>(pre0, pre1, (const bf16_raw*)whh, B, T, t0, t1,
                                                                     (bf16_raw*)h, (bf16_raw*)gates, c);
    SEHIP_CHECK_LAUNCH("lstm_fwd");
    return 0;
}

extern "C" int sehip_lstm_fwd(const float* pre0, const float* pre1, const void* whh, int B, int T, int hidden, void* h,
                              void* gates, float* c, void* stream) {
    return sehip_lstm_fwd_chunk(pre0, pre1, whh, B, T, hidden, 0, T, h, gates, c, stream);
}

// state: 4 * ceil(B/16) * 256 * 8 floats carried between chunks (needed unless the chunk is the whole sequence)
extern "C" int sehip_lstm_bwd_chunk(const void* dh_a, const void* dh_b, const void* whhT, const void* gates, const float* c, int B,
                                    int T, int hidden, int t0, int t1, float* state, void* dpre0, void* dpre1, void* stream) {
    SEHIP_REQUIRE(hidden == H, "lstm_bwd: only hidden size 64 (rnn_units=128) is built, got %d", hidden);
    SEHIP_REQUIRE(B > 0 && T > 0, "lstm_bwd: empty input");
    SEHIP_REQUIRE(0 <= t0 && t0 < t1 && t1 <= T, "lstm_bwd: bad step range [%d, %d) of %d", t0, t1, T);
    SEHIP_REQUIRE(state != nullptr || (t0 == 0 && t1 == T), "lstm_bwd: a partial step range needs the state buffer");
    lstm_bwd_kernel<<<4 * cdiv(B, 16), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)dh_a, (const bf16_raw*)dh_b,
                                                                     (const bf16_raw*)whhT, (const bf16_raw*)gates, c, B, T, t0,
                                                                     t1, state, (bf16_raw*)dpre0, (bf16_raw*)dpre1);
    SEHIP_CHECK_LAUNCH("lstm_bwd");
    return 0;
}

extern "C" int sehip_lstm_bwd(const void* dh_a, const void* dh_b, const void* whhT, const void* gates, const float* c, int B,
                              int T, int hidden, void* dpre0, void* dpre1, void* stream) {
    return sehip_lstm_bwd_chunk(dh_a, dh_b, whhT, gates, c, B, T, hidden, 0, T, nullptr, dpre0, dpre1, stream);
}
