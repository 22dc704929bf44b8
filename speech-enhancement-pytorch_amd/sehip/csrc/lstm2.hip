// The TWO stacked NavieComplexLSTM layers of DCCRN (src/model/dccrn.py:264-302, wired as in :170-191) as ONE persistent launch per
// direction of the step (round 4).
//
// Round 3 ran: input product of layer 1 (GEMM) -> recurrence 1 (32 workgroups, T serial steps) -> input product of layer 2 (GEMM over
// h1) -> recurrence 2 -> projection: two latency-bound recurrences strictly one after the other with 224 CUs idle, and in the
// backward pass the same in reverse with the dx2 product between them.  Here 64 workgroups run at once: the 32 of layer 1 as before
// and 32 of layer 2 that start step t as soon as layer 1's h1(t) exists, so the two recurrences overlap almost completely
// (layer 2 trails by about a dozen steps).  What moves into the recurrence:
//   forward : layer 2's input product.  x2_r = h1[r,real] - h1[i,imag], x2_i = h1[i,real] + h1[r,imag] (the complex combination of
//             :286-291) is formed by the consumer as ONE bf16 tile per step and multiplied by W_ih (resident in registers beside
//             W_hh) two steps ahead of its use; its result, with b_ih + b_hh, is the C operand the recurrent MFMAs accumulate onto.
//   backward: layer 2's input gradient.  The workgroup that has the gate gradients dg(t) of (part, lstm) in LDS for its recurrent
//             product W_hh^T dg also forms W_ih^T dg (8 more MFMAs on the same B operand) and hands the fp32 partial to the two
//             layer-1 workgroups that need it; a layer-1 lane adds the partials of the two lstms: its dh, exactly its own element.
// Hand-off between workgroups: data-tagged 8-byte granules (cdna_hip_programming.md section 6 Guideline 16, form R2: the data IS the
// flag): {tag, 32-bit value} written by ONE sc1 (write-through) 8-byte store, read by sc1 8-byte loads until the tag matches.  No
// flag, no fence, no wait in the producer -- its own prefetch loads stay in flight -- and results never depend on placement (for
// speed the eight workgroups of a batch tile get equal blockIdx % 8 = one XCD under round-robin dispatch).  Producers never wait
// for consumers (every step has its own granule slots), so there is no cycle; a consumer's spin is bounded and a time-out sets
// the sticky word sync[0] (the launch ends with garbage, the optimizer's device-side guard skips the step: FlatOptimizer / sehip_opt_*_g).
// tag = (epoch << 16) | (step index + 1) with a per-call epoch from the host, so the granule arrays are never cleared between calls
// (every slot is rewritten by every call); under stream capture the host clears them with a memset node instead (a captured epoch is
// frozen).
#include <stdlib.h>
#include "common.h"

#define H 64
#define G4 256
#define HP 72      // LDS pitch of an h / x2 tile (bf16 elements)
#define DGP 264    // LDS pitch of the gate-gradient tile
#define NBT 4      // batch rows per workgroup
#define PD 8       // prefetch distance in time steps (see lstm.hip)
#define L2_TMO 0   // sync word: sticky time-out
#define L2_LIM 1   // sync word: test hook, replaces the spin limit (0xffffffff: the first wait times out)
#define L2_SPIN_LIMIT (1u << 18)   // re-loads of ~1 us each

typedef __attribute__((address_space(1))) unsigned long long l2_gu64;
typedef __attribute__((address_space(1))) unsigned l2_gu32;
#define L2_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ void l2_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// s_waitcnt vmcnt(0) the compiler KNOWS about (the builtin, not inline asm).  Placed once in front of a software-pipelined loop: hipcc's
// counted waits inside the loop merge the loop-entry state with the back-edge state per register and keep the more recent one; the
// prologue's ring loads are recent on the entry path, so without this the first step of every unrolled group of PD steps drained the
// wave's whole memory queue (one full memory latency per PD steps: fused forward 158 us instead of ~125).
__device__ __forceinline__ void l2_drain_known() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0) expcnt(7) lgkmcnt(15)
// Explicit guard for the register rings of plain loads (pre-gates, gate / cell records): inside these software-pipelined loops hipcc
// emits NO vmcnt wait for a value requested PD steps earlier (lstm.hip's loops have none either), so a load slower than PD steps
// would be read before it lands.  N = memory instructions a wave issues in PD - 2 steps, rounded down: in steady state the wait is
// already satisfied (everything older than ~6 steps has returned) and costs nothing; if memory is slower than that it stalls
// instead of reading a register in flight.  (The granule rings need none: a value read too early fails its tag check.  The backward
// kernels carry none: vmcnt retires in order, their sc1 granule loads take ~2 us and held the counter above any useful N -- with
// a guard of 36 the fused backward launch took 187 us instead of 174.)
template <int N>
__device__ __forceinline__ void l2_guard() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ float l2_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float l2_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
__device__ __forceinline__ float l2_pick(const f32x4& a, int rs) { return rs == 0 ? a[0] : rs == 1 ? a[1] : rs == 2 ? a[2] : a[3]; }

__device__ __forceinline__ void l2_store_granule(unsigned long long* g, unsigned tag, unsigned value) {
    __hip_atomic_store((l2_gu64*)g, ((unsigned long long)tag << 32) | value, L2_RLX_AGENT);       // ONE aligned 8-byte sc1 store
}
__device__ __forceinline__ unsigned long long l2_load_granule(const unsigned long long* g) {
    return __hip_atomic_load((l2_gu64*)g, L2_RLX_AGENT);                                           // sc1: never served by this CU's L1
}
// wave-uniform bounded wait for the wave's two granules of one step.  (a, b) arrive from the prefetch ring; the FIRST check is
// straight-line code, so that the compiler's counted s_waitcnt for those ring loads is exact (with the check at the head of the spin
// loop it merged the ring loads' state with the re-loads' and drained the wave's whole memory queue every time step: fused kernel
// 222 us instead of ~130).  The spin itself works on its own temporaries.
__device__ __forceinline__ void l2_wait_pair(const unsigned long long* ga, const unsigned long long* gb, unsigned tag,
                                             unsigned long long& a, unsigned long long& b, unsigned* sync, bool& dead, unsigned limit) {
    if (dead) return;
    if (__builtin_expect(__all((unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag), 1)) return;
    if (limit == 0xffffffffu) {          // test hook: "time out" at the first wait that is not already satisfied
        __hip_atomic_store((l2_gu32*)(sync + L2_TMO), 1u, L2_RLX_AGENT);
        dead = true;
        return;
    }
    unsigned spins = 0;
    unsigned long long x, y;
    do {
        if (++spins > limit || ((spins & 255u) == 0 && __hip_atomic_load((l2_gu32*)(sync + L2_TMO), L2_RLX_AGENT) != 0)) {
            __hip_atomic_store((l2_gu32*)(sync + L2_TMO), 1u, L2_RLX_AGENT);
            dead = true;          // give up for the rest of the sequence: garbage out, but the launch ends and the guard word is set
            return;
        }
        __builtin_amdgcn_s_sleep(2);
        x = l2_load_granule(ga);
        y = l2_load_granule(gb);
    } while (!__all((unsigned)(x >> 32) == tag && (unsigned)(y >> 32) == tag));
    a = x;
    b = y;
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward.  blockIdx = cl * ntiles + tile, cl = layer * 4 + combo (layer-1 workgroups first), combo = part * 2 + lstm.
// gran: [4 combos of layer 1][ntiles][T][128] granules; granule i of a step = h1 of batch row i >> 5, units 2 (i & 31), +1.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm2_fwd_kernel(const float* __restrict__ pre0, const float* __restrict__ pre1,
                                                        const bf16_raw* __restrict__ whh1, const bf16_raw* __restrict__ whh2,
                                                        const bf16_raw* __restrict__ wih2, const float* __restrict__ bias2, int B, int T,
                                                        bf16_raw* __restrict__ h1out, bf16_raw* __restrict__ gates1, float* __restrict__ c1out,
                                                        bf16_raw* __restrict__ h2out, bf16_raw* __restrict__ gates2, float* __restrict__ c2out,
                                                        unsigned long long* gran, unsigned* sync, unsigned epoch) {
    __shared__ __attribute__((aligned(16))) bf16_raw hbuf[2][NBT * HP];
    __shared__ __attribute__((aligned(16))) bf16_raw xbuf[2][NBT * HP];
    // (round 6: s_setprio 3 here -- the waves of this latency-bound chain winning their SIMDs' issue arbitration over co-resident
    //  throughput work -- measured no effect, forward beside the decoders' skip halves 213 vs 216 us, backward 204 vs 203: what the
    //  recurrence loses beside other kernels is hand-off latency through L2, not issue slots)
    const int ntiles = gridDim.x >> 3;
    const int cl = blockIdx.x / ntiles, tile = blockIdx.x - cl * ntiles;
    const int layer = cl >> 2, combo = cl & 3;
    const int part = combo >> 1, lstm = combo & 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, ug = lane >> 4;
    const int bl = m & (NBT - 1), rs = m >> 2;
    const int unit = 16 * w + 4 * ug + rs;
    const int b = tile * NBT + bl;
    const bool bvalid = b < B;
    const int bc = bvalid ? b : B - 1;
    const size_t obase = ((size_t)combo * B + bc) * T;
    const size_t rbase = (size_t)(combo * ntiles + tile) * T;
    const unsigned tag0 = epoch << 16;
    for (int i = threadIdx.x; i < NBT * HP; i += 256) { hbuf[0][i] = 0; hbuf[1][i] = 0; xbuf[0][i] = 0; xbuf[1][i] = 0; }
    float c = 0.f;

    if (layer == 0) {
        // ---- layer 1: lstm_fwd_kernel of lstm.hip + the publication of h1(t - 1) at the top of step t
        const float* pre = (part ? pre1 : pre0) + ((size_t)bc * T) * (2 * G4) + lstm * G4 + unit;
        bf16x8 wf[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                wf[g][s] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const uint4*>(whh1 + ((size_t)lstm * G4 + g * H + 16 * w + m) * H + 32 * s + 8 * ug));
        float pf[PD][4];
#pragma unroll
        for (int k = 0; k < PD; ++k)
            if (k < T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) pf[k][g] = pre[(size_t)k * (2 * G4) + g * H];
            }
        __syncthreads();
        // publisher lanes: waves 0 and 1, granule i = 64 w + lane
        const int gi_ = 64 * w + lane, prow = gi_ >> 5, ppair = gi_ & 31;
        unsigned long long* gout = gran + ((size_t)(combo * ntiles + tile) * T) * 128 + gi_;
        int cur = 0;
        auto step = [&](int t, float (&pq)[4]) {
            l2_guard<42>();                      // 7-8 memory instructions per step
            if (t + PD >= T && t + PD < T + 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last requests: nothing is issued behind them
            const float p0 = pq[0], p1 = pq[1], p2 = pq[2], p3 = pq[3];
            if (t + PD < T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) pq[g] = pre[(size_t)(t + PD) * (2 * G4) + g * H];
            }
            bf16x8 hf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
                hf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&hbuf[cur][bl * HP + 32 * s + 8 * ug]));
            unsigned pubv = 0;
            if (w < 2) pubv = *reinterpret_cast<const unsigned*>(&hbuf[cur][prow * HP + 2 * ppair]);     // h1(t - 1), the tile every wave reads
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][0], hf[0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][1], hf[1], acc[g], 0, 0, 0);
            }
            if (w < 2 && t > 0) l2_store_granule(gout + (size_t)(t - 1) * 128, tag0 | (unsigned)t, pubv);
            const float gi = l2_sigmoid(l2_pick(acc[0], rs) + p0);
            const float gf = l2_sigmoid(l2_pick(acc[1], rs) + p1);
            const float gg = l2_tanh(l2_pick(acc[2], rs) + p2);
            const float go = l2_sigmoid(l2_pick(acc[3], rs) + p3);
            c = gf * c + gi * gg;
            const bf16_raw hb = f2bf(go * l2_tanh(c));
            hbuf[cur ^ 1][bl * HP + unit] = hb;
            if (bvalid) h1out[(obase + t) * H + unit] = hb;
            const size_t rec = (rbase + t) * 256 + threadIdx.x;
            c1out[rec] = c;
            *reinterpret_cast<uint2*>(gates1 + rec * 4) = make_uint2(pack_bf2(gi, gf), pack_bf2(gg, go));
            l2_lds_barrier();
            cur ^= 1;
        };
        for (int t = 0; t < T; t += PD) {
#pragma unroll
            for (int k = 0; k < PD; ++k)
                if (t + k < T) step(t + k, pf[k]);
        }
        if (w < 2) {
            const unsigned pubv = *reinterpret_cast<const unsigned*>(&hbuf[cur][prow * HP + 2 * ppair]);
            l2_store_granule(gout + (size_t)(T - 1) * 128, tag0 | (unsigned)T, pubv);
        }
        return;
    }

    // ---- layer 2.  x2 of `part`: part 0 (real input): h1[combo 0] - h1[combo 3]; part 1: h1[combo 2] + h1[combo 1]
    const int ca = part ? 2 : 0, cb = part ? 1 : 3;
    const float sgn = part ? 1.f : -1.f;
    bf16x8 wf[4][2], wi[4][2];
    f32x4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            wf[g][s] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const uint4*>(whh2 + ((size_t)lstm * G4 + g * H + 16 * w + m) * H + 32 * s + 8 * ug));
            wi[g][s] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const uint4*>(wih2 + ((size_t)lstm * G4 + g * H + 16 * w + m) * H + 32 * s + 8 * ug));
        }
        // D rows = units 16 w + 4 ug + (0..3) of gate g
        bq[g] = *reinterpret_cast<const f32x4*>(bias2 + lstm * G4 + g * H + 16 * w + 4 * ug);
    }
    bool dead = false;
    const unsigned limit = sync[L2_LIM] ? sync[L2_LIM] : L2_SPIN_LIMIT;
    // stager lanes: waves 0 and 1; lane's granule i = 64 w + lane of both sources
    // (every wave stages: waves 2 and 3 repeat the loads and the LDS writes of waves 0 and 1 with the same values.  Nothing about the
    //  memory instructions of a step is conditional -- step indices are clamped instead of branched on -- because hipcc's counted
    //  s_waitcnt for a ring load assumes that conditionally issued instructions behind it were NOT issued: with `if (stager && s < T)`
    //  around the loads it waited for all but the newest five operations, i.e. for the previous step's loads, on every time step)
    const int gi_ = 64 * (w & 1) + lane, prow = gi_ >> 5, ppair = gi_ & 31;
    const unsigned long long* ga = gran + ((size_t)(ca * ntiles + tile) * T) * 128 + gi_;
    const unsigned long long* gb = gran + ((size_t)(cb * ntiles + tile) * T) * 128 + gi_;
    unsigned long long ra[PD], rb[PD];      // granules of steps s .. s + PD - 1 (ring indexed by s % PD after unrolling)
#pragma unroll
    for (int k = 0; k < PD; ++k) {
        const int kk = k < T ? k : T - 1;
        ra[k] = l2_load_granule(ga + (size_t)kk * 128); rb[k] = l2_load_granule(gb + (size_t)kk * 128);
    }
    // stage(s, slot): x2(s) -> xbuf[s & 1]; then the slot is re-armed with the granules of step s + PD (past the end: the last step's again)
    auto stage = [&](int s, unsigned long long& qa, unsigned long long& qb) {
        const int sc = s < T ? s : T - 1;
        unsigned long long ua = qa, ub = qb;
        l2_wait_pair(ga + (size_t)sc * 128, gb + (size_t)sc * 128, tag0 | (unsigned)(sc + 1), ua, ub, sync, dead, limit);
        const unsigned va = (unsigned)ua, vb = (unsigned)ub;
        const float x0 = __uint_as_float(va << 16) + sgn * __uint_as_float(vb << 16);
        const float x1 = __uint_as_float(va & 0xffff0000u) + sgn * __uint_as_float(vb & 0xffff0000u);
        *reinterpret_cast<unsigned*>(&xbuf[s & 1][prow * HP + 2 * ppair]) = pack_bf2(x0, x1);
        const int sn = s + PD < T ? s + PD : T - 1;
        qa = l2_load_granule(ga + (size_t)sn * 128); qb = l2_load_granule(gb + (size_t)sn * 128);
    };
    // ih(s): W_ih x2(s) + b  (reads xbuf[s & 1])
    auto ih = [&](int s, f32x4 (&out)[4]) {
        bf16x8 xf[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
            xf[q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&xbuf[s & 1][bl * HP + 32 * q + 8 * ug]));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            out[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wi[g][0], xf[0], bq[g], 0, 0, 0);
            out[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wi[g][1], xf[1], out[g], 0, 0, 0);
        }
    };
    __syncthreads();
    f32x4 accih[4];          // W_ih x2(t) + b of the step about to run
    stage(0, ra[0], rb[0]);
    l2_lds_barrier();
    ih(0, accih);
    stage(1, ra[1 % PD], rb[1 % PD]);
    l2_lds_barrier();
    l2_drain_known();
    int cur = 0;
    auto step = [&](int t, unsigned long long& qa, unsigned long long& qb) {      // (qa, qb): ring slot of step t + 2
        bf16x8 hf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
            hf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&hbuf[cur][bl * HP + 32 * s + 8 * ug]));
        f32x4 acc[4], nxt[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][0], hf[0], accih[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][1], hf[1], acc[g], 0, 0, 0);
        }
        // off the dependent chain: the input product of step t + 1 (x2(t + 1) was staged in step t - 1) goes into the matrix pipe right
        // behind the recurrent MFMAs and runs under the gate arithmetic below (issued after the stores it delayed the barrier by its
        // whole 8 x 32 cycles: 154 us per launch instead of ~125)
        ih(t + 1, nxt);
        const float gi = l2_sigmoid(l2_pick(acc[0], rs));
        const float gf = l2_sigmoid(l2_pick(acc[1], rs));
        const float gg = l2_tanh(l2_pick(acc[2], rs));
        const float go = l2_sigmoid(l2_pick(acc[3], rs));
        c = gf * c + gi * gg;
        const bf16_raw hb = f2bf(go * l2_tanh(c));
        hbuf[cur ^ 1][bl * HP + unit] = hb;
        h2out[(obase + t) * H + unit] = hb;        // (rows past the batch repeat row B - 1 with the same value: no branch, see above)
        const size_t rec = (rbase + t) * 256 + threadIdx.x;
        c2out[rec] = c;
        *reinterpret_cast<uint2*>(gates2 + rec * 4) = make_uint2(pack_bf2(gi, gf), pack_bf2(gg, go));
        // the staging of x2(t + 2) (in front of the gate arithmetic instead, "under the MFMAs", the launch took 154 us instead of 143:
        // its tag check and LDS write then sit between the MFMA issue and the first use of their results)
        stage(t + 2, qa, qb);
#pragma unroll
        for (int g = 0; g < 4; ++g) accih[g] = nxt[g];
        l2_lds_barrier();
        cur ^= 1;
    };
    for (int t = 0; t < T; t += PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k)
            if (t + k < T) step(t + k, ra[(k + 2) % PD], rb[(k + 2) % PD]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward.  blockIdx = cl * ntiles + tile, cl < 4: LAYER 2 (the producers run first in the backward pass), cl >= 4: layer 1.
// gran: [4 combos of layer 2][ntiles][T][256] granules; granule of thread i at step t = fp32 (W_ih^T dg(t))[unit(i)][batch row(i)].
// dh_a / dh_b: gradients of the layer-2 outputs out_r = h2[r,real] - h2[i,imag], out_i = h2[i,real] + h2[r,imag]:
//   combo 0: +dh_a   1: +dh_b   2: +dh_b   3: -dh_a     (and the same rule for layer 1 with dx2_r / dx2_i, which arrive as granules)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm2_bwd_kernel(const bf16_raw* __restrict__ dh_a, const bf16_raw* __restrict__ dh_b,
                                                        const bf16_raw* __restrict__ whhT1, const bf16_raw* __restrict__ whhT2,
                                                        const bf16_raw* __restrict__ wihT2, const bf16_raw* __restrict__ gates1,
                                                        const float* __restrict__ c1, const bf16_raw* __restrict__ gates2,
                                                        const float* __restrict__ c2, int B, int T, bf16_raw* __restrict__ dpre1_0,
                                                        bf16_raw* __restrict__ dpre1_1, bf16_raw* __restrict__ dpre2_0,
                                                        bf16_raw* __restrict__ dpre2_1, unsigned long long* gran, unsigned* sync,
                                                        unsigned epoch, int abl) {
    __shared__ __attribute__((aligned(16))) bf16_raw dgbuf[2][NBT * DGP];
    const int ntiles = gridDim.x >> 3;
    const int cl = blockIdx.x / ntiles, tile = blockIdx.x - cl * ntiles;
    const bool second = cl < 4;                     // layer 2
    const int combo = cl & 3;
    const int part = combo >> 1, lstm = combo & 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, ug = lane >> 4;
    const int bl = m & (NBT - 1), rs = m >> 2;
    const int unit = 16 * w + 4 * ug + rs;
    const int b = tile * NBT + bl;
    const int bc = b < B ? b : B - 1;
    const float sign = combo == 3 ? -1.f : 1.f;
    const bool use_a = combo == 0 || combo == 3;
    const unsigned tag0 = epoch << 16;
    const int srow = tile * NBT + w;
    bf16_raw* dpre = (second ? (part ? dpre2_1 : dpre2_0) : (part ? dpre1_1 : dpre1_0)) +
                     ((size_t)(srow < B ? srow : 0) * T) * (2 * G4) + lstm * G4 + lane * 4;
    const bool svalid = srow < B;
    const bf16_raw* whhT = second ? whhT2 : whhT1;
    const bf16_raw* gates = second ? gates2 : gates1;
    const float* cst = second ? c2 : c1;
    bf16x8 wf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
        wf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(whhT + ((size_t)lstm * H + 16 * w + m) * G4 + 32 * s + 8 * ug));
    const size_t rbase = (size_t)(combo * ntiles + tile) * T;
    const float* cbase = cst + rbase * 256 + threadIdx.x;
    float dc = 0.f, dhrec = 0.f;
    int cur = 0;

    if (second) {
        bf16x8 wx[8];        // W_ih^T fragments: rows k (input dimension) = 16 w + (lane & 15), reduction over the 256 gates
#pragma unroll
        for (int s = 0; s < 8; ++s)
            wx[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wihT2 + ((size_t)lstm * H + 16 * w + m) * G4 + 32 * s + 8 * ug));
        const bf16_raw* dhp = (use_a ? dh_a : dh_b) + ((size_t)bc * T) * H + unit;
        unsigned long long* gout = gran + ((size_t)(combo * ntiles + tile) * T) * 256 + threadIdx.x;
        struct StepIn { uint2 g; bf16_raw dh; float c, cp; };
        auto load_step = [&](int t) {
            StepIn v;
            v.g = *reinterpret_cast<const uint2*>(gates + ((rbase + t) * 256 + threadIdx.x) * 4);
            v.dh = dhp[(size_t)t * H];
            v.c = cbase[(size_t)t * 256];
            v.cp = t > 0 ? cbase[(size_t)(t - 1) * 256] : 0.f;
            return v;
        };
        StepIn ring[PD];
#pragma unroll
        for (int k = 0; k < PD; ++k)
            if (T - 1 - k >= 0) ring[k] = load_step(T - 1 - k);
        f32x4 x0 = (f32x4){0.f, 0.f, 0.f, 0.f}, x1 = x0;       // the input-gradient partial of the step before (stored one step late)
        auto step = [&](int t, StepIn& slot) {
            const StepIn in = slot;
            if (t - PD >= 0) slot = load_step(t - PD);
            const float gi = bf2f(in.g.x & 0xffff), gf = bf2f(in.g.x >> 16), gg = bf2f(in.g.y & 0xffff), go = bf2f(in.g.y >> 16);
            const float dhv = sign * bf2f(in.dh) + dhrec;
            const float tc = l2_tanh(in.c);
            const float d_o = dhv * tc;
            const float dcv = dc + dhv * go * (1.f - tc * tc);
            const bf16_raw di = f2bf(dcv * gg * gi * (1.f - gi));
            const bf16_raw df = f2bf(dcv * in.cp * gf * (1.f - gf));
            const bf16_raw dg = f2bf(dcv * gi * (1.f - gg * gg));
            const bf16_raw dob = f2bf(d_o * go * (1.f - go));
            dc = dcv * gf;
            bf16_raw* lb = &dgbuf[cur][bl * DGP + unit];
            lb[0] = di; lb[H] = df; lb[2 * H] = dg; lb[3 * H] = dob;
            l2_lds_barrier();
            // the previous step's input-gradient partial leaves now (its MFMAs ran under this step's gate arithmetic)
            if (t < T - 1 && !(abl & 2)) l2_store_granule(gout + (size_t)(t + 1) * 256, tag0 | (unsigned)(T - t - 1), __float_as_uint(l2_pick(x0, rs) + l2_pick(x1, rs)));
            bf16x8 gq[8];
#pragma unroll
            for (int s = 0; s < 8; ++s)
                gq[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&dgbuf[cur][bl * DGP + 32 * s + 8 * ug]));
            f32x4 r0 = (f32x4){0.f, 0.f, 0.f, 0.f}, r1 = r0;
            x0 = r0; x1 = r0;
            // the recurrent product first (its result is the next step's dh), the input gradient's partial behind it on the same B
            // operands (interleaved with it the partial delayed dhrec by six MFMAs per step)
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], gq[s], r0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s + 1], gq[s + 1], r1, 0, 0, 0);
            }
            if (!(abl & 1)) {
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                x0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wx[s], gq[s], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wx[s + 1], gq[s + 1], x1, 0, 0, 0);
            }
            }
            {
                const uint2 row = *reinterpret_cast<const uint2*>(&dgbuf[cur][w * DGP + lane * 4]);
                if (svalid) *reinterpret_cast<uint2*>(dpre + (size_t)t * (2 * G4)) = row;
            }
            dhrec = l2_pick(r0, rs) + l2_pick(r1, rs);
            cur ^= 1;
        };
        for (int t = T - 1; t >= 0; t -= PD) {
#pragma unroll
            for (int k = 0; k < PD; ++k)
                if (t - k >= 0) step(t - k, ring[k]);
        }
        l2_store_granule(gout, tag0 | (unsigned)T, __float_as_uint(l2_pick(x0, rs) + l2_pick(x1, rs)));     // step 0's partial
        return;
    }

    // ---- layer 1: its dh is dx2 of part' (r for combos 0 and 3, i for 1 and 2) = the partials of layer-2 workgroups (part', lstm 0 / 1)
    const int pp = use_a ? 0 : 1;
    const unsigned long long* ga = gran + ((size_t)((2 * pp) * ntiles + tile) * T) * 256 + threadIdx.x;
    const unsigned long long* gb = gran + ((size_t)((2 * pp + 1) * ntiles + tile) * T) * 256 + threadIdx.x;
    bool dead = false;
    const unsigned limit = sync[L2_LIM] ? sync[L2_LIM] : L2_SPIN_LIMIT;
    struct StepIn1 { uint2 g; float c, cp; unsigned long long qa, qb; };
    auto load_step = [&](int t) {
        StepIn1 v;
        v.g = *reinterpret_cast<const uint2*>(gates + ((rbase + t) * 256 + threadIdx.x) * 4);
        v.c = cbase[(size_t)t * 256];
        v.cp = t > 0 ? cbase[(size_t)(t - 1) * 256] : 0.f;
        v.qa = l2_load_granule(ga + (size_t)t * 256);
        v.qb = l2_load_granule(gb + (size_t)t * 256);
        return v;
    };
    StepIn1 ring[PD];
#pragma unroll
    for (int k = 0; k < PD; ++k) ring[k] = load_step(T - 1 - k >= 0 ? T - 1 - k : 0);
    l2_drain_known();
    auto step = [&](int t, StepIn1& slot) {
        StepIn1 in = slot;
        if (!(abl & 4)) l2_wait_pair(ga + (size_t)t * 256, gb + (size_t)t * 256, tag0 | (unsigned)(T - t), in.qa, in.qb, sync, dead, limit);
        slot = load_step(t - PD >= 0 ? t - PD : 0);       // (clamped, not branched on: see the forward consumer)
        const float gi = bf2f(in.g.x & 0xffff), gf = bf2f(in.g.x >> 16), gg = bf2f(in.g.y & 0xffff), go = bf2f(in.g.y >> 16);
        const float dx = __uint_as_float((unsigned)in.qa) + __uint_as_float((unsigned)in.qb);
        const float dhv = sign * dx + dhrec;
        const float tc = l2_tanh(in.c);
        const float d_o = dhv * tc;
        const float dcv = dc + dhv * go * (1.f - tc * tc);
        const bf16_raw di = f2bf(dcv * gg * gi * (1.f - gi));
        const bf16_raw df = f2bf(dcv * in.cp * gf * (1.f - gf));
        const bf16_raw dg = f2bf(dcv * gi * (1.f - gg * gg));
        const bf16_raw dob = f2bf(d_o * go * (1.f - go));
        dc = dcv * gf;
        bf16_raw* lb = &dgbuf[cur][bl * DGP + unit];
        lb[0] = di; lb[H] = df; lb[2 * H] = dg; lb[3 * H] = dob;
        l2_lds_barrier();
        f32x4 r0 = (f32x4){0.f, 0.f, 0.f, 0.f}, r1 = r0;
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            const bf16x8 g0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&dgbuf[cur][bl * DGP + 32 * s + 8 * ug]));
            const bf16x8 g1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&dgbuf[cur][bl * DGP + 32 * s + 32 + 8 * ug]));
            r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], g0, r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s + 1], g1, r1, 0, 0, 0);
        }
        {
            const uint2 row = *reinterpret_cast<const uint2*>(&dgbuf[cur][w * DGP + lane * 4]);
            if (svalid) *reinterpret_cast<uint2*>(dpre + (size_t)t * (2 * G4)) = row;
        }
        dhrec = l2_pick(r0, rs) + l2_pick(r1, rs);
        cur ^= 1;
    };
    for (int t = T - 1; t >= 0; t -= PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k)
            if (t - k >= 0) step(t - k, ring[k]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
extern "C" long sehip_lstm2_gran_bytes(int B, int T, int backward) {
    const long ntiles = (B + NBT - 1) / NBT;
    return 4L * ntiles * T * (backward ? 256 : 128) * 8;
}
extern "C" int sehip_lstm2_sync_bytes(void) { return 64; }

static int lstm2_check(const char* who, int B, int T, int hidden, unsigned epoch) {
    SEHIP_REQUIRE(hidden == H, "%s: only hidden size 64 (rnn_units=128) is built, got %d", who, hidden);
    SEHIP_REQUIRE(B > 0 && T > 0 && T < 65535, "%s: bad sizes B=%d T=%d (T < 65535: the step index shares the tag with the epoch)", who, B, T);
    SEHIP_REQUIRE(epoch != 0 && epoch < 65536, "%s: epoch must be in [1, 65535]", who);
    // every workgroup must be resident at once (consumers wait for producers inside the launch): 8 per batch tile
    SEHIP_REQUIRE(8 * ((B + NBT - 1) / NBT) <= 1024, "%s: batch %d needs more than 1024 co-resident workgroups", who, B);
    return 0;
}

// pre_r / pre_i: layer 1's pre-gates fp32 [B][T][512] (from the input GEMM); whh1 / whh2 bf16 [2 lstm][256][64]; wih2 bf16 [2][256][64];
// bias2 fp32 [2][256] = b_ih + b_hh of layer 2; h*: [4 combos][B][T][64] bf16; gates* / c*: the records of sehip_lstm_fwd;
// gran: sehip_lstm2_gran_bytes(B, T, 0) bytes, zeroed ONCE at allocation; sync: sehip_lstm2_sync_bytes() bytes (word 0: sticky
// time-out, word 1: test hook), zeroed at allocation; epoch in [1, 65535], different from the previous call's on the same gran.
extern "C" int sehip_lstm2_fwd(const float* pre_r, const float* pre_i, const void* whh1, const void* whh2, const void* wih2,
                               const float* bias2, int B, int T, int hidden, void* h1, void* gates1, float* c1, void* h2, void* gates2,
                               float* c2, void* gran, unsigned* sync, unsigned epoch, void* stream) {
    if (int e = lstm2_check("lstm2_fwd", B, T, hidden, epoch)) return e;
    SEHIP_REQUIRE(gran && sync, "lstm2_fwd: missing granule / sync buffers");
    lstm2_fwd_kernel<<<8 * cdiv(B, NBT), 256, 0, (hipStream_t)stream>>>(
        pre_r, pre_i, (const bf16_raw*)whh1, (const bf16_raw*)whh2, (const bf16_raw*)wih2, bias2, B, T, (bf16_raw*)h1, (bf16_raw*)gates1,
        c1, (bf16_raw*)h2, (bf16_raw*)gates2, c2, (unsigned long long*)gran, sync, epoch);
    SEHIP_CHECK_LAUNCH("lstm2_fwd");
    return 0;
}

// dh_a / dh_b: gradients of layer 2's two outputs bf16 [B][T][64]; whhT* bf16 [2][64][256]; wihT2 bf16 [2][64][256] (layer 2's W_ih
// transposed); dpre*: gate gradients bf16 [B][T][512] per part (the operands of the weight gradients and of layer 1's input gradient).
extern "C" int sehip_lstm2_bwd(const void* dh_a, const void* dh_b, const void* whhT1, const void* whhT2, const void* wihT2,
                               const void* gates1, const float* c1, const void* gates2, const float* c2, int B, int T, int hidden,
                               void* dpre1_r, void* dpre1_i, void* dpre2_r, void* dpre2_i, void* gran, unsigned* sync, unsigned epoch,
                               void* stream) {
    if (int e = lstm2_check("lstm2_bwd", B, T, hidden, epoch)) return e;
    SEHIP_REQUIRE(gran && sync, "lstm2_bwd: missing granule / sync buffers");
    lstm2_bwd_kernel<<<8 * cdiv(B, NBT), 256, 0, (hipStream_t)stream>>>(
        (const bf16_raw*)dh_a, (const bf16_raw*)dh_b, (const bf16_raw*)whhT1, (const bf16_raw*)whhT2, (const bf16_raw*)wihT2,
        (const bf16_raw*)gates1, c1, (const bf16_raw*)gates2, c2, B, T, (bf16_raw*)dpre1_r, (bf16_raw*)dpre1_i, (bf16_raw*)dpre2_r,
        (bf16_raw*)dpre2_i, (unsigned long long*)gran, sync, epoch,
#ifdef SEHIP_TOOLS_BUILD
        getenv("SEHIP_L2_ABL") ? atoi(getenv("SEHIP_L2_ABL")) : 0      // timing ablations (wrong results), tools builds only
#else
        0
#endif
    );
    SEHIP_CHECK_LAUNCH("lstm2_bwd");
    return 0;
}
