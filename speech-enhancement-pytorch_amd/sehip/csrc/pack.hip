// Table-driven weight packing / gradient un-packing between the reference's parameter tensors (one flat fp32
// buffer holding every tensor of the state_dict, src/model/dccrn.py:62-137) and the GEMM-side layouts:
//   - packed bf16 weights  W[n][k]  (complex block matrix [[Wr,-Wi],[Wi,Wr]], tap/channel order of the K table,
//     LSTM column permutations, [W|-W] sign-folded concatenations)
//   - packed fp32 biases   (ComplexConv2d adds each real conv's bias twice-signed: real b_r-b_i, imag b_r+b_i,
//     src/model/dccrn.py:374-382; LSTM b_ih+b_hh)
//   - parameter gradients  g[j] = sum of up to 4 signed entries of the packed-gradient buffer.
// An entry e encodes (index << 1) | negate, -1 = absent.  The tables are built once on the host.
#include "common.h"
#include <stdlib.h>

__device__ __forceinline__ float term(const float* __restrict__ src, int e) {
    // (unconditional load -- entry 0 stands in for an absent one: a predicated load is waited for at once and serialises its neighbours)
    const float v = src[(e < 0 ? 0 : e) >> 1];
    return e < 0 ? 0.f : ((e & 1) ? -v : v);
}

// eight outputs per thread: two 16-byte table loads, eight gathers in flight, one 16-byte store (one element per thread
// with 2-byte stores took 42 us for the 4 M packed weights of DCCRN at the head of every step)
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ params, const int* __restrict__ tab, long n,
                                                        bf16_raw* __restrict__ out) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const int4 e0 = *reinterpret_cast<const int4*>(tab + 8 * i), e1 = *reinterpret_cast<const int4*>(tab + 8 * i + 4);
        const int e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = params[e[j] < 0 ? 0 : e[j] >> 1];   // unconditional loads, fixed up below
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = e[j] < 0 ? 0.f : ((e[j] & 1) ? -v[j] : v[j]);
        *reinterpret_cast<uint4*>(out + 8 * i) =
            make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
    if (blockIdx.x == 0)
        for (long i = (n8 << 3) + threadIdx.x; i < n; i += 256) out[i] = f2bf(term(params, tab[i]));
}

__global__ __launch_bounds__(256) void pack_f32_kernel(const float* __restrict__ params, const int2* __restrict__ tab, long n,
                                                       float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int2 e = tab[i];
        out[i] = term(params, e.x) + term(params, e.y);
    }
}

// the head of a step in one launch: both packings and the clearing of an accumulator region (three kernel boundaries -> one)
__global__ __launch_bounds__(256) void pack_head_kernel(const float* __restrict__ params, const int* __restrict__ wtab, long nw,
                                                        bf16_raw* __restrict__ wout, const int2* __restrict__ btab, long nb,
                                                        float* __restrict__ bout, float* __restrict__ zero, long nz) {
    const long nw8 = nw >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nw8; i += (long)gridDim.x * 256) {
        const int4 e0 = *reinterpret_cast<const int4*>(wtab + 8 * i), e1 = *reinterpret_cast<const int4*>(wtab + 8 * i + 4);
        const int e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = params[e[j] < 0 ? 0 : e[j] >> 1];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = e[j] < 0 ? 0.f : ((e[j] & 1) ? -v[j] : v[j]);
        *reinterpret_cast<uint4*>(wout + 8 * i) =
            make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
    if (blockIdx.x == 0)
        for (long i = (nw8 << 3) + threadIdx.x; i < nw; i += 256) wout[i] = f2bf(term(params, wtab[i]));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (long)gridDim.x * 256) {
        const int2 e = btab[i];
        bout[i] = term(params, e.x) + term(params, e.y);
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nz; i += (long)gridDim.x * 256) zero[i] = 0.f;
}

__global__ __launch_bounds__(256) void unpack_grad_kernel(const float* __restrict__ packed, const int4* __restrict__ tab, long n,
                                                          float* __restrict__ grads) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int4 e = tab[i];
        grads[i] = term(packed, e.x) + term(packed, e.y) + term(packed, e.z) + term(packed, e.w);
    }
}

// unpack_grad + the optimizer's two reductions over the SAME values while they are in registers (single-replica steps: the sums
// must be taken after the data-parallel all-reduce otherwise): the clipping norm's sum of squares and the per-tensor sums of the
// reference's logged grad_norm (src/solver.py:487-498).  Workgroup b owns a contiguous range of parameters and flushes its running
// sum at every tensor boundary inside it (as tensor_sums_flat_kernel); workgroup 0 also advances the optimizer's device step counter
// (what sehip_opt_begin did in a launch of its own).  Three launches less on the tail of the step's chain.
// perm (round 6, optional): position i of a tensor's range un-packs parameter perm[i] OF THE SAME TENSOR, the table is in position
// order.  With the positions of a tensor sorted by the address of their first packed entry, the lanes of a wave gather neighbouring
// floats of gpack (a convolution weight [co][ci][kf][kt] reads dW[n][(kt, kf, ci)]: in parameter order consecutive lanes are 5 C
// floats apart and every 4-byte read pulls its own 64-byte sector -- 281 MB of traffic to re-lay 8.3 MB of gradients at the DCCRN
// headline shape) and the scattered side becomes the 4-byte store, which L2 merges.  The per-tensor sums do not notice.
__global__ __launch_bounds__(256) void unpack_grad_sums_kernel(const float* __restrict__ packed, const int4* __restrict__ tab, long n,
                                                               float* __restrict__ grads, const long* __restrict__ offsets, int ntensors,
                                                               double* __restrict__ sumsq, float* __restrict__ tsums,
                                                               int* __restrict__ counter, const unsigned* __restrict__ guard,
                                                               const int* __restrict__ perm) {
    __shared__ float red[4];
    __shared__ int first;
    if (blockIdx.x == 0 && threadIdx.x == 0 && counter && !(guard && guard[0] != 0u)) counter[0] += 1;
    const long per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    long lo = (long)blockIdx.x * per;
    const long hi = min(n, lo + per);
    if (lo >= hi) return;
    // the tensor that holds parameter lo (round 6: every thread tests its share of the boundaries -- ONE round trip; the binary search
    // by one thread was eight dependent ones before the block's first useful load)
    for (int i = threadIdx.x; i < ntensors; i += 256)
        if (offsets[i] <= lo && lo < offsets[i + 1]) first = i;
    __syncthreads();
    int t = first;
    float q = 0.f;
    while (lo < hi) {
        const long e = min(hi, offsets[t + 1]);
        if (e > lo) {
            float acc = 0.f;
            long i = lo + threadIdx.x;
            // (round 6: four table rows, then their up to sixteen gathers, requested before the first use -- the walk was two dependent
            //  round trips per parameter, eight parameters per thread: 54 us whatever the traffic, 281 MB or 122 MB)
            for (; i + 3 * 256 < e; i += 4 * 256) {
                int4 en[4];
                long dst[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { en[u] = tab[i + u * 256]; dst[u] = perm ? (long)perm[i + u * 256] : i + u * 256; }
                float g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = term(packed, en[u].x) + term(packed, en[u].y) + term(packed, en[u].z) + term(packed, en[u].w);
#pragma unroll
                for (int u = 0; u < 4; ++u) { grads[dst[u]] = g[u]; acc += g[u]; q += g[u] * g[u]; }
            }
            for (; i < e; i += 256) {
                const int4 en = tab[i];
                const float g = term(packed, en.x) + term(packed, en.y) + term(packed, en.z) + term(packed, en.w);
                grads[perm ? (long)perm[i] : i] = g;
                acc += g;
                q += g * g;
            }
            if (e - lo > 256) {        // (one atomic per block and tensor: per-wave atomics on a large tensor's ONE address cost more -- 61 vs 49 us)
                acc = block_sum<4>(acc, red);
                if (threadIdx.x == 0) atomicAdd(&tsums[t], acc);
            } else {                   // a run of small tensors (biases, BatchNorm / PReLU parameters): per wave, no block barriers
                acc = wave_sum(acc);
                if ((threadIdx.x & 63) == 0 && lo + (threadIdx.x & ~63) < e) atomicAdd(&tsums[t], acc);
            }
            lo = e;
        }
        ++t;
    }
    q = block_sum<4>(q, red);
    if (threadIdx.x == 0) atomicAdd(sumsq, (double)q);
}

// The common case of the above -- ONE entry per parameter -- with a 4-byte table entry instead of 16 (133.7 M parameters of Demucs:
// 0.54 GB of table instead of 2.1 GB per step); four parameters per thread.
__global__ __launch_bounds__(256) void unpack_grad1_kernel(const float* __restrict__ packed, const int* __restrict__ tab, long n,
                                                           float* __restrict__ grads) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int4 e = *reinterpret_cast<const int4*>(tab + 4 * i);
        *reinterpret_cast<float4*>(grads + 4 * i) = make_float4(term(packed, e.x), term(packed, e.y), term(packed, e.z), term(packed, e.w));
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) grads[i] = term(packed, tab[i]);
}
// ... and the parameters with several entries, by index: grads[list[i]] = sum of the 4 entries of tab[i]
__global__ __launch_bounds__(256) void unpack_grad_list_kernel(const float* __restrict__ packed, const int* __restrict__ list,
                                                               const int4* __restrict__ tab, long m, float* __restrict__ grads) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long)gridDim.x * 256) {
        const int4 e = tab[i];
        grads[list[i]] = term(packed, e.x) + term(packed, e.y) + term(packed, e.z) + term(packed, e.w);
    }
}

// Packing from run descriptors: 8 consecutive outputs = 8 parameters at base + j * stride (one int2 per 16 bytes of output instead
// of eight 4-byte indices: the layouts are permutations with long regular runs -- taps / channels swapped, transposes).
// run.x >= 0: base index, run.y = stride; run.x == -1: eight zeros (padding); run.x <= -2: irregular, the eight ordinary
// entries ((index << 1) | negate, -1 = absent) are at side[8 * (-2 - run.x)].
// dst (optional): the run a table entry produces, out[8 * dst[i] ..] -- the table may then be stored in ANY order.  Sorted by base
// address, the lanes of a wave read neighbouring parameters: a transposing layout (8 taps x 8 channels of a convolution weight:
// eight runs of stride 8 whose bases are consecutive) touches each 32-byte sector once per wave instruction instead of once per
// lane -- the table-ordered launch moved 8x the parameter bytes between L2 and L1 (Demucs: 2.4 ms per step for 134 M parameters).
__global__ __launch_bounds__(256) void pack_bf16_runs_kernel(const float* __restrict__ params, const int2* __restrict__ runs,
                                                             const int* __restrict__ dst, const int* __restrict__ side, long n8,
                                                             bf16_raw* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const int2 r = runs[i];
        const long o = dst ? (long)dst[i] : i;
        float v[8];
        if (r.x >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = params[(long)r.x + (long)j * r.y];
        } else if (r.x == -1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        } else {
            const int* e = side + 8L * (-2 - r.x);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = term(params, e[j]);
        }
        *reinterpret_cast<uint4*>(out + 8 * o) =
            make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
}

// The one-entry-per-parameter un-pack (Demucs) WITH the sums of the fused optimizer tail, as unpack_grad_sums_kernel: every block
// takes a contiguous range, walks the tensors inside it (16-byte table / gradient accesses where a tensor's part of the range allows,
// single elements at its unaligned ends) and leaves the per-tensor sums and the sum of squares in the optimizer's accumulators.
__global__ __launch_bounds__(256) void unpack_grad1_sums_kernel(const float* __restrict__ packed, const int* __restrict__ tab, long n,
                                                                float* __restrict__ grads, const long* __restrict__ offsets, int ntensors,
                                                                double* __restrict__ sumsq, float* __restrict__ tsums,
                                                                int* __restrict__ counter, const unsigned* __restrict__ guard) {
    __shared__ float red[4];
    __shared__ int first;
    if (blockIdx.x == 0 && threadIdx.x == 0 && counter && !(guard && guard[0] != 0u)) counter[0] += 1;
    const long per = ((n + gridDim.x - 1) / gridDim.x + 1023) / 1024 * 1024;
    long lo = (long)blockIdx.x * per;
    const long hi = min(n, lo + per);
    if (lo >= hi) return;
    if (threadIdx.x == 0) {          // the tensor that holds parameter lo: largest t with offsets[t] <= lo
        int a = 0, b = ntensors - 1;
        while (a < b) {
            const int mid = (a + b + 1) >> 1;
            if (offsets[mid] <= lo) a = mid; else b = mid - 1;
        }
        first = a;
    }
    __syncthreads();
    int t = first;
    float q = 0.f;
    while (lo < hi) {
        const long e = min(hi, offsets[t + 1]);
        if (e > lo) {
            float acc = 0.f;
            const long a4 = min(e, (lo + 3) & ~3L), b4 = max(a4, e & ~3L);       // [lo, a4) single, [a4, b4) by four, [b4, e) single
            for (long i = lo + threadIdx.x; i < a4; i += 256) { const float g = term(packed, tab[i]); grads[i] = g; acc += g; q += g * g; }
            for (long v = (a4 >> 2) + threadIdx.x; v < (b4 >> 2); v += 256) {
                const int4 en = *reinterpret_cast<const int4*>(tab + 4 * v);
                const float4 g = make_float4(term(packed, en.x), term(packed, en.y), term(packed, en.z), term(packed, en.w));
                *reinterpret_cast<float4*>(grads + 4 * v) = g;
                acc += (g.x + g.y) + (g.z + g.w);
                q += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
            }
            for (long i = b4 + threadIdx.x; i < e; i += 256) { const float g = term(packed, tab[i]); grads[i] = g; acc += g; q += g * g; }
            acc = block_sum<4>(acc, red);
            if (threadIdx.x == 0) atomicAdd(&tsums[t], acc);
            lo = e;
        }
        ++t;
    }
    q = block_sum<4>(q, red);
    if (threadIdx.x == 0) atomicAdd(sumsq, (double)q);
}
// ... and the parameters with several entries: the launch above left the FIRST entry's value in grads and in the sums; this one
// writes the whole sum and corrects the accumulators by the difference
__global__ __launch_bounds__(256) void unpack_grad_list_sums_kernel(const float* __restrict__ packed, const int* __restrict__ list,
                                                                    const int4* __restrict__ tab, long m, float* __restrict__ grads,
                                                                    const long* __restrict__ offsets, int ntensors,
                                                                    double* __restrict__ sumsq, float* __restrict__ tsums) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long)gridDim.x * 256) {
        const int4 e = tab[i];
        const float old = term(packed, e.x), g = old + term(packed, e.y) + term(packed, e.z) + term(packed, e.w);
        const long pi = list[i];
        grads[pi] = g;
        int a = 0, b = ntensors - 1;
        while (a < b) {
            const int mid = (a + b + 1) >> 1;
            if (offsets[mid] <= pi) a = mid; else b = mid - 1;
        }
        atomicAdd(&tsums[a], g - old);
        atomicAdd(sumsq, (double)g * g - (double)old * old);
    }
}

static int grid_of(long n) { long g = (n + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1; return (int)g; }

extern "C" int sehip_pack_bf16(const float* params, const int* table, long n, void* out_bf16, void* stream) {
    SEHIP_REQUIRE(n >= 0, "pack_bf16: negative size");
    if (n == 0) return 0;
    SEHIP_REQUIRE(((((uintptr_t)table) | ((uintptr_t)out_bf16)) & 15) == 0, "pack_bf16: table / output must be 16-byte aligned");
    pack_bf16_kernel<<<grid_of(n >> 3), 256, 0, (hipStream_t)stream>>>(params, table, n, (bf16_raw*)out_bf16);
    SEHIP_CHECK_LAUNCH("pack_bf16");
    return 0;
}

// sehip_pack_bf16 + sehip_pack_f32 + clearing nz floats at `zero` (may be NULL / 0) in ONE launch
extern "C" int sehip_pack_head(const float* params, const int* wtable, long nw, void* wout_bf16, const int* btable2, long nb, float* bout,
                               float* zero, long nz, void* stream) {
    SEHIP_REQUIRE(nw >= 0 && nb >= 0 && nz >= 0, "pack_head: negative size");
    SEHIP_REQUIRE(((((uintptr_t)wtable) | ((uintptr_t)wout_bf16)) & 15) == 0, "pack_head: table / output must be 16-byte aligned");
    if (nw + nb + nz == 0) return 0;
    pack_head_kernel<<<grid_of((nw >> 3) + 1), 256, 0, (hipStream_t)stream>>>(params, wtable, nw, (bf16_raw*)wout_bf16, (const int2*)btable2,
                                                                            nb, bout, zero, nz);
    SEHIP_CHECK_LAUNCH("pack_head");
    return 0;
}

extern "C" int sehip_pack_f32(const float* params, const int* table2, long n, float* out, void* stream) {
    SEHIP_REQUIRE(n >= 0, "pack_f32: negative size");
    if (n == 0) return 0;
    pack_f32_kernel<<<grid_of(n), 256, 0, (hipStream_t)stream>>>(params, (const int2*)table2, n, out);
    SEHIP_CHECK_LAUNCH("pack_f32");
    return 0;
}

extern "C" int sehip_unpack_grad(const float* packed, const int* table4, long n, float* grads, void* stream) {
    SEHIP_REQUIRE(n >= 0, "unpack_grad: negative size");
    if (n == 0) return 0;
    unpack_grad_kernel<<<grid_of(n), 256, 0, (hipStream_t)stream>>>(packed, (const int4*)table4, n, grads);
    SEHIP_CHECK_LAUNCH("unpack_grad");
    return 0;
}

extern "C" int sehip_unpack_grad_sums(const float* packed, const int* table4, long n, float* grads, const long* offsets, int ntensors,
                                      double* sumsq, float* tensor_sums, int* counter, const unsigned* guard, void* stream) {
    SEHIP_REQUIRE(n >= 0 && ntensors > 0 && offsets && sumsq && tensor_sums, "unpack_grad_sums: bad arguments");
    SEHIP_REQUIRE(!sehip_deterministic(), "unpack_grad_sums: not part of the deterministic schedule (fp32 / double atomics): use "
                                          "sehip_unpack_grad + sehip_grad_sumsq + sehip_grad_metric");
    if (n == 0) return 0;
    unpack_grad_sums_kernel<<<1024, 256, 0, (hipStream_t)stream>>>(packed, (const int4*)table4, n, grads, offsets, ntensors, sumsq,
                                                                  tensor_sums, counter, guard, nullptr);
    SEHIP_CHECK_LAUNCH("unpack_grad_sums");
    return 0;
}

extern "C" int sehip_unpack_grad_sums_perm(const float* packed, const int* table4, const int* perm, long n, float* grads,
                                           const long* offsets, int ntensors, double* sumsq, float* tensor_sums, int* counter,
                                           const unsigned* guard, void* stream) {
    SEHIP_REQUIRE(n >= 0 && ntensors > 0 && offsets && sumsq && tensor_sums && perm, "unpack_grad_sums_perm: bad arguments");
    SEHIP_REQUIRE(!sehip_deterministic(), "unpack_grad_sums_perm: not part of the deterministic schedule (fp32 / double atomics): use "
                                          "sehip_unpack_grad + sehip_grad_sumsq + sehip_grad_metric");
    if (n == 0) return 0;
    sehip_note_kernel("unpack_grad_sums_kernel<perm>");
    unpack_grad_sums_kernel<<<1024, 256, 0, (hipStream_t)stream>>>(packed, (const int4*)table4, n, grads, offsets, ntensors, sumsq,
                                                                  tensor_sums, counter, guard, perm);
    SEHIP_CHECK_LAUNCH("unpack_grad_sums_perm");
    return 0;
}

extern "C" int sehip_unpack_grad1(const float* packed, const int* table1, long n, float* grads, void* stream) {
    SEHIP_REQUIRE(n >= 0, "unpack_grad1: negative size");
    if (n == 0) return 0;
    SEHIP_REQUIRE(((((uintptr_t)table1) | ((uintptr_t)grads)) & 15) == 0, "unpack_grad1: table / gradients must be 16-byte aligned");
    unpack_grad1_kernel<<<grid_of(n >> 2), 256, 0, (hipStream_t)stream>>>(packed, table1, n, grads);
    SEHIP_CHECK_LAUNCH("unpack_grad1");
    return 0;
}

extern "C" int sehip_unpack_grad_list(const float* packed, const int* list, const int* table4, long m, float* grads, void* stream) {
    SEHIP_REQUIRE(m >= 0, "unpack_grad_list: negative size");
    if (m == 0) return 0;
    unpack_grad_list_kernel<<<grid_of(m), 256, 0, (hipStream_t)stream>>>(packed, list, (const int4*)table4, m, grads);
    SEHIP_CHECK_LAUNCH("unpack_grad_list");
    return 0;
}

// sehip_unpack_grad1 / sehip_unpack_grad_list over the WHOLE parameter vector with the sums of the fused optimizer tail
// (sehip_unpack_grad_sums for the one-entry tables of Demucs; same accumulators, same counter / guard semantics).  The list launch
// must follow the main one on the same stream.
extern "C" int sehip_unpack_grad1_sums(const float* packed, const int* table1, long n, float* grads, const long* offsets, int ntensors,
                                       double* sumsq, float* tensor_sums, int* counter, const unsigned* guard, void* stream) {
    SEHIP_REQUIRE(n >= 0 && ntensors > 0 && offsets && sumsq && tensor_sums, "unpack_grad1_sums: bad arguments");
    SEHIP_REQUIRE(!sehip_deterministic(), "unpack_grad1_sums: not part of the deterministic schedule (fp32 / double atomics)");
    SEHIP_REQUIRE(((((uintptr_t)table1) | ((uintptr_t)grads)) & 15) == 0, "unpack_grad1_sums: table / gradients must be 16-byte aligned");
    if (n == 0) return 0;
    unpack_grad1_sums_kernel<<<2048, 256, 0, (hipStream_t)stream>>>(packed, table1, n, grads, offsets, ntensors, sumsq, tensor_sums, counter, guard);
    SEHIP_CHECK_LAUNCH("unpack_grad1_sums");
    return 0;
}
extern "C" int sehip_unpack_grad_list_sums(const float* packed, const int* list, const int* table4, long m, float* grads, const long* offsets,
                                           int ntensors, double* sumsq, float* tensor_sums, void* stream) {
    SEHIP_REQUIRE(m >= 0 && ntensors > 0 && offsets && sumsq && tensor_sums, "unpack_grad_list_sums: bad arguments");
    if (m == 0) return 0;
    unpack_grad_list_sums_kernel<<<grid_of(m), 256, 0, (hipStream_t)stream>>>(packed, list, (const int4*)table4, m, grads, offsets, ntensors, sumsq,
                                                                              tensor_sums);
    SEHIP_CHECK_LAUNCH("unpack_grad_list_sums");
    return 0;
}

static int pack_runs_launch(const float* params, const int* runs2, const int* dst_run, const int* side, long n, void* out_bf16, void* stream) {
    SEHIP_REQUIRE(n >= 0 && (n & 7) == 0, "pack_bf16_runs: n=%ld must be a multiple of 8", n);
    if (n == 0) return 0;
    SEHIP_REQUIRE(((((uintptr_t)runs2) & 7) | (((uintptr_t)out_bf16) & 15)) == 0, "pack_bf16_runs: run table / output misaligned");
    // this packing runs on the second stream beside the step's dependent chain: 256 workgroups leave the chain its bandwidth
    // (Demucs: 24.15 ms per step at 4096 workgroups, 23.95 at 512, 23.88 at 256)
    static const int cap = getenv("SEHIP_PACK_WGS") ? atoi(getenv("SEHIP_PACK_WGS")) : 256;
    int g = grid_of(n >> 3);
    if (g > cap) g = cap;
    pack_bf16_runs_kernel<<<g, 256, 0, (hipStream_t)stream>>>(params, (const int2*)runs2, dst_run, side, n >> 3, (bf16_raw*)out_bf16);
    SEHIP_CHECK_LAUNCH("pack_bf16_runs");
    return 0;
}
extern "C" int sehip_pack_bf16_runs(const float* params, const int* runs2, const int* side, long n, void* out_bf16, void* stream) {
    return pack_runs_launch(params, runs2, nullptr, side, n, out_bf16, stream);
}
// the same with the run table in any order: entry i produces out_base[8 * dst_run[i] ..] (dst_run: absolute run numbers)
extern "C" int sehip_pack_bf16_runs_to(const float* params, const int* runs2, const int* dst_run, const int* side, long n, void* out_base,
                                       void* stream) {
    SEHIP_REQUIRE(dst_run != nullptr, "pack_bf16_runs_to: missing destination table");
    return pack_runs_launch(params, runs2, dst_run, side, n, out_base, stream);
}
