// Optimizer path on ONE flat fp32 parameter / gradient buffer
// (reference: src/solver.py:487-498 clip_grad_norm_ + optimizer.step + the sum-based grad_norm metric,
//  src/distrib.py:244-261 Adam/SGD construction).
//
//   1. sumsq:   total = sum g^2                         (double accumulator in HBM, no host sync)
//   2. adam:    coef = min(1, max_norm/(sqrt(total)+1e-6)); g *= coef (written back: the reference's
//               p.grad is the clipped gradient afterwards); m,v,p update with torch.optim.Adam's formula.
//   3. tensor_sums: per-parameter-tensor sum(g) for the reference's grad_norm metric
//               sqrt(sum_p (sum g_p)^2).
// All HBM-bound single passes over 2.07 M floats (DCCRN); nothing here touches the host.
#include <stdlib.h>
#include "common.h"

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
        float4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, (double)acc);
}

// mode 0: Adam (torch.optim.Adam, amsgrad False, weight_decay wd added to g as L2);
// mode 1: SGD with momentum (torch.optim.SGD, dampening 0, nesterov False), m = momentum buffer.
__global__ __launch_bounds__(256) void opt_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, const double* __restrict__ sumsq,
                                                       float max_norm, float lr, float b1, float b2, float eps, float bc1,
                                                       float sqrt_bc2, float wd, int mode, int first_step,
                                                       const int* __restrict__ step_dev, float gscale,
                                                       const unsigned* __restrict__ guard,
                                                       const float* __restrict__ tsums, int ntensors, float* __restrict__ metric,
                                                       double* __restrict__ next_sumsq, float* __restrict__ next_tsums) {
    // (the fused tail, sehip_opt_step_m: the accumulators of the NEXT step are cleared here -- the other one of two sets, which nobody
    //  reads during this step -- and workgroup 0 derives the logged metrics from the per-tensor sums sehip_unpack_grad_sums left)
    if (next_tsums && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < ntensors; i += 256) next_tsums[i] = 0.f;
        if (threadIdx.x == 0 && next_sumsq) next_sumsq[0] = 0.0;
    }
    if (metric && blockIdx.x == 0 && threadIdx.x < 64) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < ntensors; i += 64) acc += tsums[i] * tsums[i];
        acc = wave_sum(acc);
        if (threadIdx.x == 0) {
            float c = gscale;                                   // the sums are of the UNCLIPPED, unscaled gradient: scale them as the
            if (max_norm > 0.f) c = gscale * fminf(1.f, max_norm / ((float)sqrt(sumsq[0]) * gscale + 1e-6f));   // update below does
            metric[0] = sqrtf(acc) * c;
            metric[1] = (float)sqrt(sumsq[0]);
        }
    }
    if (guard && guard[0] != 0u) return;   // the step's gradients were declared invalid on the device: nothing is applied
    if (step_dev) {  // step counter lives on the device (graph replay): bias corrections computed here
        const int step = step_dev[0];
        first_step = step == 1;
        if (mode == 0) {
            bc1 = (float)(1.0 - pow((double)b1, (double)step));
            sqrt_bc2 = (float)sqrt(1.0 - pow((double)b2, (double)step));
        }
    }
    // gscale: 1/world of the data-parallel mean (the all-reduce SUMS the replicas' gradients); the norm that is clipped is the
    // norm of the scaled gradient, as the reference clips the reduced gradient (src/solver.py:487-490)
    float coef = gscale;
    if (max_norm > 0.f) {
        const float total = (float)sqrt(sumsq[0]) * gscale;
        coef = gscale * fminf(1.f, max_norm / (total + 1e-6f));
    }
    const float step_size = lr / bc1;
    // 16 bytes per lane and array (the four-byte form moved Demucs' 134 M parameters at 4.1 TB/s: 0.92 ms at the end of the step with
    // nothing beside it); the scaled gradient is written back only when the scale is not exactly 1 (unclipped single-replica steps:
    // one of the eight streams less)
    const bool wr_g = coef != 1.f;
    const long n4 = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) ? (n >> 2) : 0;
    auto upd = [&](float gi_in, float pi, float& mi, float& vi, float& gout) {
        float gi = gi_in * coef;
        gout = gi;
        if (wd != 0.f) gi += wd * pi;
        if (mode == 0) {
            mi = mi * b1 + (1.f - b1) * gi;
            vi = vi * b2 + (1.f - b2) * gi * gi;
            const float denom = sqrtf(vi) / sqrt_bc2 + eps;
            return pi - step_size * (mi / denom);
        }
        mi = first_step ? gi : (mi * b1 + gi);
        return pi - lr * (b1 != 0.f ? mi : gi);
    };
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
        float4 g4 = reinterpret_cast<const float4*>(g)[i], p4 = reinterpret_cast<const float4*>(p)[i];
        float4 m4 = reinterpret_cast<const float4*>(m)[i];
        float4 v4 = mode == 0 ? reinterpret_cast<const float4*>(v)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 go;
        p4.x = upd(g4.x, p4.x, m4.x, v4.x, go.x); p4.y = upd(g4.y, p4.y, m4.y, v4.y, go.y);
        p4.z = upd(g4.z, p4.z, m4.z, v4.z, go.z); p4.w = upd(g4.w, p4.w, m4.w, v4.w, go.w);
        if (wr_g) reinterpret_cast<float4*>(g)[i] = go;
        reinterpret_cast<float4*>(m)[i] = m4;
        if (mode == 0) reinterpret_cast<float4*>(v)[i] = v4;
        reinterpret_cast<float4*>(p)[i] = p4;
    }
    for (long i = 4 * n4 + blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {      // the last n % 4 (or everything, unaligned)
        float mi = m[i], vi = mode == 0 ? v[i] : 0.f, go;
        const float pn = upd(g[i], p[i], mi, vi, go);
        if (wr_g) g[i] = go;
        m[i] = mi;
        if (mode == 0) v[i] = vi;
        p[i] = pn;
    }
}

#define TS_CHUNK 8192
__global__ __launch_bounds__(256) void tensor_sums_kernel(const float* __restrict__ g, const long* __restrict__ offsets,
                                                          int ntensors, float* __restrict__ sums) {
    __shared__ float red[4];
    const int t = blockIdx.x;
    const long lo = offsets[t] + (long)blockIdx.y * TS_CHUNK;
    const long hi = min(offsets[t + 1], lo + TS_CHUNK);
    if (lo >= hi) return;
    float acc = 0.f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) acc += g[i];
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) atomicAdd(&sums[t], acc);
}

// The same over the FLAT gradient buffer: workgroup b sums a contiguous range and flushes at every tensor boundary inside it.  The
// grid above is (tensors x chunks of the LARGEST tensor): for Demucs (284 tensors, the largest 25 M elements) 870 000 workgroups,
// nearly all of them empty, and 3 000 atomics on the sum of each large tensor: 363 us for a 535-MB read.
__global__ __launch_bounds__(256) void tensor_sums_flat_kernel(const float* __restrict__ g, const long* __restrict__ offsets,
                                                               int ntensors, float* __restrict__ sums) {
    __shared__ float red[4];
    __shared__ int first;
    const long begin = offsets[0], end = offsets[ntensors];
    const long per = (((end - begin) + gridDim.x - 1) / gridDim.x + 1023) / 1024 * 1024;
    long lo = begin + (long)blockIdx.x * per;
    const long hi = min(end, lo + per);
    if (lo >= hi) return;
    if (threadIdx.x == 0) {          // the tensor that holds element lo: largest t with offsets[t] <= lo
        int a = 0, b = ntensors - 1;
        while (a < b) {
            const int mid = (a + b + 1) >> 1;
            if (offsets[mid] <= lo) a = mid; else b = mid - 1;
        }
        first = a;
    }
    __syncthreads();
    int t = first;
    while (lo < hi) {
        const long e = min(hi, offsets[t + 1]);
        if (e > lo) {
            float acc = 0.f;
            for (long i = (lo & ~3L) + threadIdx.x * 4L; i < e; i += 1024) {          // 16-byte aligned quads; ragged ends by element
                if (i >= lo && i + 4 <= e) {
                    const float4 v = *reinterpret_cast<const float4*>(g + i);
                    acc += (v.x + v.y) + (v.z + v.w);
                } else {
                    for (long k = max(i, lo); k < min(e, i + 4); ++k) acc += g[k];
                }
            }
            acc = block_sum<4>(acc, red);
            if (threadIdx.x == 0) atomicAdd(&sums[t], acc);
            lo = e;
        }
        ++t;
    }
}

// metric[0] = sqrt(sum_t sums[t]^2)  (src/solver.py:494-498); metric[1] = sqrt(sumsq) (pre-clip L2 norm)
__global__ void grad_metric_kernel(const float* __restrict__ sums, int ntensors, const double* __restrict__ sumsq,
                                   float* __restrict__ metric) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < ntensors; i += 64) acc += sums[i] * sums[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) {
        metric[0] = sqrtf(acc);
        metric[1] = sumsq ? (float)sqrt(sumsq[0]) : 0.f;
    }
}

static int grad_sumsq_impl(const float* grads, long n, double* sumsq_out, bool clear, void* stream) {
    SEHIP_REQUIRE(n >= 0, "grad_sumsq: negative size");
    hipStream_t st = (hipStream_t)stream;
    if (clear) {
        hipError_t e = hipMemsetAsync(sumsq_out, 0, sizeof(double), st);
        SEHIP_REQUIRE(e == hipSuccess, "grad_sumsq: memset failed: %s", hipGetErrorString(e));
    }
    if (n == 0) return 0;
    SEHIP_REQUIRE((((uintptr_t)grads) & 15) == 0, "grad_sumsq: gradient buffer must be 16-byte aligned");
    int grid = cdiv(n, 256 * 16);
    if (grid > 1024) grid = 1024;
    if (sehip_deterministic()) grid = 1;          // one workgroup: one contribution, a fixed order of additions
    sumsq_kernel<<<grid, 256, 0, st>>>(grads, n, sumsq_out);
    SEHIP_CHECK_LAUNCH("grad_sumsq");
    return 0;
}

extern "C" int sehip_grad_sumsq(const float* grads, long n, double* sumsq_out, void* stream) {
    return grad_sumsq_impl(grads, n, sumsq_out, true, stream);
}
// the same without clearing the accumulator (sehip_opt_begin has done it)
extern "C" int sehip_grad_sumsq_acc(const float* grads, long n, double* sumsq_out, void* stream) {
    return grad_sumsq_impl(grads, n, sumsq_out, false, stream);
}

__global__ void counter_add_kernel(int* p, int v) { p[0] += v; }

// One launch in front of the optimizer kernels: step counter += value, sumsq = 0, tensor_sums[0 .. ntensors) = 0.  (Each of the
// three was a launch of its own -- two of them runtime memsets, each a 20-25 us bubble on the step's dependent chain.)
__global__ void opt_begin_kernel(int* counter, int value, double* sumsq, float* tensor_sums, int ntensors, const unsigned* guard) {
    if (threadIdx.x == 0) {
        if (counter && !(guard && guard[0] != 0u)) counter[0] += value;   // a guarded-out step does not count
        if (sumsq) sumsq[0] = 0.0;
    }
    if (tensor_sums)
        for (int i = threadIdx.x; i < ntensors; i += blockDim.x) tensor_sums[i] = 0.f;
}
extern "C" int sehip_opt_begin_g(int* counter, int value, double* sumsq, float* tensor_sums, int ntensors, const unsigned* guard,
                                 void* stream) {
    SEHIP_REQUIRE(ntensors >= 0, "opt_begin: negative tensor count");
    opt_begin_kernel<<<1, 256, 0, (hipStream_t)stream>>>(counter, value, sumsq, tensor_sums, ntensors, guard);
    SEHIP_CHECK_LAUNCH("opt_begin");
    return 0;
}
extern "C" int sehip_opt_begin(int* counter, int value, double* sumsq, float* tensor_sums, int ntensors, void* stream) {
    return sehip_opt_begin_g(counter, value, sumsq, tensor_sums, ntensors, nullptr, stream);
}

// Up to four buffers cleared by ONE launch (a train step's accumulators -- packed gradients, normalisation sums, an overlap-add
// output -- were four runtime fill kernels with their host gaps on ConvTasNet's chain: round 6).  Sizes in bytes, multiples of 4;
// 16-byte aligned pointers (whole allocations).
struct ZeroRegions { float* p[4]; long n[4]; };
__global__ __launch_bounds__(256) void zero_regions_kernel(ZeroRegions r) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float* p = r.p[k];
        const long n = r.n[k], n4 = n >> 2;
        if (!p) continue;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
            reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (blockIdx.x == 0 && (long)threadIdx.x < n - (n4 << 2)) p[(n4 << 2) + threadIdx.x] = 0.f;
    }
}
extern "C" int sehip_zero_regions(void* p0, long b0, void* p1, long b1, void* p2, long b2, void* p3, long b3, void* stream) {
    ZeroRegions r;
    void* ps[4] = {p0, p1, p2, p3};
    const long bs[4] = {b0, b1, b2, b3};
    long most = 0;
    for (int k = 0; k < 4; ++k) {
        SEHIP_REQUIRE(bs[k] >= 0 && (bs[k] & 3) == 0 && (reinterpret_cast<size_t>(ps[k]) & 15) == 0,
                      "zero_regions: region %d needs a 16-byte aligned pointer and a multiple of 4 bytes (%ld)", k, bs[k]);
        r.p[k] = bs[k] > 0 ? reinterpret_cast<float*>(ps[k]) : nullptr;
        r.n[k] = bs[k] >> 2;
        if (r.p[k] && r.n[k] > most) most = r.n[k];
    }
    if (most == 0) return 0;
    long g = (most / 4 + 255) / 256;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    zero_regions_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(r);
    SEHIP_CHECK_LAUNCH("zero_regions");
    return 0;
}

extern "C" int sehip_counter_add(int* counter, int value, void* stream) {
    SEHIP_REQUIRE(counter != nullptr, "counter_add: null counter");
    counter_add_kernel<<<1, 1, 0, (hipStream_t)stream>>>(counter, value);
    SEHIP_CHECK_LAUNCH("counter_add");
    return 0;
}

extern "C" int sehip_opt_step_g(float* params, float* grads, float* m, float* v, long n, const double* sumsq,
                                float max_norm, float lr, float beta1, float beta2, float eps, int step, const int* step_dev,
                                float weight_decay, int mode, float grad_scale, const unsigned* guard, void* stream) {
    SEHIP_REQUIRE(n >= 0 && (step >= 1 || step_dev != nullptr), "opt_step: bad n/step (n=%ld step=%d)", n, step);
    if (step < 1) step = 1;
    SEHIP_REQUIRE(mode == 0 || mode == 1, "opt_step: mode must be 0 (adam) or 1 (sgd)");
    SEHIP_REQUIRE(grad_scale > 0.f, "opt_step: grad_scale must be positive (1 for a single replica, 1/world after the all-reduce)");
    SEHIP_REQUIRE(max_norm <= 0.f || sumsq != nullptr, "opt_step: clipping needs the sumsq buffer");
    if (n == 0) return 0;
    const double bc1 = mode == 0 ? 1.0 - pow((double)beta1, step) : 1.0;
    const double bc2 = mode == 0 ? 1.0 - pow((double)beta2, step) : 1.0;
    int grid = cdiv(n, 256 * 16);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    opt_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(params, grads, m, v, n, sumsq, max_norm, lr, beta1, beta2, eps,
                                                           (float)bc1, (float)sqrt(bc2), weight_decay, mode, step == 1, step_dev,
                                                           grad_scale, guard, nullptr, 0, nullptr, nullptr, nullptr);
    SEHIP_CHECK_LAUNCH("opt_step");
    return 0;
}
// sehip_opt_step_g for a step whose gradients came through sehip_unpack_grad_sums (sumsq and tensor_sums hold this step's sums): the
// same update, plus metric[0] = the reference's sqrt(sum_t (sum g_t)^2) of the CLIPPED gradient and metric[1] = sqrt(sumsq)
// (sehip_grad_metric's outputs), plus the clearing of next_sumsq / next_tensor_sums (the set the next step will add to; may be
// NULL).  No sehip_opt_begin, sehip_grad_sumsq or sehip_grad_metric launch is needed around it.
extern "C" int sehip_opt_step_m(float* params, float* grads, float* m, float* v, long n, const double* sumsq,
                                float max_norm, float lr, float beta1, float beta2, float eps, int step, const int* step_dev,
                                float weight_decay, int mode, float grad_scale, const unsigned* guard, const float* tensor_sums,
                                int ntensors, float* metric, double* next_sumsq, float* next_tensor_sums, void* stream) {
    SEHIP_REQUIRE(n > 0 && (step >= 1 || step_dev != nullptr), "opt_step_m: bad n/step (n=%ld step=%d)", n, step);
    if (step < 1) step = 1;
    SEHIP_REQUIRE(mode == 0 || mode == 1, "opt_step_m: mode must be 0 (adam) or 1 (sgd)");
    SEHIP_REQUIRE(grad_scale > 0.f && sumsq && tensor_sums && metric && ntensors > 0, "opt_step_m: bad arguments");
    const double bc1 = mode == 0 ? 1.0 - pow((double)beta1, step) : 1.0;
    const double bc2 = mode == 0 ? 1.0 - pow((double)beta2, step) : 1.0;
    int grid = cdiv(n, 256 * 16);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    opt_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(params, grads, m, v, n, sumsq, max_norm, lr, beta1, beta2, eps,
                                                           (float)bc1, (float)sqrt(bc2), weight_decay, mode, step == 1, step_dev,
                                                           grad_scale, guard, tensor_sums, ntensors, metric, next_sumsq, next_tensor_sums);
    SEHIP_CHECK_LAUNCH("opt_step_m");
    return 0;
}
extern "C" int sehip_opt_step(float* params, float* grads, float* m, float* v, long n, const double* sumsq,
                              float max_norm, float lr, float beta1, float beta2, float eps, int step, const int* step_dev,
                              float weight_decay, int mode, float grad_scale, void* stream) {
    return sehip_opt_step_g(params, grads, m, v, n, sumsq, max_norm, lr, beta1, beta2, eps, step, step_dev, weight_decay, mode,
                            grad_scale, nullptr, stream);
}

static int grad_metric_impl(const float* grads, const long* offsets, int ntensors, long max_tensor, const double* sumsq,
                            float* tensor_sums, float* metric, bool clear, void* stream) {
    SEHIP_REQUIRE(ntensors > 0 && max_tensor > 0, "grad_metric: no tensors");
    hipStream_t st = (hipStream_t)stream;
    if (clear) {
        hipError_t e = hipMemsetAsync(tensor_sums, 0, sizeof(float) * ntensors, st);
        SEHIP_REQUIRE(e == hipSuccess, "grad_metric: memset failed: %s", hipGetErrorString(e));
    }
    static const bool chunked = getenv("SEHIP_TENSOR_SUMS_CHUNKED") != nullptr;
    if (chunked && !sehip_deterministic()) tensor_sums_kernel<<<dim3(ntensors, cdiv(max_tensor, TS_CHUNK)), 256, 0, st>>>(grads, offsets, ntensors, tensor_sums);
    else tensor_sums_flat_kernel<<<sehip_deterministic() ? 1 : 1024, 256, 0, st>>>(grads, offsets, ntensors, tensor_sums);
    grad_metric_kernel<<<1, 64, 0, st>>>(tensor_sums, ntensors, sumsq, metric);
    SEHIP_CHECK_LAUNCH("grad_metric");
    return 0;
}

extern "C" int sehip_grad_metric(const float* grads, const long* offsets, int ntensors, long max_tensor, const double* sumsq,
                                 float* tensor_sums, float* metric, void* stream) {
    return grad_metric_impl(grads, offsets, ntensors, max_tensor, sumsq, tensor_sums, metric, true, stream);
}
// the same without clearing tensor_sums (sehip_opt_begin has done it)
extern "C" int sehip_grad_metric_acc(const float* grads, const long* offsets, int ntensors, long max_tensor, const double* sumsq,
                                     float* tensor_sums, float* metric, void* stream) {
    return grad_metric_impl(grads, offsets, ntensors, max_tensor, sumsq, tensor_sums, metric, false, stream);
}

// ---- the deterministic schedule's second launch (csrc/det.h): one wave per (group, value); lane l adds the slots l, l + 64, ... in
// increasing order, the lane sums meet in a fixed xor tree
#include "det.h"
__global__ __launch_bounds__(64) void det_sum_slots_kernel(const double* __restrict__ part, int nb, int nvals, double* __restrict__ dst,
                                                          long dst_stride) {
    const int g = blockIdx.x, i = blockIdx.y, lane = threadIdx.x;
    const double* base = part + (size_t)g * nb * nvals + i;
    double s = 0.0;
    for (int b = lane; b < nb; b += 64) s += base[(size_t)b * nvals];
    s = wave_sum_d(s);
    if (lane == 0) dst[(size_t)g * dst_stride + i] += s;
}
int sehip_det_finish(hipStream_t st, DetCtx dc, int ngroups, int nb, int nvals, double* dst, long dst_stride) {
    if (dc.part == nullptr) return 0;
    det_sum_slots_kernel<<<dim3((unsigned)ngroups, (unsigned)nvals), 64, 0, st>>>(dc.part, nb, nvals, dst, dst_stride);
    SEHIP_CHECK_LAUNCH("det_finish");
    return 0;
}
