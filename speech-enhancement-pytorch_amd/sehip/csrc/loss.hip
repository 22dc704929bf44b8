// SI-SNR loss forward/backward (reference: src/loss.py:14-29 -- l2_norm, si_snr, loss_sisdr).
//
//   s_t   = <x,s>/(<s,s>+eps) * s          (no zero-mean, as in the reference)
//   e     = x - s_t
//   l_row = 10*log10( |s_t|^2 / (|e|^2 + eps) + eps )
//   loss  = -mean_rows(l_row)
//
// HBM-bound: one 256-thread workgroup per utterance row; the row is read twice (dot products, then the
// exact residual energy -- the second read hits L2), the backward is one fused a*x+b*s pass whose two row
// coefficients were produced by the forward.
#include "common.h"

#define EPS 1e-8f

// rowstat[r] = {a, b, l_row, unused}; d loss / d x_j = upstream * (a*x_j + b*s_j)
// (1024 threads: a row is two dependent passes of one workgroup, 32 rows per step -- latency, not bandwidth: 23.6 us with 256 threads)
__global__ __launch_bounds__(1024) void sisnr_fwd_kernel(const float* __restrict__ est, const float* __restrict__ ref,
                                                        int n, int rows, float4* __restrict__ rowstat) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    const float* x = est + (size_t)r * n;
    const float* s = ref + (size_t)r * n;
    float xs = 0.f, ss = 0.f;
    const int n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* s4 = reinterpret_cast<const float4*>(s);
    const bool vec = ((n & 3) == 0) && ((((uintptr_t)x | (uintptr_t)s) & 15) == 0);
    if (vec) {
        for (int i = threadIdx.x; i < n4; i += 1024) {
            float4 a = x4[i], b = s4[i];
            xs += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            ss += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 1024) { xs += x[i] * s[i]; ss += s[i] * s[i]; }
    }
    xs = block_sum<16>(xs, red);
    ss = block_sum<16>(ss, red);
    const float alpha = xs / (ss + EPS);
    float et = 0.f, en = 0.f, es = 0.f;
    if (vec) {
        for (int i = threadIdx.x; i < n4; i += 1024) {
            float4 a = x4[i], b = s4[i];
            float t0 = alpha * b.x, t1 = alpha * b.y, t2 = alpha * b.z, t3 = alpha * b.w;
            float e0 = a.x - t0, e1 = a.y - t1, e2 = a.z - t2, e3 = a.w - t3;
            et += t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3;
            en += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
            es += e0 * b.x + e1 * b.y + e2 * b.z + e3 * b.w;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 1024) {
            float t = alpha * s[i], e = x[i] - t;
            et += t * t; en += e * e; es += e * s[i];
        }
    }
    et = block_sum<16>(et, red);
    en = block_sum<16>(en, red);
    es = block_sum<16>(es, red);
    if (threadIdx.x == 0) {
        const float ratio = et / (en + EPS);
        const float l = 10.f * log10f(ratio + EPS);
        // d l / d ratio, with the -1/rows of loss = -mean folded in
        const float c = -(1.0f / rows) * 10.f / (2.302585092994046f * (ratio + EPS));
        const float q = et / ((en + EPS) * (en + EPS));
        const float a = c * (-2.f * q);
        const float b = c * (2.f * alpha * ss / ((ss + EPS) * (en + EPS)) + 2.f * q * alpha + 2.f * q * es / (ss + EPS));
        rowstat[r] = make_float4(a, b, l, alpha);
    }
}

__global__ void sisnr_finalize_kernel(const float4* __restrict__ rowstat, int rows, float* __restrict__ loss) {
    // single wave, deterministic order
    float acc = 0.f;
    for (int i = threadIdx.x; i < rows; i += 64) acc += rowstat[i].z;
    acc = wave_sum(acc);
    if (threadIdx.x == 0) loss[0] = -acc / rows;
}

__global__ __launch_bounds__(256) void sisnr_bwd_kernel(const float* __restrict__ est, const float* __restrict__ ref,
                                                        const float4* __restrict__ rowstat, const float* __restrict__ upstream,
                                                        int n, float* __restrict__ dest) {
    const int r = blockIdx.y;
    const float4 st = rowstat[r];
    const float up = upstream ? upstream[0] : 1.f;
    const float a = st.x * up, b = st.y * up;
    const size_t base = (size_t)r * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        dest[base + i] = a * est[base + i] + b * ref[base + i];
}

// SI-SDR validation metric (src/metric.py:92-123, SI_SDR): per row alpha = <e,r> / (<r,r> + eps), ratio = |alpha r|^2 /
// (|e - alpha r|^2 + eps); result 10 log10(mean_rows(ratio) + eps) with eps = float32 machine epsilon (np.finfo of the
// reference's float32 arrays).  Same two-pass row kernel as the loss; ratios[rows] is scratch.
#define SDR_EPS 1.1920929e-07f
__global__ __launch_bounds__(256) void sisdr_metric_kernel(const float* __restrict__ ref, const float* __restrict__ est, int n,
                                                           float* __restrict__ ratios) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float* s = ref + (size_t)r * n;
    const float* x = est + (size_t)r * n;
    float xs = 0.f, ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { xs += x[i] * s[i]; ss += s[i] * s[i]; }
    xs = block_sum<4>(xs, red);
    ss = block_sum<4>(ss, red);
    const float alpha = xs / (ss + SDR_EPS);
    float et = 0.f, en = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float t = alpha * s[i], e = x[i] - t;
        et += t * t; en += e * e;
    }
    et = block_sum<4>(et, red);
    en = block_sum<4>(en, red);
    if (threadIdx.x == 0) ratios[r] = et / (en + SDR_EPS);
}

__global__ void sisdr_metric_finalize_kernel(const float* __restrict__ ratios, int rows, float* __restrict__ out) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < rows; i += 64) acc += ratios[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[0] = 10.f * log10f(acc / rows + SDR_EPS);
}

extern "C" int sehip_sisdr_metric(const float* reference, const float* estimation, int rows, int n, float* ratios, float* out,
                                  void* stream) {
    SEHIP_REQUIRE(rows > 0 && n > 0, "sisdr_metric: empty input (rows=%d n=%d)", rows, n);
    hipStream_t st = (hipStream_t)stream;
    sisdr_metric_kernel<<<rows, 256, 0, st>>>(reference, estimation, n, ratios);
    sisdr_metric_finalize_kernel<<<1, 64, 0, st>>>(ratios, rows, out);
    SEHIP_CHECK_LAUNCH("sisdr_metric");
    return 0;
}

extern "C" int sehip_sisnr_fwd(const float* est, const float* ref, int rows, int n, float* rowstat, float* loss,
                               void* stream) {
    SEHIP_REQUIRE(rows > 0 && n > 0, "sisnr_fwd: empty input (rows=%d n=%d)", rows, n);
    hipStream_t st = (hipStream_t)stream;
    sisnr_fwd_kernel<<<rows, 1024, 0, st>>>(est, ref, n, rows, (float4*)rowstat);
    sisnr_finalize_kernel<<<1, 64, 0, st>>>((const float4*)rowstat, rows, loss);
    SEHIP_CHECK_LAUNCH("sisnr_fwd");
    return 0;
}

extern "C" int sehip_sisnr_bwd(const float* est, const float* ref, const float* rowstat, const float* upstream, int rows,
                               int n, float* dest, void* stream) {
    SEHIP_REQUIRE(rows > 0 && n > 0, "sisnr_bwd: empty input (rows=%d n=%d)", rows, n);
    dim3 grid(cdiv(n, 256 * 8), rows);
    sisnr_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(est, ref, (const float4*)rowstat, upstream, n, dest);
    SEHIP_CHECK_LAUNCH("sisnr_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Permutation-invariant SI-SNR: src/loss.py:58-100 (UtterenceBaasedPermutationInvariantTraining with loss_function =
// loss_sisdr).  est / ref are [B][S][C][n] (speaker axis 1).  The reference fills an S x S matrix with the BATCH-mean loss of
// (estimated speaker i, target speaker j) (:70-71), takes the permutation with the smallest sum, walking
// itertools.permutations in its lexicographic order with a strict '<' (:73-86), and returns the mean of the chosen pairs'
// losses (:88-96).  One launch computes the S*S*B*C row records, one single-wave launch the matrix, the permutation and the
// loss; the backward pass reads the permutation from device memory (no host round trip).
//   rowstat [S*S][R] float4 (R = B*C rows per pair; pair index i*S + j), pairloss [S*S], perm [S] (perm[j] = estimated speaker
//   matched with target j), loss [1]
// ------------------------------------------------------------------------------------------------
#define PIT_MAXS 6
__global__ __launch_bounds__(256) void sisnr_pit_rows_kernel(const float* __restrict__ est, const float* __restrict__ ref, int B, int S,
                                                             int C, int n, float4* __restrict__ rowstat) {
    __shared__ float red[4];
    const int R = B * C;
    const int r = blockIdx.x, pair = blockIdx.y;
    const int i = pair / S, j = pair - i * S;
    const int b = r / C, c = r - b * C;
    const float* x = est + (((size_t)b * S + i) * C + c) * n;
    const float* s = ref + (((size_t)b * S + j) * C + c) * n;
    float xs = 0.f, ss = 0.f;
    for (int k = threadIdx.x; k < n; k += 256) { xs += x[k] * s[k]; ss += s[k] * s[k]; }
    xs = block_sum<4>(xs, red);
    ss = block_sum<4>(ss, red);
    const float alpha = xs / (ss + EPS);
    float et = 0.f, en = 0.f, es = 0.f;
    for (int k = threadIdx.x; k < n; k += 256) {
        const float t = alpha * s[k], e = x[k] - t;
        et += t * t; en += e * e; es += e * s[k];
    }
    et = block_sum<4>(et, red);
    en = block_sum<4>(en, red);
    es = block_sum<4>(es, red);
    if (threadIdx.x == 0) {
        const float ratio = et / (en + EPS);
        const float l = 10.f * log10f(ratio + EPS);
        // loss = (1/S) sum over chosen pairs of -mean_rows(l): d loss / d l = -1 / (S R)
        const float cc = -(1.0f / ((float)S * (float)R)) * 10.f / (2.302585092994046f * (ratio + EPS));
        const float q = et / ((en + EPS) * (en + EPS));
        const float a = cc * (-2.f * q);
        const float bb = cc * (2.f * alpha * ss / ((ss + EPS) * (en + EPS)) + 2.f * q * alpha + 2.f * q * es / (ss + EPS));
        rowstat[(size_t)pair * R + r] = make_float4(a, bb, l, alpha);
    }
}

__global__ void sisnr_pit_select_kernel(const float4* __restrict__ rowstat, int R, int S, float* __restrict__ pairloss,
                                        int* __restrict__ perm, float* __restrict__ loss) {
    __shared__ float m[PIT_MAXS * PIT_MAXS];
    for (int pair = 0; pair < S * S; ++pair) {            // single wave, fixed summation order
        float acc = 0.f;
        for (int r = threadIdx.x; r < R; r += 64) acc += rowstat[(size_t)pair * R + r].z;
        acc = wave_sum(acc);
        if (threadIdx.x == 0) { m[pair] = -acc / R; pairloss[pair] = -acc / R; }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int cur[PIT_MAXS], best[PIT_MAXS];
    for (int k = 0; k < S; ++k) cur[k] = best[k] = k;
    float lmin = 1e9f;
    for (;;) {
        float l = 0.f;
        for (int j = 0; j < S; ++j) l += m[cur[j] * S + j];
        if (lmin > l) {
            lmin = l;
            for (int k = 0; k < S; ++k) best[k] = cur[k];
        }
        // next permutation in lexicographic order (the order of itertools.permutations(range(S)))
        int k = S - 2;
        while (k >= 0 && cur[k] > cur[k + 1]) --k;
        if (k < 0) break;
        int l2 = S - 1;
        while (cur[l2] < cur[k]) --l2;
        int t = cur[k]; cur[k] = cur[l2]; cur[l2] = t;
        for (int a = k + 1, b = S - 1; a < b; ++a, --b) { t = cur[a]; cur[a] = cur[b]; cur[b] = t; }
    }
    float tot = 0.f;
    for (int j = 0; j < S; ++j) { perm[j] = best[j]; tot += m[best[j] * S + j]; }
    loss[0] = tot / S;
}

__global__ __launch_bounds__(256) void sisnr_pit_bwd_kernel(const float* __restrict__ est, const float* __restrict__ ref,
                                                            const float4* __restrict__ rowstat, const int* __restrict__ perm,
                                                            const float* __restrict__ upstream, int B, int S, int C, int n,
                                                            float* __restrict__ dest) {
    const int R = B * C;
    const int r = blockIdx.y, i = blockIdx.z;
    int j = 0;
    for (int k = 0; k < S; ++k)
        if (perm[k] == i) j = k;                           // the target this estimated speaker was matched with
    const int b = r / C, c = r - b * C;
    const float4 st = rowstat[(size_t)(i * S + j) * R + r];
    const float up = upstream ? upstream[0] : 1.f;
    const float a = st.x * up, bb = st.y * up;
    const size_t xb = (((size_t)b * S + i) * C + c) * n, sb = (((size_t)b * S + j) * C + c) * n;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) dest[xb + k] = a * est[xb + k] + bb * ref[sb + k];
}

extern "C" int sehip_sisnr_pit_fwd(const float* est, const float* ref, int B, int S, int C, int n, float* rowstat, float* pairloss,
                                   int* perm, float* loss, void* stream) {
    SEHIP_REQUIRE(B > 0 && C > 0 && n > 0 && S >= 1 && S <= PIT_MAXS, "sisnr_pit_fwd: bad shape (B=%d S=%d C=%d n=%d; S <= %d)", B, S,
                  C, n, PIT_MAXS);
    hipStream_t st = (hipStream_t)stream;
    sisnr_pit_rows_kernel<<<dim3(B * C, S * S), 256, 0, st>>>(est, ref, B, S, C, n, (float4*)rowstat);
    sisnr_pit_select_kernel<<<1, 64, 0, st>>>((const float4*)rowstat, B * C, S, pairloss, perm, loss);
    SEHIP_CHECK_LAUNCH("sisnr_pit_fwd");
    return 0;
}

extern "C" int sehip_sisnr_pit_bwd(const float* est, const float* ref, const float* rowstat, const int* perm, const float* upstream,
                                   int B, int S, int C, int n, float* dest, void* stream) {
    SEHIP_REQUIRE(B > 0 && C > 0 && n > 0 && S >= 1 && S <= PIT_MAXS, "sisnr_pit_bwd: bad shape (B=%d S=%d C=%d n=%d)", B, S, C, n);
    dim3 grid(cdiv(n, 256 * 8), B * C, S);
    sisnr_pit_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(est, ref, (const float4*)rowstat, perm, upstream, B, S, C, n, dest);
    SEHIP_CHECK_LAUNCH("sisnr_pit_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// l1 / mse (the reference uses torch.nn.functional.l1_loss / mse_loss with reduction 'mean', src/distrib.py:263-268)
// mode 0: mean |x - y|     mode 1: mean (x - y)^2
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pointwise_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long n,
                                                                 int mode, double* __restrict__ acc) {
    __shared__ float red[4];
    float a = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const float d = x[i] - y[i];
        a += mode == 0 ? fabsf(d) : d * d;
    }
    a = block_sum<4>(a, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)a);
}

__global__ void pointwise_loss_finalize_kernel(const double* __restrict__ acc, long n, float* __restrict__ loss) {
    loss[0] = (float)(acc[0] / (double)n);
}

__global__ __launch_bounds__(256) void pointwise_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long n,
                                                                 int mode, const float* __restrict__ upstream,
                                                                 float* __restrict__ dx) {
    const float up = (upstream ? upstream[0] : 1.f) / (float)n;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const float d = x[i] - y[i];
        dx[i] = mode == 0 ? (d > 0.f ? up : (d < 0.f ? -up : 0.f)) : 2.f * d * up;
    }
}

// ------------------------------------------------------------------------------------------------
// phase-sensitive spectral approximation (src/loss.py:32-56, `optim.loss: psa`, STFT-domain models; the Solver hands the mixture's
// spectrum as the third argument, src/solver.py:480).  Per complex element (last axis = (real, imaginary)):
//   d = |E| - |T| cos(tanh(Ti / (Tr + 1e-9)) - tanh(Mi / (Mr + 1e-9)));   loss = mean d^2
// (sic: tanh of the ratio, not atan -- the reference's formula is restated, not corrected).  Backward w.r.t. the enhanced spectrum:
// dE = upstream * 2 d / n * E / |E|; an element with |E| = 0 gets 0 (torch's sqrt backward gives NaN there: 0 * inf).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float psa_residual(float2 e, float2 t, float2 m, float& ae) {
    const float am = tanhf(m.y / (m.x + 1e-9f)), at = tanhf(t.y / (t.x + 1e-9f));
    ae = sqrtf(e.y * e.y + e.x * e.x);
    return ae - sqrtf(t.y * t.y + t.x * t.x) * cosf(at - am);
}

__global__ __launch_bounds__(256) void psa_loss_fwd_kernel(const float2* __restrict__ enh, const float2* __restrict__ tgt,
                                                           const float2* __restrict__ mix, long n, double* __restrict__ acc) {
    __shared__ float red[4];
    float a = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        float ae;
        const float d = psa_residual(enh[i], tgt[i], mix[i], ae);
        a += d * d;
    }
    a = block_sum<4>(a, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)a);
}

__global__ __launch_bounds__(256) void psa_loss_bwd_kernel(const float2* __restrict__ enh, const float2* __restrict__ tgt,
                                                           const float2* __restrict__ mix, long n, const float* __restrict__ upstream,
                                                           float2* __restrict__ denh) {
    const float up = 2.f * (upstream ? upstream[0] : 1.f) / (float)n;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const float2 e = enh[i];
        float ae;
        const float d = psa_residual(e, tgt[i], mix[i], ae);
        const float k = ae > 0.f ? up * d / ae : 0.f;
        denh[i] = make_float2(k * e.x, k * e.y);
    }
}

extern "C" int sehip_psa_loss_fwd(const float* enh, const float* tgt, const float* mix, long ncomplex, double* acc, float* loss, void* stream) {
    SEHIP_REQUIRE(ncomplex > 0 && enh && tgt && mix, "psa_loss_fwd: bad arguments (n=%ld)", ncomplex);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(double), st);
    SEHIP_REQUIRE(e == hipSuccess, "psa_loss_fwd: memset failed: %s", hipGetErrorString(e));
    int grid = cdiv(ncomplex, 256 * 8);
    if (grid > 1024) grid = 1024;
    if (sehip_deterministic()) grid = 1;
    psa_loss_fwd_kernel<<<grid, 256, 0, st>>>((const float2*)enh, (const float2*)tgt, (const float2*)mix, ncomplex, acc);
    pointwise_loss_finalize_kernel<<<1, 1, 0, st>>>(acc, ncomplex, loss);
    SEHIP_CHECK_LAUNCH("psa_loss_fwd");
    return 0;
}

extern "C" int sehip_psa_loss_bwd(const float* enh, const float* tgt, const float* mix, long ncomplex, const float* upstream, float* denh,
                                  void* stream) {
    SEHIP_REQUIRE(ncomplex > 0 && enh && tgt && mix && denh, "psa_loss_bwd: bad arguments (n=%ld)", ncomplex);
    int grid = cdiv(ncomplex, 256 * 4);
    if (grid > 2048) grid = 2048;
    psa_loss_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const float2*)enh, (const float2*)tgt, (const float2*)mix, ncomplex, upstream,
                                                             (float2*)denh);
    SEHIP_CHECK_LAUNCH("psa_loss_bwd");
    return 0;
}

extern "C" int sehip_pointwise_loss_fwd(const float* x, const float* y, long n, int mode, double* acc, float* loss,
                                        void* stream) {
    SEHIP_REQUIRE(n > 0 && (mode == 0 || mode == 1), "pointwise_loss_fwd: bad arguments (n=%ld mode=%d)", n, mode);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(double), st);
    SEHIP_REQUIRE(e == hipSuccess, "pointwise_loss_fwd: memset failed: %s", hipGetErrorString(e));
    int grid = cdiv(n, 256 * 16);
    if (grid > 1024) grid = 1024;
    if (sehip_deterministic()) grid = 1;
    pointwise_loss_fwd_kernel<<<grid, 256, 0, st>>>(x, y, n, mode, acc);
    pointwise_loss_finalize_kernel<<<1, 1, 0, st>>>(acc, n, loss);
    SEHIP_CHECK_LAUNCH("pointwise_loss_fwd");
    return 0;
}

extern "C" int sehip_pointwise_loss_bwd(const float* x, const float* y, long n, int mode, const float* upstream, float* dx,
                                        void* stream) {
    SEHIP_REQUIRE(n > 0 && (mode == 0 || mode == 1), "pointwise_loss_bwd: bad arguments (n=%ld mode=%d)", n, mode);
    int grid = cdiv(n, 256 * 8);
    if (grid > 2048) grid = 2048;
    pointwise_loss_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, n, mode, upstream, dx);
    SEHIP_CHECK_LAUNCH("pointwise_loss_bwd");
    return 0;
}
