// The data path in front of the train step, on the device:
//   WavDataset.__getitem__   src/dataset.py:95-170  per-utterance z-score (:147-152; torch.std = UNBIASED) / linear-scale
//                                                   (:154-160) and the aligned random crop (src/utils.py:63-87: zero pad_last when the
//                                                   utterance is shorter than the sample, window [start, start + sample_length))
//   collate_fn_pad           src/distrib.py:38-98   pad to one segment / cut or pad to a whole number of segments, view as
//                                                   segments, concatenate the utterances' segments along the batch axis
// The reference does this per utterance on CPU workers, tensor by tensor.  Here the raw samples of a whole batch arrive as ONE
// flat fp32 buffer (one H2D copy); one launch takes the row statistics, one launch writes the batch tensors in their final
// layout: an output row (segment k of row r of utterance i) is a window of the raw row, normalised on the fly, zeros where
// the reference pads.  HBM-bound streaming: 4 B read + 4 B written per output sample.
#include "common.h"

// rows: raw[row_off[r] .. row_off[r + 1]); stats[r] = {mean, std (unbiased), min, max}
__global__ __launch_bounds__(256) void wav_row_stats_kernel(const float* __restrict__ raw, const long* __restrict__ row_off,
                                                            float4* __restrict__ stats) {
    __shared__ float red[4];
    __shared__ double dred[4];
    const int r = blockIdx.x;
    const long lo = row_off[r], n = row_off[r + 1] - lo;
    const float* x = raw + lo;
    double s = 0.0;
    float mn = 3.4e38f, mx = -3.4e38f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float v = x[i];
        s += (double)v; mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    s = wave_sum_d(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) dred[w] = s;
    __syncthreads();
    const double mean = n > 0 ? (dred[0] + dred[1] + dred[2] + dred[3]) / (double)n : 0.0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = mn;
    __syncthreads();
    mn = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    double q = 0.0;                                             // second pass (the row is in L2): exact centred sum of squares
    for (long i = threadIdx.x; i < n; i += 256) {
        const double dlt = (double)x[i] - mean;
        q += dlt * dlt;
    }
    q = wave_sum_d(q);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) dred[w] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double var = n > 1 ? (dred[0] + dred[1] + dred[2] + dred[3]) / (double)(n - 1) : 0.0;   // torch.std: Bessel's correction
        stats[r] = make_float4((float)mean, (float)sqrt(var), mn, mx);
    }
}

// out[o][0 .. seg) = norm(raw row out_row[o], samples out_start[o] .. + seg), zero beyond the row's end (pad_last) and beyond
// out_valid[o] samples (the crop window's end).  mode 0: none, 1: z-score (x - mean) / (std + eps), 2: linear (x - min) / (max - min + eps)
__global__ __launch_bounds__(256) void wav_collate_kernel(const float* __restrict__ raw, const long* __restrict__ row_off,
                                                          const int* __restrict__ out_row, const long* __restrict__ out_start,
                                                          const int* __restrict__ out_valid, const float4* __restrict__ stats, int mode,
                                                          float eps, int seg, float* __restrict__ out) {
    const int o = blockIdx.y;
    const int r = out_row[o];
    const long lo = row_off[r], n = row_off[r + 1] - lo, st = out_start[o];
    const int valid = out_valid[o];
    float a = 1.f, b = 0.f;                                    // y = (x - b) * a
    if (mode == 1) { const float4 s = stats[r]; b = s.x; a = 1.f / (s.y + eps); }
    else if (mode == 2) { const float4 s = stats[r]; b = s.z; a = 1.f / (s.w - s.z + eps); }
    float* y = out + (size_t)o * seg;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < seg; t += gridDim.x * 256) {
        const long i = st + t;
        y[t] = (t < valid && i < n) ? (raw[lo + i] - b) * a : 0.f;
    }
}

extern "C" int sehip_wav_row_stats(const float* raw, const long* row_off, int rows, float* stats, void* stream) {
    SEHIP_REQUIRE(rows > 0, "wav_row_stats: no rows");
    wav_row_stats_kernel<<<rows, 256, 0, (hipStream_t)stream>>>(raw, row_off, (float4*)stats);
    SEHIP_CHECK_LAUNCH("wav_row_stats");
    return 0;
}

extern "C" int sehip_wav_collate(const float* raw, const long* row_off, const int* out_row, const long* out_start, const int* out_valid,
                                 const float* stats, int mode, float eps, int seg, int out_rows, float* out, void* stream) {
    SEHIP_REQUIRE(out_rows > 0 && seg > 0 && mode >= 0 && mode <= 2, "wav_collate: bad arguments (out_rows=%d seg=%d mode=%d)", out_rows, seg, mode);
    SEHIP_REQUIRE(mode == 0 || stats != nullptr, "wav_collate: normalisation needs the row statistics");
    int gx = cdiv(seg, 256 * 8);
    if (gx < 1) gx = 1;
    wav_collate_kernel<<<dim3(gx, out_rows), 256, 0, (hipStream_t)stream>>>(raw, row_off, out_row, out_start, out_valid, (const float4*)stats,
                                                                              mode, eps, seg, out);
    SEHIP_CHECK_LAUNCH("wav_collate");
    return 0;
}
