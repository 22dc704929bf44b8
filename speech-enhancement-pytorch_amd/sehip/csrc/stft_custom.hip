// stft_custom / istft_custom of the STFT-domain models (reference: src/evaluate.py:101-128 and :130-162; called twice per
// train step from src/solver.py:457-458 for DCUNet / DNN / ...).  The reference calls torch.stft / torch.istft with a
// periodic hann window of win_length (zero-padded, centred, to n_fft), center=True -> reflect padding by n_fft/2,
// one-sided output [rows][n_fft/2+1][T][2], and divides / multiplies by win_length itself.  n_fft = 512 (both shipped
// configurations: 512/128/512 and 512/256/512) takes the wave64 FFT below; any other n_fft <= 4096 (even or odd) a direct DFT per frame.
//
// One wavefront transforms one frame (fft512.h).  The reference's layout puts TIME innermost, so a frame's 257 bins are
// 257 strided 8-byte elements: a workgroup therefore owns 16 consecutive frames of one row and passes its results
// through an LDS tile [16 frames][257 bins], so that global memory sees 128-byte runs (16 frames of one bin) in both
// directions.  HBM-bound: 4 B/sample in, 8.03 B/sample x n_fft/hop out.
#include "fft512.h"

#define SC_TF 16          // frames per workgroup
#define SC_PITCH 258      // float2 per frame row of the tile (257 + 1: odd pitch in 8-byte words)

// window value of sample n of a frame: periodic hann of win_length centred in n_fft
__device__ __forceinline__ float sc_window(int n, int left, int win_length) {
    const int m = n - left;
    if (m < 0 || m >= win_length) return 0.f;
    return 0.5f - 0.5f * cospif(2.0f * (float)m / (float)win_length);
}

__global__ __launch_bounds__(256) void stft_custom_kernel(const float* __restrict__ wav, int N, int T, int hop, int win_length,
                                                          int center, float2* __restrict__ spec) {
    __shared__ float2 tile[SC_TF * SC_PITCH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.y, t0 = blockIdx.x * SC_TF;
    const float* x = wav + (size_t)row * N;
    FftTw tw;
    fft_twiddles<-1>(tw, lane);
    const int left = (FFT_N - win_length) >> 1;
    const float scale = 1.0f / (float)win_length;          // the reference's  tensor_stft /= win_length
    float wv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) wv[r] = sc_window(lane + 64 * r, left, win_length) * scale;
    const int pad = center ? FFT_N / 2 : 0;
    constexpr int brev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    // two frames per transform (fft_pair_split); a frame beyond T or an odd tail transforms the last valid frame twice
    const int ntl = min(SC_TF, T - t0);
    for (int tp = 2 * w; tp < ntl; tp += 8) {
        const int tla = tp, tlb = tp + 1 < ntl ? tp + 1 : tp;
        float re[8], im[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int sa = (t0 + tla) * hop + lane + 64 * r - pad, sb = (t0 + tlb) * hop + lane + 64 * r - pad;
            if (sa < 0) sa = -sa;                            // reflect (no edge repeat), as torch's pad_mode="reflect"
            if (sa >= N) sa = 2 * (N - 1) - sa;
            if (sb < 0) sb = -sb;
            if (sb >= N) sb = 2 * (N - 1) - sb;
            re[r] = x[sa] * wv[r];
            im[r] = x[sb] * wv[r];
        }
        fft512_wave<-1>(re, im, tw, lane);
        float ar[8], ai[8], br[8], bi[8];
        fft_pair_split(re, im, lane, ar, ai, br, bi);
        const int k0 = 8 * brev6(lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + brev3[j];
            if (k <= 256) {
                tile[tla * SC_PITCH + k] = make_float2(ar[j], ai[j]);
                if (tlb != tla) tile[tlb * SC_PITCH + k] = make_float2(br[j], bi[j]);
            }
        }
    }
    __syncthreads();
    const int nt = min(SC_TF, T - t0);
    float2* out = spec + (size_t)row * NBIN * T + t0;
    for (int idx = threadIdx.x; idx < NBIN * SC_TF; idx += 256) {
        const int f = idx >> 4, tl = idx & (SC_TF - 1);
        if (tl < nt) out[(size_t)f * T + tl] = tile[tl * SC_PITCH + f];
    }
}

// spectrum tile -> windowed synthesis frames [rows][T][512]
__global__ __launch_bounds__(256) void istft_custom_frames_kernel(const float2* __restrict__ spec, int T, int win_length,
                                                                  float* __restrict__ frames) {
    __shared__ float2 tile[SC_TF * SC_PITCH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.y, t0 = blockIdx.x * SC_TF;
    const int nt = min(SC_TF, T - t0);
    const float2* in = spec + (size_t)row * NBIN * T + t0;
    for (int idx = threadIdx.x; idx < NBIN * SC_TF; idx += 256) {
        const int f = idx >> 4, tl = idx & (SC_TF - 1);
        if (tl < nt) tile[tl * SC_PITCH + f] = in[(size_t)f * T + tl];
    }
    __syncthreads();
    FftTw tw;
    fft_twiddles<1>(tw, lane);
    const int left = (FFT_N - win_length) >> 1;
    // tensor * win_length (src/evaluate.py:131), the 1/n_fft of the inverse transform, and the synthesis window
    const float scale = (float)win_length / (float)FFT_N;
    constexpr int brev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    // TWO frames per transform: both Hermitian-extended spectra are packed as H1 + i H2; the inverse transform of a
    // Hermitian spectrum is real, so frame 1 comes out in the real part and frame 2 in the imaginary part.
    for (int tp = 2 * w; tp < nt; tp += 8) {
        const int ta = tp, tb = tp + 1 < nt ? tp + 1 : tp;   // an odd tail transforms its last frame twice
        float re[8], im[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = lane + 64 * r;
            // Hermitian extension of the one-sided spectrum; a c2r transform ignores Im of DC and Nyquist
            const int kk = k <= 256 ? k : FFT_N - k;
            const float2 za = tile[ta * SC_PITCH + kk], zb = tile[tb * SC_PITCH + kk];
            const float sg = (k == 0 || k == 256) ? 0.f : (k < 256 ? 1.f : -1.f);
            re[r] = za.x - sg * zb.y;
            im[r] = sg * za.y + zb.x;
        }
        fft512_wave<1>(re, im, tw, lane);
        const int n0 = 8 * brev6(lane);
        float ua[8], ub[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ua[brev3[j]] = re[j]; ub[brev3[j]] = im[j]; }
        float wq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) wq[q] = scale * sc_window(n0 + q, left, win_length);
        float* fa = frames + ((size_t)row * T + t0 + ta) * FFT_N + n0;
        *reinterpret_cast<float4*>(fa) = make_float4(ua[0] * wq[0], ua[1] * wq[1], ua[2] * wq[2], ua[3] * wq[3]);
        *reinterpret_cast<float4*>(fa + 4) = make_float4(ua[4] * wq[4], ua[5] * wq[5], ua[6] * wq[6], ua[7] * wq[7]);
        if (tb != ta) {
            float* fb = frames + ((size_t)row * T + t0 + tb) * FFT_N + n0;
            *reinterpret_cast<float4*>(fb) = make_float4(ub[0] * wq[0], ub[1] * wq[1], ub[2] * wq[2], ub[3] * wq[3]);
            *reinterpret_cast<float4*>(fb + 4) = make_float4(ub[4] * wq[4], ub[5] * wq[5], ub[6] * wq[6], ub[7] * wq[7]);
        }
    }
}

// overlap-add, divide by the overlap-added squared window, drop the leading centre padding, keep `length` samples
// (zero beyond the overlap-added signal) -- what torch.istft does with an explicit length
__global__ __launch_bounds__(256) void istft_custom_ola_kernel(const float* __restrict__ frames, int T, int NFFT, int hop, int win_length,
                                                               int center, int length, float* __restrict__ wav) {
    const int row = blockIdx.y;
    const int left = (NFFT - win_length) >> 1;
    const int start = center ? NFFT / 2 : 0;
    const int total = NFFT + hop * (T - 1);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < length; i += gridDim.x * 256) {
        const int pos = i + start;
        float v = 0.f;
        if (pos < total) {
            int t_hi = pos / hop;
            if (t_hi > T - 1) t_hi = T - 1;
            int t_lo = pos - NFFT + 1 <= 0 ? 0 : (pos - NFFT + hop) / hop;     // ceil((pos - n_fft + 1) / hop)
            float acc = 0.f, env = 0.f;
            for (int t = t_lo; t <= t_hi; ++t) {
                const int n = pos - t * hop;
                const float wn = sc_window(n, left, win_length);
                acc += frames[((size_t)row * T + t) * NFFT + n];
                env += wn * wn;
            }
            v = acc / env;
        }
        wav[(size_t)row * length + i] = v;
    }
}

// ---- any other n_fft (src/evaluate.py:101-162 hands config.n_fft / hop_length / win_length to torch.stft / torch.istft as they are):
// the transform as a direct DFT per frame -- one workgroup per (row, frame), the windowed frame (the one-sided spectrum) and one period
// of cos / sin in LDS, thread k (n) walks the table with stride k (n) modulo n_fft, so every twiddle is an exact table entry.
// O(n_fft^2) per frame: the evaluation-time path of the configurations nobody ships; the 512-point path above is the fast one.
#define SC_MAX_NFFT 4096
__global__ __launch_bounds__(256) void stft_custom_dft_kernel(const float* __restrict__ wav, int N, int T, int n_fft, int hop, int win_length,
                                                              int center, float2* __restrict__ spec) {
    extern __shared__ float sc_sm[];                    // frame [n_fft] | cos [n_fft] | sin [n_fft]
    float *frame = sc_sm, *ct = sc_sm + n_fft, *st = sc_sm + 2 * n_fft;
    const int row = blockIdx.y, t = blockIdx.x;
    const float* x = wav + (size_t)row * N;
    const int left = (n_fft - win_length) >> 1, pad = center ? n_fft / 2 : 0;
    const float scale = 1.0f / (float)win_length;       // the reference's  tensor_stft /= win_length
    for (int n = threadIdx.x; n < n_fft; n += 256) {
        int sidx = t * hop + n - pad;
        if (sidx < 0) sidx = -sidx;                      // reflect (no edge repeat), as torch's pad_mode="reflect"
        if (sidx >= N) sidx = 2 * (N - 1) - sidx;
        frame[n] = x[sidx] * sc_window(n, left, win_length) * scale;
        const float a = 2.0f * (float)n / (float)n_fft;
        ct[n] = cospif(a); st[n] = sinpif(a);
    }
    __syncthreads();
    const int F = n_fft / 2 + 1;
    for (int k = threadIdx.x; k < F; k += 256) {
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int n = 0; n < n_fft; ++n) {
            const float v = frame[n];
            re += v * ct[idx];
            im -= v * st[idx];
            idx += k;
            if (idx >= n_fft) idx -= n_fft;
        }
        spec[((size_t)row * F + k) * T + t] = make_float2(re, im);
    }
}

// one-sided spectrum of frame t -> windowed synthesis frame [rows][T][n_fft] (the c2r transform torch.istft runs: the imaginary
// parts of DC and, for an even n_fft, of the Nyquist bin are ignored)
__global__ __launch_bounds__(256) void istft_custom_frames_dft_kernel(const float2* __restrict__ spec, int T, int n_fft, int win_length,
                                                                      float* __restrict__ frames) {
    extern __shared__ float sc_sm[];                    // Re [F] | Im [F] | cos [n_fft] | sin [n_fft]
    const int F = n_fft / 2 + 1;
    float *xr = sc_sm, *xi = sc_sm + F, *ct = sc_sm + 2 * F, *st = sc_sm + 2 * F + n_fft;
    const int row = blockIdx.y, t = blockIdx.x;
    for (int k = threadIdx.x; k < F; k += 256) {
        const float2 z = spec[((size_t)row * F + k) * T + t];
        xr[k] = z.x; xi[k] = z.y;
    }
    for (int n = threadIdx.x; n < n_fft; n += 256) {
        const float a = 2.0f * (float)n / (float)n_fft;
        ct[n] = cospif(a); st[n] = sinpif(a);
    }
    __syncthreads();
    const int left = (n_fft - win_length) >> 1;
    const bool even = (n_fft & 1) == 0;
    const int kmax = even ? n_fft / 2 - 1 : n_fft / 2;  // bins with a conjugate partner
    // tensor * win_length (src/evaluate.py:131), the 1/n_fft of the inverse transform, and the synthesis window
    const float scale = (float)win_length / (float)n_fft;
    for (int n = threadIdx.x; n < n_fft; n += 256) {
        float acc = 0.f;
        int idx = n >= n_fft ? 0 : n;                    // k = 1
        for (int k = 1; k <= kmax; ++k) {
            acc += xr[k] * ct[idx] - xi[k] * st[idx];
            idx += n;
            if (idx >= n_fft) idx -= n_fft;
        }
        acc = 2.f * acc + xr[0];
        if (even) acc += (n & 1) ? -xr[n_fft / 2] : xr[n_fft / 2];
        frames[((size_t)row * T + t) * n_fft + n] = acc * scale * sc_window(n, left, win_length);
    }
}

static int sc_check(const char* who, int n_fft, int hop, int win_length) {
    SEHIP_REQUIRE(n_fft >= 2 && n_fft <= SC_MAX_NFFT, "%s: n_fft %d outside [2, %d]", who, n_fft, SC_MAX_NFFT);
    SEHIP_REQUIRE(hop >= 1 && win_length >= 1 && win_length <= n_fft, "%s: bad hop_length %d / win_length %d", who, hop, win_length);
    return 0;
}

extern "C" int sehip_stft_custom_frames(int n_samples, int n_fft, int hop, int center) {
    if (hop < 1) return 0;
    // (centred: the signal is padded by n_fft / 2 on both sides -- for an odd n_fft that is one sample less than n_fft in all)
    const long padded = center ? (long)n_samples + 2L * (n_fft / 2) : (long)n_samples;
    return padded >= n_fft ? (int)(1 + (padded - n_fft) / hop) : 0;
}

extern "C" int sehip_stft_custom_fwd(const float* wav, int rows, int n_samples, int n_fft, int hop, int win_length, int center,
                                     float* spec, void* stream) {
    if (int e = sc_check("stft_custom", n_fft, hop, win_length)) return e;
    SEHIP_REQUIRE(rows > 0 && n_samples > 0, "stft_custom: empty input");
    // torch.stft: reflect padding needs n_fft/2 < n_samples; without centring the signal must hold one frame
    SEHIP_REQUIRE(center ? n_samples > n_fft / 2 : n_samples >= n_fft, "stft_custom: %d samples are too few for n_fft %d",
                  n_samples, n_fft);
    const int T = sehip_stft_custom_frames(n_samples, n_fft, hop, center);
    if (T < 1) return 0;
    if (n_fft != FFT_N) {
        stft_custom_dft_kernel<<<dim3(T, rows), 256, (size_t)3 * n_fft * sizeof(float), (hipStream_t)stream>>>(wav, n_samples, T, n_fft, hop, win_length, center,
                                                                                                             (float2*)spec);
        SEHIP_CHECK_LAUNCH("stft_custom(dft)");
        return 0;
    }
    dim3 grid(cdiv(T, SC_TF), rows);
    stft_custom_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(wav, n_samples, T, hop, win_length, center, (float2*)spec);
    SEHIP_CHECK_LAUNCH("stft_custom");
    return 0;
}

extern "C" int sehip_istft_custom_fwd(const float* spec, int rows, int n_frames, int n_fft, int hop, int win_length, int center,
                                      int length, float* frames_ws, float* wav, void* stream) {
    if (int e = sc_check("istft_custom", n_fft, hop, win_length)) return e;
    SEHIP_REQUIRE(rows > 0 && n_frames > 0 && length > 0, "istft_custom: empty input");
    // NOLA, as torch.istft checks it: the overlap-added squared window must not vanish on the kept interval
    {
        const int left = (n_fft - win_length) / 2, start = center ? n_fft / 2 : 0;
        const long total = n_fft + (long)hop * (n_frames - 1);
        const long end = start + (long)length < total ? start + (long)length : total;
        // the envelope is periodic in hop away from the two ends: checking the first and last n_fft positions covers it
        for (long pos = start; pos < end; ++pos) {
            if (pos >= start + n_fft && pos < end - n_fft) { pos = end - n_fft - 1; continue; }
            double env = 0.0;
            long t_hi = pos / hop; if (t_hi > n_frames - 1) t_hi = n_frames - 1;
            long t_lo = pos - n_fft + 1 <= 0 ? 0 : (pos - n_fft + hop) / hop;
            for (long t = t_lo; t <= t_hi; ++t) {
                const long m = pos - t * hop - left;
                if (m >= 0 && m < win_length) {
                    const double wv = 0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * (double)m / win_length);
                    env += wv * wv;
                }
            }
            SEHIP_REQUIRE(env > 1e-11, "istft_custom: window overlap-add is zero at sample %ld (torch.istft raises here as well)",
                          pos - start);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (n_fft != FFT_N) {
        const size_t lds = ((size_t)2 * (n_fft / 2 + 1) + 2 * n_fft) * sizeof(float);
        istft_custom_frames_dft_kernel<<<dim3(n_frames, rows), 256, lds, st>>>((const float2*)spec, n_frames, n_fft, win_length, frames_ws);
    } else {
        dim3 grid(cdiv(n_frames, SC_TF), rows);
        istft_custom_frames_kernel<<<grid, 256, 0, st>>>((const float2*)spec, n_frames, win_length, frames_ws);
    }
    int gx = cdiv(length, 256 * 4);
    if (gx < 1) gx = 1;
    istft_custom_ola_kernel<<<dim3(gx, rows), 256, 0, st>>>(frames_ws, n_frames, n_fft, hop, win_length, center, length, wav);
    SEHIP_CHECK_LAUNCH("istft_custom");
    return 0;
}
