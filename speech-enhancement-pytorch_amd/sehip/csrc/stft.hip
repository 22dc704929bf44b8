// Conv-STFT / conv-iSTFT front-end of DCCRN as wave64 FFTs
// (reference: src/model/dccrn.py:649-747 init_kernels/ConvSTFT/ConviSTFT, and the glue of
//  DCCRN.forward src/model/dccrn.py:145-154 (mag/phase/DC drop) and :198-229 (mask, istft, clamp)).
//
// The reference multiplies every frame by a dense [514 x 400] basis.  That basis is
//   analysis : rows = (cos, -sin)(2 pi k n / 512) * hann[n]      == 512-point FFT of the windowed frame
//   synthesis: pinv(K)^T * hann,  K = un-windowed analysis; K^T K = 256 I + (11^T + ss^T)/2, s_n = (-1)^n
//              => pinv(K) = (1/256) (I - c (11^T + ss^T)) K^T,  c = 1/(512 + win_len)   (win_len even)
// so both directions are one 512-point complex FFT per frame plus a rank-2 correction.
//
// One wavefront owns one frame: lane l holds points l + 64 r (r = 0..7) in registers, does the radix-8
// part in registers, one twiddle multiply, and the 64-point part across lanes with six __shfl_xor
// butterfly stages.  Outputs land 8 consecutive bins (or samples) per lane, in bit-reversed lane order,
// so every store is a full 32/64-byte segment.  HBM-bound; no LDS.
#include "common.h"

#include "fft512.h"
#include "mask.h"

// ------------------------------------------------------------------------------------------------
// STFT forward: wav [B][N] fp32 -> spec [B][T][257] float2 (re, im) and the encoder input
// [B][T][256][2] bf16 (bins 1..256, channels-last real|imag).       src/model/dccrn.py:687-694,147-154
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_fwd_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                       int B, int N, int T, int win, int hop,
                                                       float2* __restrict__ spec, unsigned* __restrict__ enc_in) {
    const int lane = threadIdx.x & 63;
    const int frame = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= B * T) return;
    const int b = frame / T, t = frame - b * T;
    FftTw tw;
    fft_twiddles<-1>(tw, lane);
    float re[8], im[8];
    const int pad = win - hop;
    const float* x = wav + (size_t)b * N;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        const int s = t * hop + n - pad;
        float v = 0.f;
        if (n < win && s >= 0 && s < N) v = x[s] * window[n];
        re[r] = v; im[r] = 0.f;
    }
    fft512_wave<-1>(re, im, tw, lane);
    // Round 6: the frame leaves through a wave-private LDS row.  The FFT leaves lane l with bins 8 brev6(l) .. + 7: stored from there,
    // every store instruction touched 33 different 64-byte segments with 8 (4) bytes each -- 16 instructions per frame, 34 us for the
    // 32 MB of a headline batch (0.9 TB/s).  From the row, lane i stores bin i + 64 j (512 contiguous bytes per instruction) and the
    // i-th 16-byte piece of the encoder input (1 KB per instruction): 6 instructions.  (Row index k + (k >> 3): the lanes' 64-byte
    // strides would otherwise meet in four banks.)
    __shared__ float2 srow[4][264 + 33];
    __shared__ unsigned senc[4][256];
    const int w = threadIdx.x >> 6;
    const int k0 = 8 * brev6(lane);
    constexpr int brev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    if (k0 <= 256) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + brev3[j];
            if (k <= 256) {
                srow[w][k + (k >> 3)] = make_float2(re[j], im[j]);
                if (k >= 1) senc[w][k - 1] = pack_bf2(re[j], im[j]);
            }
        }
    }
    // (the row belongs to this wave alone and a wave's LDS operations complete in order: a compiler-level barrier is enough)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float2* so = spec + (size_t)frame * NBIN;
    unsigned* eo = enc_in + (size_t)frame * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        so[k] = srow[w][k + (k >> 3)];
    }
    if (lane == 0) so[256] = srow[w][256 + 32];
    *reinterpret_cast<uint4*>(eo + 4 * lane) = *reinterpret_cast<const uint4*>(&senc[w][4 * lane]);
}

// ------------------------------------------------------------------------------------------------
// mask ('E' polar / 'C' complex / 'R' real) + inverse transform to windowed frames.
//   spec [B][T][257] float2, mask [B][T][256][2] fp32 (bins 1..256) -> frames [B][T][win] fp32
// 'E' without trigonometry: cos(phase)=re/|z|, cos(mask_phase)=m_r/|m| (atan2(0,0)=0 conventions kept).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void istft_frames_kernel(const float2* __restrict__ spec, const float2* __restrict__ mask,
                                                           const float* __restrict__ window, int nframes, int win, int mode,
                                                           float* __restrict__ frames) {
    const int lane = threadIdx.x & 63;
    const int frame = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= nframes) return;
    FftTw tw;
    fft_twiddles<1>(tw, lane);
    float re[8], im[8];
    const float2* sp = spec + (size_t)frame * NBIN;
    const float2* mk = mask + (size_t)frame * 256;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        float er = 0.f, ei = 0.f;
        if (k >= 1 && k <= 256) {  // DC: the mask row is zero padding -> estimate is exactly 0
            const float2 z = sp[k];
            const float2 m = mk[k - 1];
            apply_mask(mode, z.x, z.y, m.x, m.y, er, ei);
        }
        re[r] = er; im[r] = ei;
    }
    fft512_wave<1>(re, im, tw, lane);
    // u[n] = Re(ifft), n = n0 + brev3(j)
    const int n0 = 8 * brev6(lane);
    constexpr int brev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    float u[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + brev3[j];
        u[brev3[j]] = re[j];
        if (n < win) { s1 += re[j]; s2 += (n & 1) ? -re[j] : re[j]; }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float c = 1.0f / (float)(FFT_N + win);
    float* fo = frames + (size_t)frame * win;
    if (n0 + 8 <= win && (win & 3) == 0) {             // the lane's eight samples as two 16-byte stores (they were eight 4-byte ones)
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = n0 + q;
            const float corr = c * (s1 + ((n & 1) ? -s2 : s2));
            o[q] = window[n] * (1.0f / 256.0f) * (u[q] - corr);
        }
        *reinterpret_cast<float4*>(fo + n0) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(fo + n0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n = n0 + q;
        if (n < win) {
            const float corr = c * (s1 + ((n & 1) ? -s2 : s2));
            fo[n] = window[n] * (1.0f / 256.0f) * (u[q] - corr);
        }
    }
}

// overlap-add / window energy, trim, [:length], clamp   (src/model/dccrn.py:733-745, :228)
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ inv_coff,
                                                        int T, int win, int hop, int length, float* __restrict__ wav) {
    const int b = blockIdx.y;
    const int pad = win - hop;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < length; i += gridDim.x * 256) {
        const int s = i + pad;
        int t_hi = s / hop;
        int t_lo = (s - win + hop) / hop;  // ceil((s - win + 1)/hop) for s >= win-1
        if (s - win + 1 <= 0) t_lo = 0;
        if (t_hi > T - 1) t_hi = T - 1;
        float acc = 0.f;
        for (int t = t_lo; t <= t_hi; ++t) acc += frames[((size_t)b * T + t) * win + (s - t * hop)];
        acc *= inv_coff[i];
        wav[(size_t)b * length + i] = fminf(1.f, fmaxf(-1.f, acc));
    }
}

// ------------------------------------------------------------------------------------------------
// backward of clamp + OLA + synthesis + mask:  d wav -> d mask  ([B][T][256][2], bf16 for the dgrad
// GEMM of the last decoder layer and fp32 copy optional).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void istft_bwd_kernel(const float* __restrict__ dwav, const float* __restrict__ wav,
                                                        const float* __restrict__ inv_coff, const float* __restrict__ window,
                                                        const float2* __restrict__ spec, const float2* __restrict__ mask,
                                                        int B, int T, int win, int hop, int length, int mode,
                                                        unsigned* __restrict__ dmask_bf16) {
    const int lane = threadIdx.x & 63;
    const int frame = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= B * T) return;
    const int b = frame / T, t = frame - b * T;
    FftTw tw;
    fft_twiddles<-1>(tw, lane);
    const int pad = win - hop;
    float re[8], im[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        const int i = t * hop + n - pad;
        float v = 0.f;
        if (n < win && i >= 0 && i < length) {
            const size_t o = (size_t)b * length + i;
            const float y = wav[o];
            if (y > -1.f && y < 1.f) v = dwav[o] * inv_coff[i] * window[n] * (1.0f / 256.0f);
        }
        re[r] = v; im[r] = 0.f;
        s1 += v; s2 += (n & 1) ? -v : v;   // n parity == lane parity (64 r is even)
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float c = 1.0f / (float)(FFT_N + win);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        if (n < win) re[r] -= c * (s1 + ((n & 1) ? -s2 : s2));
    }
    // Round 6 (as stft_fwd_kernel): the spectrum and mask rows of the frame are fetched LINEARLY (lane i takes bins i + 64 j: 512
    // contiguous bytes per instruction, requested before the transform and parked in a wave-private LDS row behind it) and the mask
    // gradient leaves as one 16-byte piece per lane; from the transform's own layout (lane l holds bins 8 brev6(l) .. + 7) these were
    // 24 instructions per frame of 33 scattered 8- / 4-byte accesses each (41 us per headline batch).
    const float2* sp = spec + (size_t)frame * NBIN;
    const float2* mk = mask + (size_t)frame * 256;
    float2 spv[4], mkv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { spv[j] = sp[1 + lane + 64 * j]; mkv[j] = mk[lane + 64 * j]; }      // bins 1 .. 256 (DC carries no mask)
    fft512_wave<-1>(re, im, tw, lane);
    __shared__ float2 ssp[4][256 + 32], smk[4][256 + 32];
    __shared__ unsigned sdm[4][256];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;                   // index k - 1
        ssp[w][i + (i >> 3)] = spv[j];
        smk[w][i + (i >> 3)] = mkv[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int k0 = 8 * brev6(lane);
    constexpr int brev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    if (k0 <= 256) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + brev3[j];
            if (k < 1 || k > 256) continue;
            const float dr = re[j], di = im[j];  // d loss / d est_real[k], d est_imag[k]
            const int i = k - 1;
            const float2 z = ssp[w][i + (i >> 3)];
            const float2 m = smk[w][i + (i >> 3)];
            float gmr, gmi;
            mask_grad(mode, z.x, z.y, m.x, m.y, dr, di, gmr, gmi);
            sdm[w][i] = pack_bf2(gmr, gmi);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned* dm = dmask_bf16 + (size_t)frame * 256;
    *reinterpret_cast<uint4*>(dm + 4 * lane) = *reinterpret_cast<const uint4*>(&sdm[w][4 * lane]);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int check_stft_cfg(const char* who, int win, int hop, int fft) {
    SEHIP_REQUIRE(fft == FFT_N, "%s: only fft_len 512 is built (got %d)", who, fft);
    SEHIP_REQUIRE(win > 0 && win <= FFT_N && (win % 2) == 0, "%s: win_len must be even and <= 512 (got %d)", who, win);
    SEHIP_REQUIRE(hop > 0 && hop <= win, "%s: bad hop %d", who, hop);
    return 0;
}

extern "C" int sehip_stft_frames(int n_samples, int win, int hop) { return (n_samples + 2 * (win - hop) - win) / hop + 1; }

extern "C" int sehip_stft_fwd(const float* wav, const float* window, int B, int N, int win, int hop, int fft, float* spec,
                              void* enc_in_bf16, void* stream) {
    if (int e = check_stft_cfg("stft_fwd", win, hop, fft)) return e;
    SEHIP_REQUIRE(B > 0 && N > 0, "stft_fwd: empty input");
    const int T = sehip_stft_frames(N, win, hop);
    SEHIP_REQUIRE(T > 0, "stft_fwd: input shorter than one frame");
    stft_fwd_kernel<<<cdiv((long)B * T, 4), 256, 0, (hipStream_t)stream>>>(wav, window, B, N, T, win, hop, (float2*)spec,
                                                                           (unsigned*)enc_in_bf16);
    SEHIP_CHECK_LAUNCH("stft_fwd");
    return 0;
}

extern "C" int sehip_istft_fwd(const float* spec, const float* mask, const float* window, const float* inv_coff, int B,
                               int T, int win, int hop, int fft, int length, int mode, float* frames_ws, float* wav,
                               void* stream) {
    if (int e = check_stft_cfg("istft_fwd", win, hop, fft)) return e;
    SEHIP_REQUIRE(B > 0 && T > 0 && length > 0, "istft_fwd: empty input");
    SEHIP_REQUIRE(mode >= 0 && mode <= 2, "istft_fwd: masking mode must be 0(E) 1(C) 2(R)");
    SEHIP_REQUIRE(length <= (T - 1) * hop + win - (win - hop), "istft_fwd: length %d exceeds the synthesised signal", length);
    hipStream_t st = (hipStream_t)stream;
    istft_frames_kernel<<<cdiv((long)B * T, 4), 256, 0, st>>>((const float2*)spec, (const float2*)mask, window, B * T, win,
                                                             mode, frames_ws);
    dim3 grid(cdiv(length, 256 * 4), B);
    istft_ola_kernel<<<grid, 256, 0, st>>>(frames_ws, inv_coff, T, win, hop, length, wav);
    SEHIP_CHECK_LAUNCH("istft_fwd");
    return 0;
}

extern "C" int sehip_istft_bwd(const float* dwav, const float* wav, const float* spec, const float* mask,
                               const float* window, const float* inv_coff, int B, int T, int win, int hop, int fft,
                               int length, int mode, void* dmask_bf16, void* stream) {
    if (int e = check_stft_cfg("istft_bwd", win, hop, fft)) return e;
    SEHIP_REQUIRE(B > 0 && T > 0 && length > 0, "istft_bwd: empty input");
    SEHIP_REQUIRE(mode >= 0 && mode <= 2, "istft_bwd: masking mode must be 0(E) 1(C) 2(R)");
    istft_bwd_kernel<<<cdiv((long)B * T, 4), 256, 0, (hipStream_t)stream>>>(dwav, wav, inv_coff, window, (const float2*)spec,
                                                                            (const float2*)mask, B, T, win, hop, length, mode,
                                                                            (unsigned*)dmask_bf16);
    SEHIP_CHECK_LAUNCH("istft_bwd");
    return 0;
}
