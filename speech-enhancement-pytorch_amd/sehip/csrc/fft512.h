// 512-point complex FFT on one wavefront (shared by stft.hip and stft_custom.hip).
// Lane l holds points l + 64 r (r = 0..7) in registers: radix-8 in registers, one twiddle multiply, then the 64-point
// part across lanes with six __shfl_xor butterfly stages.  Outputs land 8 consecutive bins (or samples) per lane, in
// bit-reversed lane order.
#pragma once
#include "common.h"

#define FFT_N 512
#define NBIN 257

struct FftTw {
    float tr[8], ti[8];  // W512^(lane * k2)
    float sr[5], si[5];  // stage twiddles W_(2h)^(lane mod h), h = 32,16,8,4,2
};

// The 13 twiddles a lane needs, as a table evaluated at COMPILE time (double-precision series, rounded once): 13 sincospif
// calls per wave were ~650 instructions, as many as the transform of the one frame a wave of the DCCRN kernels owns.
//   rows 0..7 : W512^(lane * k) = cos / sin (2 pi lane k / 512)        rows 8..12: W_(2h)^(lane mod h), h = 32,16,8,4,2
namespace fft_detail {
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double cos_series(double x) {  // |x| <= pi/4
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int i = 1; i <= 12; ++i) { term *= -x2 / ((2 * i - 1) * (2 * i)); sum += term; }
    return sum;
}
constexpr double sin_series(double x) {  // |x| <= pi/4
    double x2 = x * x, term = x, sum = x;
    for (int i = 1; i <= 12; ++i) { term *= -x2 / ((2 * i) * (2 * i + 1)); sum += term; }
    return sum;
}
// cos / sin of 2 pi num / den with exact octant reduction on the integers
constexpr void cossin_frac(long num, long den, double& c, double& s) {
    num %= den; if (num < 0) num += den;
    const long oct = (8 * num) / den;                 // octant 0..7
    const double x = 2.0 * kPi * (double)(8 * num - oct * den) / (double)(8 * den);  // angle within the octant, [0, pi/4)
    const double cx = cos_series(x), sx = sin_series(x);
    const double h = 0.70710678118654752440084436210484903928;
    // rotate by oct * pi/4
    switch (oct) {
        case 0: c = cx; s = sx; break;
        case 1: c = h * (cx - sx); s = h * (cx + sx); break;
        case 2: c = -sx; s = cx; break;
        case 3: c = -h * (cx + sx); s = h * (cx - sx); break;
        case 4: c = -cx; s = -sx; break;
        case 5: c = -h * (cx - sx); s = -h * (cx + sx); break;
        case 6: c = sx; s = -cx; break;
        default: c = h * (cx + sx); s = -h * (cx - sx); break;
    }
}
struct TwTable { float c[13][64], s[13][64]; };
constexpr TwTable make_table() {
    TwTable t{};
    for (int lane = 0; lane < 64; ++lane) {
        for (int k = 0; k < 8; ++k) {
            double c = 0, s = 0;
            cossin_frac((long)lane * k, 512, c, s);
            t.c[k][lane] = (float)c; t.s[k][lane] = (float)s;
        }
        for (int st = 0; st < 5; ++st) {
            const int h = 32 >> st;
            double c = 0, s = 0;
            cossin_frac(lane & (h - 1), 2 * h, c, s);
            t.c[8 + st][lane] = (float)c; t.s[8 + st][lane] = (float)s;
        }
    }
    return t;
}
}  // namespace fft_detail
__device__ const fft_detail::TwTable g_fft_tw = fft_detail::make_table();

template <int SIGN>
__device__ __forceinline__ void fft_twiddles(FftTw& w, int lane) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { w.tr[k] = g_fft_tw.c[k][lane]; w.ti[k] = SIGN * g_fft_tw.s[k][lane]; }
#pragma unroll
    for (int st = 0; st < 5; ++st) { w.sr[st] = g_fft_tw.c[8 + st][lane]; w.si[st] = SIGN * g_fft_tw.s[8 + st][lane]; }
}

// in : lane l, register r  <->  element l + 64 r
// out: lane l, register j  <->  element 8*brev6(l) + brev3(j)
template <int SIGN>
__device__ __forceinline__ void fft512_wave(float (&re)[8], float (&im)[8], const FftTw& w, int lane) {
    const float h = 0.70710678118654752f;
    // ---- radix-8 DIF over the register index ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ar = re[i], ai = im[i], br = re[i + 4], bi = im[i + 4];
        re[i] = ar + br; im[i] = ai + bi;
        float dr = ar - br, di = ai - bi;
        // * W8^i : (1), (h, s*h), (0, s), (-h, s*h)  with s = SIGN
        if (i == 0) { re[4] = dr; im[4] = di; }
        if (i == 1) { re[5] = h * (dr - SIGN * di); im[5] = h * (di + SIGN * dr); }
        if (i == 2) { re[6] = -SIGN * di; im[6] = SIGN * dr; }
        if (i == 3) { re[7] = h * (-dr - SIGN * di); im[7] = h * (-di + SIGN * dr); }
    }
#pragma unroll
    for (int b = 0; b < 8; b += 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float ar = re[b + i], ai = im[b + i], br = re[b + i + 2], bi = im[b + i + 2];
            re[b + i] = ar + br; im[b + i] = ai + bi;
            float dr = ar - br, di = ai - bi;
            if (i == 0) { re[b + 2] = dr; im[b + 2] = di; }
            else        { re[b + 3] = -SIGN * di; im[b + 3] = SIGN * dr; }
        }
    }
#pragma unroll
    for (int b = 0; b < 8; b += 2) {
        float ar = re[b], ai = im[b], br = re[b + 1], bi = im[b + 1];
        re[b] = ar + br; im[b] = ai + bi;
        re[b + 1] = ar - br; im[b + 1] = ai - bi;
    }
    // register j now holds k2 = brev3(j); twiddle by W512^(lane*k2)
    constexpr int brev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
#pragma unroll
    for (int j = 1; j < 8; ++j) {
        const float tr = w.tr[brev3[j]], ti = w.ti[brev3[j]];
        const float xr = re[j], xi = im[j];
        re[j] = xr * tr - xi * ti;
        im[j] = xr * ti + xi * tr;
    }
    // ---- 64-point DIF across lanes ----
    // Branch-free butterflies: lower lane  x + p,  upper lane  (p - x) * w.  With sg = -1 / +1 and the stage twiddle of the
    // lower lanes set to 1 both are  (sg * x + p) * w'  (an if / else per element cost two divergent branches each:
    // 96 per frame, 116 s_cbranch in the kernel).
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        const int hh = 32 >> st;
        const bool upper = (lane & hh) != 0;
        const float sg = upper ? -1.f : 1.f;
        const float sr = (upper && st < 5) ? w.sr[st] : 1.f, si = (upper && st < 5) ? w.si[st] : 0.f;
        float pr[8], pi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { pr[j] = __shfl_xor(re[j], hh, 64); pi[j] = __shfl_xor(im[j], hh, 64); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ar = sg * re[j] + pr[j], ai = sg * im[j] + pi[j];
            re[j] = ar * sr - ai * si;
            im[j] = ar * si + ai * sr;
        }
    }
}

__device__ __forceinline__ int brev6(int l) { return (int)(__brev((unsigned)l) >> 26); }

// Two REAL 512-point transforms for the price of one.  Forward: transform z = x1 + i x2, then
//   X1[k] = (Z[k] + conj(Z[512-k])) / 2,   X2[k] = (Z[k] - conj(Z[512-k])) / (2i).
// In the output layout of fft512_wave (lane l, register j <-> k = 8 brev6(l) + brev3(j)) bin 512-k lives in lane l ^ 63,
// register jp[j] for brev3(j) != 0, and in lane brev6((64 - brev6(l)) & 63), register 0 for j = 0: 16 shuffles instead
// of a second transform's 96.  Inverse: the inverse transform of a Hermitian spectrum is real, so H1 + i H2 comes back as
// frame 1 in the real part and frame 2 in the imaginary part with no exchange at all.
__device__ __forceinline__ void fft_pair_split(const float (&zr)[8], const float (&zi)[8], int lane, float (&ar)[8],
                                               float (&ai)[8], float (&br)[8], float (&bi)[8]) {
    constexpr int jp[8] = {0, 1, 3, 2, 7, 6, 5, 4};
    const int mir = lane ^ 63;
    const int l0 = brev6((64 - brev6(lane)) & 63);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int src = j == 0 ? l0 : mir;
        const float pr = __shfl(zr[jp[j]], src, 64), pi = __shfl(zi[jp[j]], src, 64);
        ar[j] = 0.5f * (zr[j] + pr); ai[j] = 0.5f * (zi[j] - pi);
        br[j] = 0.5f * (zi[j] + pi); bi[j] = -0.5f * (zr[j] - pr);
    }
}

