// Masking of a complex spectrum by a complex mask, modes 'E' (polar) / 'C' (complex product) / 'R' (real), and its gradient
// with respect to the mask: DCCRN.forward src/model/dccrn.py:203-221 and DCUnet.forward src/model/dcunet.py:136-159 (the
// same formulas).  'E' without trigonometry: cos(phase) = re/|z|, cos(mask_phase) = m_r/|m| with atan2(0,0) = 0 kept.
#pragma once
#include "common.h"

__device__ __forceinline__ void apply_mask(int mode, float re, float im, float mr, float mi, float& er, float& ei) {
    if (mode == 0) {  // E
        const float z2 = re * re + im * im;
        const float mags = sqrtf(z2 + 1e-8f);
        const float zabs = sqrtf(z2);
        const float cp = zabs > 0.f ? re / zabs : 1.f, sp = zabs > 0.f ? im / zabs : 0.f;
        const float rho = sqrtf(mr * mr + mi * mi);
        const float cm = rho > 0.f ? mr / rho : 1.f, sm = rho > 0.f ? mi / rho : 0.f;
        const float a = tanhf(rho) * mags;
        er = a * (cp * cm - sp * sm);
        ei = a * (sp * cm + cp * sm);
    } else if (mode == 1) {  // C
        er = re * mr - im * mi;
        ei = re * mi + im * mr;
    } else {  // R
        er = re * mr;
        ei = im * mi;
    }
}


// d loss / d (er, ei) = (dr, di)  ->  d loss / d (mr, mi).  |m| -> 0 in mode 'E': the reference's autograd yields inf/NaN
// there, which its F.pad / zero rows discard; defined as 0 here.
__device__ __forceinline__ void mask_grad(int mode, float re, float im, float mr, float mi, float dr, float di, float& gmr,
                                          float& gmi) {
    if (mode == 0) {
        const float z2 = re * re + im * im;
        const float mags = sqrtf(z2 + 1e-8f);
        const float zabs = sqrtf(z2);
        const float cp = zabs > 0.f ? re / zabs : 1.f, sp_ = zabs > 0.f ? im / zabs : 0.f;
        const float rho = sqrtf(mr * mr + mi * mi);
        if (rho > 0.f) {
            const float cm = mr / rho, sm = mi / rho;
            const float ce = cp * cm - sp_ * sm, se = sp_ * cm + cp * sm;  // cos/sin(phase + mask phase)
            const float th = tanhf(rho);
            const float g_rho = (dr * ce + di * se) * mags * (1.f - th * th);
            const float g_mu = th * mags * (-dr * se + di * ce);
            gmr = g_rho * cm - g_mu * sm / rho;
            gmi = g_rho * sm + g_mu * cm / rho;
        } else { gmr = 0.f; gmi = 0.f; }
    } else if (mode == 1) {
        gmr = dr * re + di * im;
        gmi = -dr * im + di * re;
    } else {
        gmr = dr * re;
        gmi = di * im;
    }
}
