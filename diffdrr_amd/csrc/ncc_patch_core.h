// ncc_patch_core.h -- NormalizedCrossCorrelation2d(patch_size = p) per window / per pixel.
//
// Reference: diffdrr/metrics.py:16-44.  `to_patches` turns every p x p window (stride 1) of an
// (H, W) image into a channel; `norm` z-scores each window on its own (mean, biased variance + eps)
// and the score is the mean over windows and window pixels of z1 z2:
//     score = (1 / W') sum_w ncc_w,   ncc_w = (1 / n) sum_i (a_i - mu_a)(b_i - mu_b) / (s_a s_b),
//     n = p^2,  W' = (H - p + 1)(W - p + 1),  s = sqrt(var + eps).
// The reference materialises both images as (B, W', p, p) tensors (256^2, p = 13: 40 MB per image
// and pose) and a dozen more of that size in autograd; here a window is two passes over its p^2
// pixels in LDS (the reference's own two-pass arithmetic: mean, then centred moments), and the
// gradient w.r.t. the moving image b follows per pixel j from four per-window coefficients:
//     d score / d b_j = (1 / (W' n)) [ a_j S1 - S2 - b_j S3 + S4 ],   sums over the windows holding j of
//     c1 = 1 / (s_a s_b),  c2 = mu_a c1,  c3 = ncc_w / s_b^2,  c4 = c3 mu_b.
// `fa(y, x)` / `fb(y, x)`: pixel (y, x) of the window (LDS tile on the device, the image on the host).
#pragma once

#include "ddrr_common.h"

namespace ddrr {

// P > 0: the window size as a compile-time constant (the rows unrolled: LDS offsets become immediates
// and the loop arithmetic goes -- the common sizes are instantiated, pose_ncc.hip), else `p`.
template <int P = 0, class FA, class FB>
DDRR_HD float ncc_patch_window(const FA &fa, const FB &fb, int p_rt, float eps, float coef[4]) {
    const int p = P > 0 ? P : p_rt;
    float sa = 0.f, sb = 0.f;
    for (int y = 0; y < p; ++y)
#pragma unroll
        for (int x = 0; x < (P > 0 ? P : p); ++x) {
            sa += fa(y, x);
            sb += fb(y, x);
        }
    const float inv_n = 1.0f / (float)(p * p);
    const float mua = sa * inv_n, mub = sb * inv_n;
    float va = 0.f, vb = 0.f, cab = 0.f;
    for (int y = 0; y < p; ++y)
#pragma unroll
        for (int x = 0; x < (P > 0 ? P : p); ++x) {
            const float da = fa(y, x) - mua, db = fb(y, x) - mub;
            va = fmaf(da, da, va);
            vb = fmaf(db, db, vb);
            cab = fmaf(da, db, cab);
        }
    const float var_b = fmaf(vb, inv_n, eps);
    const float stda = sqrtf(fmaf(va, inv_n, eps)), stdb = sqrtf(var_b);
    const float c1 = 1.0f / (stda * stdb);
    const float ncc = cab * inv_n * c1;
    coef[0] = c1;
    coef[1] = mua * c1;
    coef[2] = ncc / var_b;
    coef[3] = coef[2] * mub;
    return ncc;
}

// `fc(wy, wx, k)`: coefficient k of the window whose top-left pixel is (wy, wx), 0 outside the
// window grid; the windows holding pixel (y, x) are wy in [y - p + 1, y], wx in [x - p + 1, x].
template <int P = 0, class FC>
DDRR_HD float ncc_patch_pixel_grad(const FC &fc, int y, int x, int p_rt, float a, float b) {
    const int p = P > 0 ? P : p_rt;
    float S1 = 0.f, S2 = 0.f, S3 = 0.f, S4 = 0.f;
    for (int wy = y - p + 1; wy <= y; ++wy)
#pragma unroll
        for (int k = 0; k < (P > 0 ? P : p); ++k) {
            const int wx = x - p + 1 + k;
            S1 += fc(wy, wx, 0);
            S2 += fc(wy, wx, 1);
            S3 += fc(wy, wx, 2);
            S4 += fc(wy, wx, 3);
        }
    return fmaf(a, S1, -S2) - fmaf(b, S3, -S4);
}

}  // namespace ddrr
