// blur_core.h -- the Gaussian blur in front of the Sobel pair of the gradient-NCC similarity (reference
// diffdrr/metrics.py:66, 88-92: torchvision.transforms.functional.gaussian_blur(img, k, sigma) with
// k = int(6 sigma + 1) | 1): the image REFLECT-padded by r = k // 2 (the edge pixel is not repeated), then
// correlated with the outer product of the k normalised taps.  torchvision is a third-party dependency that is
// neither vendored by the reference nor installed in this image: restated from its published algorithm.  The
// taps come from the host (diffdrr_amd/metrics.py computes them as torchvision does, in the image's dtype).
//   blurred[o] = sum_t taps[t] x[reflect(o + t - r)]          per axis (the 2-D kernel is separable)
// and the adjoint folds the padding back:
//   d x[i] = sum_o g[o] w(i, o),   w(i, o) = sum_t taps[t] [reflect(o + t - r) == i]
// Shared by the device kernels (pose_ncc.hip) and the host emulation (tests/emu).
#pragma once

#include "ddrr_common.h"

namespace ddrr {

constexpr int kBlurMaxTaps = 31;  // sigma <= 5 (the reference's default is 1: 7 taps)

// index of the padded sample j in [-r, n - 1 + r] (r < n: one fold), clamped for the tiles' overhang
DDRR_HD int blur_reflect(int j, int n) {
    j = j < 0 ? -j : j;
    j = j >= n ? 2 * (n - 1) - j : j;
    return j < 0 ? 0 : (j >= n ? n - 1 : j);
}

DDRR_HD float blur_tap(const float *taps, int k, int t) { return (t >= 0 && t < k) ? taps[t] : 0.f; }

// w(i, o) of the header: the padded samples that are pixel i are j = i, j = -i (1 <= i) and
// j = 2 (n - 1) - i (i <= n - 2); output o reads j with tap t = j - o + r.  Non-zero only for |o - i| <= r.
DDRR_HD float blur_adjoint_weight(const float *taps, int k, int i, int o, int n) {
    const int r = k >> 1;
    float w = blur_tap(taps, k, i - o + r);
    if (i >= 1) w += blur_tap(taps, k, -i - o + r);
    if (i <= n - 2) w += blur_tap(taps, k, 2 * (n - 1) - i - o + r);
    return w;
}

// away from the borders only the first term is there: w(i, o) = taps[i - o + r]
DDRR_HD bool blur_adjoint_plain(int i, int n, int r) { return i > r && i < n - 1 - r; }

}  // namespace ddrr
