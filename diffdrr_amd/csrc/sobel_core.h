// sobel_core.h -- the 3x3 Sobel pair of the gradient-NCC similarity (reference
// diffdrr/metrics.py:69-94: torch.nn.Conv2d(1, 2, 3, padding=1, bias=False) with
//   Gx = [[1, 0, -1], [2, 0, -2], [1, 0, -1]],  Gy = [[1, 2, 1], [0, 0, 0], [-1, -2, -1]],
// cross-correlation as conv2d computes it, zero padding) and its adjoint.
#pragma once

#include "ddrr_common.h"

namespace ddrr {

// pixel (i, j) of an H x W image, 0 outside (the conv's zero padding)
DDRR_HD float sobel_at(const float *img, int H, int W, int i, int j) {
    return (i >= 0 && i < H && j >= 0 && j < W) ? img[i * W + j] : 0.f;
}

// gx, gy from the 3 x 3 neighbourhood f(di, dj), di, dj in {-1, 0, 1}:
// out[c] = sum_{u, v} G_c[u, v] f(u - 1, v - 1)
template <class F>
DDRR_HD void sobel_of(const F &f, float &gx, float &gy) {
    const float a = f(-1, -1), b = f(-1, 0), c = f(-1, 1), d = f(0, -1);
    const float e = f(0, 1), g = f(1, -1), h = f(1, 0), k = f(1, 1);
    gx = (a - c) + 2.f * (d - e) + (g - k);
    gy = (a + 2.f * b + c) - (g + 2.f * h + k);
}

// adjoint: d loss / d img[i, j] = sum_c sum_{u, v} G_c[u, v] g[c, i - u + 1, j - v + 1], from the two
// channels' neighbourhoods fx(di, dj), fy(di, dj)
template <class FX, class FY>
DDRR_HD float sobel_adjoint_of(const FX &fx, const FY &fy) {
    const float x = (fx(1, 1) - fx(1, -1)) + 2.f * (fx(0, 1) - fx(0, -1)) + (fx(-1, 1) - fx(-1, -1));
    const float y = (fy(1, 1) + 2.f * fy(1, 0) + fy(1, -1)) - (fy(-1, 1) + 2.f * fy(-1, 0) + fy(-1, -1));
    return x + y;
}

// the same on an H x W image in memory, zero outside
DDRR_HD void sobel_pixel(const float *img, int H, int W, int i, int j, float &gx, float &gy) {
    sobel_of([&](int di, int dj) { return sobel_at(img, H, W, i + di, j + dj); }, gx, gy);
}

DDRR_HD float sobel_pixel_adjoint(const float *gx, const float *gy, int H, int W, int i, int j) {
    return sobel_adjoint_of([&](int di, int dj) { return sobel_at(gx, H, W, i + di, j + dj); },
                            [&](int di, int dj) { return sobel_at(gy, H, W, i + di, j + dj); });
}

}  // namespace ddrr
