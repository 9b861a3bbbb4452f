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

// gx, gy at (i, j): out[c, i, j] = sum_{u, v} G_c[u, v] img[i + u - 1, j + v - 1]
DDRR_HD void sobel_pixel(const float *img, int H, int W, int i, int j, float &gx, float &gy) {
    const float a = sobel_at(img, H, W, i - 1, j - 1), b = sobel_at(img, H, W, i - 1, j);
    const float c = sobel_at(img, H, W, i - 1, j + 1), d = sobel_at(img, H, W, i, j - 1);
    const float f = sobel_at(img, H, W, i, j + 1), g = sobel_at(img, H, W, i + 1, j - 1);
    const float h = sobel_at(img, H, W, i + 1, j), k = sobel_at(img, H, W, i + 1, j + 1);
    gx = (a - c) + 2.f * (d - f) + (g - k);
    gy = (a + 2.f * b + c) - (g + 2.f * h + k);
}

// adjoint: d loss / d img[i, j] = sum_c sum_{u, v} G_c[u, v] g[c, i - u + 1, j - v + 1]
DDRR_HD float sobel_pixel_adjoint(const float *gx, const float *gy, int H, int W, int i, int j) {
    // gx contributions: G_x[u, v] at output pixel (i - u + 1, j - v + 1)
    const float x = (sobel_at(gx, H, W, i + 1, j + 1) - sobel_at(gx, H, W, i + 1, j - 1)) +
                    2.f * (sobel_at(gx, H, W, i, j + 1) - sobel_at(gx, H, W, i, j - 1)) +
                    (sobel_at(gx, H, W, i - 1, j + 1) - sobel_at(gx, H, W, i - 1, j - 1));
    const float y = (sobel_at(gy, H, W, i + 1, j + 1) + 2.f * sobel_at(gy, H, W, i + 1, j) +
                     sobel_at(gy, H, W, i + 1, j - 1)) -
                    (sobel_at(gy, H, W, i - 1, j + 1) + 2.f * sobel_at(gy, H, W, i - 1, j) +
                     sobel_at(gy, H, W, i - 1, j - 1));
    return x + y;
}

}  // namespace ddrr
