// brick_walk.h -- the lean per-ray code of the volume-stationary Siddon kernel.
//
// The brick kernel is bound by vector-ALU issue (a gfx950 SIMD retires one wave64 VALU
// instruction per 4 cycles), so everything a lane executes per candidate pixel, per
// (ray, brick) hit and per voxel step is written out here with the instruction count in
// mind instead of reusing the general walk of siddon_core.h:
//   phase A  brick_candidate():  arithmetic-only conservative slab test of a detector
//            pixel against the brick, from the pose's affine detector model (no loads);
//   phase B  the exact clip of the real ray, its entry cell and the 3-way merge of the
//            plane crossings reading voxels from the LDS brick: brick_step.h.
// Reference semantics: diffdrr/renderers.py:34-76, 94-113 (see siddon_core.h).
#pragma once

#include "brick_core.h"
#include "ddrr_common.h"
#include "siddon_core.h"

namespace ddrr {

// Affine model of a pose's detector grid: target(i, j) ~ t00 + i ei + j ej
// (detector.py:126, 147-153).  The steps are taken from the far corners so that the
// model is within ~3e-4 voxel of the stored fp32 targets everywhere on the detector.
struct PoseGrid {
    float s[3], t00[3], ei[3], ej[3];
};

DDRR_HD PoseGrid pose_grid(const float *src, const float *tgt, int det_h, int det_w) {
    PoseGrid g;
    const float *tH = tgt + (long)3 * det_w * (det_h - 1), *tW = tgt + (long)3 * (det_w - 1);
    const float rh = 1.0f / (float)(det_h - 1), rw = 1.0f / (float)(det_w - 1);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g.s[a] = src[a];
        g.t00[a] = tgt[a];
        g.ei[a] = (tH[a] - tgt[a]) * rh;
        g.ej[a] = (tW[a] - tgt[a]) * rw;
    }
    return g;
}

// A box between planes given in (fractional) plane-index units: plane k sits at x = k - shift.
// Siddon bricks: integer planes lo .. hi; trilinear bricks: the base-cell box in index
// coordinates, lo + 1/2 .. lo + 31 + 1/2.
struct BoxF {
    float lo[3], hi[3];
};

DDRR_HD BoxF boxf(const Box &b) {
    BoxF f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        f.lo[a] = (float)b.lo[a];
        f.hi[a] = (float)b.hi[a];
    }
    return f;
}

// Pixel bounding box of the lines through the source that meet the box `b` (plane indices;
// plane k sits at x = k - shift): the 8 corners are projected onto the pixel lattice.
DDRR_HD PixBox project_brick_grid(const PoseGrid &g, int det_h, int det_w, const BoxF &b,
                                  float shift) {
    float r[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) r[a] = g.t00[a] - g.s[a];
    const float *ei = g.ei, *ej = g.ej;
    // solve lambda * w - i * ei - j * ej = r for every corner w = p - src (Cramer)
    const float n[3] = {ei[1] * ej[2] - ei[2] * ej[1], ei[2] * ej[0] - ei[0] * ej[2],
                        ei[0] * ej[1] - ei[1] * ej[0]};  // ei x ej
    const float rxej[3] = {r[1] * ej[2] - r[2] * ej[1], r[2] * ej[0] - r[0] * ej[2],
                           r[0] * ej[1] - r[1] * ej[0]};
    const float rxei[3] = {r[1] * ei[2] - r[2] * ei[1], r[2] * ei[0] - r[0] * ei[2],
                           r[0] * ei[1] - r[1] * ei[0]};
    float imin = INFINITY, imax = -INFINITY, jmin = INFINITY, jmax = -INFINITY;
    int npos = 0, nneg = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float w[3] = {((c & 1) ? b.hi[0] : b.lo[0]) - shift - g.s[0],
                            ((c & 2) ? b.hi[1] : b.lo[1]) - shift - g.s[1],
                            ((c & 4) ? b.hi[2] : b.lo[2]) - shift - g.s[2]};
        const float det = w[0] * n[0] + w[1] * n[1] + w[2] * n[2];
        npos += det > 0.f;
        nneg += det < 0.f;
        const float inv = 1.0f / det;
        const float i = -(w[0] * rxej[0] + w[1] * rxej[1] + w[2] * rxej[2]) * inv;
        const float j = (w[0] * rxei[0] + w[1] * rxei[1] + w[2] * rxei[2]) * inv;
        imin = fminf(imin, i);
        imax = fmaxf(imax, i);
        jmin = fminf(jmin, j);
        jmax = fmaxf(jmax, j);
    }
    PixBox pb;
    if (npos != 8 && nneg != 8) {
        // the plane through the source parallel to the detector cuts the box: its
        // projection is unbounded -> every pixel is a candidate
        pb.i0 = 0;
        pb.i1 = det_h - 1;
        pb.j0 = 0;
        pb.j1 = det_w - 1;
        return pb;
    }
    // pixel centres are the integer (i, j); keep a small guard band for rounding
    const float gb = 0.02f;
    const float fi0 = fmaxf(floorf(imin - gb), 0.f), fi1 = fminf(ceilf(imax + gb), (float)(det_h - 1));
    const float fj0 = fmaxf(floorf(jmin - gb), 0.f), fj1 = fminf(ceilf(jmax + gb), (float)(det_w - 1));
    if (!(fi0 <= fi1) || !(fj0 <= fj1)) {  // also catches NaN
        pb.i0 = pb.j0 = 0;
        pb.i1 = pb.j1 = -1;
        return pb;
    }
    pb.i0 = (int)fi0;
    pb.i1 = (int)fi1;
    pb.j0 = (int)fj0;
    pb.j1 = (int)fj1;
    return pb;
}

// One row of the per-(pose, brick) table phase A reads (22 words).
constexpr int kRowWords = 22;
struct BrickRow {
    float D0[3];   // (t00 - s) + eps : direction of pixel (0, 0)
    float ei[3], ej[3];
    float P0[3];   // (lo - margin) - shift - s : numerators of the slab planes
    float P1[3];   // (hi + margin) - shift - s
    float inv_w;   // 1 / width of the pixel box
    float nscale;  // > 0: work estimate = (exit - entry) * nscale (samples); else crossings
    int i0, j0, w, count;
    int perm_k;    // candidate order: local -> (local * perm_k) mod count (1 = row-major)
};

// Scatter kernels want the 64 rays of a wave FAR APART: neighbouring rays cross the same
// voxels, and 64 LDS atomics on a handful of addresses serialise.  A multiplicative shuffle
// of the candidate order (a bijection of 0..count-1 when gcd(k, count) = 1) puts consecutive
// candidates ~3 detector rows and ~3 columns apart.
DDRR_HD int scatter_perm_k(int w, int count) {
    if (count < 64) return 1;
    int k = 3 * w + 3;
    if (k >= count) k = count / 2 + 1;
    for (int tries = 0; tries < 64; ++tries, ++k) {
        int a = k, b = count;
        while (b) {
            const int r = a % b;
            a = b;
            b = r;
        }
        if (a == 1) return k;
    }
    return 1;
}

DDRR_HD int scatter_perm(int local, int k, int count, float inv_count) {
    if (k == 1) return local;
    // (local * k) mod count; the product stays below 2^24 for every pixel box that fits a
    // detector of up to 4096^2 / bricks... guard: fall back to exact integer arithmetic
    const int prod = local * k;
    if (prod < (1 << 23)) {
        int q = (int)((float)prod * inv_count);
        int r = prod - q * count;
        r += r < 0 ? count : 0;
        r -= r >= count ? count : 0;
        return r;
    }
    return (int)(((long)local * k) % count);
}

// With a float backward record a brick kernel is bound by its atomics, which cost per 64-byte
// request at the memory side (0.4 ms per record plane at 512^3 / 32 poses): candidate rows that
// start and end on 8-pixel boundaries make every lane group of 8 one 32-byte-aligned run of each
// plane -- 1.0 request per group and plane instead of 1.44 -- at the price of ~30 % more
// candidates to test (measured: forward + record 2.00 -> 1.83 ms).
DDRR_HD PixBox align_pixbox_rows(PixBox pb, int det_w) {
    if (pb.j1 < pb.j0 || pb.i1 < pb.i0) return pb;
    pb.j0 &= ~7;
    pb.j1 = (pb.j1 | 7) < det_w ? (pb.j1 | 7) : det_w - 1;
    return pb;
}

// margin (voxels) by which phase A inflates the brick: covers the affine model's distance
// from the stored targets, so a ray with a real chord in the brick is never rejected
constexpr float kBrickMargin = 0.01f;

DDRR_HD BrickRow brick_row(const PoseGrid &g, const PixBox &pb, const BoxF &box, float shift,
                           float eps, float nscale) {
    BrickRow r;
    r.nscale = nscale;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        r.D0[a] = (g.t00[a] - g.s[a]) + eps;
        r.ei[a] = g.ei[a];
        r.ej[a] = g.ej[a];
        r.P0[a] = (box.lo[a] - kBrickMargin) - shift - g.s[a];
        r.P1[a] = (box.hi[a] + kBrickMargin) - shift - g.s[a];
    }
    r.i0 = pb.i0;
    r.j0 = pb.j0;
    r.w = pb.j1 - pb.j0 + 1;
    r.count = pixbox_count(pb);
    r.inv_w = 1.0f / (float)(r.w > 0 ? r.w : 1);
    r.perm_k = 1;
    return r;
}

DDRR_HD float approx_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// Phase A for candidate `local` (row-major index into the pixel box, < count).
// Returns whether the pixel's ray may cross the (inflated) brick; pix = i * det_w + j;
// n_est ~ number of plane crossings inside the brick (for length classes only).
DDRR_HD bool brick_candidate(const BrickRow &r, int local, int det_w, int &pix, float &n_est) {
    // (local + 0.5) / w is at least 0.5 / w away from an integer: the float product
    // truncates to the exact quotient for every box that fits a detector
    const int di = (int)(((float)local + 0.5f) * r.inv_w);
    const int i = r.i0 + di, j = r.j0 + (local - di * r.w);
    pix = i * det_w + j;
    const float fi = (float)i, fj = (float)j;
    float entry = -INFINITY, exit = INFINITY, l1 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float d = fmaf(fj, r.ej[a], fmaf(fi, r.ei[a], r.D0[a]));
        const float inv = approx_rcp(d);
        const float a0 = r.P0[a] * inv, a1 = r.P1[a] * inv;
        entry = fmaxf(entry, fminf(a0, a1));
        exit = fminf(exit, fmaxf(a0, a1));
        l1 += fabsf(d);
    }
    n_est = (exit - entry) * (r.nscale > 0.f ? r.nscale : l1);
    return entry < exit;  // false for NaN
}

// ------------------------------------------------------------------ phase B

struct BrickGeom {
    float lof[3], hif[3];  // the brick's first / last plane index per axis, as floats
    float stridef[3];      // BYTE strides of the LDS copy (as floats: offsets stay < 2^24)
};

DDRR_HD BrickGeom brick_geom(const Box &box, const BrickLayout &lay) {
    BrickGeom G;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        G.lof[a] = (float)box.lo[a];
        G.hif[a] = (float)box.hi[a];
    }
    G.stridef[0] = (float)(lay.sx * 4);
    G.stridef[1] = (float)(lay.sy * 4);
    G.stridef[2] = 4.f;
    return G;
}

DDRR_HD float med3f(float v, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(v, lo, hi);  // one v_med3_f32, lo <= hi
#else
    return fminf(fmaxf(v, lo), hi);
#endif
}

}  // namespace ddrr
