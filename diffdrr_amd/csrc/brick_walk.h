// brick_walk.h -- the lean per-ray code of the volume-stationary Siddon kernel.
//
// The brick kernel is bound by vector-ALU issue (a gfx950 SIMD retires one wave64 VALU
// instruction per 4 cycles), so everything a lane executes per candidate pixel, per
// (ray, brick) hit and per voxel step is written out here with the instruction count in
// mind instead of reusing the general walk of siddon_core.h:
//   phase A  brick_candidate():  arithmetic-only conservative slab test of a detector
//            pixel against the brick, from the pose's affine detector model (no loads);
//   phase B  brick_trace():      exact clip of the real ray (same expressions as
//            siddon_core.h, so bricks meet exactly), entry cell, and the 3-way merge of
//            the plane crossings reading voxels from the LDS brick.
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

// Pixel bounding box of the lines through the source that meet the box `b` (plane indices;
// plane k sits at x = k - shift): the 8 corners are projected onto the pixel lattice.
DDRR_HD PixBox project_brick_grid(const PoseGrid &g, int det_h, int det_w, const Box &b,
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
        const float w[3] = {(float)((c & 1) ? b.hi[0] : b.lo[0]) - shift - g.s[0],
                            (float)((c & 2) ? b.hi[1] : b.lo[1]) - shift - g.s[1],
                            (float)((c & 4) ? b.hi[2] : b.lo[2]) - shift - g.s[2]};
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

// One row of the per-(pose, brick) table phase A reads (20 words).
constexpr int kRowWords = 20;
struct BrickRow {
    float D0[3];   // (t00 - s) + eps : direction of pixel (0, 0)
    float ei[3], ej[3];
    float P0[3];   // (lo - margin) - shift - s : numerators of the slab planes
    float P1[3];   // (hi + margin) - shift - s
    float inv_w;   // 1 / width of the pixel box
    int i0, j0, w, count;
};

// margin (voxels) by which phase A inflates the brick: covers the affine model's distance
// from the stored targets, so a ray with a real chord in the brick is never rejected
constexpr float kBrickMargin = 0.01f;

DDRR_HD BrickRow brick_row(const PoseGrid &g, const PixBox &pb, const Box &box, float shift,
                           float eps) {
    BrickRow r;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        r.D0[a] = (g.t00[a] - g.s[a]) + eps;
        r.ei[a] = g.ei[a];
        r.ej[a] = g.ej[a];
        r.P0[a] = ((float)box.lo[a] - kBrickMargin) - shift - g.s[a];
        r.P1[a] = ((float)box.hi[a] + kBrickMargin) - shift - g.s[a];
    }
    r.i0 = pb.i0;
    r.j0 = pb.j0;
    r.w = pb.j1 - pb.j0 + 1;
    r.count = pixbox_count(pb);
    r.inv_w = 1.0f / (float)(r.w > 0 ? r.w : 1);
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
    n_est = (exit - entry) * l1;
    return entry < exit;  // false for NaN
}

// ------------------------------------------------------------------ phase B

struct BrickGeom {
    float lof[3], hif[3];  // the brick's first / last plane index per axis, as floats
    int stride[3];         // BYTE strides of the LDS copy
};

DDRR_HD BrickGeom brick_geom(const Box &box, const BrickLayout &lay) {
    BrickGeom G;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        G.lof[a] = (float)box.lo[a];
        G.hif[a] = (float)box.hi[a];
    }
    G.stride[0] = lay.sx * 4;
    G.stride[1] = lay.sy * 4;
    G.stride[2] = 4;
    return G;
}

DDRR_HD float med3f(float v, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(v, lo, hi);  // one v_med3_f32, lo <= hi
#else
    return fminf(fmaxf(v, lo), hi);
#endif
}

// Exact clip + walk of one ray through one brick.  `fetch(byte offset)` reads the LDS
// copy.  Returns false if the ray does not cross the brick (phase A's margin let it
// through).  I = sum V dalpha over the brick.  With AUX, rec = {S0x, S0z, S1x, S1z} of the
// brick-local backward record (voxels outside the brick count as 0, so that the records
// of the bricks along a ray add up to the whole ray's: siddon_core.h SIDDON_AUX).
template <bool AUX, class Fetch>
DDRR_HD bool brick_trace(const Fetch &fetch, const BrickGeom &G, const float s[3],
                         const float t[3], float shift, float eps, float &I, float rec[4]) {
    float d[3], inv[3], c[3], mn[3];
    float entry = -INFINITY, exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // the same expressions as siddon_setup_fast: a plane shared by two bricks gets the
        // same alpha in both
        d[a] = (t[a] - s[a]) + eps;
#if defined(__HIP_DEVICE_COMPILE__)
        const float r0 = __builtin_amdgcn_rcpf(d[a]);
        inv[a] = fmaf(fmaf(-d[a], r0, 1.0f), r0, r0);
#else
        inv[a] = 1.0f / d[a];
#endif
        const float num = -shift - s[a];
        const float c0 = num * inv[a];
        c[a] = fmaf(fmaf(-c0, d[a], num), inv[a], c0);
        const float a_lo = fmaf(G.lof[a], inv[a], c[a]);
        const float a_hi = fmaf(G.hif[a], inv[a], c[a]);
        mn[a] = fminf(a_lo, a_hi);
        entry = fmaxf(entry, mn[a]);
        exit = fminf(exit, fmaxf(a_lo, a_hi));
    }
    I = 0.f;
    if (AUX) rec[0] = rec[1] = rec[2] = rec[3] = 0.f;
    if (!(entry < exit)) return false;  // also NaN

    // entry cell per axis (siddon_enter, incl. its alpha-order consistency rule)
    float k[3], an[3], dirf[3];
    int dstep[3];
    float offf = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const bool pos = d[a] > 0.f;
        const float p01 = pos ? 1.f : 0.f;
        dirf[a] = pos ? 1.f : -1.f;
        const float cmax = G.hif[a] - 1.f;
        float u = med3f(floorf(fmaf(entry, d[a], s[a] + shift)), G.lof[a], cmax);
        const float a_ahead = fmaf(u + p01, inv[a], c[a]);
        const float a_behind = fmaf(u + (1.f - p01), inv[a], c[a]);
        const float adj = (a_ahead < entry ? dirf[a] : 0.f) - (a_behind > entry ? dirf[a] : 0.f);
        u = med3f(u + adj, G.lof[a], cmax);
        u = (mn[a] == entry) ? (pos ? G.lof[a] : cmax) : u;  // entering axis: its face cell
        k[a] = u + p01;
        an[a] = fmaf(k[a], inv[a], c[a]);
        offf = fmaf(u - G.lof[a], (float)G.stride[a], offf);  // exact: < 2^24
        dstep[a] = pos ? G.stride[a] : -G.stride[a];
    }
    unsigned off = (unsigned)(int)offf;

    // which crossing opened the first segment, exclusive x > y > z (for the record)
    bool ox = mn[0] == entry;
    bool oz = !ox && !(mn[1] == entry);
    float a_cur = entry, acc = 0.f;
    float v = fetch(off), v_prev = 0.f;
    float S0x = 0.f, S1x = 0.f, S0z = 0.f, S1z = 0.f;
    for (int it = 0; it < 3 * BRICK + 3; ++it) {  // a brick holds < 3 * BRICK crossings
        const float a_next = fminf(fminf(an[0], an[1]), an[2]);
        const bool m0 = an[0] <= a_next, m1 = an[1] <= a_next, m2 = an[2] <= a_next;
        const bool cont = a_next < exit;
        const unsigned noff = off + (unsigned)((m0 ? dstep[0] : 0) + (m1 ? dstep[1] : 0) +
                                               (m2 ? dstep[2] : 0));
        // request the next voxel before the arithmetic of this step (stay put on the last)
        const float vn = fetch(cont ? noff : off);
        k[0] += m0 ? dirf[0] : 0.f;
        k[1] += m1 ? dirf[1] : 0.f;
        k[2] += m2 ? dirf[2] : 0.f;
        an[0] = fmaf(k[0], inv[0], c[0]);
        an[1] = fmaf(k[1], inv[1], c[1]);
        an[2] = fmaf(k[2], inv[2], c[2]);
        acc = fmaf(v, a_next - a_cur, acc);
        if (AUX) {
            // the crossing at a_cur that opened this segment: V_before - V_after
            const float dv = v_prev - v;
            const float dx = ox ? dv : 0.f, dz = oz ? dv : 0.f;
            S0x += dx;
            S1x = fmaf(dx, a_cur, S1x);
            S0z += dz;
            S1z = fmaf(dz, a_cur, S1z);
            ox = m0;
            oz = m2 && !m0 && !m1;
            v_prev = v;
        }
        a_cur = a_next;
        if (!cont) break;
        off = noff;
        v = vn;
    }
    I = acc;
    if (AUX) {
        // the crossing through which the ray leaves the brick (V_after = 0)
        const float dx = ox ? v_prev : 0.f, dz = oz ? v_prev : 0.f;
        rec[0] = S0x + dx;
        rec[1] = S0z + dz;
        rec[2] = fmaf(dx, a_cur, S1x);
        rec[3] = fmaf(dz, a_cur, S1z);
    }
    return true;
}

}  // namespace ddrr
