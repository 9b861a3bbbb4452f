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

// State of a ray at its entry into a brick: the exact clip (same expressions as
// siddon_setup_fast: a plane shared by two bricks gets the same alpha in both) and the
// entry cell (siddon_enter, incl. its alpha-order consistency rule).
struct BrickEntry {
    float inv[3], c[3], mn[3];
    float k[3], an[3], dirf[3];
    float entry, exit;
    float offc;  // voxel byte offset = offc + sum_a k_a stride_a
    bool hit;
};

// The voxel's byte offset is an affine function of the three plane counters,
//   off = sum_a (k_a - p01_a - lo_a) stride_a  (+ base: what the accessor wants added),
// exact in fp32 (< 2^24): three FMAs and a convert per step instead of three selects and
// two integer adds, and no per-axis step registers.
DDRR_HD BrickEntry brick_enter(const BrickGeom &G, const float s[3], const float t[3],
                               float shift, float eps, float base) {
    BrickEntry E;
    float d[3];
    E.entry = -INFINITY;
    E.exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;
#if defined(__HIP_DEVICE_COMPILE__)
        const float r0 = __builtin_amdgcn_rcpf(d[a]);
        E.inv[a] = fmaf(fmaf(-d[a], r0, 1.0f), r0, r0);
#else
        E.inv[a] = 1.0f / d[a];
#endif
        const float num = -shift - s[a];
        const float c0 = num * E.inv[a];
        E.c[a] = fmaf(fmaf(-c0, d[a], num), E.inv[a], c0);
        const float a_lo = fmaf(G.lof[a], E.inv[a], E.c[a]);
        const float a_hi = fmaf(G.hif[a], E.inv[a], E.c[a]);
        E.mn[a] = fminf(a_lo, a_hi);
        E.entry = fmaxf(E.entry, E.mn[a]);
        E.exit = fminf(E.exit, fmaxf(a_lo, a_hi));
    }
    E.hit = E.entry < E.exit;  // false for NaN
    E.offc = base;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const bool pos = d[a] > 0.f;
        const float p01 = pos ? 1.f : 0.f;
        E.dirf[a] = pos ? 1.f : -1.f;
        const float cmax = G.hif[a] - 1.f;
        float u = med3f(floorf(fmaf(E.entry, d[a], s[a] + shift)), G.lof[a], cmax);
        const float a_ahead = fmaf(u + p01, E.inv[a], E.c[a]);
        const float a_behind = fmaf(u + (1.f - p01), E.inv[a], E.c[a]);
        const float adj = (a_ahead < E.entry ? E.dirf[a] : 0.f) -
                          (a_behind > E.entry ? E.dirf[a] : 0.f);
        u = med3f(u + adj, G.lof[a], cmax);
        u = (E.mn[a] == E.entry) ? (pos ? G.lof[a] : cmax) : u;  // entering axis: face cell
        E.k[a] = u + p01;
        E.an[a] = fmaf(E.k[a], E.inv[a], E.c[a]);
        E.offc = fmaf(-(p01 + G.lof[a]), G.stridef[a], E.offc);
    }
    return E;
}

// Volume gradient of one ray through one brick: adds w * dalpha_k to the LDS cell of every
// voxel the ray crosses (d out / d V[k] = L dalpha_k, reference: grid_sampler_3d_backward
// behind renderers.py:159-164).  `add(base + byte offset, value)` is the scatter.
template <class Add>
DDRR_HD bool brick_scatter(const Add &add, float add_base, const BrickGeom &G, const float s[3],
                           const float t[3], float shift, float eps, float w) {
    const BrickEntry E = brick_enter(G, s, t, shift, eps, add_base);
    if (!E.hit) return false;
    float k[3] = {E.k[0], E.k[1], E.k[2]}, an[3] = {E.an[0], E.an[1], E.an[2]};
    float a_cur = E.entry;
    for (int it = 0; it < 3 * BRICK + 3; ++it) {
        const unsigned off = (unsigned)(int)fmaf(
            k[0], G.stridef[0], fmaf(k[1], G.stridef[1], fmaf(k[2], G.stridef[2], E.offc)));
        const float a_next = fminf(fminf(an[0], an[1]), an[2]);
        add(off, w * (a_next - a_cur));
        if (!(a_next < E.exit)) break;
        k[0] += an[0] <= a_next ? E.dirf[0] : 0.f;
        k[1] += an[1] <= a_next ? E.dirf[1] : 0.f;
        k[2] += an[2] <= a_next ? E.dirf[2] : 0.f;
        an[0] = fmaf(k[0], E.inv[0], E.c[0]);
        an[1] = fmaf(k[1], E.inv[1], E.c[1]);
        an[2] = fmaf(k[2], E.inv[2], E.c[2]);
        a_cur = a_next;
    }
    return true;
}

// Exact clip + walk of one ray through one brick.  `fetch(fetch_base + byte offset)` reads
// the LDS copy (fetch_base: 0 for a pointer-relative fetch, the brick's LDS address for
// LdsAbsFetch; the sum stays an exact fp32 integer).
// Returns false if the ray does not cross the brick (phase A's margin let it
// through).  I = sum V dalpha over the brick.  With AUX, rec = {S0x, S0z, S1x, S1z} of the
// brick-local backward record (voxels outside the brick count as 0, so that the records
// of the bricks along a ray add up to the whole ray's: siddon_core.h SIDDON_AUX).
template <bool AUX, class Fetch>
DDRR_HD bool brick_trace(const Fetch &fetch, float fetch_base, const BrickGeom &G,
                         const float s[3], const float t[3], float shift, float eps, float &I,
                         float rec[4]) {
    const BrickEntry E = brick_enter(G, s, t, shift, eps, fetch_base);
    I = 0.f;
    if (AUX) rec[0] = rec[1] = rec[2] = rec[3] = 0.f;
    if (!E.hit) return false;
    float k[3] = {E.k[0], E.k[1], E.k[2]}, an[3] = {E.an[0], E.an[1], E.an[2]};
    const float inv[3] = {E.inv[0], E.inv[1], E.inv[2]}, c[3] = {E.c[0], E.c[1], E.c[2]};
    const float dirf[3] = {E.dirf[0], E.dirf[1], E.dirf[2]}, mn[3] = {E.mn[0], E.mn[1], E.mn[2]};
    const float entry = E.entry, exit = E.exit, offc = E.offc;
#define DDRR_BRICK_OFF() \
    ((unsigned)(int)fmaf(k[0], G.stridef[0], fmaf(k[1], G.stridef[1], fmaf(k[2], G.stridef[2], offc))))
    unsigned off = DDRR_BRICK_OFF();

    // The walk, software-pipelined by hand.  Step i closes segment i (the ray inside voxel
    // i): it needs that voxel's VALUE only for the products, never for the geometry, so a
    // voxel is requested one step ahead (its address is known once the plane counters
    // have moved) and consumed one step late: the LDS latency (~100+ cycles with bank
    // conflicts) is covered by a full step of arithmetic instead of stalling every step.
    // Three value registers rotate (r[i % 3] holds voxel i); with the backward record the
    // crossing that OPENED segment i-1 is settled together with it: V_before - V_after =
    // voxel i-2 - voxel i-1, weighted by 1 and by the crossing's alpha for its axis
    // (exclusive attribution x > y > z, as in siddon_core.h).
    float r0 = fetch(off), r1 = 0.f, r2 = 0.f;  // voxel 0 requested; voxels -1, -2 := 0
    float len_p = 0.f;       // length of segment i-1
    float aop_p = entry;     // alpha of the crossing that opened segment i-1
    bool ox_p = false, oz_p = false;  // ... and its axis (x / z; y follows from the sums)
    // the crossing that opens segment 0 is the entry into the brick
    bool ox_c = mn[0] == entry;
    bool oz_c = !ox_c && !(mn[1] == entry);
    float a_cur = entry, acc = 0.f;
    float S0x = 0.f, S1x = 0.f, S0z = 0.f, S1z = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
    // (x, z) pairs of the record as 2-vectors: one v_pk_add_f32 and one v_pk_fma_f32 per step
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f S0v = {0.f, 0.f}, S1v = {0.f, 0.f};
#define DDRR_REC_ADD(dx, dz, al)                                   \
    {                                                              \
        const v2f dv2 = {dx, dz}, al2 = {al, al};                  \
        S0v += dv2;                                                \
        S1v = __builtin_elementwise_fma(dv2, al2, S1v);            \
    }
#else
#define DDRR_REC_ADD(dx, dz, al)  \
    {                             \
        S0x += dx;                \
        S1x = fmaf(dx, al, S1x);  \
        S0z += dz;                \
        S1z = fmaf(dz, al, S1z);  \
    }
#endif
    // put aside when the ray leaves: voxels i and i-1, the axis of the crossing that opened
    // segment i (e_fx) and of the exit crossing (e_fz)
    float e_rc = 0.f, e_rp = 0.f, e_fx = 0.f, e_fz = 0.f;

// RC: voxel i (requested during step i-1)   RN: voxel i+1 (requested now; still holds
// voxel i-2 at the top of the step)         RP: voxel i-1
#if defined(__HIP_DEVICE_COMPILE__)
#define DDRR_PIN(x) asm volatile("" : "+v"(x))
#else
#define DDRR_PIN(x) (void)(x)
#endif
#define DDRR_BRICK_STEP(IDX, RC, RN, RP)                                                          \
    {                                                                                          \
        /* settle segment i-1 and the crossing that opened it */                               \
        acc = fmaf(RP, len_p, acc);                                                            \
        if (AUX) {                                                                             \
            const float dv = RN - RP;                                                          \
            const float dx = ox_p ? dv : 0.f, dz = oz_p ? dv : 0.f;                            \
            DDRR_REC_ADD(dx, dz, aop_p);                                                       \
        }                                                                                      \
        /* geometry of step i */                                                               \
        const float a_next = fminf(fminf(an[0], an[1]), an[2]);                                \
        const bool m0 = an[0] <= a_next, m1 = an[1] <= a_next, m2 = an[2] <= a_next;           \
        const bool cont = a_next < exit;                                                       \
        k[0] += m0 ? dirf[0] : 0.f;                                                            \
        k[1] += m1 ? dirf[1] : 0.f;                                                            \
        k[2] += m2 ? dirf[2] : 0.f;                                                            \
        const unsigned noff = DDRR_BRICK_NEXT_OFF();                                           \
        RN = fetch(noff); /* voxel i+1 */                                                      \
        an[0] = fmaf(k[0], inv[0], c[0]);                                                      \
        an[1] = fmaf(k[1], inv[1], c[1]);                                                      \
        an[2] = fmaf(k[2], inv[2], c[2]);                                                      \
        len_p = a_next - a_cur;                                                                \
        aop_p = a_cur;                                                                         \
        ox_p = ox_c;                                                                           \
        oz_p = oz_c;                                                                           \
        ox_c = m0;                                                                             \
        oz_c = m2 && !m0 && !m1;                                                               \
        a_cur = a_next;                                                                        \
        if (!cont) {                                                                           \
            /* the ray leaves the brick.  This block runs at every step at which ANY lane of */ \
            /* the wave leaves, so it only puts aside what the settling after the loop needs */ \
            /* (the value registers rotate, the axis flags live in scalar masks).  DDRR_PIN   */ \
            /* keeps the uses of the voxel values inside this branch: hoisted above it they  */ \
            /* would make every step wait for the load it has just issued.                   */ \
            e_rc = RC;                                                                         \
            e_rp = RP;                                                                         \
            DDRR_PIN(e_rc);                                                                    \
            DDRR_PIN(e_rp);                                                                    \
            if (AUX) { /* axis of the two crossings: 1 x, 2 z, 0 y (x and z exclude each other) */ \
                e_fx = ox_p ? 1.f : (oz_p ? 2.f : 0.f);                                        \
                e_fz = ox_c ? 1.f : (oz_c ? 2.f : 0.f);                                        \
            }                                                                                  \
            break;                                                                             \
        }                                                                                      \
        off = noff;                                                                            \
    }
#if defined(__HIP_DEVICE_COMPILE__)
// (a ray that is leaving requests one cell beyond the brick: inside the workgroup's LDS
// allocation, or out of its range, which the hardware answers with 0; never used)
#define DDRR_BRICK_NEXT_OFF() DDRR_BRICK_OFF()
#else
#define DDRR_BRICK_NEXT_OFF() (cont ? DDRR_BRICK_OFF() : off)
#endif
    for (int it = 0; it < BRICK + 2; ++it) {  // a brick holds < 3 * BRICK crossings
        DDRR_BRICK_STEP(0, r0, r1, r2)
        DDRR_BRICK_STEP(1, r1, r2, r0)
        DDRR_BRICK_STEP(2, r2, r0, r1)
    }
#undef DDRR_BRICK_STEP
#undef DDRR_PIN
#undef DDRR_BRICK_NEXT_OFF
#undef DDRR_BRICK_OFF
#if defined(__HIP_DEVICE_COMPILE__)
    if (AUX) {
        S0x = S0v.x, S0z = S0v.y;
        S1x = S1v.x, S1z = S1v.y;
    }
#endif
#undef DDRR_REC_ADD
    // settle segment i, the crossing that opened it, and the exit crossing (V_after = 0)
    acc = fmaf(e_rc, len_p, acc);
    if (AUX) {
        const float dv = e_rp - e_rc;
        const bool xp = e_fx == 1.f, zp = e_fx == 2.f;  // the crossing that opened segment i
        const bool xc = e_fz == 1.f, zc = e_fz == 2.f;  // the exit crossing
        const float dx = xp ? dv : 0.f, dz = zp ? dv : 0.f;
        const float ex = xc ? e_rc : 0.f, ez = zc ? e_rc : 0.f;
        S0x += dx + ex;
        S1x = fmaf(ex, a_cur, fmaf(dx, aop_p, S1x));
        S0z += dz + ez;
        S1z = fmaf(ez, a_cur, fmaf(dz, aop_p, S1z));
    }
    I = acc;
    if (AUX) {
        rec[0] = S0x;
        rec[1] = S0z;
        rec[2] = S1x;
        rec[3] = S1z;
    }
    return true;
}

}  // namespace ddrr
