// tri_brick.h -- the trilinear ray-marcher on the volume-stationary brick machinery.
//
// Reference semantics: diffdrr/renderers.py:205-241 (Trilinear.forward, mask=None,
// mode="bilinear", reducefn="sum", align_corners=False): P samples at
// alpha_m = alphamin + u_m (alphamax - alphamin), zero-padded trilinear lookups
// (aten grid_sampler_3d), out = L * step * sum_m T_m; see trilinear_core.h.
//
// A sample belongs to the brick that holds its BASE CORNER floor(g) (g = index
// coordinates): bricks are boxes of 31^3 base cells starting at index -1 (a sample with
// floor(g) = -1 still touches voxel 0), staged as 32^3 voxels [lo, lo + 32) -- the +1 halo
// is what the far corners need; voxels outside the volume are staged as zeros, which IS the
// zero padding.  Every brick evaluates a sample's g with the same expression and keeps it
// only if its base corner is inside, so the bricks partition the samples exactly.
//
// The volume gradient uses OWNER bricks instead (tri_owner_scatter): a brick owns the 32^3
// voxels [lo, lo + 32) -- the plain Siddon brick grid -- and visits every sample whose 8-cell
// touches one of them (base cell in [lo - 1, lo + 31]), adding only the corners it owns.
// Samples next to a brick face are visited by 2 (4, 8) bricks, ~20 % more visits in all, but a
// voxel's gradient is complete in one brick's LDS accumulator and is STORED: no halo, no global
// atomics (the halo layers of base-cell bricks cost 5768 scattered atomics per brick: 1.7 of
// 2.6 ms at 512^3, one pose).
#pragma once

#include "brick_core.h"
#include "ddrr_common.h"
#include "trilinear_core.h"

namespace ddrr {

constexpr int TRI_CELLS = BRICK - 1;  // base cells per brick edge

struct TriGeom {
    float lo[3];       // first base cell of the brick (>= -1), as float
    float stridef[3];  // BYTE strides of the LDS copy
};

DDRR_HD BrickGrid tri_brick_grid(const Dims D) {
    BrickGrid g;  // base cells run over -1 .. D - 1  (D + 1 of them)
    g.nx = (D.x + 1 + TRI_CELLS - 1) / TRI_CELLS;
    g.ny = (D.y + 1 + TRI_CELLS - 1) / TRI_CELLS;
    g.nz = (D.z + 1 + TRI_CELLS - 1) / TRI_CELLS;
    return g;
}

// first base cell of brick `id` per axis (the stored voxels are lo .. lo + 31)
DDRR_HD void tri_brick_lo(const BrickGrid &g, int id, int lo[3]) {
    const int bz = id % g.nz, by = (id / g.nz) % g.ny, bx = id / (g.nz * g.ny);
    lo[0] = bx * TRI_CELLS - 1;
    lo[1] = by * TRI_CELLS - 1;
    lo[2] = bz * TRI_CELLS - 1;
}

DDRR_HD TriGeom tri_geom(const int lo[3], const BrickLayout &lay) {
    TriGeom G;
#pragma unroll
    for (int a = 0; a < 3; ++a) G.lo[a] = (float)lo[a];
    G.stridef[0] = (float)(lay.sx * 4);
    G.stridef[1] = (float)(lay.sy * 4);
    G.stridef[2] = 4.f;
    return G;
}

// March one ray through one brick.
//   SCATTER = false: sumT += T of every sample whose base corner is in the brick
//                    (`acc(addr)` fetches a voxel of the LDS copy);
//   SCATTER = true : the sample's 8 corner weights times `w` are added to the LDS
//                    accumulator (`acc(addr, value)`): the volume gradient, d out / d V[c] =
//                    L step w_c (grid_sampler_3d_backward, bilinear).
// `base` is what the accessor wants added to the brick-relative byte offset.
template <bool SCATTER, class Acc>
DDRR_HD bool tri_brick_march(const Acc &acc, float base, const TriGeom &G, const float s[3],
                             const float t[3], float shift, float eps, int P, float amin,
                             float amax, float w, float &sumT) {
    sumT = 0.f;
    const float go = shift - 0.5f;  // align_corners = False: g = x + shift - 1/2
    float d[3], entry = -INFINITY, exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;
        const float g0 = s[a] + go;
        const float a1 = (G.lo[a] - g0) / d[a], a2 = (G.lo[a] + (float)TRI_CELLS - g0) / d[a];
        entry = fmaxf(entry, fminf(a1, a2));
        exit = fminf(exit, fmaxf(a1, a2));
    }
    const float span = amax - amin;
    if (!(entry < exit) || !(span > 0.f)) return false;
    const float lstep = 1.0f / (float)(P - 1), sc = (float)(P - 1) / span;
    // samples that may fall in the brick: one of slack on both sides, the membership test
    // below is what decides
    const float f0 = fminf(fmaxf(floorf((entry - amin) * sc) - 1.f, 0.f), (float)P);
    const float f1 = fminf(fmaxf(ceilf((exit - amin) * sc) + 1.f, -1.f), (float)(P - 1));
    if (!(f0 <= f1)) return false;
    const int m0 = (int)f0, m1 = (int)f1;
    const float offc = fmaf(-G.lo[0], G.stridef[0],
                            fmaf(-G.lo[1], G.stridef[1], fmaf(-G.lo[2], G.stridef[2], base)));
    const float sx = G.stridef[0], sy = G.stridef[1];
    float sum = 0.f;
    for (int m = m0; m <= m1; ++m) {
        const float al = fmaf(lin01(m, P, lstep), span, amin);  // renderers.py:224-225
        const float gx = fmaf(al, d[0], s[0]) + go;
        const float gy = fmaf(al, d[1], s[1]) + go;
        const float gz = fmaf(al, d[2], s[2]) + go;
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        const bool in = fx >= G.lo[0] && fx < G.lo[0] + (float)TRI_CELLS && fy >= G.lo[1] &&
                        fy < G.lo[1] + (float)TRI_CELLS && fz >= G.lo[2] &&
                        fz < G.lo[2] + (float)TRI_CELLS;
        if (!in) continue;
        const float ax = gx - fx, ay = gy - fy, az = gz - fz;
        const float o00 = fmaf(fx, sx, fmaf(fy, sy, fmaf(fz, 4.f, offc)));  // exact, < 2^24
        const unsigned a00 = (unsigned)(int)o00;
        const unsigned a10 = (unsigned)(int)(o00 + sx), a01 = (unsigned)(int)(o00 + sy);
        const unsigned a11 = (unsigned)(int)(o00 + sx + sy);
        if constexpr (!SCATTER) {
            // aten grid_sampler_3d, corner order of trilinear_core.h fetch_trilinear
            float T = 0.f;
            {
                const float v0 = acc(a00), v1 = acc(a00 + 4u);
                T = fmaf((1.f - ax) * (1.f - ay), fmaf(az, v1 - v0, v0), T);
            }
            {
                const float v0 = acc(a10), v1 = acc(a10 + 4u);
                T = fmaf(ax * (1.f - ay), fmaf(az, v1 - v0, v0), T);
            }
            {
                const float v0 = acc(a01), v1 = acc(a01 + 4u);
                T = fmaf((1.f - ax) * ay, fmaf(az, v1 - v0, v0), T);
            }
            {
                const float v0 = acc(a11), v1 = acc(a11 + 4u);
                T = fmaf(ax * ay, fmaf(az, v1 - v0, v0), T);
            }
            sum += T;
        } else {
            const float wx0 = 1.f - ax, wy0 = 1.f - ay, wz0 = 1.f - az;
            acc(a00, w * (wx0 * wy0 * wz0));
            acc(a00 + 4u, w * (wx0 * wy0 * az));
            acc(a10, w * (ax * wy0 * wz0));
            acc(a10 + 4u, w * (ax * wy0 * az));
            acc(a01, w * (wx0 * ay * wz0));
            acc(a01 + 4u, w * (wx0 * ay * az));
            acc(a11, w * (ax * ay * wz0));
            acc(a11 + 4u, w * (ax * ay * az));
        }
    }
    sumT = sum;
    return true;
}

// Volume gradient of one ray inside one OWNER brick: voxels [lo, hi) per axis (hi - lo <= 32),
// staged at LDS offset (v - lo) . stride.  `acc(addr, value)` adds into the LDS accumulator;
// corners outside [lo, hi) belong to a neighbour (or lie outside the volume) and are skipped.
template <class Acc>
DDRR_HD void tri_owner_scatter(const Acc &acc, float base, const float lo[3], const float hi[3],
                               const float stridef[3], const float s[3], const float t[3],
                               float shift, float eps, int P, float amin, float amax, float w) {
    const float go = shift - 0.5f;  // align_corners = False: g = x + shift - 1/2
    float d[3], entry = -INFINITY, exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;
        const float g0 = s[a] + go;
        // base cells lo - 1 .. hi - 1  <=>  g in [lo - 1, hi)
        const float a1 = (lo[a] - 1.f - g0) / d[a], a2 = (hi[a] - g0) / d[a];
        entry = fmaxf(entry, fminf(a1, a2));
        exit = fminf(exit, fmaxf(a1, a2));
    }
    const float span = amax - amin;
    if (!(entry < exit) || !(span > 0.f)) return;
    const float lstep = 1.0f / (float)(P - 1), sc = (float)(P - 1) / span;
    const float f0 = fminf(fmaxf(floorf((entry - amin) * sc) - 1.f, 0.f), (float)P);
    const float f1 = fminf(fmaxf(ceilf((exit - amin) * sc) + 1.f, -1.f), (float)(P - 1));
    if (!(f0 <= f1)) return;
    const int m0 = (int)f0, m1 = (int)f1;
    const float offc = fmaf(-lo[0], stridef[0], fmaf(-lo[1], stridef[1], fmaf(-lo[2], stridef[2], base)));
    const float sx = stridef[0], sy = stridef[1];
    for (int m = m0; m <= m1; ++m) {
        const float al = fmaf(lin01(m, P, lstep), span, amin);  // renderers.py:224-225
        const float gx = fmaf(al, d[0], s[0]) + go;
        const float gy = fmaf(al, d[1], s[1]) + go;
        const float gz = fmaf(al, d[2], s[2]) + go;
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        // which of the two corners per axis this brick owns
        const bool x0 = fx >= lo[0] && fx < hi[0], x1 = fx + 1.f >= lo[0] && fx + 1.f < hi[0];
        const bool y0 = fy >= lo[1] && fy < hi[1], y1 = fy + 1.f >= lo[1] && fy + 1.f < hi[1];
        const bool z0 = fz >= lo[2] && fz < hi[2], z1 = fz + 1.f >= lo[2] && fz + 1.f < hi[2];
        if (!((x0 || x1) && (y0 || y1) && (z0 || z1))) continue;
        const float ax = gx - fx, ay = gy - fy, az = gz - fz;
        const float o00 = fmaf(fx, sx, fmaf(fy, sy, fmaf(fz, 4.f, offc)));
        // (offsets of corners that are not owned may be negative: never converted or used)
        const float wx0 = 1.f - ax, wy0 = 1.f - ay, wz0 = 1.f - az;
        if (x0 && y0) {
            const unsigned a00 = (unsigned)(int)(o00 + (z0 ? 0.f : 4.f));
            if (z0) acc(a00, w * (wx0 * wy0 * wz0));
            if (z1) acc(z0 ? a00 + 4u : a00, w * (wx0 * wy0 * az));
        }
        if (x1 && y0) {
            const unsigned a10 = (unsigned)(int)(o00 + sx + (z0 ? 0.f : 4.f));
            if (z0) acc(a10, w * (ax * wy0 * wz0));
            if (z1) acc(z0 ? a10 + 4u : a10, w * (ax * wy0 * az));
        }
        if (x0 && y1) {
            const unsigned a01 = (unsigned)(int)(o00 + sy + (z0 ? 0.f : 4.f));
            if (z0) acc(a01, w * (wx0 * ay * wz0));
            if (z1) acc(z0 ? a01 + 4u : a01, w * (wx0 * ay * az));
        }
        if (x1 && y1) {
            const unsigned a11 = (unsigned)(int)(o00 + sx + sy + (z0 ? 0.f : 4.f));
            if (z0) acc(a11, w * (ax * ay * wz0));
            if (z1) acc(z0 ? a11 + 4u : a11, w * (ax * ay * az));
        }
    }
}

}  // namespace ddrr
