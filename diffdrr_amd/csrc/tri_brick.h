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
#include "brick_step.h"  // the packed word format (pack_voxel_label), bit casts
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

constexpr int TRI_AUX_PLANES = 7;  // sum T, sum dT (3), sum alpha dT (3)

// March one ray through one brick: sumT += T of every sample whose base corner is in the brick
// (`acc(addr)` fetches a voxel of the LDS copy).  AUX: also the backward record of those
// samples, rec = {sum dT_x, dT_y, dT_z, sum alpha dT_x, dT_y, dT_z} with dT the gradient of the
// zero-padded trilinear interpolant w.r.t. the index coordinates (fetch_trilinear's `grad`);
// everything trilinear_backward_ray needs follows from it: d = t - s + eps is constant along
// the ray, so sum (dT . d) = (sum dT) . d and sum u (dT . d) = ((sum alpha dT) . d - alphamin
// (sum dT) . d) / (alphamax - alphamin).
// `base` is what the accessor wants added to the brick-relative byte offset.
template <bool AUX, class Acc>
DDRR_HD bool tri_brick_march(const Acc &acc, float base, const TriGeom &G, const float s[3],
                             const float t[3], float shift, float eps, int P, float amin,
                             float amax, float &sumT, float rec[6]) {
    sumT = 0.f;
    if (AUX) rec[0] = rec[1] = rec[2] = rec[3] = rec[4] = rec[5] = 0.f;
    const float go = shift - 0.5f;  // align_corners = False: g = x + shift - 1/2
    float d[3], entry = -INFINITY, exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;
        const float g0 = s[a] + go;
        // (v_rcp_f32, 1 ulp: these bounds only pick the run of samples that is looked at -- with one
        // sample of slack on both sides -- the membership test below decides)
        const float inv = approx_rcp(d[a]);
        const float a1 = (G.lo[a] - g0) * inv, a2 = (G.lo[a] + (float)TRI_CELLS - g0) * inv;
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
    float sum = 0.f, Ax = 0.f, Ay = 0.f, Az = 0.f, Bx = 0.f, By = 0.f, Bz = 0.f;
    for (int m = m0; m <= m1; ++m) {
        const float al = fmaf(lin01(m, P, lstep), span, amin);  // renderers.py:224-225
        const float gx = fmaf(al, d[0], s[0]) + go;
        const float gy = fmaf(al, d[1], s[1]) + go;
        const float gz = fmaf(al, d[2], s[2]) + go;
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        // (one test, not a chain of three divergent regions: samples handed to a brick are mostly in it)
        const bool in = (fx >= G.lo[0]) & (fx < G.lo[0] + (float)TRI_CELLS) & (fy >= G.lo[1]) &
                        (fy < G.lo[1] + (float)TRI_CELLS) & (fz >= G.lo[2]) &
                        (fz < G.lo[2] + (float)TRI_CELLS);
        if (!in) continue;
        const float ax = gx - fx, ay = gy - fy, az = gz - fz;
        const float o00 = fmaf(fx, sx, fmaf(fy, sy, fmaf(fz, 4.f, offc)));  // exact, < 2^24
        const unsigned a00 = (unsigned)(int)o00;
        const unsigned a10 = (unsigned)(int)(o00 + sx), a01 = (unsigned)(int)(o00 + sy);
        const unsigned a11 = (unsigned)(int)(o00 + sx + sy);
        // aten grid_sampler_3d, corner order of trilinear_core.h fetch_trilinear: the four
        // (x, y) corner columns, each interpolated along z first
        const float v000 = acc(a00), v001 = acc(a00 + 4u), v100 = acc(a10), v101 = acc(a10 + 4u);
        const float v010 = acc(a01), v011 = acc(a01 + 4u), v110 = acc(a11), v111 = acc(a11 + 4u);
        const float wx0 = 1.f - ax, wy0 = 1.f - ay;
        const float dz00 = v001 - v000, dz10 = v101 - v100, dz01 = v011 - v010, dz11 = v111 - v110;
        const float l00 = fmaf(az, dz00, v000), l10 = fmaf(az, dz10, v100);
        const float l01 = fmaf(az, dz01, v010), l11 = fmaf(az, dz11, v110);
        float T = (wx0 * wy0) * l00;
        T = fmaf(ax * wy0, l10, T);
        T = fmaf(wx0 * ay, l01, T);
        T = fmaf(ax * ay, l11, T);
        sum += T;
        if (AUX) {
            // fetch_trilinear's gX, gY, gZ in the same summation order
            float gX = -wy0 * l00;
            gX = fmaf(wy0, l10, gX);
            gX = fmaf(-ay, l01, gX);
            gX = fmaf(ay, l11, gX);
            float gY = -wx0 * l00;
            gY = fmaf(-ax, l10, gY);
            gY = fmaf(wx0, l01, gY);
            gY = fmaf(ax, l11, gY);
            float gZ = (wx0 * wy0) * dz00;
            gZ = fmaf(ax * wy0, dz10, gZ);
            gZ = fmaf(wx0 * ay, dz01, gZ);
            gZ = fmaf(ax * ay, dz11, gZ);
            Ax += gX;
            Ay += gY;
            Az += gZ;
            Bx = fmaf(al, gX, Bx);
            By = fmaf(al, gY, By);
            Bz = fmaf(al, gZ, Bz);
        }
    }
    sumT = sum;
    if (AUX) {
        rec[0] = Ax, rec[1] = Ay, rec[2] = Az;
        rec[3] = Bx, rec[4] = By, rec[5] = Bz;
    }
    return true;
}

// mask_to_channels of the marcher on the bricks (renderers.py:242-252): the brick holds packed
// words -- value with a 16-bit mantissa, the voxel's label in the low byte (brick_step.h
// pack_voxel_label) -- and every sample of the ray whose base corner lies in the brick goes to
// the channel of the label of its NEAREST voxel.  That lookup is discontinuous: the voxel is
// rint() of the index coordinate by the reference's own chain of separately rounded fp32
// operations (trilinear_core.h march_exact_coord, as in the per-ray kernel); it is one of the
// sample's 8 corners, all staged (the halo; zeros -- label 0 -- outside the volume: the zero
// padding).  Runs of samples with one label are summed in a register and handed to
// `flush(label, sum of T)` when the label changes and at the end.
template <class Acc, class Flush>
DDRR_HD bool tri_brick_march_channels(const Acc &acc, float base, const TriGeom &G, const Dims D,
                                      const float s[3], const float t[3], float shift, float eps,
                                      int P, float amin, float amax, const Flush &flush) {
    const float go = shift - 0.5f;  // align_corners = False: g = x + shift - 1/2
    MarchSetup q;
    float entry = -INFINITY, exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q.d[a] = (t[a] - s[a]) + eps;
        const float g0 = s[a] + go;
        const float inv = approx_rcp(q.d[a]);  // (bounds of the run of samples looked at, see tri_brick_march)
        const float a1 = (G.lo[a] - g0) * inv, a2 = (G.lo[a] + (float)TRI_CELLS - g0) * inv;
        entry = fmaxf(entry, fminf(a1, a2));
        exit = fminf(exit, fmaxf(a1, a2));
    }
    q.span = amax - amin;
    if (!(entry < exit) || !(q.span > 0.f)) return false;
    const float lstep = 1.0f / (float)(P - 1), sc = (float)(P - 1) / q.span;
    const float f0 = fminf(fmaxf(floorf((entry - amin) * sc) - 1.f, 0.f), (float)P);
    const float f1 = fminf(fmaxf(ceilf((exit - amin) * sc) + 1.f, -1.f), (float)(P - 1));
    if (!(f0 <= f1)) return false;
    const int m0 = (int)f0, m1 = (int)f1;
    const float offc = fmaf(-G.lo[0], G.stridef[0],
                            fmaf(-G.lo[1], G.stridef[1], fmaf(-G.lo[2], G.stridef[2], base)));
    const float sx = G.stridef[0], sy = G.stridef[1];
    auto val = [](float w) { return bits_as_float(float_bits(w) & 0xffffff00u); };
    int cur = -1;
    float run = 0.f;
    for (int m = m0; m <= m1; ++m) {
        const float lin = lin01(m, P, lstep);
        const float al = fmaf(lin, q.span, amin);  // renderers.py:224-225
        const float gx = fmaf(al, q.d[0], s[0]) + go;
        const float gy = fmaf(al, q.d[1], s[1]) + go;
        const float gz = fmaf(al, q.d[2], s[2]) + go;
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        // (one test, not a chain of three divergent regions: samples handed to a brick are mostly in it)
        const bool in = (fx >= G.lo[0]) & (fx < G.lo[0] + (float)TRI_CELLS) & (fy >= G.lo[1]) &
                        (fy < G.lo[1] + (float)TRI_CELLS) & (fz >= G.lo[2]) &
                        (fz < G.lo[2] + (float)TRI_CELLS);
        if (!in) continue;
        const float ax = gx - fx, ay = gy - fy, az = gz - fz;
        const float o00 = fmaf(fx, sx, fmaf(fy, sy, fmaf(fz, 4.f, offc)));  // exact, < 2^24
        const unsigned a00 = (unsigned)(int)o00;
        const unsigned a10 = (unsigned)(int)(o00 + sx), a01 = (unsigned)(int)(o00 + sy);
        const unsigned a11 = (unsigned)(int)(o00 + sx + sy);
        const float v000 = val(acc(a00)), v001 = val(acc(a00 + 4u));
        const float v100 = val(acc(a10)), v101 = val(acc(a10 + 4u));
        const float v010 = val(acc(a01)), v011 = val(acc(a01 + 4u));
        const float v110 = val(acc(a11)), v111 = val(acc(a11 + 4u));
        const float wx0 = 1.f - ax, wy0 = 1.f - ay;
        const float l00 = fmaf(az, v001 - v000, v000), l10 = fmaf(az, v101 - v100, v100);
        const float l01 = fmaf(az, v011 - v010, v010), l11 = fmaf(az, v111 - v110, v110);
        float T = (wx0 * wy0) * l00;
        T = fmaf(ax * wy0, l10, T);
        T = fmaf(wx0 * ay, l01, T);
        T = fmaf(ax * ay, l11, T);
        // the label: nearest voxel by the reference's arithmetic, among the 8 corners
        float un[3];
        march_exact_coord(D, lin, q, amin, s, shift, false, un);
        const float rx = fminf(fmaxf(rintf(un[0]), fx), fx + 1.f);
        const float ry = fminf(fmaxf(rintf(un[1]), fy), fy + 1.f);
        const float rz = fminf(fmaxf(rintf(un[2]), fz), fz + 1.f);
        const unsigned an = (unsigned)(int)fmaf(rx, sx, fmaf(ry, sy, fmaf(rz, 4.f, offc)));
        const int lab = (int)(float_bits(acc(an)) & 0xffu);
        if (lab != cur) {
            if (cur >= 0) flush((unsigned)cur, run);
            cur = lab;
            run = 0.f;
        }
        run += T;
    }
    if (cur >= 0) flush((unsigned)cur, run);
    return true;
}

// Backward of tri_brick_march_channels w.r.t. the rays: the record of tri_brick_march<true> with
// every sample (its T and dT) multiplied by `weight(label)` -- the incoming gradient of the
// channel the sample's nearest voxel selects, fetched when the label changes along the ray.
// The brick holds the volume's own fp32 values (dT is made of DIFFERENCES of neighbouring voxels:
// on a smooth volume the packed words' 16-bit mantissas would cost it three digits); the label of
// the nearest voxel comes from `label(rx, ry, rz)` (the label map; 0 outside the volume).
template <class Acc, class Label, class Weight>
DDRR_HD bool tri_brick_march_weighted(const Acc &acc, const Label &label, float base,
                                      const TriGeom &G, const Dims D, const float s[3],
                                      const float t[3], float shift, float eps, int P, float amin,
                                      float amax, const Weight &weight, float &sumT,
                                      float rec[6]) {
    sumT = 0.f;
    rec[0] = rec[1] = rec[2] = rec[3] = rec[4] = rec[5] = 0.f;
    const float go = shift - 0.5f;
    MarchSetup q;
    float entry = -INFINITY, exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q.d[a] = (t[a] - s[a]) + eps;
        const float g0 = s[a] + go;
        const float inv = approx_rcp(q.d[a]);  // (bounds of the run of samples looked at, see tri_brick_march)
        const float a1 = (G.lo[a] - g0) * inv, a2 = (G.lo[a] + (float)TRI_CELLS - g0) * inv;
        entry = fmaxf(entry, fminf(a1, a2));
        exit = fminf(exit, fmaxf(a1, a2));
    }
    q.span = amax - amin;
    if (!(entry < exit) || !(q.span > 0.f)) return false;
    const float lstep = 1.0f / (float)(P - 1), sc = (float)(P - 1) / q.span;
    const float f0 = fminf(fmaxf(floorf((entry - amin) * sc) - 1.f, 0.f), (float)P);
    const float f1 = fminf(fmaxf(ceilf((exit - amin) * sc) + 1.f, -1.f), (float)(P - 1));
    if (!(f0 <= f1)) return false;
    const int m0 = (int)f0, m1 = (int)f1;
    const float offc = fmaf(-G.lo[0], G.stridef[0],
                            fmaf(-G.lo[1], G.stridef[1], fmaf(-G.lo[2], G.stridef[2], base)));
    const float sx = G.stridef[0], sy = G.stridef[1];
    int cur = -1;
    float g = 0.f;
    float sum = 0.f, Ax = 0.f, Ay = 0.f, Az = 0.f, Bx = 0.f, By = 0.f, Bz = 0.f;
    for (int m = m0; m <= m1; ++m) {
        const float lin = lin01(m, P, lstep);
        const float al = fmaf(lin, q.span, amin);  // renderers.py:224-225
        const float gx = fmaf(al, q.d[0], s[0]) + go;
        const float gy = fmaf(al, q.d[1], s[1]) + go;
        const float gz = fmaf(al, q.d[2], s[2]) + go;
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        // (one test, not a chain of three divergent regions: samples handed to a brick are mostly in it)
        const bool in = (fx >= G.lo[0]) & (fx < G.lo[0] + (float)TRI_CELLS) & (fy >= G.lo[1]) &
                        (fy < G.lo[1] + (float)TRI_CELLS) & (fz >= G.lo[2]) &
                        (fz < G.lo[2] + (float)TRI_CELLS);
        if (!in) continue;
        const float ax = gx - fx, ay = gy - fy, az = gz - fz;
        const float o00 = fmaf(fx, sx, fmaf(fy, sy, fmaf(fz, 4.f, offc)));  // exact, < 2^24
        const unsigned a00 = (unsigned)(int)o00;
        const unsigned a10 = (unsigned)(int)(o00 + sx), a01 = (unsigned)(int)(o00 + sy);
        const unsigned a11 = (unsigned)(int)(o00 + sx + sy);
        const float v000 = acc(a00), v001 = acc(a00 + 4u), v100 = acc(a10), v101 = acc(a10 + 4u);
        const float v010 = acc(a01), v011 = acc(a01 + 4u), v110 = acc(a11), v111 = acc(a11 + 4u);
        // the label as tri_brick_march_channels finds it
        float un[3];
        march_exact_coord(D, lin, q, amin, s, shift, false, un);
        const float rx = fminf(fmaxf(rintf(un[0]), fx), fx + 1.f);
        const float ry = fminf(fmaxf(rintf(un[1]), fy), fy + 1.f);
        const float rz = fminf(fmaxf(rintf(un[2]), fz), fz + 1.f);
        const int lab = (int)label(rx, ry, rz);
        if (lab != cur) {
            cur = lab;
            g = weight((unsigned)lab);
        }
        const float wx0 = 1.f - ax, wy0 = 1.f - ay;
        const float dz00 = v001 - v000, dz10 = v101 - v100, dz01 = v011 - v010, dz11 = v111 - v110;
        const float l00 = fmaf(az, dz00, v000), l10 = fmaf(az, dz10, v100);
        const float l01 = fmaf(az, dz01, v010), l11 = fmaf(az, dz11, v110);
        float T = (wx0 * wy0) * l00;
        T = fmaf(ax * wy0, l10, T);
        T = fmaf(wx0 * ay, l01, T);
        T = fmaf(ax * ay, l11, T);
        sum = fmaf(g, T, sum);
        float gX = -wy0 * l00;
        gX = fmaf(wy0, l10, gX);
        gX = fmaf(-ay, l01, gX);
        gX = fmaf(ay, l11, gX);
        float gY = -wx0 * l00;
        gY = fmaf(-ax, l10, gY);
        gY = fmaf(wx0, l01, gY);
        gY = fmaf(ax, l11, gY);
        float gZ = (wx0 * wy0) * dz00;
        gZ = fmaf(ax * wy0, dz10, gZ);
        gZ = fmaf(wx0 * ay, dz01, gZ);
        gZ = fmaf(ax * ay, dz11, gZ);
        gX *= g, gY *= g, gZ *= g;
        Ax += gX;
        Ay += gY;
        Az += gZ;
        Bx = fmaf(al, gX, Bx);
        By = fmaf(al, gY, By);
        Bz = fmaf(al, gZ, Bz);
    }
    sumT = sum;
    rec[0] = Ax, rec[1] = Ay, rec[2] = Az;
    rec[3] = Bx, rec[4] = By, rec[5] = Bz;
    return true;
}

// Backward of the sum-reduced march from the record of tri_brick_march<AUX> (the sums over
// ALL bricks): what trilinear_backward_ray computes by marching (align_corners = False, so
// d(index coordinate)/dx = 1).  gl = grad_out * ray length.
DDRR_HD MarchGrad trilinear_backward_from_record(float sumT, const float A[3], const float Bv[3],
                                                 const float s[3], const float t[3], float eps,
                                                 int P, float amin, float amax, float gl) {
    MarchGrad r;
    const float span = amax - amin, step = span / (float)(P - 1);
    const float k = gl * step;
    float Cd = 0.f, Bd = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float d = (t[a] - s[a]) + eps;
        r.gs[a] = k * (A[a] - Bv[a]);  // sum (1 - alpha) dT
        r.gt[a] = k * Bv[a];           // sum alpha dT
        Cd = fmaf(A[a], d, Cd);
        Bd = fmaf(Bv[a], d, Bd);
    }
    const float Cu = span > 0.f ? (Bd - amin * Cd) / span : 0.f;  // sum u (dT . d)
    const float ws = gl * sumT / (float)(P - 1);  // through step = (amax - amin)/(P-1)
    r.g_amin = k * (Cd - Cu) - ws;
    r.g_amax = k * Cu + ws;
    r.sumT = sumT;
    return r;
}

// One sample's eight corner updates inside an OWNER brick, without a branch per corner: a corner that
// is not owned adds 0 to the owned corner next to it (per axis: the plane of the owned corner, weight
// 0) -- eight unconditional adds cost less than the twelve divergent regions that pick the owned
// ones (round 5: the marcher's volume gradient 0.895 -> 0.777 ms at 512^3 -> 512^2, one pose), and
// adding 0 changes no sum.  x0 / x1 ...: the brick owns the corner at fx / fx + 1 ... (at least one per
// axis); (ax, ay, az): the sample's fractions; o = byte offset of (0, 0, 0); w: the sample's weight
// through acc_scale.
template <class Acc>
DDRR_HD void tri_owner_add8(const Acc &acc, bool x0, bool x1, bool y0, bool y1, bool z0, bool z1, float fx,
                            float fy, float fz, float ax, float ay, float az, float sx, float sy, float offc,
                            float w) {
    const float wxa = x0 ? 1.f - ax : 0.f, wxb = x1 ? ax : 0.f;
    const float wya = y0 ? 1.f - ay : 0.f, wyb = y1 ? ay : 0.f;
    const float wza = z0 ? 1.f - az : 0.f, wzb = z1 ? az : 0.f;
    // the first corner's plane per axis: fx if owned, else fx + 1 (then that is the owned one); the
    // second corner lies one stride further only if BOTH are owned
    const float o000 =
        fmaf(x0 ? fx : fx + 1.f, sx, fmaf(y0 ? fy : fy + 1.f, sy, fmaf(z0 ? fz : fz + 1.f, 4.f, offc)));
    const unsigned a000 = (unsigned)(int)o000;  // exact, < 2^24
    const unsigned dx = x0 && x1 ? (unsigned)(int)sx : 0u, dy = y0 && y1 ? (unsigned)(int)sy : 0u;
    const unsigned dz = z0 && z1 ? 4u : 0u;
    const float waa = w * (wxa * wya), wba = w * (wxb * wya), wab = w * (wxa * wyb), wbb = w * (wxb * wyb);
    acc_add(acc, a000, waa * wza, 0);
    acc_add(acc, a000 + dz, waa * wzb, 0);
    acc_add(acc, a000 + dx, wba * wza, 0);
    acc_add(acc, a000 + dx + dz, wba * wzb, 0);
    acc_add(acc, a000 + dy, wab * wza, 0);
    acc_add(acc, a000 + dy + dz, wab * wzb, 0);
    acc_add(acc, a000 + dx + dy, wbb * wza, 0);
    acc_add(acc, a000 + dx + dy + dz, wbb * wzb, 0);
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
        const float inv = approx_rcp(d[a]);  // (bounds of the run of samples looked at, see tri_brick_march)
        const float a1 = (lo[a] - 1.f - g0) * inv, a2 = (hi[a] - g0) * inv;
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
    const float ws = acc_scale(acc, w, 0);
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
        tri_owner_add8(acc, x0, x1, y0, y1, z0, z1, fx, fy, fz, ax, ay, az, sx, sy, offc, ws);
    }
}

// The marcher's mask_to_channels volume gradient inside one OWNER brick (reference
// renderers.py:242-252, autograd w.r.t. the volume): tri_owner_scatter with every sample weighted
// by the incoming gradient of the channel its NEAREST voxel's label selects -- the voxel
// tri_brick_march_channels picks (the reference's own coordinate chain, held to the sample's 8
// corners).  That voxel may lie one layer outside the owned box: `label(rx, ry, rz, owned, addr)`
// gets its index coordinates and, for an owned voxel, its LDS address; voxels outside the volume
// carry label 0 (the zero padding).
// `weight(label)` is gathered when the label changes along the ray; k = L step.
template <class Acc, class Label, class Weight>
DDRR_HD void tri_owner_scatter_weighted(const Acc &acc, const Label &label, const Weight &weight,
                                        float base, const float lo[3], const float hi[3],
                                        const float stridef[3], const Dims D, const float s[3],
                                        const float t[3], float shift, float eps, int P, float amin,
                                        float amax, float k) {
    const float go = shift - 0.5f;  // align_corners = False: g = x + shift - 1/2
    MarchSetup q;
    float entry = -INFINITY, exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q.d[a] = (t[a] - s[a]) + eps;
        const float g0 = s[a] + go;
        const float inv = approx_rcp(q.d[a]);  // (bounds of the run of samples looked at, see tri_brick_march)
        const float a1 = (lo[a] - 1.f - g0) * inv, a2 = (hi[a] - g0) * inv;
        entry = fmaxf(entry, fminf(a1, a2));
        exit = fminf(exit, fmaxf(a1, a2));
    }
    q.span = amax - amin;
    if (!(entry < exit) || !(q.span > 0.f)) return;
    const float lstep = 1.0f / (float)(P - 1), sc = (float)(P - 1) / q.span;
    const float f0 = fminf(fmaxf(floorf((entry - amin) * sc) - 1.f, 0.f), (float)P);
    const float f1 = fminf(fmaxf(ceilf((exit - amin) * sc) + 1.f, -1.f), (float)(P - 1));
    if (!(f0 <= f1)) return;
    const int m0 = (int)f0, m1 = (int)f1;
    const float offc = fmaf(-lo[0], stridef[0], fmaf(-lo[1], stridef[1], fmaf(-lo[2], stridef[2], base)));
    const float sx = stridef[0], sy = stridef[1];
    int cur = -1;
    float w = 0.f;
    for (int m = m0; m <= m1; ++m) {
        const float lin = lin01(m, P, lstep);
        const float al = fmaf(lin, q.span, amin);  // renderers.py:224-225
        const float gx = fmaf(al, q.d[0], s[0]) + go;
        const float gy = fmaf(al, q.d[1], s[1]) + go;
        const float gz = fmaf(al, q.d[2], s[2]) + go;
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        const bool x0 = fx >= lo[0] && fx < hi[0], x1 = fx + 1.f >= lo[0] && fx + 1.f < hi[0];
        const bool y0 = fy >= lo[1] && fy < hi[1], y1 = fy + 1.f >= lo[1] && fy + 1.f < hi[1];
        const bool z0 = fz >= lo[2] && fz < hi[2], z1 = fz + 1.f >= lo[2] && fz + 1.f < hi[2];
        if (!((x0 || x1) && (y0 || y1) && (z0 || z1))) continue;
        // the sample's label: its nearest voxel by the reference's arithmetic, among the 8 corners
        float un[3];
        march_exact_coord(D, lin, q, amin, s, shift, false, un);
        const float rx = fminf(fmaxf(rintf(un[0]), fx), fx + 1.f);
        const float ry = fminf(fmaxf(rintf(un[1]), fy), fy + 1.f);
        const float rz = fminf(fmaxf(rintf(un[2]), fz), fz + 1.f);
        const bool owned = rx >= lo[0] && rx < hi[0] && ry >= lo[1] && ry < hi[1] && rz >= lo[2] &&
                           rz < hi[2];
        const float an = fmaf(rx, sx, fmaf(ry, sy, fmaf(rz, 4.f, offc)));
        const int lab = (int)label(rx, ry, rz, owned, owned ? (unsigned)(int)an : 0u);
        if (lab != cur) {
            cur = lab;
            w = acc_scale(acc, weight((unsigned)lab) * k, 0);
        }
        const float ax = gx - fx, ay = gy - fy, az = gz - fz;
        tri_owner_add8(acc, x0, x1, y0, y1, z0, z1, fx, fy, fz, ax, ay, az, sx, sy, offc, w);
    }
}

}  // namespace ddrr
