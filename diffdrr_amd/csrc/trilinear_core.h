// trilinear_core.h -- per-ray trilinear ray-marcher for one lane.
//
// What it replaces: the reference's Trilinear renderer,
//   diffdrr/renderers.py:205-241  Trilinear.forward (mask=None branch)
//   diffdrr/renderers.py:143-169  _get_xyzs / _get_voxel (grid_sample "bilinear", zeros padding)
// n_points samples at alpha_m = alphamin + u_m (alphamax - alphamin), u = linspace(0, 1, P),
// rectangular rule with step (alphamax - alphamin)/(P - 1).  alphamin/alphamax
// are inputs: the host side evaluates the reference's batch-global range
// (renderers.py:220-223) or forwards the caller's values.
//
// A lane owns a ray; samples whose 8-cell cannot touch the volume are skipped
// (they are exact zeros in the reference), the rest cost four 8-byte fetches
// (the two z-neighbours of a corner pair are adjacent in memory).
#pragma once

#include "ddrr_common.h"
#include "siddon_core.h"  // GridMap, fetch_nearest, fetch_trilinear

namespace ddrr {

// torch.linspace(0, 1, P)[m] as aten's scalar (and GPU) kernel evaluates it.
DDRR_HD float lin01(int m, int P, float lstep) {
    // (both halves computed and one selected: as a branch it is a divergent region per sample)
    const float up = (float)m * lstep, down = 1.0f - (float)(P - 1 - m) * lstep;
    return m < P / 2 ? up : down;
}

// The marching range of one ray: first / last intersection with the volume enlarged by one
// voxel, clipped to [0, 1] (reference renderers.py:124-140, the same operations in the same
// order: IEEE division of (plane - s) by (t - s) + eps; far plane at dims + 1 - shift).
DDRR_HD void ray_alpha_range(const Dims D, const float s[3], const float t[3], float shift,
                             float eps, float &amin, float &amax) {
    const float hi[3] = {(float)D.x + 1.f - shift, (float)D.y + 1.f - shift,
                         (float)D.z + 1.f - shift};
    amin = -INFINITY;
    amax = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float sdd = (t[a] - s[a]) + eps;
        const float a0 = (-shift - s[a]) / sdd, a1 = (hi[a] - s[a]) / sdd;
        // (torch.minimum / maximum propagate NaN; fminf / fmaxf drop it: a NaN here is a ray
        // with s == t on an axis, which the march handles by its own clip)
        amin = fmaxf(amin, fminf(a0, a1));
        amax = fminf(amax, fmaxf(a0, a1));
    }
    amin = fmaxf(amin, 0.f);
    amax = fminf(amax, 1.f);
}

struct MarchSetup {
    float d[3];
    float span, step, lstep;
    int m_lo, m_hi;  // inclusive range of samples that may touch the volume
};

DDRR_HD MarchSetup march_setup(const Dims D, const GridMap &g, const float s[3], const float t[3],
                               float eps, int P, float amin, float amax) {
    MarchSetup q;
    const int Dn[3] = {D.x, D.y, D.z};
    float lo = -INFINITY, hi = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q.d[a] = (t[a] - s[a]) + eps;
        // index coordinate along the ray: gi(alpha) = g0 + alpha * gd; a sample can
        // contribute only while -1 < gi < D on every axis
        const float g0 = fmaf(s[a], g.k[a], g.o[a]);
        const float gd = q.d[a] * g.k[a];
        const float a1 = (-1.f - g0) / gd, a2 = ((float)Dn[a] - g0) / gd;
        lo = fmaxf(lo, fminf(a1, a2));
        hi = fminf(hi, fmaxf(a1, a2));
    }
    q.span = amax - amin;
    q.step = q.span / (float)(P - 1);  // renderers.py:235
    q.lstep = 1.0f / (float)(P - 1);
    q.m_lo = 0;
    q.m_hi = P - 1;
    if (q.span > 0.f) {
        const float sc = (float)(P - 1) / q.span;
        // two samples of slack on both sides; the per-corner bounds tests stay exact
        const float flo = fminf(fmaxf(floorf((lo - amin) * sc) - 2.f, 0.f), (float)P);
        const float fhi = fminf(fmaxf(ceilf((hi - amin) * sc) + 2.f, -1.f), (float)(P - 1));
        if (flo == flo) q.m_lo = (int)flo;  // NaN guard
        if (fhi == fhi) q.m_hi = (int)fhi;
    }
    return q;
}

// The index coordinate of a sample by the reference's own chain of separately rounded fp32
// tensor operations: alphas = linspace * (alphamax - alphamin) + alphamin (renderers.py:225),
// x = s + alpha d, 2 (x + shift) / D - 1 (:148-152), then aten's un-normalise.  For the
// discontinuous lookups (labels, mode="nearest"), where the side of a voxel boundary a sample
// falls on is decided by exactly this arithmetic.
DDRR_HD void march_exact_coord(const Dims D, float lin, const MarchSetup &q, float amin,
                               const float s[3], float shift, bool align_corners, float un[3]) {
    const float al_ref = add_rn(mul_rn(lin, q.span), amin);
    const float Dn[3] = {(float)D.x, (float)D.y, (float)D.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = add_rn(s[a], mul_rn(al_ref, q.d[a]));
        const float nrm = add_rn(mul_rn(2.f, add_rn(x, shift)) / Dn[a], -1.f);
        un[a] = align_corners ? mul_rn(add_rn(nrm, 1.f) / 2.f, Dn[a] - 1.f)
                              : add_rn(mul_rn(add_rn(nrm, 1.f), Dn[a]), -1.f) / 2.f;
    }
}

// mode="nearest" of the marcher: the voxel lookup is as discontinuous as the label lookup, so
// the sample's index coordinate is the reference's own chain as well (gx, gy, gz are replaced)
#define DDRR_MARCH_NEAREST_COORD(lin)                                              \
    if (NEAREST) {                                                                 \
        float un_[3];                                                              \
        march_exact_coord(D, (lin), q, amin, s, shift, align_corners, un_);        \
        gx = un_[0];                                                               \
        gy = un_[1];                                                               \
        gz = un_[2];                                                               \
    }

template <int REDUCE, bool NEAREST>
DDRR_HD float trilinear_forward_ray(const float *__restrict__ vol, const Dims D, const float s[3],
                                    const float t[3], float shift, float eps, int P, float amin,
                                    float amax, bool align_corners) {
    const GridMap g = make_gridmap(D, shift, align_corners);
    const MarchSetup q = march_setup(D, g, s, t, eps, P, amin, amax);
    const bool skipped = q.m_lo > 0 || q.m_hi < P - 1;
    float acc = (REDUCE == REDUCE_SUM || skipped) ? 0.f : -INFINITY;
    for (int m = q.m_lo; m <= q.m_hi; ++m) {
        const float al = fmaf(lin01(m, P, q.lstep), q.span, amin);  // renderers.py:224-225
        float gx = fmaf(fmaf(al, q.d[0], s[0]), g.k[0], g.o[0]);
        float gy = fmaf(fmaf(al, q.d[1], s[1]), g.k[1], g.o[1]);
        float gz = fmaf(fmaf(al, q.d[2], s[2]), g.k[2], g.o[2]);
        DDRR_MARCH_NEAREST_COORD(lin01(m, P, q.lstep))
        const float v = NEAREST ? fetch_nearest(vol, D, gx, gy, gz)
                                : fetch_trilinear(vol, D, gx, gy, gz, nullptr, false);
        if (REDUCE == REDUCE_SUM)
            acc += v;
        else
            acc = fmaxf(acc, v);
    }
    return acc * q.step;  // caller multiplies by the ray length
}

// Label of the sample with linspace value `lin` (mask_to_channels, renderers.py:242-252).
// The label lookup is discontinuous, and the first sample of the ray that sets the
// batch-global alphamin sits exactly ON the volume's face (g = -1/2): which side it falls on
// is decided by the reference's fp32 round trip through normalised coordinates
// (renderers.py:152, then aten's un-normalisation), so that is restated here for the label
// (the interpolated value is continuous and does not need it).  alphas = linspace *
// (alphamax - alphamin) + alphamin; x = s + alpha * d; each operation rounded on its own, as
// the reference's tensor ops are.  0 outside the volume.
DDRR_HD int march_label(const unsigned char *__restrict__ labels, const Dims D, float lin,
                        const MarchSetup &q, float amin, const float s[3], float shift,
                        bool align_corners) {
    float un[3];
    march_exact_coord(D, lin, q, amin, s, shift, align_corners, un);
    const float rx = rintf(un[0]), ry = rintf(un[1]), rz = rintf(un[2]);  // == nearbyint
    const bool in = rx >= 0.f && rx < (float)D.x && ry >= 0.f && ry < (float)D.y && rz >= 0.f &&
                    rz < (float)D.z;
    return in ? (int)labels[((int)rx * D.y + (int)ry) * D.z + (int)rz] : 0;
}

// The per-sample tensor the reference hands to a callable reducefn (renderers.py:226-238):
// samples[m] = L * step * T(V, x(alpha_m)), m = 0 .. P - 1 (zero outside the volume).
template <bool NEAREST>
DDRR_HD void trilinear_samples_ray(const float *__restrict__ vol, const Dims D, const float s[3],
                                   const float t[3], float shift, float eps, int P, float amin,
                                   float amax, bool align_corners, float L,
                                   float *__restrict__ samples, long stride) {
    const GridMap g = make_gridmap(D, shift, align_corners);
    const MarchSetup q = march_setup(D, g, s, t, eps, P, amin, amax);
    const float k = L * q.step;
    for (int m = 0; m < P; ++m) {
        float v = 0.f;
        if (m >= q.m_lo && m <= q.m_hi) {
            const float al = fmaf(lin01(m, P, q.lstep), q.span, amin);
            float gx = fmaf(fmaf(al, q.d[0], s[0]), g.k[0], g.o[0]);
            float gy = fmaf(fmaf(al, q.d[1], s[1]), g.k[1], g.o[1]);
            float gz = fmaf(fmaf(al, q.d[2], s[2]), g.k[2], g.o[2]);
            DDRR_MARCH_NEAREST_COORD(lin01(m, P, q.lstep))
            v = NEAREST ? fetch_nearest(vol, D, gx, gy, gz)
                        : fetch_trilinear(vol, D, gx, gy, gz, nullptr, false);
        }
        samples[m * stride] = k * v;
    }
}

// mask_to_channels for the marcher (renderers.py:242-252): every sample's value goes to
// the channel of the label found by a NEAREST lookup of the mask at the sample point
// (0 outside the volume).  The ray owns its output column: a run of samples with one
// label is summed in a register and handed to `flush(label, sum)` when the label changes.
template <class Flush>
DDRR_HD void trilinear_channels_ray(const float *__restrict__ vol,
                                    const unsigned char *__restrict__ labels, const Dims D,
                                    const float s[3], const float t[3], float shift, float eps,
                                    int P, float amin, float amax, bool align_corners,
                                    Flush flush) {
    const GridMap g = make_gridmap(D, shift, align_corners);
    const MarchSetup q = march_setup(D, g, s, t, eps, P, amin, amax);
    int cur = -1;
    float run = 0.f;
    for (int m = q.m_lo; m <= q.m_hi; ++m) {
        const float al = fmaf(lin01(m, P, q.lstep), q.span, amin);
        const float gx = fmaf(fmaf(al, q.d[0], s[0]), g.k[0], g.o[0]);
        const float gy = fmaf(fmaf(al, q.d[1], s[1]), g.k[1], g.o[1]);
        const float gz = fmaf(fmaf(al, q.d[2], s[2]), g.k[2], g.o[2]);
        const float v = fetch_trilinear(vol, D, gx, gy, gz, nullptr, false);
        const int lab = march_label(labels, D, lin01(m, P, q.lstep), q, amin, s, shift, align_corners);
        if (lab != cur) {
            if (cur >= 0) flush(cur, run * q.step);
            cur = lab;
            run = 0.f;
        }
        run += v;
    }
    if (cur >= 0) flush(cur, run * q.step);
}

struct MarchGrad {
    float gs[3], gt[3];  // through the sample positions x = s + alpha (t - s + eps)
    float g_amin, g_amax;
    float sumT;  // sum of samples (d out / d img = g * step * sumT)
};

// Per-sample weight of the backward pass: 1 for the plain march; with mask_to_channels the
// incoming gradient of the channel the sample's label selects (grad_out is (B, C, N): `gcol`
// points at [b, 0, n], channels are `stride` apart).  The label has no gradient of its own.
struct UnitWeight {
    DDRR_HD float operator()(int, float, const MarchSetup &) const { return 1.f; }
};
// a callable reducefn: the incoming gradient of sample m itself (grad is (B, P, N): `g` points
// at [b, 0, n], samples are `stride` apart)
struct SampleWeight {
    const float *g;
    long stride;
    DDRR_HD float operator()(int m, float, const MarchSetup &) const { return g[m * stride]; }
};
struct LabelWeight {
    const unsigned char *labels;
    Dims D;
    const float *gcol;
    long stride;
    int C;
    float amin, shift;
    float s[3];
    bool align_corners;
    DDRR_HD float operator()(int, float lin, const MarchSetup &q) const {
        const int lab = march_label(labels, D, lin, q, amin, s, shift, align_corners);
        return lab < C ? gcol[lab * stride] : 0.f;
    }
};

// reducefn = "max" (renderers.py:178-179): autograd routes the gradient to the arg-max sample
// alone (the first one on ties).  Index of that sample, or -1 if the maximum is one of the
// skipped samples outside the volume (value 0, no gradient).
template <bool NEAREST>
DDRR_HD int trilinear_argmax_ray(const float *__restrict__ vol, const Dims D, const float s[3],
                                 const float t[3], float shift, float eps, int P, float amin,
                                 float amax, bool align_corners) {
    const GridMap g = make_gridmap(D, shift, align_corners);
    const MarchSetup q = march_setup(D, g, s, t, eps, P, amin, amax);
    float best = q.m_lo > 0 ? 0.f : -INFINITY;
    int idx = -1;
    for (int m = q.m_lo; m <= q.m_hi; ++m) {
        const float al = fmaf(lin01(m, P, q.lstep), q.span, amin);
        float gx = fmaf(fmaf(al, q.d[0], s[0]), g.k[0], g.o[0]);
        float gy = fmaf(fmaf(al, q.d[1], s[1]), g.k[1], g.o[1]);
        float gz = fmaf(fmaf(al, q.d[2], s[2]), g.k[2], g.o[2]);
        DDRR_MARCH_NEAREST_COORD(lin01(m, P, q.lstep))
        const float v = NEAREST ? fetch_nearest(vol, D, gx, gy, gz)
                                : fetch_trilinear(vol, D, gx, gy, gz, nullptr, false);
        if (v > best) {
            best = v;
            idx = m;
        }
    }
    if (q.m_hi < P - 1 && 0.f > best) idx = -1;
    return idx;
}
struct OneSampleWeight {
    int m_star;
    DDRR_HD float operator()(int m, float, const MarchSetup &) const {
        return m == m_star ? 1.f : 0.f;
    }
};

// Backward of the sum-reduced march for one ray.
// gl = grad_out * ray length (the ray length alone when `wt` carries the gradient).
template <bool NEAREST, bool WANT_VOL, class Add, class Weight = UnitWeight>
DDRR_HD MarchGrad trilinear_backward_ray(const float *__restrict__ vol, const Dims D,
                                         const float s[3], const float t[3], float shift,
                                         float eps, int P, float amin, float amax,
                                         bool align_corners, float gl, Add add,
                                         Weight wt = Weight()) {
    const GridMap g = make_gridmap(D, shift, align_corners);
    const MarchSetup q = march_setup(D, g, s, t, eps, P, amin, amax);
    MarchGrad r;
    float A[3] = {0.f, 0.f, 0.f};   // sum dT
    float Bv[3] = {0.f, 0.f, 0.f};  // sum alpha dT
    float Cu = 0.f, Cd = 0.f;       // sum u (dT . d), sum (dT . d)
    float sumT = 0.f;
    const float k = gl * q.step;
    for (int m = q.m_lo; m <= q.m_hi; ++m) {
        const float u = lin01(m, P, q.lstep);
        const float al = fmaf(u, q.span, amin);
        float gx = fmaf(fmaf(al, q.d[0], s[0]), g.k[0], g.o[0]);
        float gy = fmaf(fmaf(al, q.d[1], s[1]), g.k[1], g.o[1]);
        float gz = fmaf(fmaf(al, q.d[2], s[2]), g.k[2], g.o[2]);
        DDRR_MARCH_NEAREST_COORD(u)
        const float w = wt(m, u, q);
        if (NEAREST) {
            sumT = fmaf(w, fetch_nearest(vol, D, gx, gy, gz), sumT);
            if (WANT_VOL) scatter_nearest(D, gx, gy, gz, k * w, add);
        } else {
            float dT[3];
            sumT = fmaf(w, fetch_trilinear(vol, D, gx, gy, gz, dT, true), sumT);
            float ddot = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                dT[a] *= g.k[a] * w;  // d(index coord)/dx
                A[a] += dT[a];
                Bv[a] = fmaf(al, dT[a], Bv[a]);
                ddot = fmaf(dT[a], q.d[a], ddot);
            }
            Cd += ddot;
            Cu = fmaf(u, ddot, Cu);
            if (WANT_VOL) scatter_trilinear(D, gx, gy, gz, k * w, add);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        r.gs[a] = k * (A[a] - Bv[a]);  // sum (1 - alpha) dT
        r.gt[a] = k * Bv[a];           // sum alpha dT
    }
    const float ws = gl * sumT / (float)(P - 1);  // through step = (amax - amin)/(P-1)
    r.g_amin = k * (Cd - Cu) - ws;
    r.g_amax = k * Cu + ws;
    r.sumT = sumT;
    return r;
}

}  // namespace ddrr
