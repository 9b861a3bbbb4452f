// general_core.h -- the MATERIALISING general path of both renderers, in float or double.
//
// The fused kernels cover what the reference's defaults and tutorials use.  The reference,
// being a chain of tensor ops, also renders every other combination of its keyword arguments
// (diffdrr/renderers.py):
//   * Siddon with a mask AND a midpoint lookup (mode="bilinear" or align_corners=True;
//     :57-60, :77-89 -- the label map is looked up with the same mode, then `.long()`),
//   * a callable reducefn over the per-segment tensor with those lookups (:70-71, :175-183),
//   * gradients of reducefn="max" / stop_gradients_through_grid_sample with those lookups
//     (:63-65),
//   * a module moved `.to(torch.float64)` (drr.py:71-75) with a mask, a callable, a midpoint
//     lookup, the marcher with mode="nearest" / reducefn="max" / align_corners=True (:205-254).
// All of these go through the tensors the reference materialises just before its reduction:
// the (B, N, M - 1) per-segment terms `img * intersection_length` (:70-71) and the (B, N, P)
// per-sample terms `img * step_size` (:235-236).  The functions below write exactly those
// tensors (or, `raw`, the looked-up values themselves: the label lookups) and apply their
// autograd for an incoming gradient of the same shape; sum / max / a callable / the channel
// scatter are then ordinary tensor ops on the result.  Rare combinations: memory and time are
// the reference's, not the fused kernels'.
//
// One lane per ray, templated on the scalar type.  Compiled for the device by
// general_rays.hip and for the host by tests/emu.
#pragma once

#include "ddrr_common.h"
#include "siddon_core.h"  // div_refined (the float plane-crossing quotient)

namespace ddrr_gen {

using ddrr::Dims;

DDRR_HD float g_rint(float x) { return rintf(x); }  // half-to-even == aten's nearbyint
DDRR_HD double g_rint(double x) { return rint(x); }
DDRR_HD float g_floor(float x) { return floorf(x); }
DDRR_HD double g_floor(double x) { return floor(x); }

// alpha of a plane: the reference's quotient (k - shift - s_a) / (t_a - s_a + eps)
// (renderers.py:97-106); float through the refined product every fp32 walk uses
DDRR_HD float plane_alpha(float num, float d, float inv) { return ddrr::div_refined(num, d, inv); }
DDRR_HD double plane_alpha(double num, double d, double) { return num / d; }

// Index coordinate of the point s + alpha d: the reference's own chain of tensor operations,
// every one rounded on its own (the nearest lookups and the truncated label lookups are
// discontinuous: which side of a voxel boundary a sample falls on is decided by exactly this
// arithmetic) -- _get_xyzs (renderers.py:148-152): x = s + alpha d, 2 (x + shift) / D - 1;
// then aten's grid_sampler un-normalise:
//   align_corners=False: ((g + 1) D - 1) / 2      (= x + shift - 1/2)
//   align_corners=True : (g + 1) / 2 (D - 1)      (= (x + shift)(D - 1) / D)
template <class T>
DDRR_HD void index_coord(const Dims D, const T s[3], const T d[3], T alpha, T shift,
                         bool align_corners, T g[3]) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const T Dn[3] = {(T)D.x, (T)D.y, (T)D.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const T prod = alpha * d[a];
        const T x = s[a] + prod;
        const T twice = (T)2 * (x + shift);
        const T nrm = twice / Dn[a] - (T)1;
        const T up = nrm + (T)1;
        g[a] = align_corners ? (up / (T)2) * (Dn[a] - (T)1) : (up * Dn[a] - (T)1) / (T)2;
    }
}

// d (index coordinate) / d x per axis, for the gradients through a trilinear lookup
template <class T>
DDRR_HD T index_scale(const Dims D, int a, bool align_corners) {
    const int Dn = a == 0 ? D.x : (a == 1 ? D.y : D.z);
    return align_corners ? (T)(Dn - 1) / (T)Dn : (T)1;
}

// alphas = linspace * (alphamax - alphamin) + alphamin (renderers.py:225), product and sum
// rounded separately
template <class T>
DDRR_HD T march_alpha(T lin, T span, T amin) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const T prod = lin * span;
    return prod + amin;
}

// grid_sample "nearest", zero padding: the value and (idx >= 0) where it came from
template <class T>
DDRR_HD T lookup_nearest(const T *__restrict__ vol, const Dims D, const T g[3], long &idx) {
    const T rx = g_rint(g[0]), ry = g_rint(g[1]), rz = g_rint(g[2]);
    const bool in = rx >= (T)0 && rx < (T)D.x && ry >= (T)0 && ry < (T)D.y && rz >= (T)0 &&
                    rz < (T)D.z;
    idx = in ? ((long)rx * D.y + (long)ry) * D.z + (long)rz : -1;
    return in ? vol[idx] : (T)0;
}

// grid_sample "bilinear" (trilinear in 3-D), zero padding per corner: the value, its gradient
// w.r.t. the index coordinate, and corner(idx, weight) for every corner inside the volume
template <class T, class Corner>
DDRR_HD T lookup_trilinear(const T *__restrict__ vol, const Dims D, const T g[3], T grad[3],
                           Corner corner) {
    const int Dn[3] = {D.x, D.y, D.z};
    grad[0] = grad[1] = grad[2] = (T)0;
    // (a sample a whole cell outside the volume touches nothing; also keeps the casts in range)
    if (!(g[0] > (T)-1 && g[0] < (T)Dn[0] && g[1] > (T)-1 && g[1] < (T)Dn[1] && g[2] > (T)-1 &&
          g[2] < (T)Dn[2]))
        return (T)0;
    T w[3];
    long i0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const T f = g_floor(g[a]);
        w[a] = g[a] - f;
        i0[a] = (long)f;
    }
    T val = (T)0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
        const long x = i0[0] + cx, y = i0[1] + cy, z = i0[2] + cz;
        if (x < 0 || x >= Dn[0] || y < 0 || y >= Dn[1] || z < 0 || z >= Dn[2]) continue;
        const T wx = cx ? w[0] : (T)1 - w[0], wy = cy ? w[1] : (T)1 - w[1];
        const T wz = cz ? w[2] : (T)1 - w[2];
        const long idx = (x * D.y + y) * D.z + z;
        const T v = vol[idx];
        val += v * (wx * wy * wz);
        grad[0] += v * ((cx ? (T)1 : (T)-1) * wy * wz);
        grad[1] += v * (wx * (cy ? (T)1 : (T)-1) * wz);
        grad[2] += v * (wx * wy * (cz ? (T)1 : (T)-1));
        corner(idx, wx * wy * wz);
    }
    return val;
}

struct NoCorner {
    template <class T>
    DDRR_HD void operator()(long, T) const {}
};

// ------------------------------------------------------------------ Siddon: all crossings
// The reference's sorted list of ALL Dx + Dy + Dz + 3 plane crossings (renderers.py:94-113),
// one at a time (equal alphas give zero-length segments, as in the sorted list; ties are
// taken x before y before z).  first(axis, alpha): the first crossing;
// visit(k, cell index or -1, a_cur, a_next, axis_next): segment k, closed by a crossing of
// `axis_next`; `cell` is the voxel the segment lies in by the count of planes crossed (the
// reference's nearest lookup at the midpoint for align_corners=False).
template <class T, class First, class Visit>
DDRR_HD void all_crossings(const Dims D, const T s[3], const T t[3], T shift, T eps, First first,
                           Visit visit) {
    const int Dn[3] = {D.x, D.y, D.z};
    T inv[3], dd[3], an[3];
    int idx[3], step[3], left[3], cell[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const T d = (t[a] - s[a]) + eps;  // renderers.py:104-106
        inv[a] = (T)1 / d;
        dd[a] = d;
        const bool up = inv[a] >= (T)0;
        idx[a] = up ? 0 : Dn[a];
        step[a] = up ? 1 : -1;
        left[a] = Dn[a] + 1;
        cell[a] = up ? -1 : Dn[a];
        an[a] = plane_alpha(((T)idx[a] - shift) - s[a], dd[a], inv[a]);
    }
    const int M = D.x + D.y + D.z + 3;
    T a_cur = (T)0;
    for (int k = -1; k < M - 1; ++k) {
        int ax = -1;
        T a_next = (T)INFINITY;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (left[a] > 0 && (ax < 0 || an[a] < a_next)) {
                ax = a;
                a_next = an[a];
            }
        if (k < 0) {
            first(ax, a_next);
        } else {
            const bool in = cell[0] >= 0 && cell[0] < D.x && cell[1] >= 0 && cell[1] < D.y &&
                            cell[2] >= 0 && cell[2] < D.z;
            visit(k, in ? ((long)cell[0] * D.y + cell[1]) * D.z + cell[2] : -1L, a_cur, a_next, ax);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (a == ax) {
                cell[a] = step[a] > 0 ? idx[a] : idx[a] - 1;
                idx[a] += step[a];
                --left[a];
                an[a] = plane_alpha(((T)idx[a] - shift) - s[a], dd[a], inv[a]);
            }
        a_cur = a_next;
    }
}

// The looked-up value of one segment (renderers.py:57-66, 156-169), its gradient w.r.t. the
// POINT x (voxel units; zero unless trilinear), and where it was read.
template <class T, int LOOKUP, class Corner>
DDRR_HD T segment_value(const T *__restrict__ vol, const Dims D, T shift, bool align_corners,
                        const T s[3], const T d[3], long cell, T a0, T a1, T &mid, T G[3],
                        long &idx, Corner corner) {
    G[0] = G[1] = G[2] = (T)0;
    mid = (T)0.5 * (a0 + a1);  // renderers.py:57
    if (LOOKUP == ddrr::LOOKUP_STEP) {
        idx = cell;
        return cell >= 0 ? vol[cell] : (T)0;
    }
    T g[3];
    index_coord(D, s, d, mid, shift, align_corners, g);
    if (LOOKUP == ddrr::LOOKUP_MID_NEAREST) return lookup_nearest(vol, D, g, idx);
    idx = -1;
    T dg[3];
    const T v = lookup_trilinear(vol, D, g, dg, corner);
#pragma unroll
    for (int a = 0; a < 3; ++a) G[a] = dg[a] * index_scale<T>(D, a, align_corners);
    return v;
}

// terms[k] = L * value_k * (alpha_{k+1} - alpha_k)   (renderers.py:66-71), or, `raw`, value_k
// alone (the label lookup, :82-84); M - 1 entries, `stride` apart.
template <class T, int LOOKUP>
DDRR_HD void siddon_segments_ray(const T *__restrict__ vol, const Dims D, const T s[3],
                                 const T t[3], T shift, T eps, bool align_corners, T L, bool raw,
                                 T *__restrict__ terms, long stride) {
    T d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = (t[a] - s[a]) + eps;
    all_crossings(
        D, s, t, shift, eps, [](int, T) {},
        [&](int k, long cell, T a0, T a1, int) {
            T mid, G[3];
            long idx;
            const T v = segment_value<T, LOOKUP>(vol, D, shift, align_corners, s, d, cell, a0, a1, mid, G, idx,
                                                 NoCorner{});
            terms[k * stride] = raw ? v : (L * v) * (a1 - a0);
        });
}

// Autograd of the terms for an incoming gradient g[k].  With seg_k = alpha_{k+1} - alpha_k,
// m_k the midpoint, T_k the looked-up value, G_k = dT/dx there (trilinear lookups only, and
// only when gradients flow THROUGH the lookup: not with stop_gradients_through_grid_sample,
// renderers.py:63-65):
//   loss = L sum_k g_k T_k seg_k
//   d loss / d alpha_c = L [(w_{c-1} - w_c) + (u_{c-1} + u_c) / 2],  w_k = g_k T_k,
//                                                                    u_k = g_k seg_k (G_k . d)
//   d alpha_c / d s_a = (alpha_c - 1) / d_a,  d alpha_c / d t_a = -alpha_c / d_a  (own axis)
//   direct: d loss / d s = L sum_k g_k seg_k (1 - m_k) G_k,   d/dt: ... m_k G_k
//   g_img = sum_k g_k T_k seg_k;   g_volume[...] += g_k L seg_k x (1 | the 8 corner weights)
template <class T, int LOOKUP, class Add>
DDRR_HD void siddon_segments_backward_ray(const T *__restrict__ vol, const Dims D, const T s[3],
                                          const T t[3], T shift, T eps, bool align_corners, T L,
                                          bool through, const T *__restrict__ g, long stride,
                                          T gs[3], T gt[3], T &g_img, bool want_volume, Add add) {
    T d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = (t[a] - s[a]) + eps;
    T S0[3] = {0, 0, 0}, S1[3] = {0, 0, 0}, Es[3] = {0, 0, 0}, Et[3] = {0, 0, 0};
    T I = 0, w_prev = 0, u_prev = 0, a_open = 0;
    int ax_open = 0;
    all_crossings(
        D, s, t, shift, eps,
        [&](int ax, T a) {
            ax_open = ax;
            a_open = a;
        },
        [&](int k, long cell, T a0, T a1, int ax_next) {
            const T gk = g[k * stride], seg = a1 - a0;
            T mid, G[3];
            long idx;
            const T wv = gk * L * seg;
            const T v = segment_value<T, LOOKUP>(vol, D, shift, align_corners, s, d, cell, a0, a1, mid, G, idx,
                                                 [&](long ci, T cw) {
                                                     if (want_volume && through) add(ci, wv * cw);
                                                 });
            if (want_volume && through && LOOKUP != ddrr::LOOKUP_MID_TRILINEAR && idx >= 0)
                add(idx, wv);
            const T w = gk * v;
            T u = 0;
            if (LOOKUP == ddrr::LOOKUP_MID_TRILINEAR && through) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    u += G[a] * d[a];
                    Es[a] += gk * seg * ((T)1 - mid) * G[a];
                    Et[a] += gk * seg * mid * G[a];
                }
                u *= gk * seg;
            }
            I += w * seg;
            const T coef = (w_prev - w) + (T)0.5 * (u_prev + u);
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (a == ax_open) {
                    S0[a] += coef;
                    S1[a] += coef * a_open;
                }
            w_prev = w;
            u_prev = u;
            ax_open = ax_next;
            a_open = a1;
        });
    // the last crossing closes the last segment
    const T coef = w_prev + (T)0.5 * u_prev;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (a == ax_open) {
            S0[a] += coef;
            S1[a] += coef * a_open;
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gs[a] = L * ((S1[a] - S0[a]) / d[a] + Es[a]);
        gt[a] = L * (-S1[a] / d[a] + Et[a]);
    }
    g_img = I;
}

// ------------------------------------------------------------------ the marcher
// torch.linspace(0, 1, P)[m] -- built in the DEFAULT dtype (fp32) and then cast `.to(volume)`
// (renderers.py:224): aten's symmetric fp32 formula, also for a float64 module.
DDRR_HD float lin01(int m, int P) {
    const float step = 1.0f / (float)(P - 1);
    return m < P / 2 ? (float)m * step : 1.0f - (float)(P - 1 - m) * step;
}

// samples[m] = L * step * T(x(alpha_m))   (renderers.py:224-236), or, `raw`, the looked-up
// value alone (the label lookup, :246-248); P entries, `stride` apart.
template <class T, bool NEAREST>
DDRR_HD void trilinear_samples_ray(const T *__restrict__ vol, const Dims D, const T s[3],
                                   const T t[3], T shift, T eps, bool align_corners, int P,
                                   T amin, T amax, T L, bool raw, T *__restrict__ samples,
                                   long stride) {
    const T span = amax - amin, step = span / (T)(P - 1);
    T d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = (t[a] - s[a]) + eps;
    for (int m = 0; m < P; ++m) {
        const T al = march_alpha((T)lin01(m, P), span, amin);
        T g[3], dg[3];
        index_coord(D, s, d, al, shift, align_corners, g);
        long idx;
        const T v = NEAREST ? lookup_nearest(vol, D, g, idx)
                            : lookup_trilinear(vol, D, g, dg, NoCorner{});
        samples[m * stride] = raw ? v : (L * v) * step;
    }
}

// Autograd of the samples for an incoming gradient g[m] (u_m = linspace, G_m = dT/dx):
//   d/ds = L step sum g_m (1 - alpha_m) G_m,   d/dt = L step sum g_m alpha_m G_m,
//   d/d amin = L [-sum g_m T_m / (P-1) + step sum g_m (1 - u_m) G_m . d],
//   d/d amax = L [+sum g_m T_m / (P-1) + step sum g_m u_m G_m . d],
//   g_img = step sum g_m T_m;   g_volume[corner] += g_m L step w_c
template <class T, bool NEAREST, class Add>
DDRR_HD void trilinear_samples_backward_ray(const T *__restrict__ vol, const Dims D, const T s[3],
                                            const T t[3], T shift, T eps, bool align_corners,
                                            int P, T amin, T amax, T L,
                                            const T *__restrict__ g, long stride, T gs[3],
                                            T gt[3], T ga[2], T &g_img, bool want_volume,
                                            Add add) {
    const T span = amax - amin, step = span / (T)(P - 1);
    T d[3], sumT = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;
        gs[a] = gt[a] = 0;
    }
    ga[0] = ga[1] = 0;
    for (int m = 0; m < P; ++m) {
        const T u = (T)lin01(m, P), al = march_alpha(u, span, amin), gm = g[m * stride];
        T gi[3], dg[3] = {0, 0, 0};
        index_coord(D, s, d, al, shift, align_corners, gi);
        const T wv = gm * L * step;
        T v;
        if (NEAREST) {
            long idx;
            v = lookup_nearest(vol, D, gi, idx);
            if (want_volume && idx >= 0) add(idx, wv);
        } else {
            v = lookup_trilinear(vol, D, gi, dg, [&](long ci, T cw) {
                if (want_volume) add(ci, wv * cw);
            });
        }
        sumT += gm * v;
        T gd = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const T G = dg[a] * index_scale<T>(D, a, align_corners);
            gd += G * d[a];
            gs[a] += gm * ((T)1 - al) * G;
            gt[a] += gm * al * G;
        }
        ga[0] += gm * ((T)1 - u) * gd;
        ga[1] += gm * u * gd;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gs[a] *= L * step;
        gt[a] *= L * step;
    }
    ga[0] = L * (-sumT / (T)(P - 1) + step * ga[0]);
    ga[1] = L * (sumT / (T)(P - 1) + step * ga[1]);
    g_img = step * sumT;
}

}  // namespace ddrr_gen
