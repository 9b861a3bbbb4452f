// segments_core.h -- the per-segment terms the reference holds just before `reduce`
// (diffdrr/renderers.py:71: img * intersection_length, shape (B, N, M - 1) with
// M = Dx + Dy + Dz + 3 sorted plane crossings), for a CALLABLE `reducefn`
// (renderers.py:175-183; notebooks/tutorials/introduction.ipynb:506-529), which needs the
// materialised tensor.  One ray walks ALL plane crossings in alpha order (three cursors, the
// crossing alphas evaluated from their integer plane indices like everywhere else); a segment
// between two consecutive crossings lies in the voxel given by how many planes of each axis
// have been crossed, and contributes V * dalpha if that voxel exists, 0 otherwise -- the
// reference's nearest lookup at the segment midpoint with zero padding.  Equal alphas are
// crossed one at a time (zero-length segments), as in the sorted list.
#pragma once

#include "ddrr_common.h"
#include "siddon_core.h"

namespace ddrr {

// visit(k, v_index (>= 0: flat voxel index, -1: outside), a_cur, a_next, axis_next):
// segment k runs from a_cur to a_next; `axis_next` is the axis of the crossing that closes it.
// visit_first(axis, alpha): the very first crossing (opens segment 0).
template <class First, class Visit>
DDRR_HD void siddon_all_crossings(const Dims D, const float s[3], const float t[3], float shift,
                                  float eps, First visit_first, Visit visit) {
    const int Dn[3] = {D.x, D.y, D.z};
    float inv[3], dd[3];
    int idx[3], step[3], left[3], cell[3];
    float an[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float d = (t[a] - s[a]) + eps;  // renderers.py:104-106
        inv[a] = 1.0f / d;
        dd[a] = d;
        const bool up = inv[a] >= 0.f;  // planes are crossed in increasing / decreasing index
        idx[a] = up ? 0 : Dn[a];
        step[a] = up ? 1 : -1;
        left[a] = Dn[a] + 1;
        cell[a] = up ? -1 : Dn[a];  // the cell the ray is in before crossing any plane
        an[a] = div_refined(((float)idx[a] - shift) - s[a], dd[a], inv[a]);  // renderers.py:97-106
    }
    const int M = D.x + D.y + D.z + 3;
    float a_cur = 0.f;
    for (int k = -1; k < M - 1; ++k) {
        // next crossing: smallest alpha among the axes that still have planes (ties: x, y, z)
        int ax = -1;
        float a_next = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (left[a] > 0 && (ax < 0 || an[a] < a_next)) {
                ax = a;
                a_next = an[a];
            }
        if (k < 0) {
            visit_first(ax, a_next);
        } else {
            const bool in = cell[0] >= 0 && cell[0] < D.x && cell[1] >= 0 && cell[1] < D.y &&
                            cell[2] >= 0 && cell[2] < D.z;
            visit(k, in ? (cell[0] * D.y + cell[1]) * D.z + cell[2] : -1, a_cur, a_next, ax);
        }
        // cross it
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (a == ax) {
                cell[a] = step[a] > 0 ? idx[a] : idx[a] - 1;
                idx[a] += step[a];
                --left[a];
                an[a] = div_refined(((float)idx[a] - shift) - s[a], dd[a], inv[a]);
            }
        a_cur = a_next;
    }
}

// terms[k] = L * V * dalpha (renderers.py:70-71); `terms` has M - 1 entries.
DDRR_HD void siddon_segments_ray(const float *__restrict__ vol, const Dims D, const float s[3],
                                 const float t[3], float shift, float eps, float L,
                                 float *__restrict__ terms, long stride) {
    siddon_all_crossings(
        D, s, t, shift, eps, [](int, float) {},
        [&](int k, int vi, float a0, float a1, int) {
            terms[k * stride] = vi >= 0 ? (L * vol[vi]) * (a1 - a0) : 0.f;
        });
}

// Backward for grad_terms g[k]: with w_k = g_k V_k, the loss gradient w.r.t. crossing c is
// L (w_{c-1} - w_c) (w = 0 before the first and after the last segment), and d alpha_c / d s_a =
// (alpha_c - 1) / d_a, d alpha_c / d t_a = -alpha_c / d_a on the crossing's own axis.
// g_img = sum_k g_k V_k dalpha_k; g_volume[voxel_k] += g_k L dalpha_k (`add`).
template <bool WANT_VOL, class Add>
DDRR_HD void siddon_segments_backward_ray(const float *__restrict__ vol, const Dims D,
                                          const float s[3], const float t[3], float shift,
                                          float eps, float L, const float *__restrict__ g,
                                          long stride, float gs[3], float gt[3], float &g_img,
                                          Add add) {
    float S0[3] = {0.f, 0.f, 0.f}, S1[3] = {0.f, 0.f, 0.f};
    float I = 0.f, w_prev = 0.f, a_open = 0.f;
    int ax_open = 0;
    siddon_all_crossings(
        D, s, t, shift, eps,
        [&](int ax, float a) {
            ax_open = ax;
            a_open = a;
        },
        [&](int k, int vi, float a0, float a1, int ax_next) {
            const float gk = g[k * stride];
            const float w = vi >= 0 ? gk * vol[vi] : 0.f;
            I = fmaf(w, a1 - a0, I);
            if (WANT_VOL && vi >= 0) add((unsigned)vi, gk * L * (a1 - a0));
            // the crossing that opened this segment (a0 == a_open, axis ax_open)
            const float dw = w_prev - w;
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (a == ax_open) {
                    S0[a] += dw;
                    S1[a] = fmaf(dw, a_open, S1[a]);
                }
            w_prev = w;
            ax_open = ax_next;
            a_open = a1;
        });
    // the last crossing closes the last segment (nothing after it)
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (a == ax_open) {
            S0[a] += w_prev;
            S1[a] = fmaf(w_prev, a_open, S1[a]);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = 1.0f / ((t[a] - s[a]) + eps);
        gs[a] = L * (S1[a] - S0[a]) * inv;
        gt[a] = -L * S1[a] * inv;
    }
    g_img = I;
}

}  // namespace ddrr
