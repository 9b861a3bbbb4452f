// f64_core.h -- the two renderers in double precision, per ray.
//
// The reference runs in whatever dtype the module holds: `DRR(...).to(torch.float64)` renders
// and differentiates in fp64 (diffdrr/drr.py:71-75 register the affines as module buffers,
// renderers.py computes in the dtype of its inputs).  Those users want accuracy, not speed: this
// is a compact per-ray restatement in double of
//   Siddon.forward    renderers.py:34-76, 94-113, 143-169 (mode="nearest", align_corners=False,
//                     reducefn sum | max) and its autograd (ray endpoints, img, volume; sum),
//   Trilinear.forward renderers.py:205-241 (mode="bilinear", align_corners=False, sum) and its
//                     autograd (ray endpoints, img, alphamin / alphamax, volume),
// with the same structure as the fp32 kernels (3-way merge of the plane crossings from the
// integer plane index, backward record I / S0_a / S1_a) but none of
// their machinery.  Compiled for the device by f64_rays.hip and for the host by tests/emu.
#pragma once

#include "ddrr_common.h"

namespace ddrr64 {

using ddrr::Dims;

constexpr int kAux = 8;  // I, S0_xyz, S1_xyz, unused -- the layout of the fp32 interleaved record

DDRR_HD double min3d(double a, double b, double c) { return fmin(fmin(a, b), c); }

// The sorted crossings of one ray with the volume's planes, visited as segments.
// on_segment(voxel index, a_cur, a_next, axis of the crossing that opened it); returns the axis
// of the exit crossing through `exit_axis` / its alpha through `a_exit`.  False: no chord.
template <class OnSegment>
DDRR_HD bool siddon_walk(const Dims D, const double s[3], const double t[3], double shift,
                         double eps, OnSegment on_segment, int &exit_axis, double &a_exit) {
    const int Dn[3] = {D.x, D.y, D.z};
    double d[3], lo[3], entry = -INFINITY, exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;  // renderers.py:104-106
        const double a0 = ((0.0 - shift) - s[a]) / d[a];
        const double aD = (((double)Dn[a] - shift) - s[a]) / d[a];
        lo[a] = fmin(a0, aD);
        entry = fmax(entry, lo[a]);
        exit = fmin(exit, fmax(a0, aD));
    }
    if (!(entry < exit)) return false;
    auto alpha = [&](int a, double k) { return ((k - shift) - s[a]) / d[a]; };
    double kf[3], dir[3], an[3];
    long stride[3] = {(long)D.y * D.z, (long)D.z, 1}, idx = 0, step[3];
    // the axis of the entry crossing: exclusive, x before y before z (as the fp32 record)
    int open_axis = lo[0] == entry ? 0 : (lo[1] == entry ? 1 : 2);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const bool pos = d[a] > 0.0;
        double u = floor(fma(entry, d[a], s[a] + shift));
        u = fmin(fmax(u, 0.0), (double)(Dn[a] - 1));
        if (lo[a] == entry) {
            u = pos ? 0.0 : (double)(Dn[a] - 1);  // entering axis: the face cell
        } else {
            // the cell must agree with the ORDER of the alphas, not only with the position
            if (alpha(a, u + (pos ? 1.0 : 0.0)) < entry) u += pos ? 1.0 : -1.0;
            else if (alpha(a, u + (pos ? 0.0 : 1.0)) > entry) u -= pos ? 1.0 : -1.0;
            u = fmin(fmax(u, 0.0), (double)(Dn[a] - 1));
        }
        idx += (long)u * stride[a];
        kf[a] = u + (pos ? 1.0 : 0.0);
        dir[a] = pos ? 1.0 : -1.0;
        step[a] = pos ? stride[a] : -stride[a];
        an[a] = alpha(a, kf[a]);
    }
    double a_cur = entry;
    const int cap = D.x + D.y + D.z + 3;
    for (int it = 0; it < cap; ++it) {
        const double a_next = min3d(an[0], an[1], an[2]);
        on_segment(idx, a_cur, a_next, open_axis);
        const bool cx = an[0] <= a_next, cy = an[1] <= a_next, cz = an[2] <= a_next;
        open_axis = cx ? 0 : (cy ? 1 : 2);
        a_cur = a_next;
        if (!(a_next < exit)) break;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const bool c = a == 0 ? cx : (a == 1 ? cy : cz);
            if (c) {
                kf[a] += dir[a];
                idx += step[a];
                an[a] = alpha(a, kf[a]);
            }
        }
    }
    exit_axis = open_axis;
    a_exit = a_cur;
    return true;
}

// Forward of one ray: returns sum V dalpha (or max V dalpha); rec (kAux doubles, may be null):
// the backward record of the sum.
DDRR_HD double siddon_forward_ray(const double *__restrict__ vol, const Dims D, const double s[3],
                                  const double t[3], double shift, double eps, bool take_max,
                                  double *rec) {
    double I = 0.0, best = 0.0, S0[3] = {0, 0, 0}, S1[3] = {0, 0, 0}, prev = 0.0;
    int exit_axis = 0;
    double a_exit = 0.0;
    const bool hit = siddon_walk(
        D, s, t, shift, eps,
        [&](long idx, double a_cur, double a_next, int open_axis) {
            const double v = vol[idx], term = v * (a_next - a_cur);
            I += term;
            best = term > best ? term : best;
            S0[open_axis] += prev - v;
            S1[open_axis] += (prev - v) * a_cur;
            prev = v;
        },
        exit_axis, a_exit);
    if (hit) {
        S0[exit_axis] += prev;  // the exit crossing: last voxel | 0
        S1[exit_axis] += prev * a_exit;
    }
    if (rec) {
        rec[0] = I;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rec[1 + a] = S0[a];
            rec[4 + a] = S1[a];
        }
        rec[7] = 0.0;
    }
    return take_max ? best : I;
}

// d out / d (s, t) from the record, gl = grad_out * img:
//   d/ds_a = gl (S1_a - S0_a) / d_a,   d/dt_a = -gl S1_a / d_a   (the img path is d/d img = g I)
DDRR_HD void siddon_backward_ray(const double *rec, const double s[3], const double t[3],
                                 double eps, double gl, double gs[3], double gt[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double d = (t[a] - s[a]) + eps;
        gs[a] = gl * (rec[4 + a] - rec[1 + a]) / d;
        gt[a] = -gl * rec[4 + a] / d;
    }
}

// d out / d V[voxel of segment k] = gl dalpha_k
template <class Add>
DDRR_HD void siddon_scatter_ray(const Dims D, const double s[3], const double t[3], double shift,
                                double eps, double gl, Add add) {
    int ea;
    double ae;
    siddon_walk(
        D, s, t, shift, eps,
        [&](long idx, double a_cur, double a_next, int) { add(idx, gl * (a_next - a_cur)); }, ea,
        ae);
}

// torch.linspace(0, 1, P)[m]: the reference builds the table in the DEFAULT dtype (fp32) and
// only then casts it `.to(volume)` (renderers.py:224), so the sample positions of an fp64
// module are fp32-rounded fractions of the range: restated as such (aten's scalar formula,
// symmetric about the middle).
DDRR_HD double lin01(int m, int P) {
    const float step = 1.0f / (float)(P - 1);
    return (double)(m < P / 2 ? (float)m * step : 1.0f - (float)(P - 1 - m) * step);
}

// Zero-padded trilinear sample at index coordinate g (align_corners=False: g = x + shift - 1/2)
// and its gradient w.r.t. g; `corner(idx, weight)` is called for the in-bounds corners.
template <class Corner>
DDRR_HD double trilinear_sample(const double *__restrict__ vol, const Dims D, const double g[3],
                                double grad[3], Corner corner) {
    const int Dn[3] = {D.x, D.y, D.z};
    double f[3], w[3];
    long i0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        f[a] = floor(g[a]);
        w[a] = g[a] - f[a];
        i0[a] = (long)f[a];
    }
    double T = 0.0;
    grad[0] = grad[1] = grad[2] = 0.0;
    // (a sample further than one cell outside the volume touches nothing)
    if (!(g[0] > -1.0 && g[0] < Dn[0] && g[1] > -1.0 && g[1] < Dn[1] && g[2] > -1.0 && g[2] < Dn[2]))
        return 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
        const long x = i0[0] + cx, y = i0[1] + cy, z = i0[2] + cz;
        if (x < 0 || x >= Dn[0] || y < 0 || y >= Dn[1] || z < 0 || z >= Dn[2]) continue;
        const double wx = cx ? w[0] : 1.0 - w[0], wy = cy ? w[1] : 1.0 - w[1];
        const double wz = cz ? w[2] : 1.0 - w[2];
        const long idx = (x * D.y + y) * D.z + z;
        const double v = vol ? vol[idx] : 0.0;
        T += v * wx * wy * wz;
        grad[0] += v * (cx ? 1.0 : -1.0) * wy * wz;
        grad[1] += v * wx * (cy ? 1.0 : -1.0) * wz;
        grad[2] += v * wx * wy * (cz ? 1.0 : -1.0);
        corner(idx, wx * wy * wz);
    }
    return T;
}

struct NoCorner {
    DDRR_HD void operator()(long, double) const {}
};

// out = img * step * sum_m T(x(alpha_m))  (renderers.py:224-236); returns sum_m T
DDRR_HD double trilinear_forward_ray(const double *__restrict__ vol, const Dims D,
                                     const double s[3], const double t[3], double shift,
                                     double eps, int P, double amin, double amax) {
    double sumT = 0.0;
    for (int m = 0; m < P; ++m) {
        const double al = amin + lin01(m, P) * (amax - amin);
        double g[3], grad[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] = (s[a] + al * ((t[a] - s[a]) + eps)) + shift - 0.5;
        sumT += trilinear_sample(vol, D, g, grad, NoCorner{});
    }
    return sumT;
}

// Backward of one ray for weight gl = grad_out * img:
//   d/ds = gl step sum (1 - alpha_m) dT_m,  d/dt = gl step sum alpha_m dT_m,
//   d/d amin = gl [-sum T_m / (P-1) + step sum (1 - u_m) dT_m . d],  d/d amax = gl [+... u_m ...],
//   d/dV[corner c of sample m] += gl step w_c;   returns sum_m T (for d/d img = g step sum T).
template <class Add>
DDRR_HD double trilinear_backward_ray(const double *__restrict__ vol, const Dims D,
                                      const double s[3], const double t[3], double shift,
                                      double eps, int P, double amin, double amax, double gl,
                                      double gs[3], double gt[3], double ga[2], bool want_volume,
                                      Add add) {
    const double span = amax - amin, step = span / (double)(P - 1);
    double sumT = 0.0, d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;
        gs[a] = gt[a] = 0.0;
    }
    ga[0] = ga[1] = 0.0;
    for (int m = 0; m < P; ++m) {
        const double u = lin01(m, P), al = amin + u * span;
        double g[3], grad[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] = (s[a] + al * d[a]) + shift - 0.5;
        const double wv = gl * step;
        const double T = trilinear_sample(vol, D, g, grad, [&](long idx, double w) {
            if (want_volume) add(idx, wv * w);
        });
        sumT += T;
        const double gd = grad[0] * d[0] + grad[1] * d[1] + grad[2] * d[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            gs[a] += (1.0 - al) * grad[a];
            gt[a] += al * grad[a];
        }
        ga[0] += (1.0 - u) * gd;
        ga[1] += u * gd;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gs[a] *= gl * step;
        gt[a] *= gl * step;
    }
    ga[0] = gl * (-sumT / (double)(P - 1) + step * ga[0]);
    ga[1] = gl * (sumT / (double)(P - 1) + step * ga[1]);
    return sumT;
}

}  // namespace ddrr64
