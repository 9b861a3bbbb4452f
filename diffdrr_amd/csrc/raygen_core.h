// raygen_core.h -- fused ray generation and its adjoint for the DRR case.
//
// What it replaces: the tensor programs between a pose and the renderer call,
//   diffdrr/detector.py:151-153  pose = reorient.compose(extrinsic); source = pose(0),
//                                target = pose(calibrated detector points)
//   diffdrr/drr.py:201           img = ||target - source||            (world mm)
//   diffdrr/drr.py:204-205       source, target <- affine_inverse(.)  (voxel coordinates)
// and, in the backward pass, torch autograd of those einsums / norm plus the reduction of
// the renderer's per-ray endpoint gradients to one 3x4 matrix gradient per pose.
//
// The operation order of the reference is kept (world-space target first, rounded to
// fp32, then the volume's inverse affine), so the rays equal the reference's up to the
// summation order inside a 3-term dot product.
#pragma once

#include "ddrr_common.h"

namespace ddrr {

// y = M[:, :3] x + M[:, 3] for a row-major 3x4 matrix
DDRR_HD void apply34(const float *M, const float x[3], float y[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
        y[a] = fmaf(M[4 * a + 2], x[2], fmaf(M[4 * a + 1], x[1], M[4 * a] * x[0])) + M[4 * a + 3];
}

struct RayGenOut {
    float tv[3];  // target, voxel coordinates
    float L;      // world-space ray length
};

// Mw: (3,4) world pose of the C-arm (extrinsic o reorient); Ainv: (3,4) world -> voxel;
// P: calibrated detector point of the pixel (detector frame).
DDRR_HD RayGenOut raygen_ray(const float *Mw, const float *Ainv, const float P[3]) {
    RayGenOut o;
    float tw[3];
    apply34(Mw, P, tw);
    const float dx = tw[0] - Mw[3], dy = tw[1] - Mw[7], dz = tw[2] - Mw[11];
    o.L = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    apply34(Ainv, tw, o.tv);
    return o;
}

// Contribution of one ray to dLoss/dMw (12 floats, row-major 3x4).
//   g_tv, g_sv: gradient w.r.t. the ray's voxel-space target / (its share of the) source
//   g_L: gradient w.r.t. the ray length `img`
DDRR_HD void raygen_ray_adjoint(const float *Mw, const float *Ainv, const float P[3],
                                const float g_tv[3], const float g_sv[3], float g_L, float L,
                                float acc[12]) {
    float g_tw[3], g_sw[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // transpose of Ainv's 3x3 block
        g_tw[a] = fmaf(Ainv[8 + a], g_tv[2], fmaf(Ainv[4 + a], g_tv[1], Ainv[a] * g_tv[0]));
        g_sw[a] = fmaf(Ainv[8 + a], g_sv[2], fmaf(Ainv[4 + a], g_sv[1], Ainv[a] * g_sv[0]));
    }
    // img = ||tw - sw||: d img / d tw = (tw - sw) / img = -d img / d sw
    const float k = L > 0.f ? g_L / L : 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float u = fmaf(Mw[4 * a + 2], P[2], fmaf(Mw[4 * a + 1], P[1], Mw[4 * a] * P[0]));
        g_tw[a] = fmaf(k, u, g_tw[a]);
        g_sw[a] = fmaf(-k, u, g_sw[a]);
        acc[4 * a + 0] = fmaf(g_tw[a], P[0], acc[4 * a + 0]);
        acc[4 * a + 1] = fmaf(g_tw[a], P[1], acc[4 * a + 1]);
        acc[4 * a + 2] = fmaf(g_tw[a], P[2], acc[4 * a + 2]);
        acc[4 * a + 3] += g_tw[a] + g_sw[a];  // tw and sw both carry the translation column
    }
}

// ------------------------------------------------------------ Euler pose -> Mw
// World pose of the C-arm from Euler angles + translation, in one step:
//   R  = E(a0, th0) E(a1, th1) E(a2, th2)         pose.py:444-470 (euler_angles_to_matrix)
//   M  = [R | R xyz]                               pose.py:155-157, 108-114 (convert)
//   Mw = M @ reorient                              detector.py:151 (reorient.compose(pose))
// and its adjoint.  axes[k] in {0: X, 1: Y, 2: Z}; Ro (3,4): top rows of the 4x4 reorient.

// elementary rotation about `axis` by th, or its derivative w.r.t. th (row-major 3x3)
DDRR_HD void elem_rot(int axis, float th, bool deriv, float E[9]) {
    // Rotation about `axis` (0 X, 1 Y, 2 Z) or its derivative w.r.t. the angle.  With
    // p = axis+1, q = axis+2 (mod 3): E[axis][axis] = 1, E[p][p] = E[q][q] = c, E[p][q] = -s,
    // E[q][p] = s.  Written as selects on compile-time (i, j) so that E stays in registers.
    const float c = cosf(th), s = sinf(th);
    const float cc = deriv ? -s : c, ss = deriv ? c : s, one = deriv ? 0.f : 1.f;
    const int p = axis == 2 ? 0 : axis + 1, q = axis == 0 ? 2 : axis - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = 0.f;
            if (i == j) v = (i == axis) ? one : cc;
            else if (i == p && j == q) v = -ss;
            else if (i == q && j == p) v = ss;
            E[3 * i + j] = v;
        }
    }
}

DDRR_HD void mat3_mul(const float A[9], const float B[9], float C[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = fmaf(A[3 * i + 2], B[6 + j], fmaf(A[3 * i + 1], B[3 + j], A[3 * i] * B[j]));
}

DDRR_HD void pose_euler_forward(const float th[3], const float xyz[3], const int axes[3],
                                const float *Ro, float Mw[12]) {
    float E0[9], E1[9], E2[9], T[9], R[9];
    elem_rot(axes[0], th[0], false, E0);
    elem_rot(axes[1], th[1], false, E1);
    elem_rot(axes[2], th[2], false, E2);
    mat3_mul(E0, E1, T);
    mat3_mul(T, E2, R);
    const float v[3] = {Ro[3] + xyz[0], Ro[7] + xyz[1], Ro[11] + xyz[2]};  // to + xyz
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Mw[4 * i + j] = fmaf(R[3 * i + 2], Ro[8 + j], fmaf(R[3 * i + 1], Ro[4 + j], R[3 * i] * Ro[j]));
        Mw[4 * i + 3] = fmaf(R[3 * i + 2], v[2], fmaf(R[3 * i + 1], v[1], R[3 * i] * v[0]));
    }
}

DDRR_HD void pose_euler_backward(const float th[3], const float xyz[3], const int axes[3],
                                 const float *Ro, const float gMw[12], float g_th[3],
                                 float g_xyz[3]) {
    float E0[9], E1[9], E2[9], D0[9], D1[9], D2[9], T[9], R[9];
    elem_rot(axes[0], th[0], false, E0);
    elem_rot(axes[1], th[1], false, E1);
    elem_rot(axes[2], th[2], false, E2);
    elem_rot(axes[0], th[0], true, D0);
    elem_rot(axes[1], th[1], true, D1);
    elem_rot(axes[2], th[2], true, D2);
    mat3_mul(E0, E1, T);
    mat3_mul(T, E2, R);
    const float v[3] = {Ro[3] + xyz[0], Ro[7] + xyz[1], Ro[11] + xyz[2]};
    // dL/dR = gMw[:, :3] Ro3^T + gMw[:, 3] (x) (to + xyz);  dL/dxyz = R^T gMw[:, 3]
    float GR[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            GR[3 * i + j] = fmaf(gMw[4 * i + 3], v[j],
                                 fmaf(gMw[4 * i + 2], Ro[4 * j + 2],
                                      fmaf(gMw[4 * i + 1], Ro[4 * j + 1], gMw[4 * i] * Ro[4 * j])));
#pragma unroll
    for (int j = 0; j < 3; ++j)
        g_xyz[j] = fmaf(R[6 + j], gMw[11], fmaf(R[3 + j], gMw[7], R[j] * gMw[3]));
    // dL/dth_k = <dL/dR, dR/dth_k>
    float A[9], Bm[9];
    auto dot9 = [&](const float X[9]) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc = fmaf(GR[k], X[k], acc);
        return acc;
    };
    mat3_mul(D0, E1, A);
    mat3_mul(A, E2, Bm);
    g_th[0] = dot9(Bm);
    mat3_mul(E0, D1, A);
    mat3_mul(A, E2, Bm);
    g_th[1] = dot9(Bm);
    mat3_mul(T, D2, Bm);
    g_th[2] = dot9(Bm);
}

}  // namespace ddrr
