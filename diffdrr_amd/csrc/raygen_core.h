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

}  // namespace ddrr
