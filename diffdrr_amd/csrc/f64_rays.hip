// f64_rays.hip -- the renderers in double precision (f64_core.h): one lane per ray, plain
// elementwise kernels.  For `DRR(...).to(torch.float64)` (reference drr.py:71-75): accuracy, not
// speed -- fp64 vector math runs at a fraction of the fp32 rate and nothing here is tuned.
#include "runtime.h"

#include "f64_core.h"

using namespace ddrr_rt;

namespace {

struct Ray64 {
    const double *vol;
    Dims D;
    const double *source;
    int src_n;
    const double *target;
    const double *img;
    long R;
    int N;
    double shift, eps;
};

__device__ __forceinline__ bool load64(const Ray64 &p, long r, double s[3], double t[3]) {
    if (r >= p.R) return false;
    const long b = r / p.N, n = r - b * p.N;
    const double *sp = p.source + (b * p.src_n + (p.src_n == 1 ? 0 : n)) * 3;
    const double *tp = p.target + r * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        s[a] = sp[a];
        t[a] = tp[a];
    }
    return true;
}

struct AtomicAdd64 {
    double *base;
    __device__ __forceinline__ void operator()(long idx, double v) const {
        unsafeAtomicAdd(base + idx, v);  // global_atomic_add_f64
    }
};

__global__ __launch_bounds__(kBlock) void siddon_fwd64_kernel(Ray64 p, int take_max,
                                                              double *__restrict__ out,
                                                              double *__restrict__ aux) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    double s[3], t[3];
    if (!load64(p, r, s, t)) return;
    double rec[ddrr64::kAux];
    const double v = ddrr64::siddon_forward_ray(p.vol, p.D, s, t, p.shift, p.eps, take_max != 0,
                                                aux ? rec : nullptr);
    out[r] = (p.img ? p.img[r] : 1.0) * v;
    if (aux)
#pragma unroll
        for (int k = 0; k < ddrr64::kAux; ++k) aux[r * ddrr64::kAux + k] = rec[k];
}

__global__ __launch_bounds__(kBlock) void siddon_bwd64_kernel(
    Ray64 p, const double *__restrict__ aux, const double *__restrict__ grad_out,
    double *__restrict__ g_source, double *__restrict__ g_target, double *__restrict__ g_img,
    double *__restrict__ g_volume) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    double s[3], t[3];
    if (!load64(p, r, s, t)) return;
    const double g = grad_out[r], L = p.img ? p.img[r] : 1.0;
    if (aux && (g_source || g_target || g_img)) {
        double gs[3], gt[3];
        ddrr64::siddon_backward_ray(aux + r * ddrr64::kAux, s, t, p.eps, g * L, gs, gt);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (g_source) g_source[r * 3 + a] = gs[a];
            if (g_target) g_target[r * 3 + a] = gt[a];
        }
        if (g_img) g_img[r] = g * aux[r * ddrr64::kAux];
    }
    if (g_volume && g * L != 0.0)
        ddrr64::siddon_scatter_ray(p.D, s, t, p.shift, p.eps, g * L, AtomicAdd64{g_volume});
}

__global__ __launch_bounds__(kBlock) void trilinear_fwd64_kernel(Ray64 p, int P,
                                                                 const double *__restrict__ amin,
                                                                 const double *__restrict__ amax,
                                                                 double *__restrict__ out) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    double s[3], t[3];
    if (!load64(p, r, s, t)) return;
    const double a0 = amin[0], a1 = amax[0];
    const double sumT = ddrr64::trilinear_forward_ray(p.vol, p.D, s, t, p.shift, p.eps, P, a0, a1);
    out[r] = (p.img ? p.img[r] : 1.0) * ((a1 - a0) / (double)(P - 1)) * sumT;  // renderers.py:235
}

__global__ __launch_bounds__(kBlock) void trilinear_bwd64_kernel(
    Ray64 p, int P, const double *__restrict__ amin, const double *__restrict__ amax,
    const double *__restrict__ grad_out, double *__restrict__ g_source,
    double *__restrict__ g_target, double *__restrict__ g_img, double *__restrict__ g_alpha,
    double *__restrict__ g_volume) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    double s[3], t[3];
    if (!load64(p, r, s, t)) return;
    const double a0 = amin[0], a1 = amax[0];
    const double g = grad_out[r], L = p.img ? p.img[r] : 1.0;
    double gs[3], gt[3], ga[2];
    const double sumT = ddrr64::trilinear_backward_ray(p.vol, p.D, s, t, p.shift, p.eps, P, a0, a1,
                                                       g * L, gs, gt, ga, g_volume != nullptr,
                                                       AtomicAdd64{g_volume});
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[r * 3 + a] = gs[a];
        if (g_target) g_target[r * 3 + a] = gt[a];
    }
    if (g_img) g_img[r] = g * ((a1 - a0) / (double)(P - 1)) * sumT;
    if (g_alpha) {
        g_alpha[r * 2] = ga[0];
        g_alpha[r * 2 + 1] = ga[1];
    }
}

int check64(const void *volume, int dx, int dy, int dz, const void *source, int src_n,
            const void *target, int B, int N) {
    if (!volume || !source || !target) return fail(-1, "null pointer");
    if (dx < 1 || dy < 1 || dz < 1) return fail(-1, "volume dims must be positive");
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (src_n != 1 && src_n != N) return fail(-1, "src_n must be 1 or N");
    return 0;
}

Ray64 make64(const double *volume, int dx, int dy, int dz, const double *source, int src_n,
             const double *target, const double *img, int B, int N, double shift, double eps) {
    Ray64 p;
    p.vol = volume;
    p.D = Dims{dx, dy, dz};
    p.source = source;
    p.src_n = src_n;
    p.target = target;
    p.img = img;
    p.R = (long)B * N;
    p.N = N;
    p.shift = shift;
    p.eps = eps;
    return p;
}

inline dim3 grid64(long R) { return dim3((unsigned)((R + kBlock - 1) / kBlock)); }

}  // namespace

extern "C" {

int ddrr_siddon_forward_f64(const double *volume, int dx, int dy, int dz, const double *source,
                            int src_n, const double *target, const double *img, int B, int N,
                            double voxel_shift, double eps, int reduce_mode, double *out,
                            double *aux, void *stream) {
    if (int rc = check64(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!out) return fail(-1, "null out pointer");
    if (reduce_mode != DDRR_REDUCE_SUM && reduce_mode != DDRR_REDUCE_MAX)
        return fail(-1, "unknown reduce_mode");
    if (aux && reduce_mode != DDRR_REDUCE_SUM) return fail(-1, "the fp64 record needs reduce sum");
    const long R = (long)B * N;
    if (R == 0) return 0;
    const Ray64 p = make64(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift, eps);
    hipLaunchKernelGGL(siddon_fwd64_kernel, grid64(R), dim3(kBlock), 0, (hipStream_t)stream, p,
                       reduce_mode == DDRR_REDUCE_MAX ? 1 : 0, out, aux);
    return finish("ddrr_siddon_forward_f64");
}

int ddrr_siddon_backward_f64(int dx, int dy, int dz, const double *source, int src_n,
                             const double *target, const double *img, const double *grad_out,
                             const double *aux, int B, int N, double voxel_shift, double eps,
                             double *g_source, double *g_target, double *g_img, double *g_volume,
                             void *stream) {
    if (int rc = check64(grad_out, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if ((g_source || g_target || g_img) && !aux) return fail(-1, "ray gradients need the record");
    const long R = (long)B * N;
    if (R == 0) return 0;
    const Ray64 p = make64(nullptr, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift, eps);
    hipLaunchKernelGGL(siddon_bwd64_kernel, grid64(R), dim3(kBlock), 0, (hipStream_t)stream, p, aux,
                       grad_out, g_source, g_target, g_img, g_volume);
    return finish("ddrr_siddon_backward_f64");
}

int ddrr_trilinear_forward_f64(const double *volume, int dx, int dy, int dz, const double *source,
                               int src_n, const double *target, const double *img, int B, int N,
                               double voxel_shift, double eps, int n_points,
                               const double *alphamin, const double *alphamax, double *out,
                               void *stream) {
    if (int rc = check64(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!out || !alphamin || !alphamax) return fail(-1, "null out / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    const long R = (long)B * N;
    if (R == 0) return 0;
    const Ray64 p = make64(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift, eps);
    hipLaunchKernelGGL(trilinear_fwd64_kernel, grid64(R), dim3(kBlock), 0, (hipStream_t)stream, p,
                       n_points, alphamin, alphamax, out);
    return finish("ddrr_trilinear_forward_f64");
}

int ddrr_trilinear_backward_f64(const double *volume, int dx, int dy, int dz,
                                const double *source, int src_n, const double *target,
                                const double *img, const double *grad_out, int B, int N,
                                double voxel_shift, double eps, int n_points,
                                const double *alphamin, const double *alphamax, double *g_source,
                                double *g_target, double *g_img, double *g_alpha, double *g_volume,
                                void *stream) {
    if (int rc = check64(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!grad_out || !alphamin || !alphamax) return fail(-1, "null grad_out / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    const long R = (long)B * N;
    if (R == 0) return 0;
    const Ray64 p = make64(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift, eps);
    hipLaunchKernelGGL(trilinear_bwd64_kernel, grid64(R), dim3(kBlock), 0, (hipStream_t)stream, p,
                       n_points, alphamin, alphamax, grad_out, g_source, g_target, g_img, g_alpha,
                       g_volume);
    return finish("ddrr_trilinear_backward_f64");
}

}  // extern "C"
