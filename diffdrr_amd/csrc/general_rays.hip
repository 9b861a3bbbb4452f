// general_rays.hip -- the materialising general path (general_core.h) behind the C ABI: the
// per-segment / per-sample tensors of the reference and their autograd, in float or double, for
// the keyword combinations the fused kernels do not cover.  One lane per ray, plain 1-D grids:
// these are the reference's rare paths, written for coverage and exactness, not tuned.
#include "runtime.h"

#include "general_core.h"

using namespace ddrr_rt;

namespace {

template <class T>
struct GenArgs {
    const T *vol;
    Dims D;
    const T *source;
    int src_n;
    const T *target;
    const T *img;
    long R;
    int N;
    T shift, eps;
    int align_corners;
};

template <class T>
__device__ __forceinline__ bool gen_load(const GenArgs<T> &p, long r, long &b, long &n, T s[3],
                                         T t[3]) {
    if (r >= p.R) return false;
    b = r / p.N;
    n = r - b * p.N;
    const T *sp = p.source + (b * p.src_n + (p.src_n == 1 ? 0 : n)) * 3;
    const T *tp = p.target + r * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        s[a] = sp[a];
        t[a] = tp[a];
    }
    return true;
}

template <class T>
struct GenAdd {
    T *base;
    __device__ __forceinline__ void operator()(long idx, T v) const {
        unsafeAtomicAdd(base + idx, v);  // global_atomic_add_f32 / _f64
    }
};

template <class T, int LOOKUP>
__global__ __launch_bounds__(kBlock) void segments_kernel(GenArgs<T> p, int raw,
                                                          T *__restrict__ terms) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    long b, n;
    T s[3], t[3];
    if (!gen_load(p, r, b, n, s, t)) return;
    const long M1 = (long)p.D.x + p.D.y + p.D.z + 2;
    ddrr_gen::siddon_segments_ray<T, LOOKUP>(p.vol, p.D, s, t, p.shift, p.eps,
                                             p.align_corners != 0, p.img ? p.img[r] : (T)1,
                                             raw != 0, terms + b * M1 * p.N + n, p.N);
}

template <class T, int LOOKUP>
__global__ __launch_bounds__(kBlock) void segments_bwd_kernel(
    GenArgs<T> p, const T *__restrict__ g_terms, int through, T *__restrict__ g_source,
    T *__restrict__ g_target, T *__restrict__ g_img, T *__restrict__ g_volume) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    long b, n;
    T s[3], t[3];
    if (!gen_load(p, r, b, n, s, t)) return;
    const long M1 = (long)p.D.x + p.D.y + p.D.z + 2;
    T gs[3], gt[3], gi;
    ddrr_gen::siddon_segments_backward_ray<T, LOOKUP>(
        p.vol, p.D, s, t, p.shift, p.eps, p.align_corners != 0, p.img ? p.img[r] : (T)1,
        through != 0, g_terms + b * M1 * p.N + n, p.N, gs, gt, gi, g_volume != nullptr,
        GenAdd<T>{g_volume});
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[r * 3 + a] = gs[a];
        if (g_target) g_target[r * 3 + a] = gt[a];
    }
    if (g_img) g_img[r] = gi;
}

template <class T, bool NEAREST>
__global__ __launch_bounds__(kBlock) void samples_kernel(GenArgs<T> p, int P,
                                                         const T *__restrict__ amin,
                                                         const T *__restrict__ amax, int raw,
                                                         T *__restrict__ samples) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    long b, n;
    T s[3], t[3];
    if (!gen_load(p, r, b, n, s, t)) return;
    ddrr_gen::trilinear_samples_ray<T, NEAREST>(p.vol, p.D, s, t, p.shift, p.eps,
                                                p.align_corners != 0, P, amin[0], amax[0],
                                                p.img ? p.img[r] : (T)1, raw != 0,
                                                samples + b * P * p.N + n, p.N);
}

template <class T, bool NEAREST>
__global__ __launch_bounds__(kBlock) void samples_bwd_kernel(
    GenArgs<T> p, int P, const T *__restrict__ amin, const T *__restrict__ amax,
    const T *__restrict__ g_samples, T *__restrict__ g_source, T *__restrict__ g_target,
    T *__restrict__ g_img, T *__restrict__ g_alpha, T *__restrict__ g_volume) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    long b, n;
    T s[3], t[3];
    if (!gen_load(p, r, b, n, s, t)) return;
    T gs[3], gt[3], ga[2], gi;
    ddrr_gen::trilinear_samples_backward_ray<T, NEAREST>(
        p.vol, p.D, s, t, p.shift, p.eps, p.align_corners != 0, P, amin[0], amax[0],
        p.img ? p.img[r] : (T)1, g_samples + b * P * p.N + n, p.N, gs, gt, ga, gi,
        g_volume != nullptr, GenAdd<T>{g_volume});
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[r * 3 + a] = gs[a];
        if (g_target) g_target[r * 3 + a] = gt[a];
    }
    if (g_img) g_img[r] = gi;
    if (g_alpha) {
        g_alpha[r * 2] = ga[0];
        g_alpha[r * 2 + 1] = ga[1];
    }
}

int check_gen(const void *volume, int dx, int dy, int dz, const void *source, int src_n,
              const void *target, int B, int N, int align_corners) {
    if (!volume || !source || !target) return fail(-1, "null pointer");
    if (dx < 1 || dy < 1 || dz < 1) return fail(-1, "volume dims must be positive");
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (src_n != 1 && src_n != N) return fail(-1, "src_n must be 1 or N");
    if (align_corners != 0 && align_corners != 1) return fail(-1, "align_corners must be 0 or 1");
    return 0;
}

int check_lookup(int lookup, int align_corners) {
    if (lookup != DDRR_LOOKUP_STEP && lookup != DDRR_LOOKUP_MID_NEAREST &&
        lookup != DDRR_LOOKUP_MID_TRILINEAR)
        return fail(-1, "unknown lookup");
    if (lookup == DDRR_LOOKUP_STEP && align_corners)
        return fail(-1, "DDRR_LOOKUP_STEP is the align_corners = 0 nearest lookup");
    return 0;
}

template <class T>
GenArgs<T> make_gen(const void *volume, int dx, int dy, int dz, const void *source, int src_n,
                    const void *target, const void *img, int B, int N, double shift, double eps,
                    int align_corners) {
    GenArgs<T> p;
    p.vol = static_cast<const T *>(volume);
    p.D = Dims{dx, dy, dz};
    p.source = static_cast<const T *>(source);
    p.src_n = src_n;
    p.target = static_cast<const T *>(target);
    p.img = static_cast<const T *>(img);
    p.R = (long)B * N;
    p.N = N;
    p.shift = (T)shift;
    p.eps = (T)eps;
    p.align_corners = align_corners;
    return p;
}

inline dim3 gen_grid(long R) { return dim3((unsigned)((R + kBlock - 1) / kBlock)); }

template <class T>
void launch_segments(const GenArgs<T> &p, int lookup, int raw, void *terms, hipStream_t st) {
    T *o = static_cast<T *>(terms);
    const dim3 grid = gen_grid(p.R), block(kBlock);
    if (lookup == DDRR_LOOKUP_STEP)
        hipLaunchKernelGGL((segments_kernel<T, LOOKUP_STEP>), grid, block, 0, st, p, raw, o);
    else if (lookup == DDRR_LOOKUP_MID_NEAREST)
        hipLaunchKernelGGL((segments_kernel<T, LOOKUP_MID_NEAREST>), grid, block, 0, st, p, raw, o);
    else
        hipLaunchKernelGGL((segments_kernel<T, LOOKUP_MID_TRILINEAR>), grid, block, 0, st, p, raw, o);
}

template <class T>
void launch_segments_bwd(const GenArgs<T> &p, int lookup, const void *g_terms, int through,
                         void *gs, void *gt, void *gi, void *gv, hipStream_t st) {
    const T *g = static_cast<const T *>(g_terms);
    T *a = static_cast<T *>(gs), *b = static_cast<T *>(gt), *c = static_cast<T *>(gi),
      *d = static_cast<T *>(gv);
    const dim3 grid = gen_grid(p.R), block(kBlock);
    if (lookup == DDRR_LOOKUP_STEP)
        hipLaunchKernelGGL((segments_bwd_kernel<T, LOOKUP_STEP>), grid, block, 0, st, p, g, through,
                           a, b, c, d);
    else if (lookup == DDRR_LOOKUP_MID_NEAREST)
        hipLaunchKernelGGL((segments_bwd_kernel<T, LOOKUP_MID_NEAREST>), grid, block, 0, st, p, g,
                           through, a, b, c, d);
    else
        hipLaunchKernelGGL((segments_bwd_kernel<T, LOOKUP_MID_TRILINEAR>), grid, block, 0, st, p, g,
                           through, a, b, c, d);
}

template <class T>
void launch_samples(const GenArgs<T> &p, int P, const void *amin, const void *amax, int nearest,
                    int raw, void *samples, hipStream_t st) {
    const T *a0 = static_cast<const T *>(amin), *a1 = static_cast<const T *>(amax);
    T *o = static_cast<T *>(samples);
    const dim3 grid = gen_grid(p.R), block(kBlock);
    if (nearest)
        hipLaunchKernelGGL((samples_kernel<T, true>), grid, block, 0, st, p, P, a0, a1, raw, o);
    else
        hipLaunchKernelGGL((samples_kernel<T, false>), grid, block, 0, st, p, P, a0, a1, raw, o);
}

template <class T>
void launch_samples_bwd(const GenArgs<T> &p, int P, const void *amin, const void *amax,
                        int nearest, const void *g_samples, void *gs, void *gt, void *gi, void *ga,
                        void *gv, hipStream_t st) {
    const T *a0 = static_cast<const T *>(amin), *a1 = static_cast<const T *>(amax);
    const T *g = static_cast<const T *>(g_samples);
    T *a = static_cast<T *>(gs), *b = static_cast<T *>(gt), *c = static_cast<T *>(gi),
      *d = static_cast<T *>(ga), *e = static_cast<T *>(gv);
    const dim3 grid = gen_grid(p.R), block(kBlock);
    if (nearest)
        hipLaunchKernelGGL((samples_bwd_kernel<T, true>), grid, block, 0, st, p, P, a0, a1, g, a, b,
                           c, d, e);
    else
        hipLaunchKernelGGL((samples_bwd_kernel<T, false>), grid, block, 0, st, p, P, a0, a1, g, a,
                           b, c, d, e);
}

}  // namespace

extern "C" {

int ddrr_siddon_segments_general(const void *volume, int f64, int dx, int dy, int dz,
                                 const void *source, int src_n, const void *target,
                                 const void *img, int B, int N, double voxel_shift, double eps,
                                 int lookup, int align_corners, int raw, void *terms,
                                 void *stream) {
    if (int rc = check_gen(volume, dx, dy, dz, source, src_n, target, B, N, align_corners)) return rc;
    if (int rc = check_lookup(lookup, align_corners)) return rc;
    if (!terms) return fail(-1, "null terms pointer");
    if ((long)B * N == 0) return 0;
    if (f64)
        launch_segments(make_gen<double>(volume, dx, dy, dz, source, src_n, target, img, B, N,
                                         voxel_shift, eps, align_corners),
                        lookup, raw, terms, (hipStream_t)stream);
    else
        launch_segments(make_gen<float>(volume, dx, dy, dz, source, src_n, target, img, B, N,
                                        voxel_shift, eps, align_corners),
                        lookup, raw, terms, (hipStream_t)stream);
    return finish("ddrr_siddon_segments_general");
}

int ddrr_siddon_segments_general_backward(const void *volume, int f64, int dx, int dy, int dz,
                                          const void *source, int src_n, const void *target,
                                          const void *img, const void *grad_terms, int B, int N,
                                          double voxel_shift, double eps, int lookup,
                                          int align_corners, int through_lookup, void *g_source,
                                          void *g_target, void *g_img, void *g_volume,
                                          void *stream) {
    if (int rc = check_gen(volume, dx, dy, dz, source, src_n, target, B, N, align_corners)) return rc;
    if (int rc = check_lookup(lookup, align_corners)) return rc;
    if (!grad_terms) return fail(-1, "null grad_terms pointer");
    if (!through_lookup && (g_volume || g_img))
        return fail(-1, "stop_gradients_through_grid_sample has no volume / img gradient");
    if ((long)B * N == 0) return 0;
    if (f64)
        launch_segments_bwd(make_gen<double>(volume, dx, dy, dz, source, src_n, target, img, B, N,
                                             voxel_shift, eps, align_corners),
                            lookup, grad_terms, through_lookup, g_source, g_target, g_img, g_volume,
                            (hipStream_t)stream);
    else
        launch_segments_bwd(make_gen<float>(volume, dx, dy, dz, source, src_n, target, img, B, N,
                                            voxel_shift, eps, align_corners),
                            lookup, grad_terms, through_lookup, g_source, g_target, g_img, g_volume,
                            (hipStream_t)stream);
    return finish("ddrr_siddon_segments_general_backward");
}

int ddrr_trilinear_samples_general(const void *volume, int f64, int dx, int dy, int dz,
                                   const void *source, int src_n, const void *target,
                                   const void *img, int B, int N, double voxel_shift, double eps,
                                   int n_points, const void *alphamin, const void *alphamax,
                                   int nearest, int align_corners, int raw, void *samples,
                                   void *stream) {
    if (int rc = check_gen(volume, dx, dy, dz, source, src_n, target, B, N, align_corners)) return rc;
    if (!samples || !alphamin || !alphamax) return fail(-1, "null samples / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if ((long)B * N == 0) return 0;
    if (f64)
        launch_samples(make_gen<double>(volume, dx, dy, dz, source, src_n, target, img, B, N,
                                        voxel_shift, eps, align_corners),
                       n_points, alphamin, alphamax, nearest, raw, samples, (hipStream_t)stream);
    else
        launch_samples(make_gen<float>(volume, dx, dy, dz, source, src_n, target, img, B, N,
                                       voxel_shift, eps, align_corners),
                       n_points, alphamin, alphamax, nearest, raw, samples, (hipStream_t)stream);
    return finish("ddrr_trilinear_samples_general");
}

int ddrr_trilinear_samples_general_backward(const void *volume, int f64, int dx, int dy, int dz,
                                            const void *source, int src_n, const void *target,
                                            const void *img, const void *grad_samples, int B,
                                            int N, double voxel_shift, double eps, int n_points,
                                            const void *alphamin, const void *alphamax,
                                            int nearest, int align_corners, void *g_source,
                                            void *g_target, void *g_img, void *g_alpha,
                                            void *g_volume, void *stream) {
    if (int rc = check_gen(volume, dx, dy, dz, source, src_n, target, B, N, align_corners)) return rc;
    if (!grad_samples || !alphamin || !alphamax)
        return fail(-1, "null grad_samples / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if ((long)B * N == 0) return 0;
    if (f64)
        launch_samples_bwd(make_gen<double>(volume, dx, dy, dz, source, src_n, target, img, B, N,
                                            voxel_shift, eps, align_corners),
                           n_points, alphamin, alphamax, nearest, grad_samples, g_source, g_target,
                           g_img, g_alpha, g_volume, (hipStream_t)stream);
    else
        launch_samples_bwd(make_gen<float>(volume, dx, dy, dz, source, src_n, target, img, B, N,
                                           voxel_shift, eps, align_corners),
                           n_points, alphamin, alphamax, nearest, grad_samples, g_source, g_target,
                           g_img, g_alpha, g_volume, (hipStream_t)stream);
    return finish("ddrr_trilinear_samples_general_backward");
}

}  // extern "C"
