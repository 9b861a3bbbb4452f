// trilinear_rays.hip -- the per-ray trilinear marcher (trilinear_core.h): forward, backward,
// mask_to_channels, and their C-ABI entries.  The volume-stationary forms live in bricks.hip.
#include "runtime.h"
#include "siddon_core.h"
#include "trilinear_core.h"

using namespace ddrr;
using namespace ddrr_rt;

namespace {

// --------------------------------------------------------------- Trilinear

template <int REDUCE, bool NEAREST>
__global__ __launch_bounds__(kBlock) void trilinear_fwd_kernel(RayArgs p, int n_points,
                                                               const float *__restrict__ amin,
                                                               const float *__restrict__ amax,
                                                               int align_corners,
                                                               float *__restrict__ out) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float I = trilinear_forward_ray<REDUCE, NEAREST>(p.vol, p.D, s, t, p.shift, p.eps,
                                                           n_points, amin[0], amax[0],
                                                           align_corners != 0);
    const float L = p.img ? p.img[id.r] : 1.f;
    out[id.r] = L * I;
}

__global__ __launch_bounds__(kBlock) void trilinear_fwd_channels_kernel(
    RayArgs p, const unsigned char *__restrict__ labels, int C, int n_points,
    const float *__restrict__ amin, const float *__restrict__ amax, int align_corners,
    float *__restrict__ out) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    float *col = out + (long)id.b * C * p.N + id.n;  // stride N between channels
    trilinear_channels_ray(p.vol, labels, p.D, s, t, p.shift, p.eps, n_points, amin[0], amax[0],
                           align_corners != 0, ColumnFlush{col, p.N, C, L});
}

template <bool NEAREST, bool WANT_VOL>
__global__ __launch_bounds__(kBlock) void trilinear_bwd_kernel(
    RayArgs p, const float *__restrict__ grad_out, int n_points, const float *__restrict__ amin,
    const float *__restrict__ amax, int align_corners, float *__restrict__ g_source,
    float *__restrict__ g_target, float *__restrict__ g_img, float *__restrict__ g_alpha,
    float *__restrict__ g_volume) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    const float g = grad_out[id.r];
    const float a0 = amin[0], a1 = amax[0];
    MarchGrad r;
    if (WANT_VOL)
        r = trilinear_backward_ray<NEAREST, true>(p.vol, p.D, s, t, p.shift, p.eps, n_points, a0,
                                                  a1, align_corners != 0, g * L,
                                                  AtomicAdder{g_volume});
    else
        r = trilinear_backward_ray<NEAREST, false>(p.vol, p.D, s, t, p.shift, p.eps, n_points,
                                                   a0, a1, align_corners != 0, g * L, NoAdd{});
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[id.r * 3 + a] = r.gs[a];
        if (g_target) g_target[id.r * 3 + a] = r.gt[a];
    }
    if (g_img) g_img[id.r] = g * r.sumT * ((a1 - a0) / (float)(n_points - 1));
    if (g_alpha) {
        g_alpha[id.r * 2 + 0] = r.g_amin;
        g_alpha[id.r * 2 + 1] = r.g_amax;
    }
}

// Backward of the marcher's mask_to_channels (renderers.py:242-252): every sample carries
// the incoming gradient of the channel its nearest label selects.
template <bool WANT_VOL>
__global__ __launch_bounds__(kBlock) void trilinear_bwd_channels_kernel(
    RayArgs p, const unsigned char *__restrict__ labels, int C,
    const float *__restrict__ grad_out, int n_points, const float *__restrict__ amin,
    const float *__restrict__ amax, int align_corners, float *__restrict__ g_source,
    float *__restrict__ g_target, float *__restrict__ g_img, float *__restrict__ g_alpha,
    float *__restrict__ g_volume) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    const float a0 = amin[0], a1 = amax[0];
    const LabelWeight wt{labels, p.D, grad_out + (long)id.b * C * p.N + id.n, p.N, C, a0,
                         p.shift, {s[0], s[1], s[2]}, align_corners != 0};
    MarchGrad r;
    if (WANT_VOL)
        r = trilinear_backward_ray<false, true>(p.vol, p.D, s, t, p.shift, p.eps, n_points, a0, a1,
                                                align_corners != 0, L, AtomicAdder{g_volume}, wt);
    else
        r = trilinear_backward_ray<false, false>(p.vol, p.D, s, t, p.shift, p.eps, n_points, a0,
                                                 a1, align_corners != 0, L, NoAdd{}, wt);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[id.r * 3 + a] = r.gs[a];
        if (g_target) g_target[id.r * 3 + a] = r.gt[a];
    }
    if (g_img) g_img[id.r] = r.sumT * ((a1 - a0) / (float)(n_points - 1));
    if (g_alpha) {
        g_alpha[id.r * 2 + 0] = r.g_amin;
        g_alpha[id.r * 2 + 1] = r.g_amax;
    }
}

// The materialised per-sample tensor for a callable reducefn; samples is (B, P, N): the ray
// owns column [b, :, n], consecutive lanes write consecutive floats.
template <bool NEAREST>
__global__ __launch_bounds__(kBlock) void trilinear_samples_kernel(
    RayArgs p, int n_points, const float *__restrict__ amin, const float *__restrict__ amax,
    int align_corners, float *__restrict__ samples) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    trilinear_samples_ray<NEAREST>(p.vol, p.D, s, t, p.shift, p.eps, n_points, amin[0], amax[0],
                                   align_corners != 0, L,
                                   samples + (long)id.b * n_points * p.N + id.n, p.N);
}

template <bool NEAREST, bool WANT_VOL>
__global__ __launch_bounds__(kBlock) void trilinear_samples_bwd_kernel(
    RayArgs p, const float *__restrict__ g_samples, int n_points, const float *__restrict__ amin,
    const float *__restrict__ amax, int align_corners, float *__restrict__ g_source,
    float *__restrict__ g_target, float *__restrict__ g_img, float *__restrict__ g_alpha,
    float *__restrict__ g_volume) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    const float a0 = amin[0], a1 = amax[0];
    const SampleWeight wt{g_samples + (long)id.b * n_points * p.N + id.n, p.N};
    MarchGrad r;
    if (WANT_VOL)
        r = trilinear_backward_ray<NEAREST, true>(p.vol, p.D, s, t, p.shift, p.eps, n_points, a0,
                                                  a1, align_corners != 0, L,
                                                  AtomicAdder{g_volume}, wt);
    else
        r = trilinear_backward_ray<NEAREST, false>(p.vol, p.D, s, t, p.shift, p.eps, n_points, a0,
                                                   a1, align_corners != 0, L, NoAdd{}, wt);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[id.r * 3 + a] = r.gs[a];
        if (g_target) g_target[id.r * 3 + a] = r.gt[a];
    }
    if (g_img) g_img[id.r] = r.sumT * ((a1 - a0) / (float)(n_points - 1));
    if (g_alpha) {
        g_alpha[id.r * 2 + 0] = r.g_amin;
        g_alpha[id.r * 2 + 1] = r.g_amax;
    }
}

// reducefn = "max": the gradient of the arg-max sample alone (one march to find it, one for
// its gradient).
template <bool NEAREST, bool WANT_VOL>
__global__ __launch_bounds__(kBlock) void trilinear_bwd_max_kernel(
    RayArgs p, const float *__restrict__ grad_out, int n_points, const float *__restrict__ amin,
    const float *__restrict__ amax, int align_corners, float *__restrict__ g_source,
    float *__restrict__ g_target, float *__restrict__ g_img, float *__restrict__ g_alpha,
    float *__restrict__ g_volume) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    const float g = grad_out[id.r];
    const float a0 = amin[0], a1 = amax[0];
    const bool ac = align_corners != 0;
    const OneSampleWeight wt{
        trilinear_argmax_ray<NEAREST>(p.vol, p.D, s, t, p.shift, p.eps, n_points, a0, a1, ac)};
    MarchGrad r;
    if (WANT_VOL)
        r = trilinear_backward_ray<NEAREST, true>(p.vol, p.D, s, t, p.shift, p.eps, n_points, a0,
                                                  a1, ac, g * L, AtomicAdder{g_volume}, wt);
    else
        r = trilinear_backward_ray<NEAREST, false>(p.vol, p.D, s, t, p.shift, p.eps, n_points, a0,
                                                   a1, ac, g * L, NoAdd{}, wt);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[id.r * 3 + a] = r.gs[a];
        if (g_target) g_target[id.r * 3 + a] = r.gt[a];
    }
    if (g_img) g_img[id.r] = g * r.sumT * ((a1 - a0) / (float)(n_points - 1));
    if (g_alpha) {
        g_alpha[id.r * 2 + 0] = r.g_amin;
        g_alpha[id.r * 2 + 1] = r.g_amax;
    }
}

// Batch-global marching range (renderers.py:220-223): min over all rays of alphamin, max of
// alphamax.  At most 64 workgroups of 1024 threads stride over the rays; a workgroup reduces its
// rays through shuffles and LDS and issues ONE atomic pair on the floats' bit patterns (round 4:
// one pair per wave, 8 k same-address atomics at 512^2 rays = 0.105 ms; now at most 128).
// alphamin >= +0: unsigned order is float order.  alphamax <= 1 may be negative (every ray leaves
// the volume behind the source): non-negative values compete as signed ints (max), negative ones
// as unsigned ints (min: the least negative wins, any non-negative beats them).  Both words start
// at 0xffffffff -- above every float as an unsigned int, -1 (below every non-negative float) as a
// signed one -- so that ONE 8-byte memset initialises both.
constexpr int kRangeBlock = 1024, kRangeBlocks = 64;
__global__ __launch_bounds__(kRangeBlock) void alpha_range_kernel(const float *__restrict__ source,
                                                                  int src_n,
                                                                  const float *__restrict__ target,
                                                                  long R, int N, Dims D, float shift,
                                                                  float eps, unsigned *range) {
    __shared__ float red[2][kRangeBlock / 64];
    float lo = INFINITY, hi = -INFINITY;
    // four rays per thread and round: the round's loads are in flight together (rays beyond the
    // batch re-read the last ray: it is a member of the min / max anyway)
    const long stride = (long)gridDim.x * kRangeBlock;
    for (long r0 = (long)blockIdx.x * kRangeBlock + threadIdx.x; r0 < R; r0 += 4 * stride) {
        float s[4][3], t[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long rk = r0 + k * stride, r = rk < R ? rk : R - 1;
            const long b = r / N, n = r - b * N;
            const float *sp = source + (b * src_n + (src_n == 1 ? 0 : n)) * 3, *tp = target + r * 3;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                s[k][a] = sp[a];
                t[k][a] = tp[a];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a0, a1;
            ddrr::ray_alpha_range(D, s[k], t[k], shift, eps, a0, a1);
            lo = fminf(lo, a0);
            hi = fmaxf(hi, a1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = lo;
        red[1][threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        lo = threadIdx.x < kRangeBlock / 64 ? red[0][threadIdx.x] : INFINITY;
        hi = threadIdx.x < kRangeBlock / 64 ? red[1][threadIdx.x] : -INFINITY;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, 64));
            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
        }
        // (a workgroup without a ray -- none is launched -- would hold +inf / -inf: harmless)
        if (threadIdx.x == 0) {
            atomicMin(range, __float_as_uint(lo + 0.f));  // (-0 + 0 = +0)
            if (hi >= 0.f)
                atomicMax(reinterpret_cast<int *>(range + 1), __float_as_int(hi));
            else
                atomicMin(range + 1, __float_as_uint(hi));
        }
    }
}

}  // namespace

extern "C" {

int ddrr_trilinear_alpha_range(const float *source, int src_n, const float *target, int B, int N,
                               int dx, int dy, int dz, float voxel_shift, float eps,
                               float *range2, void *stream) {
    if (!source || !target || !range2) return fail(-1, "null pointer");
    if (dx < 1 || dy < 1 || dz < 1) return fail(-1, "volume dims must be positive");
    if (B < 1 || N < 1) return fail(-1, "the marching range of an empty ray batch is undefined");
    if (src_n != 1 && src_n != N) return fail(-1, "src_n must be 1 or N");
    hipStream_t st = (hipStream_t)stream;
    // both words 0xffffffff (see alpha_range_kernel)
    const hipError_t e = hipMemsetAsync(range2, 0xff, 2 * sizeof(float), st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    const long R = (long)B * N;
    const long blocks = (R + kRangeBlock - 1) / kRangeBlock;
    hipLaunchKernelGGL(alpha_range_kernel, dim3((unsigned)(blocks < kRangeBlocks ? blocks : kRangeBlocks)),
                       dim3(kRangeBlock), 0, st, source, src_n, target, R, N, Dims{dx, dy, dz},
                       voxel_shift, eps, reinterpret_cast<unsigned *>(range2));
    return finish("ddrr_trilinear_alpha_range");
}

int ddrr_trilinear_forward(const float *volume, int dx, int dy, int dz, const float *source,
                           int src_n, const float *target, const float *img, int B, int N,
                           float voxel_shift, float eps, int n_points, const float *alphamin,
                           const float *alphamax, int mode_nearest, int reduce_mode,
                           int align_corners, int det_h, int det_w, int tile_h, int tile_w,
                           float *out, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!out || !alphamin || !alphamax) return fail(-1, "null out / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, det_h, det_w, tile_h, tile_w);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_for(p)), block(kBlock);
    const bool sum = reduce_mode == DDRR_REDUCE_SUM;
    if (!sum && reduce_mode != DDRR_REDUCE_MAX) return fail(-1, "bad reduce_mode");
#define LAUNCH(R, NN)                                                                          \
    hipLaunchKernelGGL((trilinear_fwd_kernel<R, NN>), grid, block, 0, st, p, n_points, alphamin, \
                       alphamax, align_corners, out)
    if (sum && !mode_nearest) LAUNCH(REDUCE_SUM, false);
    else if (sum) LAUNCH(REDUCE_SUM, true);
    else if (!mode_nearest) LAUNCH(REDUCE_MAX, false);
    else LAUNCH(REDUCE_MAX, true);
#undef LAUNCH
    return finish("ddrr_trilinear_forward");
}

int ddrr_trilinear_forward_channels(const float *volume, const unsigned char *labels, int dx,
                                    int dy, int dz, const float *source, int src_n,
                                    const float *target, const float *img, int B, int N, int C,
                                    float voxel_shift, float eps, int n_points,
                                    const float *alphamin, const float *alphamax,
                                    int align_corners, int det_h, int det_w, int tile_h,
                                    int tile_w, float *out, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!labels || !out || C < 1) return fail(-1, "null labels/out or C < 1");
    if (!alphamin || !alphamax) return fail(-1, "null alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (B == 0 || N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C * N, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, det_h, det_w, tile_h, tile_w);
    hipLaunchKernelGGL(trilinear_fwd_channels_kernel, dim3(grid_for(p)), dim3(kBlock), 0, st, p,
                       labels, C, n_points, alphamin, alphamax, align_corners, out);
    return finish("ddrr_trilinear_forward_channels");
}

int ddrr_trilinear_backward(const float *volume, int dx, int dy, int dz, const float *source,
                            int src_n, const float *target, const float *img,
                            const float *grad_out, int B, int N, float voxel_shift, float eps,
                            int n_points, const float *alphamin, const float *alphamax,
                            int mode_nearest, int align_corners, int det_h, int det_w, int tile_h,
                            int tile_w, float *g_source, float *g_target, float *g_img,
                            float *g_alpha, float *g_volume, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!grad_out || !alphamin || !alphamax) return fail(-1, "null grad_out / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, det_h, det_w, tile_h, tile_w);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_for(p)), block(kBlock);
#define LAUNCH(NN, WV)                                                                          \
    hipLaunchKernelGGL((trilinear_bwd_kernel<NN, WV>), grid, block, 0, st, p, grad_out, n_points, \
                       alphamin, alphamax, align_corners, g_source, g_target, g_img, g_alpha,     \
                       g_volume)
    if (mode_nearest && g_volume) LAUNCH(true, true);
    else if (mode_nearest) LAUNCH(true, false);
    else if (g_volume) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    return finish("ddrr_trilinear_backward");
}

int ddrr_trilinear_backward_channels(const float *volume, const unsigned char *labels, int dx,
                                     int dy, int dz, const float *source, int src_n,
                                     const float *target, const float *img,
                                     const float *grad_out, int B, int N, int C,
                                     float voxel_shift, float eps, int n_points,
                                     const float *alphamin, const float *alphamax,
                                     int align_corners, int det_h, int det_w, int tile_h,
                                     int tile_w, float *g_source, float *g_target, float *g_img,
                                     float *g_alpha, float *g_volume, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!labels || !grad_out || !alphamin || !alphamax || C < 1)
        return fail(-1, "null labels / grad_out / alphamin / alphamax or C < 1");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, det_h, det_w, tile_h, tile_w);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_for(p)), block(kBlock);
    if (g_volume)
        hipLaunchKernelGGL((trilinear_bwd_channels_kernel<true>), grid, block, 0, st, p, labels, C,
                           grad_out, n_points, alphamin, alphamax, align_corners, g_source,
                           g_target, g_img, g_alpha, g_volume);
    else
        hipLaunchKernelGGL((trilinear_bwd_channels_kernel<false>), grid, block, 0, st, p, labels, C,
                           grad_out, n_points, alphamin, alphamax, align_corners, g_source,
                           g_target, g_img, g_alpha, g_volume);
    return finish("ddrr_trilinear_backward_channels");
}

int ddrr_trilinear_samples(const float *volume, int dx, int dy, int dz, const float *source,
                           int src_n, const float *target, const float *img, int B, int N,
                           float voxel_shift, float eps, int n_points, const float *alphamin,
                           const float *alphamax, int mode_nearest, int align_corners,
                           float *samples, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!samples || !alphamin || !alphamax) return fail(-1, "null samples / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, 0, 0, 1, 64);
    const dim3 grid(grid_for(p)), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    if (mode_nearest)
        hipLaunchKernelGGL((trilinear_samples_kernel<true>), grid, block, 0, st, p, n_points,
                           alphamin, alphamax, align_corners, samples);
    else
        hipLaunchKernelGGL((trilinear_samples_kernel<false>), grid, block, 0, st, p, n_points,
                           alphamin, alphamax, align_corners, samples);
    return finish("ddrr_trilinear_samples");
}

int ddrr_trilinear_samples_backward(const float *volume, int dx, int dy, int dz,
                                    const float *source, int src_n, const float *target,
                                    const float *img, const float *grad_samples, int B, int N,
                                    float voxel_shift, float eps, int n_points,
                                    const float *alphamin, const float *alphamax,
                                    int mode_nearest, int align_corners, float *g_source,
                                    float *g_target, float *g_img, float *g_alpha,
                                    float *g_volume, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!grad_samples || !alphamin || !alphamax)
        return fail(-1, "null grad_samples / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, 0, 0, 1, 64);
    const dim3 grid(grid_for(p)), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(NN, WV)                                                                            \
    hipLaunchKernelGGL((trilinear_samples_bwd_kernel<NN, WV>), grid, block, 0, st, p, grad_samples, \
                       n_points, alphamin, alphamax, align_corners, g_source, g_target, g_img,     \
                       g_alpha, g_volume)
    if (mode_nearest && g_volume) LAUNCH(true, true);
    else if (mode_nearest) LAUNCH(true, false);
    else if (g_volume) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    return finish("ddrr_trilinear_samples_backward");
}

int ddrr_trilinear_backward_max(const float *volume, int dx, int dy, int dz, const float *source,
                                int src_n, const float *target, const float *img,
                                const float *grad_out, int B, int N, float voxel_shift, float eps,
                                int n_points, const float *alphamin, const float *alphamax,
                                int mode_nearest, int align_corners, float *g_source,
                                float *g_target, float *g_img, float *g_alpha, float *g_volume,
                                void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!grad_out || !alphamin || !alphamax) return fail(-1, "null grad_out / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, 0, 0, 1, 64);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_for(p)), block(kBlock);
#define LAUNCH(NN, WV)                                                                           \
    hipLaunchKernelGGL((trilinear_bwd_max_kernel<NN, WV>), grid, block, 0, st, p, grad_out,       \
                       n_points, alphamin, alphamax, align_corners, g_source, g_target, g_img,    \
                       g_alpha, g_volume)
    if (mode_nearest && g_volume) LAUNCH(true, true);
    else if (mode_nearest) LAUNCH(true, false);
    else if (g_volume) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    return finish("ddrr_trilinear_backward_max");
}

}  // extern "C"
