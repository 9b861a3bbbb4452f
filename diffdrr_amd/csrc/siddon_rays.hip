// siddon_rays.hip -- the per-ray Siddon kernels (one detector ray per lane; no LDS, no MFMA:
// a gather-bound line integral) and their C-ABI entries: the generic walk
// and its backward passes, mask_to_channels, the materialised
// per-segment tensor.  The volume-stationary kernels live in bricks.hip.
#include "runtime.h"
#include "siddon_core.h"
#include "record_pack.h"
#include "record_layout.h"
#include "segments_core.h"

using namespace ddrr;
using namespace ddrr_rt;

namespace {

// ------------------------------------------------------------------ Siddon

template <int REDUCE, bool AUX, bool COUNT>
__global__ __launch_bounds__(kBlock) void siddon_fwd_kernel(RayArgs p, float *__restrict__ out,
                                                            float *__restrict__ aux,
                                                            int *__restrict__ n_vox) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    float rec[SIDDON_AUX];
    int cnt = 0;
    const float I = siddon_forward_ray<REDUCE, AUX, COUNT>(p.vol, p.D, full_box(p.D), s, t, p.shift,
                                                           p.eps, rec, &cnt);
    const float L = p.img ? p.img[id.r] : 1.f;
    out[id.r] = L * I;
    if (AUX) {
        float4 *a4 = reinterpret_cast<float4 *>(aux + id.r * SIDDON_AUX);
        a4[0] = make_float4(rec[0], rec[1], rec[2], rec[3]);
        a4[1] = make_float4(rec[4], rec[5], rec[6], rec[7]);
    }
    if (COUNT) n_vox[id.r] = cnt;
}

template <int REDUCE, int LOOKUP>
__global__ __launch_bounds__(kBlock) void siddon_fwd_mid_kernel(RayArgs p, int align_corners,
                                                                float *__restrict__ out) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float I = siddon_forward_ray_midpoint<REDUCE, LOOKUP>(p.vol, p.D, s, t, p.shift, p.eps,
                                                                align_corners != 0);
    const float L = p.img ? p.img[id.r] : 1.f;
    out[id.r] = L * I;
}

template <int REDUCE>
__global__ __launch_bounds__(kBlock) void siddon_bwd_rays_kernel(
    const float *__restrict__ aux, const float *__restrict__ grad_out,
    const float *__restrict__ source, int src_n, const float *__restrict__ target,
    const float *__restrict__ img, long R, int N, float eps, int layout,
    float *__restrict__ g_source, float *__restrict__ g_target, float *__restrict__ g_img) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    if (r >= R) return;
    const long b = r / N;
    const int n = (int)(r - b * N);
    const float *sp = source + (b * src_n + (src_n == 1 ? 0 : n)) * 3;
    const float *tp = target + r * 3;
    const float s[3] = {sp[0], sp[1], sp[2]}, t[3] = {tp[0], tp[1], tp[2]};
    float rec[SIDDON_AUX];
    if (layout == DDRR_AUX_BLOCKED) {
        // float record of the brick kernel (record_layout.h): I, S0x, S0z, S1x, S1z; the y
        // components follow from sum_a S0_a = 0, sum_a S1_a = I
        rec_blocked_load(aux, r, rec);
    } else if (layout == DDRR_AUX_PACKED) {
        // its packed fixed-point form (record_pack.h): planes of R elements
        float S0x, S0z, S1x, S1z;
        const long long *X = reinterpret_cast<const long long *>(aux);
        const float q = aux[6 * R], qa = q / aux[5 * R + r];
        record_unpack(X[r], q, qa, S0x, S1x);
        record_unpack(X[R + r], q, qa, S0z, S1z);
        const float I = aux[4 * R + r];
        rec[0] = I;
        rec[1] = S0x;
        rec[2] = -(S0x + S0z);
        rec[3] = S0z;
        rec[4] = S1x;
        rec[5] = I - (S1x + S1z);
        rec[6] = S1z;
        rec[7] = 0.f;
    } else {
        const float4 *a4 = reinterpret_cast<const float4 *>(aux + r * SIDDON_AUX);
        const float4 lo = a4[0], hi = a4[1];
        rec[0] = lo.x, rec[1] = lo.y, rec[2] = lo.z, rec[3] = lo.w;
        rec[4] = hi.x, rec[5] = hi.y, rec[6] = hi.z, rec[7] = hi.w;
    }
    const float g = grad_out[r];
    const float L = img ? img[r] : 1.f;
    float gs[3], gt[3];
    siddon_backward_ray<REDUCE>(rec, s, t, eps, g * L, gs, gt);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[r * 3 + a] = gs[a];
        if (g_target) g_target[r * 3 + a] = gt[a];
    }
    if (g_img) g_img[r] = g * rec[0];
}

template <int REDUCE>
__global__ __launch_bounds__(kBlock) void siddon_bwd_volume_kernel(
    RayArgs p, const float *__restrict__ grad_out, float *__restrict__ g_volume) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    const float gl = grad_out[id.r] * L;
    if (gl == 0.f) return;
    siddon_scatter_ray<REDUCE>(p.vol, p.D, s, t, p.shift, p.eps, gl, AtomicAdder{g_volume});
}

// mask_to_channels (renderers.py:77-89): the ray owns column out[b, :, n]; runs
// of one label are flushed with a plain read-modify-write (siddon_channels_ray).
__global__ __launch_bounds__(kBlock) void siddon_fwd_channels_kernel(
    RayArgs p, const unsigned char *__restrict__ labels, int C, float *__restrict__ out) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    float *col = out + (long)id.b * C * p.N + id.n;  // stride N between channels
    siddon_channels_ray(p.vol, labels, p.D, s, t, p.shift, p.eps, ColumnFlush{col, p.N, C, L});
}

// Backward of mask_to_channels (what autograd of renderers.py:77-89 returns): the loss
// gradient of a segment is that of the channel its label selects, so the ray is walked
// once more over the WEIGHTED volume v * grad_out[b, label, n]; the record of that walk
// gives the endpoint gradients exactly as in the single-channel case (gl = ray length).
struct ChannelFetch {
    const float *vol;
    const unsigned char *labels;
    const float *gcol;  // grad_out + [b, 0, n]
    long stride;        // N: distance between channels
    int C;
    __device__ __forceinline__ float operator()(unsigned boff) const {
        const unsigned idx = boff >> 2;
        const int lab = labels[idx];
        return lab < C ? vol[idx] * gcol[lab * stride] : 0.f;
    }
};

struct ChannelAdder {
    float *g_volume;
    const unsigned char *labels;
    const float *gcol;
    long stride;
    int C;
    __device__ __forceinline__ void operator()(unsigned idx, float v) const {
        const int lab = labels[idx];
        if (lab < C) unsafeAtomicAdd(g_volume + idx, v * gcol[lab * stride]);
    }
};

__global__ __launch_bounds__(kBlock) void siddon_bwd_channels_kernel(
    RayArgs p, const unsigned char *__restrict__ labels, int C,
    const float *__restrict__ grad_out, float *__restrict__ g_source,
    float *__restrict__ g_target, float *__restrict__ g_img, float *__restrict__ g_volume) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    const float *gcol = grad_out + (long)id.b * C * p.N + id.n;
    if (g_source || g_target || g_img) {
        float rec[SIDDON_AUX];
        siddon_forward_ray_t<REDUCE_SUM, true, false>(ChannelFetch{p.vol, labels, gcol, p.N, C},
                                                      global_store(p.D), full_box(p.D), s, t,
                                                      p.shift, p.eps, rec, nullptr);
        float gs[3], gt[3];
        siddon_backward_ray<REDUCE_SUM>(rec, s, t, p.eps, L, gs, gt);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (g_source) g_source[id.r * 3 + a] = gs[a];
            if (g_target) g_target[id.r * 3 + a] = gt[a];
        }
        if (g_img) g_img[id.r] = rec[0];
    }
    if (g_volume)
        siddon_scatter_ray<REDUCE_SUM>(p.vol, p.D, s, t, p.shift, p.eps, L,
                                       ChannelAdder{g_volume, labels, gcol, p.N, C});
}

// The materialised per-segment terms for a callable reducefn (segments_core.h).  terms is
// (B, M - 1, N): the ray owns column [b, :, n], consecutive lanes write consecutive floats.
__global__ __launch_bounds__(kBlock) void siddon_segments_kernel(RayArgs p,
                                                                 float *__restrict__ terms) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    const long M1 = (long)p.D.x + p.D.y + p.D.z + 2;
    siddon_segments_ray(p.vol, p.D, s, t, p.shift, p.eps, L, terms + (long)id.b * M1 * p.N + id.n,
                        p.N);
}

template <bool WANT_VOL>
__global__ __launch_bounds__(kBlock) void siddon_segments_bwd_kernel(
    RayArgs p, const float *__restrict__ g_terms, float *__restrict__ g_source,
    float *__restrict__ g_target, float *__restrict__ g_img, float *__restrict__ g_volume) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    const long M1 = (long)p.D.x + p.D.y + p.D.z + 2;
    float gs[3], gt[3], gi;
    siddon_segments_backward_ray<WANT_VOL>(p.vol, p.D, s, t, p.shift, p.eps, L,
                                           g_terms + (long)id.b * M1 * p.N + id.n, p.N, gs, gt, gi,
                                           AtomicAdder{g_volume});
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[id.r * 3 + a] = gs[a];
        if (g_target) g_target[id.r * 3 + a] = gt[a];
    }
    if (g_img) g_img[id.r] = gi;
}

// Backward of the midpoint-lookup forms (align_corners = True, mode "bilinear"): one more walk
// (siddon_backward_ray_midpoint), no record.
template <int LOOKUP, bool WANT_VOL>
__global__ __launch_bounds__(kBlock) void siddon_bwd_mid_kernel(
    RayArgs p, const float *__restrict__ grad_out, int align_corners, float *__restrict__ g_source,
    float *__restrict__ g_target, float *__restrict__ g_img, float *__restrict__ g_volume) {
    const RayId id = ray_id(p);
    if (id.n < 0) return;
    float s[3], t[3];
    load_ray(p, id, s, t);
    const float L = p.img ? p.img[id.r] : 1.f;
    const float g = grad_out[id.r];
    float gs[3], gt[3];
    const float I = siddon_backward_ray_midpoint<LOOKUP, WANT_VOL>(
        p.vol, p.D, s, t, p.shift, p.eps, align_corners != 0, g * L, gs, gt, AtomicAdder{g_volume});
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[id.r * 3 + a] = gs[a];
        if (g_target) g_target[id.r * 3 + a] = gt[a];
    }
    if (g_img) g_img[id.r] = g * I;
}

}  // namespace

extern "C" {

int ddrr_siddon_forward(const float *volume, int dx, int dy, int dz, const float *source,
                        int src_n, const float *target, const float *img, int B, int N,
                        float voxel_shift, float eps, int reduce_mode, int lookup_mode,
                        int align_corners, int det_h, int det_w, int tile_h, int tile_w,
                        float *out, float *aux, int *n_vox, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!out) return fail(-1, "null out pointer");
    if (reduce_mode != DDRR_REDUCE_SUM && reduce_mode != DDRR_REDUCE_MAX)
        return fail(-1, "reduce_mode must be DDRR_REDUCE_SUM or DDRR_REDUCE_MAX");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, det_h, det_w, tile_h, tile_w);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_for(p)), block(kBlock);
    if (lookup_mode == DDRR_LOOKUP_STEP) {
        if (align_corners) return fail(-1, "DDRR_LOOKUP_STEP requires align_corners=0");
        const bool sum = reduce_mode == DDRR_REDUCE_SUM;
#define LAUNCH(R, A, C) \
    hipLaunchKernelGGL((siddon_fwd_kernel<R, A, C>), grid, block, 0, st, p, out, aux, n_vox)
        if (n_vox) {
            if (aux) return fail(-1, "aux and n_vox cannot be requested together");
            if (sum) LAUNCH(REDUCE_SUM, false, true);
            else LAUNCH(REDUCE_MAX, false, true);
        } else if (aux) {
            if (sum) LAUNCH(REDUCE_SUM, true, false);
            else LAUNCH(REDUCE_MAX, true, false);
        } else {
            if (sum) LAUNCH(REDUCE_SUM, false, false);
            else LAUNCH(REDUCE_MAX, false, false);
        }
#undef LAUNCH
    } else if (lookup_mode == DDRR_LOOKUP_MID_NEAREST || lookup_mode == DDRR_LOOKUP_MID_TRILINEAR) {
        if (aux || n_vox) return fail(-1, "aux / n_vox are only produced by DDRR_LOOKUP_STEP");
        const bool sum = reduce_mode == DDRR_REDUCE_SUM;
        const bool tri = lookup_mode == DDRR_LOOKUP_MID_TRILINEAR;
#define LAUNCH(R, K) \
    hipLaunchKernelGGL((siddon_fwd_mid_kernel<R, K>), grid, block, 0, st, p, align_corners, out)
        if (sum && tri) LAUNCH(REDUCE_SUM, LOOKUP_MID_TRILINEAR);
        else if (sum) LAUNCH(REDUCE_SUM, LOOKUP_MID_NEAREST);
        else if (tri) LAUNCH(REDUCE_MAX, LOOKUP_MID_TRILINEAR);
        else LAUNCH(REDUCE_MAX, LOOKUP_MID_NEAREST);
#undef LAUNCH
    } else {
        return fail(-1, "unknown lookup_mode");
    }
    return finish("ddrr_siddon_forward");
}

int ddrr_siddon_backward_rays(const float *aux, int aux_layout, const float *grad_out,
                              const float *source, int src_n, const float *target,
                              const float *img, int B, int N, float eps, int reduce_mode,
                              float *g_source, float *g_target, float *g_img, void *stream) {
    if (!aux || !grad_out || !source || !target) return fail(-1, "null pointer");
    if (src_n != 1 && src_n != N) return fail(-1, "src_n must be 1 or N");
    if (aux_layout != DDRR_AUX_INTERLEAVED && aux_layout != DDRR_AUX_BLOCKED &&
        aux_layout != DDRR_AUX_PACKED)
        return fail(-1, "bad aux_layout");
    if (aux_layout != DDRR_AUX_INTERLEAVED && reduce_mode != DDRR_REDUCE_SUM)
        return fail(-1, "the brick kernel's records exist for reduce sum only");
    const long R = (long)B * N;
    if (R == 0) return 0;
    const dim3 grid((unsigned)((R + kBlock - 1) / kBlock)), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    if (reduce_mode == DDRR_REDUCE_SUM)
        hipLaunchKernelGGL((siddon_bwd_rays_kernel<REDUCE_SUM>), grid, block, 0, st, aux, grad_out,
                           source, src_n, target, img, R, N, eps, aux_layout, g_source, g_target,
                           g_img);
    else if (reduce_mode == DDRR_REDUCE_MAX)
        hipLaunchKernelGGL((siddon_bwd_rays_kernel<REDUCE_MAX>), grid, block, 0, st, aux, grad_out,
                           source, src_n, target, img, R, N, eps, 0, g_source, g_target, g_img);
    else
        return fail(-1, "bad reduce_mode");
    return finish("ddrr_siddon_backward_rays");
}

int ddrr_siddon_backward_volume(const float *volume, int dx, int dy, int dz, const float *source,
                                int src_n, const float *target, const float *img,
                                const float *grad_out, int B, int N, float voxel_shift, float eps,
                                int reduce_mode, int det_h, int det_w, int tile_h, int tile_w,
                                float *g_volume, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!grad_out || !g_volume) return fail(-1, "null grad_out / g_volume");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, det_h, det_w, tile_h, tile_w);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_for(p)), block(kBlock);
    if (reduce_mode == DDRR_REDUCE_SUM)
        hipLaunchKernelGGL((siddon_bwd_volume_kernel<REDUCE_SUM>), grid, block, 0, st, p, grad_out,
                           g_volume);
    else if (reduce_mode == DDRR_REDUCE_MAX)
        hipLaunchKernelGGL((siddon_bwd_volume_kernel<REDUCE_MAX>), grid, block, 0, st, p, grad_out,
                           g_volume);
    else
        return fail(-1, "bad reduce_mode");
    return finish("ddrr_siddon_backward_volume");
}

int ddrr_siddon_forward_channels(const float *volume, const unsigned char *labels, int dx, int dy,
                                 int dz, const float *source, int src_n, const float *target,
                                 const float *img, int B, int N, int C, float voxel_shift,
                                 float eps, int det_h, int det_w, int tile_h, int tile_w,
                                 float *out, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!labels || !out || C < 1) return fail(-1, "null labels/out or C < 1");
    if (B == 0 || N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C * N, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, det_h, det_w, tile_h, tile_w);
    hipLaunchKernelGGL(siddon_fwd_channels_kernel, dim3(grid_for(p)), dim3(kBlock), 0, st, p,
                       labels, C, out);
    return finish("ddrr_siddon_forward_channels");
}

int ddrr_siddon_backward_channels(const float *volume, const unsigned char *labels, int dx,
                                  int dy, int dz, const float *source, int src_n,
                                  const float *target, const float *img, const float *grad_out,
                                  int B, int N, int C, float voxel_shift, float eps, int det_h,
                                  int det_w, int tile_h, int tile_w, float *g_source,
                                  float *g_target, float *g_img, float *g_volume, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!labels || !grad_out || C < 1) return fail(-1, "null labels/grad_out or C < 1");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, det_h, det_w, tile_h, tile_w);
    hipLaunchKernelGGL(siddon_bwd_channels_kernel, dim3(grid_for(p)), dim3(kBlock), 0,
                       (hipStream_t)stream, p, labels, C, grad_out, g_source, g_target, g_img,
                       g_volume);
    return finish("ddrr_siddon_backward_channels");
}

int ddrr_siddon_segments(const float *volume, int dx, int dy, int dz, const float *source,
                         int src_n, const float *target, const float *img, int B, int N,
                         float voxel_shift, float eps, float *terms, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!terms) return fail(-1, "null terms pointer");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, 0, 0, 1, 64);
    hipLaunchKernelGGL(siddon_segments_kernel, dim3(grid_for(p)), dim3(kBlock), 0,
                       (hipStream_t)stream, p, terms);
    return finish("ddrr_siddon_segments");
}

int ddrr_siddon_segments_backward(const float *volume, int dx, int dy, int dz, const float *source,
                                  int src_n, const float *target, const float *img,
                                  const float *grad_terms, int B, int N, float voxel_shift,
                                  float eps, float *g_source, float *g_target, float *g_img,
                                  float *g_volume, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!grad_terms) return fail(-1, "null grad_terms pointer");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, 0, 0, 1, 64);
    const dim3 grid(grid_for(p)), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    if (g_volume)
        hipLaunchKernelGGL((siddon_segments_bwd_kernel<true>), grid, block, 0, st, p, grad_terms,
                           g_source, g_target, g_img, g_volume);
    else
        hipLaunchKernelGGL((siddon_segments_bwd_kernel<false>), grid, block, 0, st, p, grad_terms,
                           g_source, g_target, g_img, g_volume);
    return finish("ddrr_siddon_segments_backward");
}

int ddrr_siddon_backward_midpoint(const float *volume, int dx, int dy, int dz, const float *source,
                                  int src_n, const float *target, const float *img,
                                  const float *grad_out, int B, int N, float voxel_shift,
                                  float eps, int lookup_mode, int align_corners, float *g_source,
                                  float *g_target, float *g_img, float *g_volume, void *stream) {
    if (int rc = check_common(volume, dx, dy, dz, source, src_n, target, B, N)) return rc;
    if (!grad_out) return fail(-1, "null grad_out pointer");
    if (lookup_mode != DDRR_LOOKUP_MID_NEAREST && lookup_mode != DDRR_LOOKUP_MID_TRILINEAR)
        return fail(-1, "lookup_mode must be a midpoint lookup");
    if (B == 0 || N == 0) return 0;
    const RayArgs p = make_args(volume, dx, dy, dz, source, src_n, target, img, B, N, voxel_shift,
                                eps, 0, 0, 1, 64);
    const dim3 grid(grid_for(p)), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(LK, WV)                                                                          \
    hipLaunchKernelGGL((siddon_bwd_mid_kernel<LK, WV>), grid, block, 0, st, p, grad_out,         \
                       align_corners, g_source, g_target, g_img, g_volume)
    const bool tri = lookup_mode == DDRR_LOOKUP_MID_TRILINEAR;
    if (tri && g_volume) LAUNCH(LOOKUP_MID_TRILINEAR, true);
    else if (tri) LAUNCH(LOOKUP_MID_TRILINEAR, false);
    else if (g_volume) LAUNCH(LOOKUP_MID_NEAREST, true);
    else LAUNCH(LOOKUP_MID_NEAREST, false);
#undef LAUNCH
    return finish("ddrr_siddon_backward_midpoint");
}

}  // extern "C"
