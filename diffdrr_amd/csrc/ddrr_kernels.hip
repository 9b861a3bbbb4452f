// ddrr_kernels.hip -- gfx950 kernels + the C ABI declared in include/diffdrr_hip.h.
//
// Launch geometry (all kernels): 256-thread workgroups = 4 wavefronts, one
// detector ray per lane, one tile of 64 rays per wavefront (ddrr_common.h
// TileMap).  The grid is 1-D over (pose, tile) with the tile index fastest;
// workgroup ids are optionally re-mapped so that each of the 8 XCDs (private
// 4 MiB L2 each) works on one contiguous range of tiles.  No LDS, no MFMA:
// the path is a gather-bound line integral.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <string.h>

#include "../../include/diffdrr_hip.h"
#include "ddrr_common.h"
#include "siddon_core.h"
#include "brick_core.h"
#include "brick_walk.h"
#include "raygen_core.h"
#include "segments_core.h"
#include "tri_brick.h"
#include "slab_core.h"
#include "trilinear_core.h"

using namespace ddrr;

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;

thread_local char g_err[512] = "";

int fail(int code, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}
int fail_hip(hipError_t e, const char *where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}

struct RayArgs {
    const float *vol;
    Dims D;
    const float *source;
    int src_n;
    const float *target;
    const float *img;
    int B, N;
    float shift, eps;
    TileMap tm;
    int total_waves;
    int xcd_swizzle;
};

// Workgroup id -> logical workgroup id.  Workgroup b is dispatched to XCD
// b % 8 (observed, used for speed only): give every XCD a contiguous chunk.
__device__ __forceinline__ int logical_block(int xcd_swizzle) {
    const int bid = blockIdx.x;
    if (!xcd_swizzle) return bid;
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = bid & 7;
    return x * q + (x < r ? x : r) + (bid >> 3);
}

struct RayId {
    int b, n;     // pose, ray within pose (n < 0: padding lane)
    long r;       // b * N + n
};

__device__ __forceinline__ RayId ray_id(const RayArgs &p) {
    RayId id;
    const int wave = logical_block(p.xcd_swizzle) * kWavesPerBlock + (threadIdx.x >> 6);
    id.b = -1;
    id.n = -1;
    id.r = -1;
    if (wave < p.total_waves) {
        id.b = wave / p.tm.waves_per_pose;
        const int w = wave - id.b * p.tm.waves_per_pose;
        id.n = tile_ray(p.tm, w, threadIdx.x & 63, p.N);
        id.r = (long)id.b * p.N + id.n;
    }
    return id;
}

__device__ __forceinline__ void load_ray(const RayArgs &p, const RayId &id, float s[3],
                                         float t[3]) {
    const float *sp = p.source + ((long)id.b * p.src_n + (p.src_n == 1 ? 0 : id.n)) * 3;
    const float *tp = p.target + id.r * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        s[a] = sp[a];
        t[a] = tp[a];
    }
}

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
    const float *__restrict__ img, long R, int N, float eps, int planar,
    float *__restrict__ g_source, float *__restrict__ g_target, float *__restrict__ g_img) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    if (r >= R) return;
    const long b = r / N;
    const int n = (int)(r - b * N);
    const float *sp = source + (b * src_n + (src_n == 1 ? 0 : n)) * 3;
    const float *tp = target + r * 3;
    const float s[3] = {sp[0], sp[1], sp[2]}, t[3] = {tp[0], tp[1], tp[2]};
    float rec[SIDDON_AUX];
    if (planar) {
        // record of the brick kernel: planes I, S0x, S0z, S1x, S1z of R floats each; the y
        // components follow from sum_a S0_a = 0, sum_a S1_a = I
        const float I = aux[r], S0x = aux[R + r], S0z = aux[2 * R + r];
        const float S1x = aux[3 * R + r], S1z = aux[4 * R + r];
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

struct AtomicAdder {
    float *base;
    __device__ __forceinline__ void operator()(unsigned off, float v) const {
        unsafeAtomicAdd(base + off, v);  // global_atomic_add_f32, no return
    }
};

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

// ------------------------------------------------------- Siddon, slab march
// The fast forward path (slab_core.h).  plan[b] = {march axis (0 x, 1 y, 2 = z:
// generic walk), major (lanes along detector rows / columns)}; shear[b][strip] =
// slope of the z-epipolar lines at each 64-pixel strip.  One wave = 64 pixels
// along such a line; lanes advance one m-slab per iteration in lockstep.

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(v, o, 64);
        v = other < v ? other : v;
    }
    return v;
}

__device__ __forceinline__ float2 load_pair(const float *__restrict__ vol, unsigned boff) {
    // 4-byte aligned 8-byte fetch (global_load_dwordx2; gfx950 has unaligned access)
    float2 r;
    __builtin_memcpy(&r, reinterpret_cast<const char *>(vol) + boff, 8);
    return r;
}

struct SlabArgs {
    const float *vol;
    Dims D;
    const float *source;  // (B, 1, 3)
    const float *target;
    const float *img;
    int B, N;
    float shift, eps;
    ShearMap sm;
    const int *plan;     // (B, 2)
    const float *shear;  // (B, max_strips)
    int max_strips;
    int total_waves;
    int xcd_swizzle;
    Box box;         // sub-box of the volume this pass covers
    int accumulate;  // add to out / aux (later passes) instead of overwriting
};

template <bool AUX>
__global__ __launch_bounds__(kBlock) void siddon_fwd_slab_kernel(SlabArgs p,
                                                                 float *__restrict__ out,
                                                                 float *__restrict__ aux) {
    const int wave = logical_block(p.xcd_swizzle) * kWavesPerBlock + (threadIdx.x >> 6);
    if (wave >= p.total_waves) return;
    const int lane = threadIdx.x & 63;
    const int b = wave / p.sm.waves_per_pose;
    const int w = wave - b * p.sm.waves_per_pose;
    const int march = p.plan[2 * b], major = p.plan[2 * b + 1];
    const int strip = shear_strip_of(p.sm, major, w);
    if (strip >= shear_strips(p.sm, major)) return;  // padding wave of this pose's major
    const float sigma = p.shear[b * p.max_strips + strip];
    const int n = shear_ray(p.sm, major, w, lane, sigma);
    const long r = (long)b * p.N + (n < 0 ? 0 : n);

    float s[3], t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        s[a] = p.source[b * 3 + a];
        t[a] = p.target[r * 3 + a];
    }
    float rec[SIDDON_AUX];
    float I = 0.f;
    if (march > 1) {  // z-dominant pose: generic per-crossing walk
        if (n >= 0) I = siddon_forward_ray<REDUCE_SUM, AUX, false>(p.vol, p.D, p.box, s, t, p.shift,
                                                                   p.eps, rec, nullptr);
    } else {
        const SlabAxes ax = make_slab_axes(p.D, march);
        SlabLane L = slab_lane_init(p.D, p.box, ax, s, t, p.shift, p.eps);
        if (n < 0) {
            L.hit = false;
            L.fast = false;
            L.done = true;
        }
        // one march direction per wave (majority); the others take the generic walk
        const unsigned long long fast0 = __ballot(L.fast);
        const unsigned long long pos = __ballot(L.fast && L.dirf_m > 0.f);
        const int dirw = 2 * __popcll(pos) >= __popcll(fast0) ? 1 : -1;
        L.fast = L.fast && ((L.dirf_m > 0.f) == (dirw > 0));
        // lanes start when the wave's common slab index reaches their entry slab
        const int key = L.fast ? L.im_in * dirw : 0x7fffffff;
        const int t_start = key - wave_min_i32(key);
        const int cap = ax.Dm + 2;
        for (int it = 0; it < cap; ++it) {
            const bool active = L.fast && !L.done && it >= t_start;
            const SlabGeo g = slab_geometry(L, ax, active);
            const float2 pa = load_pair(p.vol, g.offA);
            float2 pb = make_float2(0.f, 0.f);
            if (__ballot(g.cx)) pb = load_pair(p.vol, g.offB);
            slab_consume<AUX>(L, g, pa.x, pa.y, pb.x, pb.y);
            if (!__ballot(L.fast && !L.done)) break;
        }
        if (L.fast) {
            I = L.acc;
            if (AUX) slab_aux_record(L, ax, rec);
        } else if (L.hit) {
            I = siddon_forward_ray<REDUCE_SUM, AUX, false>(p.vol, p.D, p.box, s, t, p.shift, p.eps,
                                                           rec, nullptr);
        } else if (AUX) {
#pragma unroll
            for (int k = 0; k < SIDDON_AUX; ++k) rec[k] = 0.f;
        }
    }
    if (n < 0) return;
    const float Lm = p.img ? p.img[r] : 1.f;
    // passes over disjoint sub-boxes run one after the other on the stream; the
    // lane owns its ray's outputs, so a plain read-modify-write accumulates them
    out[r] = (p.accumulate ? out[r] : 0.f) + Lm * I;
    if (AUX) {
        float4 *a4 = reinterpret_cast<float4 *>(aux + r * SIDDON_AUX);
        float4 lo = make_float4(rec[0], rec[1], rec[2], rec[3]);
        float4 hi = make_float4(rec[4], rec[5], rec[6], rec[7]);
        if (p.accumulate) {
            const float4 plo = a4[0], phi = a4[1];
            lo = make_float4(lo.x + plo.x, lo.y + plo.y, lo.z + plo.z, lo.w + plo.w);
            hi = make_float4(hi.x + phi.x, hi.y + phi.y, hi.z + phi.z, hi.w + phi.w);
        }
        a4[0] = lo;
        a4[1] = hi;
    }
}

// ------------------------------------------------- Siddon, brick-stationary
// One workgroup per 32^3 brick: stage the brick in LDS (padded layout), then trace from
// LDS the part of every ray of every pose that crosses it (brick_core.h, brick_walk.h).
// 1024 threads and ~159 KiB of LDS -> one workgroup per CU, 4 waves per SIMD.
//
// Work distribution inside the workgroup (no block-wide barriers in the hot loop):
//  * per pose the brick's 8 corners are projected onto the detector: a pixel box of
//    candidates; a unit = 64 consecutive candidates of one pose; waves pull units from one
//    LDS counter in increasing order, so the unit -> pose lookup is a forward cursor;
//  * phase A (all 64 lanes, arithmetic only): conservative slab test of the candidate
//    against the brick from the pose's affine detector model; the hits are compacted
//    (ballot + mbcnt) into the wave's private LDS queues, one queue per length class
//    (estimated number of crossings), so that a wave walks rays of similar length;
//  * phase B: as soon as a queue holds 64 hits their real rays are clipped exactly and
//    walked with every lane busy; the remainders are walked together at the end.
//    A queue entry is (pose << pix_bits) | pixel.

constexpr int kBrickThreads = 1024;
constexpr int kBrickWaves = kBrickThreads / 64;
constexpr int kPoseChunk = 32;
constexpr int kQueueCap = 128;
constexpr int kBuckets = 3;
constexpr int kBrickAuxPlanes = 5;  // I, S0x, S0z, S1x, S1z (y follows from the sums)

struct BrickArgs {
    const float *vol;
    Dims D;
    const float *source;  // (B, 1, 3)
    const float *target;  // (B, N, 3), row-major det_h x det_w grid
    const float *img;
    int B, det_h, det_w;
    float shift, eps;
    BrickLayout lay;
    unsigned aux_plane;  // elements per plane of the planar backward record (B * N)
    int pix_bits;        // queue entry = (pose << pix_bits) | pixel
    float t1, t2;        // length-class thresholds on the estimated crossing count
    int dbg;             // experiment switches (0 in production)
    int *work;           // global brick counter of this launch (zero at launch)
    const float *grad_out;  // *_VOLGRAD: dLoss/dout (B, N)
    float *g_volume;        // *_VOLGRAD: dLoss/dvolume
    int n_points;           // BRICK_TRI_*: samples per ray
    const float *amin, *amax;  // BRICK_TRI_*: device scalars (renderers.py:220-223)
};

// what a brick launch computes
constexpr int BRICK_FWD = 0;      // out
constexpr int BRICK_FWD_AUX = 1;  // out + planar backward record
constexpr int BRICK_VOLGRAD = 2;  // g_volume (the brick in LDS is the accumulator)
constexpr int BRICK_TRI_FWD = 3;      // trilinear marcher: out
constexpr int BRICK_TRI_VOLGRAD = 4;  // trilinear marcher: g_volume
constexpr int BRICK_TRI_FWD_AUX = 5;  // trilinear marcher: planar backward record (out follows from it)

inline size_t brick_lds_bytes(const BrickLayout &lay) {
    return (size_t)brick_floats(lay) * 4 + (size_t)kBrickWaves * kBuckets * kQueueCap * 4 +
           (size_t)(kPoseChunk * kRowWords + 2) * 4;
}

#if defined(__HIPCC__)
// Scatter into the LDS accumulator by absolute LDS byte address.  LDS float atomics run at
// ~0.7 lane per clock on gfx950 (measured: ds_add_f32 occupies the LDS pipe ~90 cycles per
// wave instruction, conflicts or not), integer ones 4x faster: the accumulator is int32
// fixed point, value = count / q, with q chosen per launch from a bound on the largest sum
// a voxel can receive (volgrad_prepare_kernel); q == 0 selects the float path (the bound
// does not exist, e.g. the source lies inside the volume).  Integer sums are associative:
// the fixed-point gradient is bit-reproducible.
struct LdsAbsAdd {
    float q;
    __device__ __forceinline__ void operator()(unsigned addr, float v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if (q != 0.f) {
            __hip_atomic_fetch_add((int *)(__attribute__((address_space(3))) int *)(unsigned long long)addr,
                                   __float2int_rn(v * q), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
        __hip_atomic_fetch_add((float *)(__attribute__((address_space(3))) float *)(unsigned long long)addr,
                               v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        (void)addr;
        (void)v;
#endif
    }
};
#endif

// Phase B for one queue entry: load the real ray, clip, walk; add to the image (forward)
// or scatter into the LDS accumulator (volume gradient).
// Offsets are 32-bit: the host checks 12 * B * N < 2^32.
template <int MODE>
__device__ __forceinline__ void brick_item(const BrickArgs &p, const float *brick,
                                           const BrickGeom &G, unsigned b, unsigned pix,
                                           float fixq, float *__restrict__ out,
                                           float *__restrict__ aux) {
    constexpr bool AUX = MODE == BRICK_FWD_AUX;
    const unsigned r = b * (unsigned)(p.det_h * p.det_w) + pix;
    const float *sp = p.source + b * 3u, *tp = p.target + r * 3u;
    const float s[3] = {sp[0], sp[1], sp[2]}, t[3] = {tp[0], tp[1], tp[2]};
    const float L = p.img ? p.img[r] : 1.f;
    const float base = (float)LdsAbsFetch::base_of(brick);
    if (MODE == BRICK_VOLGRAD) {
        const float w = p.grad_out[r] * L;
        if (w != 0.f) brick_scatter(LdsAbsAdd{fixq}, base, G, s, t, p.shift, p.eps, w);
        return;
    }
    if (MODE == BRICK_TRI_VOLGRAD) {
        const float a0 = p.amin[0], a1 = p.amax[0];
        const float w = p.grad_out[r] * L * ((a1 - a0) / (float)(p.n_points - 1));
        if (w != 0.f)
            tri_owner_scatter(LdsAbsAdd{fixq}, base, G.lof, G.hif, G.stridef, s, t, p.shift, p.eps,
                              p.n_points, a0, a1, w);
        return;
    }
    if (MODE == BRICK_TRI_FWD || MODE == BRICK_TRI_FWD_AUX) {
        TriGeom T;  // G.lof holds the first base cell here (set by the kernel)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            T.lo[a] = G.lof[a];
            T.stridef[a] = G.stridef[a];
        }
        const float a0 = p.amin[0], a1 = p.amax[0];
        float sumT, rec[6];
        if (!tri_brick_march<MODE == BRICK_TRI_FWD_AUX>(LdsAbsFetch{}, base, T, s, t, p.shift, p.eps,
                                                        p.n_points, a0, a1, sumT, rec))
            return;
        if (MODE == BRICK_TRI_FWD) {
            const float step = (a1 - a0) / (float)(p.n_points - 1);  // renderers.py:235
            unsafeAtomicAdd(out + r, L * step * sumT);
        } else {
            // the record alone: out = L step sumT is formed from plane 0 afterwards
            unsafeAtomicAdd(aux + r, sumT);
#pragma unroll
            for (int k = 0; k < 6; ++k) unsafeAtomicAdd(aux + (unsigned)(k + 1) * p.aux_plane + r, rec[k]);
        }
        return;
    }
    float I, rec[4];
    if (!brick_trace<AUX>(LdsAbsFetch{}, base, G, s, t, p.shift, p.eps, I, rec)) return;
    // with the record, out = L * (plane I) is formed afterwards (siddon_out_from_record_kernel)
    if (!AUX && !(p.dbg & 2)) unsafeAtomicAdd(out + r, L * I);
    if (AUX && !(p.dbg & 1)) {
        unsafeAtomicAdd(aux + r, I);
        unsafeAtomicAdd(aux + p.aux_plane + r, rec[0]);
        unsafeAtomicAdd(aux + 2u * p.aux_plane + r, rec[1]);
        unsafeAtomicAdd(aux + 3u * p.aux_plane + r, rec[2]);
        unsafeAtomicAdd(aux + 4u * p.aux_plane + r, rec[3]);
    }
}

__device__ __forceinline__ int lane_rank(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int MODE>
__global__ __launch_bounds__(kBrickThreads) void siddon_brick_kernel(
    BrickArgs p, float *__restrict__ out, float *__restrict__ aux) {
    constexpr bool AUX = MODE == BRICK_FWD_AUX;
    // TRI: bricks of 31^3 base cells + halo (the marcher's forward); the marcher's volume
    // gradient (TRI_OWNER) runs on the plain 32^3 voxel bricks, see tri_brick.h
    constexpr bool TRI = MODE == BRICK_TRI_FWD || MODE == BRICK_TRI_FWD_AUX;
    constexpr bool TRI_OWNER = MODE == BRICK_TRI_VOLGRAD;
    constexpr bool GRAD = MODE == BRICK_VOLGRAD || MODE == BRICK_TRI_VOLGRAD;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *brick = reinterpret_cast<float *>(smem_raw);
    unsigned *queue = reinterpret_cast<unsigned *>(brick + brick_floats(p.lay));
    float *rows = reinterpret_cast<float *>(queue + kBrickWaves * kBuckets * kQueueCap);
    int *counter = reinterpret_cast<int *>(rows + kPoseChunk * kRowWords);  // [0] unit, [1] brick

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const BrickGrid bg = TRI ? tri_brick_grid(p.D) : brick_grid(p.D);
    const int n_bricks = bg.nx * bg.ny * bg.nz;
    const int N = p.det_h * p.det_w;
    // wave-private: written and read by lanes of the same wave only.  LDS operations of a
    // wave execute in order; wave_fence() keeps the compiler from reordering them.
    unsigned *myq = queue + wave * kBuckets * kQueueCap;
    const unsigned pix_mask = (1u << p.pix_bits) - 1u;
    const int n_chunks = (p.B + kPoseChunk - 1) / kPoseChunk;
    const bool vec_ok = (p.D.z & 3) == 0 && (reinterpret_cast<uintptr_t>(p.vol) & 15) == 0;
    const bool vec_out = GRAD && (p.D.z & 3) == 0 && (reinterpret_cast<uintptr_t>(p.g_volume) & 15) == 0;
    // fixed-point scale of the LDS accumulator (volume-gradient modes): the largest sum a
    // voxel can receive is n_sum (contributions) * wmax (each) -- volgrad_prepare_kernel
    float fixq = 0.f;
    if (GRAD && !(p.dbg & 32)) {
        const float wmax = __uint_as_float((unsigned)p.work[1]);
        const float n_sum = reinterpret_cast<const float *>(p.work)[2];
        if (wmax > 0.f && wmax < 1e30f && n_sum > 0.f && n_sum <= 16384.f)
            fixq = 2.0e9f / (n_sum * wmax);
    }

  // Persistent workgroups: bricks are handed out by a global counter, so a CU that drew
  // light bricks (far from the sources: fewer rays cross them) simply takes more of them.
  for (;;) {
    __syncthreads();  // every wave is done with the previous brick's LDS
    if (tid == 0) counter[1] = atomicAdd(p.work, 1);
    __syncthreads();
    const int brick_id = counter[1];
    if (brick_id >= n_bricks) break;
    // `box`: the voxels staged in LDS; `cells`: the planes the candidates are clipped against
    Box box;
    BoxF cells;
    if (TRI) {
        int lo[3];
        tri_brick_lo(bg, brick_id, lo);
        const int Dn[3] = {p.D.x, p.D.y, p.D.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            box.lo[a] = lo[a];  // may be -1: staged as zeros (the zero padding)
            box.hi[a] = lo[a] + BRICK < Dn[a] ? lo[a] + BRICK : Dn[a];
            cells.lo[a] = (float)lo[a] + 0.5f;  // g = lo  <=>  plane index lo + 1/2
            cells.hi[a] = (float)(lo[a] + TRI_CELLS) + 0.5f;
        }
    } else {
        box = brick_box(p.D, bg, brick_id);
        cells = boxf(box);
        if (TRI_OWNER) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                // samples whose base cell is lo - 1 .. hi - 1 (g = c <=> plane index c + 1/2)
                cells.lo[a] = (float)(box.lo[a] - 1) + 0.5f;
                cells.hi[a] = (float)box.hi[a] + 0.5f;
            }
        }
    }
    BrickGeom G = brick_geom(box, p.lay);
    if (TRI) {
#pragma unroll
        for (int a = 0; a < 3; ++a) G.lof[a] = (float)box.lo[a];
    }
    const float nscale = (TRI || TRI_OWNER) ? (float)(p.n_points - 1) / (p.amax[0] - p.amin[0]) : 0.f;
    int qn0 = 0, qn1 = 0, qn2 = 0;  // hits waiting per length class (wave-uniform)

    for (int ch = 0; ch < n_chunks; ++ch) {
        const int b0 = ch * kPoseChunk;
        const int nb = p.B - b0 < kPoseChunk ? p.B - b0 : kPoseChunk;
        const bool last_chunk = ch == n_chunks - 1;
        if (ch > 0) __syncthreads();  // previous chunk's table no longer in use
        // The chunk's row table (one thread per pose) and, for the first chunk, the brick: a
        // thread owns 8 quads of 4 floats; all of its loads are issued before the first LDS
        // store, so a brick costs one memory round trip instead of eight.
        constexpr int kQuads = BRICK * BRICK * 8 / kBrickThreads;
        static_assert(kQuads * kBrickThreads == BRICK * BRICK * 8, "brick staging");
        const int q4 = (tid & 7) * 4, z = box.lo[2] + q4;
        float *const d0 = brick + q4;
        const bool stage_vec = ch == 0 && !GRAD && !TRI && vec_ok;
        // z and D.z are multiples of 4: a quad is wholly inside or wholly outside
        const bool in_z = z + 4 <= box.hi[2];
        if (tid < nb) {
            const PoseGrid pg = pose_grid(p.source + (long)(b0 + tid) * 3,
                                          p.target + (long)(b0 + tid) * N * 3, p.det_h, p.det_w);
            const PixBox pb = project_brick_grid(pg, p.det_h, p.det_w, cells, p.shift);
            BrickRow r = brick_row(pg, pb, cells, p.shift, p.eps, nscale);
            if (GRAD && !(p.dbg & 16)) r.perm_k = scatter_perm_k(r.w, r.count);
            *reinterpret_cast<BrickRow *>(rows + tid * kRowWords) = r;
        }
        if (tid == 0) counter[0] = 0;
        float4 q[kQuads];
        if (stage_vec) {
#pragma unroll
            for (int it = 0; it < kQuads; ++it) {
                const int row = (tid >> 3) + it * (kBrickThreads >> 3);
                const int lx = row / BRICK, ly = row - lx * BRICK;
                const int x = box.lo[0] + lx, y = box.lo[1] + ly;
                // clamped (always readable) address; what lies outside is zeroed below
                const int xc = x < p.D.x ? x : p.D.x - 1, yc = y < p.D.y ? y : p.D.y - 1;
                q[it] = *reinterpret_cast<const float4 *>(
                    p.vol + ((long)xc * p.D.y + yc) * p.D.z + (in_z ? z : 0));
            }
        }
        if (stage_vec) {
#pragma unroll
            for (int it = 0; it < kQuads; ++it) {
                const int row = (tid >> 3) + it * (kBrickThreads >> 3);
                const int lx = row / BRICK, ly = row - lx * BRICK;
                const bool in = in_z && box.lo[0] + lx < box.hi[0] && box.lo[1] + ly < box.hi[1];
                float *d = d0 + lx * p.lay.sx + ly * p.lay.sy;
                d[0] = in ? q[it].x : 0.f;
                d[1] = in ? q[it].y : 0.f;
                d[2] = in ? q[it].z : 0.f;
                d[3] = in ? q[it].w : 0.f;
            }
        } else if (ch == 0) {
            // general path (halo bricks of the trilinear marcher, unaligned volumes, and the
            // zero fill of the gradient accumulator), two quads in flight
#pragma unroll 1
            for (int h = 0; h < kQuads; h += 2) {
                float v[2][4];
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int row = (tid >> 3) + (h + it) * (kBrickThreads >> 3);
                    const int lx = row / BRICK, ly = row - lx * BRICK;
                    const int x = box.lo[0] + lx, y = box.lo[1] + ly;
                    const bool in_xy = !GRAD && x >= 0 && y >= 0 && x < box.hi[0] && y < box.hi[1];
                    const int xc = clampi(x, 0, p.D.x - 1), yc = clampi(y, 0, p.D.y - 1);
                    const float *g = p.vol + ((long)xc * p.D.y + yc) * p.D.z;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool in = in_xy && z + k >= 0 && z + k < box.hi[2];
                        v[it][k] = in ? g[z + k] : 0.f;
                    }
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int row = (tid >> 3) + (h + it) * (kBrickThreads >> 3);
                    const int lx = row / BRICK, ly = row - lx * BRICK;
                    float *d = d0 + lx * p.lay.sx + ly * p.lay.sy;
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = v[it][k];
                }
            }
        }
        __syncthreads();
        // units per pose -> inclusive prefix, held by every wave in registers (lane k: pose k)
        int incl = lane < nb ? (reinterpret_cast<const BrickRow *>(rows + lane * kRowWords)->count +
                                63) >> 6
                             : 0;
#pragma unroll
        for (int o = 1; o < kPoseChunk; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            incl += lane >= o ? up : 0;
        }
        const int units = __builtin_amdgcn_readlane(incl, kPoseChunk - 1);
        int cur = 0, cur_lo = 0, cur_hi = __builtin_amdgcn_readlane(incl, 0);
        for (;;) {
            int u = 0;
            if (lane == 0) u = atomicAdd(&counter[0], 1);
            u = uni(u);
            const bool drain = u >= units;  // no unit left in this chunk
            if (drain && !last_chunk) break;
            if (!drain) {
                while (u >= cur_hi) {  // units arrive in increasing order: forward cursor
                    ++cur;
                    cur_lo = cur_hi;
                    cur_hi = __builtin_amdgcn_readlane(incl, uni(cur));
                }
                const BrickRow r = *reinterpret_cast<const BrickRow *>(rows + cur * kRowWords);
                int local = (u - cur_lo) * 64 + lane;
                const bool valid = local < uni(r.count);
                if (GRAD && valid)
                    local = scatter_perm(local, uni(r.perm_k), uni(r.count),
                                         1.0f / (float)uni(r.count));
                int pix = 0;
                float n_est = 0.f;
                const bool hit = valid && brick_candidate(r, local, p.det_w, pix, n_est);
                // With the backward record (5 atomics per hit instead of 1) the length class
                // of a hit is that of the longest hit among its 8 neighbours in candidate
                // order (consecutive pixels of a detector row): a batch is then made of runs
                // of >= 8 adjacent pixels and its atomics touch few cache lines -- their cost
                // is per line, not per lane (measured: 3.55 -> 2.74 ms; without the record
                // the per-lane classes win, 1.87 vs 2.01 ms).
                float n_grp = hit ? n_est : 0.f;
                if ((AUX || MODE == BRICK_TRI_FWD_AUX) && !(p.dbg & 8)) {
                    n_grp = fmaxf(n_grp, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, n_grp), 0xB1, 0xf, 0xf, true)));   // lane ^ 1
                    n_grp = fmaxf(n_grp, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, n_grp), 0x4E, 0xf, 0xf, true)));   // lane ^ 2
                    n_grp = fmaxf(n_grp, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, n_grp), 0x141, 0xf, 0xf, true)));  // 7 - lane
                }
                const bool c0 = n_grp < p.t1, c1 = !c0 && n_grp < p.t2;
                const unsigned long long m0 = __ballot(hit && c0);
                const unsigned long long m1 = __ballot(hit && c1);
                const unsigned long long m2 = __ballot(hit && !c0 && !c1);
                if (hit) {
                    const int r0 = lane_rank(m0), r1 = lane_rank(m1), r2 = lane_rank(m2);
                    const int slot = c0 ? qn0 + r0 : (c1 ? kQueueCap + qn1 + r1
                                                         : 2 * kQueueCap + qn2 + r2);
                    myq[slot] = ((unsigned)(b0 + cur) << p.pix_bits) | (unsigned)pix;
                }
                qn0 = uni(qn0 + (int)__popcll(m0));
                qn1 = uni(qn1 + (int)__popcll(m1));
                qn2 = uni(qn2 + (int)__popcll(m2));
                wave_fence();
            }
            // walk every full batch of 64 hits of one class; when draining, what is left
            // of all classes together (longest first), 64 at a time
            for (;;) {
                int k = -1, n = 0;
                if (qn0 >= 64) k = 0, n = 64;
                else if (qn1 >= 64) k = 1, n = 64;
                else if (qn2 >= 64) k = 2, n = 64;
                unsigned e = 0;
                if (k >= 0) {
                    const int base = (k == 0 ? qn0 : (k == 1 ? qn1 : qn2)) - 64;
                    qn0 -= k == 0 ? 64 : 0;
                    qn1 -= k == 1 ? 64 : 0;
                    qn2 -= k == 2 ? 64 : 0;
                    e = myq[k * kQueueCap + base + lane];
                } else if (drain && qn0 + qn1 + qn2 > 0) {
                    // virtual queue [class 2 | class 1 | class 0], taken from the front
                    const int tot = qn0 + qn1 + qn2;
                    n = tot < 64 ? tot : 64;
                    const int i2 = lane, i1 = lane - qn2, i0 = lane - qn2 - qn1;
                    if (lane < n)
                        e = i2 < qn2 ? myq[2 * kQueueCap + qn2 - 1 - i2]
                                     : (i1 < qn1 ? myq[kQueueCap + qn1 - 1 - i1]
                                                 : myq[qn0 - 1 - i0]);
                    // consumed from the tops of the stacks
                    const int t2 = qn2 < n ? qn2 : n;
                    const int t1 = qn1 < n - t2 ? qn1 : n - t2;
                    qn2 -= t2;
                    qn1 -= t1;
                    qn0 -= n - t2 - t1;
                } else {
                    break;
                }
                if (lane < n)
                    brick_item<MODE>(p, brick, G, e >> p.pix_bits, e & pix_mask, fixq, out, aux);
                wave_fence();
            }
            if (drain) break;
        }
    }
    if (GRAD) {
        // every ray of every pose has been scattered into the LDS accumulator, and the brick
        // owns its voxels (Siddon bricks, and the marcher's owner bricks): the gradient is
        // complete and is stored, 16 bytes per thread
        __syncthreads();
        for (int row = tid >> 3; row < BRICK * BRICK; row += kBrickThreads >> 3) {
            const int lx = row / BRICK, ly = row - lx * BRICK, q4 = (tid & 7) * 4;
            const int x = box.lo[0] + lx, y = box.lo[1] + ly, z = box.lo[2] + q4;
            if (x < box.hi[0] && y < box.hi[1]) {
                const float *src = brick + lx * p.lay.sx + ly * p.lay.sy + q4;
                float *g = p.g_volume + ((long)x * p.D.y + y) * p.D.z + z;
                float val[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    val[k] = fixq != 0.f ? (float)__float_as_int(src[k]) / fixq : src[k];
                if (vec_out && z + 4 <= box.hi[2]) {
                    *reinterpret_cast<float4 *>(g) = make_float4(val[0], val[1], val[2], val[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (z + k < box.hi[2]) g[k] = val[k];
                }
            }
        }
    }
  }
}

// mask_to_channels (renderers.py:77-89): the ray owns column out[b, :, n]; runs
// of one label are flushed with a plain read-modify-write (siddon_channels_ray).
struct ColumnFlush {
    float *col;
    long stride;
    int C;
    float L;
    __device__ __forceinline__ void operator()(int label, float run) const {
        if (label < C) col[label * stride] += L * run;
    }
};

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

// ------------------------------------------- volume-gradient fixed-point bound
// work[1] = bits of max over rays of the largest single contribution a ray can make to a
// voxel; work[2] = float: bound on the number of such contributions a voxel can receive in
// this launch (sum over poses of the rays that can cross one voxel); see LdsAbsAdd.
//   Siddon:    |g| L dalpha,  dalpha |d| <= sqrt(3)           ->  sqrt(3) |g| L / |d|
//   trilinear: |g| L step per sample, at most 2 sqrt(3) / (step |d|) + 1 samples of a ray
//              touch one voxel                                 ->  |g| L (2 sqrt(3) / |d| + step)
// Rays of a pose through one voxel: those whose pixel lies in the voxel's (8-cell's)
// shadow, at most (extent * |t - s| / rho_min / e_min + 2)^2 with rho_min the distance from
// the source to the volume (0: no bound, the float path is taken).
__global__ __launch_bounds__(kBlock) void volgrad_prepare_kernel(
    int tri, const float *__restrict__ source, const float *__restrict__ target,
    const float *__restrict__ img, const float *__restrict__ grad_out, int N, int det_w, Dims D,
    float shift, float eps, int n_points, const float *__restrict__ amin,
    const float *__restrict__ amax, int *__restrict__ work) {
    const int b = blockIdx.y;
    const float s[3] = {source[b * 3], source[b * 3 + 1], source[b * 3 + 2]};
    const float step = tri ? (amax[0] - amin[0]) / (float)(n_points - 1) : 0.f;
    float wmax = 0.f;
    for (int n = blockIdx.x * kBlock + threadIdx.x; n < N; n += gridDim.x * kBlock) {
        const long r = (long)b * N + n;
        const float dx = target[r * 3] - s[0] + eps, dy = target[r * 3 + 1] - s[1] + eps;
        const float dz = target[r * 3 + 2] - s[2] + eps;
        const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
        const float L = img ? img[r] : 1.f;
        const float c = tri ? (3.4642f / dn + step) : 1.7321f / dn;
        wmax = fmaxf(wmax, fabsf(grad_out[r]) * L * c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned *>(work) + 1, __float_as_uint(wmax));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // distance from the source to the volume box (voxel coordinates, planes at k - shift)
        const float lo = -shift, hi[3] = {(float)D.x - shift, (float)D.y - shift, (float)D.z - shift};
        float rho2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float gap = fmaxf(fmaxf(lo - s[a], s[a] - hi[a]), 0.f);
            rho2 += gap * gap;
        }
        const float *t0 = target + (long)b * N * 3;
        float e_i = 0.f, e_j = 0.f, dst = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float di = t0[(long)det_w * 3 + a] - t0[a], dj = t0[3 + a] - t0[a];
            const float dt = t0[a] - s[a];
            e_i += di * di;
            e_j += dj * dj;
            dst += dt * dt;
        }
        // the detector point farthest from the source bounds |t - s| (corner pixels)
        const float *tc = t0 + (long)(N - 1) * 3;
        float dst2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) dst2 += (tc[a] - s[a]) * (tc[a] - s[a]);
        const float reach = sqrtf(fmaxf(dst, dst2)), e_min = sqrtf(fminf(e_i, e_j));
        const float extent = tri ? 3.4642f : 1.7321f;
        float R = INFINITY;
        if (rho2 > 1.f && e_min > 0.f) {
            const float side = extent * reach / (sqrtf(rho2) * e_min) + 2.f;
            R = side * side;
        }
        atomicAdd(reinterpret_cast<float *>(work) + 2, R);
    }
}

// ------------------------------------------------ fused ray generation (DRR case)

// source_v (B,3), target_v (B,N,3), img (B,N) from one world pose per DRR (raygen_core.h)
__global__ __launch_bounds__(kBlock) void raygen_fwd_kernel(
    const float *__restrict__ Mw, const float *__restrict__ Ainv, const float *__restrict__ P,
    int N, float *__restrict__ source_v, float *__restrict__ target_v, float *__restrict__ img) {
    const int b = blockIdx.y, n = blockIdx.x * kBlock + threadIdx.x;
    const float *M = Mw + (long)b * 12;
    if (n == 0) {
        const float sw[3] = {M[3], M[7], M[11]};
        float sv[3];
        apply34(Ainv, sw, sv);
        source_v[b * 3 + 0] = sv[0];
        source_v[b * 3 + 1] = sv[1];
        source_v[b * 3 + 2] = sv[2];
    }
    if (n >= N) return;
    const float Pn[3] = {P[n * 3], P[n * 3 + 1], P[n * 3 + 2]};
    const RayGenOut o = raygen_ray(M, Ainv, Pn);
    const long r = (long)b * N + n;
    target_v[r * 3 + 0] = o.tv[0];
    target_v[r * 3 + 1] = o.tv[1];
    target_v[r * 3 + 2] = o.tv[2];
    img[r] = o.L;
}

// dLoss/dMw (B,3,4) from the forward's backward record: the renderer's per-ray endpoint
// gradients (siddon_backward_ray) are chained through the ray generation and reduced per
// pose inside the kernel; no per-ray gradient tensor is written.  A block covers
// kPoseRaysPerBlock rays of one pose and adds its 12 partial sums with atomics.
constexpr int kPoseRaysPerBlock = 4096;

__global__ __launch_bounds__(kBlock) void siddon_bwd_pose_kernel(
    const float *__restrict__ aux, int planar, const float *__restrict__ grad_out,
    const float *__restrict__ source_v, const float *__restrict__ target_v,
    const float *__restrict__ img, const float *__restrict__ Mw, const float *__restrict__ Ainv,
    const float *__restrict__ P, int B, int N, float eps, int with_img_path,
    float *__restrict__ gMw) {
    __shared__ float red[kWavesPerBlock][12];
    const int b = blockIdx.y;
    const long R = (long)B * N;
    const float *M = Mw + (long)b * 12;
    const float s[3] = {source_v[b * 3], source_v[b * 3 + 1], source_v[b * 3 + 2]};
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    const int n_end = min(N, (int)(blockIdx.x + 1) * kPoseRaysPerBlock);
    for (int n = blockIdx.x * kPoseRaysPerBlock + threadIdx.x; n < n_end; n += kBlock) {
        const long r = (long)b * N + n;
        float rec[SIDDON_AUX];
        if (planar) {
            const float I = aux[r], S0x = aux[R + r], S0z = aux[2 * R + r];
            const float S1x = aux[3 * R + r], S1z = aux[4 * R + r];
            rec[0] = I, rec[1] = S0x, rec[2] = -(S0x + S0z), rec[3] = S0z;
            rec[4] = S1x, rec[5] = I - (S1x + S1z), rec[6] = S1z, rec[7] = 0.f;
        } else {
            const float4 *a4 = reinterpret_cast<const float4 *>(aux + r * SIDDON_AUX);
            const float4 lo = a4[0], hi = a4[1];
            rec[0] = lo.x, rec[1] = lo.y, rec[2] = lo.z, rec[3] = lo.w;
            rec[4] = hi.x, rec[5] = hi.y, rec[6] = hi.z, rec[7] = hi.w;
        }
        const float t[3] = {target_v[r * 3], target_v[r * 3 + 1], target_v[r * 3 + 2]};
        const float Pn[3] = {P[n * 3], P[n * 3 + 1], P[n * 3 + 2]};
        const float g = grad_out[r], L = img[r];
        float gs[3], gt[3];
        siddon_backward_ray<REDUCE_SUM>(rec, s, t, eps, g * L, gs, gt);
        raygen_ray_adjoint(M, Ainv, Pn, gt, gs, with_img_path ? g * rec[0] : 0.f, L, acc);
    }
    // 12 sums over the block: wave butterflies, then the 4 waves through LDS
#pragma unroll
    for (int k = 0; k < 12; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) v += red[w][threadIdx.x];
        unsafeAtomicAdd(gMw + (long)b * 12 + threadIdx.x, v);
    }
}

// -------------------------------------------------- Euler pose -> world matrix
__global__ __launch_bounds__(kBlock) void pose_euler_fwd_kernel(
    const float *__restrict__ rot, const float *__restrict__ xyz, int a0, int a1, int a2,
    const float *__restrict__ Ro, int B, float *__restrict__ Mw) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const float th[3] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2]};
    const float t[3] = {xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
    const int axes[3] = {a0, a1, a2};
    float M[12];
    pose_euler_forward(th, t, axes, Ro, M);
#pragma unroll
    for (int k = 0; k < 12; ++k) Mw[b * 12 + k] = M[k];
}

__global__ __launch_bounds__(kBlock) void pose_euler_bwd_kernel(
    const float *__restrict__ rot, const float *__restrict__ xyz, int a0, int a1, int a2,
    const float *__restrict__ Ro, const float *__restrict__ gMw, int B, float *__restrict__ g_rot,
    float *__restrict__ g_xyz) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const float th[3] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2]};
    const float t[3] = {xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
    const int axes[3] = {a0, a1, a2};
    float g[12], gt[3], gx[3];
#pragma unroll
    for (int k = 0; k < 12; ++k) g[k] = gMw[b * 12 + k];
    pose_euler_backward(th, t, axes, Ro, g, gt, gx);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g_rot[b * 3 + k] = gt[k];
        g_xyz[b * 3 + k] = gx[k];
    }
}

// ------------------------------------------------- fused NCC (sweep / registration)
// NormalizedCrossCorrelation2d with patch_size = None (reference metrics.py:21-44):
// ncc_b = mean(z1 * z2), z = (x - mean) / sqrt(var + eps), one value per image pair.
// One workgroup per pair, two passes over the pair (means, then centred moments: no
// cancellation); stats[b] = {mu1, s1, mu2, s2, ncc}.  x1 may be one image shared by the
// whole batch (x1_stride = 0).
constexpr int kNccThreads = 1024;

__device__ __forceinline__ float block_sum(float v, float *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // red may still be read by the previous call
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kNccThreads / 64; ++w) t += red[w];
    return t;
}

__global__ __launch_bounds__(kNccThreads) void ncc_fwd_kernel(
    const float *__restrict__ x1, long x1_stride, const float *__restrict__ x2, int N, float eps,
    float *__restrict__ out, float *__restrict__ stats) {
    __shared__ float red[kNccThreads / 64];
    const int b = blockIdx.x;
    const float *p1 = x1 + b * x1_stride, *p2 = x2 + (long)b * N;
    float a1 = 0.f, a2 = 0.f;
    for (int n = threadIdx.x; n < N; n += kNccThreads) {
        a1 += p1[n];
        a2 += p2[n];
    }
    const float inv_n = 1.0f / (float)N;
    const float mu1 = block_sum(a1, red) * inv_n, mu2 = block_sum(a2, red) * inv_n;
    float v1 = 0.f, v2 = 0.f, c12 = 0.f;
    for (int n = threadIdx.x; n < N; n += kNccThreads) {
        const float d1 = p1[n] - mu1, d2 = p2[n] - mu2;
        v1 = fmaf(d1, d1, v1);
        v2 = fmaf(d2, d2, v2);
        c12 = fmaf(d1, d2, c12);
    }
    const float s1 = sqrtf(block_sum(v1, red) * inv_n + eps);
    const float s2 = sqrtf(block_sum(v2, red) * inv_n + eps);
    const float ncc = block_sum(c12, red) * inv_n / (s1 * s2);
    if (threadIdx.x == 0) {
        out[b] = ncc;
        stats[b * 5 + 0] = mu1;
        stats[b * 5 + 1] = s1;
        stats[b * 5 + 2] = mu2;
        stats[b * 5 + 3] = s2;
        stats[b * 5 + 4] = ncc;
    }
}

// d ncc / d x2[n] = (z1[n] - z2[n] ncc) / (N s2), and symmetrically for x1 (per pair; a
// shared x1 gets no gradient from this kernel).
__global__ __launch_bounds__(kBlock) void ncc_bwd_kernel(
    const float *__restrict__ x1, long x1_stride, const float *__restrict__ x2,
    const float *__restrict__ stats, const float *__restrict__ g_out, int N,
    float *__restrict__ g_x1, float *__restrict__ g_x2) {
    const int b = blockIdx.y, n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float mu1 = stats[b * 5], s1 = stats[b * 5 + 1], mu2 = stats[b * 5 + 2];
    const float s2 = stats[b * 5 + 3], ncc = stats[b * 5 + 4];
    const float z1 = (x1[b * x1_stride + n] - mu1) / s1, z2 = (x2[(long)b * N + n] - mu2) / s2;
    const float g = g_out[b] / (float)N;
    if (g_x2) g_x2[(long)b * N + n] = g * (z1 - z2 * ncc) / s2;
    if (g_x1) g_x1[(long)b * N + n] = g * (z2 - z1 * ncc) / s1;
}

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

struct NoAdd {
    __device__ __forceinline__ void operator()(unsigned, float) const {}
};

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

// out = L * I from plane 0 of the Siddon planar record (the record launch leaves `out` alone:
// one atomic less per ray and brick).
__global__ __launch_bounds__(kBlock) void siddon_out_from_record_kernel(
    const float *__restrict__ aux, const float *__restrict__ img, long R, float *__restrict__ out) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    if (r < R) out[r] = (img ? img[r] : 1.f) * aux[r];
}

// out = L * step * sumT from plane 0 of the marcher's planar record (the record launch
// does not touch `out`).
__global__ __launch_bounds__(kBlock) void tri_out_from_record_kernel(
    const float *__restrict__ aux, const float *__restrict__ img, long R, int n_points,
    const float *__restrict__ amin, const float *__restrict__ amax, float *__restrict__ out) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    if (r >= R) return;
    const float step = (amax[0] - amin[0]) / (float)(n_points - 1);  // renderers.py:235
    out[r] = (img ? img[r] : 1.f) * step * aux[r];
}

// Ray / range gradients of the march from the planar record of ddrr_trilinear_forward_bricks
// (planes sumT, sum dT_xyz, sum alpha dT_xyz of R floats each): elementwise.
__global__ __launch_bounds__(kBlock) void trilinear_bwd_record_kernel(
    const float *__restrict__ aux, const float *__restrict__ grad_out,
    const float *__restrict__ source, const float *__restrict__ target,
    const float *__restrict__ img, long R, int N, float eps, int n_points,
    const float *__restrict__ amin, const float *__restrict__ amax, float *__restrict__ g_source,
    float *__restrict__ g_target, float *__restrict__ g_img, float *__restrict__ g_alpha) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    if (r >= R) return;
    const long b = r / N;
    const float *sp = source + b * 3, *tp = target + r * 3;
    const float s[3] = {sp[0], sp[1], sp[2]}, t[3] = {tp[0], tp[1], tp[2]};
    const float A[3] = {aux[R + r], aux[2 * R + r], aux[3 * R + r]};
    const float Bv[3] = {aux[4 * R + r], aux[5 * R + r], aux[6 * R + r]};
    const float g = grad_out[r], L = img ? img[r] : 1.f;
    const float a0 = amin[0], a1 = amax[0];
    const MarchGrad m = trilinear_backward_from_record(aux[r], A, Bv, s, t, eps, n_points, a0, a1,
                                                       g * L);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (g_source) g_source[r * 3 + a] = m.gs[a];
        if (g_target) g_target[r * 3 + a] = m.gt[a];
    }
    if (g_img) g_img[r] = g * m.sumT * ((a1 - a0) / (float)(n_points - 1));
    if (g_alpha) {
        g_alpha[r * 2 + 0] = m.g_amin;
        g_alpha[r * 2 + 1] = m.g_amax;
    }
}

// ------------------------------------------------------------------ host side

int g_xcd_swizzle = 1;
// The slab kernel is fastest with the natural round-robin placement: all XCDs then
// sweep the same poses at the same time, which the shared Infinity Cache likes
// (profiles/r01/sweep_v2_slab_512.txt: 4.8 ms vs 6.8 ms on the bench workload).
int g_xcd_swizzle_slab = 0;
// LDS layout of a brick (floats): rows padded 32 -> 33, planes 32*33 -> 1057, so that
// x-, y- and z-neighbours all fall in different banks.
BrickLayout g_brick_layout = {33, 32 * 33 + 1};
// length classes of brick hits (estimated plane crossings inside the brick)
float g_brick_t1 = 18.f, g_brick_t2 = 40.f;
float g_tri_t1 = 10.f, g_tri_t2 = 22.f;  // same for the marcher, in samples per brick
int g_brick_dbg = 0;

int check_common(const float *volume, int dx, int dy, int dz, const float *source, int src_n,
                 const float *target, int B, int N) {
    if (!volume || !source || !target) return fail(-1, "null volume/source/target pointer");
    if (dx < 1 || dy < 1 || dz < 1) return fail(-1, "volume dims must be >= 1");
    if ((long)dx * dy * dz > (1L << 30)) return fail(-1, "volume larger than 2^30 voxels");
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (src_n != 1 && src_n != N) return fail(-1, "src_n must be 1 or N");
    return 0;
}

RayArgs make_args(const float *volume, int dx, int dy, int dz, const float *source, int src_n,
                  const float *target, const float *img, int B, int N, float shift, float eps,
                  int det_h, int det_w, int tile_h, int tile_w) {
    RayArgs p;
    p.vol = volume;
    p.D = Dims{dx, dy, dz};
    p.source = source;
    p.src_n = src_n;
    p.target = target;
    p.img = img;
    p.B = B;
    p.N = N;
    p.shift = shift;
    p.eps = eps;
    p.tm = make_tilemap(N, det_h, det_w, tile_h, tile_w);
    p.total_waves = B * p.tm.waves_per_pose;
    p.xcd_swizzle = g_xcd_swizzle;
    return p;
}

inline int grid_for(const RayArgs &p) {
    return (p.total_waves + kWavesPerBlock - 1) / kWavesPerBlock;
}

int finish(const char *where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, where);
    return 0;
}

}  // namespace

extern "C" {

int ddrr_abi_version(void) { return DDRR_ABI_VERSION; }
const char *ddrr_last_error(void) { return g_err; }

// Experiment knob (not part of the renderer contract): 0/1 XCD-contiguous
// workgroup mapping.
int ddrr_set_xcd_swizzle(int on) {
    int old = g_xcd_swizzle;
    g_xcd_swizzle = on ? 1 : 0;
    return old;
}
// Experiment knob: LDS strides (floats) of a staged brick; sy >= 32, sx >= 32 * sy.
int ddrr_set_brick_layout(int sy, int sx) {
    if (sy < BRICK || sx < BRICK * sy) return -1;
    BrickLayout lay = {sy, sx};
    if (brick_lds_bytes(lay) > 160 * 1024) return -1;
    g_brick_layout = lay;
    return 0;
}
// Experiment switches of the brick kernels (0 in production; results are wrong with 1, 2):
//   1 skip the record's atomics, 2 skip the image atomic, 8 per-lane length classes also with
//   the record (no groups of 8 pixels), 16 no scatter permutation, 32 float LDS accumulation.
// profiles/r01/exp_record_cost.txt holds the decomposition these gave for the record.
int ddrr_set_brick_debug(int flags) {
    g_brick_dbg = flags;
    return 0;
}
int ddrr_set_brick_classes(float t1, float t2) {
    g_brick_t1 = t1;
    g_brick_t2 = t2;
    return 0;
}
int ddrr_set_xcd_swizzle_slab(int on) {
    int old = g_xcd_swizzle_slab;
    g_xcd_swizzle_slab = on ? 1 : 0;
    return old;
}

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

int ddrr_siddon_forward_slab(const float *volume, int dx, int dy, int dz, const float *source,
                             const float *target, const float *img, int B, int det_h, int det_w,
                             float voxel_shift, float eps, const int *plan, const float *shear,
                             int max_strips, const int *box, int accumulate, float *out,
                             float *aux, void *stream) {
    const int N = det_h * det_w;
    if (int rc = check_common(volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (!out || !plan || !shear) return fail(-1, "null out / plan / shear pointer");
    if (det_h < 1 || det_w < 1) return fail(-1, "detector must be at least 1x1");
    if (B == 0) return 0;
    SlabArgs p;
    p.vol = volume;
    p.D = Dims{dx, dy, dz};
    p.source = source;
    p.target = target;
    p.img = img;
    p.B = B;
    p.N = N;
    p.shift = voxel_shift;
    p.eps = eps;
    p.sm = make_shearmap(det_h, det_w);
    p.plan = plan;
    p.shear = shear;
    p.max_strips = max_strips;
    const int need = (det_h > det_w ? det_h : det_w);
    if (max_strips < (need + 63) / 64) return fail(-1, "shear table has too few strips");
    p.total_waves = B * p.sm.waves_per_pose;
    p.xcd_swizzle = g_xcd_swizzle_slab;
    p.box = full_box(p.D);
    if (box) {
        const int Dn[3] = {dx, dy, dz};
        for (int a = 0; a < 3; ++a) {
            p.box.lo[a] = box[a];
            p.box.hi[a] = box[3 + a];
            if (box[a] < 0 || box[3 + a] > Dn[a] || box[a] >= box[3 + a])
                return fail(-1, "box must satisfy 0 <= lo < hi <= dims");
        }
    }
    p.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((p.total_waves + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlock);
    if (aux)
        hipLaunchKernelGGL((siddon_fwd_slab_kernel<true>), grid, block, 0, st, p, out, aux);
    else
        hipLaunchKernelGGL((siddon_fwd_slab_kernel<false>), grid, block, 0, st, p, out, aux);
    return finish("ddrr_siddon_forward_slab");
}

namespace {
int launch_bricks(int mode, const float *volume, int dx, int dy, int dz, const float *source,
                  const float *target, const float *img, const float *grad_out, int B, int det_h,
                  int det_w, float voxel_shift, float eps, float *out, float *aux,
                  float *g_volume, hipStream_t st, const char *who, int n_points = 0,
                  const float *amin = nullptr, const float *amax = nullptr) {
    const int N = det_h * det_w;
    BrickArgs p;
    p.vol = volume;
    p.D = Dims{dx, dy, dz};
    p.source = source;
    p.target = target;
    p.img = img;
    p.B = B;
    p.det_h = det_h;
    p.det_w = det_w;
    p.shift = voxel_shift;
    p.eps = eps;
    p.lay = g_brick_layout;
    if ((long)B * N * 12 >= (1L << 32))
        return fail(-1, "B * N too large for one brick launch (12 B N must stay below 2^32): "
                        "split the pose batch");
    p.aux_plane = (unsigned)((long)B * N);
    p.pix_bits = 1;
    while ((1L << p.pix_bits) < N) ++p.pix_bits;
    if (((long)B << p.pix_bits) > (1L << 32))
        return fail(-1, "B * 2^ceil(log2 N) exceeds 2^32: split the pose batch");
    p.t1 = g_brick_t1;
    p.t2 = g_brick_t2;
    p.dbg = g_brick_dbg;
    p.grad_out = grad_out;
    p.g_volume = g_volume;
    p.n_points = n_points;
    p.amin = amin;
    p.amax = amax;
    if (mode == BRICK_TRI_FWD || mode == BRICK_TRI_VOLGRAD || mode == BRICK_TRI_FWD_AUX) {
        p.t1 = g_tri_t1;
        p.t2 = g_tri_t2;
    }
    const size_t lds = brick_lds_bytes(p.lay);
    hipError_t e;
    static bool attr_set = false;  // raise the dynamic-LDS limit once per process
    if (!attr_set) {
        const void *fns[6] = {reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_TRI_FWD_AUX>),
                              reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_FWD>),
                              reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_FWD_AUX>),
                              reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_VOLGRAD>),
                              reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_TRI_FWD>),
                              reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_TRI_VOLGRAD>)};
        for (const void *fn : fns)
            if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024)) != hipSuccess)
                return fail_hip(e, "hipFuncSetAttribute");
        attr_set = true;
    }
    // one brick counter per launch, from a small per-device ring (launches in flight on
    // different streams must not share one); zeroed on the launch's stream
    constexpr int kRing = 64, kMaxDev = 64;
    static int *ring[kMaxDev] = {nullptr};
    static int n_cu[kMaxDev] = {0};
    static unsigned slot[kMaxDev] = {0};
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return fail_hip(e, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDev) return fail(-1, "device index out of range");
    if (!ring[dev]) {
        if ((e = hipMalloc(reinterpret_cast<void **>(&ring[dev]), kRing * 4 * sizeof(int))) != hipSuccess)
            return fail_hip(e, "hipMalloc(brick counters)");
        if ((e = hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev)) !=
            hipSuccess)
            return fail_hip(e, "hipDeviceGetAttribute");
    }
    p.work = ring[dev] + 4 * (slot[dev]++ % kRing);  // {brick counter, wmax bits, n_sum, -}
    if ((e = hipMemsetAsync(p.work, 0, 4 * sizeof(int), st)) != hipSuccess)
        return fail_hip(e, "hipMemsetAsync");
    if (mode == BRICK_VOLGRAD || mode == BRICK_TRI_VOLGRAD) {
        const int tri = mode == BRICK_TRI_VOLGRAD;
        int bx = (N + kBlock - 1) / kBlock;
        bx = bx > 64 ? 64 : bx;
        hipLaunchKernelGGL(volgrad_prepare_kernel, dim3(bx, B), dim3(kBlock), 0, st, tri, source,
                           target, img, grad_out, N, det_w, p.D, voxel_shift, eps, n_points, amin,
                           amax, p.work);
    }
    const BrickGrid bg = (mode == BRICK_TRI_FWD || mode == BRICK_TRI_FWD_AUX) ? tri_brick_grid(p.D)
                                                                              : brick_grid(p.D);
    const int n_bricks = bg.nx * bg.ny * bg.nz;
    const dim3 grid(n_bricks < n_cu[dev] ? n_bricks : n_cu[dev]), block(kBrickThreads);
    if (mode == BRICK_TRI_FWD)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_TRI_FWD>, grid, block, lds, st, p, out, aux);
    else if (mode == BRICK_TRI_FWD_AUX)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_TRI_FWD_AUX>, grid, block, lds, st, p, out,
                           aux);
    else if (mode == BRICK_TRI_VOLGRAD)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_TRI_VOLGRAD>, grid, block, lds, st, p, out,
                           aux);
    else if (mode == BRICK_FWD)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_FWD>, grid, block, lds, st, p, out, aux);
    else if (mode == BRICK_FWD_AUX)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_FWD_AUX>, grid, block, lds, st, p, out, aux);
    else
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_VOLGRAD>, grid, block, lds, st, p, out, aux);
    return finish(who);
}
}  // namespace

int ddrr_siddon_forward_bricks(const float *volume, int dx, int dy, int dz, const float *source,
                               const float *target, const float *img, int B, int det_h,
                               int det_w, float voxel_shift, float eps, float *out, float *aux,
                               void *stream) {
    const int N = det_h * det_w;
    if (int rc = check_common(volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (!out) return fail(-1, "null out pointer");
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const long R = (long)B * N;
    hipError_t e = hipMemsetAsync(aux ? aux : out, 0,
                                  sizeof(float) * (size_t)R * (aux ? kBrickAuxPlanes : 1), st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    if (int rc = launch_bricks(aux ? BRICK_FWD_AUX : BRICK_FWD, volume, dx, dy, dz, source, target,
                               img, nullptr, B, det_h, det_w, voxel_shift, eps, out, aux, nullptr,
                               st, "ddrr_siddon_forward_bricks"))
        return rc;
    if (!aux) return 0;
    hipLaunchKernelGGL(siddon_out_from_record_kernel, dim3((unsigned)((R + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, st, aux, img, R, out);
    return finish("ddrr_siddon_forward_bricks");
}

int ddrr_siddon_backward_volume_bricks(int dx, int dy, int dz, const float *source,
                                       const float *target, const float *img,
                                       const float *grad_out, int B, int det_h, int det_w,
                                       float voxel_shift, float eps, float *g_volume,
                                       void *stream) {
    const int N = det_h * det_w;
    if (!g_volume || !grad_out) return fail(-1, "null grad_out / g_volume");
    if (int rc = check_common(g_volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {  // nothing contributes: the gradient is zero
        hipError_t e = hipMemsetAsync(g_volume, 0, sizeof(float) * (size_t)dx * dy * dz, st);
        return e == hipSuccess ? 0 : fail_hip(e, "hipMemsetAsync");
    }
    return launch_bricks(BRICK_VOLGRAD, nullptr, dx, dy, dz, source, target, img, grad_out, B,
                         det_h, det_w, voxel_shift, eps, nullptr, nullptr, g_volume, st,
                         "ddrr_siddon_backward_volume_bricks");
}

int ddrr_trilinear_forward_bricks(const float *volume, int dx, int dy, int dz,
                                  const float *source, const float *target, const float *img,
                                  int B, int det_h, int det_w, float voxel_shift, float eps,
                                  int n_points, const float *alphamin, const float *alphamax,
                                  float *out, float *aux, void *stream) {
    const int N = det_h * det_w;
    if (int rc = check_common(volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (!out || !alphamin || !alphamax) return fail(-1, "null out / alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const long R = (long)B * N;
    hipError_t e = hipMemsetAsync(aux ? aux : out, 0,
                                  sizeof(float) * (size_t)R * (aux ? DDRR_TRI_AUX_PLANES : 1), st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    if (!aux)
        return launch_bricks(BRICK_TRI_FWD, volume, dx, dy, dz, source, target, img, nullptr, B,
                             det_h, det_w, voxel_shift, eps, out, nullptr, nullptr, st,
                             "ddrr_trilinear_forward_bricks", n_points, alphamin, alphamax);
    if (int rc = launch_bricks(BRICK_TRI_FWD_AUX, volume, dx, dy, dz, source, target, img, nullptr,
                               B, det_h, det_w, voxel_shift, eps, out, aux, nullptr, st,
                               "ddrr_trilinear_forward_bricks", n_points, alphamin, alphamax))
        return rc;
    hipLaunchKernelGGL(tri_out_from_record_kernel, dim3((unsigned)((R + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, st, aux, img, R, n_points, alphamin, alphamax, out);
    return finish("ddrr_trilinear_forward_bricks");
}

int ddrr_trilinear_backward_rays(const float *aux, const float *grad_out, const float *source,
                                 const float *target, const float *img, int B, int N, float eps,
                                 int n_points, const float *alphamin, const float *alphamax,
                                 float *g_source, float *g_target, float *g_img, float *g_alpha,
                                 void *stream) {
    if (!aux || !grad_out || !source || !target || !alphamin || !alphamax)
        return fail(-1, "null pointer");
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    const long R = (long)B * N;
    if (R == 0) return 0;
    hipLaunchKernelGGL(trilinear_bwd_record_kernel, dim3((unsigned)((R + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, aux, grad_out, source, target, img, R,
                       N, eps, n_points, alphamin, alphamax, g_source, g_target, g_img, g_alpha);
    return finish("ddrr_trilinear_backward_rays");
}

int ddrr_trilinear_backward_volume_bricks(int dx, int dy, int dz, const float *source,
                                          const float *target, const float *img,
                                          const float *grad_out, int B, int det_h, int det_w,
                                          float voxel_shift, float eps, int n_points,
                                          const float *alphamin, const float *alphamax,
                                          float *g_volume, void *stream) {
    const int N = det_h * det_w;
    if (!g_volume || !grad_out || !alphamin || !alphamax)
        return fail(-1, "null grad_out / g_volume / alphamin / alphamax");
    if (int rc = check_common(g_volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        hipError_t e = hipMemsetAsync(g_volume, 0, sizeof(float) * (size_t)dx * dy * dz, st);
        return e == hipSuccess ? 0 : fail_hip(e, "hipMemsetAsync");
    }
    return launch_bricks(BRICK_TRI_VOLGRAD, nullptr, dx, dy, dz, source, target, img, grad_out, B,
                         det_h, det_w, voxel_shift, eps, nullptr, nullptr, g_volume, st,
                         "ddrr_trilinear_backward_volume_bricks", n_points, alphamin, alphamax);
}

int ddrr_siddon_backward_rays(const float *aux, int aux_layout, const float *grad_out,
                              const float *source, int src_n, const float *target,
                              const float *img, int B, int N, float eps, int reduce_mode,
                              float *g_source, float *g_target, float *g_img, void *stream) {
    if (!aux || !grad_out || !source || !target) return fail(-1, "null pointer");
    if (src_n != 1 && src_n != N) return fail(-1, "src_n must be 1 or N");
    if (aux_layout != DDRR_AUX_INTERLEAVED && aux_layout != DDRR_AUX_PLANAR)
        return fail(-1, "bad aux_layout");
    if (aux_layout == DDRR_AUX_PLANAR && reduce_mode != DDRR_REDUCE_SUM)
        return fail(-1, "the planar record exists for reduce sum only");
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

int ddrr_raygen_forward(const float *Mw, const float *Ainv, const float *P, int B, int N,
                        float *source_v, float *target_v, float *img, void *stream) {
    if (!Mw || !Ainv || !P || !source_v || !target_v || !img) return fail(-1, "null pointer");
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (B == 0 || N == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 poses per call");
    const dim3 grid((N + kBlock - 1) / kBlock, B), block(kBlock);
    hipLaunchKernelGGL(raygen_fwd_kernel, grid, block, 0, (hipStream_t)stream, Mw, Ainv, P, N,
                       source_v, target_v, img);
    return finish("ddrr_raygen_forward");
}

int ddrr_siddon_backward_pose(const float *aux, int aux_layout, const float *grad_out,
                              const float *source_v, const float *target_v, const float *img,
                              const float *Mw, const float *Ainv, const float *P, int B, int N,
                              float eps, int with_img_path, float *gMw, void *stream) {
    if (!aux || !grad_out || !source_v || !target_v || !img || !Mw || !Ainv || !P || !gMw)
        return fail(-1, "null pointer");
    if (aux_layout != DDRR_AUX_INTERLEAVED && aux_layout != DDRR_AUX_PLANAR)
        return fail(-1, "bad aux_layout");
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (B == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 poses per call");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(gMw, 0, sizeof(float) * 12 * (size_t)B, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    if (N == 0) return 0;
    const dim3 grid((N + kPoseRaysPerBlock - 1) / kPoseRaysPerBlock, B), block(kBlock);
    hipLaunchKernelGGL(siddon_bwd_pose_kernel, grid, block, 0, st, aux, aux_layout, grad_out,
                       source_v, target_v, img, Mw, Ainv, P, B, N, eps, with_img_path, gMw);
    return finish("ddrr_siddon_backward_pose");
}

int ddrr_ncc_forward(const float *x1, long x1_stride, const float *x2, int B, int N, float eps,
                     float *out, float *stats, void *stream) {
    if (!x1 || !x2 || !out || !stats) return fail(-1, "null pointer");
    if (B < 0 || N < 1) return fail(-1, "bad batch / image size");
    if (B == 0) return 0;
    hipLaunchKernelGGL(ncc_fwd_kernel, dim3(B), dim3(kNccThreads), 0, (hipStream_t)stream, x1,
                       x1_stride, x2, N, eps, out, stats);
    return finish("ddrr_ncc_forward");
}

int ddrr_ncc_backward(const float *x1, long x1_stride, const float *x2, const float *stats,
                      const float *g_out, int B, int N, float *g_x1, float *g_x2, void *stream) {
    if (!x1 || !x2 || !stats || !g_out) return fail(-1, "null pointer");
    if (g_x1 && x1_stride == 0) return fail(-1, "a shared x1 gets no gradient here");
    if (B < 0 || N < 1) return fail(-1, "bad batch / image size");
    if (B == 0 || B > 65535) return B == 0 ? 0 : fail(-1, "at most 65535 pairs per call");
    hipLaunchKernelGGL(ncc_bwd_kernel, dim3((N + kBlock - 1) / kBlock, B), dim3(kBlock), 0,
                       (hipStream_t)stream, x1, x1_stride, x2, stats, g_out, N, g_x1, g_x2);
    return finish("ddrr_ncc_backward");
}

int ddrr_pose_euler_forward(const float *rot, const float *xyz, int a0, int a1, int a2,
                            const float *reorient34, int B, float *Mw, void *stream) {
    if (!rot || !xyz || !reorient34 || !Mw) return fail(-1, "null pointer");
    if (a0 < 0 || a0 > 2 || a1 < 0 || a1 > 2 || a2 < 0 || a2 > 2 || a1 == a0 || a1 == a2)
        return fail(-1, "invalid Euler convention");
    if (B <= 0) return B == 0 ? 0 : fail(-1, "negative batch");
    hipLaunchKernelGGL(pose_euler_fwd_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, rot, xyz, a0, a1, a2, reorient34, B, Mw);
    return finish("ddrr_pose_euler_forward");
}

int ddrr_pose_euler_backward(const float *rot, const float *xyz, int a0, int a1, int a2,
                             const float *reorient34, const float *gMw, int B, float *g_rot,
                             float *g_xyz, void *stream) {
    if (!rot || !xyz || !reorient34 || !gMw || !g_rot || !g_xyz) return fail(-1, "null pointer");
    if (a0 < 0 || a0 > 2 || a1 < 0 || a1 > 2 || a2 < 0 || a2 > 2 || a1 == a0 || a1 == a2)
        return fail(-1, "invalid Euler convention");
    if (B <= 0) return B == 0 ? 0 : fail(-1, "negative batch");
    hipLaunchKernelGGL(pose_euler_bwd_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, rot, xyz, a0, a1, a2, reorient34, gMw, B, g_rot, g_xyz);
    return finish("ddrr_pose_euler_backward");
}

}  // extern "C"
