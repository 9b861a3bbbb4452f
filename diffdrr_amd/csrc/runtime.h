// runtime.h -- what the translation units behind the C ABI share: launch geometry of the
// per-ray kernels (256-thread workgroups = 4 wavefronts, one detector ray per lane, one tile
// of 64 rays per wavefront: ddrr_common.h TileMap; 1-D grid over (pose, tile), workgroup ids
// optionally re-mapped so that each of the 8 XCDs works on one contiguous range of tiles),
// argument checks and the thread-local error string of ddrr_last_error().
#pragma once

#include <hip/hip_runtime.h>

#include <stdio.h>
#include <string.h>

#include "../../include/diffdrr_hip.h"
#include "ddrr_common.h"

namespace ddrr_rt {

using namespace ddrr;

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;

int fail(int code, const char *what);
int fail_hip(hipError_t e, const char *where);
int finish(const char *where);  // hipGetLastError() after a launch -> 0 or the error code
const char *last_error();

extern int g_xcd_swizzle;  // experiment knob: XCD-contiguous workgroup mapping (default on)

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

struct AtomicAdder {
    float *base;
    __device__ __forceinline__ void operator()(unsigned off, float v) const {
        unsafeAtomicAdd(base + off, v);  // global_atomic_add_f32, no return
    }
};

struct NoAdd {
    __device__ __forceinline__ void operator()(unsigned, float) const {}
};

// mask_to_channels: the ray owns column out[b, :, n] (zero-filled by the entry point); a run of
// one label is flushed with a fire-and-forget atomic add (siddon_channels_ray /
// trilinear_channels_ray).  Nobody else touches the address -- the atomic is there because a
// plain read-modify-write makes the lane wait ~1 us for the load at every label change (a ray
// through the 119-label example map changes label ~20 times: 0.98 -> see profiles/r02).
struct ColumnFlush {
    float *col;
    long stride;
    int C;
    float L;
    __device__ __forceinline__ void operator()(int label, float run) const {
        if (label < C) unsafeAtomicAdd(col + label * stride, L * run);
    }
};

int check_common(const float *volume, int dx, int dy, int dz, const float *source, int src_n,
                 const float *target, int B, int N);
RayArgs make_args(const float *volume, int dx, int dy, int dz, const float *source, int src_n,
                  const float *target, const float *img, int B, int N, float shift, float eps,
                  int det_h, int det_w, int tile_h, int tile_w);
inline int grid_for(const RayArgs &p) {
    return (p.total_waves + kWavesPerBlock - 1) / kWavesPerBlock;
}

}  // namespace ddrr_rt
