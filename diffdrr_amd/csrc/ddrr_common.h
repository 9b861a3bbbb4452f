// ddrr_common.h -- shared definitions for the MI355X (gfx950) DRR kernels.
//
// The per-ray math lives in *_core.h as `DDRR_HD` functions so that the very
// same source can also be compiled for the host by the test-only emulation
// build (tests/emu), which lets the traversal logic be checked against the
// oracle in the GPU-less build container.  The product only ever runs the
// __global__ kernels of the *.hip translation units.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DDRR_HD __host__ __device__ __forceinline__
#else
#define DDRR_HD inline
#endif

namespace ddrr {

enum Reduce : int { REDUCE_SUM = 0, REDUCE_MAX = 1 };
enum Lookup : int {
    LOOKUP_STEP = 0,      // exact voxel stepping (nearest, align_corners=False): the fast path
    LOOKUP_MID_NEAREST = 1,   // re-derive the voxel from each segment midpoint (any align_corners)
    LOOKUP_MID_TRILINEAR = 2  // trilinear interpolation at each segment midpoint
};

struct Dims {
    int x, y, z;
};

// Separately rounded product / sum (never contracted into an FMA): for the few places that
// restate a reference expression bit for bit.
// (hipcc contracts a * b + c by default and __fmul_rn / __fadd_rn do not stop it: the
// operations are emitted without the `contract` flag through the pragma.)
DDRR_HD float mul_rn(float a, float b) {
#if defined(__clang__)
#pragma clang fp contract(off)
    return a * b;
#else
    volatile float r = a * b;
    return r;
#endif
}
DDRR_HD float add_rn(float a, float b) {
#if defined(__clang__)
#pragma clang fp contract(off)
    return a + b;
#else
    volatile float r = a + b;
    return r;
#endif
}

DDRR_HD float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
DDRR_HD float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
DDRR_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Detector tiling: which ray a lane owns.  A wavefront covers a tile of
// tile_h x tile_w detector pixels (tile_h * tile_w == 64) so that its 64
// voxel fetches per step fall into as few cache lines as possible (the
// detector direction that maps onto the volume's fastest axis, z, should be
// the long side of the tile).  det_h == 0 means "plain ray list": lane l of
// wave w owns ray 64*w + l.
struct TileMap {
    int det_h, det_w;      // 0,0 -> linear
    int tile_h, tile_w;    // product 64
    int tiles_x;           // ceil(det_w / tile_w)
    int waves_per_pose;    // number of 64-ray groups per pose
};

DDRR_HD TileMap make_tilemap(int N, int det_h, int det_w, int tile_h, int tile_w) {
    TileMap m;
    if (det_h <= 0 || det_w <= 0 || (long)det_h * det_w != N || tile_h * tile_w != 64) {
        m.det_h = m.det_w = 0;
        m.tile_h = 1;
        m.tile_w = 64;
        m.tiles_x = 0;
        m.waves_per_pose = (N + 63) / 64;
    } else {
        m.det_h = det_h;
        m.det_w = det_w;
        m.tile_h = tile_h;
        m.tile_w = tile_w;
        m.tiles_x = (det_w + tile_w - 1) / tile_w;
        m.waves_per_pose = m.tiles_x * ((det_h + tile_h - 1) / tile_h);
    }
    return m;
}

// Ray index within the pose for (wave, lane), or -1 if the lane is padding.
DDRR_HD int tile_ray(const TileMap &m, int wave, int lane, int N) {
    if (m.det_h == 0) {
        int n = wave * 64 + lane;
        return n < N ? n : -1;
    }
    int ty = wave / m.tiles_x, tx = wave - ty * m.tiles_x;
    int ly = lane / m.tile_w, lx = lane - ly * m.tile_w;
    int i = ty * m.tile_h + ly, j = tx * m.tile_w + lx;
    return (i < m.det_h && j < m.det_w) ? i * m.det_w + j : -1;
}

}  // namespace ddrr
