// brick_core.h -- volume-stationary ("brick") form of the Siddon forward pass.
//
// The per-ray kernels stream the volume once per pose; at 512^3 a batch of 32
// poses re-reads every voxel ~10 times through the L2/Infinity-Cache fabric,
// which is what bounds them (profiles/r01).  Here the roles are swapped: a
// workgroup stages one 32^3-voxel brick (128 KiB, fits the 160 KiB LDS of a CU)
// and traces, from LDS, the part of EVERY ray of EVERY pose of the batch that
// crosses that brick; the partial line integrals are added to the image with
// fp32 atomics.  The volume is then read from HBM exactly once per batch, and
// the ~10 reads per voxel are LDS reads (random-access bandwidth ~8x the
// texture path's).
//
// Which rays cross a brick: the detector grid of a pose is an affine image of
// the pixel lattice, target(i, j) = o + i e_i + j e_j (reference
// detector.py:126, 147-153), so the brick's 8 corners project through the
// source onto a convex pixel region whose bounding box is enumerated.  The box
// only selects candidates -- each candidate ray is then clipped against the
// brick with the same plane-crossing arithmetic as the full walk
// (siddon_setup on the brick's Box), so brick boundaries are exact: adjacent
// bricks evaluate the shared plane's alpha with the same expression.
#pragma once

#include "ddrr_common.h"
#include "siddon_core.h"

namespace ddrr {

constexpr int BRICK = 32;  // brick edge in voxels

// Fingerprint of the volume a cached 16-bit brick workspace was built from: the bit patterns of
// kFingerprintWords voxels spread over the flat volume.  The build pass stores them in the
// workspace; every launch that is handed the workspace as valid compares them with the live volume
// (siddon_fwd_brick_kernel: one voxel per thread, one round trip, ~2 us per launch) and, on a
// mismatch, renders every brick from the volume's own fp32 values -- the caller's validity test
// cannot see edits that bypass its version counter (PyTorch: `volume.data[...] = x`).  Catches any
// edit that touches a sampled voxel (a replaced or rescaled volume: always); an edit of a few
// voxels between the samples is not seen.
// Sample i lies in the i-th of 1024 equal cells of the flat volume, at a hashed offset inside its
// cell: an even stride alone is a multiple of the row length for every power-of-two volume (512^3:
// every sample in slice z = 0 -- air, in a CT) and would see nothing of an edit of the body.
constexpr int kFingerprintWords = 1024;
DDRR_HD long fingerprint_index(int i, long n_vox) {
    const unsigned long long cell = n_vox >> 10 > 0 ? (unsigned long long)(n_vox >> 10) : 1ull;
    // offsets below the largest power of two in the cell: a mask, not a division
    const unsigned long long mask = (1ull << (63 - __builtin_clzll(cell))) - 1ull;
    const unsigned long long h = ((unsigned)i * 2654435761u + 0x9e3779b9u) >> 4;
    const long idx = (long)((unsigned long long)i * cell + (h & mask));
    return idx < n_vox ? idx : n_vox - 1;
}

struct BrickGrid {
    int nx, ny, nz;
};

DDRR_HD BrickGrid brick_grid(const Dims D) {
    BrickGrid g;
    g.nx = (D.x + BRICK - 1) / BRICK;
    g.ny = (D.y + BRICK - 1) / BRICK;
    g.nz = (D.z + BRICK - 1) / BRICK;
    return g;
}

DDRR_HD Box brick_box(const Dims D, const BrickGrid &g, int id) {
    const int bz = id % g.nz, by = (id / g.nz) % g.ny, bx = id / (g.nz * g.ny);
    Box b;
    b.lo[0] = bx * BRICK;
    b.lo[1] = by * BRICK;
    b.lo[2] = bz * BRICK;
    b.hi[0] = b.lo[0] + BRICK < D.x ? b.lo[0] + BRICK : D.x;
    b.hi[1] = b.lo[1] + BRICK < D.y ? b.lo[1] + BRICK : D.y;
    b.hi[2] = b.lo[2] + BRICK < D.z ? b.lo[2] + BRICK : D.z;
    return b;
}

// Staging by quads of four voxels along z (brick_shared.h quad_load / quad_fix), for volumes of
// at least four slices (D.z >= 4; thinner ones take the scalar staging of the general kernel,
// bricks.hip).  A quad is one 16-byte load from element `at` of the volume, clamped to the
// volume's last four voxels; with D.z >= 4 the only quad this can happen to is the one of the
// volume's LAST ROW that reaches beyond it (none when D.z is a multiple of 4): its words are
// shifted into place where they are used.  (x, y, z): the quad's first voxel (x, y beyond the
// volume: the caller loads from a clamped row and masks the result).
// (Round 5 tried the form that also serves D.z < 4 -- quads of earlier rows and slabs are clamped
// there too -- inside the kernels: 17 more VGPRs in the staging, 4-5 spilled registers in the
// headline kernel.  Hence the split.)
DDRR_HD long quad_clamped_at(const Dims &D, long at) {
    const long last = (long)D.x * D.y * D.z - 4;
    return at < last ? at : last;
}
DDRR_HD int quad_shift(const Dims &D, int x, int y, int z) {
    return x >= D.x - 1 && y >= D.y - 1 ? z + 4 - D.z : 0;  // (> 0: shift; D.z >= 4)
}
DDRR_HD bool quads_serve(const Dims &D) { return D.z >= 4; }

// How a brick is laid out in LDS: z contiguous, rows and planes padded so that the
// voxels a wave reads in one step (neighbouring rays: a small patch perpendicular to the
// rays) spread over the 32 banks.  Strides in floats.
struct BrickLayout {
    int sy, sx;
};

DDRR_HD int brick_floats(const BrickLayout &lay) { return lay.sx * BRICK; }

DDRR_HD Store brick_store(const Box &b, const BrickLayout &lay) {
    Store st;
    st.dims = Dims{BRICK, BRICK, BRICK};
    st.org[0] = b.lo[0];
    st.org[1] = b.lo[1];
    st.org[2] = b.lo[2];
    st.stride[0] = lay.sx * 4;
    st.stride[1] = lay.sy * 4;
    st.stride[2] = 4;
    return st;
}

struct PixBox {
    int i0, i1, j0, j1;  // inclusive; empty when i1 < i0 or j1 < j0
};

DDRR_HD int pixbox_count(const PixBox &b) {
    return (b.i1 < b.i0 || b.j1 < b.j0) ? 0 : (b.i1 - b.i0 + 1) * (b.j1 - b.j0 + 1);
}

// Pixel bounding box of the lines through `src` that meet the box `b`
// (plane indices; a plane k sits at x = k - shift).  tgt points at the pose's
// (det_h * det_w, 3) target grid.
DDRR_HD PixBox project_brick(const float *src, const float *tgt, int det_h, int det_w,
                             const Box &b, float shift) {
    const float *t00 = tgt, *t01 = tgt + 3, *t10 = tgt + 3 * det_w;
    float ei[3], ej[3], r[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ei[a] = t10[a] - t00[a];
        ej[a] = t01[a] - t00[a];
        r[a] = t00[a] - src[a];
    }
    // solve lambda * w - i * ei - j * ej = r for every corner w = p - src (Cramer)
    const float n[3] = {ei[1] * ej[2] - ei[2] * ej[1], ei[2] * ej[0] - ei[0] * ej[2],
                        ei[0] * ej[1] - ei[1] * ej[0]};  // ei x ej
    const float rxej[3] = {r[1] * ej[2] - r[2] * ej[1], r[2] * ej[0] - r[0] * ej[2],
                           r[0] * ej[1] - r[1] * ej[0]};
    const float rxei[3] = {r[1] * ei[2] - r[2] * ei[1], r[2] * ei[0] - r[0] * ei[2],
                           r[0] * ei[1] - r[1] * ei[0]};
    float imin = INFINITY, imax = -INFINITY, jmin = INFINITY, jmax = -INFINITY;
    int npos = 0, nneg = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float w[3] = {(float)((c & 1) ? b.hi[0] : b.lo[0]) - shift - src[0],
                            (float)((c & 2) ? b.hi[1] : b.lo[1]) - shift - src[1],
                            (float)((c & 4) ? b.hi[2] : b.lo[2]) - shift - src[2]};
        const float det = w[0] * n[0] + w[1] * n[1] + w[2] * n[2];
        npos += det > 0.f;
        nneg += det < 0.f;
        const float inv = 1.0f / det;
        const float i = -(w[0] * rxej[0] + w[1] * rxej[1] + w[2] * rxej[2]) * inv;
        const float j = (w[0] * rxei[0] + w[1] * rxei[1] + w[2] * rxei[2]) * inv;
        imin = fminf(imin, i);
        imax = fmaxf(imax, i);
        jmin = fminf(jmin, j);
        jmax = fmaxf(jmax, j);
    }
    PixBox pb;
    if (npos != 8 && nneg != 8) {
        // the plane through the source parallel to the detector cuts the box: its
        // projection is unbounded -> every pixel is a candidate
        pb.i0 = 0;
        pb.i1 = det_h - 1;
        pb.j0 = 0;
        pb.j1 = det_w - 1;
        return pb;
    }
    // pixel centres are the integer (i, j); keep a small guard band for rounding
    const float g = 0.02f;
    const float fi0 = fmaxf(floorf(imin - g), 0.f), fi1 = fminf(ceilf(imax + g), (float)(det_h - 1));
    const float fj0 = fmaxf(floorf(jmin - g), 0.f), fj1 = fminf(ceilf(jmax + g), (float)(det_w - 1));
    if (!(fi0 <= fi1) || !(fj0 <= fj1)) {  // also catches NaN
        pb.i0 = pb.j0 = 0;
        pb.i1 = pb.j1 = -1;
        return pb;
    }
    pb.i0 = (int)fi0;
    pb.i1 = (int)fi1;
    pb.j0 = (int)fj0;
    pb.j1 = (int)fj1;
    return pb;
}

struct LdsFetch {
    const float *brick;  // BrickLayout-strided floats
    DDRR_HD float operator()(unsigned boff) const {
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(brick) + boff);
    }
};

// Phase-A test of a candidate ray against a brick: a cheap slab clip (hardware reciprocal,
// no refinement) with a little slack, so that it never rejects a ray whose exact clip
// (siddon_setup_fast, evaluated later on the survivors) is a chord longer than rounding
// noise; false positives cost one idle lane in the walk.  `n_est` ~ number of plane
// crossings inside the brick: used only to group hits of similar length into one wave.
DDRR_HD bool brick_maybe_hit(const Box &box, const float s[3], const float t[3], float shift,
                             float eps, float &n_est) {
    float entry = -INFINITY, exit = INFINITY, l1 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float d = (t[a] - s[a]) + eps;
#if defined(__HIP_DEVICE_COMPILE__)
        const float inv = __builtin_amdgcn_rcpf(d);
#else
        const float inv = 1.0f / d;
#endif
        const float a0 = ((float)box.lo[a] - shift - s[a]) * inv;
        const float a1 = ((float)box.hi[a] - shift - s[a]) * inv;
        entry = fmaxf(entry, fminf(a0, a1));
        exit = fminf(exit, fmaxf(a0, a1));
        l1 += fabsf(d);
    }
    const float slack = 1e-6f * fmaxf(fabsf(entry), fabsf(exit));
    n_est = (exit - entry) * l1;
    return entry < exit + slack;  // false for NaN
}

#if defined(__HIPCC__)
// Fetch by ABSOLUTE LDS byte address (the brick's base is folded into the offset the walk
// carries, so a step issues `ds_read_b32 v, addr` with no address arithmetic).
struct LdsAbsFetch {
    static __device__ __forceinline__ unsigned base_of(const float *brick) {
#if defined(__HIP_DEVICE_COMPILE__)
        return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float *)brick;
#else
        (void)brick;
        return 0u;  // (host pass of the single-source compile: never executed)
#endif
    }
    __device__ __forceinline__ float operator()(unsigned addr) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return *(const __attribute__((address_space(3))) float *)(unsigned long long)addr;
#else
        (void)addr;
        return 0.f;
#endif
    }
};
#endif

// Candidate pixel `local` (row-major index into the pixel box) -> (i, j).
DDRR_HD void pixbox_pixel(int i0, int j0, int w, float inv_w, int local, int &i, int &j) {
    // exact for the box sizes that occur (local < 2^22): no integer division
    int di = (int)(((float)local + 0.5f) * inv_w);
    di -= (di * w > local) ? 1 : 0;
    di += ((di + 1) * w <= local) ? 1 : 0;
    i = i0 + di;
    j = j0 + (local - di * w);
}

}  // namespace ddrr
