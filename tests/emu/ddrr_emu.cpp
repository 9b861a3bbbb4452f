// ddrr_emu.cpp -- TEST-ONLY host build of the per-ray kernel cores.
//
// Compiles diffdrr_amd/csrc/{siddon,trilinear}_core.h for the CPU (the headers
// are written as __host__ __device__ code) behind the same C ABI as
// include/diffdrr_hip.h, with every pointer a HOST pointer.  It lets the
// traversal logic of the HIP kernels be checked against the oracle in the
// GPU-less build container (tests/test_emu_vs_oracle.py).  It is NOT a CPU
// fallback: nothing in diffdrr_amd/ can load it, and it is built into
// tests/emu/_build only by the tests.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "../../diffdrr_amd/csrc/ddrr_common.h"
#include "../../diffdrr_amd/csrc/siddon_core.h"
#include "../../diffdrr_amd/csrc/brick_core.h"
#include "../../diffdrr_amd/csrc/brick_walk.h"
#include "../../diffdrr_amd/csrc/brick_step.h"
#include "../../diffdrr_amd/csrc/raygen_core.h"
#include "../../diffdrr_amd/csrc/record_pack.h"
#include "../../diffdrr_amd/csrc/record_layout.h"
#include "../../diffdrr_amd/csrc/segments_core.h"
#include "../../diffdrr_amd/csrc/ncc_patch_core.h"
#include "../../diffdrr_amd/csrc/sobel_core.h"
#include "../../diffdrr_amd/csrc/blur_core.h"
#include "../../diffdrr_amd/csrc/tri_brick.h"
#include "../../diffdrr_amd/csrc/trilinear_core.h"
#include "../../diffdrr_amd/csrc/f64_core.h"
#include "../../diffdrr_amd/csrc/general_core.h"
#include "../../include/diffdrr_hip.h"

using namespace ddrr;

namespace {

struct Ray {
    float s[3], t[3], L;
};

// Visit rays in the same (wave, lane) order as the kernels so that the tile
// map is exercised too.
template <class F>
void for_each_ray(const float *source, int src_n, const float *target, const float *img, int B,
                  int N, int det_h, int det_w, int tile_h, int tile_w, F f) {
    const TileMap tm = make_tilemap(N, det_h, det_w, tile_h, tile_w);
    std::vector<char> seen((size_t)B * N, 0);
    for (int b = 0; b < B; ++b)
        for (int w = 0; w < tm.waves_per_pose; ++w)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = tile_ray(tm, w, lane, N);
                if (n < 0) continue;
                const long r = (long)b * N + n;
                seen[r]++;
                Ray ray;
                const float *sp = source + ((long)b * src_n + (src_n == 1 ? 0 : n)) * 3;
                for (int a = 0; a < 3; ++a) {
                    ray.s[a] = sp[a];
                    ray.t[a] = target[r * 3 + a];
                }
                ray.L = img ? img[r] : 1.f;
                f(b, n, r, ray);
            }
    for (size_t i = 0; i < seen.size(); ++i)
        if (seen[i] != 1) abort();  // the tile map must be a bijection
}

struct HostAdd {
    float *base;
    void operator()(unsigned off, float v) const { base[off] += v; }
};
struct NoAdd {
    void operator()(unsigned, float) const {}
};

}  // namespace

namespace {
// Host emulation of the marcher's forward brick kernel: per brick of 31^3 base cells (32^3
// staged voxels, zeros outside the volume), every candidate pixel of every pose marches its
// samples; accumulates out, or (aux != NULL) the 7-plane backward record.
int tri_bricks_host(const float *volume, int dx, int dy, int dz, const float *source,
                    const float *target, const float *img, int B, int det_h, int det_w,
                    float voxel_shift, float eps, int n_points, float amin, float amax,
                    float *out, float *aux, const unsigned char *labels = nullptr, int C = 0,
                    const float *grad_cols = nullptr) {
    // labels: mask_to_channels (out is (B, C, N)): packed words, tri_brick_march_channels;
    // grad_cols (B, C, N): its ray backward instead -- the weighted record (tri_brick_march_weighted)
    const Dims D{dx, dy, dz};
    const int N = det_h * det_w;
    const long R = (long)B * N;
    const BrickGrid bg = tri_brick_grid(D);
    const BrickLayout lay{33, 32 * 33 + 1};
    std::vector<float> brick((size_t)brick_floats(lay));
    struct HostFetch {
        const float *base;
        float operator()(unsigned off) const { return base[off >> 2]; }
    };
    const float step = (amax - amin) / (float)(n_points - 1);
    const float nscale = (float)(n_points - 1) / (amax - amin);
    for (int id = 0; id < bg.nx * bg.ny * bg.nz; ++id) {
        int lo[3];
        tri_brick_lo(bg, id, lo);
        BoxF cells;
        for (int a = 0; a < 3; ++a) {
            cells.lo[a] = (float)lo[a] + 0.5f;
            cells.hi[a] = (float)(lo[a] + TRI_CELLS) + 0.5f;
        }
        const TriGeom G = tri_geom(lo, lay);
        std::fill(brick.begin(), brick.end(), 0.f);
        for (int lx = 0; lx < BRICK; ++lx)
            for (int ly = 0; ly < BRICK; ++ly)
                for (int lz = 0; lz < BRICK; ++lz) {
                    const int x = lo[0] + lx, y = lo[1] + ly, z = lo[2] + lz;
                    if (x < 0 || y < 0 || z < 0 || x >= dx || y >= dy || z >= dz) continue;
                    const long at = ((long)x * dy + y) * dz + z;
                    brick[lx * lay.sx + ly * lay.sy + lz] =
                        labels && !grad_cols ? pack_voxel_label(volume[at], labels[at]) : volume[at];
                }
        for (int b = 0; b < B; ++b) {
            const PoseGrid pg = pose_grid(source + (long)b * 3, target + (long)b * N * 3, det_h,
                                          det_w);
            PixBox pb = project_brick_grid(pg, det_h, det_w, cells, voxel_shift);
            if (aux) pb = align_pixbox_rows(pb, det_w);  // as the kernel does with a float record
            const BrickRow row = brick_row(pg, pb, cells, voxel_shift, eps, nscale);
            std::vector<char> cand((size_t)N, 0);
            auto ray = [&](int pix, float s[3], float t[3]) {
                const long r = (long)b * N + pix;
                for (int a = 0; a < 3; ++a) {
                    s[a] = source[(long)b * 3 + a];
                    t[a] = target[r * 3 + a];
                }
                return r;
            };
            for (int local = 0; local < row.count; ++local) {
                int pix;
                float n_est;
                if (!brick_candidate(row, local, det_w, pix, n_est)) continue;
                cand[pix] = 1;
                float s[3], t[3], sumT, rec[6];
                const long r = ray(pix, s, t);
                const float L = img ? img[r] : 1.f;
                if (labels && grad_cols) {
                    const float *col = grad_cols + (long)b * C * N + pix;
                    if (tri_brick_march_weighted(
                            HostFetch{brick.data()},
                            [&](float rx, float ry, float rz) -> unsigned {
                                const bool in = rx >= 0.f && ry >= 0.f && rz >= 0.f && rx < (float)dx &&
                                                ry < (float)dy && rz < (float)dz;
                                return in ? labels[((long)(int)rx * dy + (int)ry) * dz + (int)rz] : 0u;
                            },
                            0.f, G, D, s, t, voxel_shift, eps, n_points, amin,
                            amax, [&](unsigned lab) { return lab < (unsigned)C ? col[(long)lab * N] : 0.f; },
                            sumT, rec)) {
                        aux[r] += sumT;
                        for (int k = 0; k < 6; ++k) aux[(k + 1) * R + r] += rec[k];
                    }
                    continue;
                }
                if (labels) {
                    float *col = out + (long)b * C * N + pix;
                    tri_brick_march_channels(HostFetch{brick.data()}, 0.f, G, D, s, t, voxel_shift, eps,
                                             n_points, amin, amax, [&](unsigned lab, float run) {
                                                 if (lab < (unsigned)C) col[(long)lab * N] += L * step * run;
                                             });
                    continue;
                }
                if (aux) {
                    if (tri_brick_march<true>(HostFetch{brick.data()}, 0.f, G, s, t, voxel_shift,
                                              eps, n_points, amin, amax, sumT, rec)) {
                        aux[r] += sumT;
                        for (int k = 0; k < 6; ++k) aux[(k + 1) * R + r] += rec[k];
                    }
                } else if (tri_brick_march<false>(HostFetch{brick.data()}, 0.f, G, s, t,
                                                  voxel_shift, eps, n_points, amin, amax, sumT,
                                                  rec)) {
                    out[r] += L * step * sumT;
                }
            }
            // phase A must not lose a pixel with samples in this brick
            for (int pix = 0; pix < N && !labels; ++pix) {
                if (cand[pix]) continue;
                float s[3], t[3], sumT, rec[6];
                ray(pix, s, t);
                if (tri_brick_march<false>(HostFetch{brick.data()}, 0.f, G, s, t, voxel_shift, eps,
                                           n_points, amin, amax, sumT, rec) &&
                    sumT != 0.f)
                    abort();
            }
        }
    }
    if (aux && out)
        for (long r = 0; r < R; ++r) out[r] = (img ? img[r] : 1.f) * step * aux[r];
    return 0;
}

// The marcher's volume gradient on OWNER bricks (tri_owner_scatter): per 32^3 voxel brick,
// every candidate pixel of every pose scatters the corners the brick owns into the staged
// accumulator, which is then stored (each voxel exactly once: g_volume needs no zero fill).
int tri_owner_host(int dx, int dy, int dz, const float *source, const float *target,
                   const float *img, const float *grad_out, int B, int det_h, int det_w,
                   float voxel_shift, float eps, int n_points, float amin, float amax,
                   float *g_volume, const unsigned char *labels = nullptr, int C = 0) {
    // labels: the channel render's volume gradient, grad_out is (B, C, N) (tri_owner_scatter_weighted;
    // the emulation looks every label up in the label map)
    const Dims D{dx, dy, dz};
    const int N = det_h * det_w;
    const BrickGrid bg = brick_grid(D);
    const BrickLayout lay{33, 32 * 33 + 1};
    std::vector<float> brick((size_t)brick_floats(lay));
    struct HostAddB {
        float *base;
        void operator()(unsigned off, float v) const { base[off >> 2] += v; }
    };
    const float step = (amax - amin) / (float)(n_points - 1);
    const float nscale = (float)(n_points - 1) / (amax - amin);
    for (int id = 0; id < bg.nx * bg.ny * bg.nz; ++id) {
        const Box box = brick_box(D, bg, id);
        const BrickGeom G = brick_geom(box, lay);
        BoxF cells;
        for (int a = 0; a < 3; ++a) {
            cells.lo[a] = (float)(box.lo[a] - 1) + 0.5f;
            cells.hi[a] = (float)box.hi[a] + 0.5f;
        }
        std::fill(brick.begin(), brick.end(), 0.f);
        for (int b = 0; b < B; ++b) {
            const PoseGrid pg = pose_grid(source + (long)b * 3, target + (long)b * N * 3, det_h,
                                          det_w);
            const PixBox pb = project_brick_grid(pg, det_h, det_w, cells, voxel_shift);
            const BrickRow row = brick_row(pg, pb, cells, voxel_shift, eps, nscale);
            std::vector<char> cand((size_t)N, 0);
            auto scatter = [&](int pix, float *acc) {
                const long r = (long)b * N + pix;
                float s[3], t[3];
                for (int a = 0; a < 3; ++a) {
                    s[a] = source[(long)b * 3 + a];
                    t[a] = target[r * 3 + a];
                }
                const float L = img ? img[r] : 1.f;
                if (labels) {
                    const float *col = grad_out + (long)b * C * N + pix;
                    tri_owner_scatter_weighted(
                        HostAddB{acc},
                        [&](float rx, float ry, float rz, bool, unsigned) -> unsigned {
                            const bool in = rx >= 0.f && ry >= 0.f && rz >= 0.f && rx < (float)dx &&
                                            ry < (float)dy && rz < (float)dz;
                            return in ? labels[((long)(int)rx * dy + (int)ry) * dz + (int)rz] : 0u;
                        },
                        [&](unsigned l) { return l < (unsigned)C ? col[(long)l * N] : 0.f; }, 0.f, G.lof,
                        G.hif, G.stridef, D, s, t, voxel_shift, eps, n_points, amin, amax, L * step);
                    return;
                }
                tri_owner_scatter(HostAddB{acc}, 0.f, G.lof, G.hif, G.stridef, s, t, voxel_shift,
                                  eps, n_points, amin, amax, grad_out[r] * L * step);
            };
            for (int local = 0; local < row.count; ++local) {
                int pix;
                float n_est;
                if (!brick_candidate(row, local, det_w, pix, n_est)) continue;
                cand[pix] = 1;
                scatter(pix, brick.data());
            }
            // phase A must not lose a pixel with samples that touch this brick
            std::vector<float> probe(brick.size());
            for (int pix = 0; pix < N; ++pix) {
                if (cand[pix] || labels || grad_out[(long)b * N + pix] == 0.f) continue;
                std::fill(probe.begin(), probe.end(), 0.f);
                scatter(pix, probe.data());
                for (float v : probe)
                    if (v != 0.f) abort();
            }
        }
        for (int lx = 0; lx < BRICK; ++lx)
            for (int ly = 0; ly < BRICK; ++ly)
                for (int lz = 0; lz < BRICK; ++lz) {
                    const int x = box.lo[0] + lx, y = box.lo[1] + ly, z = box.lo[2] + lz;
                    if (x >= box.hi[0] || y >= box.hi[1] || z >= box.hi[2]) continue;
                    g_volume[((long)x * dy + y) * dz + z] = brick[lx * lay.sx + ly * lay.sy + lz];
                }
    }
    return 0;
}
}  // namespace

// The brick kernel's record of ray r as the interleaved 8 floats (the blocked float record of
// record_layout.h, or the packed fixed-point form of record_pack.h).
static void planar_record(const float *aux, int aux_layout, long R, long r, float rec[SIDDON_AUX]) {
    if (aux_layout == DDRR_AUX_BLOCKED) {
        rec_blocked_load(aux, r, rec);
        return;
    }
    float I, S0x, S0z, S1x, S1z;
    const long long *X = reinterpret_cast<const long long *>(aux);
    const float q = aux[6 * R], qa = q / aux[5 * R + r];
    record_unpack(X[r], q, qa, S0x, S1x);
    record_unpack(X[R + r], q, qa, S0z, S1z);
    I = aux[4 * R + r];
    const float v[SIDDON_AUX] = {I, S0x, -(S0x + S0z), S0z, S1x, I - (S1x + S1z), S1z, 0.f};
    memcpy(rec, v, sizeof(v));
}

extern "C" {

int ddrr_abi_version(void) { return DDRR_ABI_VERSION; }
const char *ddrr_last_error(void) { return ""; }

int ddrr_siddon_forward(const float *volume, int dx, int dy, int dz, const float *source,
                        int src_n, const float *target, const float *img, int B, int N,
                        float voxel_shift, float eps, int reduce_mode, int lookup_mode,
                        int align_corners, int det_h, int det_w, int tile_h, int tile_w,
                        float *out, float *aux, int *n_vox, void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(source, src_n, target, img, B, N, det_h, det_w, tile_h, tile_w,
                 [&](int, int, long r, const Ray &ray) {
                     float rec[SIDDON_AUX] = {0};
                     int cnt = 0;
                     float I;
                     const bool sum = reduce_mode == DDRR_REDUCE_SUM;
                     if (lookup_mode == DDRR_LOOKUP_STEP) {
                         if (n_vox)
                             I = sum ? siddon_forward_ray<REDUCE_SUM, false, true>(
                                           volume, D, full_box(D), ray.s, ray.t, voxel_shift, eps, rec, &cnt)
                                     : siddon_forward_ray<REDUCE_MAX, false, true>(
                                           volume, D, full_box(D), ray.s, ray.t, voxel_shift, eps, rec, &cnt);
                         else if (aux)
                             I = sum ? siddon_forward_ray<REDUCE_SUM, true, false>(
                                           volume, D, full_box(D), ray.s, ray.t, voxel_shift, eps, rec, &cnt)
                                     : siddon_forward_ray<REDUCE_MAX, true, false>(
                                           volume, D, full_box(D), ray.s, ray.t, voxel_shift, eps, rec, &cnt);
                         else
                             I = sum ? siddon_forward_ray<REDUCE_SUM, false, false>(
                                           volume, D, full_box(D), ray.s, ray.t, voxel_shift, eps, rec, &cnt)
                                     : siddon_forward_ray<REDUCE_MAX, false, false>(
                                           volume, D, full_box(D), ray.s, ray.t, voxel_shift, eps, rec, &cnt);
                     } else if (lookup_mode == DDRR_LOOKUP_MID_TRILINEAR) {
                         I = sum ? siddon_forward_ray_midpoint<REDUCE_SUM, LOOKUP_MID_TRILINEAR>(
                                       volume, D, ray.s, ray.t, voxel_shift, eps, align_corners)
                                 : siddon_forward_ray_midpoint<REDUCE_MAX, LOOKUP_MID_TRILINEAR>(
                                       volume, D, ray.s, ray.t, voxel_shift, eps, align_corners);
                     } else {
                         I = sum ? siddon_forward_ray_midpoint<REDUCE_SUM, LOOKUP_MID_NEAREST>(
                                       volume, D, ray.s, ray.t, voxel_shift, eps, align_corners)
                                 : siddon_forward_ray_midpoint<REDUCE_MAX, LOOKUP_MID_NEAREST>(
                                       volume, D, ray.s, ray.t, voxel_shift, eps, align_corners);
                     }
                     out[r] = ray.L * I;
                     if (aux) memcpy(aux + r * SIDDON_AUX, rec, sizeof(rec));
                     if (n_vox) n_vox[r] = cnt;
                 });
    return 0;
}

// Host emulation of siddon_fwd_brick_kernel: same per-pose table (affine detector model,
// projected pixel box), same arithmetic phase-A test, hits compacted into three
// length-class queues and walked in batches of 64 like a wave does, same exact clip and
// walk (brick_walk.h); LDS brick replaced by a local padded copy (same BrickLayout
// strides), atomics by plain adds.
// (the emulation keeps no packed copy: the packed storage is the same arithmetic from a
// different staging source; the workspace is sized like the product's)
long ddrr_brick_workspace_bytes(int dx, int dy, int dz, int brick_storage) {
    if (brick_storage == DDRR_BRICKS_F32 || dx < 1 || dy < 1 || dz < 1) return 0;
    const long n32 = (long)((dx + 31) / 32) * ((dy + 31) / 32) * ((dz + 31) / 32);
    long n = 256 + (n32 * 12 + 255) / 256 * 256;  // header, (min, max) and fallback flag per brick
    n += kFingerprintWords * 4;                   // the volume's fingerprint (brick_core.h)
    if (brick_storage == DDRR_BRICKS_Q16_PACKED)
        n += (long)((dx + 31) / 32) * ((dy + 31) / 32) * ((dz + 63) / 64) * 133184;
    return n;
}

// (the emulation keeps no per-launch device state; the size is the product's)
long ddrr_brick_launch_workspace_bytes(int dx, int dy, int dz) {
    if (dx < 1 || dy < 1 || dz < 1) return 0;
    const long n32 = (long)((dx + 31) / 32) * ((dy + 31) / 32) * ((dz + 31) / 32);
    return 256 + (n32 * 8 + 255) / 256 * 256;
}

static int emu_siddon_forward_bricks(const float *volume, int dx, int dy, int dz, const float *source,
                                     const float *target, const float *img, int B, int det_h,
                                     int det_w, float voxel_shift, float eps, float *out, float *aux,
                                     float record_vmax, int brick_storage, float *brick_ws,
                                     int ranges_valid, const unsigned *pixel_mask);

int ddrr_siddon_forward_bricks(const float *volume, int dx, int dy, int dz, const float *source,
                               const float *target, const float *img, int B, int det_h,
                               int det_w, float voxel_shift, float eps, float *out, float *aux,
                               float record_vmax, int brick_storage, float *brick_ws,
                               int ranges_valid, void * /*launch_ws*/, void *) {
    return emu_siddon_forward_bricks(volume, dx, dy, dz, source, target, img, B, det_h, det_w, voxel_shift, eps,
                                     out, aux, record_vmax, brick_storage, brick_ws, ranges_valid, nullptr);
}

int ddrr_siddon_forward_bricks_masked(const float *volume, int dx, int dy, int dz, const float *source,
                                      const float *target, const float *img, int B, int det_h,
                                      int det_w, float voxel_shift, float eps, float *out, float *aux,
                                      float record_vmax, int brick_storage, float *brick_ws,
                                      int ranges_valid, void * /*launch_ws*/, const unsigned *pixel_mask,
                                      void *) {
    return emu_siddon_forward_bricks(volume, dx, dy, dz, source, target, img, B, det_h, det_w, voxel_shift, eps,
                                     out, aux, record_vmax, brick_storage, brick_ws, ranges_valid, pixel_mask);
}

static int emu_siddon_forward_bricks(const float *volume, int dx, int dy, int dz, const float *source,
                                     const float *target, const float *img, int B, int det_h,
                                     int det_w, float voxel_shift, float eps, float *out, float *aux,
                                     float record_vmax, int brick_storage, float *brick_ws,
                                     int ranges_valid, const unsigned *pixel_mask) {
    ranges_valid &= 1;  // (DDRR_BRICKS_CLEARED: the emulation clears its outputs anyway)
    const Dims D{dx, dy, dz};
    // workspace layout of the product (bricks_fwd.hip): header, (min, max) per brick, fallback
    // flags.  The emulation walks 32^3 bricks; like the product it TRUSTS a workspace handed
    // over as valid (ranges_valid) instead of looking at the volume again.
    const long n32 = (long)((dx + 31) / 32) * ((dy + 31) / 32) * ((dz + 31) / 32);
    int *ws_header = reinterpret_cast<int *>(brick_ws);
    float *brick_ranges = brick_ws ? brick_ws + 64 : nullptr;
    int *brick_fallback = brick_ws ? reinterpret_cast<int *>(brick_ranges + 2 * n32) : nullptr;
    int n_fallback = 0;
    const bool q16 = brick_storage == DDRR_BRICKS_Q16 || brick_storage == DDRR_BRICKS_Q16_PACKED;
    // the fingerprint of the volume the workspace was built from (brick_core.h): a workspace handed
    // over as valid whose volume has changed renders every brick from its fp32 values
    unsigned *fingerprint = brick_ws ? reinterpret_cast<unsigned *>(
        reinterpret_cast<unsigned char *>(brick_ws) + 256 + (n32 * 12 + 255) / 256 * 256) : nullptr;
    bool stale = false;
    if (q16 && fingerprint) {
        const long n_vox = (long)dx * dy * dz;
        for (int i = 0; i < kFingerprintWords; ++i) {
            unsigned bits;
            memcpy(&bits, volume + fingerprint_index(i, n_vox), 4);
            if (!ranges_valid) fingerprint[i] = bits;
            else stale = stale || fingerprint[i] != bits;
        }
        if (!ranges_valid) ws_header[2] = 0;
        else if (stale) ws_header[2] += 1;
    }
    // 16-bit bricks (bricks_fwd.hip CfgQ16x2): rows and planes padded by one element
    const int qsy = 32 * 2 + 2, qsx = 32 * qsy + 2;
    std::vector<unsigned short> qbrick((size_t)qsx * 32 / 2);
    const int N = det_h * det_w;
    const long plane = (long)B * N;
    // (out may be NULL with aux: the record alone)
    std::vector<float> out_scratch;
    if (!out) {
        if (!aux) return -1;
        out_scratch.resize((size_t)B * N);
        out = out_scratch.data();
    }
    memset(out, 0, sizeof(float) * (size_t)B * N);
    // packed fixed-point record (record_pack.h): planes 5 (A per ray) and 6 (q) first
    const bool packed = aux && record_vmax > 0.f;
    if (aux)
        memset(aux, 0, sizeof(float) * (packed ? (size_t)5 * B * N : (size_t)rec_blocked_floats(plane)));
    const float rec_q = packed ? record_scale(record_vmax, D) : 0.f;
    long long *packedX = reinterpret_cast<long long *>(aux);
    if (packed) {
        aux[6 * plane] = rec_q;
        for (long r = 0; r < plane; ++r)
            aux[5 * plane + r] = record_alpha_bound(D, source + (r / N) * 3, target + r * 3,
                                                    voxel_shift, eps);
    }
    const BrickGrid bg = brick_grid(D);
    const BrickLayout lay{33, 32 * 33 + 1};
    std::vector<float> brick((size_t)brick_floats(lay));
    std::vector<std::pair<int, int>> queues[3];
    for (int id = 0; id < bg.nx * bg.ny * bg.nz; ++id) {
        const Box box = brick_box(D, bg, id);
        const StepGeom SG = step_geom(box, lay);
        std::fill(brick.begin(), brick.end(), 0.f);
        float vmin = INFINITY, vmax = -INFINITY;
        bool bad = false;  // (min / max drop NaNs)
        for (int x = box.lo[0]; x < box.hi[0]; ++x)
            for (int y = box.lo[1]; y < box.hi[1]; ++y)
                for (int z = box.lo[2]; z < box.hi[2]; ++z) {
                    const float v = volume[((long)x * dy + y) * dz + z];
                    brick[(x - box.lo[0]) * lay.sx + (y - box.lo[1]) * lay.sy + (z - box.lo[2])] = v;
                    vmin = fminf(vmin, v);
                    vmax = fmaxf(vmax, v);
                    bad = bad || !(fabsf(v) < INFINITY);
                }
        if (bad) vmax = NAN;
        // the brick's level: smallest mean |V| of its 4^3 blocks (brick_range_kernel)
        float level = INFINITY;
        for (int x4 = box.lo[0]; x4 < box.hi[0]; x4 += 4)
            for (int y4 = box.lo[1]; y4 < box.hi[1]; y4 += 4)
                for (int z4 = box.lo[2]; z4 < box.hi[2]; z4 += 4) {
                    float sum = 0.f;
                    int nnz = 0, cnt = 0;
                    for (int x = x4; x < x4 + 4 && x < box.hi[0]; ++x)
                        for (int y = y4; y < y4 + 4 && y < box.hi[1]; ++y)
                            for (int z = z4; z < z4 + 4 && z < box.hi[2]; ++z) {
                                const float v = volume[((long)x * dy + y) * dz + z];
                                sum += fabsf(v);
                                nnz += v != 0.f;
                                ++cnt;
                            }
                    const int n_eff = vmin == 0.f ? nnz : cnt;
                    if (n_eff > 0) level = fminf(level, sum / (float)n_eff);
                }
        bool fallback = !q16_usable(vmin, vmax, level);
        if (q16 && brick_ranges) {
            if (ranges_valid) {
                vmin = brick_ranges[2 * id];
                vmax = brick_ranges[2 * id + 1];
                fallback = stale || brick_fallback[id] != 0;
            } else {
                brick_ranges[2 * id] = vmin;
                brick_ranges[2 * id + 1] = vmax;
                brick_fallback[id] = fallback;
            }
        }
        n_fallback += fallback;
        const bool q16_here = q16 && !fallback;
        Q16Range range = q16_range(vmin, vmax);
        StepGeom SGq = SG;
        if (q16_here) {
            SGq.strideb[0] = bits_as_float((unsigned)qsx);
            SGq.strideb[1] = bits_as_float((unsigned)qsy);
            SGq.strideb[2] = bits_as_float(2u);
            std::fill(qbrick.begin(), qbrick.end(), (unsigned short)0);
            for (int x = box.lo[0]; x < box.hi[0]; ++x)
                for (int y = box.lo[1]; y < box.hi[1]; ++y)
                    for (int z = box.lo[2]; z < box.hi[2]; ++z)
                        qbrick[((x - box.lo[0]) * qsx + (y - box.lo[1]) * qsy) / 2 + (z - box.lo[2])] =
                            (unsigned short)q16_encode(volume[((long)x * dy + y) * dz + z], range);
        }
        auto item = [&](int b, int pix) {
            const long r = (long)b * N + pix;
            float s[3], t[3];
            for (int a = 0; a < 3; ++a) {
                s[a] = source[(long)b * 3 + a];
                t[a] = target[r * 3 + a];
            }
            float I, rec[4];
            bool hit;
            if (q16_here)
                hit = aux ? step_trace_q16<true, 100>(LdsFetch16{qbrick.data()}, 0u, SGq, range, s, t,
                                                      voxel_shift, eps, I, rec)
                          : step_trace_q16<false, 100>(LdsFetch16{qbrick.data()}, 0u, SGq, range, s,
                                                       t, voxel_shift, eps, I, rec);
            else
                hit = aux ? step_trace<true>(LdsFetch{brick.data()}, 0u, SG, s, t, voxel_shift,
                                             eps, I, rec)
                          : step_trace<false>(LdsFetch{brick.data()}, 0u, SG, s, t,
                                              voxel_shift, eps, I, rec);
            if (!hit) return;  // phase A's margin let a non-crossing ray through
            out[r] += (img ? img[r] : 1.f) * I;
            if (packed) {
                const float qa = rec_q / aux[5 * plane + r];
                packedX[r] += record_pack(rec[0], rec[2], rec_q, qa);
                packedX[plane + r] += record_pack(rec[1], rec[3], rec_q, qa);
                aux[4 * plane + r] += I;
            } else if (aux) {
                aux[rec_index(r, 0)] += I;
                for (int k = 0; k < 4; ++k) aux[rec_index(r, k + 1)] += rec[k];
            }
        };
        for (auto &qk : queues) qk.clear();
        for (int b = 0; b < B; ++b) {
            const PoseGrid pg = pose_grid(source + (long)b * 3, target + (long)b * N * 3, det_h,
                                          det_w);
            PixBox pb = project_brick_grid(pg, det_h, det_w, boxf(box), voxel_shift);
            if (aux && !packed) pb = align_pixbox_rows(pb, det_w);  // as the kernel does
            const BrickRow row = brick_row(pg, pb, boxf(box), voxel_shift, eps, 0.f);
            // phase A must never lose a ray the exact clip accepts: check EVERY pixel of the
            // pose against the exact clip, inside and outside the projected pixel box
            std::vector<char> cand((size_t)N, 0);
            for (int local = 0; local < row.count; ++local) {
                int pix;
                float n_est;
                const bool maybe = brick_candidate(row, local, det_w, pix, n_est);
                if (pix != (row.i0 + local / row.w) * det_w + row.j0 + local % row.w) abort();
                if (!maybe) continue;
                cand[pix] = 1;
                // (a subsample: the pixel's bit decides after the conservative test, as in the kernel)
                if (pixel_mask && !((pixel_mask[pix >> 5] >> (pix & 31)) & 1u)) continue;
                auto &qk = queues[n_est < 14.f ? 0 : (n_est < 34.f ? 1 : 2)];
                qk.emplace_back(b, pix);
                if (qk.size() >= 64) {  // a full wave of hits of one length class: walk them
                    for (size_t k = qk.size() - 64; k < qk.size(); ++k)
                        item(qk[k].first, qk[k].second);
                    qk.resize(qk.size() - 64);
                }
            }
            for (int pix = 0; pix < N; ++pix) {
                if (cand[pix]) continue;
                const long r = (long)b * N + pix;
                float s[3], t[3];
                for (int a = 0; a < 3; ++a) {
                    s[a] = source[(long)b * 3 + a];
                    t[a] = target[r * 3 + a];
                }
                float I, rec[4];
                if (step_trace<false>(LdsFetch{brick.data()}, 0u, SG, s, t, voxel_shift, eps, I, rec) &&
                    I != 0.f)
                    abort();
            }
        }
        for (int k = 2; k >= 0; --k)
            for (auto &it : queues[k]) item(it.first, it.second);
    }
    if (q16 && ws_header && !ranges_valid) {
        ws_header[0] = n_fallback;
        ws_header[1] = bg.nx * bg.ny * bg.nz;
    }
    return 0;
}

// Host emulation of the volume-gradient brick kernel: per brick, every candidate of every
// pose scatters w * dalpha into a local padded accumulator, which is then stored.
int ddrr_siddon_backward_volume_bricks(int dx, int dy, int dz, const float *source,
                                       const float *target, const float *img,
                                       const float *grad_out, int B, int det_h, int det_w,
                                       float voxel_shift, float eps, float *g_volume, void *,
                                       void *) {
    const Dims D{dx, dy, dz};
    const int N = det_h * det_w;
    const BrickGrid bg = brick_grid(D);
    const BrickLayout lay{33, 32 * 33 + 1};
    std::vector<float> brick((size_t)brick_floats(lay));
    struct HostAddBytes {
        float *base;
        void operator()(unsigned off, float v) const { base[off >> 2] += v; }
    };
    for (int id = 0; id < bg.nx * bg.ny * bg.nz; ++id) {
        const Box box = brick_box(D, bg, id);
        const BrickGeom G = brick_geom(box, lay);
        std::fill(brick.begin(), brick.end(), 0.f);
        for (int b = 0; b < B; ++b) {
            const PoseGrid pg = pose_grid(source + (long)b * 3, target + (long)b * N * 3, det_h,
                                          det_w);
            const PixBox pb = project_brick_grid(pg, det_h, det_w, boxf(box), voxel_shift);
            const BrickRow row = brick_row(pg, pb, boxf(box), voxel_shift, eps, 0.f);
            for (int local = 0; local < row.count; ++local) {
                int pix;
                float n_est;
                if (!brick_candidate(row, local, det_w, pix, n_est)) continue;
                const long r = (long)b * N + pix;
                float s[3], t[3];
                for (int a = 0; a < 3; ++a) {
                    s[a] = source[(long)b * 3 + a];
                    t[a] = target[r * 3 + a];
                }
                const float w = grad_out[r] * (img ? img[r] : 1.f);
                if (w != 0.f)
                    step_scatter(HostAddBytes{brick.data()}, 0u, step_geom(box, lay), s, t, voxel_shift,
                                 eps, w);
            }
        }
        for (int x = box.lo[0]; x < box.hi[0]; ++x)
            for (int y = box.lo[1]; y < box.hi[1]; ++y)
                for (int z = box.lo[2]; z < box.hi[2]; ++z)
                    g_volume[((long)x * dy + y) * dz + z] =
                        brick[(x - box.lo[0]) * lay.sx + (y - box.lo[1]) * lay.sy + (z - box.lo[2])];
    }
    return 0;
}

// The channel render's volume gradient on the bricks: as above with the weight of the voxel's
// own label (step_scatter_weighted); the emulation keeps the labels in a plane of its own (the
// device packs them into the accumulator's words) and poisons the result first: every voxel must
// be stored exactly once.
int ddrr_siddon_backward_channels_volume_bricks(const unsigned char *labels, int dx, int dy, int dz,
                                                const float *source, const float *target,
                                                const float *img, const float *grad_out, int B,
                                                int det_h, int det_w, int C, float voxel_shift,
                                                float eps, float *g_volume, void *, void *) {
    const Dims D{dx, dy, dz};
    const int N = det_h * det_w;
    const BrickGrid bg = brick_grid(D);
    const BrickLayout lay{33, 32 * 33 + 1};
    std::vector<float> brick((size_t)brick_floats(lay));
    std::vector<unsigned char> lab((size_t)brick_floats(lay));
    for (size_t i = 0; i < (size_t)dx * dy * dz; ++i) g_volume[i] = NAN;
    for (int id = 0; id < bg.nx * bg.ny * bg.nz; ++id) {
        const Box box = brick_box(D, bg, id);
        std::fill(brick.begin(), brick.end(), 0.f);
        std::fill(lab.begin(), lab.end(), (unsigned char)0);
        for (int x = box.lo[0]; x < box.hi[0]; ++x)
            for (int y = box.lo[1]; y < box.hi[1]; ++y)
                for (int z = box.lo[2]; z < box.hi[2]; ++z)
                    lab[(x - box.lo[0]) * lay.sx + (y - box.lo[1]) * lay.sy + (z - box.lo[2])] =
                        labels[((long)x * dy + y) * dz + z];
        for (int b = 0; b < B; ++b) {
            const PoseGrid pg = pose_grid(source + (long)b * 3, target + (long)b * N * 3, det_h,
                                          det_w);
            const PixBox pb = project_brick_grid(pg, det_h, det_w, boxf(box), voxel_shift);
            const BrickRow row = brick_row(pg, pb, boxf(box), voxel_shift, eps, 0.f);
            for (int local = 0; local < row.count; ++local) {
                int pix;
                float n_est;
                if (!brick_candidate(row, local, det_w, pix, n_est)) continue;
                const long r = (long)b * N + pix;
                float s[3], t[3];
                for (int a = 0; a < 3; ++a) {
                    s[a] = source[(long)b * 3 + a];
                    t[a] = target[r * 3 + a];
                }
                const float *col = grad_out + (long)b * C * N + pix;
                step_scatter_weighted(
                    [&](unsigned off, float v) { brick[off >> 2] += v; },
                    [&](unsigned off) { return (unsigned)lab[off >> 2]; },
                    [&](unsigned l) { return l < (unsigned)C ? col[(long)l * N] : 0.f; }, 0u,
                    step_geom(box, lay), s, t, voxel_shift, eps, img ? img[r] : 1.f);
            }
        }
        for (int x = box.lo[0]; x < box.hi[0]; ++x)
            for (int y = box.lo[1]; y < box.hi[1]; ++y)
                for (int z = box.lo[2]; z < box.hi[2]; ++z)
                    g_volume[((long)x * dy + y) * dz + z] =
                        brick[(x - box.lo[0]) * lay.sx + (y - box.lo[1]) * lay.sy + (z - box.lo[2])];
    }
    return 0;
}

int ddrr_trilinear_forward_bricks(const float *volume, int dx, int dy, int dz,
                                  const float *source, const float *target, const float *img,
                                  int B, int det_h, int det_w, float voxel_shift, float eps,
                                  int n_points, const float *alphamin, const float *alphamax,
                                  float *out, float *aux, void *, void *) {
    const size_t R = (size_t)B * det_h * det_w;
    memset(out, 0, sizeof(float) * R);
    if (aux) memset(aux, 0, sizeof(float) * R * DDRR_TRI_AUX_PLANES);
    return tri_bricks_host(volume, dx, dy, dz, source, target, img, B, det_h, det_w, voxel_shift,
                           eps, n_points, *alphamin, *alphamax, out, aux);
}

int ddrr_trilinear_forward_channels_bricks(const float *volume, const unsigned char *labels,
                                           int dx, int dy, int dz, const float *source,
                                           const float *target, const float *img, int B,
                                           int det_h, int det_w, int C, float voxel_shift,
                                           float eps, int n_points, const float *alphamin,
                                           const float *alphamax, float *out, void *, void *) {
    memset(out, 0, sizeof(float) * (size_t)B * C * det_h * det_w);
    return tri_bricks_host(volume, dx, dy, dz, source, target, img, B, det_h, det_w, voxel_shift,
                           eps, n_points, *alphamin, *alphamax, out, nullptr, labels, C);
}

int ddrr_trilinear_backward_channels_bricks(const float *volume, const unsigned char *labels,
                                            int dx, int dy, int dz, const float *source,
                                            const float *target, const float *grad_out, int B,
                                            int det_h, int det_w, int C, float voxel_shift,
                                            float eps, int n_points, const float *alphamin,
                                            const float *alphamax, float *aux, void *, void *) {
    memset(aux, 0, sizeof(float) * (size_t)B * det_h * det_w * DDRR_TRI_AUX_PLANES);
    return tri_bricks_host(volume, dx, dy, dz, source, target, nullptr, B, det_h, det_w,
                           voxel_shift, eps, n_points, *alphamin, *alphamax, nullptr, aux, labels, C,
                           grad_out);
}

int ddrr_trilinear_backward_rays(const float *aux, const float *grad_out, const float *source,
                                 const float *target, const float *img, int B, int N, float eps,
                                 int n_points, const float *alphamin, const float *alphamax,
                                 float *g_source, float *g_target, float *g_img, float *g_alpha,
                                 void *) {
    const long R = (long)B * N;
    for (long r = 0; r < R; ++r) {
        const long b = r / N;
        const float *s = source + b * 3, *t = target + r * 3;
        const float A[3] = {aux[R + r], aux[2 * R + r], aux[3 * R + r]};
        const float Bv[3] = {aux[4 * R + r], aux[5 * R + r], aux[6 * R + r]};
        const float g = grad_out[r], L = img ? img[r] : 1.f;
        const MarchGrad m = trilinear_backward_from_record(aux[r], A, Bv, s, t, eps, n_points,
                                                           *alphamin, *alphamax, g * L);
        for (int a = 0; a < 3; ++a) {
            if (g_source) g_source[r * 3 + a] = m.gs[a];
            if (g_target) g_target[r * 3 + a] = m.gt[a];
        }
        if (g_img) g_img[r] = g * m.sumT * ((*alphamax - *alphamin) / (float)(n_points - 1));
        if (g_alpha) {
            g_alpha[r * 2 + 0] = m.g_amin;
            g_alpha[r * 2 + 1] = m.g_amax;
        }
    }
    return 0;
}

int ddrr_trilinear_backward_volume_bricks(int dx, int dy, int dz, const float *source,
                                          const float *target, const float *img,
                                          const float *grad_out, int B, int det_h, int det_w,
                                          float voxel_shift, float eps, int n_points,
                                          const float *alphamin, const float *alphamax,
                                          float *g_volume, void *, void *) {
    // poison: every voxel must be stored by exactly one owner brick
    for (size_t i = 0; i < (size_t)dx * dy * dz; ++i) g_volume[i] = NAN;
    return tri_owner_host(dx, dy, dz, source, target, img, grad_out, B, det_h, det_w, voxel_shift,
                          eps, n_points, *alphamin, *alphamax, g_volume);
}

int ddrr_trilinear_backward_channels_volume_bricks(const unsigned char *labels, int dx, int dy,
                                                   int dz, const float *source, const float *target,
                                                   const float *img, const float *grad_out, int B,
                                                   int det_h, int det_w, int C, float voxel_shift,
                                                   float eps, int n_points, const float *alphamin,
                                                   const float *alphamax, float *g_volume, void *,
                                                   void *) {
    for (size_t i = 0; i < (size_t)dx * dy * dz; ++i) g_volume[i] = NAN;
    return tri_owner_host(dx, dy, dz, source, target, img, grad_out, B, det_h, det_w, voxel_shift,
                          eps, n_points, *alphamin, *alphamax, g_volume, labels, C);
}

int ddrr_siddon_backward_rays(const float *aux, int aux_layout, const float *grad_out,
                              const float *source, int src_n, const float *target,
                              const float *img, int B, int N, float eps, int reduce_mode,
                              float *g_source, float *g_target, float *g_img, void *) {
    const long R = (long)B * N;
    for_each_ray(source, src_n, target, img, B, N, 0, 0, 0, 0,
                 [&](int, int, long r, const Ray &ray) {
                     float gs[3], gt[3];
                     float planar[SIDDON_AUX];
                     if (aux_layout != DDRR_AUX_INTERLEAVED)
                         planar_record(aux, aux_layout, R, r, planar);
                     const float *rec =
                         aux_layout != DDRR_AUX_INTERLEAVED ? planar : aux + r * SIDDON_AUX;
                     if (reduce_mode == DDRR_REDUCE_SUM)
                         siddon_backward_ray<REDUCE_SUM>(rec, ray.s, ray.t, eps,
                                                         grad_out[r] * ray.L, gs, gt);
                     else
                         siddon_backward_ray<REDUCE_MAX>(rec, ray.s, ray.t, eps,
                                                         grad_out[r] * ray.L, gs, gt);
                     for (int a = 0; a < 3; ++a) {
                         if (g_source) g_source[r * 3 + a] = gs[a];
                         if (g_target) g_target[r * 3 + a] = gt[a];
                     }
                     if (g_img) g_img[r] = grad_out[r] * rec[0];
                 });
    return 0;
}

int ddrr_siddon_backward_volume(const float *volume, int dx, int dy, int dz, const float *source,
                                int src_n, const float *target, const float *img,
                                const float *grad_out, int B, int N, float voxel_shift, float eps,
                                int reduce_mode, int det_h, int det_w, int tile_h, int tile_w,
                                float *g_volume, void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(source, src_n, target, img, B, N, det_h, det_w, tile_h, tile_w,
                 [&](int, int, long r, const Ray &ray) {
                     const float gl = grad_out[r] * ray.L;
                     if (reduce_mode == DDRR_REDUCE_SUM)
                         siddon_scatter_ray<REDUCE_SUM>(volume, D, ray.s, ray.t, voxel_shift, eps,
                                                        gl, HostAdd{g_volume});
                     else
                         siddon_scatter_ray<REDUCE_MAX>(volume, D, ray.s, ray.t, voxel_shift, eps,
                                                        gl, HostAdd{g_volume});
                 });
    return 0;
}

int ddrr_siddon_forward_channels(const float *volume, const unsigned char *labels, int dx, int dy,
                                 int dz, const float *source, int src_n, const float *target,
                                 const float *img, int B, int N, int C, float voxel_shift,
                                 float eps, int det_h, int det_w, int tile_h, int tile_w,
                                 float *out, void *) {
    const Dims D{dx, dy, dz};
    memset(out, 0, sizeof(float) * (size_t)B * C * N);
    for_each_ray(source, src_n, target, img, B, N, det_h, det_w, tile_h, tile_w,
                 [&](int b, int n, long, const Ray &ray) {
                     float *col = out + (long)b * C * N + n;
                     siddon_channels_ray(volume, labels, D, ray.s, ray.t, voxel_shift, eps,
                                         [&](int label, float run) {
                                             if (label < C) col[(long)label * N] += ray.L * run;
                                         });
                 });
    return 0;
}

// Host emulation of the channel render on the bricks: per brick, the packed value | label words
// are staged exactly as the kernel stages them, and every ray of every pose is clipped and
// walked with the kernel's own step_walk_channels.
long ddrr_channel_words_state_bytes(void) { return (long)(2 + 2 * kFingerprintWords) * 4; }

int ddrr_channel_words(const float *volume, const unsigned char *labels, long n_voxels, int C, float *words,
                       void *state_raw, int force, void *) {
    int *state = static_cast<int *>(state_raw);
    unsigned *fp = reinterpret_cast<unsigned *>(state + 2);
    bool stale = force != 0;
    for (int i = 0; i < kFingerprintWords && n_voxels > 0; ++i) {
        const long at = fingerprint_index(i, n_voxels);
        unsigned bits;
        memcpy(&bits, volume + at, 4);
        stale = stale || bits != fp[i] || (unsigned)labels[at] != fp[kFingerprintWords + i];
    }
    if (!stale) return 0;
    for (long i = 0; i < n_voxels; ++i) words[i] = pack_voxel_label_below(volume[i], labels[i], (unsigned)C);
    for (int i = 0; i < kFingerprintWords && n_voxels > 0; ++i) {
        const long at = fingerprint_index(i, n_voxels);
        memcpy(&fp[i], volume + at, 4);
        fp[kFingerprintWords + i] = (unsigned)labels[at];
    }
    state[1] += 1;
    return 0;
}

// (labels == NULL: `volume` holds the ready-packed words of ddrr_channel_words)
static int emu_channels_bricks(const float *volume, const unsigned char *labels, int dx,
                               int dy, int dz, const float *source, const float *target,
                               const float *img, int B, int det_h, int det_w, int C,
                               float voxel_shift, float eps, float *out) {
    const Dims D{dx, dy, dz};
    const int N = det_h * det_w;
    memset(out, 0, sizeof(float) * (size_t)B * C * N);
    const BrickGrid bg = brick_grid(D);
    const BrickLayout lay{33, 32 * 33 + 1};
    std::vector<float> brick((size_t)brick_floats(lay));
    for (int id = 0; id < bg.nx * bg.ny * bg.nz; ++id) {
        const Box box = brick_box(D, bg, id);
        const StepGeom SG = step_geom(box, lay);
        std::fill(brick.begin(), brick.end(), 0.f);
        for (int x = box.lo[0]; x < box.hi[0]; ++x)
            for (int y = box.lo[1]; y < box.hi[1]; ++y)
                for (int z = box.lo[2]; z < box.hi[2]; ++z) {
                    const long at = ((long)x * dy + y) * dz + z;
                    brick[(x - box.lo[0]) * lay.sx + (y - box.lo[1]) * lay.sy + (z - box.lo[2])] =
                        labels ? pack_voxel_label_below(volume[at], labels[at], (unsigned)C) : volume[at];
                }
        for (int b = 0; b < B; ++b)
            for (int pix = 0; pix < N; ++pix) {
                const long r = (long)b * N + pix;
                float s[3], t[3];
                for (int a = 0; a < 3; ++a) {
                    s[a] = source[(long)b * 3 + a];
                    t[a] = target[r * 3 + a];
                }
                const StepEntry E = step_enter(SG, s, t, voxel_shift, eps, 0u);
                if (!E.hit) continue;
                const float L = img ? img[r] : 1.f;
                float *col = out + (long)b * C * N + pix;
                step_walk_channels(LdsFetch{brick.data()}, SG, E, [&](unsigned lab, float run) {
                    col[(long)lab * N] += run;  // (labels >= C were staged as value 0 | label 0)
                }, L);
            }
    }
    return 0;
}

int ddrr_siddon_forward_channels_bricks(const float *volume, const unsigned char *labels, int dx,
                                        int dy, int dz, const float *source, const float *target,
                                        const float *img, int B, int det_h, int det_w, int C,
                                        float voxel_shift, float eps, float *out, void *,
                                        void *) {
    if (!labels) return -1;
    return emu_channels_bricks(volume, labels, dx, dy, dz, source, target, img, B, det_h, det_w, C, voxel_shift,
                               eps, out);
}

int ddrr_siddon_forward_channels_bricks_words(const float *words, int dx, int dy, int dz, const float *source,
                                              const float *target, const float *img, int B, int det_h,
                                              int det_w, int C, float voxel_shift, float eps, float *out,
                                              void *, void *) {
    return emu_channels_bricks(words, nullptr, dx, dy, dz, source, target, img, B, det_h, det_w, C, voxel_shift,
                               eps, out);
}

// Host emulation of the channel render's ray backward on the bricks: the blocked record of the
// volume weighted by grad_out[b, label, n], with the kernel's own step_walk_weighted.
int ddrr_siddon_backward_channels_bricks(const float *volume, const unsigned char *labels, int dx,
                                         int dy, int dz, const float *source, const float *target,
                                         const float *grad_out, int B, int det_h, int det_w, int C,
                                         float voxel_shift, float eps, float *aux, void *, void *) {
    const Dims D{dx, dy, dz};
    const int N = det_h * det_w;
    const long R = (long)B * N;
    memset(aux, 0, sizeof(float) * (size_t)rec_blocked_floats(R));
    const BrickGrid bg = brick_grid(D);
    const BrickLayout lay{33, 32 * 33 + 1};
    std::vector<float> brick((size_t)brick_floats(lay));
    for (int id = 0; id < bg.nx * bg.ny * bg.nz; ++id) {
        const Box box = brick_box(D, bg, id);
        const StepGeom SG = step_geom(box, lay);
        std::fill(brick.begin(), brick.end(), 0.f);
        for (int x = box.lo[0]; x < box.hi[0]; ++x)
            for (int y = box.lo[1]; y < box.hi[1]; ++y)
                for (int z = box.lo[2]; z < box.hi[2]; ++z) {
                    const long at = ((long)x * dy + y) * dz + z;
                    brick[(x - box.lo[0]) * lay.sx + (y - box.lo[1]) * lay.sy + (z - box.lo[2])] =
                        pack_voxel_label_below(volume[at], labels[at], (unsigned)C);
                }
        for (int b = 0; b < B; ++b)
            for (int pix = 0; pix < N; ++pix) {
                const long r = (long)b * N + pix;
                float s[3], t[3];
                for (int a = 0; a < 3; ++a) {
                    s[a] = source[(long)b * 3 + a];
                    t[a] = target[r * 3 + a];
                }
                const StepEntry E = step_enter(SG, s, t, voxel_shift, eps, 0u);
                if (!E.hit) continue;
                const float *col = grad_out + (long)b * C * N + pix;
                float I, rec[4];
                step_walk_weighted(LdsFetch{brick.data()}, SG, E,
                                   [&](unsigned lab) { return col[(long)lab * N]; }, I, rec);
                aux[rec_index(r, 0)] += I;
                for (int k = 0; k < 4; ++k) aux[rec_index(r, k + 1)] += rec[k];
            }
    }
    return 0;
}

int ddrr_siddon_backward_channels(const float *volume, const unsigned char *labels, int dx,
                                  int dy, int dz, const float *source, int src_n,
                                  const float *target, const float *img, const float *grad_out,
                                  int B, int N, int C, float voxel_shift, float eps, int det_h,
                                  int det_w, int tile_h, int tile_w, float *g_source,
                                  float *g_target, float *g_img, float *g_volume, void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(
        source, src_n, target, img, B, N, det_h, det_w, tile_h, tile_w,
        [&](int b, int n, long r, const Ray &ray) {
            const float *gcol = grad_out + (long)b * C * N + n;
            if (g_source || g_target || g_img) {
                float rec[SIDDON_AUX];
                auto fetch = [&](unsigned boff) {
                    const unsigned idx = boff >> 2;
                    const int lab = labels[idx];
                    return lab < C ? volume[idx] * gcol[(long)lab * N] : 0.f;
                };
                siddon_forward_ray_t<REDUCE_SUM, true, false>(fetch, global_store(D), full_box(D),
                                                              ray.s, ray.t, voxel_shift, eps, rec,
                                                              nullptr);
                float gs[3], gt[3];
                siddon_backward_ray<REDUCE_SUM>(rec, ray.s, ray.t, eps, ray.L, gs, gt);
                for (int a = 0; a < 3; ++a) {
                    if (g_source) g_source[r * 3 + a] = gs[a];
                    if (g_target) g_target[r * 3 + a] = gt[a];
                }
                if (g_img) g_img[r] = rec[0];
            }
            if (g_volume)
                siddon_scatter_ray<REDUCE_SUM>(volume, D, ray.s, ray.t, voxel_shift, eps, ray.L,
                                               [&](unsigned idx, float v) {
                                                   const int lab = labels[idx];
                                                   if (lab < C)
                                                       g_volume[idx] += v * gcol[(long)lab * N];
                                               });
        });
    return 0;
}

int ddrr_trilinear_backward_channels(const float *volume, const unsigned char *labels, int dx,
                                     int dy, int dz, const float *source, int src_n,
                                     const float *target, const float *img,
                                     const float *grad_out, int B, int N, int C,
                                     float voxel_shift, float eps, int n_points,
                                     const float *alphamin, const float *alphamax,
                                     int align_corners, int det_h, int det_w, int tile_h,
                                     int tile_w, float *g_source, float *g_target, float *g_img,
                                     float *g_alpha, float *g_volume, void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(
        source, src_n, target, img, B, N, det_h, det_w, tile_h, tile_w,
        [&](int b, int n, long r, const Ray &ray) {
            const bool ac = align_corners != 0;
            const LabelWeight wt{labels, D, grad_out + (long)b * C * N + n, N, C, *alphamin,
                                 voxel_shift, {ray.s[0], ray.s[1], ray.s[2]}, ac};
            MarchGrad m;
            if (g_volume)
                m = trilinear_backward_ray<false, true>(volume, D, ray.s, ray.t, voxel_shift, eps,
                                                        n_points, *alphamin, *alphamax, ac, ray.L,
                                                        HostAdd{g_volume}, wt);
            else
                m = trilinear_backward_ray<false, false>(volume, D, ray.s, ray.t, voxel_shift, eps,
                                                         n_points, *alphamin, *alphamax, ac, ray.L,
                                                         NoAdd{}, wt);
            for (int a = 0; a < 3; ++a) {
                if (g_source) g_source[r * 3 + a] = m.gs[a];
                if (g_target) g_target[r * 3 + a] = m.gt[a];
            }
            if (g_img) g_img[r] = m.sumT * ((*alphamax - *alphamin) / (float)(n_points - 1));
            if (g_alpha) {
                g_alpha[r * 2 + 0] = m.g_amin;
                g_alpha[r * 2 + 1] = m.g_amax;
            }
        });
    return 0;
}

int ddrr_siddon_backward_midpoint(const float *volume, int dx, int dy, int dz, const float *source,
                                  int src_n, const float *target, const float *img,
                                  const float *grad_out, int B, int N, float voxel_shift,
                                  float eps, int lookup_mode, int align_corners, float *g_source,
                                  float *g_target, float *g_img, float *g_volume, void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(
        source, src_n, target, img, B, N, 0, 0, 1, 64, [&](int, int, long r, const Ray &ray) {
            const float g = grad_out[r];
            const bool ac = align_corners != 0;
            float gs[3], gt[3], I;
            if (lookup_mode == DDRR_LOOKUP_MID_TRILINEAR)
                I = g_volume ? siddon_backward_ray_midpoint<LOOKUP_MID_TRILINEAR, true>(
                                   volume, D, ray.s, ray.t, voxel_shift, eps, ac, g * ray.L, gs, gt,
                                   HostAdd{g_volume})
                             : siddon_backward_ray_midpoint<LOOKUP_MID_TRILINEAR, false>(
                                   volume, D, ray.s, ray.t, voxel_shift, eps, ac, g * ray.L, gs, gt,
                                   NoAdd{});
            else
                I = g_volume ? siddon_backward_ray_midpoint<LOOKUP_MID_NEAREST, true>(
                                   volume, D, ray.s, ray.t, voxel_shift, eps, ac, g * ray.L, gs, gt,
                                   HostAdd{g_volume})
                             : siddon_backward_ray_midpoint<LOOKUP_MID_NEAREST, false>(
                                   volume, D, ray.s, ray.t, voxel_shift, eps, ac, g * ray.L, gs, gt,
                                   NoAdd{});
            for (int a = 0; a < 3; ++a) {
                if (g_source) g_source[r * 3 + a] = gs[a];
                if (g_target) g_target[r * 3 + a] = gt[a];
            }
            if (g_img) g_img[r] = g * I;
        });
    return 0;
}

int ddrr_siddon_segments(const float *volume, int dx, int dy, int dz, const float *source,
                         int src_n, const float *target, const float *img, int B, int N,
                         float voxel_shift, float eps, float *terms, void *) {
    const Dims D{dx, dy, dz};
    const long M1 = (long)dx + dy + dz + 2;
    for_each_ray(source, src_n, target, img, B, N, 0, 0, 1, 64,
                 [&](int b, int n, long, const Ray &ray) {
                     siddon_segments_ray(volume, D, ray.s, ray.t, voxel_shift, eps, ray.L,
                                         terms + (long)b * M1 * N + n, N);
                 });
    return 0;
}

int ddrr_siddon_segments_backward(const float *volume, int dx, int dy, int dz, const float *source,
                                  int src_n, const float *target, const float *img,
                                  const float *grad_terms, int B, int N, float voxel_shift,
                                  float eps, float *g_source, float *g_target, float *g_img,
                                  float *g_volume, void *) {
    const Dims D{dx, dy, dz};
    const long M1 = (long)dx + dy + dz + 2;
    for_each_ray(source, src_n, target, img, B, N, 0, 0, 1, 64,
                 [&](int b, int n, long r, const Ray &ray) {
                     float gs[3], gt[3], gi;
                     const float *g = grad_terms + (long)b * M1 * N + n;
                     if (g_volume)
                         siddon_segments_backward_ray<true>(volume, D, ray.s, ray.t, voxel_shift,
                                                            eps, ray.L, g, N, gs, gt, gi,
                                                            HostAdd{g_volume});
                     else
                         siddon_segments_backward_ray<false>(volume, D, ray.s, ray.t, voxel_shift,
                                                             eps, ray.L, g, N, gs, gt, gi, NoAdd{});
                     for (int a = 0; a < 3; ++a) {
                         if (g_source) g_source[r * 3 + a] = gs[a];
                         if (g_target) g_target[r * 3 + a] = gt[a];
                     }
                     if (g_img) g_img[r] = gi;
                 });
    return 0;
}

int ddrr_trilinear_backward_max(const float *volume, int dx, int dy, int dz, const float *source,
                                int src_n, const float *target, const float *img,
                                const float *grad_out, int B, int N, float voxel_shift, float eps,
                                int n_points, const float *alphamin, const float *alphamax,
                                int mode_nearest, int align_corners, float *g_source,
                                float *g_target, float *g_img, float *g_alpha, float *g_volume,
                                void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(
        source, src_n, target, img, B, N, 0, 0, 1, 64, [&](int, int, long r, const Ray &ray) {
            const float g = grad_out[r];
            const bool ac = align_corners != 0;
            MarchGrad m;
            if (mode_nearest) {
                const OneSampleWeight wt{trilinear_argmax_ray<true>(
                    volume, D, ray.s, ray.t, voxel_shift, eps, n_points, *alphamin, *alphamax, ac)};
                if (g_volume)
                    m = trilinear_backward_ray<true, true>(volume, D, ray.s, ray.t, voxel_shift,
                                                           eps, n_points, *alphamin, *alphamax, ac,
                                                           g * ray.L, HostAdd{g_volume}, wt);
                else
                    m = trilinear_backward_ray<true, false>(volume, D, ray.s, ray.t, voxel_shift,
                                                            eps, n_points, *alphamin, *alphamax,
                                                            ac, g * ray.L, NoAdd{}, wt);
            } else {
                const OneSampleWeight wt{trilinear_argmax_ray<false>(
                    volume, D, ray.s, ray.t, voxel_shift, eps, n_points, *alphamin, *alphamax, ac)};
                if (g_volume)
                    m = trilinear_backward_ray<false, true>(volume, D, ray.s, ray.t, voxel_shift,
                                                            eps, n_points, *alphamin, *alphamax,
                                                            ac, g * ray.L, HostAdd{g_volume}, wt);
                else
                    m = trilinear_backward_ray<false, false>(volume, D, ray.s, ray.t, voxel_shift,
                                                             eps, n_points, *alphamin, *alphamax,
                                                             ac, g * ray.L, NoAdd{}, wt);
            }
            for (int a = 0; a < 3; ++a) {
                if (g_source) g_source[r * 3 + a] = m.gs[a];
                if (g_target) g_target[r * 3 + a] = m.gt[a];
            }
            if (g_img)
                g_img[r] = g * m.sumT * ((*alphamax - *alphamin) / (float)(n_points - 1));
            if (g_alpha) {
                g_alpha[r * 2 + 0] = m.g_amin;
                g_alpha[r * 2 + 1] = m.g_amax;
            }
        });
    return 0;
}

int ddrr_trilinear_samples(const float *volume, int dx, int dy, int dz, const float *source,
                           int src_n, const float *target, const float *img, int B, int N,
                           float voxel_shift, float eps, int n_points, const float *alphamin,
                           const float *alphamax, int mode_nearest, int align_corners,
                           float *samples, void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(source, src_n, target, img, B, N, 0, 0, 1, 64,
                 [&](int b, int n, long, const Ray &ray) {
                     float *col = samples + (long)b * n_points * N + n;
                     if (mode_nearest)
                         trilinear_samples_ray<true>(volume, D, ray.s, ray.t, voxel_shift, eps,
                                                     n_points, *alphamin, *alphamax,
                                                     align_corners != 0, ray.L, col, N);
                     else
                         trilinear_samples_ray<false>(volume, D, ray.s, ray.t, voxel_shift, eps,
                                                      n_points, *alphamin, *alphamax,
                                                      align_corners != 0, ray.L, col, N);
                 });
    return 0;
}

int ddrr_trilinear_samples_backward(const float *volume, int dx, int dy, int dz,
                                    const float *source, int src_n, const float *target,
                                    const float *img, const float *grad_samples, int B, int N,
                                    float voxel_shift, float eps, int n_points,
                                    const float *alphamin, const float *alphamax,
                                    int mode_nearest, int align_corners, float *g_source,
                                    float *g_target, float *g_img, float *g_alpha,
                                    float *g_volume, void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(
        source, src_n, target, img, B, N, 0, 0, 1, 64, [&](int b, int n, long r, const Ray &ray) {
            const bool ac = align_corners != 0;
            const SampleWeight wt{grad_samples + (long)b * n_points * N + n, N};
            MarchGrad m;
            if (mode_nearest) {
                if (g_volume)
                    m = trilinear_backward_ray<true, true>(volume, D, ray.s, ray.t, voxel_shift,
                                                           eps, n_points, *alphamin, *alphamax, ac,
                                                           ray.L, HostAdd{g_volume}, wt);
                else
                    m = trilinear_backward_ray<true, false>(volume, D, ray.s, ray.t, voxel_shift,
                                                            eps, n_points, *alphamin, *alphamax,
                                                            ac, ray.L, NoAdd{}, wt);
            } else {
                if (g_volume)
                    m = trilinear_backward_ray<false, true>(volume, D, ray.s, ray.t, voxel_shift,
                                                            eps, n_points, *alphamin, *alphamax,
                                                            ac, ray.L, HostAdd{g_volume}, wt);
                else
                    m = trilinear_backward_ray<false, false>(volume, D, ray.s, ray.t, voxel_shift,
                                                             eps, n_points, *alphamin, *alphamax,
                                                             ac, ray.L, NoAdd{}, wt);
            }
            for (int a = 0; a < 3; ++a) {
                if (g_source) g_source[r * 3 + a] = m.gs[a];
                if (g_target) g_target[r * 3 + a] = m.gt[a];
            }
            if (g_img) g_img[r] = m.sumT * ((*alphamax - *alphamin) / (float)(n_points - 1));
            if (g_alpha) {
                g_alpha[r * 2 + 0] = m.g_amin;
                g_alpha[r * 2 + 1] = m.g_amax;
            }
        });
    return 0;
}

int ddrr_trilinear_forward_channels(const float *volume, const unsigned char *labels, int dx,
                                    int dy, int dz, const float *source, int src_n,
                                    const float *target, const float *img, int B, int N, int C,
                                    float voxel_shift, float eps, int n_points,
                                    const float *alphamin, const float *alphamax,
                                    int align_corners, int det_h, int det_w, int tile_h,
                                    int tile_w, float *out, void *) {
    const Dims D{dx, dy, dz};
    memset(out, 0, sizeof(float) * (size_t)B * C * N);
    for_each_ray(source, src_n, target, img, B, N, det_h, det_w, tile_h, tile_w,
                 [&](int b, int n, long, const Ray &ray) {
                     float *col = out + (long)b * C * N + n;
                     trilinear_channels_ray(volume, labels, D, ray.s, ray.t, voxel_shift, eps,
                                            n_points, *alphamin, *alphamax, align_corners != 0,
                                            [&](int label, float run) {
                                                if (label < C) col[(long)label * N] += ray.L * run;
                                            });
                 });
    return 0;
}

int ddrr_trilinear_alpha_range(const float *source, int src_n, const float *target, int B, int N,
                               int dx, int dy, int dz, float voxel_shift, float eps,
                               float *range2, void *) {
    const Dims D{dx, dy, dz};
    float lo = INFINITY, hi = -INFINITY;
    for (long r = 0; r < (long)B * N; ++r) {
        const long b = r / N, n = r - b * N;
        const float *s = source + (b * src_n + (src_n == 1 ? 0 : n)) * 3, *t = target + r * 3;
        float a0, a1;
        ray_alpha_range(D, s, t, voxel_shift, eps, a0, a1);
        lo = fminf(lo, a0);
        hi = fmaxf(hi, a1);
    }
    range2[0] = lo;
    range2[1] = hi;
    return 0;
}

int ddrr_trilinear_forward(const float *volume, int dx, int dy, int dz, const float *source,
                           int src_n, const float *target, const float *img, int B, int N,
                           float voxel_shift, float eps, int n_points, const float *alphamin,
                           const float *alphamax, int mode_nearest, int reduce_mode,
                           int align_corners, int det_h, int det_w, int tile_h, int tile_w,
                           float *out, void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(
        source, src_n, target, img, B, N, det_h, det_w, tile_h, tile_w,
        [&](int, int, long r, const Ray &ray) {
            float I;
            const bool sum = reduce_mode == DDRR_REDUCE_SUM;
            const bool ac = align_corners != 0;
            if (mode_nearest)
                I = sum ? trilinear_forward_ray<REDUCE_SUM, true>(volume, D, ray.s, ray.t,
                                                                  voxel_shift, eps, n_points,
                                                                  *alphamin, *alphamax, ac)
                        : trilinear_forward_ray<REDUCE_MAX, true>(volume, D, ray.s, ray.t,
                                                                  voxel_shift, eps, n_points,
                                                                  *alphamin, *alphamax, ac);
            else
                I = sum ? trilinear_forward_ray<REDUCE_SUM, false>(volume, D, ray.s, ray.t,
                                                                   voxel_shift, eps, n_points,
                                                                   *alphamin, *alphamax, ac)
                        : trilinear_forward_ray<REDUCE_MAX, false>(volume, D, ray.s, ray.t,
                                                                   voxel_shift, eps, n_points,
                                                                   *alphamin, *alphamax, ac);
            out[r] = ray.L * I;
        });
    return 0;
}

int ddrr_trilinear_backward(const float *volume, int dx, int dy, int dz, const float *source,
                            int src_n, const float *target, const float *img,
                            const float *grad_out, int B, int N, float voxel_shift, float eps,
                            int n_points, const float *alphamin, const float *alphamax,
                            int mode_nearest, int align_corners, int det_h, int det_w, int tile_h,
                            int tile_w, float *g_source, float *g_target, float *g_img,
                            float *g_alpha, float *g_volume, void *) {
    const Dims D{dx, dy, dz};
    for_each_ray(
        source, src_n, target, img, B, N, det_h, det_w, tile_h, tile_w,
        [&](int, int, long r, const Ray &ray) {
            const float g = grad_out[r];
            const bool ac = align_corners != 0;
            MarchGrad m;
            if (mode_nearest) {
                if (g_volume)
                    m = trilinear_backward_ray<true, true>(volume, D, ray.s, ray.t, voxel_shift,
                                                           eps, n_points, *alphamin, *alphamax, ac,
                                                           g * ray.L, HostAdd{g_volume});
                else
                    m = trilinear_backward_ray<true, false>(volume, D, ray.s, ray.t, voxel_shift,
                                                            eps, n_points, *alphamin, *alphamax,
                                                            ac, g * ray.L, NoAdd{});
            } else {
                if (g_volume)
                    m = trilinear_backward_ray<false, true>(volume, D, ray.s, ray.t, voxel_shift,
                                                            eps, n_points, *alphamin, *alphamax,
                                                            ac, g * ray.L, HostAdd{g_volume});
                else
                    m = trilinear_backward_ray<false, false>(volume, D, ray.s, ray.t, voxel_shift,
                                                             eps, n_points, *alphamin, *alphamax,
                                                             ac, g * ray.L, NoAdd{});
            }
            for (int a = 0; a < 3; ++a) {
                if (g_source) g_source[r * 3 + a] = m.gs[a];
                if (g_target) g_target[r * 3 + a] = m.gt[a];
            }
            if (g_img)
                g_img[r] = g * m.sumT * ((*alphamax - *alphamin) / (float)(n_points - 1));
            if (g_alpha) {
                g_alpha[r * 2 + 0] = m.g_amin;
                g_alpha[r * 2 + 1] = m.g_amax;
            }
        });
    return 0;
}

int ddrr_raygen_forward(const float *Mw, const float *Ainv, const float *P, int B, int N,
                        float *source_v, float *target_v, float *img, void *) {
    for (int b = 0; b < B; ++b) {
        const float *M = Mw + (long)b * 12;
        const float sw[3] = {M[3], M[7], M[11]};
        apply34(Ainv, sw, source_v + b * 3);
        for (int n = 0; n < N; ++n) {
            const RayGenOut o = raygen_ray(M, Ainv, P + (long)n * 3);
            const long r = (long)b * N + n;
            for (int a = 0; a < 3; ++a) target_v[r * 3 + a] = o.tv[a];
            img[r] = o.L;
        }
    }
    return 0;
}

int ddrr_siddon_backward_pose(const float *aux, int aux_layout, const float *grad_out,
                              const float *source_v, const float *target_v, const float *img,
                              const float *Mw, const float *Ainv, const float *P, int B, int N,
                              float eps, int with_img_path, float *gMw, void *) {
    const long R = (long)B * N;
    for (int b = 0; b < B; ++b) {
        float acc[12] = {0};
        const float *s = source_v + b * 3;
        for (int n = 0; n < N; ++n) {
            const long r = (long)b * N + n;
            float rec[SIDDON_AUX];
            if (aux_layout != DDRR_AUX_INTERLEAVED) {
                planar_record(aux, aux_layout, R, r, rec);
            } else {
                memcpy(rec, aux + r * SIDDON_AUX, sizeof(rec));
            }
            float gs[3], gt[3];
            siddon_backward_ray<REDUCE_SUM>(rec, s, target_v + r * 3, eps, grad_out[r] * img[r],
                                            gs, gt);
            raygen_ray_adjoint(Mw + (long)b * 12, Ainv, P + (long)n * 3, gt, gs,
                               with_img_path ? grad_out[r] * rec[0] : 0.f, img[r], acc);
        }
        memcpy(gMw + (long)b * 12, acc, sizeof(acc));
    }
    return 0;
}

int ddrr_sobel_forward(const float *img, int B, int H, int W, float *out, void *) {
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j)
                sobel_pixel(img + (long)b * H * W, H, W, i, j, out[((long)b * 2) * H * W + i * W + j],
                            out[((long)b * 2 + 1) * H * W + i * W + j]);
    return 0;
}

int ddrr_sobel_backward(const float *g, int B, int H, int W, float *g_img, void *) {
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j)
                g_img[(long)b * H * W + i * W + j] = sobel_pixel_adjoint(
                    g + ((long)b * 2) * H * W, g + ((long)b * 2 + 1) * H * W, H, W, i, j);
    return 0;
}

// blur_core.h's passes over whole images (the device does them per tile through LDS, pose_ncc.hip)
int ddrr_blur_sobel_forward(const float *img, long img_stride, int B, int H, int W, const float *taps, int k,
                            float *out, void *) {
    if (k < 1 || k > kBlurMaxTaps || !(k & 1) || (k >> 1) >= H || (k >> 1) >= W) return -1;
    const int r = k >> 1;
    std::vector<float> tmp((size_t)H * W), bl((size_t)H * W);
    for (int b = 0; b < B; ++b) {
        const float *src = img + b * img_stride;
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                float s = 0.f;
                for (int t = 0; t < k; ++t) s = fmaf(taps[t], src[(long)i * W + blur_reflect(j + t - r, W)], s);
                tmp[(size_t)i * W + j] = s;
            }
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                float s = 0.f;
                for (int t = 0; t < k; ++t) s = fmaf(taps[t], tmp[(size_t)blur_reflect(i + t - r, H) * W + j], s);
                bl[(size_t)i * W + j] = s;
            }
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j)
                sobel_pixel(bl.data(), H, W, i, j, out[((long)b * 2) * H * W + i * W + j],
                            out[((long)b * 2 + 1) * H * W + i * W + j]);
    }
    return 0;
}

int ddrr_blur_sobel_backward(const float *g, int B, int H, int W, const float *taps, int k, float *g_img, void *) {
    if (k < 1 || k > kBlurMaxTaps || !(k & 1) || (k >> 1) >= H || (k >> 1) >= W) return -1;
    const int r = k >> 1;
    std::vector<float> gb((size_t)H * W), tmp((size_t)H * W);
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j)
                gb[(size_t)i * W + j] = sobel_pixel_adjoint(g + ((long)b * 2) * H * W, g + ((long)b * 2 + 1) * H * W,
                                                            H, W, i, j);
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                float s = 0.f;
                for (int o = i - r < 0 ? 0 : i - r; o <= i + r && o < H; ++o)
                    s = fmaf(blur_adjoint_weight(taps, k, i, o, H), gb[(size_t)o * W + j], s);
                tmp[(size_t)i * W + j] = s;
            }
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                float s = 0.f;
                for (int o = j - r < 0 ? 0 : j - r; o <= j + r && o < W; ++o)
                    s = fmaf(blur_adjoint_weight(taps, k, j, o, W), tmp[(size_t)i * W + o], s);
                g_img[(long)b * H * W + i * W + j] = s;
            }
    }
    return 0;
}

int ddrr_ncc_forward(const float *x1, long x1_stride, const float *x2, int B, int N, float eps,
                     float *out, float *stats, void *) {
    for (int b = 0; b < B; ++b) {
        const float *p1 = x1 + b * x1_stride, *p2 = x2 + (long)b * N;
        double a1 = 0, a2 = 0;
        for (int n = 0; n < N; ++n) a1 += p1[n], a2 += p2[n];
        const float mu1 = (float)(a1 / N), mu2 = (float)(a2 / N);
        double v1 = 0, v2 = 0, c12 = 0;
        for (int n = 0; n < N; ++n) {
            const double d1 = p1[n] - mu1, d2 = p2[n] - mu2;
            v1 += d1 * d1, v2 += d2 * d2, c12 += d1 * d2;
        }
        const float s1 = sqrtf((float)(v1 / N) + eps), s2 = sqrtf((float)(v2 / N) + eps);
        const float ncc = (float)(c12 / N) / (s1 * s2);
        out[b] = ncc;
        const float st[5] = {mu1, s1, mu2, s2, ncc};
        memcpy(stats + b * 5, st, sizeof(st));
    }
    return 0;
}

int ddrr_ncc_backward(const float *x1, long x1_stride, const float *x2, const float *stats,
                      const float *g_out, int g_stride, int B, int N, float *g_x1, float *g_x2,
                      void *) {
    for (int b = 0; b < B; ++b) {
        const float *st = stats + b * 5;
        const float g = g_out[b * g_stride] / (float)N;
        for (int n = 0; n < N; ++n) {
            const float z1 = (x1[b * x1_stride + n] - st[0]) / st[1];
            const float z2 = (x2[(long)b * N + n] - st[2]) / st[3];
            if (g_x2) g_x2[(long)b * N + n] = g * (z1 - z2 * st[4]) / st[3];
            if (g_x1) g_x1[(long)b * N + n] = g * (z2 - z1 * st[4]) / st[1];
        }
    }
    return 0;
}

int ddrr_ncc_patch_forward(const float *x1, long x1_stride, const float *x2, int B, int H, int W, int p,
                           float eps, float *out, float *coef, void *) {
    if (p < 1 || p > H || p > W || p > 64) return -1;
    const int hw = H - p + 1, ww = W - p + 1;
    for (int b = 0; b < B; ++b) {
        const float *a = x1 + b * x1_stride, *m = x2 + (long)b * H * W;
        double total = 0.;
        for (int wy = 0; wy < hw; ++wy)
            for (int wx = 0; wx < ww; ++wx) {
                float c[4];
                total += ncc_patch_window([&](int y, int x) { return a[(long)(wy + y) * W + wx + x]; },
                                          [&](int y, int x) { return m[(long)(wy + y) * W + wx + x]; }, p, eps, c);
                if (coef) memcpy(coef + (((long)b * hw + wy) * ww + wx) * 4, c, sizeof(c));
            }
        out[b] = (float)(total / ((double)hw * ww));
    }
    return 0;
}

int ddrr_ncc_patch_backward(const float *x1, long x1_stride, const float *x2, const float *coef,
                            const float *g_out, int g_stride, int B, int H, int W, int p, float *g_x2,
                            void *) {
    if (p < 1 || p > H || p > W || p > 64) return -1;
    const int hw = H - p + 1, ww = W - p + 1;
    for (int b = 0; b < B; ++b) {
        const float g = g_out[b * g_stride] / ((float)hw * (float)ww * (float)(p * p));
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float s = ncc_patch_pixel_grad(
                    [&](int wy, int wx, int k) {
                        return (wy >= 0 && wx >= 0 && wy < hw && wx < ww)
                                   ? coef[(((long)b * hw + wy) * ww + wx) * 4 + k] : 0.f;
                    },
                    y, x, p, x1[b * x1_stride + (long)y * W + x], x2[((long)b * H + y) * W + x]);
                g_x2[((long)b * H + y) * W + x] = g * s;
            }
    }
    return 0;
}

int ddrr_pose_euler_forward(const float *rot, const float *xyz, int a0, int a1, int a2,
                            const float *reorient34, int B, float *Mw, void *) {
    const int axes[3] = {a0, a1, a2};
    for (int b = 0; b < B; ++b)
        pose_euler_forward(rot + b * 3, xyz + b * 3, axes, reorient34, Mw + b * 12);
    return 0;
}

int ddrr_pose_euler_backward(const float *rot, const float *xyz, int a0, int a1, int a2,
                             const float *reorient34, const float *gMw, int B, float *g_rot,
                             float *g_xyz, void *) {
    const int axes[3] = {a0, a1, a2};
    for (int b = 0; b < B; ++b)
        pose_euler_backward(rot + b * 3, xyz + b * 3, axes, reorient34, gMw + b * 12, g_rot + b * 3,
                            g_xyz + b * 3);
    return 0;
}

// ---- the fused registration step (csrc/pose_ncc.hip): on the host simply the entries it fuses,
// one after the other (the workspace stays untouched: zero in, zero out)
int ddrr_pose_adam_step(float *rot, float *xyz, const float *g_rot, const float *g_xyz, float *m_rot,
                        float *v_rot, float *m_xyz, float *v_xyz, float *step_rot, float *step_xyz, int B,
                        float lr_rot, float lr_xyz, float beta1, float beta2, float eps, int maximize, void *) {
    if (!rot || !xyz || !g_rot || !g_xyz || !m_rot || !v_rot || !m_xyz || !v_xyz || !step_rot || !step_xyz)
        return -1;
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f) || !(lr_rot >= 0.f) ||
        !(lr_xyz >= 0.f) || B < 0)
        return -1;
    for (int k = 0; k < 2; ++k) {
        float *p = k ? xyz : rot, *m = k ? m_xyz : m_rot, *v = k ? v_xyz : v_rot, *st = k ? step_xyz : step_rot;
        const float *gp = k ? g_xyz : g_rot;
        if (B == 0) continue;
        const float step = st[0] + 1.f;
        const float ss = (k ? lr_xyz : lr_rot) / (1.f - powf(beta1, step));
        const float rb2 = 1.f / sqrtf(1.f - powf(beta2, step));
        for (int j = 0; j < 3 * B; ++j) {
            const float g = maximize ? -gp[j] : gp[j];
            m[j] = fmaf(g - m[j], 1.f - beta1, m[j]);
            v[j] = fmaf(beta2, v[j], (1.f - beta2) * g * g);
            p[j] -= ss * m[j] / fmaf(sqrtf(v[j]), rb2, eps);
        }
        st[0] = step;
    }
    return 0;
}

long ddrr_siddon_ncc_workspace_bytes(int B) { return B < 1 ? 0 : (long)B * (5 * 8 + 12 * 4 + 2 * 4) + 16; }

int ddrr_pose_raygen_forward(const float *rot, const float *xyz, int a0, int a1, int a2,
                             const float *reorient34, const float *Ainv, const float *P, int B, int N,
                             float *Mw, float *source_v, float *target_v, float *img, float *clear,
                             long clear_floats, void *clear_launch_ws, void *st) {
    if (clear_floats < 0 || (clear_floats > 0 && !clear)) return -1;
    if ((clear_floats > 0 || clear_launch_ws) && (B == 0 || N == 0)) return -1;
    if (clear_floats > 0) memset(clear, 0, sizeof(float) * (size_t)clear_floats);
    if (clear_launch_ws) memset(clear_launch_ws, 0, 16);
    if (int rc = ddrr_pose_euler_forward(rot, xyz, a0, a1, a2, reorient34, B, Mw, st)) return rc;
    return ddrr_raygen_forward(Mw, Ainv, P, B, N, source_v, target_v, img, st);
}

int ddrr_siddon_ncc_forward(const float *aux, const float *img, const float *x1, long x1_stride, int B,
                            int N, float eps, void *, float *ncc, float *stats, float *out, float *ncc_sum,
                            void *st) {
    std::vector<float> x2((size_t)B * N);
    for (long r = 0; r < (long)B * N; ++r) x2[r] = img[r] * aux[rec_index(r, 0)];
    if (out) memcpy(out, x2.data(), sizeof(float) * x2.size());
    if (int rc = ddrr_ncc_forward(x1, x1_stride, x2.data(), B, N, eps, ncc, stats, st)) return rc;
    if (ncc_sum) {
        double total = 0.;
        for (int b = 0; b < B; ++b) total += (double)ncc[b];
        *ncc_sum = (float)total;
    }
    return 0;
}

int ddrr_siddon_ncc_backward_pose(const float *aux, const float *img, const float *x1, long x1_stride,
                                  const float *stats, const float *g_out, int g_stride,
                                  const float *source_v, const float *target_v, const float *Mw,
                                  const float *Ainv, const float *P, const float *rot, const float *xyz,
                                  int a0, int a1, int a2, const float *reorient34, int B, int N,
                                  float eps, int with_img_path, void *, float *g_rot, float *g_xyz,
                                  void *st) {
    std::vector<float> x2((size_t)B * N), g_x2((size_t)B * N), gMw((size_t)B * 12);
    for (long r = 0; r < (long)B * N; ++r) x2[r] = img[r] * aux[rec_index(r, 0)];
    if (int rc = ddrr_ncc_backward(x1, x1_stride, x2.data(), stats, g_out, g_stride, B, N, nullptr,
                                   g_x2.data(), st))
        return rc;
    if (int rc = ddrr_siddon_backward_pose(aux, DDRR_AUX_BLOCKED, g_x2.data(), source_v, target_v, img, Mw,
                                           Ainv, P, B, N, eps, with_img_path, gMw.data(), st))
        return rc;
    return ddrr_pose_euler_backward(rot, xyz, a0, a1, a2, reorient34, gMw.data(), B, g_rot, g_xyz, st);
}

int ddrr_siddon_backward_pose_euler(const float *aux, const float *grad_out, const float *source_v,
                                    const float *Mw, const float *Ainv, const float *P, const float *rot,
                                    const float *xyz, int a0, int a1, int a2, const float *reorient34, int B,
                                    int N, float eps, int with_img_path, void *, float *g_rot, float *g_xyz,
                                    void *st) {
    // (the rays' targets and lengths as the device regenerates them: raygen_core.h raygen_ray)
    std::vector<float> target((size_t)B * N * 3), img((size_t)B * N), gMw((size_t)B * 12);
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const RayGenOut ray = raygen_ray(Mw + (long)b * 12, Ainv, P + (long)n * 3);
            for (int a = 0; a < 3; ++a) target[((long)b * N + n) * 3 + a] = ray.tv[a];
            img[(long)b * N + n] = ray.L;
        }
    if (int rc = ddrr_siddon_backward_pose(aux, DDRR_AUX_BLOCKED, grad_out, source_v, target.data(), img.data(), Mw,
                                           Ainv, P, B, N, eps, with_img_path, gMw.data(), st))
        return rc;
    return ddrr_pose_euler_backward(rot, xyz, a0, a1, a2, reorient34, gMw.data(), B, g_rot, g_xyz, st);
}

// ---- double precision (csrc/f64_rays.hip): the same per-ray cores, host loops
static void ray64(const double *source, int src_n, const double *target, long r, int N, double s[3],
                  double t[3]) {
    const long b = r / N, n = r - b * N;
    for (int a = 0; a < 3; ++a) {
        s[a] = source[(b * src_n + (src_n == 1 ? 0 : n)) * 3 + a];
        t[a] = target[r * 3 + a];
    }
}

int ddrr_siddon_forward_f64(const double *volume, int dx, int dy, int dz, const double *source,
                            int src_n, const double *target, const double *img, int B, int N,
                            double voxel_shift, double eps, int reduce_mode, double *out,
                            double *aux, void *) {
    const Dims D{dx, dy, dz};
    for (long r = 0; r < (long)B * N; ++r) {
        double s[3], t[3];
        ray64(source, src_n, target, r, N, s, t);
        out[r] = (img ? img[r] : 1.0) *
                 ddrr64::siddon_forward_ray(volume, D, s, t, voxel_shift, eps,
                                            reduce_mode == DDRR_REDUCE_MAX,
                                            aux ? aux + r * ddrr64::kAux : nullptr);
    }
    return 0;
}

int ddrr_siddon_backward_f64(int dx, int dy, int dz, const double *source, int src_n,
                             const double *target, const double *img, const double *grad_out,
                             const double *aux, int B, int N, double voxel_shift, double eps,
                             double *g_source, double *g_target, double *g_img, double *g_volume,
                             void *) {
    const Dims D{dx, dy, dz};
    for (long r = 0; r < (long)B * N; ++r) {
        double s[3], t[3];
        ray64(source, src_n, target, r, N, s, t);
        const double g = grad_out[r], L = img ? img[r] : 1.0;
        if (aux && (g_source || g_target || g_img)) {
            double gs[3], gt[3];
            ddrr64::siddon_backward_ray(aux + r * ddrr64::kAux, s, t, eps, g * L, gs, gt);
            for (int a = 0; a < 3; ++a) {
                if (g_source) g_source[r * 3 + a] = gs[a];
                if (g_target) g_target[r * 3 + a] = gt[a];
            }
            if (g_img) g_img[r] = g * aux[r * ddrr64::kAux];
        }
        if (g_volume && g * L != 0.0)
            ddrr64::siddon_scatter_ray(D, s, t, voxel_shift, eps, g * L,
                                       [&](long idx, double v) { g_volume[idx] += v; });
    }
    return 0;
}

int ddrr_trilinear_forward_f64(const double *volume, int dx, int dy, int dz, const double *source,
                               int src_n, const double *target, const double *img, int B, int N,
                               double voxel_shift, double eps, int n_points,
                               const double *alphamin, const double *alphamax, double *out,
                               void *) {
    const Dims D{dx, dy, dz};
    const double a0 = alphamin[0], a1 = alphamax[0];
    for (long r = 0; r < (long)B * N; ++r) {
        double s[3], t[3];
        ray64(source, src_n, target, r, N, s, t);
        out[r] = (img ? img[r] : 1.0) * ((a1 - a0) / (double)(n_points - 1)) *
                 ddrr64::trilinear_forward_ray(volume, D, s, t, voxel_shift, eps, n_points, a0, a1);
    }
    return 0;
}

int ddrr_trilinear_backward_f64(const double *volume, int dx, int dy, int dz,
                                const double *source, int src_n, const double *target,
                                const double *img, const double *grad_out, int B, int N,
                                double voxel_shift, double eps, int n_points,
                                const double *alphamin, const double *alphamax, double *g_source,
                                double *g_target, double *g_img, double *g_alpha, double *g_volume,
                                void *) {
    const Dims D{dx, dy, dz};
    const double a0 = alphamin[0], a1 = alphamax[0];
    for (long r = 0; r < (long)B * N; ++r) {
        double s[3], t[3], gs[3], gt[3], ga[2];
        ray64(source, src_n, target, r, N, s, t);
        const double g = grad_out[r], L = img ? img[r] : 1.0;
        const double sumT = ddrr64::trilinear_backward_ray(
            volume, D, s, t, voxel_shift, eps, n_points, a0, a1, g * L, gs, gt, ga,
            g_volume != nullptr, [&](long idx, double v) { g_volume[idx] += v; });
        for (int a = 0; a < 3; ++a) {
            if (g_source) g_source[r * 3 + a] = gs[a];
            if (g_target) g_target[r * 3 + a] = gt[a];
        }
        if (g_img) g_img[r] = g * ((a1 - a0) / (double)(n_points - 1)) * sumT;
        if (g_alpha) {
            g_alpha[r * 2] = ga[0];
            g_alpha[r * 2 + 1] = ga[1];
        }
    }
    return 0;
}

}  // extern "C"

// ---- the materialising general path (csrc/general_rays.hip): the same templated cores
namespace {

template <class T>
void gen_ray(const void *source, int src_n, const void *target, long r, int N, T s[3], T t[3]) {
    const T *sp = static_cast<const T *>(source), *tp = static_cast<const T *>(target);
    const long b = r / N, n = r - b * N;
    for (int a = 0; a < 3; ++a) {
        s[a] = sp[(b * src_n + (src_n == 1 ? 0 : n)) * 3 + a];
        t[a] = tp[r * 3 + a];
    }
}

template <class T, int LOOKUP>
void gen_segments(const void *volume, Dims D, const void *source, int src_n, const void *target,
                  const void *img, int B, int N, double shift, double eps, int ac, int raw,
                  void *terms) {
    const T *L = static_cast<const T *>(img);
    const long M1 = (long)D.x + D.y + D.z + 2;
    for (long r = 0; r < (long)B * N; ++r) {
        T s[3], t[3];
        gen_ray<T>(source, src_n, target, r, N, s, t);
        const long b = r / N, n = r - b * N;
        ddrr_gen::siddon_segments_ray<T, LOOKUP>(static_cast<const T *>(volume), D, s, t, (T)shift,
                                                 (T)eps, ac != 0, L ? L[r] : (T)1, raw != 0,
                                                 static_cast<T *>(terms) + b * M1 * N + n, N);
    }
}

template <class T, int LOOKUP>
void gen_segments_bwd(const void *volume, Dims D, const void *source, int src_n,
                      const void *target, const void *img, const void *g_terms, int B, int N,
                      double shift, double eps, int ac, int through, void *g_source,
                      void *g_target, void *g_img, void *g_volume) {
    const T *L = static_cast<const T *>(img);
    T *gsrc = static_cast<T *>(g_source), *gtgt = static_cast<T *>(g_target);
    T *gim = static_cast<T *>(g_img), *gvol = static_cast<T *>(g_volume);
    const long M1 = (long)D.x + D.y + D.z + 2;
    for (long r = 0; r < (long)B * N; ++r) {
        T s[3], t[3], gs[3], gt[3], gi;
        gen_ray<T>(source, src_n, target, r, N, s, t);
        const long b = r / N, n = r - b * N;
        ddrr_gen::siddon_segments_backward_ray<T, LOOKUP>(
            static_cast<const T *>(volume), D, s, t, (T)shift, (T)eps, ac != 0, L ? L[r] : (T)1,
            through != 0, static_cast<const T *>(g_terms) + b * M1 * N + n, N, gs, gt, gi,
            gvol != nullptr, [&](long idx, T v) { gvol[idx] += v; });
        for (int a = 0; a < 3; ++a) {
            if (gsrc) gsrc[r * 3 + a] = gs[a];
            if (gtgt) gtgt[r * 3 + a] = gt[a];
        }
        if (gim) gim[r] = gi;
    }
}

template <class T, bool NEAREST>
void gen_samples(const void *volume, Dims D, const void *source, int src_n, const void *target,
                 const void *img, int B, int N, double shift, double eps, int P, const void *amin,
                 const void *amax, int ac, int raw, void *samples) {
    const T *L = static_cast<const T *>(img);
    const T a0 = *static_cast<const T *>(amin), a1 = *static_cast<const T *>(amax);
    for (long r = 0; r < (long)B * N; ++r) {
        T s[3], t[3];
        gen_ray<T>(source, src_n, target, r, N, s, t);
        const long b = r / N, n = r - b * N;
        ddrr_gen::trilinear_samples_ray<T, NEAREST>(static_cast<const T *>(volume), D, s, t,
                                                    (T)shift, (T)eps, ac != 0, P, a0, a1,
                                                    L ? L[r] : (T)1, raw != 0,
                                                    static_cast<T *>(samples) + b * P * N + n, N);
    }
}

template <class T, bool NEAREST>
void gen_samples_bwd(const void *volume, Dims D, const void *source, int src_n, const void *target,
                     const void *img, const void *g_samples, int B, int N, double shift,
                     double eps, int P, const void *amin, const void *amax, int ac, void *g_source,
                     void *g_target, void *g_img, void *g_alpha, void *g_volume) {
    const T *L = static_cast<const T *>(img);
    const T a0 = *static_cast<const T *>(amin), a1 = *static_cast<const T *>(amax);
    T *gsrc = static_cast<T *>(g_source), *gtgt = static_cast<T *>(g_target);
    T *gim = static_cast<T *>(g_img), *gal = static_cast<T *>(g_alpha);
    T *gvol = static_cast<T *>(g_volume);
    for (long r = 0; r < (long)B * N; ++r) {
        T s[3], t[3], gs[3], gt[3], ga[2], gi;
        gen_ray<T>(source, src_n, target, r, N, s, t);
        const long b = r / N, n = r - b * N;
        ddrr_gen::trilinear_samples_backward_ray<T, NEAREST>(
            static_cast<const T *>(volume), D, s, t, (T)shift, (T)eps, ac != 0, P, a0, a1,
            L ? L[r] : (T)1, static_cast<const T *>(g_samples) + b * P * N + n, N, gs, gt, ga, gi,
            gvol != nullptr, [&](long idx, T v) { gvol[idx] += v; });
        for (int a = 0; a < 3; ++a) {
            if (gsrc) gsrc[r * 3 + a] = gs[a];
            if (gtgt) gtgt[r * 3 + a] = gt[a];
        }
        if (gim) gim[r] = gi;
        if (gal) {
            gal[r * 2] = ga[0];
            gal[r * 2 + 1] = ga[1];
        }
    }
}

}  // namespace

#define GEN_BY_LOOKUP(FN, T, ...)                                              \
    do {                                                                       \
        if (lookup == DDRR_LOOKUP_STEP) FN<T, LOOKUP_STEP>(__VA_ARGS__);       \
        else if (lookup == DDRR_LOOKUP_MID_NEAREST) FN<T, LOOKUP_MID_NEAREST>(__VA_ARGS__); \
        else FN<T, LOOKUP_MID_TRILINEAR>(__VA_ARGS__);                         \
    } while (0)

extern "C" {

int ddrr_siddon_segments_general(const void *volume, int f64, int dx, int dy, int dz,
                                 const void *source, int src_n, const void *target,
                                 const void *img, int B, int N, double voxel_shift, double eps,
                                 int lookup, int align_corners, int raw, void *terms, void *) {
    const Dims D{dx, dy, dz};
    if (lookup == DDRR_LOOKUP_STEP && align_corners) return -1;
    if (f64)
        GEN_BY_LOOKUP(gen_segments, double, volume, D, source, src_n, target, img, B, N,
                      voxel_shift, eps, align_corners, raw, terms);
    else
        GEN_BY_LOOKUP(gen_segments, float, volume, D, source, src_n, target, img, B, N,
                      voxel_shift, eps, align_corners, raw, terms);
    return 0;
}

int ddrr_siddon_segments_general_backward(const void *volume, int f64, int dx, int dy, int dz,
                                          const void *source, int src_n, const void *target,
                                          const void *img, const void *grad_terms, int B, int N,
                                          double voxel_shift, double eps, int lookup,
                                          int align_corners, int through_lookup, void *g_source,
                                          void *g_target, void *g_img, void *g_volume, void *) {
    const Dims D{dx, dy, dz};
    if (lookup == DDRR_LOOKUP_STEP && align_corners) return -1;
    if (!through_lookup && (g_volume || g_img)) return -1;
    if (f64)
        GEN_BY_LOOKUP(gen_segments_bwd, double, volume, D, source, src_n, target, img, grad_terms,
                      B, N, voxel_shift, eps, align_corners, through_lookup, g_source, g_target,
                      g_img, g_volume);
    else
        GEN_BY_LOOKUP(gen_segments_bwd, float, volume, D, source, src_n, target, img, grad_terms,
                      B, N, voxel_shift, eps, align_corners, through_lookup, g_source, g_target,
                      g_img, g_volume);
    return 0;
}

int ddrr_trilinear_samples_general(const void *volume, int f64, int dx, int dy, int dz,
                                   const void *source, int src_n, const void *target,
                                   const void *img, int B, int N, double voxel_shift, double eps,
                                   int n_points, const void *alphamin, const void *alphamax,
                                   int nearest, int align_corners, int raw, void *samples,
                                   void *) {
    const Dims D{dx, dy, dz};
#define GEN_SAMPLES(T, NN)                                                                       \
    gen_samples<T, NN>(volume, D, source, src_n, target, img, B, N, voxel_shift, eps, n_points, \
                       alphamin, alphamax, align_corners, raw, samples)
    if (f64) {
        if (nearest) GEN_SAMPLES(double, true); else GEN_SAMPLES(double, false);
    } else {
        if (nearest) GEN_SAMPLES(float, true); else GEN_SAMPLES(float, false);
    }
#undef GEN_SAMPLES
    return 0;
}

int ddrr_trilinear_samples_general_backward(const void *volume, int f64, int dx, int dy, int dz,
                                            const void *source, int src_n, const void *target,
                                            const void *img, const void *grad_samples, int B,
                                            int N, double voxel_shift, double eps, int n_points,
                                            const void *alphamin, const void *alphamax,
                                            int nearest, int align_corners, void *g_source,
                                            void *g_target, void *g_img, void *g_alpha,
                                            void *g_volume, void *) {
    const Dims D{dx, dy, dz};
#define GEN_SAMPLES_BWD(T, NN)                                                                   \
    gen_samples_bwd<T, NN>(volume, D, source, src_n, target, img, grad_samples, B, N,           \
                           voxel_shift, eps, n_points, alphamin, alphamax, align_corners,       \
                           g_source, g_target, g_img, g_alpha, g_volume)
    if (f64) {
        if (nearest) GEN_SAMPLES_BWD(double, true); else GEN_SAMPLES_BWD(double, false);
    } else {
        if (nearest) GEN_SAMPLES_BWD(float, true); else GEN_SAMPLES_BWD(float, false);
    }
#undef GEN_SAMPLES_BWD
    return 0;
}

}  // extern "C"

// Test hook (tests/test_host_api.py): the arithmetic of the brick kernels' staging by quads --
// brick_core.h quad_clamped_at / quad_shift, which brick_shared.h quad_load / quad_fix are made of
// -- on a host volume: the four words a lane ends up with for the quad whose first voxel is
// (x, y, z), before the caller's masks.
extern "C" int ddrr_emu_quad_stage(const float *volume, int dx, int dy, int dz, int x, int y, int z,
                                   float *out4) {
    const Dims D{dx, dy, dz};
    if ((long)dx * dy * dz < 4 || !quads_serve(D)) return -1;  // (the host sends these elsewhere)
    const int xc = x < dx ? x : dx - 1, yc = y < dy ? y : dy - 1;
    const long a = quad_clamped_at(D, ((long)xc * dy + yc) * dz + z);
    float w[8] = {volume[a], volume[a + 1], volume[a + 2], volume[a + 3], 0.f, 0.f, 0.f, 0.f};
    const int sh = quad_shift(D, x, y, z), shift = sh > 0 ? sh : 0;  // (quad_fix: only if > 0)
    for (int i = 0; i < 4; ++i) out4[i] = shift + i < 8 ? w[shift + i] : 0.f;
    return 0;
}
