// brick_shared.h -- what the translation units of the volume-stationary kernels share
// (bricks.hip: every mode on 32^3 fp32 bricks; bricks_fwd.hip: the configurable Siddon forward /
// forward + record kernel): launch arguments, the per-phase profile of tools builds, wave
// helpers, and the delivery of the float backward record.
#pragma once

#include "runtime.h"
#include "brick_core.h"
#include "brick_walk.h"
#include "record_layout.h"

namespace ddrr_brick {

using namespace ddrr;

constexpr int kQueueCap = 128;  // entries per wave and length class (63 waiting + 64 arriving)
constexpr int kBuckets = 3;     // length classes

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
    float rec_q;         // > 0: the record is the packed fixed-point form (record_pack.h), scale q
    int pix_bits;        // queue entry = (pose << pix_bits) | pixel
    float t1, t2;        // length-class thresholds on the estimated crossing count
    int dbg;             // experiment switches (tools builds with -DDDRR_EXPERIMENTS; else 0)
    int *work;           // global brick counter of this launch (zero at launch)
    const float *grad_out;  // *_VOLGRAD: dLoss/dout (B, N); BRICK_[TRI_]CHANNELS_AUX, BRICK_CHANNELS_VOLGRAD: (B, n_channels, N)
    float *g_volume;        // *_VOLGRAD: dLoss/dvolume
    int n_points;           // BRICK_TRI_*: samples per ray
    const float *amin, *amax;  // BRICK_TRI_*: device scalars (renderers.py:220-223)
    unsigned long long *prof;  // DDRR_BRICK_PROFILE builds: per-phase wave-cycle totals
    const unsigned char *labels;  // BRICK_CHANNELS: label of every voxel
    int n_channels;               // BRICK_CHANNELS: out is (B, n_channels, N)
    const float *ranges;          // 16-bit bricks: (vmin, vmax) per brick (bricks_fwd.hip)
    const int *fallback;          // ... 1: the brick is rendered from its fp32 values (q16_usable)
    int *ws_header;               // ... the workspace's header words
    int ranges_valid;             // ... already computed for this volume
    const unsigned char *packed;  // 16-bit bricks: their LDS images, brick after brick (or null)
    const unsigned *pix_mask;     // the forward kernels: bit n of word n / 32 = detector pixel n is rendered (NULL: every pixel)
    const unsigned *fingerprint;  // 16-bit bricks: bit patterns of kFingerprintWords voxels of the volume the workspace was built from
    int vec;                      // bricks_fwd.hip brick_range_kernel: 16-byte loads serve the volume
    const int *order;             // bricks_fwd.hip: k-th brick handed out (NULL: k itself)
    int *order_ws;                // ... this launch's workspace for it: order_cap weights, order_cap ints
    int order_cap;
    float *clear;                 // host side of bricks_fwd.hip: floats to zero in front of the launch (or NULL) ...
    long clear_n;                 // ... so many, by the launch that zeroes the brick counter (brick_clear_kernel);
                                  // -1: the caller has cleared them and the counter (DDRR_BRICKS_CLEARED)
    unsigned *brick_times;        // profiling builds: duration of every brick (10 ns ticks), or NULL
    int split_t, split_s;         // ... the last split_t bricks are handed out in split_s pose parts
};

// Phase timing of the brick kernel (tools/ builds with -DDDRR_BRICK_PROFILE only): s_memtime
// deltas per wave, added up per phase.  Compiled out of the product library.
#if defined(DDRR_BRICK_PROFILE)
struct BrickProf {
    unsigned long long t[16];
    unsigned long long last, born;  // born: s_memrealtime (100 MHz, one clock for all XCDs)
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = 0;
        last = __builtin_amdgcn_s_memtime();
        born = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void mark(int i) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        t[i] += now - last;
        last = now;
    }
    __device__ __forceinline__ void count(int i, unsigned long long n) { t[i] += n; }
};
#if defined(DDRR_TRACE_ONLY)
// (-DDDRR_TRACE_ONLY: the per-brick stage stamps of DDRR_TRACE alone -- the phase marks cost a
// scalar-memory round trip each and stretch a one-pose brick by a quarter)
#define DDRR_PROF(i)
#define DDRR_PROF_COUNT(i, n)
#define DDRR_PROF_WAIT_VMEM()
#else
#define DDRR_PROF(i) prof.mark(i)
#define DDRR_PROF_COUNT(i, n) prof.count(i, n)
#define DDRR_PROF_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#else
struct BrickProf {};
#define DDRR_PROF(i)
#define DDRR_PROF_COUNT(i, n)
#define DDRR_PROF_WAIT_VMEM()
#endif
enum {
    PROF_STAGE = 0,   // brick id, row table, staging, barrier after it
    PROF_PULL = 1,    // unit counter, cursor, row read
    PROF_PHASE_A = 2, // candidate test, classes, queue push
    PROF_POP = 3,     // batch selection, queue read
    PROF_LOADS = 4,   // ray loads (issue + wait)
    PROF_SETUP = 5,   // exact clip, entry cell
    PROF_WALK = 6,
    PROF_DELIVER = 7, // atomics
    PROF_BARRIER = 8, // waiting for the other waves at the end of a brick
    PROF_N_BATCH = 9, PROF_N_STEPS = 10, PROF_N_UNITS = 11, PROF_N_HITS = 12,
    PROF_CLAIM = 13,   // (part of staging) brick id known
    PROF_ROWS = 14,    // (part of staging) row table written
    PROF_STORE = 15,   // (part of staging) brick stored to LDS; PROF_STAGE then is the barrier + prefix
};

// what a brick launch computes
constexpr int BRICK_FWD = 0;      // out
constexpr int BRICK_FWD_AUX = 1;  // out + planar backward record
constexpr int BRICK_VOLGRAD = 2;  // g_volume (the brick in LDS is the accumulator)
constexpr int BRICK_TRI_FWD = 3;      // trilinear marcher: out
constexpr int BRICK_TRI_VOLGRAD = 4;  // trilinear marcher: g_volume
constexpr int BRICK_TRI_FWD_AUX = 5;  // trilinear marcher: planar backward record (out follows from it)
constexpr int BRICK_CHANNELS = 6;     // out (B, C, N): one line integral per label (mask_to_channels)
constexpr int BRICK_TRI_CHANNELS = 7; // trilinear marcher, out (B, C, N): samples by the label of their nearest voxel
constexpr int BRICK_CHANNELS_AUX = 8; // backward of BRICK_CHANNELS w.r.t. the rays: the record of the volume weighted by grad_out[b, label, n]
constexpr int BRICK_TRI_CHANNELS_AUX = 9;  // backward of BRICK_TRI_CHANNELS w.r.t. the rays: the marcher's record weighted by grad_out[b, label, n]
constexpr int BRICK_CHANNELS_VOLGRAD = 10;  // backward of BRICK_CHANNELS w.r.t. the volume: the LDS accumulator's words carry the voxel's label in their low byte
constexpr int BRICK_CHANNELS_WORDS = 12;  // BRICK_CHANNELS from a volume of ready-packed words (value | label): staged like a plain brick
constexpr int BRICK_TRI_CHANNELS_VOLGRAD = 11;  // the same for the marcher (owner bricks; labels outside the owned box from the label map)


#if defined(__HIPCC__)
constexpr int kRor8 = 0x128;  // DPP row_ror:8: lane ^ 8 within each row of 16

// The float backward record of one batch of hits, delivered to the blocked layout of
// record_layout.h.  Called by ALL lanes of the wave (ok: the lane holds a hit): lane l swaps
// plane 1 (3) of its hit against plane 0 (2) of lane l ^ 8's, so that the lower half of a
// 16-lane row carries planes 0 | 1 of its eight hits and the upper half those of its own eight --
// when the eight are one run of adjacent pixels (they are: length classes are formed per run of 8)
// each half row is one contiguous 64-byte line: 2 + 2 + 1 atomic instructions as before, but
// whole lines instead of half lines.  v = {I, S0x, S0z, S1x, S1z}.
__device__ __forceinline__ void deliver_record_blocked(float *__restrict__ aux, bool ok,
                                                       unsigned r, const float v[5]) {
    const unsigned lo = rec_off01(r) | (ok ? 0u : 0x80000000u);  // (sign bit: nothing to add)
    const int o0 = (int)lo, o1 = (int)(lo + 8u);
    // X: lanes 0-7 of a row own plane 0 | lanes 8-15 their partner's plane 1
    // Y: lanes 0-7 their partner's plane 0 | lanes 8-15 own plane 1
    const int xo = __builtin_amdgcn_update_dpp(o0, o1, kRor8, 0xf, 0xC, false);
    const int yo = __builtin_amdgcn_update_dpp(o1, o0, kRor8, 0xf, 0x3, false);
    auto swap_hi = [](float own, float other) {  // lanes 8-15 <- partner's `other`
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
            __builtin_bit_cast(int, own), __builtin_bit_cast(int, other), kRor8, 0xf, 0xC, false));
    };
    auto swap_lo = [](float own, float other) {  // lanes 0-7 <- partner's `other`
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
            __builtin_bit_cast(int, own), __builtin_bit_cast(int, other), kRor8, 0xf, 0x3, false));
    };
    const float x01 = swap_hi(v[0], v[1]), y01 = swap_lo(v[1], v[0]);
    const float x23 = swap_hi(v[2], v[3]), y23 = swap_lo(v[3], v[2]);
    if (xo >= 0) {
        unsafeAtomicAdd(aux + xo, x01);
        unsafeAtomicAdd(aux + xo + 16, x23);
    }
    if (yo >= 0) {
        unsafeAtomicAdd(aux + yo, y01);
        unsafeAtomicAdd(aux + yo + 16, y23);
    }
    if (ok) unsafeAtomicAdd(aux + rec_off4(r), v[4]);
}

// Four voxels along z from element `at` of the volume (LABELS: and their labels): one 16-byte
// load from a dword-aligned address -- global memory takes any dword alignment, and the label
// dword any alignment -- so that any D.z >= 4 is staged like a multiple of 4 (the reference's
// example CT has 133 slices).  The address is clamped to the volume's last four voxels; the one
// quad this happens to (it reaches beyond the volume's last row: none does when D.z is a multiple
// of 4) has its words shifted into place by quad_fix -- where the words are USED, so that a
// round's loads stay in flight together.  What a load picks up behind the quad's row is for the
// caller to mask.  (Volumes of fewer than four slices: brick_core.h quads_serve -- the host sends
// them to the general kernel's scalar staging.)
typedef unsigned int quad_u32x4 __attribute__((ext_vector_type(4)));  // (a native vector: plain loads / stores)
typedef unsigned int quad_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <bool LABELS>
__device__ __forceinline__ void quad_load(const float *__restrict__ vol,
                                          const unsigned char *__restrict__ label_map, const Dims &D,
                                          long at, quad_u32x4 &v, unsigned &labels) {
    const long a = quad_clamped_at(D, at);
    v = *reinterpret_cast<const quad_u32x4_a4 *>(vol + a);
    if (LABELS) __builtin_memcpy(&labels, label_map + a, 4);
}
// (x, y, z: the quad's first voxel; brick_core.h quad_shift says which quads were clamped.  One
// that starts behind a row's end is masked by the caller whatever it holds.)
__device__ __forceinline__ void quad_fix(const Dims &D, int x, int y, int z, quad_u32x4 &v,
                                         unsigned &labels) {
    const int shift = quad_shift(D, x, y, z);
    if (shift > 0) {
        if (shift >= 2) v = quad_u32x4{v.z, v.w, 0u, 0u};
        if (shift & 1) v = quad_u32x4{v.y, v.z, v.w, 0u};
        labels = shift < 4 ? labels >> (8 * shift) : 0u;
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

#endif  // __HIPCC__

// This launch's device-side state, carved out of the caller's launch workspace (bricks.hip): the
// CU count and the brick counter {brick id, wmax bits, n_sum, -}, zeroed on `st`;
// order_ws / order_cap: the room for the hand-out order of the bricks (bricks_fwd.hip).
long brick_launch_workspace_bytes(int dx, int dy, int dz);
int brick_launch_resources(hipStream_t st, void *launch_ws, int dx, int dy, int dz, int &n_cu,
                           int *&work, int **order_ws = nullptr, int *order_cap = nullptr,
                           bool zero_work = true);

// The 32^3 fp32 launch path of bricks.hip (every mode); bricks_fwd.hip falls back to it.
bool order_bricks(BrickArgs &q, int BX, int BY, int BZ, int nby, int nbz, int n_bricks, int slots,
                  hipStream_t st, bool zero_counter = false, int min_poses = 8);
int launch_bricks(int mode, const float *volume, int dx, int dy, int dz, const float *source,
                  const float *target, const float *img, const float *grad_out, int B, int det_h,
                  int det_w, float voxel_shift, float eps, float *out, float *aux,
                  float *g_volume, hipStream_t st, void *launch_ws, const char *who, int n_points = 0,
                  const float *amin = nullptr, const float *amax = nullptr, float rec_q = 0.f,
                  const unsigned char *labels = nullptr, int n_channels = 0,
                  const unsigned *pix_mask = nullptr);

// The configurable Siddon forward / forward + record kernel (bricks_fwd.hip).  variant:
// DDRR_BRICKS_F32 / DDRR_BRICKS_Q16 (fp32 bricks at fewer than 8 poses, and volumes of fewer than four
// voxels, take launch_bricks).
// brick_ranges: the DDRR_BRICKS_Q16 workspace (header, (min, max) and fallback flag per brick),
// ranges_valid: it already holds this volume's.
// packed: the workspace also holds the bricks' LDS images (DDRR_BRICKS_Q16_PACKED) behind the ranges.
// clear, clear_n: floats the launch has to find zeroed (the image or the record its atomics add to):
// cleared by the launch that clears the brick counter -- one launch instead of two memsets, each
// of which is a launch of its own (5 us apiece in front of a 140 us one-pose render).
long brick_workspace_bytes(int dx, int dy, int dz, int brick_storage);
int launch_fwd_bricks(int variant, int packed, float *brick_ranges, int ranges_valid, const float *volume,
                      int dx, int dy, int dz, const float *source, const float *target,
                      const float *img, int B, int det_h, int det_w, float voxel_shift, float eps,
                      float *out, float *aux, float rec_q, hipStream_t st, void *launch_ws,
                      const char *who, float *clear = nullptr, long clear_n = 0,
                      const unsigned *pix_mask = nullptr);

// experiment switches (tools builds: mutable; product: constants)
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
extern float g_brick_t1, g_brick_t2;
extern int g_brick_dbg;
extern int g_brick_variant;
extern float g_brick_sq_width;
extern const int *g_brick_order;
extern unsigned *g_brick_times;  // profiling builds: per-brick duration, 10 ns ticks
extern int g_brick_split_t, g_brick_split_s;
#else
constexpr float g_brick_t1 = 18.f, g_brick_t2 = 40.f;
constexpr int g_brick_dbg = 0;
#endif
#if defined(DDRR_BRICK_PROFILE)
extern unsigned long long *g_brick_prof;  // 16 device counters, see BrickProf
#endif

}  // namespace ddrr_brick
