// bricks.hip -- the volume-stationary kernels (brick_core.h, brick_walk.h, tri_brick.h): one
// persistent 1024-thread workgroup per CU stages 32^3 bricks in LDS and traces every ray of
// every pose through them; Siddon forward (+ backward record), Siddon and trilinear volume
// gradients, trilinear forward (+ record); and the elementwise kernels that consume the records.
#include "runtime.h"

#include <mutex>
#include "siddon_core.h"
#include "brick_core.h"
#include "brick_walk.h"
#include "brick_step.h"
#include "record_pack.h"
#include "record_layout.h"
#include "tri_brick.h"
#include "brick_shared.h"
#include "trilinear_core.h"

using namespace ddrr;
using namespace ddrr_rt;
using namespace ddrr_brick;

namespace {

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

inline size_t brick_lds_bytes(const BrickLayout &lay) {
    return (size_t)brick_floats(lay) * 4 + (size_t)kBrickWaves * kBuckets * kQueueCap * 4 +
           (size_t)(kPoseChunk * kRowWords + 4) * 4;
}

#if defined(__HIPCC__)
// Scatter into the LDS accumulator by absolute LDS byte address.  LDS float atomics run at
// ~0.7 lane per clock on gfx950 (measured: ds_add_f32 occupies the LDS pipe ~90 cycles per
// wave instruction, conflicts or not), integer ones 4x faster: the accumulator is int32
// fixed point, value = count / q, with q chosen per launch from a bound on the largest sum
// a voxel can receive (volgrad_prepare_kernel); q == 0 selects the float path (the bound
// does not exist, e.g. the source lies inside the volume).  Integer sums are associative:
// the fixed-point gradient is bit-reproducible.
// The choice between the two is made where the accumulator is built, not per add (a test per add is
// a branch per corner of the marcher's samples, per step of a walk), and the scale is folded into
// the ray's weight (brick_step.h acc_scale / acc_add): one conversion per add, round half up.
template <bool FIXED>
struct LdsAbsAddT {
    float q;
    __device__ __forceinline__ float scale(float w) const { return FIXED ? w * q : w; }
    __device__ __forceinline__ void add_scaled(unsigned addr, float v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if (FIXED) {
            int c;
            asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(c) : "v"(v));  // floor(v + 1/2)
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
            if (q < 0.f) {  // (timing experiment: a plain store instead of the atomic; the result is garbage)
                *(__attribute__((address_space(3))) int *)(unsigned long long)addr = c;
                return;
            }
#endif
            __hip_atomic_fetch_add((int *)(__attribute__((address_space(3))) int *)(unsigned long long)addr, c,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add((float *)(__attribute__((address_space(3))) float *)(unsigned long long)addr,
                                   v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#else
        (void)addr;
        (void)v;
#endif
    }
};

// The same into the upper 24 bits of the word (BRICK_CHANNELS_VOLGRAD): the low byte holds the
// voxel's label, which sums of multiples of 256 never touch.  q: counts per unit, for 23 bits.
// (Fewer label bits for fewer channels -- variable shifts -- cost 20 % and bought no accuracy: a
// voxel's gradient is one or two segment lengths, and those carry the 1e-4 of a difference of two
// fp32 alphas whatever the accumulator.)
struct LdsAbsAddHigh {
    float q;
    __device__ __forceinline__ float scale(float w) const { return w * q; }
    __device__ __forceinline__ void add_scaled(unsigned addr, float v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        int c;
        asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(c) : "v"(v));  // floor(v + 1/2)
        __hip_atomic_fetch_add((int *)(__attribute__((address_space(3))) int *)(unsigned long long)addr, c << 8,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        (void)addr;
        (void)v;
#endif
    }
    __device__ __forceinline__ void operator()(unsigned addr, float v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_add((int *)(__attribute__((address_space(3))) int *)(unsigned long long)addr,
                               __float2int_rn(v * q) << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        (void)addr;
        (void)v;
#endif
    }
};
// ... and the label of the voxel at an LDS address: that low byte,
struct LdsLabelLow {
    __device__ __forceinline__ unsigned operator()(unsigned addr) const {
        return float_bits(LdsAbsFetch{}(addr)) & 0xffu;
    }
};
// or -- no bound on a voxel's sum, float accumulators (see LdsAbsAddT) -- the label map itself,
// at the voxel the address belongs to (byte strides of the LDS copy: 4 sx, 4 sy, 4).
struct GlobalLabelOf {
    const unsigned char *labels;
    Dims D;
    int lo[3];
    unsigned base, sx4, sy4;
    __device__ __forceinline__ unsigned operator()(unsigned addr) const {
        const unsigned off = addr - base, lx = off / sx4, r = off - lx * sx4, ly = r / sy4;
        const unsigned lz = (r - ly * sy4) >> 2;
        return labels[((long)(lo[0] + (int)lx) * D.y + (lo[1] + (int)ly)) * D.z + (lo[2] + (int)lz)];
    }
};

// The marcher's sample label (BRICK_TRI_CHANNELS_VOLGRAD): from the label map, 0 outside the
// volume.  (Labels in the accumulator words' low byte, as for Siddon, were built and measured:
// the marcher adds ~100 corner weights per voxel under a bound of thousands, and 24 bits leave
// 1e-4 .. 4e-4 of the largest gradient as rounding -- the 31-bit accumulator 4e-6.  The byte loads
// hit L1 / L2: neighbouring samples share their nearest voxel.)
struct TriLabelOf {
    const unsigned char *labels;
    Dims D;
    __device__ __forceinline__ unsigned operator()(float rx, float ry, float rz, bool, unsigned) const {
        const bool in = rx >= 0.f && ry >= 0.f && rz >= 0.f && rx < (float)D.x && ry < (float)D.y &&
                        rz < (float)D.z;
        return in ? labels[((long)(int)rx * D.y + (int)ry) * D.z + (int)rz] : 0u;
    }
};

// A finished label run of a ray's walk through a brick goes to the ray's output column
// (B, C, N): one fire-and-forget atomic per run.  32-bit offsets: the host checks B C N < 2^30.
// SCALED: L multiplies every run (else the walk carries it); CHECKED: labels >= C are dropped here
// (else the staging has already removed them: pack_voxel_label_below).
template <bool SCALED = true, bool CHECKED = true>
struct BrickColumnFlush {
    float *out;
    unsigned colb;  // byte offset of (b, 0, pixel): 4 (b C N + pixel)
    unsigned N4, C;  // byte stride between channels
    float L;         // SCALED: factor of every run (else the walk carries it, step_walk_channels)
    __device__ __forceinline__ void operator()(unsigned lab, float run) const {
        // (a 32-bit byte offset from the wave-uniform base: ONE 24-bit multiply-add -- left to the
        // compiler this becomes a 64-bit v_mad_u64_u32 with a scalar operand -- and the atomic's
        // scalar-base addressing)
        if (!CHECKED || lab < C) {
#if defined(__HIP_DEVICE_COMPILE__)
            unsigned off, n4 = N4;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(lab), "v"(n4), "v"(colb));
#else
            const unsigned off = colb + lab * N4;
#endif
            unsafeAtomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(out) + off),
                            SCALED ? L * run : run);
        }
    }
#if defined(DDRR_CHANNELS_MASKED_FLUSH)
    // EXPERIMENT, measured and not adopted (tools builds with -DDDRR_CHANNELS_MASKED_FLUSH;
    // profiles/r06/channel_masked_flush_experiment.txt).  The flush of a label change inside the step,
    // branch-free ("lanes whose label changed flush under an exec mask in the step's
    // own iteration"): compare, narrow the exec mask to the lanes that changed, one multiply-add for
    // the address, the fire-and-forget atomic, restore -- six instructions EVERY step instead of a
    // divergent block behind a branch whenever any lane of the wave changes label.  1.5 % slower on
    // the reference's label map (0.281 against 0.276 ms at 8 poses), 1.6 % on synthetic blocks: the
    // branch is not what the flush costs.  -> the run to go on with (0 where it was handed over).
    // Unscaled, unchecked flushes only (the Siddon channel render).
    static constexpr bool kMaskedFlush = !SCALED && !CHECKED;
    __device__ __forceinline__ float masked(unsigned lab, unsigned cur, float run) const {
        unsigned long long saved;
        unsigned off;
        float next;
        asm volatile(
            "v_cmp_ne_u32_e32 vcc, %[lab], %[cur]\n\t"
            "s_and_saveexec_b64 %[sv], vcc\n\t"
            "v_mad_u32_u24 %[off], %[cur], %[n4], %[colb]\n\t"
            "global_atomic_add_f32 %[off], %[run], %[base]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "v_cndmask_b32_e64 %[next], %[run], 0, vcc"
            : [sv] "=&s"(saved), [off] "=&v"(off), [next] "=&v"(next)
            : [lab] "v"(lab), [cur] "v"(cur), [n4] "v"(N4), [colb] "v"(colb), [run] "v"(run), [base] "s"(out)
            : "vcc", "memory");
        return next;
    }
#endif
};

// The incoming gradient of the ray's own output column, by label (BRICK_CHANNELS_AUX): a gather
// at a 32-bit byte offset from the wave-uniform base.
// CHECKED: labels without a channel (>= C) weigh nothing (the Siddon staging has already turned
// them into value 0 | label 0).
template <bool CHECKED = false>
struct BrickColumnWeight {
    const float *g;
    unsigned colb, N4, C;
    __device__ __forceinline__ float operator()(unsigned lab) const {
        if (CHECKED && lab >= C) return 0.f;
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(g) + (colb + __umul24(lab, N4)));
    }
};
#endif

// Phase B for one queue entry: load the real ray, clip, walk; add to the image (forward)
// or scatter into the LDS accumulator (volume gradient).  Called by every lane of the wave;
// `active`: the lane holds a queue entry.
// Offsets are 32-bit: the host checks 12 * B * N < 2^32.
template <int MODE>
__device__ __forceinline__ void brick_item(const BrickArgs &p, const float *brick,
                                           const BrickGeom &G, const StepGeom &SG, bool active,
                                           unsigned b, unsigned pix, float fixq,
                                           float *__restrict__ out, float *__restrict__ aux,
                                           BrickProf &prof) {
    constexpr bool AUX = MODE == BRICK_FWD_AUX || MODE == BRICK_CHANNELS_AUX;
    const unsigned r = b * (unsigned)(p.det_h * p.det_w) + pix;
    if (AUX) {
        // the walk under the lanes that hold an entry, the record's delivery by all of them
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        bool ok = false;
        if (active) {
            const float *sp = p.source + b * 3u, *tp = p.target + r * 3u;
            const float s[3] = {sp[0], sp[1], sp[2]}, t[3] = {tp[0], tp[1], tp[2]};
            DDRR_PROF_WAIT_VMEM();
            DDRR_PROF(PROF_LOADS);
            const StepEntry E = step_enter(SG, s, t, p.shift, p.eps, LdsAbsFetch::base_of(brick));
            DDRR_PROF(PROF_SETUP);
            int steps = 0;
            if (MODE == BRICK_CHANNELS_AUX) {
                const unsigned N = (unsigned)(p.det_h * p.det_w), C = (unsigned)p.n_channels;
                if (E.hit)
                    steps = step_walk_weighted(LdsAbsFetch{}, SG, E,
                                               BrickColumnWeight<false>{p.grad_out, (b * C * N + pix) * 4u, N * 4u, C},
                                               v[0], v + 1);
            } else if (E.hit) {
                steps = step_walk<true>(LdsAbsFetch{}, SG, E, v[0], v + 1);
            }
            DDRR_PROF(PROF_WALK);
            DDRR_PROF_COUNT(PROF_N_STEPS, (unsigned long long)__builtin_amdgcn_readfirstlane(steps));
            (void)steps;
            ok = E.hit;  // (phase A's margin lets a few non-crossing rays through)
        }
        // with the record, out = L * (plane I) is formed afterwards (siddon_out_from_record_kernel)
        if (p.rec_q > 0.f) {
            if (ok) {
                // packed record: (S1x : S0x) and (S1z : S0z) as two 64-bit integer atomics
                const float qa = p.rec_q / aux[5u * p.aux_plane + r];
                unsigned long long *X = reinterpret_cast<unsigned long long *>(aux);
                atomicAdd(X + r, (unsigned long long)record_pack(v[1], v[3], p.rec_q, qa));
                atomicAdd(X + p.aux_plane + r,
                          (unsigned long long)record_pack(v[2], v[4], p.rec_q, qa));
                unsafeAtomicAdd(aux + 4u * p.aux_plane + r, v[0]);
            }
        } else if (p.dbg & 64) {
            // (experiment builds: the record's delivery with one atomic instead of five -- what the
            // kernel would cost if the atomics were free; the record is wrong)
            if (ok) unsafeAtomicAdd(aux + r, v[0] + v[1] + v[2] + v[3] + v[4]);
        } else if (p.dbg & 128) {
            // (experiment builds: the blocked layout without the lane swap: half lines)
            if (ok) {
                const unsigned o = rec_off01(r);
                unsafeAtomicAdd(aux + o, v[0]);
                unsafeAtomicAdd(aux + o + 8, v[1]);
                unsafeAtomicAdd(aux + o + 16, v[2]);
                unsafeAtomicAdd(aux + o + 24, v[3]);
                unsafeAtomicAdd(aux + rec_off4(r), v[4]);
            }
        } else {
            deliver_record_blocked(aux, ok, r, v);
        }
        DDRR_PROF(PROF_DELIVER);
        return;
    }
    if (!active) return;
    const float *sp = p.source + b * 3u, *tp = p.target + r * 3u;
    const float s[3] = {sp[0], sp[1], sp[2]}, t[3] = {tp[0], tp[1], tp[2]};
    const float L = p.img ? p.img[r] : 1.f;
    const float base = (float)LdsAbsFetch::base_of(brick);
    if (MODE == BRICK_VOLGRAD) {
        const float w = p.grad_out[r] * L;
        if (w != 0.f) {
            if (fixq != 0.f)
                step_scatter(LdsAbsAddT<true>{fixq}, LdsAbsFetch::base_of(brick), SG, s, t, p.shift, p.eps, w);
            else
                step_scatter(LdsAbsAddT<false>{0.f}, LdsAbsFetch::base_of(brick), SG, s, t, p.shift, p.eps, w);
        }
        return;
    }
    if (MODE == BRICK_CHANNELS_VOLGRAD) {
        const unsigned N = (unsigned)(p.det_h * p.det_w), C = (unsigned)p.n_channels;
        const BrickColumnWeight<true> weight{p.grad_out, (b * C * N + pix) * 4u, N * 4u, C};
        const unsigned lbase = LdsAbsFetch::base_of(brick);
        if (fixq != 0.f) {
            step_scatter_weighted(LdsAbsAddHigh{fixq}, LdsLabelLow{}, weight, lbase, SG, s, t, p.shift,
                                  p.eps, L);
        } else {
            const GlobalLabelOf label{p.labels, p.D, {(int)G.lof[0], (int)G.lof[1], (int)G.lof[2]}, lbase,
                                      (unsigned)p.lay.sx * 4u, (unsigned)p.lay.sy * 4u};
            step_scatter_weighted(LdsAbsAddT<false>{0.f}, label, weight, lbase, SG, s, t, p.shift, p.eps, L);
        }
        return;
    }
    if (MODE == BRICK_TRI_CHANNELS_VOLGRAD) {
        const unsigned N = (unsigned)(p.det_h * p.det_w), C = (unsigned)p.n_channels;
        const BrickColumnWeight<true> weight{p.grad_out, (b * C * N + pix) * 4u, N * 4u, C};
        const float a0 = p.amin[0], a1 = p.amax[0];
        const float k = L * ((a1 - a0) / (float)(p.n_points - 1));
        if (fixq != 0.f)
            tri_owner_scatter_weighted(LdsAbsAddT<true>{fixq}, TriLabelOf{p.labels, p.D}, weight, base, G.lof, G.hif,
                                       G.stridef, p.D, s, t, p.shift, p.eps, p.n_points, a0, a1, k);
        else
            tri_owner_scatter_weighted(LdsAbsAddT<false>{0.f}, TriLabelOf{p.labels, p.D}, weight, base, G.lof, G.hif,
                                       G.stridef, p.D, s, t, p.shift, p.eps, p.n_points, a0, a1, k);
        return;
    }
    if (MODE == BRICK_TRI_VOLGRAD) {
        const float a0 = p.amin[0], a1 = p.amax[0];
        const float w = p.grad_out[r] * L * ((a1 - a0) / (float)(p.n_points - 1));
        if (w != 0.f) {
            if (fixq != 0.f)
                tri_owner_scatter(LdsAbsAddT<true>{fixq}, base, G.lof, G.hif, G.stridef, s, t, p.shift, p.eps,
                                  p.n_points, a0, a1, w);
            else
                tri_owner_scatter(LdsAbsAddT<false>{0.f}, base, G.lof, G.hif, G.stridef, s, t, p.shift, p.eps,
                                  p.n_points, a0, a1, w);
        }
        return;
    }
    if (MODE == BRICK_TRI_CHANNELS) {
        TriGeom T;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            T.lo[a] = G.lof[a];
            T.stridef[a] = G.stridef[a];
        }
        const float a0 = p.amin[0], a1 = p.amax[0];
        const unsigned N = (unsigned)(p.det_h * p.det_w), C = (unsigned)p.n_channels;
        const float step = (a1 - a0) / (float)(p.n_points - 1);  // renderers.py:235
        tri_brick_march_channels(LdsAbsFetch{}, base, T, p.D, s, t, p.shift, p.eps, p.n_points, a0, a1,
                                 BrickColumnFlush<true>{out, (b * C * N + pix) * 4u, N * 4u, C, L * step});
        return;
    }
    if (MODE == BRICK_TRI_CHANNELS_AUX) {
        TriGeom T;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            T.lo[a] = G.lof[a];
            T.stridef[a] = G.stridef[a];
        }
        const unsigned N = (unsigned)(p.det_h * p.det_w), C = (unsigned)p.n_channels;
        float sumT, rec[6];
        const TriLabelOf tl{p.labels, p.D};
        if (!tri_brick_march_weighted(LdsAbsFetch{}, [&](float rx, float ry, float rz) { return tl(rx, ry, rz, false, 0u); },
                                      base, T, p.D, s, t, p.shift, p.eps, p.n_points, p.amin[0], p.amax[0],
                                      BrickColumnWeight<true>{p.grad_out, (b * C * N + pix) * 4u, N * 4u, C},
                                      sumT, rec))
            return;
        unsafeAtomicAdd(aux + r, sumT);
#pragma unroll
        for (int k = 0; k < 6; ++k) unsafeAtomicAdd(aux + (unsigned)(k + 1) * p.aux_plane + r, rec[k]);
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
    DDRR_PROF_WAIT_VMEM();
    DDRR_PROF(PROF_LOADS);
    const StepEntry E = step_enter(SG, s, t, p.shift, p.eps, LdsAbsFetch::base_of(brick));
    DDRR_PROF(PROF_SETUP);
    if (MODE == BRICK_CHANNELS || MODE == BRICK_CHANNELS_WORDS) {
        const unsigned N = (unsigned)(p.det_h * p.det_w), C = (unsigned)p.n_channels;
        unsigned n4 = N * 4u;
        asm volatile("" : "+v"(n4));  // (in a vector register before the loop, not moved there per flush)
        if (E.hit)
            step_walk_channels(LdsAbsFetch{}, SG, E,
                               BrickColumnFlush<false, false>{out, (b * C * N + pix) * 4u, n4, C, 1.f}, L);
        DDRR_PROF(PROF_WALK);
        return;
    }
    float I = 0.f, a_end;
    int steps = 0;
    if (E.hit) steps = step_walk_fwd(LdsAbsFetch{}, SG, E, I, a_end);  // forward only
    DDRR_PROF(PROF_WALK);
    DDRR_PROF_COUNT(PROF_N_STEPS, (unsigned long long)__builtin_amdgcn_readfirstlane(steps));
    (void)steps;
    if (!E.hit) return;  // phase A's margin let a non-crossing ray through
    unsafeAtomicAdd(out + r, L * I);
    DDRR_PROF(PROF_DELIVER);
}

// (profiling builds: stamps of a brick's stages, 10 ns ticks from its claim: [n_bricks + 16 brick + k];
// how = 0: wave 0, 1: the latest wave.  k = 0 claimed, 1 / 2 rows + staging issued, 3 behind the staging
// barrier, 4 / 5 out of units, 6 / 7 walks done, 8 behind the gradient's barrier, 9 / 10 stored)
#if defined(DDRR_BRICK_PROFILE)
#define DDRR_STAMP(k, how)                                                                              \
    if (p.brick_times && lane == 0) {                                                                   \
        const unsigned dt_ = (unsigned)__builtin_amdgcn_s_memrealtime() - (unsigned)counter[3];         \
        unsigned *slot_ = p.brick_times + n_bricks + 16 * brick_id + (k);                               \
        if (how) atomicMax(slot_, dt_);                                                                 \
        else if (wave == 0) *slot_ = dt_;                                                               \
    }
#else
#define DDRR_STAMP(k, how) {}
#endif

template <int MODE>
__global__ __launch_bounds__(kBrickThreads) void siddon_brick_kernel(
    BrickArgs p, float *__restrict__ out, float *__restrict__ aux) {
    constexpr bool AUX = MODE == BRICK_FWD_AUX || MODE == BRICK_CHANNELS_AUX;
    // TRI: bricks of 31^3 base cells + halo (the marcher's forward); the marcher's volume
    // gradient (TRI_OWNER) runs on the plain 32^3 voxel bricks, see tri_brick.h
    constexpr bool TRI = MODE == BRICK_TRI_FWD || MODE == BRICK_TRI_FWD_AUX || MODE == BRICK_TRI_CHANNELS ||
                         MODE == BRICK_TRI_CHANNELS_AUX;
    constexpr bool TRI_OWNER = MODE == BRICK_TRI_VOLGRAD || MODE == BRICK_TRI_CHANNELS_VOLGRAD;
    constexpr bool GRAD = MODE == BRICK_VOLGRAD || MODE == BRICK_TRI_VOLGRAD || MODE == BRICK_CHANNELS_VOLGRAD ||
                          MODE == BRICK_TRI_CHANNELS_VOLGRAD;
    // ... with the labels in the accumulator's words
    constexpr bool GRADL = MODE == BRICK_CHANNELS_VOLGRAD;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *brick = reinterpret_cast<float *>(smem_raw);
    unsigned *queue = reinterpret_cast<unsigned *>(brick + brick_floats(p.lay));
    float *rows = reinterpret_cast<float *>(queue + kBrickWaves * kBuckets * kQueueCap);
    int *counter = reinterpret_cast<int *>(rows + kPoseChunk * kRowWords);  // [0] unit, [1] brick, [2] brick not empty

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const BrickGrid bg = TRI ? tri_brick_grid(p.D) : brick_grid(p.D);
    const int n_bricks = bg.nx * bg.ny * bg.nz;
    const int N = p.det_h * p.det_w;
    // wave-private: written and read by lanes of the same wave only.  LDS operations of a
    // wave execute in order; wave_fence() keeps the compiler from reordering them.
    unsigned *myq = queue + wave * kBuckets * kQueueCap;
    const unsigned pix_mask = (1u << p.pix_bits) - 1u;
    const int n_chunks = (p.B + kPoseChunk - 1) / kPoseChunk;
    // (quads of four voxels from dword-aligned addresses, brick_shared.h quad_load: any D.z)
    // (fewer than four slices: quads of rows before the last one would be clamped too -- scalar staging)
    const bool vec_ok = quads_serve(p.D) && (long)p.D.x * p.D.y * p.D.z >= 4 &&
                        (reinterpret_cast<uintptr_t>(p.vol) & 3) == 0;
    const bool vec_out = GRAD && (p.D.z & 3) == 0 && (reinterpret_cast<uintptr_t>(p.g_volume) & 15) == 0;
    const bool labels_dword_ok = (MODE == BRICK_CHANNELS || MODE == BRICK_TRI_CHANNELS ||
                                  MODE == BRICK_CHANNELS_AUX || GRADL) &&
                                 (reinterpret_cast<uintptr_t>(p.labels) & 3) == 0;
    // fixed-point scale of the LDS accumulator (volume-gradient modes): the largest sum a
    // voxel can receive is n_sum (contributions) * wmax (each) -- volgrad_prepare_kernel
    float fixq = 0.f;
    if (GRAD && !(p.dbg & 32)) {
        const float wmax = __uint_as_float((unsigned)p.work[1]);
        const float n_sum = reinterpret_cast<const float *>(p.work)[2];
        // (the channel gradient keeps the voxel's label in the word's low byte: 24 bits)
        if (wmax > 0.f && wmax < 1e30f && n_sum > 0.f && n_sum <= 16384.f)
            fixq = (GRADL ? 7.8e6f : 2.0e9f) / (n_sum * wmax);
    }
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
    if (GRAD && (p.dbg & (1 << 23))) fixq = -fixq;  // (LdsAbsAddT: plain stores)
#endif

  // Persistent workgroups: bricks are handed out by a global counter, so a CU that drew
  // light bricks (far from the sources: fewer rays cross them) simply takes more of them.
  BrickProf prof;
#if defined(DDRR_BRICK_PROFILE)
  prof.start();
#endif
  for (;;) {
    __syncthreads();  // every wave is done with the previous brick's LDS
    DDRR_PROF(PROF_BARRIER);
#if defined(DDRR_BRICK_PROFILE)
    if (tid == 0) counter[3] = (int)(unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    if (tid == 0) {
        counter[1] = atomicAdd(p.work, 1);
        counter[2] = 0;  // "a staged voxel is non-zero"
    }
    __syncthreads();
    if (counter[1] >= n_bricks) break;
    const int brick_id = p.order ? p.order[counter[1]] : counter[1];  // (heaviest first)
    DDRR_PROF(PROF_CLAIM);
    DDRR_STAMP(0, 0)
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
    const StepGeom SG = step_geom(box, p.lay);
    if (TRI) {
#pragma unroll
        for (int a = 0; a < 3; ++a) G.lof[a] = (float)box.lo[a];
    }
    const float nscale = (TRI || TRI_OWNER) ? (float)(p.n_points - 1) / (p.amax[0] - p.amin[0]) : 0.f;
    int qn0 = 0, qn1 = 0, qn2 = 0;  // hits waiting per length class (wave-uniform)
    bool brick_empty = false;        // every staged voxel is zero (set with the first chunk)
    // the float backward records: length classes per group of 8 lanes = 8 adjacent pixels (below)
    // (the channel render gains nothing from it: runs of 8 adjacent pixels, whose label changes
    // coincide, measured 0.314 vs 0.302 ms at 8 poses -- profiles/r04/channels.txt)
    const bool GROUPED = ((AUX && p.rec_q == 0.f) || MODE == BRICK_TRI_FWD_AUX ||
                          MODE == BRICK_TRI_CHANNELS_AUX) && !(p.dbg & 8);

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
        // (recomputed per brick from an opaque copy of the thread id: hoisted out of the brick
        // loop these few values are what tips the channel instance into scratch spills)
        int tid_here = tid;
        asm volatile("" : "+v"(tid_here));
        const int q4 = (tid_here & 7) * 4, z = box.lo[2] + q4;
        float *const d0 = brick + q4;
        // (the marcher's channel backward stages the volume's own values and reads the label map)
        constexpr bool LABELS = MODE == BRICK_CHANNELS || MODE == BRICK_TRI_CHANNELS ||
                                MODE == BRICK_CHANNELS_AUX || GRADL;
        // (Siddon channels: labels without a channel are staged as value 0 | label 0; the channel
        // gradient's accumulator: an integer 0 over the label, or a float 0 where there is no bound)
        auto pack_word = [&](float v, unsigned lab) {
            if (GRADL) return fixq != 0.f ? bits_as_float(lab & 0xffu) : 0.f;
            return MODE != BRICK_TRI_CHANNELS
                       ? pack_voxel_label_below(v, lab, (unsigned)p.n_channels)
                                          : pack_voxel_label(v, lab);
        };
        const bool stage_vec = ch == 0 && !GRAD && !TRI && vec_ok;
        if (tid < nb) {
            const PoseGrid pg = pose_grid(p.source + (long)(b0 + tid) * 3,
                                          p.target + (long)(b0 + tid) * N * 3, p.det_h, p.det_w);
            PixBox pb = project_brick_grid(pg, p.det_h, p.det_w, cells, p.shift);
            if (GROUPED) pb = align_pixbox_rows(pb, p.det_w);
            BrickRow r = brick_row(pg, pb, cells, p.shift, p.eps, nscale);
            if (GRAD && !(p.dbg & 16)) r.perm_k = scatter_perm_k(r.w, r.count);
            *reinterpret_cast<BrickRow *>(rows + tid * kRowWords) = r;
        }
        if (tid == 0) counter[0] = 0;
        unsigned nz = 0u;  // OR of the bits of every voxel this thread stages
        DDRR_PROF(PROF_ROWS);
        // (with labels: two rounds of four quads -- all eight at once plus their labels do not
        // fit the register budget next to the kernel's loop invariants)
        constexpr int kRounds = LABELS ? 2 : 1, kPer = kQuads / kRounds;
        if (stage_vec) {
#pragma unroll
            for (int h0 = 0; h0 < kQuads; h0 += kPer) {
                quad_u32x4 qv[kPer];
                unsigned ql[LABELS ? kPer : 1] = {};  // BRICK_CHANNELS: the quads' four labels
#pragma unroll
                for (int i = 0; i < kPer; ++i) {
                    const int row = (tid >> 3) + (h0 + i) * (kBrickThreads >> 3);
                    const int lx = row / BRICK, ly = row - lx * BRICK;
                    const int x = box.lo[0] + lx, y = box.lo[1] + ly;
                    // clamped (always readable) address; what lies outside is zeroed below
                    const int xc = x < p.D.x ? x : p.D.x - 1, yc = y < p.D.y ? y : p.D.y - 1;
                    quad_load<LABELS>(p.vol, p.labels, p.D, ((long)xc * p.D.y + yc) * p.D.z + z, qv[i],
                                      ql[LABELS ? i : 0]);
                }
#pragma unroll
                for (int i = 0; i < kPer; ++i) {
                    const int row = (tid >> 3) + (h0 + i) * (kBrickThreads >> 3);
                    const int lx = row / BRICK, ly = row - lx * BRICK;
                    const int x = box.lo[0] + lx, y = box.lo[1] + ly;
                    const bool in = x < box.hi[0] && y < box.hi[1];
                    float *d = d0 + lx * p.lay.sx + ly * p.lay.sy;
                    quad_u32x4 w = qv[i];
                    unsigned lab4 = ql[LABELS ? i : 0];
                    quad_fix(p.D, x, y, z, w, lab4);
                    float4 q = make_float4(bits_as_float(w.x), bits_as_float(w.y), bits_as_float(w.z),
                                           bits_as_float(w.w));
                    if (LABELS) {
                        q.x = pack_word(q.x, lab4);
                        q.y = pack_word(q.y, lab4 >> 8);
                        q.z = pack_word(q.z, lab4 >> 16);
                        q.w = pack_word(q.w, lab4 >> 24);
                    }
                    d[0] = in && z < box.hi[2] ? q.x : 0.f;
                    d[1] = in && z + 1 < box.hi[2] ? q.y : 0.f;
                    d[2] = in && z + 2 < box.hi[2] ? q.z : 0.f;
                    d[3] = in && z + 3 < box.hi[2] ? q.w : 0.f;
                    nz |= __float_as_uint(d[0]) | __float_as_uint(d[1]) | __float_as_uint(d[2]) |
                          __float_as_uint(d[3]);
                }
                if (LABELS) __builtin_amdgcn_sched_barrier(0);  // keep the rounds apart
            }
        } else if (ch == 0) {
            // general path (halo bricks of the trilinear marcher, volumes of fewer than four voxels,
            // and the zero fill of the gradient accumulator), kFly quads in flight.  Every load is issued
            // unconditionally from a clamped (always readable) address and what lies outside is
            // zeroed afterwards: loads under per-lane conditions end up in separate round trips
            // (measured with the labels: staging 4x the plain brick's).  The quad's four labels
            // come as two aligned dwords + a byte alignment -- the halo shifts the rows to every
            // alignment -- or, at the array's last bytes and for an unaligned label pointer, as
            // four byte loads.
            constexpr int kFly = LABELS ? 2 : 4;  // quads in flight per thread
            const long total4 = ((long)p.D.x * p.D.y * p.D.z) & ~3L;
            // Plain halo bricks (the marcher's forward, with and without the record): a quad is ONE
            // 16-byte load from a dword-aligned address, like the Siddon bricks' (brick_shared.h
            // quad_load) -- the halo puts the rows at every alignment, and as four dword loads a
            // brick is 512 wave-loads of which every cache line is asked for four times.  The quad
            // is read from inside its row (z clamped to 0 .. D.z - 4) and shifted where that moved it
            // (the bricks at the volume's two ends only); what lies outside is masked below as before.
            const bool quads = !GRAD && !LABELS && p.D.z >= 4 && (reinterpret_cast<uintptr_t>(p.vol) & 3) == 0;
            const int zq = clampi(z, 0, p.D.z - 4), zsh = z - zq;
#pragma unroll 1
            for (int h = 0; h < kQuads; h += kFly) {
                float v[kFly][4];
                unsigned lw[LABELS ? kFly : 1][2], lb[LABELS ? kFly : 1][4];
                bool in_xy[kFly], wide[kFly];
                int sh[kFly];
#pragma unroll
                for (int it = 0; it < kFly; ++it) {
                    const int row = (tid >> 3) + (h + it) * (kBrickThreads >> 3);
                    const int lx = row / BRICK, ly = row - lx * BRICK;
                    const int x = box.lo[0] + lx, y = box.lo[1] + ly;
                    in_xy[it] = (!GRAD || GRADL) && x >= 0 && y >= 0 && x < box.hi[0] && y < box.hi[1];
                    const int xc = clampi(x, 0, p.D.x - 1), yc = clampi(y, 0, p.D.y - 1);
                    const long rowbase = ((long)xc * p.D.y + yc) * p.D.z;
                    const float *g = p.vol + rowbase;
                    if (quads) {
                        const quad_u32x4 q = *reinterpret_cast<const quad_u32x4_a4 *>(g + zq);
                        v[it][0] = bits_as_float(q.x), v[it][1] = bits_as_float(q.y);
                        v[it][2] = bits_as_float(q.z), v[it][3] = bits_as_float(q.w);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[it][k] = GRAD ? 0.f : g[clampi(z + k, 0, p.D.z - 1)];
                    }
                    if (LABELS) {
                        const long a = rowbase + (z > 0 ? z : 0), a4 = a & ~3L;
                        wide[it] = labels_dword_ok && z >= 0 && a4 + 8 <= total4;
                        sh[it] = (int)(a - a4);
                        const unsigned *w2 = reinterpret_cast<const unsigned *>(
                            p.labels + (wide[it] ? a4 : 0L));
                        if (labels_dword_ok && total4 >= 8) {
                            lw[it][0] = w2[0];
                            lw[it][1] = w2[1];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) lb[it][k] = 0u;
                    }
                }
                if (LABELS) {
                    // (rare: the last bytes of the array, the marcher's halo at z = -1)
                    bool bytes = false;
#pragma unroll
                    for (int it = 0; it < kFly; ++it) bytes = bytes || (in_xy[it] && !wide[it]);
                    if (__ballot(bytes)) {
#pragma unroll
                        for (int it = 0; it < kFly; ++it) {
                            const int row = (tid >> 3) + (h + it) * (kBrickThreads >> 3);
                            const int lx = row / BRICK, ly = row - lx * BRICK;
                            const int xc = clampi(box.lo[0] + lx, 0, p.D.x - 1);
                            const int yc = clampi(box.lo[1] + ly, 0, p.D.y - 1);
                            const unsigned char *lr = p.labels + ((long)xc * p.D.y + yc) * p.D.z;
#pragma unroll
                            for (int k = 0; k < 4; ++k) lb[it][k] = lr[clampi(z + k, 0, p.D.z - 1)];
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < kFly; ++it) {
                    const int row = (tid >> 3) + (h + it) * (kBrickThreads >> 3);
                    const int lx = row / BRICK, ly = row - lx * BRICK;
                    float *d = d0 + lx * p.lay.sx + ly * p.lay.sy;
                    if (quads && zsh != 0) {  // voxel z + k is element k + zsh of what was read
                        const float q0 = v[it][0], q1 = v[it][1], q2 = v[it][2], q3 = v[it][3];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int e = k + zsh;
                            v[it][k] = e == 0 ? q0 : (e == 1 ? q1 : (e == 2 ? q2 : (e == 3 ? q3 : 0.f)));
                        }
                    }
                    unsigned lab4 = 0u;
                    if (LABELS)
                        lab4 = wide[it] ? __builtin_amdgcn_alignbyte(lw[it][1], lw[it][0], (unsigned)sh[it])
                                        : (lb[it][0] | (lb[it][1] << 8) | (lb[it][2] << 16) | (lb[it][3] << 24));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool in = in_xy[it] && z + k >= 0 && z + k < box.hi[2];
                        float val = in ? v[it][k] : 0.f;
                        if (LABELS && in) val = pack_word(val, lab4 >> (8 * k));
                        d[k] = val;
                        nz |= __float_as_uint(val);
                    }
                }
            }
        }
        DDRR_PROF(PROF_STORE);
        // Empty space: a brick of zeros (air around the patient: HU -> density maps it to exactly
        // 0, reference data.py:214-227) adds nothing to any line integral, record or channel, so
        // none of its candidates is looked at.  (Sign and, for the channel words, label bits do
        // not make a voxel non-zero; the volume-gradient modes have no such shortcut.)
        constexpr unsigned kValueBits = LABELS ? 0x7fffff00u : 0x7fffffffu;
        if (!GRAD && ch == 0 && (nz & kValueBits) != 0u) counter[2] = 1;  // (cleared with the claim)
        if (ch == 0) DDRR_STAMP(1, 0)
        if (ch == 0) DDRR_STAMP(2, 1)
        __syncthreads();
        if (ch == 0) DDRR_STAMP(3, 0)
        if (!GRAD && ch == 0) brick_empty = counter[2] == 0;
        // units per pose -> inclusive prefix, held by every wave in registers (lane k: pose k)
        int incl = lane < nb ? (reinterpret_cast<const BrickRow *>(rows + lane * kRowWords)->count +
                                63) >> 6
                             : 0;
#pragma unroll
        for (int o = 1; o < kPoseChunk; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            incl += lane >= o ? up : 0;
        }
        const int units = brick_empty ? 0 : __builtin_amdgcn_readlane(incl, kPoseChunk - 1);
        int cur = 0, cur_lo = 0, cur_hi = __builtin_amdgcn_readlane(incl, 0);
        DDRR_PROF(PROF_STAGE);
        for (;;) {
            int u = 0;
            if (lane == 0) u = atomicAdd(&counter[0], 1);
            u = uni(u);
            const bool drain = u >= units;  // no unit left in this chunk
            if (drain && !last_chunk) break;
            if (drain) {
                DDRR_STAMP(4, 0)
                DDRR_STAMP(5, 1)
            }
            if (!drain) {
                while (u >= cur_hi) {  // units arrive in increasing order: forward cursor
                    ++cur;
                    cur_lo = cur_hi;
                    cur_hi = __builtin_amdgcn_readlane(incl, uni(cur));
                }
                const BrickRow r = *reinterpret_cast<const BrickRow *>(rows + cur * kRowWords);
                DDRR_PROF(PROF_PULL);
                DDRR_PROF_COUNT(PROF_N_UNITS, 1);
                int local = (u - cur_lo) * 64 + lane;
                const bool valid = local < uni(r.count);
                if (GRAD && valid)
                    local = scatter_perm(local, uni(r.perm_k), uni(r.count),
                                         1.0f / (float)uni(r.count));
                int pix = 0;
                float n_est = 0.f;
                bool hit = valid && brick_candidate(r, local, p.det_w, pix, n_est);
                // a subsample of the detector (reference drr.py:36-39: p_subsample): pixels whose bit is
                // not set are no candidates
                if (p.pix_mask != nullptr) hit = hit && ((p.pix_mask[pix >> 5] >> (pix & 31)) & 1u) != 0u;
                // With the float backward record (5 atomics per hit instead of 1; the packed
                // record's 3 are cheap enough to go per lane) the length class
                // of a hit is that of the longest hit among its 8 neighbours in candidate
                // order (consecutive pixels of a detector row): a batch is then made of runs
                // of >= 8 adjacent pixels and its atomics touch few cache lines -- their cost
                // is per line, not per lane (measured: 3.55 -> 2.74 ms; without the record
                // the per-lane classes win, 1.87 vs 2.01 ms).
                float n_grp = hit ? n_est : 0.f;
                if (GROUPED) {
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
                DDRR_PROF_COUNT(PROF_N_HITS, __popcll(m0) + __popcll(m1) + __popcll(m2));
                DDRR_PROF(PROF_PHASE_A);
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
                DDRR_PROF(PROF_POP);
                DDRR_PROF_COUNT(PROF_N_BATCH, 1);
                brick_item<MODE>(p, brick, G, SG, lane < n, e >> p.pix_bits, e & pix_mask, fixq,
                                 out, aux, prof);
                wave_fence();
            }
            if (drain) break;
        }
    }
    DDRR_STAMP(6, 0)
    DDRR_STAMP(7, 1)
    if (GRAD) {
        // every ray of every pose has been scattered into the LDS accumulator, and the brick
        // owns its voxels (Siddon bricks, and the marcher's owner bricks): the gradient is
        // complete and is stored, 16 bytes per thread
        __syncthreads();
        DDRR_STAMP(8, 0)
        const float unfix = fixq != 0.f ? 1.0f / fixq : 0.f;  // (one division per brick, not one per voxel)
        for (int row = tid >> 3; row < BRICK * BRICK; row += kBrickThreads >> 3) {
            const int lx = row / BRICK, ly = row - lx * BRICK, q4 = (tid & 7) * 4;
            const int x = box.lo[0] + lx, y = box.lo[1] + ly, z = box.lo[2] + q4;
            if (x < box.hi[0] && y < box.hi[1]) {
                const float *src = brick + lx * p.lay.sx + ly * p.lay.sy + q4;
                float *g = p.g_volume + ((long)x * p.D.y + y) * p.D.z + z;
                float val[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    val[k] = fixq != 0.f ? (float)(GRADL ? __float_as_int(src[k]) >> 8 : __float_as_int(src[k])) * unfix
                                         : src[k];
                if (vec_out && z + 4 <= box.hi[2]) {
                    *reinterpret_cast<float4 *>(g) = make_float4(val[0], val[1], val[2], val[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (z + k < box.hi[2]) g[k] = val[k];
                }
            }
        }
        DDRR_STAMP(9, 0)
        DDRR_STAMP(10, 1)
    }
  }
#if defined(DDRR_BRICK_PROFILE)
  DDRR_PROF(PROF_BARRIER);
  if (lane == 0 && p.prof)
      for (int i = 0; i < 16; ++i) atomicAdd(p.prof + i, prof.t[i]);
#endif
}

// ------------------------------------------- volume-gradient fixed-point bound
// work[1] = bits of max over rays of the largest single contribution a ray can make to a
// voxel; work[2] = float: bound on the number of such contributions a voxel can receive in
// this launch (sum over poses of the rays that can cross one voxel); see LdsAbsAddT.
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
    const float *__restrict__ amax, int *__restrict__ work, int n_channels) {
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
        float g = 0.f;
        if (n_channels > 0) {  // (B, C, N): the largest weight any label of the ray's column carries
            for (int ch = 0; ch < n_channels; ++ch)
                g = fmaxf(g, fabsf(grad_out[((long)b * n_channels + ch) * N + n]));
        } else {
            g = fabsf(grad_out[r]);
        }
        wmax = fmaxf(wmax, g * L * c);
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

// Per-ray alpha bound A and the scale q of the packed record (record_pack.h): planes 5 and 6.
__global__ __launch_bounds__(kBlock) void record_prepare_kernel(
    const float *__restrict__ source, const float *__restrict__ target, long R, int N, Dims D,
    float shift, float eps, float q, float *__restrict__ aux) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    if (r == 0) aux[6 * R] = q;
    if (r >= R) return;
    const long b = r / N;
    const float s[3] = {source[b * 3], source[b * 3 + 1], source[b * 3 + 2]};
    const float t[3] = {target[r * 3], target[r * 3 + 1], target[r * 3 + 2]};
    aux[5 * R + r] = record_alpha_bound(D, s, t, shift, eps);
}

// out = L * I from plane 0 of the Siddon planar record (the record launch leaves `out` alone:
// one atomic less per ray and brick).
// (blocked: the float record of record_layout.h; else plane I of the packed record)
__global__ __launch_bounds__(kBlock) void siddon_out_from_record_kernel(
    const float *__restrict__ aux, int blocked, const float *__restrict__ img, long R,
    float *__restrict__ out) {
    const long r = (long)blockIdx.x * kBlock + threadIdx.x;
    if (r < R) out[r] = (img ? img[r] : 1.f) * aux[blocked ? rec_index(r, 0) : r];
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

// LDS layout of a brick (floats): rows padded 32 -> 33, planes 32*33 -> 1057, so that
// x-, y- and z-neighbours all fall in different banks; length classes of brick hits
// (estimated plane crossings inside the brick; for the marcher: samples per brick).
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
BrickLayout g_brick_layout = {33, 32 * 33 + 1};
float g_tri_t1 = 10.f, g_tri_t2 = 22.f;
#else
constexpr BrickLayout g_brick_layout = {33, 32 * 33 + 1};
constexpr float g_tri_t1 = 10.f, g_tri_t2 = 22.f;
#endif

}  // namespace

namespace ddrr_brick {

#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
float g_brick_t1 = 18.f, g_brick_t2 = 40.f;
int g_brick_dbg = 0;
int g_brick_variant = -2;
float g_brick_sq_width = 8.f;
const int *g_brick_order = nullptr;
unsigned *g_brick_times = nullptr;
int g_brick_split_t = 0, g_brick_split_s = 1;
#endif
#if defined(DDRR_BRICK_PROFILE)
unsigned long long *g_brick_prof = nullptr;  // 16 device counters, see BrickProf
#endif

// This launch's device-side state lives in the CALLER's launch workspace
// (ddrr_brick_launch_workspace_bytes): [header: 64 words -- {brick counter, wmax bits, n_sum, -},
// zeroed here on the launch's stream] [weights: one float per 32^3 brick] [hand-out order: one
// int per 32^3 brick].  Nothing on the device is shared between launches: two streams, or a
// captured graph and eager launches, can never meet on one brick counter.  (Per process only the
// CU count of each device is remembered.)
long brick_launch_workspace_bytes(int dx, int dy, int dz) {
    const long n32 = (long)((dx + 31) / 32) * ((dy + 31) / 32) * ((dz + 31) / 32);
    return 256 + (n32 * 2 * (long)sizeof(int) + 255) / 256 * 256;
}

int brick_launch_resources(hipStream_t st, void *launch_ws, int dx, int dy, int dz, int &n_cu_out,
                           int *&work, int **order_ws, int *order_cap, bool zero_work) {
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static int n_cu[kMaxDev] = {0};
    hipError_t e;
    int dev = 0;
    if (!launch_ws) return fail(-1, "null launch workspace (ddrr_brick_launch_workspace_bytes)");
    if (reinterpret_cast<uintptr_t>(launch_ws) & 15) return fail(-1, "launch workspace not 16-byte aligned");
    if ((e = hipGetDevice(&dev)) != hipSuccess) return fail_hip(e, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDev) return fail(-1, "device index out of range");
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!n_cu[dev] &&
            (e = hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev)) !=
                hipSuccess)
            return fail_hip(e, "hipDeviceGetAttribute");
        n_cu_out = n_cu[dev];
    }
    work = reinterpret_cast<int *>(launch_ws);
    if (order_ws) {
        *order_ws = work + 64;
        *order_cap = (int)(((long)((dx + 31) / 32) * ((dy + 31) / 32)) * ((dz + 31) / 32));
    }
    // (zero_work = false: the caller clears the counter itself, bricks_fwd.hip launch_cfg)
    if (zero_work && (e = hipMemsetAsync(work, 0, 4 * sizeof(int), st)) != hipSuccess)
        return fail_hip(e, "hipMemsetAsync");
    return 0;
}

int launch_bricks(int mode, const float *volume, int dx, int dy, int dz, const float *source,
                  const float *target, const float *img, const float *grad_out, int B, int det_h,
                  int det_w, float voxel_shift, float eps, float *out, float *aux,
                  float *g_volume, hipStream_t st, void *launch_ws, const char *who, int n_points,
                  const float *amin, const float *amax, float rec_q,
                  const unsigned char *labels, int n_channels, const unsigned *pix_mask) {
    const int N = det_h * det_w;
    BrickArgs p;
    p.pix_mask = pix_mask;
    p.fingerprint = nullptr;
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
    p.rec_q = rec_q;
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
    p.prof = nullptr;
    p.labels = labels;
    p.n_channels = n_channels;
    p.ranges = nullptr;
    p.fallback = nullptr;
    p.ws_header = nullptr;
    p.ranges_valid = 0;
    p.brick_times = nullptr;
#if defined(DDRR_BRICK_PROFILE)
    p.brick_times = g_brick_times;
#endif
    p.order = nullptr;
    p.order_ws = nullptr;
    p.order_cap = 0;
    p.split_t = 0;
    p.split_s = 1;
#if defined(DDRR_BRICK_PROFILE)
    p.prof = g_brick_prof;
#endif
    if (mode == BRICK_TRI_FWD || mode == BRICK_TRI_VOLGRAD || mode == BRICK_TRI_FWD_AUX ||
        mode == BRICK_TRI_CHANNELS || mode == BRICK_TRI_CHANNELS_AUX || mode == BRICK_TRI_CHANNELS_VOLGRAD) {
        p.t1 = g_tri_t1;
        p.t2 = g_tri_t2;
    }
    const size_t lds = brick_lds_bytes(p.lay);
    hipError_t e;
    // the raised dynamic-LDS limit of the brick kernels, once per device
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static bool attr_set[kMaxDev] = {false};
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return fail_hip(e, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDev) return fail(-1, "device index out of range");
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            const void *fns[13] = {
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_CHANNELS_WORDS>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_TRI_CHANNELS_VOLGRAD>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_CHANNELS_VOLGRAD>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_TRI_CHANNELS_AUX>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_CHANNELS_AUX>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_TRI_CHANNELS>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_CHANNELS>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_TRI_FWD_AUX>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_FWD>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_FWD_AUX>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_VOLGRAD>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_TRI_FWD>),
                reinterpret_cast<const void *>(&siddon_brick_kernel<BRICK_TRI_VOLGRAD>)};
            for (const void *fn : fns)
                if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024)) != hipSuccess)
                    return fail_hip(e, "hipFuncSetAttribute");
            attr_set[dev] = true;
        }
    }
    int n_cu_dev = 0;
    if (int rc = brick_launch_resources(st, launch_ws, dx, dy, dz, n_cu_dev, p.work, &p.order_ws, &p.order_cap))
        return rc;
    if (g_brick_dbg & 512) p.order_ws = nullptr;  // (bricks in id order)
    if (mode == BRICK_VOLGRAD || mode == BRICK_TRI_VOLGRAD || mode == BRICK_CHANNELS_VOLGRAD ||
        mode == BRICK_TRI_CHANNELS_VOLGRAD) {
        const int tri = mode == BRICK_TRI_VOLGRAD || mode == BRICK_TRI_CHANNELS_VOLGRAD;
        int bx = (N + kBlock - 1) / kBlock;
        bx = bx > 64 ? 64 : bx;
        hipLaunchKernelGGL(volgrad_prepare_kernel, dim3(bx, B), dim3(kBlock), 0, st, tri, source,
                           target, img, grad_out, N, det_w, p.D, voxel_shift, eps, n_points, amin,
                           amax, p.work,
                           mode == BRICK_CHANNELS_VOLGRAD || mode == BRICK_TRI_CHANNELS_VOLGRAD ? n_channels : 0);
    }
    const BrickGrid bg = (mode == BRICK_TRI_FWD || mode == BRICK_TRI_FWD_AUX ||
                          mode == BRICK_TRI_CHANNELS || mode == BRICK_TRI_CHANNELS_AUX)
                             ? tri_brick_grid(p.D)
                                                                              : brick_grid(p.D);
    const int n_bricks = bg.nx * bg.ny * bg.nz;
    // the Siddon modes hand their bricks out heaviest first like the forward kernels (the marcher's
    // bricks are cells + halo with their own boxes: id order)
    if (mode == BRICK_FWD || mode == BRICK_FWD_AUX || mode == BRICK_VOLGRAD || mode == BRICK_CHANNELS ||
        mode == BRICK_CHANNELS_WORDS || mode == BRICK_CHANNELS_AUX || mode == BRICK_CHANNELS_VOLGRAD)
        order_bricks(p, BRICK, BRICK, BRICK, bg.ny, bg.nz, n_bricks, n_cu_dev, st);
    // (the marcher's volume gradient: owner bricks = the plain grid; a brick's walks are long
    // whatever the pose count)
    if (mode == BRICK_TRI_VOLGRAD) order_bricks(p, BRICK, BRICK, BRICK, bg.ny, bg.nz, n_bricks, n_cu_dev, st, false, 1);
    const dim3 grid(n_bricks < n_cu_dev ? n_bricks : n_cu_dev), block(kBrickThreads);
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
    else if (mode == BRICK_CHANNELS)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_CHANNELS>, grid, block, lds, st, p, out, aux);
    else if (mode == BRICK_CHANNELS_WORDS)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_CHANNELS_WORDS>, grid, block, lds, st, p, out, aux);
    else if (mode == BRICK_CHANNELS_AUX)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_CHANNELS_AUX>, grid, block, lds, st, p, out, aux);
    else if (mode == BRICK_TRI_CHANNELS)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_TRI_CHANNELS>, grid, block, lds, st, p, out, aux);
    else if (mode == BRICK_TRI_CHANNELS_AUX)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_TRI_CHANNELS_AUX>, grid, block, lds, st, p, out,
                           aux);
    else if (mode == BRICK_CHANNELS_VOLGRAD)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_CHANNELS_VOLGRAD>, grid, block, lds, st, p, out,
                           aux);
    else if (mode == BRICK_TRI_CHANNELS_VOLGRAD)
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_TRI_CHANNELS_VOLGRAD>, grid, block, lds, st, p,
                           out, aux);
    else
        hipLaunchKernelGGL(siddon_brick_kernel<BRICK_VOLGRAD>, grid, block, lds, st, p, out, aux);
    return finish(who);
}

}  // namespace ddrr_brick

extern "C" {

#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
// tools/ builds only (tools/build_variant.py): process-wide experiment switches.  The product
// library does not contain them: its entry points keep no state between calls but the
// per-device launch resources (include/diffdrr_hip.h).
//   layout: LDS strides (floats) of a staged brick; sy >= 32, sx >= 32 * sy
//   debug flags: 8 per-lane length classes also with the record (no groups of 8 pixels),
//                16 no scatter permutation, 32 float LDS accumulation,
//                64 the float record delivered with one atomic instead of five (timing only: wrong record)
int ddrr_set_brick_layout(int sy, int sx) {
    if (sy < BRICK || sx < BRICK * sy) return -1;
    BrickLayout lay = {sy, sx};
    if (brick_lds_bytes(lay) > 160 * 1024) return -1;
    g_brick_layout = lay;
    return 0;
}
int ddrr_set_brick_debug(int flags) {
    g_brick_dbg = flags;
    return 0;
}
int ddrr_set_brick_classes(float t1, float t2) {
    g_brick_t1 = t1;
    g_brick_t2 = t2;
    return 0;
}
#endif
#if defined(DDRR_BRICK_PROFILE)
// zero / read the phase counters of the brick launches since the last reset
int ddrr_brick_profile_reset() {
    if (!g_brick_prof && hipMalloc(reinterpret_cast<void **>(&g_brick_prof), 20 * 8) != hipSuccess)
        return -1;
    return hipMemset(g_brick_prof, 0, 20 * 8) == hipSuccess ? 0 : -1;
}
// (slots 16 .. 19: sums of the waves' start / end ticks, earliest start (inverted), latest end
// -- bricks_fwd.hip only)
int ddrr_brick_profile_read20(unsigned long long *host20) {
    if (!g_brick_prof) return -1;
    return hipMemcpy(host20, g_brick_prof, 20 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
int ddrr_brick_profile_read(unsigned long long *host16) {
    if (!g_brick_prof) return -1;
    return hipMemcpy(host16, g_brick_prof, 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

static int siddon_forward_bricks_impl(const float *volume, int dx, int dy, int dz, const float *source,
                                      const float *target, const float *img, int B, int det_h,
                                      int det_w, float voxel_shift, float eps, float *out, float *aux,
                                      float record_vmax, int brick_storage, float *brick_ranges,
                                      int ranges_valid, void *launch_ws, const unsigned *pixel_mask,
                                      void *stream) {
    const int N = det_h * det_w;
    if (int rc = check_common(volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (!out && !aux) return fail(-1, "null out pointer");  // (the record alone: ddrr_siddon_ncc_forward forms the image)
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if (!(record_vmax >= 0.f)) return fail(-1, "record_vmax must be >= 0");
    if (brick_storage != DDRR_BRICKS_F32 && brick_storage != DDRR_BRICKS_Q16 &&
        brick_storage != DDRR_BRICKS_Q16_PACKED)
        return fail(-1, "brick_storage must be DDRR_BRICKS_F32, DDRR_BRICKS_Q16 or DDRR_BRICKS_Q16_PACKED");
    if (brick_storage != DDRR_BRICKS_F32 && !brick_ranges)
        return fail(-1, "DDRR_BRICKS_Q16 needs the brick_ranges workspace");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const long R = (long)B * N;
    // (DDRR_BRICKS_CLEARED: the record / image and the brick counter are zero already)
    const bool cleared = (ranges_valid & DDRR_BRICKS_CLEARED) != 0;
    ranges_valid &= 1;
    const bool packed = aux && record_vmax > 0.f;
    // (the packed record's planes 5, 6 are written by record_prepare_kernel)
    const size_t fill = !aux ? (size_t)R : (packed ? (size_t)R * 5 : (size_t)rec_blocked_floats(R));
    // (the packed record's preparation writes behind the zeros: cleared here; else with the brick
    // counter, by one launch: launch_fwd_bricks)
    float rec_q = 0.f;
    if (packed) {
        const hipError_t e = cleared ? hipSuccess : hipMemsetAsync(aux, 0, sizeof(float) * fill, st);
        if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
        rec_q = record_scale(record_vmax, Dims{dx, dy, dz});
        hipLaunchKernelGGL(record_prepare_kernel, dim3((unsigned)((R + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, st, source, target, R, N, Dims{dx, dy, dz}, voxel_shift,
                           eps, rec_q, aux);
    }
    const int packed_bricks = brick_storage == DDRR_BRICKS_Q16_PACKED;
    if (int rc = launch_fwd_bricks(packed_bricks ? DDRR_BRICKS_Q16 : brick_storage, packed_bricks,
                                   brick_ranges, ranges_valid, volume, dx, dy, dz,
                                   source, target, img, B, det_h, det_w, voxel_shift, eps, out, aux,
                                   rec_q, st, launch_ws, "ddrr_siddon_forward_bricks",
                                   packed || cleared ? nullptr : (aux ? aux : out),
                                   cleared ? -1 : (packed ? 0 : (long)fill), pixel_mask))
        return rc;
    if (!aux) return 0;
    if (out)
        hipLaunchKernelGGL(siddon_out_from_record_kernel, dim3((unsigned)((R + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, st, aux + (packed ? 4 * R : 0), packed ? 0 : 1, img, R, out);
    return finish("ddrr_siddon_forward_bricks");
}

int ddrr_siddon_forward_bricks(const float *volume, int dx, int dy, int dz, const float *source,
                               const float *target, const float *img, int B, int det_h,
                               int det_w, float voxel_shift, float eps, float *out, float *aux,
                               float record_vmax, int brick_storage, float *brick_ranges,
                               int ranges_valid, void *launch_ws, void *stream) {
    return siddon_forward_bricks_impl(volume, dx, dy, dz, source, target, img, B, det_h, det_w, voxel_shift,
                                      eps, out, aux, record_vmax, brick_storage, brick_ranges, ranges_valid,
                                      launch_ws, nullptr, stream);
}

int ddrr_siddon_forward_bricks_masked(const float *volume, int dx, int dy, int dz, const float *source,
                                      const float *target, const float *img, int B, int det_h,
                                      int det_w, float voxel_shift, float eps, float *out, float *aux,
                                      float record_vmax, int brick_storage, float *brick_ranges,
                                      int ranges_valid, void *launch_ws, const unsigned *pixel_mask,
                                      void *stream) {
    if (pixel_mask && (reinterpret_cast<uintptr_t>(pixel_mask) & 3) != 0)
        return fail(-1, "pixel_mask must be 4-byte aligned");
    return siddon_forward_bricks_impl(volume, dx, dy, dz, source, target, img, B, det_h, det_w, voxel_shift,
                                      eps, out, aux, record_vmax, brick_storage, brick_ranges, ranges_valid,
                                      launch_ws, pixel_mask, stream);
}

long ddrr_brick_workspace_bytes(int dx, int dy, int dz, int brick_storage) {
    if (dx < 1 || dy < 1 || dz < 1) return 0;
    return brick_workspace_bytes(dx, dy, dz, brick_storage);
}

long ddrr_brick_launch_workspace_bytes(int dx, int dy, int dz) {
    if (dx < 1 || dy < 1 || dz < 1) return 0;
    return brick_launch_workspace_bytes(dx, dy, dz);
}

int ddrr_siddon_forward_channels_bricks(const float *volume, const unsigned char *labels, int dx,
                                        int dy, int dz, const float *source, const float *target,
                                        const float *img, int B, int det_h, int det_w, int C,
                                        float voxel_shift, float eps, float *out, void *launch_ws,
                                        void *stream) {
    const int N = det_h * det_w;
    if (int rc = check_common(volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (!labels || !out || C < 1) return fail(-1, "null labels/out or C < 1");
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if ((long)B * C * N >= (1L << 30) || N >= (1 << 22))
        return fail(-1, "B * C * N must stay below 2^30 (and N below 2^22) for one channel launch "
                        "on the bricks: split the pose batch or use ddrr_siddon_forward_channels");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C * N, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    return launch_bricks(BRICK_CHANNELS, volume, dx, dy, dz, source, target, img, nullptr, B, det_h,
                         det_w, voxel_shift, eps, out, nullptr, nullptr, st, launch_ws,
                         "ddrr_siddon_forward_channels_bricks", 0, nullptr, nullptr, 0.f, labels, C);
}

// The channel render's staged words -- value rounded to a 16-bit mantissa | label, labels without a
// channel as the value 0 under label 0 (brick_step.h pack_voxel_label_below) -- for a whole volume at
// once: what ddrr_siddon_forward_channels_bricks_words stages with straight 16-byte copies.
// Launched in front of EVERY render from the words, and self-healing: ONE workgroup compares the
// fingerprint the last repack left (brick_core.h kFingerprintWords voxels: the volume's bits and the
// label bytes) with the live volume and label map and leaves its verdict in state[0]; the pack launch
// behind it ends at once unless the verdict -- an edit the caller's bookkeeping cannot see (PyTorch:
// `volume.data[...] = x`) -- or `force` says otherwise.  (One launch with a last-workgroup ticket
// was measured first: 4096 same-address atomics, 0.26 ms per render.)  state: [0] verdict of the
// launch in flight, [1] repacks so far (both int), then 2 x kFingerprintWords words.
constexpr int kChannelWordsStateWords = 2 + 2 * kFingerprintWords;

__global__ __launch_bounds__(1024) void channel_words_check_kernel(const float *__restrict__ vol,
                                                                   const unsigned char *__restrict__ labels,
                                                                   long n, int *__restrict__ state, int force) {
    unsigned *fp = reinterpret_cast<unsigned *>(state + 2);
    int bad = force;
    unsigned v = 0u, l = 0u;
    if (threadIdx.x < kFingerprintWords) {
        const long at = fingerprint_index(threadIdx.x, n);
        v = __float_as_uint(vol[at]);
        l = (unsigned)labels[at];
        bad |= (v != fp[threadIdx.x]) | (l != fp[kFingerprintWords + threadIdx.x]);
    }
    const bool stale = __syncthreads_or(bad) != 0;
    if (stale && threadIdx.x < kFingerprintWords) {  // (nobody else reads the fingerprint: renewed here)
        fp[threadIdx.x] = v;
        fp[kFingerprintWords + threadIdx.x] = l;
    }
    if (threadIdx.x == 0) {
        state[0] = stale ? 1 : 0;
        if (stale) state[1] += 1;
    }
}

__global__ __launch_bounds__(1024) void channel_words_kernel(const float *__restrict__ vol,
                                                             const unsigned char *__restrict__ labels, long n,
                                                             unsigned n_channels, float *__restrict__ words,
                                                             const int *__restrict__ state) {
    if (state[0] == 0) return;  // (workgroup-uniform: the check's verdict)
    const long stride = (long)gridDim.x * 1024;
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride)
        words[i] = pack_voxel_label_below(vol[i], labels[i], n_channels);
}

long ddrr_channel_words_state_bytes(void) { return (long)kChannelWordsStateWords * 4; }

int ddrr_channel_words(const float *volume, const unsigned char *labels, long n_voxels, int C, float *words,
                       void *state, int force, void *stream) {
    if (!volume || !labels || !words || !state || C < 1 || n_voxels < 0)
        return fail(-1, "null pointer or C < 1");
    if (n_voxels == 0) return 0;
    static_assert(kFingerprintWords <= 1024, "one sample per thread of the check");
    const long blocks = (n_voxels + 4095) / 4096;
    hipLaunchKernelGGL(channel_words_check_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, volume, labels,
                       n_voxels, reinterpret_cast<int *>(state), force ? 1 : 0);
    hipLaunchKernelGGL(channel_words_kernel, dim3((unsigned)(blocks > 512 ? 512 : blocks)), dim3(1024), 0,
                       (hipStream_t)stream, volume, labels, n_voxels, (unsigned)C, words,
                       reinterpret_cast<const int *>(state));
    return finish("ddrr_channel_words");
}

int ddrr_siddon_forward_channels_bricks_words(const float *words, int dx, int dy, int dz, const float *source,
                                              const float *target, const float *img, int B, int det_h,
                                              int det_w, int C, float voxel_shift, float eps, float *out,
                                              void *launch_ws, void *stream) {
    const int N = det_h * det_w;
    if (int rc = check_common(words, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (!out || C < 1) return fail(-1, "null out or C < 1");
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if ((long)B * C * N >= (1L << 30) || N >= (1 << 22))
        return fail(-1, "B * C * N must stay below 2^30 (and N below 2^22) for one channel launch "
                        "on the bricks: split the pose batch or use ddrr_siddon_forward_channels");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C * N, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    return launch_bricks(BRICK_CHANNELS_WORDS, words, dx, dy, dz, source, target, img, nullptr, B, det_h,
                         det_w, voxel_shift, eps, out, nullptr, nullptr, st, launch_ws,
                         "ddrr_siddon_forward_channels_bricks_words", 0, nullptr, nullptr, 0.f, nullptr, C);
}

int ddrr_siddon_backward_channels_bricks(const float *volume, const unsigned char *labels, int dx,
                                         int dy, int dz, const float *source, const float *target,
                                         const float *grad_out, int B, int det_h, int det_w, int C,
                                         float voxel_shift, float eps, float *aux, void *launch_ws,
                                         void *stream) {
    const int N = det_h * det_w;
    if (int rc = check_common(volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (!labels || !grad_out || !aux || C < 1) return fail(-1, "null labels/grad_out/aux or C < 1");
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if ((long)B * C * N >= (1L << 30) || N >= (1 << 22))
        return fail(-1, "B * C * N must stay below 2^30 (and N below 2^22) for one channel launch "
                        "on the bricks: split the pose batch or use ddrr_siddon_backward_channels");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(aux, 0, sizeof(float) * (size_t)rec_blocked_floats((long)B * N), st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    return launch_bricks(BRICK_CHANNELS_AUX, volume, dx, dy, dz, source, target, nullptr, grad_out, B,
                         det_h, det_w, voxel_shift, eps, nullptr, aux, nullptr, st, launch_ws,
                         "ddrr_siddon_backward_channels_bricks", 0, nullptr, nullptr, 0.f, labels, C);
}

int ddrr_trilinear_forward_channels_bricks(const float *volume, const unsigned char *labels,
                                           int dx, int dy, int dz, const float *source,
                                           const float *target, const float *img, int B,
                                           int det_h, int det_w, int C, float voxel_shift,
                                           float eps, int n_points, const float *alphamin,
                                           const float *alphamax, float *out, void *launch_ws,
                                           void *stream) {
    const int N = det_h * det_w;
    if (int rc = check_common(volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (!labels || !out || C < 1) return fail(-1, "null labels/out or C < 1");
    if (!alphamin || !alphamax) return fail(-1, "null alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if ((long)B * C * N >= (1L << 30) || N >= (1 << 22))
        return fail(-1, "B * C * N must stay below 2^30 (and N below 2^22) for one channel launch "
                        "on the bricks: split the pose batch or use ddrr_trilinear_forward_channels");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C * N, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    return launch_bricks(BRICK_TRI_CHANNELS, volume, dx, dy, dz, source, target, img, nullptr, B,
                         det_h, det_w, voxel_shift, eps, out, nullptr, nullptr, st, launch_ws,
                         "ddrr_trilinear_forward_channels_bricks", n_points, alphamin, alphamax, 0.f,
                         labels, C);
}

int ddrr_trilinear_backward_channels_bricks(const float *volume, const unsigned char *labels,
                                            int dx, int dy, int dz, const float *source,
                                            const float *target, const float *grad_out, int B,
                                            int det_h, int det_w, int C, float voxel_shift,
                                            float eps, int n_points, const float *alphamin,
                                            const float *alphamax, float *aux, void *launch_ws,
                                            void *stream) {
    const int N = det_h * det_w;
    if (int rc = check_common(volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (!labels || !grad_out || !aux || C < 1) return fail(-1, "null labels/grad_out/aux or C < 1");
    if (!alphamin || !alphamax) return fail(-1, "null alphamin / alphamax");
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if ((long)B * C * N >= (1L << 30) || N >= (1 << 22))
        return fail(-1, "B * C * N must stay below 2^30 (and N below 2^22) for one channel launch "
                        "on the bricks: split the pose batch or use ddrr_trilinear_backward_channels");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(aux, 0, sizeof(float) * (size_t)B * N * DDRR_TRI_AUX_PLANES, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    return launch_bricks(BRICK_TRI_CHANNELS_AUX, volume, dx, dy, dz, source, target, nullptr,
                         grad_out, B, det_h, det_w, voxel_shift, eps, nullptr, aux, nullptr, st,
                         launch_ws, "ddrr_trilinear_backward_channels_bricks", n_points, alphamin,
                         alphamax, 0.f, labels, C);
}

int ddrr_siddon_backward_channels_volume_bricks(const unsigned char *labels, int dx, int dy, int dz,
                                                const float *source, const float *target,
                                                const float *img, const float *grad_out, int B,
                                                int det_h, int det_w, int C, float voxel_shift,
                                                float eps, float *g_volume, void *launch_ws,
                                                void *stream) {
    const int N = det_h * det_w;
    if (!g_volume || !grad_out || !labels || C < 1)
        return fail(-1, "null labels / grad_out / g_volume or C < 1");
    if (int rc = check_common(g_volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if ((long)B * C * N >= (1L << 30) || N >= (1 << 22))
        return fail(-1, "B * C * N must stay below 2^30 (and N below 2^22) for one channel launch "
                        "on the bricks: split the pose batch or use ddrr_siddon_backward_channels");
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {  // nothing contributes: the gradient is zero
        hipError_t e = hipMemsetAsync(g_volume, 0, sizeof(float) * (size_t)dx * dy * dz, st);
        return e == hipSuccess ? 0 : fail_hip(e, "hipMemsetAsync");
    }
    return launch_bricks(BRICK_CHANNELS_VOLGRAD, nullptr, dx, dy, dz, source, target, img, grad_out,
                         B, det_h, det_w, voxel_shift, eps, nullptr, nullptr, g_volume, st, launch_ws,
                         "ddrr_siddon_backward_channels_volume_bricks", 0, nullptr, nullptr, 0.f,
                         labels, C);
}

int ddrr_siddon_backward_volume_bricks(int dx, int dy, int dz, const float *source,
                                       const float *target, const float *img,
                                       const float *grad_out, int B, int det_h, int det_w,
                                       float voxel_shift, float eps, float *g_volume,
                                       void *launch_ws, void *stream) {
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
                         det_h, det_w, voxel_shift, eps, nullptr, nullptr, g_volume, st, launch_ws,
                         "ddrr_siddon_backward_volume_bricks");
}

int ddrr_trilinear_forward_bricks(const float *volume, int dx, int dy, int dz,
                                  const float *source, const float *target, const float *img,
                                  int B, int det_h, int det_w, float voxel_shift, float eps,
                                  int n_points, const float *alphamin, const float *alphamax,
                                  float *out, float *aux, void *launch_ws, void *stream) {
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
                             det_h, det_w, voxel_shift, eps, out, nullptr, nullptr, st, launch_ws,
                             "ddrr_trilinear_forward_bricks", n_points, alphamin, alphamax);
    if (int rc = launch_bricks(BRICK_TRI_FWD_AUX, volume, dx, dy, dz, source, target, img, nullptr,
                               B, det_h, det_w, voxel_shift, eps, out, aux, nullptr, st, launch_ws,
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

int ddrr_trilinear_backward_channels_volume_bricks(const unsigned char *labels, int dx, int dy,
                                                   int dz, const float *source, const float *target,
                                                   const float *img, const float *grad_out, int B,
                                                   int det_h, int det_w, int C, float voxel_shift,
                                                   float eps, int n_points, const float *alphamin,
                                                   const float *alphamax, float *g_volume,
                                                   void *launch_ws, void *stream) {
    const int N = det_h * det_w;
    if (!g_volume || !grad_out || !labels || !alphamin || !alphamax || C < 1)
        return fail(-1, "null labels / grad_out / g_volume / alphamin / alphamax or C < 1");
    if (int rc = check_common(g_volume, dx, dy, dz, source, 1, target, B, N)) return rc;
    if (n_points < 2) return fail(-1, "n_points must be >= 2");
    if (det_h < 2 || det_w < 2) return fail(-1, "the brick path needs a detector of at least 2x2");
    if ((long)B * C * N >= (1L << 30) || N >= (1 << 22))
        return fail(-1, "B * C * N must stay below 2^30 (and N below 2^22) for one channel launch "
                        "on the bricks: split the pose batch or use ddrr_trilinear_backward_channels");
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        hipError_t e = hipMemsetAsync(g_volume, 0, sizeof(float) * (size_t)dx * dy * dz, st);
        return e == hipSuccess ? 0 : fail_hip(e, "hipMemsetAsync");
    }
    return launch_bricks(BRICK_TRI_CHANNELS_VOLGRAD, nullptr, dx, dy, dz, source, target, img,
                         grad_out, B, det_h, det_w, voxel_shift, eps, nullptr, nullptr, g_volume, st,
                         launch_ws, "ddrr_trilinear_backward_channels_volume_bricks", n_points, alphamin,
                         alphamax, 0.f, labels, C);
}

int ddrr_trilinear_backward_volume_bricks(int dx, int dy, int dz, const float *source,
                                          const float *target, const float *img,
                                          const float *grad_out, int B, int det_h, int det_w,
                                          float voxel_shift, float eps, int n_points,
                                          const float *alphamin, const float *alphamax,
                                          float *g_volume, void *launch_ws, void *stream) {
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
                         det_h, det_w, voxel_shift, eps, nullptr, nullptr, g_volume, st, launch_ws,
                         "ddrr_trilinear_backward_volume_bricks", n_points, alphamin, alphamax);
}

}  // extern "C"
