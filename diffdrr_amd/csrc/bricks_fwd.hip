// bricks_fwd.hip -- the Siddon forward / forward + record kernel of the DRR path on
// volume-stationary bricks, with the brick's shape, storage and the workgroup size as
// compile-time configuration (bricks.hip keeps every other mode on 32^3 fp32 bricks).
//
// Same orchestration as siddon_brick_kernel (bricks.hip): persistent workgroups pull bricks
// from a global counter; per brick the voxels are staged in LDS, every pose's candidate pixels
// are tested arithmetically (phase A), the hits are compacted into per-wave length-class
// queues and walked from LDS 64 at a time (phase B); partial integrals (and the backward
// record) are combined with fp32 atomics.  What is configurable:
//   * Q16: the brick is staged as 16-bit block-quantised voxels (brick_step.h: one (vmin, step)
//     pair per brick, from a pass over the volume that the caller caches; the walk reads them
//     as denormal floats at no extra instruction).  A brick of TWICE the volume then fits the
//     LDS of a CU: 32 x 32 x 64 voxels in 130 KiB, 17 % fewer (ray, brick) pairs, i.e. less of
//     the per-hit work (exact clip, ray loads, queueing, record atomics) per voxel visited:
//     forward 1.30 -> 1.21 ms, forward + record 1.82 -> 1.70 ms at 512^3 / 32 poses;
//   * BX x BY x BZ, THREADS: brick extent and workgroup size.  Measured and not adopted
//     (profiles/r03, tools builds keep them as variants): two 512-thread workgroups per CU on
//     32^3 16-bit bricks or on 32 x 32 x 16 fp32 half bricks (their barrier waits halve, their
//     staging time doubles: no gain), anisotropic fp32 bricks 16 x 64 x 32 ... (3-8 % slower on
//     mixed poses), workgroup-shared rings of 12 length classes (see below).
// Reference: diffdrr/renderers.py:34-76, 94-113 (Siddon.forward, mask=None, sum, nearest).
#include "runtime.h"

#include <mutex>
#include "siddon_core.h"
#include "brick_core.h"
#include "brick_walk.h"
#include "brick_step.h"
#include "record_pack.h"
#include "record_layout.h"
#include "brick_shared.h"

using namespace ddrr;
using namespace ddrr_rt;
using namespace ddrr_brick;

namespace {

template <int BX_, int BY_, int BZ_, int THREADS_, bool Q16_, int ROWPAD_ = -1>
struct FwdCfg {
    static constexpr int BX = BX_, BY = BY_, BZ = BZ_, THREADS = THREADS_, WAVES = THREADS_ / 64;
    static constexpr bool Q16 = Q16_;
    static constexpr int ES = Q16 ? 2 : 4;  // bytes per staged voxel
    // rows and planes padded by one element so that x-, y- and z-neighbours fall in different
    // banks (ROWPAD_ = 0: no row padding, where the LDS budget of a half CU has no room for it)
    static constexpr int ROWPAD = ROWPAD_ < 0 ? ES : ROWPAD_;
    static constexpr int SY = BZ * ES + ROWPAD;  // byte strides
    static constexpr int SX = BY * SY + ES;
    static constexpr int BRICK_BYTES = (BX * SX + 15) / 16 * 16;
    static constexpr int WGS_PER_CU = 1024 / THREADS;
    static constexpr int LDS_BUDGET = 160 * 1024 / WGS_PER_CU;
    static constexpr int QUEUE_BYTES = WAVES * kBuckets * kQueueCap * 4;
    // A 16-bit brick whose values the block quantisation would not keep (brick_step.h
    // q16_usable: range large against the smallest 4^3-block level, non-finite values) is
    // rendered from the volume's own fp32 values instead: its two halves along z, one after
    // the other, as fp32 bricks BX x BY x BZ/2 (rows and planes padded by one float) in the
    // same LDS.  MIXED: the budget has the room (every product configuration; not two
    // workgroups per CU).
    static constexpr int HZ = BZ / 2;
    static constexpr int FSY = HZ * 4 + 4, FSX = BY * FSY + 4;  // byte strides of a fallback half
    static constexpr int FALLBACK_BYTES = (BX * FSX + 15) / 16 * 16;
    static constexpr bool MIXED =
        Q16 && HZ % 4 == 0 && (BX * BY * (HZ / 4)) % THREADS == 0 &&
        FALLBACK_BYTES + QUEUE_BYTES + 16 + 8 * 20 * 4 <= LDS_BUDGET;
    using Half = FwdCfg<BX_, BY_, BZ_ / 2, THREADS_, false>;  // (used where MIXED only)
    static constexpr int LDS_BRICK = MIXED && FALLBACK_BYTES > BRICK_BYTES ? FALLBACK_BYTES : BRICK_BYTES;
    // poses per row-table chunk: what the LDS left by the brick and the queues holds
    static constexpr int ROW_ROOM = (LDS_BUDGET - LDS_BRICK - QUEUE_BYTES - 48) / (20 * 4);
    static constexpr int CHUNK = ROW_ROOM >= 32 ? 32 : ROW_ROOM;
    static constexpr int LDS_MIN = LDS_BRICK + QUEUE_BYTES + CHUNK * 20 * 4 + 16 + 32;  // (+ counters, look-ahead slots)
    // what the waves have left in their queues at the end of a brick is pooled (one count per
    // wave and class) where the budget has the room for the counts
    static constexpr int POOL_BYTES = WAVES * kBuckets * 4;
    static constexpr bool POOL = LDS_MIN + POOL_BYTES <= LDS_BUDGET && WAVES * kBuckets <= 64;
    static constexpr int LDS = LDS_MIN + (POOL ? POOL_BYTES : 0);
    static constexpr int MAXSTEPS = BX + BY + BZ + 4;
    static_assert(CHUNK >= 8, "no room for the row table");
    static_assert(BZ % 4 == 0 && (BX * BY * (BZ / 4)) % THREADS == 0, "staging");
};

// A row of the per-(pose, brick) table phase A reads, without the fields only the other modes
// use (brick_walk.h BrickRow): 20 words.
struct FwdRow {
    float D0[3], ei[3], ej[3], P0[3], P1[3];
    float inv_w;
    int i0, j0, w, count;
};
static_assert(sizeof(FwdRow) == 80, "FwdRow");

__device__ __forceinline__ FwdRow fwd_row(const BrickRow &r) {
    FwdRow f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        f.D0[a] = r.D0[a];
        f.ei[a] = r.ei[a];
        f.ej[a] = r.ej[a];
        f.P0[a] = r.P0[a];
        f.P1[a] = r.P1[a];
    }
    f.inv_w = r.inv_w;
    f.i0 = r.i0;
    f.j0 = r.j0;
    f.w = r.w;
    f.count = r.count;
    return f;
}

__device__ __forceinline__ BrickRow brick_row_of(const FwdRow &f) {
    BrickRow r;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        r.D0[a] = f.D0[a];
        r.ei[a] = f.ei[a];
        r.ej[a] = f.ej[a];
        r.P0[a] = f.P0[a];
        r.P1[a] = f.P1[a];
    }
    r.inv_w = f.inv_w;
    r.nscale = 0.f;
    r.i0 = f.i0;
    r.j0 = f.j0;
    r.w = f.w;
    r.count = f.count;
    r.perm_k = 1;
    return r;
}

// Fetch of a 16-bit voxel by absolute LDS byte address: q in the low half of the register =
// the denormal float q 2^-149 (brick_step.h).
struct LdsAbsFetch16 {
    __device__ __forceinline__ float operator()(unsigned addr) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned q =
            *(const __attribute__((address_space(3))) unsigned short *)(unsigned long long)addr;
        return __uint_as_float(q);
#else
        (void)addr;
        return 0.f;
#endif
    }
};

template <class C>
__device__ __forceinline__ Box cfg_brick_box(const Dims D, int nby, int nbz, int id) {
    const int bz = id % nbz, by = (id / nbz) % nby, bx = id / (nbz * nby);
    Box b;
    b.lo[0] = bx * C::BX;
    b.lo[1] = by * C::BY;
    b.lo[2] = bz * C::BZ;
    b.hi[0] = b.lo[0] + C::BX < D.x ? b.lo[0] + C::BX : D.x;
    b.hi[1] = b.lo[1] + C::BY < D.y ? b.lo[1] + C::BY : D.y;
    b.hi[2] = b.lo[2] + C::BZ < D.z ? b.lo[2] + C::BZ : D.z;
    return b;
}

// Phase B for one batch entry (all lanes call; `active`: the lane holds an entry).
template <bool AUX, class C>
__device__ __forceinline__ void fwd_item(const BrickArgs &p, unsigned lds_base,
                                         const StepGeom &SG, const Q16Range &range, bool active,
                                         unsigned b, unsigned pix, float *__restrict__ out,
                                         float *__restrict__ aux, BrickProf &prof,
                                         const FwdRow *rows = nullptr, int b0 = 0,
                                         bool f32_brick = !C::Q16) {
    // (f32_brick: wave-uniform -- the staged brick holds the volume's own fp32 values)
    const bool q16 = C::Q16 && !f32_brick;
    const unsigned r = b * (unsigned)(p.det_h * p.det_w) + pix;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    bool ok = false;
    if (active) {
        const float *sp = p.source + b * 3u, *tp = p.target + r * 3u;
        float s[3], t[3], L = 1.f;
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
        if ((p.dbg & 256) && rows) {
            // (timing experiment: NO ray loads -- the ray from the pose's affine detector model in
            // the LDS row table, target = s + D0 + i ei + j ej; results differ in the last bits)
            const FwdRow &w = rows[b - b0];
            const float fi = floorf(((float)pix + 0.5f) / (float)p.det_w);
            const float fj = (float)pix - fi * (float)p.det_w;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                s[a] = (float)(a == 0 ? SG.lof[0] : (a == 1 ? SG.lof[1] : SG.lof[2])) - 0.01f - p.shift -
                       w.P0[a];  // P0 = (lo - margin) - shift - s
                t[a] = s[a] + (fmaf(fj, w.ej[a], fmaf(fi, w.ei[a], w.D0[a])) - p.eps);
            }
            L = sqrtf((t[0] - s[0]) * (t[0] - s[0]) + (t[1] - s[1]) * (t[1] - s[1]) +
                      (t[2] - s[2]) * (t[2] - s[2]));
        } else
#endif
        {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                s[a] = sp[a];
                t[a] = tp[a];
            }
            L = (!AUX && p.img) ? p.img[r] : 1.f;
        }
        DDRR_PROF_WAIT_VMEM();
        DDRR_PROF(PROF_LOADS);
        StepEntry E = step_enter(SG, s, t, p.shift, p.eps, lds_base);
        if (q16) q16_scale_entry(E);
        DDRR_PROF(PROF_SETUP);
        int steps = 0;
        float ex[2] = {0.f, 0.f};
        if (E.hit) {
            if (!AUX) {
                // forward only: the accumulating walk (brick_step.h step_walk_fwd)
                float a_end;
                if (q16) {
                    steps = step_walk_fwd<C::MAXSTEPS>(LdsAbsFetch16{}, SG, E, v[0], a_end);
                    v[0] = q16_finish_fwd(range, E, v[0], a_end);
                } else {
                    steps = step_walk_fwd<C::MAXSTEPS>(LdsAbsFetch{}, SG, E, v[0], a_end);
                }
            } else if (q16) {
                steps = step_walk<AUX, C::MAXSTEPS>(LdsAbsFetch16{}, SG, E, v[0], v + 1, ex);
                q16_finish<AUX>(range, E, ex, v[0], v + 1);
            } else {
                steps = step_walk<AUX, C::MAXSTEPS>(LdsAbsFetch{}, SG, E, v[0], v + 1);
            }
        }
        DDRR_PROF(PROF_WALK);
        DDRR_PROF_COUNT(PROF_N_STEPS, (unsigned long long)__builtin_amdgcn_readfirstlane(steps));
        (void)steps;
        ok = E.hit;  // (phase A's margin lets a few non-crossing rays through)
        if (!AUX && ok) unsafeAtomicAdd(out + r, L * v[0]);
    }
    if (AUX) {
        // with the record, out = L * (plane I) is formed afterwards (siddon_out_from_record)
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
        } else {
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
            // (timing experiment: the record computed but NOT delivered -- the bound on what any other
            // delivery, e.g. through LDS, could gain: profiles/r05/record_structural_candidates.txt)
            if (p.dbg & 131072) ok = ok && v[0] == 12345.678f;
#endif
            deliver_record_blocked(aux, ok, r, v);
        }
    }
    DDRR_PROF(PROF_DELIVER);
}

// Stage brick `brick_id` (voxels `box`) in LDS: a thread owns NQ quads of 4 voxels along z, PER
// of them in flight at a time (all of a round's loads are issued before its first LDS store).
// 16-bit bricks: `range` / `brick_empty` from the ranges brick_range_kernel left; fp32 bricks:
// counter[2] is raised if any staged voxel is non-zero (read after the next barrier).
// Any D.z (the reference's example CT has 133 slices): a quad is ONE 16-byte load from a
// dword-aligned address (brick_shared.h quad_load); what it reads beyond its row is masked per
// voxel.
template <class C>
__device__ __forceinline__ void fwd_stage_brick(const BrickArgs &p, unsigned char *brick,
                                                const Box &box, int brick_id, int tid,
                                                Q16Range &range, bool &brick_empty, int *counter) {
    constexpr int QZ = C::BZ / 4, NQ = C::BX * C::BY * QZ / C::THREADS;
    constexpr int PER = NQ > 8 ? 8 : NQ;
    static_assert(NQ % PER == 0, "staging rounds");
    constexpr int ROWS_PER_PASS = C::THREADS / QZ;
    const int qz4 = (tid % QZ) * 4, row0 = tid / QZ;
    const int z = box.lo[2] + qz4;
    if (C::Q16) {
        const float lo = p.ranges[2 * brick_id], hi = p.ranges[2 * brick_id + 1];
        range = q16_range(lo, hi);
        brick_empty = lo == 0.f && hi == 0.f;
    }
    unsigned nz = 0u;
#pragma unroll
    for (int h0 = 0; h0 < NQ; h0 += PER) {
        quad_u32x4 q[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int row = row0 + (h0 + i) * ROWS_PER_PASS;
            const int lx = row / C::BY, ly = row - lx * C::BY;
            const int x = box.lo[0] + lx, y = box.lo[1] + ly;
            // clamped (always readable) addresses; what lies outside is zeroed below
            const int xc = x < p.D.x ? x : p.D.x - 1, yc = y < p.D.y ? y : p.D.y - 1;
            unsigned none = 0u;
            quad_load<false>(p.vol, nullptr, p.D, ((long)xc * p.D.y + yc) * p.D.z + z, q[i], none);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int row = row0 + (h0 + i) * ROWS_PER_PASS;
            const int lx = row / C::BY, ly = row - lx * C::BY;
            const int x = box.lo[0] + lx, y = box.lo[1] + ly;
            const bool in = x < box.hi[0] && y < box.hi[1];
            quad_u32x4 w = q[i];
            unsigned none = 0u;
            quad_fix(p.D, x, y, z, w, none);
            if (!(in && z < box.hi[2])) w.x = 0u;
            if (!(in && z + 1 < box.hi[2])) w.y = 0u;
            if (!(in && z + 2 < box.hi[2])) w.z = 0u;
            if (!(in && z + 3 < box.hi[2])) w.w = 0u;
            if (C::Q16) {
                unsigned short *d =
                    reinterpret_cast<unsigned short *>(brick + lx * C::SX + ly * C::SY + qz4 * 2);
                d[0] = (unsigned short)q16_encode(bits_as_float(w.x), range);
                d[1] = (unsigned short)q16_encode(bits_as_float(w.y), range);
                d[2] = (unsigned short)q16_encode(bits_as_float(w.z), range);
                d[3] = (unsigned short)q16_encode(bits_as_float(w.w), range);
            } else {
                unsigned *df = reinterpret_cast<unsigned *>(brick + lx * C::SX + ly * C::SY + qz4 * 4);
                df[0] = w.x;
                df[1] = w.y;
                df[2] = w.z;
                df[3] = w.w;
                nz |= (w.x | w.y) | (w.z | w.w);
            }
        }
    }
    // Empty space: a brick of zeros adds nothing to any integral or record
    if (!C::Q16 && (nz & 0x7fffffffu) != 0u) counter[2] = 1;  // (cleared with the claim)
}

// A volume that is rendered again and again (a registration, a pose sweep) can keep its 16-bit
// bricks in HBM as they lie in LDS -- padding and all, brick after brick (brick_pack_kernel, into
// the caller's workspace): staging is then a straight 16-byte copy of half the bytes, without the
// conversion and without the 2-byte LDS stores of the padded rows.
template <class C>
__device__ __forceinline__ void fwd_stage_packed(const BrickArgs &p, unsigned char *brick,
                                                 int brick_id, int tid, Q16Range &range,
                                                 bool &brick_empty) {
    constexpr int NV = C::BRICK_BYTES / 16, G = 4, FULL = NV / (G * C::THREADS);
    const float lo = p.ranges[2 * brick_id], hi = p.ranges[2 * brick_id + 1];
    range = q16_range(lo, hi);
    brick_empty = lo == 0.f && hi == 0.f;
    if (brick_empty) return;  // (nothing will read the brick)
    const uint4 *src = reinterpret_cast<const uint4 *>(p.packed + (size_t)brick_id * C::BRICK_BYTES);
    uint4 *dst = reinterpret_cast<uint4 *>(brick);
    // groups of G loads in flight per thread, then what is left
#pragma unroll
    for (int g = 0; g < FULL; ++g) {
        uint4 v[G];
#pragma unroll
        for (int i = 0; i < G; ++i) v[i] = src[(g * G + i) * C::THREADS + tid];
#pragma unroll
        for (int i = 0; i < G; ++i) dst[(g * G + i) * C::THREADS + tid] = v[i];
    }
    for (int k = FULL * G * C::THREADS + tid; k < NV; k += C::THREADS) dst[k] = src[k];
}

// The packed image of brick `brick_id` into registers (issued by a wave that has nothing left to
// walk in the brick in hand: the loads fly while the others finish) and from there into LDS.
// (Native vector registers by name: HIP's uint4 is copied with memcpy, which keeps an object that
// lives across the brick loop in scratch.)
typedef quad_u32x4 pf_u32x4;
template <class C>
struct PackedPrefetch {
    static constexpr int NV = C::BRICK_BYTES / 16, N = (NV + C::THREADS - 1) / C::THREADS;
    static constexpr bool FITS = N <= 9;
    pf_u32x4 v0, v1, v2, v3, v4, v5, v6, v7, v8;
    template <int I>
    __device__ __forceinline__ void load1(pf_u32x4 &v, const pf_u32x4 *src, int tid) {
        if constexpr (I < N) {
            const int k = I * C::THREADS + tid;
            v = src[(I + 1) * C::THREADS <= NV || k < NV ? k : NV - 1];  // (clamped: unconditional)
        }
    }
    template <int I>
    __device__ __forceinline__ void store1(const pf_u32x4 &v, pf_u32x4 *dst, int tid) const {
        if constexpr (I < N) {
            const int k = I * C::THREADS + tid;
            if ((I + 1) * C::THREADS <= NV || k < NV) dst[k] = v;
        }
    }
    __device__ __forceinline__ void load(const BrickArgs &p, int brick_id, int tid) {
        const pf_u32x4 *src = reinterpret_cast<const pf_u32x4 *>(p.packed + (size_t)brick_id * C::BRICK_BYTES);
        load1<0>(v0, src, tid), load1<1>(v1, src, tid), load1<2>(v2, src, tid);
        load1<3>(v3, src, tid), load1<4>(v4, src, tid), load1<5>(v5, src, tid);
        load1<6>(v6, src, tid), load1<7>(v7, src, tid), load1<8>(v8, src, tid);
    }
    template <int I>
    __device__ __forceinline__ pf_u32x4 &reg() {
        if constexpr (I == 0) return v0;
        else if constexpr (I == 1) return v1;
        else if constexpr (I == 2) return v2;
        else if constexpr (I == 3) return v3;
        else if constexpr (I == 4) return v4;
        else if constexpr (I == 5) return v5;
        else if constexpr (I == 6) return v6;
        else if constexpr (I == 7) return v7;
        else return v8;
    }
    // A brick on the fp32 path: the half `box` (H: its configuration, C::Half) from the volume's own
    // values, one round of fwd_stage_brick<H>'s loads held in the same registers, and its stores.
    template <class H, int I>
    __device__ __forceinline__ void half1(const BrickArgs &p, unsigned char *brick, const Box &box,
                                          int tid, bool load) {
        constexpr int QZ = H::BZ / 4, NQ = H::BX * H::BY * QZ / H::THREADS;
        constexpr int ROWS_PER_PASS = H::THREADS / QZ;
        static_assert(NQ <= 9, "a half's quads per thread must fit the prefetch registers");
        if constexpr (I < NQ) {
            const int qz4 = (tid % QZ) * 4, row = tid / QZ + I * ROWS_PER_PASS;
            const int lx = row / H::BY, ly = row - lx * H::BY;
            const int x = box.lo[0] + lx, y = box.lo[1] + ly, z = box.lo[2] + qz4;
            unsigned none = 0u;
            if (load) {
                // clamped (always readable) addresses; what lies outside is zeroed by the store
                const int xc = x < p.D.x ? x : p.D.x - 1, yc = y < p.D.y ? y : p.D.y - 1;
                quad_load<false>(p.vol, nullptr, p.D, ((long)xc * p.D.y + yc) * p.D.z + z, reg<I>(), none);
            } else {
                pf_u32x4 q = reg<I>();
                quad_fix(p.D, x, y, z, q, none);
                const bool in = x < box.hi[0] && y < box.hi[1];
                if (!(in && z < box.hi[2])) q.x = 0u;
                if (!(in && z + 1 < box.hi[2])) q.y = 0u;
                if (!(in && z + 2 < box.hi[2])) q.z = 0u;
                if (!(in && z + 3 < box.hi[2])) q.w = 0u;
                unsigned *d = reinterpret_cast<unsigned *>(brick + lx * H::SX + ly * H::SY + qz4 * 4);
                d[0] = q.x, d[1] = q.y, d[2] = q.z, d[3] = q.w;
            }
        }
    }
    template <class H>
    __device__ __forceinline__ void half(const BrickArgs &p, unsigned char *brick, const Box &box, int tid,
                                         bool load) {
        half1<H, 0>(p, brick, box, tid, load), half1<H, 1>(p, brick, box, tid, load);
        half1<H, 2>(p, brick, box, tid, load), half1<H, 3>(p, brick, box, tid, load);
        half1<H, 4>(p, brick, box, tid, load), half1<H, 5>(p, brick, box, tid, load);
        half1<H, 6>(p, brick, box, tid, load), half1<H, 7>(p, brick, box, tid, load);
        half1<H, 8>(p, brick, box, tid, load);
    }
    // (nothing held: without this on the paths that do not load, the old contents would stay live
    // through the whole next iteration -- 36 registers the walk does not have)
    __device__ __forceinline__ void clear() {
        v0 = v1 = v2 = v3 = v4 = v5 = v6 = v7 = v8 = pf_u32x4{0u, 0u, 0u, 0u};
    }
    __device__ __forceinline__ void store(unsigned char *brick, int tid) const {
        pf_u32x4 *dst = reinterpret_cast<pf_u32x4 *>(brick);
        store1<0>(v0, dst, tid), store1<1>(v1, dst, tid), store1<2>(v2, dst, tid);
        store1<3>(v3, dst, tid), store1<4>(v4, dst, tid), store1<5>(v5, dst, tid);
        store1<6>(v6, dst, tid), store1<7>(v7, dst, tid), store1<8>(v8, dst, tid);
    }
};

// One workgroup per brick: stage it exactly as the render kernel would and store the LDS image.
template <class C>
__global__ __launch_bounds__(C::THREADS) void brick_pack_kernel(BrickArgs p, int nby, int nbz,
                                                                unsigned char *__restrict__ packed) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ int counter[4];
    const int tid = threadIdx.x, brick_id = blockIdx.x;
    if (p.fallback && p.fallback[brick_id]) return;  // rendered from the volume itself
    constexpr int NV = C::BRICK_BYTES / 16;
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    for (int k = tid; k < NV; k += C::THREADS) lds[k] = make_uint4(0u, 0u, 0u, 0u);  // (the padding)
    __syncthreads();
    const Box box = cfg_brick_box<C>(p.D, nby, nbz, brick_id);
    Q16Range range = {0.f, 0.f, 0.f};
    bool empty = false;
    fwd_stage_brick<C>(p, smem_raw, box, brick_id, tid, range, empty, counter);
    __syncthreads();
    uint4 *dst = reinterpret_cast<uint4 *>(packed + (size_t)brick_id * C::BRICK_BYTES);
    for (int k = tid; k < NV; k += C::THREADS) dst[k] = lds[k];
}

// (profiling builds: stamps of a brick's stages behind the per-brick durations, 10 ns ticks from
// the brick's start: [n_bricks + 16 brick + k]; k = 0 staged, 1 at the pool's barrier, 2 behind
// it, 3 wave 0 out of work, 4 / 7 wave 0 / the last wave at the staging barrier, 5 last wave out
// of work, 6 last walk done, 8 / 10 wave 0's / the last wave's share of the image stored, 9 wave 0:
// row table written.  A stamp is a scalar-memory round trip and a global store: with
// -DDDRR_TRACE_ONLY (no phase marks) a one-pose launch takes 117 instead of 108 us.)
#if defined(DDRR_BRICK_PROFILE)
#define DDRR_TRACE(k, how)                                                                      \
    if (p.brick_times && lane == 0) {                                                           \
        const unsigned dt_ = (unsigned)__builtin_amdgcn_s_memrealtime() - (unsigned)ahead[6];   \
        unsigned *slot_ = p.brick_times + n_bricks + 16 * brick_id + (k);                        \
        if (how) atomicMax(slot_, dt_);                                                         \
        else if (wave == 0) *slot_ = dt_;                                                       \
    }
#else
// (a statement of its own: `if (c) DDRR_TRACE(..)` in front of a barrier must not swallow it --
// round 5 lost the staging barrier of every pose chunk but the first that way for an hour)
#define DDRR_TRACE(k, how) {}
#endif

// PRE (launches of a few poses on the 16-bit storages): the workgroup's first claim is made in front
// of the loop, see the fingerprint comparison below.
// SUB: the launch renders a subsample of the detector (p.pix_mask; an instantiation of its own: as a
// uniform branch per unit the test cost the one-pose launches 1 - 2 us of 111).
template <bool AUX, class C, bool PRE = false, bool SUB = false>
__global__ __launch_bounds__(C::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void
siddon_fwd_brick_kernel(BrickArgs p, float *__restrict__ out, float *__restrict__ aux) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned char *brick = smem_raw;
    unsigned *queue = reinterpret_cast<unsigned *>(smem_raw + C::LDS_BRICK);
    FwdRow *rows = reinterpret_cast<FwdRow *>(queue + C::WAVES * kBuckets * kQueueCap);
    // [0] unit, [1] brick, [2] non-zero, [3] pooled batch; then the pool's counts
    int *counter = reinterpret_cast<int *>(rows + C::CHUNK);
    // the look-ahead: [0] the item after the one in hand (-1: none), [1] its brick | kLaStage,
    // [2], [3] the brick's (min, max); [4] the item after that (held until it is looked at)
    volatile int *ahead = counter + 4;
    int *pool = counter + 12;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nbx = (p.D.x + C::BX - 1) / C::BX, nby = (p.D.y + C::BY - 1) / C::BY;
    const int nbz = (p.D.z + C::BZ - 1) / C::BZ;
    const int n_bricks = nbx * nby * nbz;
    const int N = p.det_h * p.det_w;
    unsigned *myq = queue + wave * kBuckets * kQueueCap;  // wave-private (see bricks.hip)
    const unsigned pix_mask = (1u << p.pix_bits) - 1u;
    const unsigned lds_base = LdsAbsFetch::base_of(reinterpret_cast<const float *>(brick));
    const bool GROUPED = AUX && p.rec_q == 0.f && !(p.dbg & 8);
    const bool POOLED = C::POOL && !(p.dbg & 2048);

    BrickProf prof;
#if defined(DDRR_BRICK_PROFILE)
    prof.start();
#endif
    // the work item in hand (workgroup-uniform): a brick, or -- a 16-bit brick rendered from fp32
    // values -- one of its `n_sub` halves along z, taken one after the other
    int brick_id = 0, sub = 0, n_sub = 1, pose_lo = 0, pose_hi = p.B;
    bool f32_brick = !C::Q16;
    // Looking ahead (packed 16-bit bricks).  A brick's fixed latencies -- the claim's atomic, the
    // table lookups behind it, the brick's own bytes -- are what a launch of a few poses spends
    // its time on (a registration step: one pose).  So while a brick is in hand, lane 0 of wave 1
    // claims the item after the next one and looks up the next one's brick (order, fallback flag,
    // range) -- nothing in one iteration depends on another request of the same iteration -- and a
    // wave that has nothing left to walk requests its share of the next brick's image into
    // registers: the loads fly while the other waves finish, and the next iteration starts with
    // LDS stores (a brick on the fp32 path: one half's values at a time).  Not in the last two rounds
    // of the launch (an item held ahead there could wait behind its workgroup while others idle),
    // and not for more poses than one chunk: bricks that take long have nothing to hide, and the
    // items held ahead cost the launch's end its balance (64 poses: +1.5 %, 512: +5 %;
    // profiles/r04/look_ahead.txt).
    constexpr int kLookRounds = 3;
    constexpr int kLaStage = 1 << 30;  // the next brick is on the fp32 path (its halves are requested one by one)
    const bool LOOK = C::Q16 && PackedPrefetch<C>::FITS && p.packed != nullptr && p.B <= C::CHUNK && !(p.dbg & 4096)
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
                      && p.split_t == 0
#endif
        ;
    const int n_wg = (int)gridDim.x;
    PackedPrefetch<C> pf;
    pf.clear();
    // (workgroup-uniform) la_item / la_brick / la_lo / la_hi: the next item as published by the
    // iteration in hand, -1: none; la_loaded: its image is in `pf` (or it is empty)
    int la_item = -1, la_brick = 0, la_after = -1;
    float la_lo = 0.f, la_hi = 0.f;
    bool la_f32 = false;  // the next item's brick is on the fp32 path
    // The cached 16-bit workspace against the live volume (brick_core.h kFingerprintWords): every
    // workgroup compares the build pass's fingerprint for itself -- one voxel per thread, one
    // barrier, once per launch -- and a launch that finds the volume changed renders EVERY brick from
    // the volume's own fp32 values (the fallback path) instead of the stale bricks: an edit that
    // bypassed the caller's version counter costs speed, not correctness.  Waited for on its own in
    // front of the loop the comparison costs a launch ~3 us: 0.2 % of a 32-pose launch, 2.7 % of a
    // one-pose launch (110 us).  PRE, the instantiation for launches of at most 8 poses, hides the
    // round trip behind the workgroup's FIRST claim, made here by thread 0 with the lookups behind
    // it (order, fallback flag, range: three dependent round trips) and handed to the loop as an item
    // already known, like the look-ahead's: one pose 0.113 -> 0.110 ms.  (As the only form it cost the
    // 32-pose launches 1 % -- 1.446 against 1.432 ms with the record, same box, the loop's register
    // allocation -- hence two instantiations: profiles/r06/fingerprint_placement_ab.txt.)
    bool stale = false;
    if constexpr (C::MIXED) {
        if (p.fingerprint != nullptr && p.ranges_valid) {
            static_assert(kFingerprintWords <= C::THREADS, "one sample per thread");
            int bad = 0;
            if (tid < kFingerprintWords)
                bad = __float_as_uint(p.vol[fingerprint_index(tid, (long)p.D.x * p.D.y * p.D.z)]) != p.fingerprint[tid];
            if constexpr (PRE) {
                if (tid == 0) {
                    const int it = atomicAdd(p.work, 1);
                    int b = 0, fb = 0;
                    float lo = 0.f, hi = 0.f;
                    if (it < n_bricks) {
                        b = p.order ? p.order[it] : it;
                        fb = p.fallback ? p.fallback[b] : 0;
                        lo = p.ranges[2 * b];
                        hi = p.ranges[2 * b + 1];
                    }
                    ahead[0] = it;
                    ahead[1] = b | (fb ? kLaStage : 0);
                    ahead[2] = __float_as_int(lo);
                    ahead[3] = __float_as_int(hi);
                }
            }
            stale = __syncthreads_or(bad) != 0;
            if constexpr (PRE) {
                la_item = uni(ahead[0]);
                const int b = uni(ahead[1]);
                la_brick = b & ~kLaStage;
                la_f32 = stale || (b & kLaStage) != 0;
                la_lo = __int_as_float(uni(ahead[2]));
                la_hi = __int_as_float(uni(ahead[3]));
            }
            if (stale && blockIdx.x == 0 && tid == 0) atomicAdd(p.ws_header + 2, 1);
        }
    }
    // what `pf` holds for the unit (brick, or half of a brick on the fp32 path) after the one in
    // hand: 0 nothing, 1 the packed image of a quantised brick (or nothing to load: air), 2 a half's
    // fp32 values
    int pf_kind = 0;
    int item = 0;
    for (;;) {
        __syncthreads();  // every wave is done with the previous brick's LDS
        DDRR_PROF(PROF_BARRIER);
        const bool next_half = sub + 1 < n_sub;
        const bool known = !next_half && la_item >= 0;  // no claim needed
        const bool loaded = pf_kind != 0;
        if (tid == 0) {
            if (!next_half && !known) counter[1] = atomicAdd(p.work, 1);
            counter[2] = 0;
            counter[3] = 0;
        }
        // (a brick that arrives in registers uses neither counter[1] nor [2]; [3] is read behind
        // later barriers)
        if (!loaded) __syncthreads();
        float cur_lo = 0.f, cur_hi = 0.f;  // (the range of a brick that arrives in registers)
        if (next_half) {
            ++sub;
        } else {
            // Work items: bricks in the order p.order hands them out (heaviest first, see
            // brick_weight_kernel), every pose
            item = known ? la_item : counter[1];
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
            // (tools builds: the last split_t bricks go out in split_s parts of the pose batch each --
            // measured, not adopted: the smaller batches cost more than the shorter tail gains)
            const int n_full = n_bricks - p.split_t;
            if (item >= n_full + p.split_t * p.split_s) break;
            const int kth = item < n_full ? item : n_full + (item - n_full) / p.split_s;
            const int part = item < n_full ? 0 : (item - n_full) % p.split_s;
            const int parts = item < n_full ? 1 : p.split_s;
            pose_lo = (int)((long)p.B * part / parts);
            pose_hi = (int)((long)p.B * (part + 1) / parts);
#else
            if (item >= n_bricks) break;
            const int kth = item;
#endif
            brick_id = known ? la_brick : (p.order ? p.order[kth] : kth);
            cur_lo = la_lo;
            cur_hi = la_hi;
            la_item = -1;  // (taken; la_after stays with the workgroup)
            sub = 0;
            n_sub = 1;
            if (C::MIXED) {
                f32_brick = stale || (known ? la_f32 : (p.fallback && p.fallback[brick_id] != 0));
                const int z0 = (brick_id % nbz) * C::BZ;
                n_sub = f32_brick && p.D.z - z0 > C::HZ ? 2 : 1;
            }
        }
        // the requests of this iteration's look-ahead (lane 0 of wave 1; published before the
        // staging barrier below): the item after the next -- unless this item's last part is not
        // in hand yet, or the launch is in its last two rounds -- and the next item's brick
        const bool last_part = sub + 1 >= n_sub;
        const bool look = LOOK && last_part;
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
        const int look_rounds = (p.dbg >> 13) & 3 ? 1 + ((p.dbg >> 13) & 3) : kLookRounds;  // (tools: 2 .. 4)
#else
        constexpr int look_rounds = kLookRounds;
#endif
        const bool claim = look && item + look_rounds * n_wg < n_bricks;
        int nx_item = la_after;  // (claimed by an earlier iteration)
        int req_item = -1, req_brick = 0, req_fb = 0;
        float req_lo = 0.f, req_hi = 0.f;
        if (tid == 64 && look) {
            // (the workgroup's first look-ahead claims both, one after the other: once per launch)
            if (nx_item < 0 && claim) nx_item = atomicAdd(p.work, 1);
            if (claim) req_item = atomicAdd(p.work, 1);
            if (nx_item >= 0 && nx_item < n_bricks) {
                req_brick = p.order ? p.order[nx_item] : nx_item;
                req_fb = stale ? 1 : (C::MIXED && p.fallback ? p.fallback[req_brick] : 0);
                req_lo = p.ranges[2 * req_brick];
                req_hi = p.ranges[2 * req_brick + 1];
            }
        }
#if defined(DDRR_BRICK_PROFILE)
        const unsigned long long brick_t0 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) ahead[6] = (int)(unsigned)brick_t0;  // (read behind the staging / pool barrier)
#endif
        const int n_chunks = (pose_hi - pose_lo + C::CHUNK - 1) / C::CHUNK;
        const int chunk = n_chunks ? (pose_hi - pose_lo + n_chunks - 1) / n_chunks : 0;
        DDRR_PROF(PROF_CLAIM);
        Box box = cfg_brick_box<C>(p.D, nby, nbz, brick_id);
        if (C::MIXED && f32_brick) {
            box.lo[2] += sub * C::HZ;
            box.hi[2] = box.lo[2] + C::HZ < box.hi[2] ? box.lo[2] + C::HZ : box.hi[2];
        }
        const BoxF cells = boxf(box);
        StepGeom SG;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            SG.lof[a] = (float)box.lo[a];
            SG.hif[a] = (float)box.hi[a];
        }
        const bool half = C::MIXED && f32_brick;
        SG.strideb[0] = bits_as_float((unsigned)(half ? C::FSX : C::SX));
        SG.strideb[1] = bits_as_float((unsigned)(half ? C::FSY : C::SY));
        SG.strideb[2] = bits_as_float((unsigned)(half ? 4 : C::ES));
        int qn0 = 0, qn1 = 0, qn2 = 0;  // hits waiting per length class (wave-uniform)
        bool brick_empty = false, pool_open = false;
        Q16Range range = {0.f, 0.f, 0.f};
        // (the thread's offsets into the image are worked out where they are used: hoisted out of
        // the brick loop they cost the walk two registers it does not have)
        int tid_pf = tid;
        asm volatile("" : "+v"(tid_pf));
        if (pf_kind == 1) {
            // the image was requested by the previous iteration: registers -> LDS (first, so that
            // the registers are free again before the row table is worked out)
            range = q16_range(cur_lo, cur_hi);
            brick_empty = cur_lo == 0.f && cur_hi == 0.f;
            if (!brick_empty) pf.store(brick, tid_pf);
        }
        if constexpr (C::MIXED) {
            // (a half that arrives in registers is walked whatever it holds: no count of non-zeros)
            if (pf_kind == 2) pf.template half<typename C::Half>(p, brick, box, tid_pf, false);
        }
        DDRR_TRACE(8, 0)
        DDRR_TRACE(10, 1)

        for (int ch = 0; ch < n_chunks; ++ch) {  // chunks of at most C::CHUNK poses, of equal size
            const int b0 = pose_lo + ch * chunk;
            const int nb = pose_hi - b0 < chunk ? pose_hi - b0 : chunk;
            const bool last_chunk = ch == n_chunks - 1;
            if (ch > 0) __syncthreads();  // previous chunk's table no longer in use
            // (per-chunk code: what it derives from the launch constants and the thread id is
            // kept out of the registers the walk needs -- the compiler would hoist it out of the
            // brick loop and spill it -- by making the inputs opaque here)
            int det_h = p.det_h, det_w = p.det_w, tid_here = tid;
            asm volatile("" : "+s"(det_h), "+s"(det_w), "+v"(tid_here));
            if (tid < nb) {
                const PoseGrid pg = pose_grid(p.source + (long)(b0 + tid) * 3,
                                              p.target + (long)(b0 + tid) * N * 3, det_h, det_w);
                PixBox pb = project_brick_grid(pg, det_h, det_w, cells, p.shift);
                if (GROUPED && !(p.dbg & 1024)) pb = align_pixbox_rows(pb, det_w);
                rows[tid] = fwd_row(brick_row(pg, pb, cells, p.shift, p.eps, 0.f));
            }
            if (tid == 0) counter[0] = 0;
            DDRR_PROF(PROF_ROWS);
            if (ch == 0) DDRR_TRACE(9, 0)
            if (ch == 0 && !loaded) {
                if constexpr (C::MIXED) {
                    if (f32_brick)
                        fwd_stage_brick<typename C::Half>(p, brick, box, brick_id, tid_here, range,
                                                          brick_empty, counter);
                    else if (p.packed)
                        fwd_stage_packed<C>(p, brick, brick_id, tid_here, range, brick_empty);
                    else
                        fwd_stage_brick<C>(p, brick, box, brick_id, tid_here, range, brick_empty, counter);
                } else if (C::Q16 && p.packed) {
                    fwd_stage_packed<C>(p, brick, brick_id, tid_here, range, brick_empty);
                } else {
                    fwd_stage_brick<C>(p, brick, box, brick_id, tid_here, range, brick_empty, counter);
                }
            }
            if (ch == 0 && look && tid == 64) {
                ahead[0] = nx_item;
                ahead[1] = req_brick | (req_fb ? kLaStage : 0);
                ahead[2] = __float_as_int(req_lo);
                ahead[3] = __float_as_int(req_hi);
                ahead[4] = req_item;
            }
            DDRR_PROF(PROF_STORE);
            if (ch == 0) DDRR_TRACE(4, 0)
            if (ch == 0) DDRR_TRACE(7, 1)
            __syncthreads();
            if (ch == 0) DDRR_TRACE(0, 0)
            if ((!C::Q16 || f32_brick) && ch == 0 && !loaded) brick_empty = counter[2] == 0;
            // units per pose -> inclusive prefix, held by every wave in registers (lane k: pose k)
            int incl = lane < nb ? (rows[lane].count + 63) >> 6 : 0;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                incl += lane >= o ? up : 0;
            }
            const int units = brick_empty ? 0 : __builtin_amdgcn_readlane(incl, 31);
            int cur = 0, cur_lo = 0, cur_hi = __builtin_amdgcn_readlane(incl, 0);
            DDRR_PROF(PROF_STAGE);
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
                    const BrickRow r = brick_row_of(rows[cur]);
                    DDRR_PROF(PROF_PULL);
                    DDRR_PROF_COUNT(PROF_N_UNITS, 1);
                    const int local = (u - cur_lo) * 64 + lane;
                    const bool valid = local < uni(r.count);
                    int pix = 0;
                    float n_est = 0.f;
                    bool hit = valid && brick_candidate(r, local, p.det_w, pix, n_est);
                    // a subsample of the detector (reference drr.py:36-39, p_subsample): pixels whose
                    // bit is not set are no candidates -- the walks, the ray loads and the atomics of
                    // nine tenths of the rays at p_subsample = 0.1
                    if constexpr (SUB) hit = hit && ((p.pix_mask[pix >> 5] >> (pix & 31)) & 1u) != 0u;
                    // float record: classes per run of 8 adjacent pixels (see bricks.hip)
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
                // Walk every full batch of 64 hits of one class.  What the waves have left at
                // the end of the brick -- on average half a batch per class each, a quarter of
                // the brick's batches at 32 poses -- is pooled: one list of segments (class,
                // wave), longest class first, cut into batches of 64 that an LDS counter hands
                // out.  Fewer, fuller, better sorted batches than every wave draining its own,
                // and the end of the brick is balanced over the waves.
                for (;;) {
                    int n = 0;
                    unsigned e = 0;
                    if (!pool_open) {
                        int k = -1;
                        if (qn0 >= 64) k = 0;
                        else if (qn1 >= 64) k = 1;
                        else if (qn2 >= 64) k = 2;
                        if (k >= 0) {
                            const int base = (k == 0 ? qn0 : (k == 1 ? qn1 : qn2)) - 64;
                            qn0 -= k == 0 ? 64 : 0;
                            qn1 -= k == 1 ? 64 : 0;
                            qn2 -= k == 2 ? 64 : 0;
                            e = myq[k * kQueueCap + base + lane];
                            n = 64;
                        } else if (drain && POOLED) {
                            // (every wave gets here exactly once per brick: drain is only seen
                            // in the last chunk)
                            if (lane < kBuckets)
                                pool[wave * kBuckets + lane] = lane == 0 ? qn0 : (lane == 1 ? qn1 : qn2);
                            DDRR_TRACE(1, 0)
                            __syncthreads();
                            DDRR_TRACE(2, 0)
                            DDRR_PROF(PROF_BARRIER);
                            pool_open = true;
                            continue;
                        } else if (drain && qn0 + qn1 + qn2 > 0) {
                            // (no room for the pool: the wave's own virtual queue
                            // [class 2 | class 1 | class 0], taken from the front)
                            const int tot = qn0 + qn1 + qn2;
                            n = tot < 64 ? tot : 64;
                            const int i2 = lane, i1 = lane - qn2, i0 = lane - qn2 - qn1;
                            if (lane < n)
                                e = i2 < qn2 ? myq[2 * kQueueCap + qn2 - 1 - i2]
                                             : (i1 < qn1 ? myq[kQueueCap + qn1 - 1 - i1]
                                                         : myq[qn0 - 1 - i0]);
                            const int t2 = qn2 < n ? qn2 : n;
                            const int t1 = qn1 < n - t2 ? qn1 : n - t2;
                            qn2 -= t2;
                            qn1 -= t1;
                            qn0 -= n - t2 - t1;
                        } else {
                            break;
                        }
                    } else {
                        constexpr int NSEG = C::WAVES * kBuckets;
                        int j = 0;
                        if (lane == 0) j = atomicAdd(&counter[3], 1);
                        j = uni(j) * 64;
                        // lane s: end of segment s = (class 2 - s / WAVES, wave s % WAVES) in the list
                        int end = lane < NSEG ? pool[(lane % C::WAVES) * kBuckets + kBuckets - 1 - lane / C::WAVES] : 0;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            const int up = __shfl_up(end, o, 64);
                            end += lane >= o ? up : 0;
                        }
                        const int total = __builtin_amdgcn_readlane(end, 63);
                        if (j >= total) break;
                        // the segments that reach into this batch are [kf, kl]
                        const int kf = (int)__popcll(__ballot(lane < NSEG && end <= j));
                        const int kl = (int)__popcll(__ballot(lane < NSEG && end <= j + 63));
                        const int g = j + lane;
                        int seg = kf, base = kf ? __builtin_amdgcn_readlane(end, uni(kf - 1)) : 0;
                        for (int k = kf; k < kl; ++k) {
                            const int ek = __builtin_amdgcn_readlane(end, uni(k));
                            if (ek <= g) seg = k + 1, base = ek;
                        }
                        n = total - j < 64 ? total - j : 64;
                        if (lane < n)
                            e = queue[((seg % C::WAVES) * kBuckets + kBuckets - 1 - seg / C::WAVES) * kQueueCap + (g - base)];
                    }
                    DDRR_PROF(PROF_POP);
                    DDRR_PROF_COUNT(PROF_N_BATCH, 1);
                    fwd_item<AUX, C>(p, lds_base, SG, range, lane < n, e >> p.pix_bits,
                                     e & pix_mask, out, aux, prof, rows, b0, f32_brick);
                    wave_fence();
                }
                if (drain) break;
            }
        }
        DDRR_TRACE(3, 0)
        DDRR_TRACE(6, 1)
        // this wave has nothing left to walk: its share of the next unit, into registers
        pf_kind = 0;
        int half_of = -1, half_sub = 0;  // (a brick on the fp32 path whose half comes next)
        if (C::MIXED && LOOK && !last_part) {
            half_of = brick_id;  // the other half of the brick in hand
            half_sub = sub + 1;
        } else if (look) {
            // what lane 0 of wave 1 published before this part's staging barrier
            la_item = uni(ahead[0]);
            la_after = uni(ahead[4]);
            const int b = uni(ahead[1]);
            la_brick = b & ~kLaStage;
            la_lo = __int_as_float(uni(ahead[2]));
            la_hi = __int_as_float(uni(ahead[3]));
            la_f32 = (b & kLaStage) != 0;
            if (la_item >= 0 && la_item < n_bricks) {
                if (la_f32) half_of = la_brick;
                else pf_kind = 1;
            }
        }
        int tid_ld = tid;
        asm volatile("" : "+v"(tid_ld));
        if (pf_kind == 1 && !(la_lo == 0.f && la_hi == 0.f)) {
            pf.load(p, la_brick, tid_ld);
        } else if (C::MIXED && half_of >= 0) {
            if constexpr (C::MIXED) {
                Box nbox = cfg_brick_box<C>(p.D, nby, nbz, half_of);
                nbox.lo[2] += half_sub * C::HZ;
                nbox.hi[2] = nbox.lo[2] + C::HZ < nbox.hi[2] ? nbox.lo[2] + C::HZ : nbox.hi[2];
                pf.clear();
                pf.template half<typename C::Half>(p, brick, nbox, tid_ld, true);
                pf_kind = 2;
            }
        } else {
            pf.clear();
        }
        DDRR_TRACE(5, 1)
#if defined(DDRR_BRICK_PROFILE)
        __syncthreads();
        if (tid == 0 && p.brick_times)
            p.brick_times[brick_id] = (unsigned)(__builtin_amdgcn_s_memrealtime() - brick_t0);
#endif
    }
#if defined(DDRR_BRICK_PROFILE)
    DDRR_PROF(PROF_BARRIER);
    if (lane == 0 && p.prof) {
        for (int i = 0; i < 16; ++i) atomicAdd(p.prof + i, prof.t[i]);
        // how long this wave lived: the launch ends with its longest-lived one (the tail)
        // [16] sum of the waves' start ticks, [17] of their end ticks, [18] earliest start (stored
        // inverted: the slots start at 0), [19] latest end
        const unsigned long long died = __builtin_amdgcn_s_memrealtime();
        atomicAdd(p.prof + 16, prof.born);
        atomicAdd(p.prof + 17, died);
        atomicMax(p.prof + 18, ~prof.born);
        atomicMax(p.prof + 19, died);
    }
#endif
}

// ------------------------------------------------------------------ shared length-class rings
// The same kernel with the hit queues SHARED by the workgroup: NCLS length classes (equal
// width in estimated crossings) instead of 3, each a ring of CAP entries in LDS with two
// counters: tail[c], tickets handed out, and head[c], batches of 64 taken -- in order.
//   * A wave's push reserves the tickets of all its classes with one LDS atomic (lane c adds
//     the unit's count of class c to tail[c]; with the float record the first hit of each run of
//     8 pixels reserves for its run, so that runs stay together) and stores ticket t's entry in
//     slot t mod CAP -- once head[c] says the slot's previous ticket, t - CAP, has been taken.
//   * The push whose reservation completes a batch (tickets 64 k .. 64 k + 63) owns it: when
//     head[c] == k it waits for the 64 slots to be written, takes them (reads, marks EMPTY),
//     sets head[c] = k + 1 and walks them.  A push takes ALL the batches it owns before it walks
//     the first one, so that nobody waits for a walk.
//   * Every wait is for a push that reserved EARLIER (a writer for the owner of a batch CAP
//     tickets back; an owner for the writers of its batch and the owner of the batch before):
//     the oldest unfinished push never waits, so there is no cycle, whatever CAP.  The spins are
//     bounded all the same (kSqSpinCap): a logic error ends in a NaN image, not a hang.
//   * When the units of the last chunk are gone the waves meet at a barrier; what is left (< 64
//     per class) is handed out 64 at a time, longest class first.
// With 16 waves feeding one set of rings a class fills 16x faster than a per-wave queue: twelve
// classes cost < 12 partial batches per brick (48 with the 3 per-wave queues), and a batch
// holds rays within one class width of each other.
constexpr unsigned kSqEmpty = 0xffffffffu;
constexpr int kSqSpinCap = 1 << 22;

template <int NCLS_, class C>
struct SqCfg {
    static constexpr int NCLS = NCLS_;
    // ring entries per class: the power of two that fits the per-wave queues' LDS
    static constexpr int ROOM = C::WAVES * kBuckets * kQueueCap / NCLS;
    static constexpr int CAP = ROOM >= 1024 ? 1024 : (ROOM >= 512 ? 512 : (ROOM >= 256 ? 256 : 128));
    static_assert(ROOM >= 128, "ring too small");
    // (the row table and the control words lie behind the rings exactly as behind the queues)
    static constexpr int LDS = C::LDS + 64 + 8 * NCLS;
};

template <bool AUX, class C, int NCLS>
__global__ __launch_bounds__(C::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void
siddon_fwd_brick_sq_kernel(BrickArgs p, float *__restrict__ out, float *__restrict__ aux) {
    using S = SqCfg<NCLS, C>;
    constexpr int CAP = S::CAP;
    static_assert(NCLS * 63 <= 64 * C::WAVES, "the leftovers of a brick: one batch per wave at most");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned char *brick = smem_raw;
    unsigned *ring = reinterpret_cast<unsigned *>(smem_raw + C::BRICK_BYTES);  // [NCLS][CAP]
    FwdRow *rows = reinterpret_cast<FwdRow *>(ring + C::WAVES * kBuckets * kQueueCap);
    int *counter = reinterpret_cast<int *>(rows + C::CHUNK);  // [0] unit, [1] brick, [2] non-zero
    volatile int *tail = counter + 4;                         // [NCLS] tickets handed out
    volatile int *head = tail + NCLS;                         // [NCLS] batches taken (in order)
    volatile int *fault = head + NCLS;                        // a spin ran into its bound

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nbx = (p.D.x + C::BX - 1) / C::BX, nby = (p.D.y + C::BY - 1) / C::BY;
    const int nbz = (p.D.z + C::BZ - 1) / C::BZ;
    const int n_bricks = nbx * nby * nbz;
    const int N = p.det_h * p.det_w;
    const unsigned pix_mask = (1u << p.pix_bits) - 1u;
    const int n_chunks = (p.B + C::CHUNK - 1) / C::CHUNK;
    const int chunk = (p.B + n_chunks - 1) / n_chunks;
    const unsigned lds_base = LdsAbsFetch::base_of(reinterpret_cast<const float *>(brick));
    const bool GROUPED = AUX && p.rec_q == 0.f && !(p.dbg & 8);
    const float inv_width = 1.0f / p.t1;  // class = estimated crossings / width

    for (int i = tid; i < NCLS * CAP; i += C::THREADS) ring[i] = kSqEmpty;
    if (tid == 0) *fault = 0;

    BrickProf prof;
#if defined(DDRR_BRICK_PROFILE)
    prof.start();
#endif
    // take the 64 entries of batch k of class c (lane l: ticket 64 k + l); in order
    auto take_batch = [&](int c, int k) -> unsigned {
        int spin = 0;
        while (head[c] != k && ++spin < kSqSpinCap) {
        }
        volatile unsigned *slot = ring + c * CAP + ((64 * k + lane) & (CAP - 1));
        unsigned e = *slot;
        while (__ballot(e == kSqEmpty) && ++spin < kSqSpinCap) e = *slot;
        if (spin >= kSqSpinCap) *fault = 1;
        *slot = kSqEmpty;
        wave_fence();
        if (lane == 0) head[c] = k + 1;  // (LDS operations of a wave are performed in order)
        return e;
    };
    auto walk_entries = [&](unsigned e, const StepGeom &SG, const Q16Range &range) {
        DDRR_PROF(PROF_POP);
        DDRR_PROF_COUNT(PROF_N_BATCH, 1);
        fwd_item<AUX, C>(p, lds_base, SG, range, e != kSqEmpty, e >> p.pix_bits, e & pix_mask, out,
                         aux, prof);
    };

    for (;;) {
        __syncthreads();  // every wave is done with the previous brick's LDS
        DDRR_PROF(PROF_BARRIER);
        if (tid == 0) {
            counter[1] = atomicAdd(p.work, 1);
            counter[2] = 0;
        }
        if (tid < NCLS) {
            tail[tid] = 0;
            head[tid] = 0;
        }
        __syncthreads();
        const int brick_id = counter[1];
        if (brick_id >= n_bricks) break;
        DDRR_PROF(PROF_CLAIM);
        const Box box = cfg_brick_box<C>(p.D, nby, nbz, brick_id);
        const BoxF cells = boxf(box);
        StepGeom SG;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            SG.lof[a] = (float)box.lo[a];
            SG.hif[a] = (float)box.hi[a];
        }
        SG.strideb[0] = bits_as_float((unsigned)C::SX);
        SG.strideb[1] = bits_as_float((unsigned)C::SY);
        SG.strideb[2] = bits_as_float((unsigned)C::ES);
        bool brick_empty = false;
        Q16Range range = {0.f, 0.f, 0.f};

        for (int ch = 0; ch < n_chunks; ++ch) {
            const int b0 = ch * chunk;
            const int nb = p.B - b0 < chunk ? p.B - b0 : chunk;
            const bool last_chunk = ch == n_chunks - 1;
            if (ch > 0) __syncthreads();  // previous chunk's table no longer in use
            if (tid < nb) {
                const PoseGrid pg = pose_grid(p.source + (long)(b0 + tid) * 3,
                                              p.target + (long)(b0 + tid) * N * 3, p.det_h, p.det_w);
                PixBox pb = project_brick_grid(pg, p.det_h, p.det_w, cells, p.shift);
                if (GROUPED) pb = align_pixbox_rows(pb, p.det_w);
                rows[tid] = fwd_row(brick_row(pg, pb, cells, p.shift, p.eps, 0.f));
            }
            if (tid == 0) counter[0] = 0;
            DDRR_PROF(PROF_ROWS);
            if (ch == 0) fwd_stage_brick<C>(p, brick, box, brick_id, tid, range, brick_empty, counter);
            DDRR_PROF(PROF_STORE);
            __syncthreads();
            if (!C::Q16 && ch == 0) brick_empty = counter[2] == 0;
            int incl = lane < nb ? (rows[lane].count + 63) >> 6 : 0;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                incl += lane >= o ? up : 0;
            }
            const int units = brick_empty ? 0 : __builtin_amdgcn_readlane(incl, 31);
            int cur = 0, cur_lo = 0, cur_hi = __builtin_amdgcn_readlane(incl, 0);
            DDRR_PROF(PROF_STAGE);
            for (;;) {
                int u = 0;
                if (lane == 0) u = atomicAdd(&counter[0], 1);
                u = uni(u);
                if (u >= units) break;  // no unit left in this chunk
                while (u >= cur_hi) {   // units arrive in increasing order: forward cursor
                    ++cur;
                    cur_lo = cur_hi;
                    cur_hi = __builtin_amdgcn_readlane(incl, uni(cur));
                }
                const BrickRow r = brick_row_of(rows[cur]);
                DDRR_PROF(PROF_PULL);
                DDRR_PROF_COUNT(PROF_N_UNITS, 1);
                const int local = (u - cur_lo) * 64 + lane;
                const bool valid = local < uni(r.count);
                int pix = 0;
                float n_est = 0.f;
                const bool hit = valid && brick_candidate(r, local, p.det_w, pix, n_est);
                float n_grp = hit ? n_est : 0.f;
                if (GROUPED) {  // float record: classes per run of 8 adjacent pixels (bricks.hip)
                    n_grp = fmaxf(n_grp, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, n_grp), 0xB1, 0xf, 0xf, true)));   // lane ^ 1
                    n_grp = fmaxf(n_grp, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, n_grp), 0x4E, 0xf, 0xf, true)));   // lane ^ 2
                    n_grp = fmaxf(n_grp, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, n_grp), 0x141, 0xf, 0xf, true)));  // 7 - lane
                }
                int cls = (int)(n_grp * inv_width);
                cls = cls < NCLS - 1 ? cls : NCLS - 1;
                cls = cls > 0 ? cls : 0;
                const unsigned long long hits = __ballot(hit);
                DDRR_PROF_COUNT(PROF_N_HITS, __popcll(hits));
                // Tickets: `own_cls` / `own_batch` on the lanes whose reservation completed a batch
                int ticket = 0, own_cls = 0, own_batch = 0;
                bool owns = false;
                if (!GROUPED) {
                    // per class present in the unit: its hits' ranks, its count on lane c
                    int cnt_mine = 0, rank = 0;
                    unsigned long long rem = hits;
                    while (rem) {
                        const int c = __builtin_amdgcn_readlane(cls, __ffsll((long long)rem) - 1);
                        const unsigned long long m = __ballot(hit && cls == c);
                        cnt_mine = lane == c ? (int)__popcll(m) : cnt_mine;
                        rank = cls == c ? lane_rank(m) : rank;
                        rem &= ~m;
                    }
                    int first = 0;  // one LDS atomic reserves for all classes: lane c, class c
                    if (lane < NCLS && cnt_mine > 0) first = atomicAdd((int *)&tail[lane], cnt_mine);
                    owns = lane < NCLS && cnt_mine > 0 && ((first + cnt_mine) >> 6) != (first >> 6);
                    own_cls = lane;
                    own_batch = first >> 6;
                    ticket = __shfl(first, cls, 64) + rank;
                } else {
                    // float record: the hits of a run of 8 pixels share a class and stay together
                    const unsigned half = lane < 32 ? (unsigned)hits : (unsigned)(hits >> 32);
                    const unsigned grp = (half >> (lane & 24)) & 0xffu;  // hits of my run of 8
                    const int n_run = __popc(grp), before = __popc(grp & ((1u << (lane & 7)) - 1u));
                    const bool leader = hit && before == 0;
                    int first = 0;
                    if (leader) first = atomicAdd((int *)&tail[cls], n_run);
                    owns = leader && ((first + n_run) >> 6) != (first >> 6);
                    own_cls = cls;
                    own_batch = first >> 6;
                    const int lead_lane = (lane & ~7) + (__ffs(grp | 0x100u) - 1);
                    ticket = __shfl(first, lead_lane & 63, 64) + before;
                }
                if (hit) {
                    // slot t mod CAP is free once batch (t - CAP) / 64 has been taken
                    const int need = (ticket >> 6) - CAP / 64 + 1;
                    int spin = 0;
                    while (head[cls] < need && ++spin < kSqSpinCap) {
                    }
                    if (spin >= kSqSpinCap) *fault = 1;
                    const_cast<volatile unsigned *>(ring)[cls * CAP + (ticket & (CAP - 1))] =
                        ((unsigned)(b0 + cur) << p.pix_bits) | (unsigned)pix;
                }
                wave_fence();
                DDRR_PROF(PROF_PHASE_A);
                // the batches this push completed are this wave's: take them all, then walk them
                unsigned long long own = __ballot(owns);
                while (own) {
                    unsigned e[4];
                    int n_own = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        e[j] = kSqEmpty;
                        if (own) {
                            const int l = __ffsll((long long)own) - 1;
                            own &= own - 1;
                            e[j] = take_batch(__builtin_amdgcn_readlane(own_cls, l),
                                              __builtin_amdgcn_readlane(own_batch, l));
                            n_own = j + 1;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < n_own) walk_entries(e[j], SG, range);
                }
            }
            if (!last_chunk) continue;
            // every push of this brick has been made: hand out what is left, longest class first
            __syncthreads();
            DDRR_PROF(PROF_BARRIER);
            int left_before = 0;  // leftovers of the classes above c
            unsigned e = kSqEmpty;
            const int pos = 64 * wave + lane;
#pragma unroll 1
            for (int c = NCLS - 1; c >= 0; --c) {
                const int tl = tail[c], left = tl & 63;  // (every complete batch has been taken)
                const int i = pos - left_before;
                if (i >= 0 && i < left) {
                    unsigned *slot = ring + c * CAP + ((tl - left + i) & (CAP - 1));
                    e = *slot;
                    *slot = kSqEmpty;
                }
                left_before += left;
            }
            if (64 * wave < left_before) {
                DDRR_PROF(PROF_POP);
                DDRR_PROF_COUNT(PROF_N_BATCH, 1);
                fwd_item<AUX, C>(p, lds_base, SG, range, e != kSqEmpty, e >> p.pix_bits, e & pix_mask,
                                 out, aux, prof);
            }
        }
    }
#if defined(DDRR_BRICK_PROFILE)
    DDRR_PROF(PROF_BARRIER);
    if (lane == 0 && p.prof)
        for (int i = 0; i < 16; ++i) atomicAdd(p.prof + i, prof.t[i]);
#endif
    // (a spin that ran into its bound: poison the result instead of returning a wrong image)
    if (tid == 0 && *fault) (AUX ? aux : out)[0] = NAN;
}

// (vmin, vmax) of every brick of a BX x BY x BZ grid, the input of the 16-bit staging
// (q16_range), and whether the brick can be quantised at all (brick_step.h q16_usable): its
// LEVEL is the smallest mean |V| of any of its 4 x 4 x 4 blocks -- over the non-zero voxels when
// the brick's minimum is 0 (zeros are then stored exactly: air after the HU -> density
// transform) -- and fallback[id] = 1 sends the brick to the fp32 path.  One workgroup per brick.
constexpr int kSubBlocks = 1024;  // 4^3 blocks of a brick, at most
__global__ __launch_bounds__(256) void brick_range_kernel(const float *__restrict__ vol, Dims D,
                                                          int BX, int BY, int BZ, int nby, int nbz,
                                                          float *__restrict__ ranges,
                                                          int *__restrict__ fallback, int vec) {
    __shared__ float red[3][4];
    __shared__ float sub_sum[kSubBlocks];
    __shared__ int sub_nnz[kSubBlocks];
    const int id = blockIdx.x;
    const int bz = id % nbz, by = (id / nbz) % nby, bx = id / (nbz * nby);
    const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
    const int nx = min(BX, D.x - x0), ny = min(BY, D.y - y0), nz = min(BZ, D.z - z0);
    const int QZ = BZ / 4, quads = nx * ny * QZ;
    const int SBY = (BY + 3) / 4, n_sub = ((BX + 3) / 4) * SBY * QZ;
    for (int k = threadIdx.x; k < kSubBlocks; k += 256) {
        sub_sum[k] = 0.f;
        sub_nnz[k] = 0;
    }
    __syncthreads();
    float tmin = INFINITY, tmax = -INFINITY;
    bool bad = false;
    for (int k = threadIdx.x; k < quads; k += 256) {
        const int qz = k % QZ, row = k / QZ, ly = row % ny, lx = row / ny;
        if (qz * 4 >= nz) continue;
        const float *g = vol + ((long)(x0 + lx) * D.y + (y0 + ly)) * D.z + z0 + qz * 4;
        const int nv = nz - qz * 4;  // voxels of the quad inside the volume (vec: always >= 4)
        float4 v;
        if (vec) {
            v = *reinterpret_cast<const float4 *>(g);
        } else {
            // (what lies outside repeats the quad's first voxel for min / max and counts 0 below)
            v.x = g[0];
            v.y = nv > 1 ? g[1] : v.x;
            v.z = nv > 2 ? g[2] : v.x;
            v.w = nv > 3 ? g[3] : v.x;
        }
        tmin = fminf(fminf(fminf(tmin, v.x), fminf(v.y, v.z)), v.w);
        tmax = fmaxf(fmaxf(fmaxf(tmax, v.x), fmaxf(v.y, v.z)), v.w);
        const float sum = (fabsf(v.x) + (nv > 1 ? fabsf(v.y) : 0.f)) +
                          ((nv > 2 ? fabsf(v.z) : 0.f) + (nv > 3 ? fabsf(v.w) : 0.f));
        bad = bad || !(sum < INFINITY);  // (min / max drop NaNs)
        const int sb = ((lx >> 2) * SBY + (ly >> 2)) * QZ + qz;
        if (n_sub <= kSubBlocks) {
            atomicAdd(&sub_sum[sb], sum);
            atomicAdd(&sub_nnz[sb], (v.x != 0.f) + (nv > 1 && v.y != 0.f) + (nv > 2 && v.z != 0.f) +
                                        (nv > 3 && v.w != 0.f));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, o, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
    }
    bad = __ballot(bad) != 0ull;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = tmin;
        red[1][wave] = bad ? NAN : tmax;
    }
    __syncthreads();
    float lo = INFINITY, hi = -INFINITY;
    bool nan = false;
    for (int w = 0; w < 4; ++w) {
        lo = fminf(lo, red[0][w]);
        hi = fmaxf(hi, red[1][w]);
        nan = nan || red[1][w] != red[1][w];
    }
    if (!(lo <= hi)) lo = hi = 0.f;  // (no voxel)
    // level: the smallest block mean; a block's mean is over the voxels the quantisation can be
    // wrong about -- all of them, or the non-zero ones when 0 is the brick's minimum (q = 0)
    float level = INFINITY;
    for (int k = threadIdx.x; k < n_sub && n_sub <= kSubBlocks; k += 256) {
        const int qz = k % QZ, sy = (k / QZ) % SBY, sx = k / (QZ * SBY);
        const int cx = min(4, nx - 4 * sx), cy = min(4, ny - 4 * sy), cz = min(4, nz - 4 * qz);
        if (cx <= 0 || cy <= 0 || cz <= 0) continue;
        const int n_eff = lo == 0.f ? sub_nnz[k] : cx * cy * cz;
        if (n_eff > 0) level = fminf(level, sub_sum[k] / (float)n_eff);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) level = fminf(level, __shfl_xor(level, o, 64));
    if ((threadIdx.x & 63) == 0) red[2][wave] = level;
    __syncthreads();
    if (threadIdx.x == 0) {
        level = fminf(fminf(red[2][0], red[2][1]), fminf(red[2][2], red[2][3]));
        if (n_sub > kSubBlocks) level = 0.f;  // (a brick shape this kernel has no blocks for)
        const float vmax = nan ? NAN : hi;
        ranges[2 * id] = lo;
        ranges[2 * id + 1] = vmax;
        fallback[id] = q16_usable(lo, vmax, level) ? 0 : 1;
    }
}

// header word 0 of the brick workspace: how many bricks take the fp32 path
__global__ __launch_bounds__(256) void brick_fallback_count_kernel(const int *__restrict__ fallback,
                                                                   int n_bricks, int *__restrict__ header) {
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int n = 0;
    for (int k = threadIdx.x; k < n_bricks; k += 256) n += fallback[k] != 0;
    atomicAdd(&total, n);
    __syncthreads();
    if (threadIdx.x == 0) {
        header[0] = total;
        header[1] = n_bricks;
        header[2] = 0;  // launches that found the volume changed under the workspace (fingerprint)
    }
}

// The build pass's fingerprint of the volume (brick_core.h kFingerprintWords).
__global__ __launch_bounds__(1024) void brick_fingerprint_kernel(const float *__restrict__ vol, long n_vox,
                                                                 unsigned *__restrict__ fp) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < kFingerprintWords) fp[i] = __float_as_uint(vol[fingerprint_index(i, n_vox)]);
}

// ------------------------------------------------------------------ heaviest bricks first
// The persistent workgroups pull bricks from one counter; a launch ends when the last of them is
// done, and with 8-16 bricks per CU a heavy brick drawn late leaves the other CUs idle for most
// of its duration (measured: 17 % of the wave-time of a 32-pose launch was that tail).  So the
// bricks are handed out by decreasing weight = sum over the poses of the brick's projected pixel
// box (what phase A will enumerate): brick_weight_kernel (one wave per brick, lanes over poses),
// brick_order_kernel (one workgroup: counting sort by 1024 weight classes).  ~5 us per launch,
// forward 1.26 -> 1.19 ms at 32 poses, 4.39 -> 4.00 ms at 128 (profiles/r03/brick_order_split.txt).
__global__ __launch_bounds__(256) void brick_weight_kernel(BrickArgs p, int BX, int BY, int BZ,
                                                           int nby, int nbz, int n_bricks,
                                                           float *__restrict__ weight) {
    const int brick = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (brick >= n_bricks) return;
    const int bz = brick % nbz, by = (brick / nbz) % nby, bx = brick / (nbz * nby);
    BoxF cells;
    cells.lo[0] = (float)(bx * BX), cells.lo[1] = (float)(by * BY), cells.lo[2] = (float)(bz * BZ);
    cells.hi[0] = (float)min(bx * BX + BX, p.D.x), cells.hi[1] = (float)min(by * BY + BY, p.D.y);
    cells.hi[2] = (float)min(bz * BZ + BZ, p.D.z);
    const int N = p.det_h * p.det_w;
    float w = 0.f;
    for (int b = lane; b < p.B; b += 64) {
        const PoseGrid pg = pose_grid(p.source + (long)b * 3, p.target + (long)b * N * 3, p.det_h, p.det_w);
        w += (float)pixbox_count(project_brick_grid(pg, p.det_h, p.det_w, cells, p.shift));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
    // (a brick on the fp32 path is walked as two halves, one after the other)
    if (lane == 0) weight[brick] = p.fallback && p.fallback[brick] ? 2.f * w : w;
}

constexpr int kOrderClasses = 1024;
__global__ __launch_bounds__(1024) void brick_order_kernel(const float *__restrict__ weight,
                                                           int n_bricks, int *__restrict__ order,
                                                           int *__restrict__ zero4) {
    // (zero4: the launch's brick counter, cleared here instead of by a memset of its own)
    if (zero4 && threadIdx.x < 4) zero4[threadIdx.x] = 0;
    __shared__ int hist[kOrderClasses];
    __shared__ float wmax_s[16];
    const int tid = threadIdx.x;
    float wmax = 0.f;
    for (int i = tid; i < n_bricks; i += 1024) wmax = fmaxf(wmax, weight[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
    if ((tid & 63) == 0) wmax_s[tid >> 6] = wmax;
    hist[tid] = 0;
    __syncthreads();
    wmax = 0.f;
    for (int k = 0; k < 16; ++k) wmax = fmaxf(wmax, wmax_s[k]);
    const float scale = wmax > 0.f ? (float)(kOrderClasses - 1) / wmax : 0.f;
    // class 0 = heaviest
    for (int i = tid; i < n_bricks; i += 1024)
        atomicAdd(&hist[kOrderClasses - 1 - (int)(weight[i] * scale)], 1);
    __syncthreads();
    // exclusive scan of the 1024 class counts (Hillis-Steele in place)
    int v = hist[tid];
    for (int o = 1; o < kOrderClasses; o <<= 1) {
        __syncthreads();
        const int add = tid >= o ? hist[tid - o] : 0;
        __syncthreads();
        hist[tid] += add;
    }
    __syncthreads();
    const int excl = hist[tid] - v;
    __syncthreads();
    hist[tid] = excl;
    __syncthreads();
    for (int i = tid; i < n_bricks; i += 1024)
        order[atomicAdd(&hist[kOrderClasses - 1 - (int)(weight[i] * scale)], 1)] = i;
}

// What a launch has to find zeroed, in one go: the image or the record its atomics add to, and
// the brick counter.
__global__ __launch_bounds__(256) void brick_clear_kernel(float *__restrict__ buf, long n, int *__restrict__ work) {
    if (work && blockIdx.x == 0 && threadIdx.x < 4) work[threadIdx.x] = 0;
    const long stride = (long)gridDim.x * 256, n4 = n >> 2;
    float4 *b4 = reinterpret_cast<float4 *>(buf);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) b4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) buf[(n4 << 2) + threadIdx.x] = 0.f;
}

template <bool AUX, class C, bool PRE = false, bool SUB = false>
int launch_cfg(const BrickArgs &p, int n_cu, float *out, float *aux, hipStream_t st) {
    static_assert(C::LDS <= C::LDS_BUDGET, "LDS budget");
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static bool attr_set[kMaxDev] = {false};
    hipError_t e;
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return fail_hip(e, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDev) return fail(-1, "device index out of range");
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            if ((e = hipFuncSetAttribute(
                     reinterpret_cast<const void *>(&siddon_fwd_brick_kernel<AUX, C, PRE, SUB>),
                     hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS)) != hipSuccess)
                return fail_hip(e, "hipFuncSetAttribute");
            attr_set[dev] = true;
        }
    }
    const int nbx = (p.D.x + C::BX - 1) / C::BX, nby = (p.D.y + C::BY - 1) / C::BY;
    const int nbz = (p.D.z + C::BZ - 1) / C::BZ;
    const int n_bricks = nbx * nby * nbz, slots = n_cu * C::WGS_PER_CU;
    if constexpr (C::Q16) {
        if (!p.ranges) return fail(-1, "DDRR_BRICKS_Q16 needs the brick_ranges workspace");
        if (!p.ranges_valid) {
            hipLaunchKernelGGL(brick_range_kernel, dim3(n_bricks), dim3(256), 0, st, p.vol, p.D, C::BX,
                               C::BY, C::BZ, nby, nbz, const_cast<float *>(p.ranges),
                               const_cast<int *>(p.fallback), p.vec);
            hipLaunchKernelGGL(brick_fallback_count_kernel, dim3(1), dim3(256), 0, st, p.fallback,
                               n_bricks, p.ws_header);
            hipLaunchKernelGGL(brick_fingerprint_kernel, dim3((kFingerprintWords + 1023) / 1024), dim3(1024), 0, st,
                               p.vol, (long)p.D.x * p.D.y * p.D.z, const_cast<unsigned *>(p.fingerprint));
            if (p.packed) {
                static bool pack_attr[kMaxDev] = {false};
                {
                    std::lock_guard<std::mutex> lock(mu);
                    if (!pack_attr[dev]) {
                        if ((e = hipFuncSetAttribute(
                                 reinterpret_cast<const void *>(&brick_pack_kernel<C>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, C::BRICK_BYTES)) !=
                            hipSuccess)
                            return fail_hip(e, "hipFuncSetAttribute");
                        pack_attr[dev] = true;
                    }
                }
                hipLaunchKernelGGL(brick_pack_kernel<C>, dim3(n_bricks), dim3(C::THREADS),
                                   C::BRICK_BYTES, st, p, nby, nbz,
                                   const_cast<unsigned char *>(p.packed));
            }
        }
    } else if (p.packed) {
        return fail(-1, "packed bricks are 16-bit bricks");
    }
    BrickArgs q = p;
    // (the launch's counter is cleared by the order kernel, or -- no order -- together with the image
    // or the record by brick_clear_kernel, or by a memset)
    const bool ordered = order_bricks(q, C::BX, C::BY, C::BZ, nby, nbz, n_bricks, slots, st, true);
    const bool clear_here = p.clear && p.clear_n > 0;
    if (p.clear_n < 0) {
        // (the caller has cleared the image / record and the counter: DDRR_BRICKS_CLEARED)
    } else if (clear_here && (reinterpret_cast<uintptr_t>(p.clear) & 15) == 0) {
        const long blocks = (p.clear_n / 4 + 255) / 256;
        hipLaunchKernelGGL(brick_clear_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks))),
                           dim3(256), 0, st, p.clear, p.clear_n, ordered ? nullptr : q.work);
    } else {
        // (a buffer at an odd address: two memsets as before)
        if (clear_here && (e = hipMemsetAsync(p.clear, 0, sizeof(float) * (size_t)p.clear_n, st)) != hipSuccess)
            return fail_hip(e, "hipMemsetAsync");
        if (!ordered && (e = hipMemsetAsync(q.work, 0, 4 * sizeof(int), st)) != hipSuccess)
            return fail_hip(e, "hipMemsetAsync");
    }
    const dim3 grid(n_bricks < slots ? n_bricks : slots), block(C::THREADS);
    hipLaunchKernelGGL((siddon_fwd_brick_kernel<AUX, C, PRE, SUB>), grid, block, C::LDS, st, q, out, aux);
    return 0;
}

template <bool AUX, class C, int NCLS>
int launch_sq(const BrickArgs &p, int n_cu, float *out, float *aux, hipStream_t st) {
    using S = SqCfg<NCLS, C>;
    static_assert(S::LDS <= C::LDS_BUDGET, "LDS budget");
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static bool attr_set[kMaxDev] = {false};
    hipError_t e;
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return fail_hip(e, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDev) return fail(-1, "device index out of range");
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            if ((e = hipFuncSetAttribute(
                     reinterpret_cast<const void *>(&siddon_fwd_brick_sq_kernel<AUX, C, NCLS>),
                     hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS)) != hipSuccess)
                return fail_hip(e, "hipFuncSetAttribute");
            attr_set[dev] = true;
        }
    }
    const int nbx = (p.D.x + C::BX - 1) / C::BX, nby = (p.D.y + C::BY - 1) / C::BY;
    const int nbz = (p.D.z + C::BZ - 1) / C::BZ;
    const int n_bricks = nbx * nby * nbz, slots = n_cu * C::WGS_PER_CU;
    if (C::Q16) {
        if (!p.ranges) return fail(-1, "DDRR_BRICKS_Q16 needs the brick_ranges workspace");
        if (!p.ranges_valid)
            hipLaunchKernelGGL(brick_range_kernel, dim3(n_bricks), dim3(256), 0, st, p.vol, p.D, C::BX,
                               C::BY, C::BZ, nby, nbz, const_cast<float *>(p.ranges),
                               const_cast<int *>(p.fallback), p.vec);
        // (the shared-ring variants have no fp32 path: tools builds only)
    }
    const dim3 grid(n_bricks < slots ? n_bricks : slots), block(C::THREADS);
    if (p.clear_n >= 0 && hipMemsetAsync(p.work, 0, 4 * sizeof(int), st) != hipSuccess)
        return fail(-1, "hipMemsetAsync");
    if (p.clear && p.clear_n > 0 &&
        hipMemsetAsync(p.clear, 0, sizeof(float) * (size_t)p.clear_n, st) != hipSuccess)
        return fail(-1, "hipMemsetAsync");
    hipLaunchKernelGGL((siddon_fwd_brick_sq_kernel<AUX, C, NCLS>), grid, block, S::LDS, st, p, out, aux);
    return 0;
}

// brick variants (DDRR_BRICKS_* of include/diffdrr_hip.h; the others exist in tools builds)
using CfgF32 = FwdCfg<32, 32, 32, 1024, false>;        // DDRR_BRICKS_F32: 32^3 fp32
using CfgQ16Z64 = FwdCfg<32, 32, 64, 1024, true>;      // DDRR_BRICKS_Q16: 32x32x64 16-bit
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
using CfgQ16x2 = FwdCfg<32, 32, 32, 512, true>;        // 32^3 16-bit, two workgroups per CU
using CfgQ16x1 = FwdCfg<32, 32, 32, 1024, true>;       // 32^3 16-bit, one workgroup per CU
using CfgF32Half = FwdCfg<32, 32, 16, 512, false, 0>;  // 32x32x16 fp32 halves, two workgroups
using CfgQ16X64 = FwdCfg<64, 32, 32, 1024, true>;      // double bricks, 16-bit
using CfgQ16Y64 = FwdCfg<32, 64, 32, 1024, true>;
using CfgF32Y64 = FwdCfg<16, 64, 32, 1024, false>;     // anisotropic fp32 bricks
using CfgF32X64 = FwdCfg<64, 16, 32, 1024, false>;
using CfgF32Z64 = FwdCfg<16, 32, 64, 1024, false>;
#endif

}  // namespace

namespace ddrr_brick {

// Heaviest bricks first (brick_weight_kernel, brick_order_kernel): fills q.order from the launch's
// order workspace.  Not with fewer bricks than workgroups (nothing to order) or a handful of
// poses (the launch is latency-bound and the two small kernels cost more than the order gains).
bool order_bricks(BrickArgs &q, int BX, int BY, int BZ, int nby, int nbz, int n_bricks, int slots,
                  hipStream_t st, bool zero_counter, int min_poses) {
    if (!(q.order_ws && n_bricks <= q.order_cap && n_bricks > slots && !q.order && q.B >= min_poses)) return false;
    float *weight = reinterpret_cast<float *>(q.order_ws);
    int *order = q.order_ws + q.order_cap;
    hipLaunchKernelGGL(brick_weight_kernel, dim3((n_bricks + 3) / 4), dim3(256), 0, st, q, BX, BY,
                       BZ, nby, nbz, n_bricks, weight);
    hipLaunchKernelGGL(brick_order_kernel, dim3(1), dim3(1024), 0, st, weight, n_bricks, order,
                       zero_counter ? q.work : nullptr);
    q.order = order;
    return true;
}

// variant: DDRR_BRICKS_F32 (0) or DDRR_BRICKS_Q16 (1); tools builds know more (g_brick_variant)
// bytes of the caller's brick workspace: the (min, max) pairs, 2 floats per 32^3 brick (any brick
// grid fits), then, for the packed storage, the LDS images of the 32 x 32 x 64 bricks
// layout: [header: 64 words, word 0 = bricks on the fp32 path, word 1 = bricks, word 2 = launches
//          that found the volume changed under the workspace]
//         [(min, max) per brick: room for every 32^3 brick] [fallback flag per brick: int]
//         [fingerprint of the volume the workspace was built from: kFingerprintWords words]
//         [packed storage: the 16-bit bricks' LDS images]
constexpr long kWsHeaderBytes = 256;
static long n32_bricks(int dx, int dy, int dz) {
    return (long)((dx + 31) / 32) * ((dy + 31) / 32) * ((dz + 31) / 32);
}
constexpr long kFingerprintBytes = kFingerprintWords * (long)sizeof(unsigned);
static long ranges_bytes(int dx, int dy, int dz) {
    return kWsHeaderBytes + (n32_bricks(dx, dy, dz) * 3 * (long)sizeof(float) + 255) / 256 * 256 + kFingerprintBytes;
}

long brick_workspace_bytes(int dx, int dy, int dz, int brick_storage) {
    if (brick_storage == DDRR_BRICKS_F32) return 0;
    long n = ranges_bytes(dx, dy, dz);
    if (brick_storage == DDRR_BRICKS_Q16_PACKED)
        n += (long)((dx + CfgQ16Z64::BX - 1) / CfgQ16Z64::BX) * ((dy + CfgQ16Z64::BY - 1) / CfgQ16Z64::BY) *
             ((dz + CfgQ16Z64::BZ - 1) / CfgQ16Z64::BZ) * CfgQ16Z64::BRICK_BYTES;
    return n;
}

int launch_fwd_bricks(int variant, int packed, float *brick_ranges, int ranges_valid,
                      const float *volume, int dx, int dy, int dz, const float *source, const float *target,
                      const float *img, int B, int det_h, int det_w, float voxel_shift, float eps,
                      float *out, float *aux, float rec_q, hipStream_t st, void *launch_ws,
                      const char *who, float *clear, long clear_n, const unsigned *pix_mask) {
    const int N = det_h * det_w;
    // (brick_range_kernel: 16-byte loads where the volume's rows are aligned, else dwords)
    const bool vec_ok = (dz & 3) == 0 && (reinterpret_cast<uintptr_t>(volume) & 15) == 0;
    // the staging reads quads of four voxels from dword-aligned addresses (quad_load)
    // (brick_core.h quads_serve: at least four slices; thinner volumes take the general kernel)
    const bool quads_ok = dz >= 4 && (long)dx * dy * dz >= 4 && (reinterpret_cast<uintptr_t>(volume) & 3) == 0;
    // fp32 bricks at a handful of poses: the general kernel (no pooled end, no hand-out order --
    // neither pays there): 7-12 % ahead at one pose, within 2 % from 8 on (profiles/r04/
    // fp32_bricks_general_vs_configurable.txt)
    bool few_f32 = variant == DDRR_BRICKS_F32 && B < 8;
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
    if (g_brick_variant != -2) {  // (-2: no override, -1: bricks.hip)
        static int last_variant = -100;
        variant = g_brick_variant;
        if (variant != last_variant) ranges_valid = 0;  // (the variants differ in their brick grids)
        last_variant = variant;
        few_f32 = false;  // (the variant asked for)
    }
#endif
    if (!quads_ok || variant < 0 || few_f32) {
        if (clear && clear_n > 0 && hipMemsetAsync(clear, 0, sizeof(float) * (size_t)clear_n, st) != hipSuccess)
            return fail(-1, "hipMemsetAsync");
        return launch_bricks(aux ? BRICK_FWD_AUX : BRICK_FWD, volume, dx, dy, dz, source, target, img,
                             nullptr, B, det_h, det_w, voxel_shift, eps, out, aux, nullptr, st,
                             launch_ws, who, 0, nullptr, nullptr, rec_q, nullptr, 0, pix_mask);
    }
    BrickArgs p = {};
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
    p.vec = vec_ok ? 1 : 0;
    p.pix_mask = pix_mask;
    if ((long)B * N * 12 >= (1L << 32))
        return fail(-1, "B * N too large for one brick launch (12 B N must stay below 2^32): "
                        "split the pose batch");
    p.aux_plane = (unsigned)((long)B * N);
    p.rec_q = rec_q;
    p.ws_header = reinterpret_cast<int *>(brick_ranges);
    p.ranges = brick_ranges ? brick_ranges + kWsHeaderBytes / sizeof(float) : nullptr;
    p.fallback = brick_ranges ? reinterpret_cast<const int *>(p.ranges + 2 * n32_bricks(dx, dy, dz)) : nullptr;
    p.ranges_valid = ranges_valid;
    p.packed = packed && brick_ranges
                   ? reinterpret_cast<const unsigned char *>(brick_ranges) + ranges_bytes(dx, dy, dz)
                   : nullptr;
    p.fingerprint = brick_ranges ? reinterpret_cast<const unsigned *>(
                                       reinterpret_cast<const unsigned char *>(brick_ranges) +
                                       ranges_bytes(dx, dy, dz) - kFingerprintBytes)
                                 : nullptr;
    p.order = nullptr;
    p.clear = clear;
    p.clear_n = clear_n;
    p.split_t = 0;
    p.split_s = 1;
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
    p.brick_times = g_brick_times;
    p.order = g_brick_order;
    p.split_t = g_brick_split_t;
    p.split_s = g_brick_split_s;
#endif
    p.pix_bits = 1;
    while ((1L << p.pix_bits) < N) ++p.pix_bits;
    if (((long)B << p.pix_bits) > (1L << 32))
        return fail(-1, "B * 2^ceil(log2 N) exceeds 2^32: split the pose batch");
    p.t1 = g_brick_t1;
    p.t2 = g_brick_t2;
    p.dbg = g_brick_dbg;
#if defined(DDRR_BRICK_PROFILE)
    p.prof = g_brick_prof;
#endif
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
    if (variant >= 16) p.t1 = g_brick_sq_width;  // shared rings: t1 = class width
#endif
    int n_cu = 0;
    if (int rc = brick_launch_resources(st, launch_ws, dx, dy, dz, n_cu, p.work, &p.order_ws, &p.order_cap,
                                        /*zero_work=*/false))
        return rc;
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
    if (g_brick_dbg & 512) p.order_ws = nullptr;  // (bricks in id order)
#endif
    int rc = 0;
#define DDRR_LAUNCH_P(C, PRE_)                                                               \
    (p.pix_mask ? (aux ? launch_cfg<true, C, PRE_, true>(p, n_cu, out, aux, st)             \
                       : launch_cfg<false, C, PRE_, true>(p, n_cu, out, aux, st))           \
                : (aux ? launch_cfg<true, C, PRE_, false>(p, n_cu, out, aux, st)            \
                       : launch_cfg<false, C, PRE_, false>(p, n_cu, out, aux, st)))
#define DDRR_LAUNCH(C) DDRR_LAUNCH_P(C, false)
    // (length-class thresholds: flat within 1.5 % around these, profiles/r03)
    if (variant == DDRR_BRICKS_Q16 || variant == 5) {
        p.t1 = g_brick_t1 * (22.f / 18.f);
        p.t2 = g_brick_t2 * (48.f / 40.f);
    }
    switch (variant) {
        case DDRR_BRICKS_F32: rc = DDRR_LAUNCH(CfgF32); break;
        case DDRR_BRICKS_Q16:
            // (a few poses: the instantiation whose first claim hides the fingerprint's round trip)
            if (B <= 8)
                rc = DDRR_LAUNCH_P(CfgQ16Z64, true);
            else
                rc = DDRR_LAUNCH(CfgQ16Z64);
            break;
#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
        case 2: rc = DDRR_LAUNCH(CfgQ16x1); break;
        case 10: rc = DDRR_LAUNCH(CfgQ16x2); break;
        case 3: rc = DDRR_LAUNCH(CfgF32Half); break;
        case 4: rc = DDRR_LAUNCH(CfgQ16X64); break;
        case 5: rc = DDRR_LAUNCH(CfgQ16Z64); break;  // (= DDRR_BRICKS_Q16)
        case 6: rc = DDRR_LAUNCH(CfgQ16Y64); break;
        case 7: rc = DDRR_LAUNCH(CfgF32Y64); break;
        case 8: rc = DDRR_LAUNCH(CfgF32X64); break;
        case 9: rc = DDRR_LAUNCH(CfgF32Z64); break;
        // workgroup-shared length-class rings: 16 + v: 12 classes, 32 + v: 8 classes
#define DDRR_LAUNCH_SQ(C, K) (aux ? launch_sq<true, C, K>(p, n_cu, out, aux, st) \
                                  : launch_sq<false, C, K>(p, n_cu, out, aux, st))
        case 16: rc = DDRR_LAUNCH_SQ(CfgF32, 12); break;
        case 18: rc = DDRR_LAUNCH_SQ(CfgQ16x1, 12); break;
        case 20: rc = DDRR_LAUNCH_SQ(CfgQ16X64, 12); break;
        case 21: rc = DDRR_LAUNCH_SQ(CfgQ16Z64, 12); break;
        case 32: rc = DDRR_LAUNCH_SQ(CfgF32, 8); break;
        case 36: rc = DDRR_LAUNCH_SQ(CfgQ16X64, 8); break;
#undef DDRR_LAUNCH_SQ
#endif
        default: return fail(-1, "unknown brick variant");
    }
#undef DDRR_LAUNCH
#undef DDRR_LAUNCH_P
    if (rc) return rc;
    return finish(who);
}

}  // namespace ddrr_brick

#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
extern "C" int ddrr_set_brick_variant(int v) {
    ddrr_brick::g_brick_variant = v;
    return 0;
}
extern "C" int ddrr_set_brick_times(unsigned *device_times) {
    ddrr_brick::g_brick_times = device_times;
    return 0;
}
extern "C" int ddrr_set_brick_order(const int *device_order) {
    ddrr_brick::g_brick_order = device_order;
    return 0;
}
extern "C" int ddrr_set_brick_split(int t, int s) {
    ddrr_brick::g_brick_split_t = t;
    ddrr_brick::g_brick_split_s = s < 1 ? 1 : s;
    return 0;
}
extern "C" int ddrr_set_brick_sq_width(float w) {
    ddrr_brick::g_brick_sq_width = w;
    return 0;
}
#endif
