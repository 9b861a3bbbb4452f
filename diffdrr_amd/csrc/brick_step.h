// brick_step.h -- the per-(ray, brick) clip and walk of the volume-stationary Siddon kernel,
// written for how gfx950 issues vector instructions.
//
// Measured on MI355X (tools/ubench/valu_rates2.hip, 4 waves per SIMD): a wave64 v_add / v_sub /
// v_mul / v_fma / v_fmac _f32 (also with neg / abs / clamp modifiers, inline constants, literals
// and denormal operands), v_mov, v_and / or / xor and v_add / sub _u32 issue in 2 cycles; v_min /
// max / min3 / med3, every v_cmp and v_cndmask, the conversions, v_floor, shifts, DPP moves, the
// packed f32 forms AND any of the fast ones with an SGPR operand take 4.  So the walk below
//   * selects with arithmetic instead of compare + cndmask: t = clamp(live - (a - m) * 2^126) is
//     1 exactly where a == m (one v_sub + one v_fma with the clamp modifier);
//   * forms the voxel's LDS address without a conversion: strides and base are handed over as
//     the DENORMAL floats whose bit patterns are the integers (n * 2^-149), the address is an
//     FMA chain over the plane counters and its bit pattern is the byte address;
//   * keeps every loop operand in a VGPR, wave-uniform ones included;
//   * has no per-lane exit: a lane whose ray has left the brick idles on its last voxel with
//     zero-length steps (live = 0), the wave leaves when no lane is live.
// One step = v_min3 + 17 fast instructions = 38 issue cycles (52 with the backward record),
// against 58+ for the compare / select / convert formulation it replaces.
//
// Exactness: see step_enter (alpha of every plane a ray reaches inside a brick is within an ulp
// of the reference's quotient (k - shift - s) / (t - s + eps), diffdrr/renderers.py:97-106).
#pragma once

#include <string.h>

#include "brick_core.h"
#include "brick_walk.h"
#include "ddrr_common.h"
#include "siddon_core.h"

namespace ddrr {

// --- the three non-standard operations, with their host restatements (tests/emu) ---

// clamp(d * nbig + live) to [0, 1]: with d >= 0, nbig = -2^126: `live` where d == 0, else 0.
DDRR_HD float sel_zero(float d, float nbig, float live) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(d), "v"(nbig), "v"(live));
    return r;
#else
    const float r = fmaf(d, nbig, live);
    return r > 0.f ? (r < 1.f ? r : 1.f) : 0.f;  // NaN -> 0, like the hardware clamp
#endif
}

// The float whose bit pattern is the integer n (n * 2^-149 for n < 2^23) and back.
DDRR_HD float bits_as_float(unsigned n) {
    float f;
    memcpy(&f, &n, 4);
    return f;
}
DDRR_HD unsigned float_bits(float f) {
    unsigned n;
    memcpy(&n, &f, 4);
    return n;
}

// Make the compiler hold a (possibly wave-uniform) value in a vector register from here on:
// an SGPR operand halves the issue rate of the instruction that reads it.
DDRR_HD float in_vgpr(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

constexpr float kSelBig = 0x1p126f;

// Geometry of the staged brick for the walk: plane range per axis and the LDS byte strides
// as bit-pattern floats (see above).
struct StepGeom {
    float lof[3], hif[3];  // first / last plane index of the brick per axis
    float strideb[3];      // bits_as_float(byte stride) per axis
};

DDRR_HD StepGeom step_geom(const Box &box, const BrickLayout &lay) {
    StepGeom G;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        G.lof[a] = (float)box.lo[a];
        G.hif[a] = (float)box.hi[a];
    }
    G.strideb[0] = bits_as_float((unsigned)lay.sx * 4u);
    G.strideb[1] = bits_as_float((unsigned)lay.sy * 4u);
    G.strideb[2] = bits_as_float(4u);
    return G;
}

// State of a ray at its entry into a brick.
struct StepEntry {
    float inv[3], a0[3];  // alpha_a(k) = fma(k - k0_a, inv_a, a0_a), k0_a: first plane ahead
    float an[3];          // alpha of the next plane ahead (= a0 at entry: k - k0 = 0)
    float dirf[3];        // +-1: how k - k0 moves
    float ent[3];         // one-hot: the axis of the entry crossing (x before y before z)
    float entry, exit;
    float lbig;           // 2^(125 - exponent of the largest |alpha| in the brick), see step_walk
    float offc;           // bit-pattern float: address = bits(offc + sum_a (k_a - k0_a) strideb_a)
    float nx[3];          // planes from the first plane ahead to the exit face of the axis (>= 0)
    bool hit;
};

// `base_bits`: what the accessor wants added to a byte offset inside the brick (the brick's
// absolute LDS address on the device, 0 for a pointer-relative fetch on the host).
//
// Conditioning.  alpha of plane k is the reference's quotient n_k / d, n_k = (k - shift) - s,
// d = (t - s) + eps (renderers.py:97-106).  For a ray that glides along a plane of axis a
// (|d_a| ~ 1e-5 |d|) n_k is tiny and exact for the plane next to the ray and n_k / d is
// accurate, while any "base + k / d" form with a far base plane cancels two numbers of
// size |k / d| >> alpha.  So the base plane of each axis is the FIRST plane ahead of the entry
// point, with its alpha by a correctly rounded division; plane k further on is
// fma(k - k0, 1/d, alpha_0), off by <= |alpha_k - alpha_0| 2^-23, i.e. by a fraction of an
// ulp of alpha for every plane the ray can reach inside the brick.  Which cell the ray enters
// is decided in the same terms, exactly: alpha_k >= entry  <=>  sign(d) (entry d - n_k) <= 0,
// one FMA whose sign is that of the real number.
// The exit alpha is the walk's own value for the exit face it reaches first (the walk
// ends exactly there); the neighbouring brick enters at the face's quotient, at most
// |chord| 2^-23 away: the pieces of a ray meet to ~1e-9 of its length.
DDRR_HD StepEntry step_enter(const StepGeom &G, const float s[3], const float t[3], float shift,
                             float eps, unsigned base_bits) {
    StepEntry E;
    float d[3], mn[3];
    E.entry = -INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;  // renderers.py:104-106
        E.inv[a] = fast_rcp(d[a]);
        const float kn = d[a] > 0.f ? G.lof[a] : G.hif[a];  // the face the ray enters through
        mn[a] = div_refined((kn - shift) - s[a], d[a], E.inv[a]);
        E.entry = fmaxf(E.entry, mn[a]);
    }
    E.exit = INFINITY;
    E.offc = bits_as_float(base_bits);
    // the axis of the entry crossing, exclusive (ties: x before y before z)
    E.ent[0] = mn[0] == E.entry ? 1.f : 0.f;
    E.ent[1] = (mn[1] == E.entry && mn[0] != E.entry) ? 1.f : 0.f;
    E.ent[2] = 1.f - E.ent[0] - E.ent[1];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const bool pos = d[a] > 0.f;
        const float p01 = pos ? 1.f : 0.f;
        E.dirf[a] = pos ? 1.f : -1.f;
        const float cmax = G.hif[a] - 1.f;
        // cell of the entry point from its position, then the alpha-order rule in exact terms:
        // the plane ahead must not lie behind the entry alpha, the plane behind not ahead of it
        float u = med3f(floorf(fmaf(E.entry, d[a], s[a] + shift)), G.lof[a], cmax);
        const float ra = fmaf(E.entry, d[a], -(((u + p01) - shift) - s[a])) * E.dirf[a];
        const float rb = fmaf(E.entry, d[a], -(((u + (1.f - p01)) - shift) - s[a])) * E.dirf[a];
        const float adj = (ra > 0.f ? E.dirf[a] : 0.f) - (rb < 0.f ? E.dirf[a] : 0.f);
        u = med3f(u + adj, G.lof[a], cmax);
        u = (mn[a] == E.entry) ? (pos ? G.lof[a] : cmax) : u;  // entering axis: the face cell
        const float k0 = u + p01;
        E.a0[a] = div_refined((k0 - shift) - s[a], d[a], E.inv[a]);
        E.an[a] = E.a0[a];
        // the exit face of this axis, as the walk will see it
        const float kx = pos ? G.hif[a] : G.lof[a];
        E.exit = fminf(E.exit, fmaf(kx - k0, E.inv[a], E.a0[a]));
        E.nx[a] = (kx - k0) * E.dirf[a];
        // address = base + sum_a (cell_a - lo_a) stride_a, cell_a = k_a - p01_a
        E.offc = fmaf((k0 - p01) - G.lof[a], G.strideb[a], E.offc);
    }
    E.hit = E.entry < E.exit;  // false for NaN
    // scale of the "is this the last segment" test, per ray: the whole line through source and
    // target is integrated, so alpha may be of any magnitude; with m = max(|entry|, |exit|) in
    // [2^e, 2^(e+1)), lbig = 2^(125 - e) keeps exit * lbig finite and (exit - a) * lbig >= 1 for
    // every float a below exit
    // (exponent field of 2^(125 - e) is 379 - biased exponent of m; capped at 2^127 for tiny m)
    const float m = fmaxf(fabsf(E.entry), fabsf(E.exit));
    const unsigned field = 379u - ((float_bits(m) >> 23) & 0xffu);
    E.lbig = bits_as_float((field < 254u ? field : 254u) << 23);
    return E;
}

// The walk of one ray through one brick from its entry state E (E.hit must hold).
// `fetch(bits)` reads the LDS copy at the byte address `bits` (absolute on the device,
// relative to the brick on the host).  I = sum V dalpha over the brick; with AUX,
// rec = {S0x, S0z, S1x, S1z} of the brick-local backward record (voxels outside the brick
// count as 0 on both sides of a face, so the records of the bricks along a ray add up to the
// ray's; y follows from sum_a S0_a = 0, sum_a S1_a = I).
// MAXSTEPS: a bound on the crossings inside the brick (+ slack); exit_axis: NULL, or with AUX
// two floats receiving the one-hot (x, y; z implied) axis of the exit crossing.
template <bool AUX, int MAXSTEPS = 3 * BRICK + 4, class Fetch>
DDRR_HD int step_walk(const Fetch &fetch, const StepGeom &G, const StepEntry &E, float &I,
                      float rec[4], float *exit_axis = nullptr) {
    float kr0 = 0.f, kr1 = 0.f, kr2 = 0.f;  // k - k0 per axis
    float an0 = E.an[0], an1 = E.an[1], an2 = E.an[2];
    const float inv0 = E.inv[0], inv1 = E.inv[1], inv2 = E.inv[2];
    const float af0 = E.a0[0], af1 = E.a0[1], af2 = E.a0[2];
    const float dir0 = E.dirf[0], dir1 = E.dirf[1], dir2 = E.dirf[2];
    // wave-uniform loop operands: into vector registers once
    const float sb0 = in_vgpr(G.strideb[0]), sb1 = in_vgpr(G.strideb[1]);
    const float sb2 = in_vgpr(G.strideb[2]);
    const float nbig = in_vgpr(-kSelBig);
    const float exit = E.exit, offc = E.offc;
    // (pinned: left alone the compiler re-derives -lbig with a v_xor inside the loop)
    const float nlbig = in_vgpr(-E.lbig), exit_big = exit * E.lbig;
    float a_cur = E.entry, acc = 0.f;
    // Vc: the voxel of the segment being closed (requested one step earlier), Vp: the one before
    float Vc = fetch(float_bits(fmaf(kr0, sb0, fmaf(kr1, sb1, fmaf(kr2, sb2, offc)))));
    float Vp = 0.f;
    // the crossing that opened the segment being closed: its axis (x, y one-hot; z implied)
    float tpx = E.ent[0], tpy = E.ent[1];
    float S0x = 0.f, S0y = 0.f, S1x = 0.f, S1y = 0.f;
    float live = 1.f;

// One step: close the segment [a_cur, a_next), move the plane counters of the axes crossed at
// a_next, request the next voxel.  With the record, the crossing that OPENED the segment
// (alpha = a_cur, axes tp*, voxels Vp | Vc) is settled here too: its two voxels are both known.
#define DDRR_STEP()                                                                       \
    {                                                                                     \
        const float a_next = fminf(fminf(an0, an1), an2);                                 \
        const float len = a_next - a_cur;                                                 \
        live = sel_zero(a_next, nlbig, exit_big); /* clamp((exit - a_next) lbig): 0 on the */ \
        /* ray's last segment; both products are exact, the sum rounds once */            \
        const float t0 = sel_zero(an0 - a_next, nbig, live);                              \
        const float t1 = sel_zero(an1 - a_next, nbig, live);                              \
        const float t2 = sel_zero(an2 - a_next, nbig, live);                              \
        kr0 = fmaf(t0, dir0, kr0);                                                        \
        kr1 = fmaf(t1, dir1, kr1);                                                        \
        kr2 = fmaf(t2, dir2, kr2);                                                        \
        const float Vn =                                                                  \
            fetch(float_bits(fmaf(kr0, sb0, fmaf(kr1, sb1, fmaf(kr2, sb2, offc)))));      \
        an0 = fmaf(kr0, inv0, af0);                                                       \
        an1 = fmaf(kr1, inv1, af1);                                                       \
        an2 = fmaf(kr2, inv2, af2);                                                       \
        acc = fmaf(Vc, len, acc);                                                         \
        if (AUX) {                                                                        \
            const float dv = Vp - Vc, dva = dv * a_cur;                                   \
            S0x = fmaf(dv, tpx, S0x);                                                     \
            S0y = fmaf(dv, tpy, S0y);                                                     \
            S1x = fmaf(dva, tpx, S1x);                                                    \
            S1y = fmaf(dva, tpy, S1y);                                                    \
            tpx = t0;                                                                     \
            tpy = fmaf(-t0, t1, t1); /* exclusive: x before y */                          \
        }                                                                                 \
        a_cur = a_next;                                                                   \
        Vp = Vc;                                                                          \
        Vc = Vn;                                                                          \
    }
    // a brick holds < 3 * BRICK + 3 crossings; the wave leaves when no lane is live (a lane
    // that is done repeats zero-length steps on its last voxel: t* = 0, nothing moves)
    // (two pairs per trip: the values a pair hands to the next one -- alpha, two voxels, the
    // crossing's axis flags -- are still live when their successors are formed, so a single
    // pair per trip ends in five register copies; unrolled, the pairs swap registers instead.
    // The build also passes -fno-slp-vectorize: v_pk_*_f32 issue at half rate on gfx950, and
    // the packed forms cost register shuffles on top.  Together 1.39 -> 1.30 ms forward.)
    int it = 0;
#if defined(DDRR_WALK_CHECK4)
#pragma unroll 1
    for (; it < MAXSTEPS; it += 4) {
        DDRR_STEP()
        DDRR_STEP()
        DDRR_STEP()
        DDRR_STEP()
        if (!__builtin_amdgcn_ballot_w64(live != 0.f)) break;
    }
    if (false)
#endif
#pragma unroll 2
    for (; it < MAXSTEPS; it += 2) {
        DDRR_STEP()
        DDRR_STEP()
#if defined(__HIP_DEVICE_COMPILE__)
        if (!__builtin_amdgcn_ballot_w64(live != 0.f)) break;
#else
        if (live == 0.f) break;
#endif
    }
#undef DDRR_STEP
    I = acc;
    if (AUX) {
        // (after a lane's last live step its counters stand still and tp* = 0: idle steps add
        // nothing, and the crossing that opened the last segment was settled by that step)
        // the exit crossing: last voxel | 0 at alpha = exit, on the axis whose face it is
        const float ex = sel_zero(an0 - exit, nbig, 1.f);
        const float ey0 = sel_zero(an1 - exit, nbig, 1.f);
        const float ey = fmaf(-ex, ey0, ey0);
        const float va = Vc * exit;
        S0x = fmaf(Vc, ex, S0x);
        S0y = fmaf(Vc, ey, S0y);
        S1x = fmaf(va, ex, S1x);
        S1y = fmaf(va, ey, S1y);
        rec[0] = S0x;
        rec[1] = -(S0x + S0y);       // sum_a S0_a = 0
        rec[2] = S1x;
        rec[3] = acc - (S1x + S1y);  // sum_a S1_a = I
        if (exit_axis) {
            // (for the quantised bricks: they finish the record themselves, q16_finish)
            rec[1] = S0y;
            rec[3] = S1y;
            exit_axis[0] = ex;
            exit_axis[1] = ey;
        }
    }
    return it + 2;  // steps the wave took (profiling builds)
}

// The forward-ONLY walk (no backward record).  Without a record nothing depends on WHICH axis a
// near-tied pair of crossings is attributed to -- the integral is continuous across ties -- so
// the plane counters of step_walk (kept there so that every alpha is the reference's quotient
// within an ulp: the record's attribution and the pose gradients rest on it) can go: alpha of
// the next plane of a crossing axis is the accumulated alpha + |1 / d|, and the LDS address moves
// by the signed stride.  16 vector instructions per step instead of 19 (+ the exit check every
// second step).
// Accuracy.  The walk runs in alpha' = alpha - entry (exact differences of the entry state's
// correctly rounded quotients), i.e. in numbers of the size of the CHORD, 4-16x smaller than
// alpha itself: every accumulation rounds at <= 1/2 ulp(chord), so after the <= 64 crossings of
// an axis inside a brick the drift stays below the 1/2 ulp(alpha) that the first plane's quotient
// carries anyway -- the segment lengths are as good as step_walk's.
// The exit.  exit' = min_a fma(n_a, |1 / d_a|, a0'_a), n_a planes to the exit face; the accumulated
// alpha' of that face differs from it by <= 64 x 1/2 ulp(exit') <= 2^-18 exit'.  The walk stops at
// the first crossing not below exit' (1 - 2^-17): the real exit crossing always stops it; a
// crossing that happens to lie inside that margin stops it early and drops a sliver of
// < 2^-17 of the chord.  a_end: alpha' the walk stopped at = the sum of the segment lengths.
template <int MAXSTEPS = 3 * BRICK + 4, class Fetch>
DDRR_HD int step_walk_fwd(const Fetch &fetch, const StepGeom &G, const StepEntry &E, float &I,
                          float &a_end) {
    // |1 / d| and the signed byte strides (bit-pattern floats: -n 2^-149 subtracts n exactly)
    const float st0 = E.inv[0] * E.dirf[0], st1 = E.inv[1] * E.dirf[1], st2 = E.inv[2] * E.dirf[2];
    float an0 = E.an[0] - E.entry, an1 = E.an[1] - E.entry, an2 = E.an[2] - E.entry;
    const float exit = fminf(fminf(fmaf(E.nx[0], st0, an0), fmaf(E.nx[1], st1, an1)),
                             fmaf(E.nx[2], st2, an2));
    const float sb0 = in_vgpr(G.strideb[0]) * E.dirf[0], sb1 = in_vgpr(G.strideb[1]) * E.dirf[1];
    const float sb2 = in_vgpr(G.strideb[2]) * E.dirf[2];
    const float nbig = in_vgpr(-kSelBig);
    // lbig = 2^(125 - exponent of exit'): (thr - a) lbig >= 1 for every float a below thr
    const unsigned field = 379u - ((float_bits(exit) >> 23) & 0xffu);
    const float lbig = bits_as_float((field < 254u ? field : 254u) << 23);
    const float thr = fmaf(-exit, 0x1p-17f, exit);
    const float nlbig = in_vgpr(-lbig), thr_big = thr * lbig;
    float addr = E.offc, a_cur = 0.f, acc = 0.f, live = 1.f;
    float Vc = fetch(float_bits(addr));
#define DDRR_STEP()                                                                       \
    {                                                                                     \
        const float a_next = fminf(fminf(an0, an1), an2);                                 \
        const float len = a_next - a_cur;                                                 \
        live = sel_zero(a_next, nlbig, thr_big); /* clamp((thr - a_next) lbig) */         \
        const float t0 = sel_zero(an0 - a_next, nbig, live);                              \
        const float t1 = sel_zero(an1 - a_next, nbig, live);                              \
        const float t2 = sel_zero(an2 - a_next, nbig, live);                              \
        addr = fmaf(t0, sb0, fmaf(t1, sb1, fmaf(t2, sb2, addr)));                         \
        const float Vn = fetch(float_bits(addr));                                         \
        an0 = fmaf(t0, st0, an0);                                                         \
        an1 = fmaf(t1, st1, an1);                                                         \
        an2 = fmaf(t2, st2, an2);                                                         \
        acc = fmaf(Vc, len, acc);                                                         \
        a_cur = a_next;                                                                   \
        Vc = Vn;                                                                          \
    }
    int it = 0;
#pragma unroll 2
    for (; it < MAXSTEPS; it += 2) {
        DDRR_STEP()
        DDRR_STEP()
#if defined(__HIP_DEVICE_COMPILE__)
        if (!__builtin_amdgcn_ballot_w64(live != 0.f)) break;
#else
        if (live == 0.f) break;
#endif
    }
#undef DDRR_STEP
    I = acc;
    a_end = a_cur;
    return it + 2;
}

// ---------------------------------------------------------------- 16-bit quantised bricks
// A brick staged as 16-bit block-quantised voxels (one (vmin, step) pair per brick,
// V ~ vmin + q step, q = 0 .. 65535) takes half the LDS of the fp32 brick, so two workgroups
// fit a CU and one's staging and end-of-brick barrier overlap the other's walk.  The walk is
// step_walk itself, unchanged and at no extra instruction: `ds_read_u16` leaves q in the low
// half of the register, which read as a float is the DENORMAL q 2^-149 -- gfx950 multiplies
// denormals at full rate -- and every alpha of the entry state is pre-scaled by 2^64
// (q16_scale_entry), so that the products q 2^-149 * len 2^64 = q len 2^-85 are normal numbers
// accumulated without loss.  q16_finish turns the sums over q back into sums over V.
constexpr float kQ16AlphaScale = 0x1p64f;  // K
constexpr float kQ16Unscale = 0x1p85f;     // 2^149 / K

struct Q16Range {
    float vmin, step;      // V ~ vmin + q * step
    float inv_step;        // 65535 / (vmax - vmin), 0 for a constant brick
};

DDRR_HD Q16Range q16_range(float vmin, float vmax) {
    Q16Range r;
    const float range = vmax - vmin;
    r.vmin = vmin;
    // a brick holding inf / nan poisons the rays through it (as the values themselves would)
    r.step = range * (1.0f / 65535.0f);
    r.inv_step = range > 0.f ? 65535.0f / range : 0.f;
    if (!(range >= 0.f) || !(range < INFINITY)) {
        r.vmin = NAN;
        r.step = NAN;
        r.inv_step = 0.f;
    }
    return r;
}

// Which bricks may be quantised.  The error of a stored voxel is at most range / 131070 in
// ABSOLUTE terms, whatever the voxel's own size: one bright voxel (a metal marker, contrast,
// un-normalised HU) sets the step for the 65 535 others of its brick, and rays that see only
// the others carry that error against a small integral.  So a brick is quantised only if its
// range is at most kQ16RangeOverLevel times its LEVEL -- the smallest mean |V| of any of its
// 4 x 4 x 4 blocks (brick_range_kernel; over the non-zero voxels when the minimum is 0, which
// q = 0 stores exactly): every voxel's error is then <= 12 / 131070 = 9.2e-5 of the mean of the
// dimmest block of its brick, even if all rounding errors along a ray had one sign: a pixel is
// within 9.2e-5 of the line integral of the brick LEVELS along its ray (+ 3e-6 of fp32 arithmetic).
// That is a bound relative to the levels, NOT to the pixel's own value: a ray that
// crosses mostly air (stored exactly, but counted in no level) and a sliver of skin inside
// tissue-level bricks carries the tissue's step against a small integral.  Measured on noisy
// CT-like volumes, fp64, quantisation alone (tests/test_brick_storage_guard.py): image-normalised
// <= 2.2e-6 (the gate: 1e-4); relative to the pixel's own value <= 1.6e-4 at pixels above 1e-3 of
// the image's maximum, where the reference's own fp32 arithmetic is 1.5e-4 ... 2.5e-4 off.
// (Rounds 4 and early 5 had 8 = 6.1e-5: it sent 644 instead of 369 of the 512^3 phantom's 2048
// bricks to the fp32 path.)  Any
// other brick -- and any brick holding inf / NaN or a range outside what the 2^64 pre-scaling
// of the walk carries -- is rendered from the volume's own fp32 values (bricks_fwd.hip MIXED).
constexpr float kQ16RangeOverLevel = 12.0f;
DDRR_HD bool q16_usable(float vmin, float vmax, float level) {
    const float range = vmax - vmin;
    if (range == 0.f) return fabsf(vmin) < 0x1p40f;  // a brick of one value: exact
    return range >= 0x1p-60f && range < 0x1p40f && fabsf(vmin) < 0x1p40f &&
           range <= kQ16RangeOverLevel * level;     // (false for NaN)
}

// q of a voxel: round to nearest by adding 2^23 (the sum's low 16 bits are the integer)
DDRR_HD unsigned q16_encode(float v, const Q16Range &r) {
    const float f = fmaf(v - r.vmin, r.inv_step, 8388608.0f);
    return float_bits(f) & 0xffffu;  // (nan -> some value: the range already poisons the brick)
}

// Scale every alpha of an entry state by K = 2^64 (exact: powers of two, no alpha of a ray
// that meets the volume is within 2^63 of the fp32 range).
DDRR_HD void q16_scale_entry(StepEntry &E) {
    const float K = kQ16AlphaScale;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        E.inv[a] *= K;
        E.a0[a] *= K;
        E.an[a] *= K;
    }
    E.entry *= K;
    E.exit *= K;
    const float m = fmaxf(fabsf(E.entry), fabsf(E.exit));
    const unsigned field = 379u - ((float_bits(m) >> 23) & 0xffu);
    E.lbig = bits_as_float((field < 254u ? field : 254u) << 23);
}

// Forward only, after step_walk_fwd: the chord is what the walk summed, entry .. a_end.
DDRR_HD float q16_finish_fwd(const Q16Range &r, const StepEntry &E, float acc, float a_end) {
    const float iK = 1.0f / kQ16AlphaScale;
    (void)E;
    return fmaf(acc, r.step * kQ16Unscale, r.vmin * (a_end * iK));
}

// Fetch of a 16-bit voxel relative to the brick (host emulation; the device fetches by absolute
// LDS address, bricks_fwd.hip LdsAbsFetch16).
struct LdsFetch16 {
    const unsigned short *brick;
    DDRR_HD float operator()(unsigned boff) const { return bits_as_float(brick[boff >> 1]); }
};

// From the walk over q (step_walk on a scaled entry state; rec = {S0x, S0y, S1x, S1y} of q,
// exit_axis from the walk) to the brick-local integral and record of V = vmin + q step inside
// the brick, 0 outside: interior crossings see differences (vmin cancels), the entry and exit
// crossings see the values themselves.  rec <- {S0x, S0z, S1x, S1z}.
template <bool AUX>
DDRR_HD void q16_finish(const Q16Range &r, const StepEntry &E, const float exit_axis[2], float &I,
                        float rec[4]) {
    const float iK = 1.0f / kQ16AlphaScale;
    const float cs = r.step * kQ16Unscale;             // sums of q 2^-149 alpha K -> V alpha
    const float entry = E.entry * iK, exit = E.exit * iK;
    I = fmaf(I, cs, r.vmin * (exit - entry));
    if (AUX) {
        const float K = kQ16AlphaScale;
        const float s0x = fmaf(rec[0] * K, cs, r.vmin * (exit_axis[0] - E.ent[0]));
        const float s0y = fmaf(rec[1] * K, cs, r.vmin * (exit_axis[1] - E.ent[1]));
        const float s1x = fmaf(rec[2], cs, r.vmin * fmaf(exit, exit_axis[0], -entry * E.ent[0]));
        const float s1y = fmaf(rec[3], cs, r.vmin * fmaf(exit, exit_axis[1], -entry * E.ent[1]));
        rec[0] = s0x;
        rec[1] = -(s0x + s0y);     // sum_a S0_a = 0
        rec[2] = s1x;
        rec[3] = I - (s1x + s1y);  // sum_a S1_a = I
    }
}

// mask_to_channels on the bricks (reference renderers.py:77-89).  The staged word of a voxel
// holds its value rounded to a 16-bit mantissa in the upper 24 bits and its label in the low
// 8: one LDS read per step serves both, and the brick needs no second plane in LDS (there
// is no room for one: 132 KiB of 160 are the values).  Rounding to nearest at 2^-17
// relative per voxel is of the size of fp32 accumulation error over a ray (measured:
// channel sum vs plain render 6e-6 of the image scale; the parity tolerance is 1e-4).
DDRR_HD float pack_voxel_label(float v, unsigned lab) {
    const unsigned b = float_bits(v);
    const bool special = (b & 0x7f800000u) == 0x7f800000u;  // inf / nan: keep the class
    const unsigned r = special ? (b | ((b & 0x007fffffu) ? 0x00400000u : 0u)) : b + 0x80u;
    return bits_as_float((r & 0xffffff00u) | (lab & 0xffu));
}

// Siddon channels: a label the caller has no channel for (>= n_channels) is staged as the value
// 0 under label 0 -- its segments add nothing, and the flush needs no range check.  (Not for the
// marcher: there a voxel's value also feeds its neighbours' interpolation.)
DDRR_HD float pack_voxel_label_below(float v, unsigned lab, unsigned n_channels) {
    lab &= 0xffu;
    return lab < n_channels ? pack_voxel_label(v, lab) : 0.f;
}

// The walk of step_walk_fwd over packed words: the integral of a run of voxels with one label is
// summed in a register and handed to `flush(label, sum)` when the label changes and at the end.
// (A lane that is done idles on its last voxel: same label, zero length.)  `scale` multiplies
// every segment length -- the ray's length L, folded into the chord-relative alphas at the entry
// (the selects only compare alphas; the lengths are their differences), so that a flush is the
// address and the atomic alone.  16 + 3 vector instructions per step; the flush -- a divergent
// block whenever ANY lane of the wave changes label -- adds 4.
// A flush functor that can hand a run over UNDER AN EXEC MASK inside the step -- no branch around a
// divergent block whenever any lane of the wave changes label -- says so with a member
// `masked(lab, cur, run) -> run'` (device only; bricks.hip BrickColumnFlush).
template <class F, class = void>
struct has_masked_flush { static constexpr bool value = false; };
template <class F>
struct has_masked_flush<F, decltype(void(F::kMaskedFlush))> { static constexpr bool value = F::kMaskedFlush; };

template <class Fetch, class Flush>
DDRR_HD void step_walk_channels(const Fetch &fetch, const StepGeom &G, const StepEntry &E,
                                const Flush &flush, float scale = 1.f) {
    const float st0 = E.inv[0] * E.dirf[0] * scale, st1 = E.inv[1] * E.dirf[1] * scale;
    const float st2 = E.inv[2] * E.dirf[2] * scale;
    float an0 = (E.an[0] - E.entry) * scale, an1 = (E.an[1] - E.entry) * scale;
    float an2 = (E.an[2] - E.entry) * scale;
    const float exit = fminf(fminf(fmaf(E.nx[0], st0, an0), fmaf(E.nx[1], st1, an1)),
                             fmaf(E.nx[2], st2, an2));
    const float sb0 = in_vgpr(G.strideb[0]) * E.dirf[0], sb1 = in_vgpr(G.strideb[1]) * E.dirf[1];
    const float sb2 = in_vgpr(G.strideb[2]) * E.dirf[2];
    const float nbig = in_vgpr(-kSelBig);
    const unsigned field = 379u - ((float_bits(exit) >> 23) & 0xffu);
    const float lbig = bits_as_float((field < 254u ? field : 254u) << 23);
    const float thr = fmaf(-exit, 0x1p-17f, exit);
    const float nlbig = in_vgpr(-lbig), thr_big = thr * lbig;
    float addr = E.offc, a_cur = 0.f, run = 0.f, live = 1.f;
    float Vc = fetch(float_bits(addr));
    unsigned cur = float_bits(Vc) & 0xffu;
#define DDRR_STEP()                                                                       \
    {                                                                                     \
        const float a_next = fminf(fminf(an0, an1), an2);                                 \
        const float len = a_next - a_cur;                                                 \
        live = sel_zero(a_next, nlbig, thr_big);                                          \
        const float t0 = sel_zero(an0 - a_next, nbig, live);                              \
        const float t1 = sel_zero(an1 - a_next, nbig, live);                              \
        const float t2 = sel_zero(an2 - a_next, nbig, live);                              \
        addr = fmaf(t0, sb0, fmaf(t1, sb1, fmaf(t2, sb2, addr)));                         \
        const float Vn = fetch(float_bits(addr));                                         \
        an0 = fmaf(t0, st0, an0);                                                         \
        an1 = fmaf(t1, st1, an1);                                                         \
        an2 = fmaf(t2, st2, an2);                                                         \
        const unsigned w = float_bits(Vc), lab = w & 0xffu;                               \
        if constexpr (has_masked_flush<Flush>::value) {                                   \
            run = flush.masked(lab, cur, run);                                            \
            cur = lab;                                                                    \
        } else if (lab != cur) {                                                          \
            flush(cur, run);                                                              \
            run = 0.f;                                                                    \
            cur = lab;                                                                    \
        }                                                                                 \
        run = fmaf(bits_as_float(w & 0xffffff00u), len, run);                             \
        a_cur = a_next;                                                                   \
        Vc = Vn;                                                                          \
    }
    for (int it = 0; it < 3 * BRICK + 4; it += 2) {
        DDRR_STEP()
        DDRR_STEP()
#if defined(__HIP_DEVICE_COMPILE__)
        if (!__builtin_amdgcn_ballot_w64(live != 0.f)) break;
#else
        if (live == 0.f) break;
#endif
    }
#undef DDRR_STEP
    flush(cur, run);
}

// Backward of the channel render w.r.t. the ray (reference: autograd of renderers.py:77-89): the
// brick-local record of step_walk<true> for the volume W(x) = V(x) g[label(x)], g = the incoming
// gradient of the ray's own output column -- d/d source, d/d target and d/d img of sum_c g_c I_c
// follow from it by the single-channel formulas (ddrr_siddon_backward_rays with grad_out = 1).
// Packed words as in step_walk_channels (labels without a channel staged as value 0); the weight
// of the current label is kept in a register and fetched by `weight(label)` -- a global gather --
// when the label of the voxel entering the walk changes: a divergent block, and the one place
// where the walk waits for memory (runs of 8 adjacent pixels change label together and read one
// 32-byte run of g).  Plane counters and quotient alphas as in step_walk: the record's attribution
// of crossings to axes rests on them.  rec = {S0x, S0z, S1x, S1z}.
template <int MAXSTEPS = 3 * BRICK + 4, class Fetch, class Weight>
DDRR_HD int step_walk_weighted(const Fetch &fetch, const StepGeom &G, const StepEntry &E,
                               const Weight &weight, float &I, float rec[4]) {
    float kr0 = 0.f, kr1 = 0.f, kr2 = 0.f;
    float an0 = E.an[0], an1 = E.an[1], an2 = E.an[2];
    const float inv0 = E.inv[0], inv1 = E.inv[1], inv2 = E.inv[2];
    const float af0 = E.a0[0], af1 = E.a0[1], af2 = E.a0[2];
    const float dir0 = E.dirf[0], dir1 = E.dirf[1], dir2 = E.dirf[2];
    const float sb0 = in_vgpr(G.strideb[0]), sb1 = in_vgpr(G.strideb[1]);
    const float sb2 = in_vgpr(G.strideb[2]);
    const float nbig = in_vgpr(-kSelBig);
    const float exit = E.exit, offc = E.offc;
    const float nlbig = in_vgpr(-E.lbig), exit_big = exit * E.lbig;
    float a_cur = E.entry, acc = 0.f;
    unsigned Wn = float_bits(fetch(float_bits(fmaf(kr0, sb0, fmaf(kr1, sb1, fmaf(kr2, sb2, offc))))));
    unsigned cur = 0xffffffffu;  // (no label yet: the first voxel fetches its weight)
    float gcur = 0.f, Vc = 0.f, Vp = 0.f;
    float tpx = E.ent[0], tpy = E.ent[1];
    float S0x = 0.f, S0y = 0.f, S1x = 0.f, S1y = 0.f;
    float live = 1.f;
    int it = 0;
    for (; it < MAXSTEPS; ++it) {
        // the voxel whose segment this step closes: its weighted value
        const unsigned lab = Wn & 0xffu;
        if (lab != cur) {
            gcur = weight(lab);
            cur = lab;
        }
        Vc = bits_as_float(Wn & 0xffffff00u) * gcur;
        const float a_next = fminf(fminf(an0, an1), an2);
        const float len = a_next - a_cur;
        live = sel_zero(a_next, nlbig, exit_big);
        const float t0 = sel_zero(an0 - a_next, nbig, live);
        const float t1 = sel_zero(an1 - a_next, nbig, live);
        const float t2 = sel_zero(an2 - a_next, nbig, live);
        kr0 = fmaf(t0, dir0, kr0);
        kr1 = fmaf(t1, dir1, kr1);
        kr2 = fmaf(t2, dir2, kr2);
        Wn = float_bits(fetch(float_bits(fmaf(kr0, sb0, fmaf(kr1, sb1, fmaf(kr2, sb2, offc))))));
        an0 = fmaf(kr0, inv0, af0);
        an1 = fmaf(kr1, inv1, af1);
        an2 = fmaf(kr2, inv2, af2);
        acc = fmaf(Vc, len, acc);
        const float dv = Vp - Vc, dva = dv * a_cur;
        S0x = fmaf(dv, tpx, S0x);
        S0y = fmaf(dv, tpy, S0y);
        S1x = fmaf(dva, tpx, S1x);
        S1y = fmaf(dva, tpy, S1y);
        tpx = t0;
        tpy = fmaf(-t0, t1, t1);  // exclusive: x before y
        a_cur = a_next;
        Vp = Vc;
#if defined(__HIP_DEVICE_COMPILE__)
        if ((it & 1) && !__builtin_amdgcn_ballot_w64(live != 0.f)) break;
#else
        if (live == 0.f) break;
#endif
    }
    I = acc;
    // (a lane that is done idles on its last voxel with zero-length steps: Vp == Vc, nothing added)
    // the exit crossing: last voxel | 0 at alpha = exit, on the axis whose face it is
    const float ex = sel_zero(an0 - exit, nbig, 1.f);
    const float ey0 = sel_zero(an1 - exit, nbig, 1.f);
    const float ey = fmaf(-ex, ey0, ey0);
    const float va = Vp * exit;
    S0x = fmaf(Vp, ex, S0x);
    S0y = fmaf(Vp, ey, S0y);
    S1x = fmaf(va, ex, S1x);
    S1y = fmaf(va, ey, S1y);
    rec[0] = S0x;
    rec[1] = -(S0x + S0y);       // sum_a S0_a = 0
    rec[2] = S1x;
    rec[3] = acc - (S1x + S1y);  // sum_a S1_a = I
    return it + 1;
}

// An accumulator may take its weights pre-scaled (the LDS fixed-point accumulator: one multiply by
// its scale per ray instead of one per corner): `scale(w)` once, `add_scaled(addr, v)` per corner;
// any other accumulator is called as acc(addr, v).
template <class Acc>
DDRR_HD auto acc_scale(const Acc &a, float w, int) -> decltype(a.scale(w)) {
    return a.scale(w);
}
template <class Acc>
DDRR_HD float acc_scale(const Acc &, float w, long) {
    return w;
}
template <class Acc>
DDRR_HD auto acc_add(const Acc &a, unsigned addr, float v, int) -> decltype(a.add_scaled(addr, v)) {
    a.add_scaled(addr, v);
}
template <class Acc>
DDRR_HD void acc_add(const Acc &a, unsigned addr, float v, long) {
    a(addr, v);
}

// Volume gradient of one ray through one brick: adds w * dalpha_k to the LDS cell of every
// voxel the ray crosses (d out / d V[k] = L dalpha_k; reference: grid_sampler_3d_backward
// behind renderers.py:159-164).  `add(bits, value)` is the scatter at byte address `bits`.
// Same clip, alphas and stepping as the forward walk: <render(V), g> == <V, gradient(g)> holds
// segment by segment.  Lanes are left at their own pace here (a scatter has nothing in flight
// to wait for, and an idle lane would still issue LDS atomics).
template <class Add>
DDRR_HD bool step_scatter(const Add &add, unsigned base_bits, const StepGeom &G, const float s[3],
                          const float t[3], float shift, float eps, float w) {
    const StepEntry E = step_enter(G, s, t, shift, eps, base_bits);
    if (!E.hit) return false;
    float kr0 = 0.f, kr1 = 0.f, kr2 = 0.f;
    float an0 = E.an[0], an1 = E.an[1], an2 = E.an[2];
    const float sb0 = in_vgpr(G.strideb[0]), sb1 = in_vgpr(G.strideb[1]);
    const float sb2 = in_vgpr(G.strideb[2]);
    const float nbig = in_vgpr(-kSelBig), one = in_vgpr(1.f);
    float a_cur = E.entry;
    const float ws = acc_scale(add, w, 0);
    for (int it = 0; it < 3 * BRICK + 4; ++it) {
        const unsigned addr =
            float_bits(fmaf(kr0, sb0, fmaf(kr1, sb1, fmaf(kr2, sb2, E.offc))));
        const float a_next = fminf(fminf(an0, an1), an2);
        acc_add(add, addr, ws * (a_next - a_cur), 0);
        if (!(a_next < E.exit)) break;
        kr0 = fmaf(sel_zero(an0 - a_next, nbig, one), E.dirf[0], kr0);
        kr1 = fmaf(sel_zero(an1 - a_next, nbig, one), E.dirf[1], kr1);
        kr2 = fmaf(sel_zero(an2 - a_next, nbig, one), E.dirf[2], kr2);
        an0 = fmaf(kr0, E.inv[0], E.a0[0]);
        an1 = fmaf(kr1, E.inv[1], E.a0[1]);
        an2 = fmaf(kr2, E.inv[2], E.a0[2]);
        a_cur = a_next;
    }
    return true;
}

// The volume gradient of the channel render inside one brick (reference renderers.py:77-89,
// autograd w.r.t. the volume): every voxel the ray crosses receives len L grad_out[b, label, n].
// step_scatter with the weight of the voxel's own label: `label(addr)` looks the label of the
// voxel at LDS byte address `addr` up, `weight(label)` gathers the incoming gradient of the ray's
// column -- only when the label changes along the ray.
template <class Add, class Label, class Weight>
DDRR_HD bool step_scatter_weighted(const Add &add, const Label &label, const Weight &weight,
                                   unsigned base_bits, const StepGeom &G, const float s[3],
                                   const float t[3], float shift, float eps, float L) {
    const StepEntry E = step_enter(G, s, t, shift, eps, base_bits);
    if (!E.hit) return false;
    float kr0 = 0.f, kr1 = 0.f, kr2 = 0.f;
    float an0 = E.an[0], an1 = E.an[1], an2 = E.an[2];
    const float sb0 = in_vgpr(G.strideb[0]), sb1 = in_vgpr(G.strideb[1]);
    const float sb2 = in_vgpr(G.strideb[2]);
    const float nbig = in_vgpr(-kSelBig), one = in_vgpr(1.f);
    float a_cur = E.entry, w = 0.f;
    int cur = -1;
    for (int it = 0; it < 3 * BRICK + 4; ++it) {
        const unsigned addr =
            float_bits(fmaf(kr0, sb0, fmaf(kr1, sb1, fmaf(kr2, sb2, E.offc))));
        const float a_next = fminf(fminf(an0, an1), an2);
        const int lab = (int)label(addr);
        if (lab != cur) {
            cur = lab;
            w = acc_scale(add, weight((unsigned)lab) * L, 0);
        }
        acc_add(add, addr, w * (a_next - a_cur), 0);
        if (!(a_next < E.exit)) break;
        kr0 = fmaf(sel_zero(an0 - a_next, nbig, one), E.dirf[0], kr0);
        kr1 = fmaf(sel_zero(an1 - a_next, nbig, one), E.dirf[1], kr1);
        kr2 = fmaf(sel_zero(an2 - a_next, nbig, one), E.dirf[2], kr2);
        an0 = fmaf(kr0, E.inv[0], E.a0[0]);
        an1 = fmaf(kr1, E.inv[1], E.a0[1]);
        an2 = fmaf(kr2, E.inv[2], E.a0[2]);
        a_cur = a_next;
    }
    return true;
}

// Exact clip + walk of a 16-bit brick (G: its byte strides as bit-pattern floats).
template <bool AUX, int MAXSTEPS, class Fetch>
DDRR_HD bool step_trace_q16(const Fetch &fetch, unsigned base_bits, const StepGeom &G,
                            const Q16Range &range, const float s[3], const float t[3], float shift,
                            float eps, float &I, float rec[4]) {
    StepEntry E = step_enter(G, s, t, shift, eps, base_bits);
    I = 0.f;
    if (AUX) rec[0] = rec[1] = rec[2] = rec[3] = 0.f;
    if (!E.hit) return false;
    q16_scale_entry(E);
    if (!AUX) {
        float acc, a_end;
        step_walk_fwd<MAXSTEPS>(fetch, G, E, acc, a_end);
        I = q16_finish_fwd(range, E, acc, a_end);
        return true;
    }
    float ex[2] = {0.f, 0.f};
    step_walk<AUX, MAXSTEPS>(fetch, G, E, I, rec, ex);
    q16_finish<AUX>(range, E, ex, I, rec);
    return true;
}

// Exact clip + walk.  Returns false if the ray does not cross the brick.
template <bool AUX, class Fetch>
DDRR_HD bool step_trace(const Fetch &fetch, unsigned base_bits, const StepGeom &G,
                        const float s[3], const float t[3], float shift, float eps, float &I,
                        float rec[4]) {
    const StepEntry E = step_enter(G, s, t, shift, eps, base_bits);
    I = 0.f;
    if (AUX) rec[0] = rec[1] = rec[2] = rec[3] = 0.f;
    if (!E.hit) return false;
    if (!AUX) {
        float a_end;
        step_walk_fwd(fetch, G, E, I, a_end);
        return true;
    }
    step_walk<AUX>(fetch, G, E, I, rec);
    return true;
}

}  // namespace ddrr
