// record_pack.h -- the fixed-point form of the brick kernel's backward record.
//
// The forward + record kernel adds its per-brick pieces of (I, S0x, S0z, S1x, S1z) to the
// ray's record with atomics, and the atomics are what bounds it (profiles/r01/exp_record_cost.txt:
// their cost is per instruction and 64-byte request).  Two 32-bit fixed-point fields per 64-bit
// integer atomic carry (S0x, S1x) and (S0z, S1z): 3 atomic instructions per ray and brick
// instead of 5, and cheap enough that hits no longer have to be batched in runs of 8 adjacent
// pixels.  Integer sums are exact and order independent, so -- unlike the float record -- the
// packed record is bit-reproducible.
//
// Scales.  With vmax = max |V| and n = Dx + Dy + Dz + 3 crossings at most,
//   |S0_a| <= 2 vmax n =: bound,   |S1_a| <= A bound,   A = max(1, max |alpha| inside the volume),
// fields hold rint(S0 q) and rint(S1 q / A) with q = 2^30 / bound: no field can overflow into its
// neighbour, and the resolution is bound / 2^30 (2.9e-6 vmax at 512^3) per added piece.
// A is per ray (alpha is unbounded when the detector plane cuts the volume), computed once per
// launch by record_prepare_kernel and kept next to the record.
//
// Layout of the packed record, (7, B, N) floats: planes 0-1: 64-bit (S1x : S0x) per ray,
// planes 2-3: (S1z : S0z), plane 4: I (float), plane 5: A, plane 6: [q, ...].
#pragma once

#include "ddrr_common.h"
#include "siddon_core.h"

namespace ddrr {

constexpr int REC_PACKED_PLANES = 7;

DDRR_HD float record_scale(float vmax, const Dims D) {
    return 1073741824.f / (2.f * vmax * (float)(D.x + D.y + D.z + 3));
}

// max |alpha| over the part of the line inside the volume, at least 1
DDRR_HD float record_alpha_bound(const Dims D, const float s[3], const float t[3], float shift,
                                 float eps) {
    const SiddonSetup q = siddon_setup(full_box(D), s, t, shift, eps);
    if (!q.hit) return 1.f;
    return fmaxf(1.f, fmaxf(fabsf(q.entry), fabsf(q.exit)));
}

DDRR_HD long long record_pack(float s0, float s1, float q, float qa) {
    const long long i0 = (long long)rintf(s0 * q), i1 = (long long)rintf(s1 * qa);
    return i1 * 4294967296LL + i0;
}

DDRR_HD void record_unpack(long long v, float q, float qa, float &s0, float &s1) {
    const int lo = (int)(unsigned)(unsigned long long)v;  // low field, sign-extended by the cast
    const long long hi = (v - (long long)lo) / 4294967296LL;  // exact: divisible
    s0 = (float)lo / q;
    s1 = (float)hi / qa;
}

}  // namespace ddrr
