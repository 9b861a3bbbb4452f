// record_layout.h -- where the float backward record of the brick kernel lives in memory.
//
// The forward + record kernel adds the brick-local pieces (I, S0x, S0z, S1x, S1z) of every
// (ray, brick) pair to the ray's record with device-scope atomics, and those are executed at
// the memory side at a rate per 64-byte REQUEST (one atomic instruction's lanes that fall into
// one 64-byte line), not per lane (profiles/r02: 3.4e7 requests per launch at 512^3 / 32 poses
// were what bounded the kernel).  Hits are walked in runs of 8 adjacent pixels (bricks.hip),
// so the record is laid out such that one run fills whole lines:
//
//   block = 16 consecutive rays (r >> 4), 80 floats = five 64-byte lines
//     line 0: rays 0-7  plane 0 | rays 0-7  plane 1
//     line 1: rays 0-7  plane 2 | rays 0-7  plane 3
//     line 2: rays 8-15 plane 0 | rays 8-15 plane 1
//     line 3: rays 8-15 plane 2 | rays 8-15 plane 3
//     line 4: rays 0-15 plane 4
//
// and the kernel swaps values between lane l and lane l ^ 8 (one DPP row rotate per plane)
// so that a 16-lane row of an atomic instruction covers one whole line: 2.5 requests per run
// of 8 instead of 5.  planes: 0 I, 1 S0x, 2 S0z, 3 S1x, 4 S1z (y follows from the sums).
#pragma once

#include "ddrr_common.h"

namespace ddrr {

constexpr int kRecBlockRays = 16;
constexpr int kRecBlockFloats = 80;

// floats of the blocked record of R rays
DDRR_HD long rec_blocked_floats(long R) {
    return (R + kRecBlockRays - 1) / kRecBlockRays * kRecBlockFloats;
}

// float index of planes 0 / 1 of ray r (planes 2 / 3: + 16); plane 1 = plane 0 + 8
DDRR_HD unsigned rec_off01(unsigned r) {
    return (r >> 4) * (unsigned)kRecBlockFloats + ((r >> 3) & 1u) * 32u + (r & 7u);
}
DDRR_HD unsigned rec_off4(unsigned r) {
    return (r >> 4) * (unsigned)kRecBlockFloats + 64u + (r & 15u);
}

// float index of (plane, ray); 64-bit ray index for the consumers
DDRR_HD long rec_index(long r, int plane) {
    const long blk = (r >> 4) * kRecBlockFloats;
    if (plane == 4) return blk + 64 + (r & 15);
    return blk + ((r >> 3) & 1) * 32 + (plane >> 1) * 16 + (plane & 1) * 8 + (r & 7);
}

// the five planes of ray r -> the 8-float record of siddon_backward_ray
// {I, S0x, S0y, S0z, S1x, S1y, S1z, -}
DDRR_HD void rec_blocked_load(const float *aux, long r, float rec[8]) {
    const float I = aux[rec_index(r, 0)], S0x = aux[rec_index(r, 1)], S0z = aux[rec_index(r, 2)];
    const float S1x = aux[rec_index(r, 3)], S1z = aux[rec_index(r, 4)];
    rec[0] = I;
    rec[1] = S0x;
    rec[2] = -(S0x + S0z);  // sum_a S0_a = 0
    rec[3] = S0z;
    rec[4] = S1x;
    rec[5] = I - (S1x + S1z);  // sum_a S1_a = I
    rec[6] = S1z;
    rec[7] = 0.f;
}

}  // namespace ddrr
