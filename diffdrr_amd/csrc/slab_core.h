// slab_core.h -- lockstep "slab march" form of the Siddon walk (the fast path).
//
// Same integral as siddon_core.h (reference diffdrr/renderers.py:34-76), but
// organised for the memory system instead of per ray: the 64 lanes of a wave
// advance together, one slab of the pose's dominant ("march") axis m per
// iteration, so that at every iteration all lanes read the SAME m-plane of the
// volume.  With the wave's rays chosen in one plane through the source that
// contains the volume's z axis (ddrr_common.h ShearMap) their voxels then sit
// in one or two (m, u) rows and differ only in z, the fastest axis: a wave's
// fetch is a handful of fully used cache lines instead of 64 scattered ones
// (profiles/r01: the per-crossing walk fetched 11x its algorithmic bytes).
//
// Inside a slab a ray that is dominant along m (|d_m| >= |d_u|, |d_z|) crosses
// at most one u-plane and one z-plane, i.e. at most three segments in the
// voxels (u,z), (u',z) or (u,z'), (u',z').  The two z-neighbours are adjacent
// in memory, so one 8-byte fetch per touched row covers them: row u always,
// row u' only in the iterations where some lane crosses a u-plane.
// Non-dominant rays (wide cones at oblique poses) are flagged and take the
// generic walk; results are identical either way.
#pragma once

#include "ddrr_common.h"
#include "siddon_core.h"

namespace ddrr {

// Per-pose constants of the march: which axis is marched (0 = x or 1 = y; z is
// never the march axis here -- z-dominant poses use the generic walk).
struct SlabAxes {
    int m, u;                // march axis, row axis ({m, u} = {0, 1})
    int Dm, Du, Dz;
    int stride_m, stride_u;  // byte strides of one voxel step
};

DDRR_HD SlabAxes make_slab_axes(const Dims D, int m) {
    SlabAxes ax;
    ax.m = m;
    ax.u = 1 - m;
    // (selects, not runtime-indexed arrays: those would live in scratch memory)
    ax.Dm = m == 0 ? D.x : D.y;
    ax.Du = m == 0 ? D.y : D.x;
    ax.Dz = D.z;
    ax.stride_m = m == 0 ? D.y * D.z * 4 : D.z * 4;
    ax.stride_u = m == 0 ? D.z * 4 : D.y * D.z * 4;
    return ax;
}

struct SlabLane {
    // per-ray constants
    float inv_m, c_m, inv_u, c_u, inv_z, c_z;
    float entry, exit;
    float dirf_m, dirf_u, dirf_z;
    int dstep_m, dstep_u, diz;
    int im_in;      // voxel index along m where the ray enters the volume
    int entry_axis; // 0 = m, 1 = u, 2 = z (exclusive, priority in x,y,z order)
    int exit_axis;
    // walk state
    float km, ku, kz;  // index of the next plane to cross, per axis
    float au, az;      // alpha of the next u / z plane
    float a_cur;
    unsigned row_off;  // byte offset of the current (m, u) row (z = 0)
    int iz;
    bool hit, fast, started, done;
    // results
    float acc;
    float S0u, S1u, S0z, S1z;
};

// Set a lane up from its ray.  `fast` = can take the slab march.
DDRR_HD SlabLane slab_lane_init(const Dims D, const Box &box, const SlabAxes &ax,
                                const float s[3], const float t[3], float shift, float eps) {
    SlabLane L;
    const SiddonSetup q = siddon_setup(box, s, t, shift, eps);
    const bool mx = ax.m == 0;  // march along x (rows along y) or along y (rows along x)
    L.hit = q.hit;
    L.acc = 0.f;
    L.S0u = L.S1u = L.S0z = L.S1z = 0.f;
    L.inv_m = mx ? q.inv[0] : q.inv[1];
    L.c_m = mx ? q.c[0] : q.c[1];
    L.inv_u = mx ? q.inv[1] : q.inv[0];
    L.c_u = mx ? q.c[1] : q.c[0];
    L.inv_z = q.inv[2];
    L.c_z = q.c[2];
    L.entry = q.entry;
    L.exit = q.exit;
    L.a_cur = q.entry;
    L.done = !q.hit;
    L.started = false;
    // dominant along m, and thick enough in z for the 8-byte pair fetch
    const float dm = mx ? q.d[0] : q.d[1], du = mx ? q.d[1] : q.d[0];
    L.fast = q.hit && fabsf(dm) >= fabsf(du) && fabsf(dm) >= fabsf(q.d[2]) && ax.Dz >= 2;
    // which plane opens the first / closes the last segment (exclusive, x > y > z);
    // roles: 0 = march axis, 1 = row axis, 2 = z
    const int role_x = mx ? 0 : 1, role_y = mx ? 1 : 0;
    L.entry_axis = q.lo[0] == q.entry ? role_x : (q.lo[1] == q.entry ? role_y : 2);
    L.exit_axis = q.hi[0] == q.exit ? role_x : (q.hi[1] == q.exit ? role_y : 2);
    L.im_in = 0;
    L.km = L.ku = L.kz = 0.f;
    L.au = L.az = 0.f;
    L.row_off = 0u;
    L.iz = 0;
    L.dirf_m = L.dirf_u = L.dirf_z = 1.f;
    L.dstep_m = L.dstep_u = 0;
    L.diz = 1;
    if (q.hit) {
        const SiddonWalk w = siddon_enter(global_store(D), box, s, shift, q);
        L.km = mx ? w.kf[0] : w.kf[1];
        L.ku = mx ? w.kf[1] : w.kf[0];
        L.kz = w.kf[2];
        L.au = mx ? w.an[1] : w.an[0];
        L.az = w.an[2];
        L.dirf_m = mx ? w.dirf[0] : w.dirf[1];
        L.dirf_u = mx ? w.dirf[1] : w.dirf[0];
        L.dirf_z = w.dirf[2];
        L.dstep_m = mx ? w.dstep[0] : w.dstep[1];
        L.dstep_u = mx ? w.dstep[1] : w.dstep[0];
        L.diz = w.dirf[2] > 0.f ? 1 : -1;
        // voxel indices at entry, recovered from the next-plane counters
        const int im = (int)L.km - (L.dirf_m > 0.f ? 1 : 0);
        const int iu = (int)L.ku - (L.dirf_u > 0.f ? 1 : 0);
        L.iz = (int)w.kf[2] - (w.dirf[2] > 0.f ? 1 : 0);
        L.im_in = im;
        L.row_off = (unsigned)(im * ax.stride_m + iu * ax.stride_u);
    }
    return L;
}

// What one slab iteration needs from memory and how to weigh it.
struct SlabGeo {
    float l0, l1, l2;     // lengths of the (up to) three segments
    unsigned offA, offB;  // byte offsets of the two 8-byte fetches (rows u and u')
    bool s0, s1;          // pair element holding z (false = first) / holding z'
    bool cx, cz, xfirst;
    bool first, last;     // lane enters / leaves the volume in this slab
    float au, az;         // alphas of the u / z crossing (valid if cx / cz)
};

// Advance an ACTIVE lane by one m-slab: everything here depends on the ray
// only, never on voxel values, so the next slab's fetches can be issued before
// the current slab's values are consumed.  Inactive lanes (not entered yet,
// finished, or slow) get zero lengths and an in-bounds dummy address.
DDRR_HD SlabGeo slab_geometry(SlabLane &L, const SlabAxes &ax, bool active) {
    SlabGeo g;
    const float a_m = fmaf(L.km, L.inv_m, L.c_m);
    const float a_out = active ? fminf(a_m, L.exit) : L.a_cur;
    g.cx = active && L.au < a_out;
    g.cz = active && L.az < a_out;
    const float ax_ = g.cx ? L.au : a_out;
    const float az_ = g.cz ? L.az : a_out;
    const float a1 = fminf(ax_, az_), a2 = fmaxf(ax_, az_);
    g.xfirst = ax_ <= az_;
    g.l0 = a1 - L.a_cur;
    g.l1 = a2 - a1;
    g.l2 = a_out - a2;
    g.au = L.au;
    g.az = L.az;
    g.first = active && !L.started;
    L.started = L.started || active;
    // 8-byte pair base along z: covers z and its neighbour in the direction of travel
    int zb = L.diz > 0 ? L.iz : L.iz - 1;
    zb = zb < 0 ? 0 : (zb > ax.Dz - 2 ? ax.Dz - 2 : zb);
    g.s0 = L.iz != zb;
    g.s1 = (L.iz + L.diz) != zb;
    g.offA = L.row_off + (unsigned)(zb * 4);
    g.offB = g.offA + (unsigned)(g.cx ? L.dstep_u : 0);
    // state update
    L.ku += g.cx ? L.dirf_u : 0.f;
    L.kz += g.cz ? L.dirf_z : 0.f;
    L.au = fmaf(L.ku, L.inv_u, L.c_u);
    L.az = fmaf(L.kz, L.inv_z, L.c_z);
    L.iz += g.cz ? L.diz : 0;
    const bool more = active && a_out < L.exit;
    g.last = active && !more;
    // stay on the last voxel's row once the ray has left (keeps addresses valid)
    L.row_off += (unsigned)((g.cx && more ? L.dstep_u : 0) + (more ? L.dstep_m : 0));
    // NB: when the ray leaves in this slab a pending u-step is not taken either:
    // its crossing would be at/after the exit plane, so cx is false then anyway.
    L.km += active ? L.dirf_m : 0.f;
    L.a_cur = a_out;
    L.done = L.done || (active && !more);
    return g;
}

// Fold the fetched pairs of one slab into the lane's sums.
template <bool AUX>
DDRR_HD void slab_consume(SlabLane &L, const SlabGeo &g, float ax0, float ax1, float bx0,
                          float bx1) {
    const float v00 = g.s0 ? ax1 : ax0;
    const float v01 = g.s1 ? ax1 : ax0;
    const float v10 = g.s0 ? bx1 : bx0;
    const float v11 = g.s1 ? bx1 : bx0;
    const float mid = g.xfirst ? v10 : v01;
    L.acc = fmaf(v00, g.l0, L.acc);
    L.acc = fmaf(mid, g.l1, L.acc);
    L.acc = fmaf(v11, g.l2, L.acc);
    if (AUX) {
        // u-plane crossing at g.au: V_before - V_after
        const float du = g.cx ? (g.xfirst ? v00 - v10 : v01 - v11) : 0.f;
        // z-plane crossing at g.az
        const float dz = g.cz ? (g.xfirst ? v10 - v11 : v00 - v01) : 0.f;
        L.S0u += du;
        L.S1u = fmaf(du, g.au, L.S1u);
        L.S0z += dz;
        L.S1z = fmaf(dz, g.az, L.S1z);
        // the planes through which the ray enters / leaves the volume
        const float vin = g.first ? v00 : 0.f;
        const float vend = (g.cx && g.cz) ? v11 : ((g.cx || g.cz) ? mid : v00);
        const float vout = g.last ? vend : 0.f;
        if (L.entry_axis == 1) {
            L.S0u -= vin;
            L.S1u = fmaf(-vin, L.entry, L.S1u);
        } else if (L.entry_axis == 2) {
            L.S0z -= vin;
            L.S1z = fmaf(-vin, L.entry, L.S1z);
        }
        if (L.exit_axis == 1) {
            L.S0u += vout;
            L.S1u = fmaf(vout, L.exit, L.S1u);
        } else if (L.exit_axis == 2) {
            L.S0z += vout;
            L.S1z = fmaf(vout, L.exit, L.S1z);
        }
    }
}

// Aux record (layout of siddon_core.h SIDDON_AUX, sum mode) from the lane sums.
DDRR_HD void slab_aux_record(const SlabLane &L, const SlabAxes &ax, float rec[SIDDON_AUX]) {
    const float S0m = -(L.S0u + L.S0z);
    const float S1m = L.acc - (L.S1u + L.S1z);
    const bool mx = ax.m == 0;
    rec[0] = L.acc;
    rec[1] = mx ? S0m : L.S0u;
    rec[2] = mx ? L.S0u : S0m;
    rec[3] = L.S0z;
    rec[4] = mx ? S1m : L.S1u;
    rec[5] = mx ? L.S1u : S1m;
    rec[6] = L.S1z;
    rec[7] = 0.f;
}

}  // namespace ddrr
