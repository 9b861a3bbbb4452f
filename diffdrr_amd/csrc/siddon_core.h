// siddon_core.h -- per-ray exact radiological path (Siddon) for one lane.
//
// What it replaces: the tensor program of the reference's Siddon renderer,
//   diffdrr/renderers.py:34-76   Siddon.forward (mask=None branch)
//   diffdrr/renderers.py:94-113  _get_alphas  (every plane crossing + torch.sort)
//   diffdrr/renderers.py:143-169 _get_xyzs / _get_voxel (midpoint -> grid_sample nearest)
//   diffdrr/renderers.py:175-183 reduce (sum / max)
// which materialises (B, N, Dx+Dy+Dz+3) tensors.  Here a lane owns one ray and
// merges the three monotone crossing sequences on the fly (no sort, no
// temporaries); the only memory traffic is one 4-byte voxel fetch per segment.
//
// Semantics kept from the reference (SURVEY.md section 7 "exact semantic quirks"):
//  * the WHOLE line through source and target is integrated (alpha is not
//    clipped to [0, 1]);
//  * planes of axis a sit at i - voxel_shift, i = 0..D_a; with "u = x + shift"
//    they are the integers and voxel i is u in [i, i+1)  (nearest-neighbour
//    grid_sample of the segment midpoint, align_corners=False);
//  * the direction is d = (t - s) + eps, used both for alpha and for positions;
//  * crossings are evaluated directly from the integer plane index,
//    alpha_a(k) = fma(k, 1/d_a, (-shift - s_a)/d_a), never accumulated, so no
//    drift builds up along a ray.
#pragma once

#include "ddrr_common.h"

namespace ddrr {

// Sub-box of the volume a pass is restricted to, in plane indices per axis
// (lo <= planes <= hi; voxels lo .. hi-1).  The full volume is {0,0,0}..{Dx,Dy,Dz}.
// Restricting a walk to a box and adding the results over a partition of the
// volume into boxes gives the same integral (and the same backward record):
// used to render a large volume in Infinity-Cache-sized passes.
struct Box {
    int lo[3], hi[3];
};

DDRR_HD Box full_box(const Dims D) {
    Box b;
    b.lo[0] = b.lo[1] = b.lo[2] = 0;
    b.hi[0] = D.x;
    b.hi[1] = D.y;
    b.hi[2] = D.z;
    return b;
}

// Where the voxels of the box being walked are stored: the whole volume in global
// memory (dims = volume dims, org = 0), or a brick staged in LDS (dims = the
// brick's storage dims, org = the brick's first voxel).  Offsets are in bytes.
struct Store {
    Dims dims;      // extent of the stored box in voxels (bounds the walk's trip count)
    int org[3];     // first voxel of the stored box
    int stride[3];  // BYTE strides of one voxel step along x, y, z
};

DDRR_HD Store global_store(const Dims D) {
    Store st;
    st.dims = D;
    st.org[0] = st.org[1] = st.org[2] = 0;
    st.stride[0] = D.y * D.z * 4;
    st.stride[1] = D.z * 4;
    st.stride[2] = 4;
    return st;
}

struct GlobalFetch {
    const float *vol;
    DDRR_HD float operator()(unsigned boff) const {
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(vol) + boff);
    }
};

// n / d from n, d and r ~ 1/d (<= 1 ulp): one residual correction of the product, i.e. the
// correctly rounded quotient in all but a vanishing fraction of cases.  Deterministic in (n, d).
DDRR_HD float div_refined(float n, float d, float r) {
    const float q0 = n * r;
    const float q1 = fmaf(fmaf(-q0, d, n), r, q0);
    return (q1 == q1) ? q1 : q0;  // d = 0 or inf: keep the plain product
}

struct SiddonSetup {
    float d[3], inv[3], s[3], lo[3], hi[3];
    float shift;
    float entry, exit;
    bool hit;
};

// alpha of plane k of axis a: the reference's own quotient (k - shift - s_a) / (t_a - s_a + eps)
// (diffdrr/renderers.py:97-106), numerator and all.  Every Siddon walk evaluates every
// crossing through this one expression from the integer plane index -- never k * (1/d) + c,
// which cancels two numbers of size |k/d| for the lateral axes (~10 ulps of alpha at 512^3,
// and any amount for a ray gliding along a plane) -- so sub-boxes meet exactly and ties fall
// where the reference's fp32 arithmetic puts them.
DDRR_HD float plane_alpha(const SiddonSetup &q, int a, float kf) {
    return div_refined((kf - q.shift) - q.s[a], q.d[a], q.inv[a]);
}

DDRR_HD float fast_rcp(float x);

template <bool FAST>
DDRR_HD SiddonSetup siddon_setup_t(const Box &box, const float s[3], const float t[3],
                                   float shift, float eps) {
    SiddonSetup q;
    q.entry = -INFINITY;
    q.exit = INFINITY;
    q.shift = shift;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q.d[a] = (t[a] - s[a]) + eps;  // renderers.py:104-106, :148
        q.inv[a] = FAST ? fast_rcp(q.d[a]) : 1.0f / q.d[a];
        q.s[a] = s[a];
        const float a0 = plane_alpha(q, a, (float)box.lo[a]);
        const float aD = plane_alpha(q, a, (float)box.hi[a]);
        q.lo[a] = fminf(a0, aD);
        q.hi[a] = fmaxf(a0, aD);
        q.entry = fmaxf(q.entry, q.lo[a]);
        q.exit = fminf(q.exit, q.hi[a]);
    }
    q.hit = q.entry < q.exit;  // false for NaN
    return q;
}

DDRR_HD SiddonSetup siddon_setup(const Box &box, const float s[3], const float t[3], float shift,
                                 float eps) {
    return siddon_setup_t<false>(box, s, t, shift, eps);
}

// Reciprocal for per-(ray, brick) setups, which run ~25x per ray: v_rcp_f32 + one Newton step
// (<= 1 ulp) instead of the IEEE division sequence.  Deterministic in its input.
DDRR_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float r1 = fmaf(fmaf(-x, r0, 1.0f), r0, r0);
    return (r1 == r1) ? r1 : r0;  // x = 0 or inf: keep the hardware's inf / 0
#else
    return 1.0f / x;
#endif
}

DDRR_HD SiddonSetup siddon_setup_fast(const Box &box, const float s[3], const float t[3],
                                      float shift, float eps) {
    return siddon_setup_t<true>(box, s, t, shift, eps);
}

// State of the 3-way merge once the ray is inside the volume.
struct SiddonWalk {
    float kf[3];    // index of the next plane to be crossed, per axis (as float)
    float dirf[3];  // +1 / -1
    float an[3];    // alpha of that plane
    int dstep[3];   // signed BYTE stride of one voxel step along the axis
    unsigned off;   // BYTE offset of the current voxel (volume <= 2^30 voxels)
};

DDRR_HD SiddonWalk siddon_enter(const Store &st, const Box &box, const float s[3], float shift,
                                const SiddonSetup &q) {
    SiddonWalk w;
    const int stride[3] = {st.stride[0], st.stride[1], st.stride[2]};
    w.off = 0u;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const bool pos = q.d[a] > 0.f;
        // position at entry in plane units; the entering axis is pinned to its
        // face voxel, the others are whatever cell the entry point lies in
        float u = floorf(fmaf(q.entry, q.d[a], s[a] + shift));
        u = fminf(fmaxf(u, (float)box.lo[a]), (float)(box.hi[a] - 1));
        int i = (int)u;
        if (q.lo[a] == q.entry) {
            i = pos ? box.lo[a] : box.hi[a] - 1;
        } else {
            // The cell must agree with the ORDER OF THE ALPHAS the walk steps by, not only
            // with the position: for a ray gliding along a plane of this axis (|d_a| tiny)
            // a position error of 1e-5 voxel is an alpha error of any size, and a cell on
            // the wrong side of an already-passed plane would open with a segment of
            // negative length (or skip one).  Move one cell if the plane ahead is already
            // behind `entry`, or the plane behind is still ahead of it.
            const float a_ahead = plane_alpha(q, a, (float)(i + (pos ? 1 : 0)));
            const float a_behind = plane_alpha(q, a, (float)(i + (pos ? 0 : 1)));
            const int di = pos ? 1 : -1;
            if (a_ahead < q.entry) i += di;
            else if (a_behind > q.entry) i -= di;
            i = i < box.lo[a] ? box.lo[a] : (i > box.hi[a] - 1 ? box.hi[a] - 1 : i);
        }
        w.off += (unsigned)((i - st.org[a]) * stride[a]);
        w.kf[a] = (float)(i + (pos ? 1 : 0));
        w.dirf[a] = pos ? 1.f : -1.f;
        w.dstep[a] = pos ? stride[a] : -stride[a];
        w.an[a] = plane_alpha(q, a, w.kf[a]);
    }
    return w;
}

// Voxel fetch by byte offset: an SGPR base + 32-bit VGPR offset global load.
DDRR_HD float vox(const float *__restrict__ vol, unsigned boff) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(vol) + boff);
}

// Layout of the 8-float per-ray record the forward kernel can emit for the
// backward pass:
//   sum : [0]=I  [1..3]=S0_xyz  [4..6]=S1_xyz  [7]=unused
//         I    = sum_k V_k (alpha_{k+1} - alpha_k)
//         S0_a = sum over crossings of axis a of (V_before - V_after)
//         S1_a = same, weighted by the crossing's alpha
//   max : [0]=V*(a_out-a_in)  [1]=V*  [2]=a_in  [3]=a_out  [4]=axis_in  [5]=axis_out
constexpr int SIDDON_AUX = 8;

// One generator step of the 3-way merge: closes the segment that starts at
// a_cur, returns its length and everything the consumer needs, and moves the
// walk to the next voxel.  Once the ray has left the volume (`live` false) it
// returns zero-length segments and stays on its last voxel, so that run-ahead
// fetches stay in bounds.  Branch-free on purpose: the forward loop must be one
// basic block for the fetches to be software-pipelined (counted vmcnt).
struct SiddonSeg {
    float seg;
    // sum-mode aux weights: S0x += w0x v, S1x += w1x v, S0z += w0z v, S1z += w1z v
    float w0x, w1x, w0z, w1z;
    // max-mode aux: the segment's two alphas and the axes of its two crossings
    float a_in, a_out, ax_in, ax_out;
};

struct SiddonGen {
    SiddonWalk w;
    float a_cur, exit;
    bool live;
    float o0x, o1x, o0z, o1z;  // what the crossing that opened the current segment contributed
    float axis_in;
};

template <int REDUCE, bool AUX>
DDRR_HD SiddonSeg siddon_step(SiddonGen &g, const SiddonSetup &q) {
    SiddonSeg r;
    SiddonWalk &w = g.w;
    const float a_next = min3f(w.an[0], w.an[1], w.an[2]);
    r.seg = g.live ? a_next - g.a_cur : 0.f;
    const float a_lim = g.live ? a_next : -INFINITY;  // dead: no axis advances
    const bool cx = w.an[0] <= a_lim, cy = w.an[1] <= a_lim, cz = w.an[2] <= a_lim;
    // advance every axis whose plane is reached (ties step together)
    w.kf[0] += cx ? w.dirf[0] : 0.f;
    w.kf[1] += cy ? w.dirf[1] : 0.f;
    w.kf[2] += cz ? w.dirf[2] : 0.f;
    w.an[0] = plane_alpha(q, 0, w.kf[0]);
    w.an[1] = plane_alpha(q, 1, w.kf[1]);
    w.an[2] = plane_alpha(q, 2, w.kf[2]);
    const unsigned noff = w.off + (unsigned)((cx ? w.dstep[0] : 0) + (cy ? w.dstep[1] : 0) +
                                             (cz ? w.dstep[2] : 0));
    if (AUX && REDUCE == REDUCE_SUM) {
        // exclusive attribution of a crossing to one axis (priority x > y > z) keeps
        // the telescoping identities sum_a S0_a = 0, sum_a S1_a = I exact at ties
        const bool ez = cz && !cx && !cy;
        const float n0x = cx ? 1.f : 0.f, n1x = cx ? a_next : 0.f;
        const float n0z = ez ? 1.f : 0.f, n1z = ez ? a_next : 0.f;
        // a dead step closes no segment: it must not charge the last voxel again
        r.w0x = g.live ? n0x - g.o0x : 0.f;
        r.w1x = g.live ? n1x - g.o1x : 0.f;
        r.w0z = g.live ? n0z - g.o0z : 0.f;
        r.w1z = g.live ? n1z - g.o1z : 0.f;
        g.o0x = n0x;
        g.o1x = n1x;
        g.o0z = n0z;
        g.o1z = n1z;
    }
    if (AUX && REDUCE == REDUCE_MAX) {
        r.a_in = g.a_cur;
        r.a_out = a_next;
        r.ax_in = g.axis_in;
        r.ax_out = cx ? 0.f : (cy ? 1.f : 2.f);
        g.axis_in = r.ax_out;
    }
    g.live = g.live && (a_next < g.exit);
    w.off = g.live ? noff : w.off;  // never step past the exit voxel
    g.a_cur = a_next;
    return r;
}

// Forward for one ray.  Returns the un-scaled line integral (sum) or the
// largest single-segment term (max); the caller multiplies by the ray length.
//
// Two voxel fetches are kept in flight per lane (slots A and B, each re-armed
// right after it is consumed), on top of the 8 waves per SIMD the 8-wave
// occupancy provides: the walk is a pure gather whose only lever against HBM /
// Infinity-Cache latency is the number of outstanding requests.
template <int REDUCE, bool AUX, bool COUNT, class Fetch>
DDRR_HD float siddon_walk_t(const Fetch &fetch, const Store &st, const Box &box,
                            const float s[3], float shift, const SiddonSetup &q, float *aux,
                            int *count) {
    const Dims D = st.dims;
    float acc = 0.f;  // sum: integral; max: best term (>= 0: segments outside the volume are 0)
    int nvis = 0;
    float S0x = 0.f, S1x = 0.f, S0z = 0.f, S1z = 0.f;             // aux (sum)
    float bV = 0.f, bIn = 0.f, bOut = 0.f, bAin = 0.f, bAout = 0.f;  // aux (max)
    if (q.hit) {
        SiddonGen g;
        g.w = siddon_enter(st, box, s, shift, q);
        g.a_cur = q.entry;
        g.exit = q.exit;
        g.live = true;
        // which crossing opened the first segment (exclusive, priority x > y > z)
        const bool ix = q.lo[0] == q.entry;
        const bool iy = !ix && q.lo[1] == q.entry;
        g.o0x = ix ? 1.f : 0.f;
        g.o1x = ix ? q.entry : 0.f;
        g.o0z = (!ix && !iy) ? 1.f : 0.f;
        g.o1z = (!ix && !iy) ? q.entry : 0.f;
        g.axis_in = ix ? 0.f : (iy ? 1.f : 2.f);

#define DDRR_CONSUME(v, r)                                        \
    do {                                                          \
        if (REDUCE == REDUCE_SUM) {                               \
            acc = fmaf(v, r.seg, acc);                            \
            if (AUX) {                                            \
                S0x = fmaf(r.w0x, v, S0x);                        \
                S1x = fmaf(r.w1x, v, S1x);                        \
                S0z = fmaf(r.w0z, v, S0z);                        \
                S1z = fmaf(r.w1z, v, S1z);                        \
            }                                                     \
        } else {                                                  \
            const float term = v * r.seg;                         \
            const bool better = term > acc;                       \
            acc = better ? term : acc;                            \
            if (AUX) {                                            \
                bV = better ? v : bV;                             \
                bIn = better ? r.a_in : bIn;                      \
                bOut = better ? r.a_out : bOut;                   \
                bAin = better ? r.ax_in : bAin;                   \
                bAout = better ? r.ax_out : bAout;                \
            }                                                     \
        }                                                         \
        if (COUNT) nvis += r.seg > 0.f ? 1 : 0;                   \
    } while (0)

        bool liveA = true;
        float vA = fetch(g.w.off);
        SiddonSeg rA = siddon_step<REDUCE, AUX>(g, q);
        float vB = fetch(g.w.off);
        SiddonSeg rB = siddon_step<REDUCE, AUX>(g, q);
        // safety net: a ray has at most Dx+Dy+Dz+3 crossings, two are retired per trip
        const int cap = (D.x + D.y + D.z + 3) / 2 + 2;
        for (int it = 0; it < cap && liveA; ++it) {
            DDRR_CONSUME(vA, rA);
            liveA = g.live;
            vA = fetch(g.w.off);
            rA = siddon_step<REDUCE, AUX>(g, q);
            DDRR_CONSUME(vB, rB);
            vB = fetch(g.w.off);
            rB = siddon_step<REDUCE, AUX>(g, q);
        }
#undef DDRR_CONSUME
    }
    if (AUX) {
        if (REDUCE == REDUCE_SUM) {
            aux[0] = acc;
            aux[1] = S0x;
            aux[2] = -(S0x + S0z);
            aux[3] = S0z;
            aux[4] = S1x;
            aux[5] = acc - (S1x + S1z);
            aux[6] = S1z;
            aux[7] = 0.f;
        } else {
            aux[0] = acc;
            aux[1] = bV;
            aux[2] = bIn;
            aux[3] = bOut;
            aux[4] = bAin;
            aux[5] = bAout;
            aux[6] = 0.f;
            aux[7] = 0.f;
        }
    }
    if (COUNT) *count = nvis;
    return acc;
}

template <int REDUCE, bool AUX, bool COUNT, class Fetch>
DDRR_HD float siddon_forward_ray_t(const Fetch &fetch, const Store &st, const Box &box,
                                   const float s[3], const float t[3], float shift, float eps,
                                   float *aux, int *count) {
    const SiddonSetup q = siddon_setup(box, s, t, shift, eps);
    return siddon_walk_t<REDUCE, AUX, COUNT>(fetch, st, box, s, shift, q, aux, count);
}

template <int REDUCE, bool AUX, bool COUNT>
DDRR_HD float siddon_forward_ray(const float *__restrict__ vol, const Dims D, const Box &box,
                                 const float s[3], const float t[3], float shift, float eps,
                                 float *aux, int *count) {
    return siddon_forward_ray_t<REDUCE, AUX, COUNT>(GlobalFetch{vol}, global_store(D), box, s, t,
                                                    shift, eps, aux, count);
}

// Gradient w.r.t. the voxel-space ray endpoints from the forward record
// (SURVEY.md section 8a).  `gl` = grad_out * ray length.  The ||t - s|| factor of the
// image is differentiated by autograd through `img`, not here.
template <int REDUCE>
DDRR_HD void siddon_backward_ray(const float *aux, const float s[3], const float t[3], float eps,
                                 float gl, float gs[3], float gt[3]) {
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) inv[a] = 1.0f / ((t[a] - s[a]) + eps);
    if (REDUCE == REDUCE_SUM) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float S0 = aux[1 + a], S1 = aux[4 + a];
            gs[a] = gl * (S1 - S0) * inv[a];  // d alpha / d s_a = (alpha - 1) / d_a
            gt[a] = -gl * S1 * inv[a];        // d alpha / d t_a = -alpha / d_a
        }
    } else {
        const float c = gl * aux[1];  // d out / d alpha_out = +L V*, d out / d alpha_in = -L V*
        const float a_in = aux[2], a_out = aux[3];
        const int ax_in = (int)aux[4], ax_out = (int)aux[5];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float vs = 0.f, vt = 0.f;
            if (a == ax_out) {
                vs += c * (a_out - 1.f);
                vt += -c * a_out;
            }
            if (a == ax_in) {
                vs += -c * (a_in - 1.f);
                vt += c * a_in;
            }
            gs[a] = vs * inv[a];
            gt[a] = vt * inv[a];
        }
    }
}

// Volume gradient for one ray: d out / d V[voxel of segment k] = L dalpha_k
// (sum), or only the arg-max segment (max).  `Add` is the scatter primitive
// (a hardware fp32 atomic on the GPU).
template <int REDUCE, class Add>
DDRR_HD void siddon_scatter_ray(const float *__restrict__ vol, const Dims D, const float s[3],
                                const float t[3], float shift, float eps, float gl, Add add) {
    const Box box = full_box(D);
    const SiddonSetup q = siddon_setup(box, s, t, shift, eps);
    if (!q.hit) return;
    SiddonWalk w = siddon_enter(global_store(D), box, s, shift, q);
    float a_cur = q.entry;
    unsigned off = w.off;  // bytes
    const int cap = D.x + D.y + D.z + 3;
    float best = 0.f, bseg = 0.f;
    long boff = -1;
    for (int it = 0; it < cap; ++it) {
        const float a_next = min3f(w.an[0], w.an[1], w.an[2]);
        const float seg = a_next - a_cur;
        const bool cx = w.an[0] <= a_next, cy = w.an[1] <= a_next, cz = w.an[2] <= a_next;
        if (REDUCE == REDUCE_SUM) {
            add(off >> 2, gl * seg);
        } else {
            const float term = vox(vol, off) * seg;
            if (term > best) {
                best = term;
                bseg = seg;
                boff = (long)(off >> 2);
            }
        }
        w.kf[0] += cx ? w.dirf[0] : 0.f;
        w.kf[1] += cy ? w.dirf[1] : 0.f;
        w.kf[2] += cz ? w.dirf[2] : 0.f;
        w.an[0] = plane_alpha(q, 0, w.kf[0]);
        w.an[1] = plane_alpha(q, 1, w.kf[1]);
        w.an[2] = plane_alpha(q, 2, w.kf[2]);
        off += (unsigned)((cx ? w.dstep[0] : 0) + (cy ? w.dstep[1] : 0) + (cz ? w.dstep[2] : 0));
        a_cur = a_next;
        if (!(a_next < q.exit)) break;
    }
    if (REDUCE == REDUCE_MAX && boff >= 0) add((unsigned)boff, gl * bseg);
}

// mask_to_channels (renderers.py:77-89): per-label line integrals of one ray.
// The ray owns its output column, so a run of segments with one label is summed
// in a register and handed to `flush(label, value)` when the label changes
// (anatomical labels are piecewise constant along a ray: a handful of flushes).
template <class Flush>
DDRR_HD void siddon_channels_ray(const float *__restrict__ vol,
                                 const unsigned char *__restrict__ labels, const Dims D,
                                 const float s[3], const float t[3], float shift, float eps,
                                 Flush flush) {
    const Box box = full_box(D);
    const SiddonSetup q = siddon_setup(box, s, t, shift, eps);
    if (!q.hit) return;
    SiddonWalk w = siddon_enter(global_store(D), box, s, shift, q);
    float a_cur = q.entry, run = 0.f;
    unsigned off = w.off;  // bytes
    int cur = -1;
    const int cap = D.x + D.y + D.z + 3;
    // Two voxels (value + label) ahead of the one being consumed are in flight: where the ray
    // goes next does not depend on what it reads, and a gather that waits for each voxel before
    // asking for the next runs at the memory latency (0.98 ms for the 119-label example render
    // at 200x200 against 0.21 ms for the plain per-ray render; profiles/r02/channels.txt).
    // A "slot" holds a requested voxel and the length of the segment it will be charged.
    struct Slot {
        float v, seg;
        int lab;
    };
    bool live = true;
    auto request = [&](Slot &sl) {
        // closes the segment that starts at a_cur, requests its voxel, steps to the next one
        const float a_next = min3f(w.an[0], w.an[1], w.an[2]);
        sl.seg = live ? a_next - a_cur : 0.f;
        sl.lab = live ? (int)labels[off >> 2] : -1;
        sl.v = live ? vox(vol, off) : 0.f;
        const bool cx = w.an[0] <= a_next, cy = w.an[1] <= a_next, cz = w.an[2] <= a_next;
        const bool more = live && a_next < q.exit;
        if (more) {
            w.kf[0] += cx ? w.dirf[0] : 0.f;
            w.kf[1] += cy ? w.dirf[1] : 0.f;
            w.kf[2] += cz ? w.dirf[2] : 0.f;
            w.an[0] = plane_alpha(q, 0, w.kf[0]);
            w.an[1] = plane_alpha(q, 1, w.kf[1]);
            w.an[2] = plane_alpha(q, 2, w.kf[2]);
            off += (unsigned)((cx ? w.dstep[0] : 0) + (cy ? w.dstep[1] : 0) + (cz ? w.dstep[2] : 0));
        }
        a_cur = a_next;
        live = more;
    };
    auto consume = [&](const Slot &sl) {
        if (sl.lab < 0) return;  // nothing was requested: the ray had ended
        if (sl.lab != cur) {
            if (cur >= 0) flush(cur, run);
            cur = sl.lab;
            run = 0.f;
        }
        run = fmaf(sl.v, sl.seg, run);
    };
    Slot s0, s1;
    request(s0);
    request(s1);
    for (int it = 0; it < cap; it += 2) {
        Slot n0, n1;
        request(n0);
        request(n1);
        consume(s0);
        consume(s1);
        s0 = n0;
        s1 = n1;
        if (s0.lab < 0) break;  // the slots ahead are empty: everything is consumed
    }
    if (cur >= 0) flush(cur, run);
}

// ---------------------------------------------------------------- generic path
// Voxel looked up from every segment's midpoint exactly as the reference does
// (renderers.py:57-60): needed for align_corners=True and for Siddon with
// mode="bilinear", where a segment does not map onto a single voxel.  Slower
// (no prefetch, more arithmetic).

struct GridMap {  // index coordinate = fma(x, k, o) per axis
    float k[3], o[3];
};

DDRR_HD GridMap make_gridmap(const Dims D, float shift, bool align_corners) {
    GridMap g;
    const int Dn[3] = {D.x, D.y, D.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // normalise 2(x+shift)/D - 1 (renderers.py:152), then aten's un-normalise:
        //   align_corners=False: ((g+1) D - 1)/2 = x + shift - 1/2
        //   align_corners=True : (g+1)/2 (D-1)   = (x + shift)(D-1)/D
        g.k[a] = align_corners ? (float)(Dn[a] - 1) / (float)Dn[a] : 1.f;
        g.o[a] = align_corners ? shift * g.k[a] : shift - 0.5f;
    }
    return g;
}

DDRR_HD float fetch_nearest(const float *__restrict__ vol, const Dims D, float gx, float gy,
                            float gz) {
    const float rx = rintf(gx), ry = rintf(gy), rz = rintf(gz);  // half-to-even == nearbyint
    const bool in = rx >= 0.f && rx < (float)D.x && ry >= 0.f && ry < (float)D.y && rz >= 0.f &&
                    rz < (float)D.z;
    if (!in) return 0.f;
    return vol[((int)rx * D.y + (int)ry) * D.z + (int)rz];
}

// aten grid_sampler_3d bilinear, padding zeros: 8 corners around floor(coord).
// The two z-neighbours are adjacent in memory and fetched as one pair.
DDRR_HD float fetch_trilinear(const float *__restrict__ vol, const Dims D, float gx, float gy,
                              float gz, float grad[3], bool want_grad) {
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    const float ax = gx - fx, ay = gy - fy, az = gz - fz;
    // keep the int conversions in range for far-away samples
    const int ix = (int)fminf(fmaxf(fx, -2.f), (float)D.x + 1.f);
    const int iy = (int)fminf(fmaxf(fy, -2.f), (float)D.y + 1.f);
    const int iz = (int)fminf(fmaxf(fz, -2.f), (float)D.z + 1.f);
    float T = 0.f, gX = 0.f, gY = 0.f, gZ = 0.f;
    const bool z0in = iz >= 0 && iz < D.z, z1in = iz + 1 >= 0 && iz + 1 < D.z;
    if (z0in || z1in) {
        const int zc = z0in ? iz : iz + 1;  // a valid z for addressing
#pragma unroll
        for (int cxy = 0; cxy < 4; ++cxy) {
            const int ox = cxy & 1, oy = cxy >> 1;
            const int x = ix + ox, y = iy + oy;
            if (x < 0 || x >= D.x || y < 0 || y >= D.y) continue;
            const float *row = vol + (x * D.y + y) * D.z;
            float v0 = 0.f, v1 = 0.f;
            if (z0in && z1in) {
                v0 = row[iz];
                v1 = row[iz + 1];
            } else if (z0in) {
                v0 = row[zc];
            } else {
                v1 = row[zc];
            }
            const float wx = ox ? ax : 1.f - ax, wy = oy ? ay : 1.f - ay;
            const float vz = fmaf(az, v1 - v0, v0);  // (1-az) v0 + az v1
            T = fmaf(wx * wy, vz, T);
            if (want_grad) {
                gX = fmaf((ox ? 1.f : -1.f) * wy, vz, gX);
                gY = fmaf((oy ? 1.f : -1.f) * wx, vz, gY);
                gZ = fmaf(wx * wy, v1 - v0, gZ);
            }
        }
    }
    if (want_grad) {
        grad[0] = gX;
        grad[1] = gY;
        grad[2] = gZ;
    }
    return T;
}

template <int REDUCE, int LOOKUP>
DDRR_HD float siddon_forward_ray_midpoint(const float *__restrict__ vol, const Dims D,
                                          const float s[3], const float t[3], float shift,
                                          float eps, bool align_corners) {
    // Walks the reference's complete sorted crossing list (all Dx+Dy+Dz+3
    // planes, including those outside the volume, whose segments sample the
    // zero padding) because with trilinear lookups or align_corners=True a
    // segment's sample is not confined to the voxel the segment lies in.
    const int Dn[3] = {D.x, D.y, D.z};
    const GridMap g = make_gridmap(D, shift, align_corners);
    float d[3], inv[3], kf[3], dirf[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;
        inv[a] = 1.0f / d[a];
        const bool pos = d[a] > 0.f;
        kf[a] = pos ? 0.f : (float)Dn[a];  // first plane met along the ray
        dirf[a] = pos ? 1.f : -1.f;
    }
    float acc = 0.f, a_cur = 0.f;
    bool have_prev = false;
    const int cap = D.x + D.y + D.z + 3;
    for (int it = 0; it < cap; ++it) {
        float an[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // planes beyond the last one of an axis never come: park them at +inf
            const bool done = kf[a] < 0.f || kf[a] > (float)Dn[a];
            an[a] = done ? INFINITY : div_refined((kf[a] - shift) - s[a], d[a], inv[a]);
        }
        const float a_next = min3f(an[0], an[1], an[2]);
        if (!(a_next < INFINITY)) break;
        if (have_prev) {
            const float mid = 0.5f * (a_cur + a_next);  // renderers.py:57
            const float gx = fmaf(fmaf(mid, d[0], s[0]), g.k[0], g.o[0]);
            const float gy = fmaf(fmaf(mid, d[1], s[1]), g.k[1], g.o[1]);
            const float gz = fmaf(fmaf(mid, d[2], s[2]), g.k[2], g.o[2]);
            float v;
            if (LOOKUP == LOOKUP_MID_TRILINEAR)
                v = fetch_trilinear(vol, D, gx, gy, gz, nullptr, false);
            else
                v = fetch_nearest(vol, D, gx, gy, gz);
            const float term = v * (a_next - a_cur);
            if (REDUCE == REDUCE_SUM)
                acc += term;
            else
                acc = fmaxf(acc, term);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) kf[a] += (an[a] <= a_next) ? dirf[a] : 0.f;
        a_cur = a_next;
        have_prev = true;
    }
    return acc;
}

// Scatter k * w_c into the 8 corners of a sample (volume gradient).
template <class Add>
DDRR_HD void scatter_trilinear(const Dims D, float gx, float gy, float gz, float k, Add add) {
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    const float ax = gx - fx, ay = gy - fy, az = gz - fz;
    const int ix = (int)fminf(fmaxf(fx, -2.f), (float)D.x + 1.f);
    const int iy = (int)fminf(fmaxf(fy, -2.f), (float)D.y + 1.f);
    const int iz = (int)fminf(fmaxf(fz, -2.f), (float)D.z + 1.f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ox = c & 1, oy = (c >> 1) & 1, oz = c >> 2;
        const int x = ix + ox, y = iy + oy, z = iz + oz;
        if (x < 0 || x >= D.x || y < 0 || y >= D.y || z < 0 || z >= D.z) continue;
        const float w = (ox ? ax : 1.f - ax) * (oy ? ay : 1.f - ay) * (oz ? az : 1.f - az);
        add((unsigned)((x * D.y + y) * D.z + z), k * w);
    }
}

template <class Add>
DDRR_HD void scatter_nearest(const Dims D, float gx, float gy, float gz, float k, Add add) {
    const float rx = rintf(gx), ry = rintf(gy), rz = rintf(gz);
    const bool in = rx >= 0.f && rx < (float)D.x && ry >= 0.f && ry < (float)D.y && rz >= 0.f &&
                    rz < (float)D.z;
    if (in) add((unsigned)(((int)rx * D.y + (int)ry) * D.z + (int)rz), k);
}

// Backward of siddon_forward_ray_midpoint (reduce sum): what autograd returns for the
// reference's midpoint formulation (renderers.py:57-71).  With segment k between crossings k and
// k+1, T_k the looked-up value at its midpoint m_k = (alpha_k + alpha_{k+1})/2 and, for the
// bilinear lookup, G_k = d T / d x there:
//   I = sum_k T_k (alpha_{k+1} - alpha_k);
//   d I = sum_c coef_c d alpha_c + sum_k seg_k G_k . ((1 - m_k) ds + m_k dt),
//   coef_c = (T_{c-1} - T_c) + (w_{c-1} + w_c) / 2,  w_k = seg_k (G_k . d)   (through d m_k),
// and d alpha_c / d s_a = (alpha_c - 1)/d_a, d alpha_c / d t_a = -alpha_c/d_a on the crossing's
// own axis (tied crossings: one axis, x before y before z).  gl = grad_out * ray length;
// I is returned for d out / d img; `add(flat voxel index, value)` scatters the volume gradient
// (the 8 corner weights for the bilinear lookup).
template <int LOOKUP, bool WANT_VOL, class Add>
DDRR_HD float siddon_backward_ray_midpoint(const float *__restrict__ vol, const Dims D,
                                           const float s[3], const float t[3], float shift,
                                           float eps, bool align_corners, float gl, float gs[3],
                                           float gt[3], Add add) {
    const int Dn[3] = {D.x, D.y, D.z};
    const GridMap g = make_gridmap(D, shift, align_corners);
    float d[3], inv[3], kf[3], dirf[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = (t[a] - s[a]) + eps;
        inv[a] = 1.0f / d[a];
        const bool pos = d[a] > 0.f;
        kf[a] = pos ? 0.f : (float)Dn[a];
        dirf[a] = pos ? 1.f : -1.f;
    }
    float S0[3] = {0.f, 0.f, 0.f}, S1[3] = {0.f, 0.f, 0.f};
    float Es[3] = {0.f, 0.f, 0.f}, Et[3] = {0.f, 0.f, 0.f};
    float I = 0.f, a_cur = 0.f, T_prev = 0.f, w_prev = 0.f;
    int ax_open = 0;
    bool have_prev = false;
    const int cap = D.x + D.y + D.z + 3;
    for (int it = 0; it < cap; ++it) {
        float an[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const bool done = kf[a] < 0.f || kf[a] > (float)Dn[a];
            an[a] = done ? INFINITY : div_refined((kf[a] - shift) - s[a], d[a], inv[a]);
        }
        const float a_next = min3f(an[0], an[1], an[2]);
        if (!(a_next < INFINITY)) break;
        if (have_prev) {
            const float seg = a_next - a_cur, mid = 0.5f * (a_cur + a_next);
            const float gx = fmaf(fmaf(mid, d[0], s[0]), g.k[0], g.o[0]);
            const float gy = fmaf(fmaf(mid, d[1], s[1]), g.k[1], g.o[1]);
            const float gz = fmaf(fmaf(mid, d[2], s[2]), g.k[2], g.o[2]);
            float T, w = 0.f;
            if (LOOKUP == LOOKUP_MID_TRILINEAR) {
                float dT[3];
                T = fetch_trilinear(vol, D, gx, gy, gz, dT, true);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float G = dT[a] * g.k[a];  // d T / d x_a
                    w = fmaf(G, d[a], w);
                    Es[a] = fmaf(seg * (1.f - mid), G, Es[a]);
                    Et[a] = fmaf(seg * mid, G, Et[a]);
                }
                w *= seg;
                if (WANT_VOL) scatter_trilinear(D, gx, gy, gz, gl * seg, add);
            } else {
                T = fetch_nearest(vol, D, gx, gy, gz);
                if (WANT_VOL) scatter_nearest(D, gx, gy, gz, gl * seg, add);
            }
            I = fmaf(T, seg, I);
            // the crossing that opened this segment
            const float coef = (T_prev - T) + 0.5f * (w_prev + w);
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (a == ax_open) {
                    S0[a] += coef;
                    S1[a] = fmaf(coef, a_cur, S1[a]);
                }
            T_prev = T;
            w_prev = w;
        }
        ax_open = an[0] <= a_next ? 0 : (an[1] <= a_next ? 1 : 2);
#pragma unroll
        for (int a = 0; a < 3; ++a) kf[a] += (an[a] <= a_next) ? dirf[a] : 0.f;
        a_cur = a_next;
        have_prev = true;
    }
    // the last crossing closes the last segment
    const float coef = T_prev + 0.5f * w_prev;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (a == ax_open && have_prev) {
            S0[a] += coef;
            S1[a] = fmaf(coef, a_cur, S1[a]);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gs[a] = gl * fmaf(S1[a] - S0[a], inv[a], Es[a]);
        gt[a] = gl * fmaf(-S1[a], inv[a], Et[a]);
    }
    return I;
}

}  // namespace ddrr
