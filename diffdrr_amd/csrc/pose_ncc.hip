// pose_ncc.hip -- the small kernels either side of the renderers on the DRR path: Euler pose ->
// world matrix, fused ray generation, the pose-gradient reduction, normalised cross-correlation.
#include "runtime.h"
#include "siddon_core.h"
#include "raygen_core.h"
#include "record_pack.h"
#include "record_layout.h"
#include "ncc_patch_core.h"
#include "sobel_core.h"
#include "blur_core.h"

using namespace ddrr;
using namespace ddrr_rt;

namespace {

// ------------------------------------------------ fused ray generation (DRR case)

// source_v (B,3), target_v (B,N,3), img (B,N) from one world pose per DRR (raygen_core.h)
__global__ __launch_bounds__(kBlock) void raygen_fwd_kernel(
    const float *__restrict__ Mw, const float *__restrict__ Ainv, const float *__restrict__ P,
    int N, float *__restrict__ source_v, float *__restrict__ target_v, float *__restrict__ img) {
    const int b = blockIdx.y, n = blockIdx.x * kBlock + threadIdx.x;
    const float *M = Mw + (long)b * 12;
    if (n == 0) {
        const float sw[3] = {M[3], M[7], M[11]};
        float sv[3];
        apply34(Ainv, sw, sv);
        source_v[b * 3 + 0] = sv[0];
        source_v[b * 3 + 1] = sv[1];
        source_v[b * 3 + 2] = sv[2];
    }
    if (n >= N) return;
    const float Pn[3] = {P[n * 3], P[n * 3 + 1], P[n * 3 + 2]};
    const RayGenOut o = raygen_ray(M, Ainv, Pn);
    const long r = (long)b * N + n;
    target_v[r * 3 + 0] = o.tv[0];
    target_v[r * 3 + 1] = o.tv[1];
    target_v[r * 3 + 2] = o.tv[2];
    img[r] = o.L;
}

// dLoss/dMw (B,3,4) from the forward's backward record: the renderer's per-ray endpoint
// gradients (siddon_backward_ray) are chained through the ray generation and reduced per
// pose inside the kernel; no per-ray gradient tensor is written.  A block covers
// kPoseRaysPerBlock rays of one pose and adds its 12 partial sums with atomics.
constexpr int kPoseRaysPerBlock = 4096;

__global__ __launch_bounds__(kBlock) void siddon_bwd_pose_kernel(
    const float *__restrict__ aux, int layout, const float *__restrict__ grad_out,
    const float *__restrict__ source_v, const float *__restrict__ target_v,
    const float *__restrict__ img, const float *__restrict__ Mw, const float *__restrict__ Ainv,
    const float *__restrict__ P, int B, int N, float eps, int with_img_path,
    float *__restrict__ gMw) {
    __shared__ float red[kWavesPerBlock][12];
    const int b = blockIdx.y;
    const long R = (long)B * N;
    const float *M = Mw + (long)b * 12;
    const float s[3] = {source_v[b * 3], source_v[b * 3 + 1], source_v[b * 3 + 2]};
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    const int n_end = min(N, (int)(blockIdx.x + 1) * kPoseRaysPerBlock);
    for (int n = blockIdx.x * kPoseRaysPerBlock + threadIdx.x; n < n_end; n += kBlock) {
        const long r = (long)b * N + n;
        float rec[SIDDON_AUX];
        if (layout == DDRR_AUX_BLOCKED) {
            rec_blocked_load(aux, r, rec);
        } else if (layout == DDRR_AUX_PACKED) {
            float S0x, S0z, S1x, S1z;
            const long long *X = reinterpret_cast<const long long *>(aux);
            const float q = aux[6 * R], qa = q / aux[5 * R + r];
            record_unpack(X[r], q, qa, S0x, S1x);
            record_unpack(X[R + r], q, qa, S0z, S1z);
            const float I = aux[4 * R + r];
            rec[0] = I, rec[1] = S0x, rec[2] = -(S0x + S0z), rec[3] = S0z;
            rec[4] = S1x, rec[5] = I - (S1x + S1z), rec[6] = S1z, rec[7] = 0.f;
        } else {
            const float4 *a4 = reinterpret_cast<const float4 *>(aux + r * SIDDON_AUX);
            const float4 lo = a4[0], hi = a4[1];
            rec[0] = lo.x, rec[1] = lo.y, rec[2] = lo.z, rec[3] = lo.w;
            rec[4] = hi.x, rec[5] = hi.y, rec[6] = hi.z, rec[7] = hi.w;
        }
        const float t[3] = {target_v[r * 3], target_v[r * 3 + 1], target_v[r * 3 + 2]};
        const float Pn[3] = {P[n * 3], P[n * 3 + 1], P[n * 3 + 2]};
        const float g = grad_out[r], L = img[r];
        float gs[3], gt[3];
        siddon_backward_ray<REDUCE_SUM>(rec, s, t, eps, g * L, gs, gt);
        raygen_ray_adjoint(M, Ainv, Pn, gt, gs, with_img_path ? g * rec[0] : 0.f, L, acc);
    }
    // 12 sums over the block: wave butterflies, then the 4 waves through LDS
#pragma unroll
    for (int k = 0; k < 12; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) v += red[w][threadIdx.x];
        unsafeAtomicAdd(gMw + (long)b * 12 + threadIdx.x, v);
    }
}

// -------------------------------------------------- Euler pose -> world matrix
__global__ __launch_bounds__(kBlock) void pose_euler_fwd_kernel(
    const float *__restrict__ rot, const float *__restrict__ xyz, int a0, int a1, int a2,
    const float *__restrict__ Ro, int B, float *__restrict__ Mw) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const float th[3] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2]};
    const float t[3] = {xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
    const int axes[3] = {a0, a1, a2};
    float M[12];
    pose_euler_forward(th, t, axes, Ro, M);
#pragma unroll
    for (int k = 0; k < 12; ++k) Mw[b * 12 + k] = M[k];
}

__global__ __launch_bounds__(kBlock) void pose_euler_bwd_kernel(
    const float *__restrict__ rot, const float *__restrict__ xyz, int a0, int a1, int a2,
    const float *__restrict__ Ro, const float *__restrict__ gMw, int B, float *__restrict__ g_rot,
    float *__restrict__ g_xyz) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const float th[3] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2]};
    const float t[3] = {xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
    const int axes[3] = {a0, a1, a2};
    float g[12], gt[3], gx[3];
#pragma unroll
    for (int k = 0; k < 12; ++k) g[k] = gMw[b * 12 + k];
    pose_euler_backward(th, t, axes, Ro, g, gt, gx);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g_rot[b * 3 + k] = gt[k];
        g_xyz[b * 3 + k] = gx[k];
    }
}


// ----------------------------------- the registration / sweep step around the renderer, fused
// A registration iteration at one pose is ~0.19 ms of brick kernel and -- as separate launches:
// pose -> matrix, matrix -> rays, record -> image, NCC, its backward, rays -> dL/dMw, dL/dMw ->
// dL/d(rot, xyz), plus two zero-fills -- 0.06 ms of small kernels at ~4.4 us each
// (profiles/r04/bench_config_4.json).  Three kernels do the same arithmetic, in the same order
// per element:
//   pose_raygen_fwd_kernel      = pose_euler_fwd_kernel + raygen_fwd_kernel (every workgroup works
//                                 its pose's matrix out again: three sincos, 30 fma);
//   siddon_ncc_fwd_kernel       = siddon_out_from_record_kernel + ncc_fwd_kernel, many workgroups per
//                                 pair: moments by double atomics into the caller's workspace, the
//                                 LAST workgroup of a pair (a ticket) finishes its statistics;
//   siddon_ncc_bwd_pose_kernel  = ncc_bwd_kernel + siddon_bwd_pose_kernel + pose_euler_bwd_kernel: the
//                                 image gradient is formed per ray from the statistics, chained
//                                 through the record and the ray generation, reduced per pose; the
//                                 last workgroup of a pose takes dL/dMw on to the pose parameters.
// The workspace (moments, dL/dMw accumulators, tickets) is caller-owned, zero when it is first
// handed over, and left zero by every call: no fills.  Reference: diffdrr/pose.py:140-190,
// detector.py:144-154, drr.py:201-205, metrics.py:21-44 and their autograd.
// Rays of one pose per workgroup of the two epilogues, by launch size (template parameters): a launch
// of many poses is bound by its bytes and wants them in flight -- forward 4096 rays per workgroup
// (four 16-byte triples per thread: 27 -> 16 us at 32 poses), backward 2048 (eight rays per thread:
// 33.5 -> 28 us) --, a launch of one pose is bound by its latency and wants workgroups (1024 rays
// each, as in round 5: with the large ones a registration iteration went 0.175 -> 0.180 ms).
constexpr int kStepWorkgroupsWanted = 512;  // two per CU before the large workgroups pay

struct NccWs {
    double *mom;   // [B][5]  sum x1, x2, x1^2, x2^2, x1 x2
    double *total; // sum of the pairs' NCC (ncc_sum)
    float *gacc;   // [B][12] dLoss/dMw
    int *tick1, *tick2;  // [B] each
    int *tick_all;       // pairs that have added their NCC to `total`
};
__host__ __device__ inline NccWs ncc_ws(void *ws, int B) {
    NccWs w;
    w.mom = reinterpret_cast<double *>(ws);
    w.total = w.mom + 5 * (long)B;
    w.gacc = reinterpret_cast<float *>(w.total + 1);
    w.tick1 = reinterpret_cast<int *>(w.gacc + 12 * (long)B);
    w.tick2 = w.tick1 + B;
    w.tick_all = w.tick2 + B;
    return w;
}

// The last workgroup of a pose (a ticket).  No agent-scope fence: on this part it writes the
// XCD's whole L2 back, per wave that executes it -- tens of microseconds with the record just
// written (measured: the first form of these kernels, with __threadfence(), ran 2x the launches it
// fuses).  What has to be ordered are atomics only: every accumulator update is a device-scope
// atomic, performed at the coherence point; the wave that issued them waits until they have been
// performed before the workgroup takes its ticket, and the last workgroup reads the accumulators
// with atomics again.  The wait is spelled out: the accumulator updates are no-return atomics, counted
// by vmcnt, and a workgroup-scope release fence lowers to `s_waitcnt lgkmcnt(0)` alone on gfx950
// (checked in the ISA) -- without `vmcnt(0)` the ticket, an atomic to another address and
// possibly another channel, could be performed before the sums it announces.
__device__ __forceinline__ bool last_workgroup_of_pose(int *ticket, int *shared_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's atomics have been performed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (and its LDS traffic, for the barrier)
    __syncthreads();
    if (threadIdx.x == 0) *shared_flag = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    __syncthreads();
    return *shared_flag != 0;
}

__global__ __launch_bounds__(kBlock) void pose_raygen_fwd_kernel(
    const float *__restrict__ rot, const float *__restrict__ xyz, int a0, int a1, int a2,
    const float *__restrict__ Ro, const float *__restrict__ Ainv, const float *__restrict__ P, int N,
    float *__restrict__ Mw, float *__restrict__ source_v, float *__restrict__ target_v,
    float *__restrict__ img, float *__restrict__ clear, long clear_n, int *__restrict__ counter) {
    __shared__ float Ms[12];
    const int b = blockIdx.y, n = blockIdx.x * kBlock + threadIdx.x;
    // what the render behind this launch has to find zeroed (its record or image, its brick
    // counter): cleared here instead of by a launch of its own
    if (clear) {
        const long stride = (long)gridDim.x * gridDim.y * kBlock, n4 = clear_n >> 2;
        float4 *c4 = reinterpret_cast<float4 *>(clear);
        const long first = ((long)blockIdx.y * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        for (long i = first; i < n4; i += stride) c4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (first < (clear_n & 3)) clear[(n4 << 2) + first] = 0.f;
    }
    if (counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) counter[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        // (once per workgroup: three sincos and a few dozen fma)
        const float th[3] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2]};
        const float t[3] = {xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
        const int axes[3] = {a0, a1, a2};
        float M[12];
        pose_euler_forward(th, t, axes, Ro, M);
#pragma unroll
        for (int k = 0; k < 12; ++k) Ms[k] = M[k];
        if (blockIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) Mw[b * 12 + k] = M[k];
            const float sw[3] = {M[3], M[7], M[11]};
            float sv[3];
            apply34(Ainv, sw, sv);
            source_v[b * 3 + 0] = sv[0];
            source_v[b * 3 + 1] = sv[1];
            source_v[b * 3 + 2] = sv[2];
        }
    }
    __syncthreads();
    if (n >= N) return;
    float M[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) M[k] = Ms[k];
    const float Pn[3] = {P[n * 3], P[n * 3 + 1], P[n * 3 + 2]};
    const RayGenOut o = raygen_ray(M, Ainv, Pn);
    const long r = (long)b * N + n;
    target_v[r * 3 + 0] = o.tv[0];
    target_v[r * 3 + 1] = o.tv[1];
    target_v[r * 3 + 2] = o.tv[2];
    img[r] = o.L;
}

template <int kNccFwdQuads>
__global__ __launch_bounds__(kBlock) void siddon_ncc_fwd_kernel(
    const float *__restrict__ aux, const float *__restrict__ img, const float *__restrict__ x1,
    long x1_stride, int B, int N, float eps, void *ws_raw, float *__restrict__ ncc_out,
    float *__restrict__ stats, float *__restrict__ out, float *__restrict__ ncc_sum) {
    __shared__ double red[5][kWavesPerBlock];
    __shared__ int last;
    const NccWs ws = ncc_ws(ws_raw, B);
    const int b = blockIdx.y;
    double m[5] = {0., 0., 0., 0., 0.};
    auto take = [&](float a, float c) {
        const double da = (double)a, dc = (double)c;
        m[0] += da;
        m[1] += dc;
        m[2] = fma(da, da, m[2]);
        m[3] = fma(dc, dc, m[3]);
        m[4] = fma(da, dc, m[4]);
    };
    const float *p1 = x1 + b * x1_stride, *pl = img + (long)b * N;
    const long r0 = (long)b * N;
    constexpr int kNccFwdRaysPerBlock = 4 * kNccFwdQuads * kBlock;
    const int n0 = blockIdx.x * kNccFwdRaysPerBlock, n_end = min(N, n0 + kNccFwdRaysPerBlock);
    // four consecutive rays per load: plane I of rays 4 k .. 4 k + 3 is one aligned 16-byte run of
    // the blocked record (record_layout.h), like the rays' lengths and the fixed image's pixels.
    // A thread takes kNccFwdQuads such quads, ALL requested before the first is summed (twelve
    // 16-byte loads in flight per thread, two workgroups per CU and pose at 256^2: the kernel
    // streams 12 B per ray and round 5's one quad per thread -- then the reduction tail, then a
    // ticket, in four times as many workgroups -- left it at 1 TB/s: 24 us for 25 MB).
    const bool vec = (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(pl) |
                                       reinterpret_cast<uintptr_t>(aux)) & 15) == 0 &&
                     (!out || (reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (vec) {
        float4 I4[kNccFwdQuads], L4[kNccFwdQuads], a4[kNccFwdQuads];
        bool in[kNccFwdQuads];
#pragma unroll
        for (int k = 0; k < kNccFwdQuads; ++k) {
            const int n_k = n0 + 4 * ((int)threadIdx.x + k * kBlock);
            in[k] = n_k < n_end;
            const int n = in[k] ? n_k : n0;  // (a quad beyond the image re-reads the first with weight 0)
            I4[k] = *reinterpret_cast<const float4 *>(aux + rec_index(r0 + n, 0));
            L4[k] = *reinterpret_cast<const float4 *>(pl + n);
            a4[k] = *reinterpret_cast<const float4 *>(p1 + n);
        }
#pragma unroll
        for (int k = 0; k < kNccFwdQuads; ++k) {
            if (!in[k]) continue;
            const float4 x4 = make_float4(L4[k].x * I4[k].x, L4[k].y * I4[k].y, L4[k].z * I4[k].z, L4[k].w * I4[k].w);
            if (out) *reinterpret_cast<float4 *>(out + r0 + n0 + 4 * ((int)threadIdx.x + k * kBlock)) = x4;
            take(a4[k].x, x4.x);
            take(a4[k].y, x4.y);
            take(a4[k].z, x4.z);
            take(a4[k].w, x4.w);
        }
    } else {
        for (int n = n0 + threadIdx.x; n < n_end; n += kBlock) {
            const float x2 = pl[n] * aux[rec_index(r0 + n, 0)];  // (= siddon_out_from_record_kernel)
            if (out) out[r0 + n] = x2;
            take(p1[n], x2);
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double v = m[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double v = 0.;
        for (int w = 0; w < kWavesPerBlock; ++w) v += red[threadIdx.x][w];
        atomicAdd(ws.mom + b * 5 + threadIdx.x, v);
    }
    if (!last_workgroup_of_pose(ws.tick1 + b, &last)) return;
    // (the five sums by five lanes: one round trip to the coherence point, not five in a row)
    if (threadIdx.x < 5) {
        red[threadIdx.x][0] = atomicAdd(ws.mom + b * 5 + threadIdx.x, 0.0);  // (a coherent read)
        ws.mom[b * 5 + threadIdx.x] = 0.0;                                    // left zero for the next call
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double t[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = red[k][0];
    ws.tick1[b] = 0;
    const double inv_n = 1.0 / (double)N;
    const double mu1 = t[0] * inv_n, mu2 = t[1] * inv_n;
    const double v1 = t[2] * inv_n - mu1 * mu1, v2 = t[3] * inv_n - mu2 * mu2;
    const double c12 = t[4] * inv_n - mu1 * mu2;
    const float s1 = sqrtf((float)v1 + eps), s2 = sqrtf((float)v2 + eps);
    const float ncc = (float)c12 / (s1 * s2);
    ncc_out[b] = ncc;
    stats[b * 5 + 0] = (float)mu1;
    stats[b * 5 + 1] = s1;
    stats[b * 5 + 2] = (float)mu2;
    stats[b * 5 + 3] = s2;
    stats[b * 5 + 4] = ncc;
    if (ncc_sum) {
        // the batch's objective: every pair adds its value, the last pair to do so hands the sum over
        // (same ordering rule as last_workgroup_of_pose: the addition has been performed before the
        // ticket is taken) and leaves the two words zero for the next call
        atomicAdd(ws.total, (double)ncc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(ws.tick_all, 1) == B - 1) {
            *ncc_sum = (float)atomicAdd(ws.total, 0.0);  // (a coherent read)
            *ws.total = 0.0;
            *ws.tick_all = 0;
        }
    }
}

// EXPLICIT: the per-pixel gradient of ANY image-space objective arrives ready-made in `x1` ((B, N),
// x1_stride = N; stats and g_out unused) instead of being formed from the NCC statistics:
// ddrr_siddon_backward_pose_euler, the differentiable `drr(rot, xyz, parameterization="euler_angles")`.
template <int kStepRaysPerBlock, bool EXPLICIT = false>
__global__ __launch_bounds__(kBlock) void siddon_ncc_bwd_pose_kernel(
    const float *__restrict__ aux, const float *__restrict__ img, const float *__restrict__ x1,
    long x1_stride, const float *__restrict__ stats, const float *__restrict__ g_out, int g_stride,
    const float *__restrict__ source_v, const float *__restrict__ target_v,
    const float *__restrict__ Mw, const float *__restrict__ Ainv, const float *__restrict__ P,
    const float *__restrict__ rot, const float *__restrict__ xyz, int a0, int a1, int a2,
    const float *__restrict__ Ro, int B, int N, float eps, int with_img_path, void *ws_raw,
    float *__restrict__ g_rot, float *__restrict__ g_xyz) {
    __shared__ float red[kWavesPerBlock][12];
    __shared__ int last;
    const NccWs ws = ncc_ws(ws_raw, B);
    const int b = blockIdx.y;
    const float *M = Mw + (long)b * 12;
    const float s[3] = {source_v[b * 3], source_v[b * 3 + 1], source_v[b * 3 + 2]};
    float mu1 = 0.f, s1 = 1.f, mu2 = 0.f, s2 = 1.f, ncc = 0.f, gn = 0.f;
    if constexpr (!EXPLICIT) {
        mu1 = stats[b * 5], s1 = stats[b * 5 + 1], mu2 = stats[b * 5 + 2];
        s2 = stats[b * 5 + 3], ncc = stats[b * 5 + 4];
        gn = g_out[b * g_stride] / (float)N;
    }
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    // a thread's rays of the workgroup's kStepRaysPerBlock: everything they read is requested before
    // the first is worked on (one round trip to memory, not one per ray); a ray beyond the image
    // re-reads the last one with weight 0
    constexpr int kPer = kStepRaysPerBlock / kBlock;
    const int n_end = min(N, (int)(blockIdx.x + 1) * kStepRaysPerBlock);
    // (a ray's voxel-space target and length are NOT read back -- 16 of the 52 bytes per ray this kernel
    // streamed: they are regenerated from the pose's matrix and the detector point, which is read
    // anyway, by the very function that wrote them, raygen_core.h raygen_ray: the same bits)
    float rec[kPer][SIDDON_AUX], x1s[kPer], Ps[kPer][3];
    bool in[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int n0 = blockIdx.x * kStepRaysPerBlock + threadIdx.x + k * kBlock;
        in[k] = n0 < n_end;
        const int n = in[k] ? n0 : n_end - 1;
        const long r = (long)b * N + n;
        rec_blocked_load(aux, r, rec[k]);
        x1s[k] = x1[b * x1_stride + n];
#pragma unroll
        for (int a = 0; a < 3; ++a) Ps[k][a] = P[n * 3 + a];
    }
    (void)img;
    (void)target_v;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const RayGenOut ray = raygen_ray(M, Ainv, Ps[k]);
        const float L = ray.L;
        // d ncc / d x2[n] (= ncc_bwd_kernel on x2 = L I)
        float g;
        if constexpr (EXPLICIT) {
            g = in[k] ? x1s[k] : 0.f;
        } else {
            const float z1 = (x1s[k] - mu1) / s1, z2 = (L * rec[k][0] - mu2) / s2;
            g = in[k] ? gn * (z1 - z2 * ncc) / s2 : 0.f;
        }
        float gs[3], gt[3];
        siddon_backward_ray<REDUCE_SUM>(rec[k], s, ray.tv, eps, g * L, gs, gt);
        raygen_ray_adjoint(M, Ainv, Ps[k], gt, gs, with_img_path ? g * rec[k][0] : 0.f, L, acc);
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) v += red[w][threadIdx.x];
        unsafeAtomicAdd(ws.gacc + (long)b * 12 + threadIdx.x, v);
    }
    if (!last_workgroup_of_pose(ws.tick2 + b, &last)) return;
    // the last workgroup of the pose: dLoss/dMw -> dLoss/d(rot, xyz) (= pose_euler_bwd_kernel)
    // (the twelve sums by twelve lanes: one round trip to the coherence point, not twelve in a row)
    if (threadIdx.x < 12) {
        red[0][threadIdx.x] = atomicAdd(ws.gacc + (long)b * 12 + threadIdx.x, 0.f);  // (a coherent read)
        ws.gacc[(long)b * 12 + threadIdx.x] = 0.f;                                    // left zero for the next call
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float g[12], gth[3], gx[3];
#pragma unroll
    for (int k = 0; k < 12; ++k) g[k] = red[0][k];
    ws.tick2[b] = 0;
    const float th[3] = {rot[b * 3], rot[b * 3 + 1], rot[b * 3 + 2]};
    const float tr[3] = {xyz[b * 3], xyz[b * 3 + 1], xyz[b * 3 + 2]};
    const int axes[3] = {a0, a1, a2};
    pose_euler_backward(th, tr, axes, Ro, g, gth, gx);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g_rot[b * 3 + k] = gth[k];
        g_xyz[b * 3 + k] = gx[k];
    }
}

// ------------------------------------------------- fused NCC (sweep / registration)
// NormalizedCrossCorrelation2d with patch_size = None (reference metrics.py:21-44):
// ncc_b = mean(z1 * z2), z = (x - mean) / sqrt(var + eps), one value per image pair.
// One workgroup per pair, two passes over the pair (means, then centred moments: no
// cancellation); stats[b] = {mu1, s1, mu2, s2, ncc}.  x1 may be one image shared by the
// whole batch (x1_stride = 0).
constexpr int kNccThreads = 1024;

// One pass over the pair: the five raw moments are accumulated in double (no cancellation
// worth mentioning: 1e-16 mu^2 / var), 16-byte loads where the rows allow it, one block
// reduction of all five.
__global__ __launch_bounds__(kNccThreads) void ncc_fwd_kernel(
    const float *__restrict__ x1, long x1_stride, const float *__restrict__ x2, int N, float eps,
    float *__restrict__ out, float *__restrict__ stats) {
    __shared__ double red[5][kNccThreads / 64];
    const int b = blockIdx.x;
    const float *p1 = x1 + b * x1_stride, *p2 = x2 + (long)b * N;
    double m[5] = {0., 0., 0., 0., 0.};  // sum x1, x2, x1^2, x2^2, x1 x2
    auto take = [&](float a, float c) {
        const double da = (double)a, dc = (double)c;
        m[0] += da;
        m[1] += dc;
        m[2] = fma(da, da, m[2]);
        m[3] = fma(dc, dc, m[3]);
        m[4] = fma(da, dc, m[4]);
    };
    const bool vec = (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0;
    if (vec) {
        const float4 *q1 = reinterpret_cast<const float4 *>(p1), *q2 = reinterpret_cast<const float4 *>(p2);
        for (int n = threadIdx.x; n < (N >> 2); n += kNccThreads) {
            const float4 a = q1[n], c = q2[n];
            take(a.x, c.x);
            take(a.y, c.y);
            take(a.z, c.z);
            take(a.w, c.w);
        }
    } else {
        for (int n = threadIdx.x; n < N; n += kNccThreads) take(p1[n], p2[n]);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double v = m[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            t[k] = 0.;
            for (int w = 0; w < kNccThreads / 64; ++w) t[k] += red[k][w];
        }
        const double inv_n = 1.0 / (double)N;
        const double mu1 = t[0] * inv_n, mu2 = t[1] * inv_n;
        const double v1 = t[2] * inv_n - mu1 * mu1, v2 = t[3] * inv_n - mu2 * mu2;
        const double c12 = t[4] * inv_n - mu1 * mu2;
        const float s1 = sqrtf((float)v1 + eps), s2 = sqrtf((float)v2 + eps);
        const float ncc = (float)c12 / (s1 * s2);
        out[b] = ncc;
        stats[b * 5 + 0] = (float)mu1;
        stats[b * 5 + 1] = s1;
        stats[b * 5 + 2] = (float)mu2;
        stats[b * 5 + 3] = s2;
        stats[b * 5 + 4] = ncc;
    }
}

// d ncc / d x2[n] = (z1[n] - z2[n] ncc) / (N s2), and symmetrically for x1 (per pair; a
// shared x1 gets no gradient from this kernel).
__global__ __launch_bounds__(kBlock) void ncc_bwd_kernel(
    const float *__restrict__ x1, long x1_stride, const float *__restrict__ x2,
    const float *__restrict__ stats, const float *__restrict__ g_out, int g_stride, int N,
    float *__restrict__ g_x1, float *__restrict__ g_x2) {
    const int b = blockIdx.y, n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float mu1 = stats[b * 5], s1 = stats[b * 5 + 1], mu2 = stats[b * 5 + 2];
    const float s2 = stats[b * 5 + 3], ncc = stats[b * 5 + 4];
    const float z1 = (x1[b * x1_stride + n] - mu1) / s1, z2 = (x2[(long)b * N + n] - mu2) / s2;
    const float g = g_out[b * g_stride] / (float)N;
    if (g_x2) g_x2[(long)b * N + n] = g * (z1 - z2 * ncc) / s2;
    if (g_x1) g_x1[(long)b * N + n] = g * (z2 - z1 * ncc) / s1;
}


// ------------------------------------------------- patch-wise NCC (multiscale NCC)
// NormalizedCrossCorrelation2d(patch_size = p), reference metrics.py:16-44 (ncc_patch_core.h has the
// formulas).  Forward: a workgroup takes 16 x 16 windows of one pair, stages the (16 + p - 1)^2
// pixels they cover of both images in LDS, every thread z-scores its own window in two passes and
// -- for the backward -- writes the window's four coefficients; the pair's score is the mean of
// the windows' values (block sums, one atomic per workgroup; the entry zeroes the B floats).
constexpr int kPatchTile = 16;
// LDS row strides: a ds_read_b32 lane group is two rows of 16 windows on 32 banks, a ds_read_b128 one (the
// backward's coefficients) rows of 16-byte slots on 64: strides of 16 mod 32 floats / a multiple of 16 float4
// keep the rows of a group on different banks (T = 28 at p = 13 has them collide: every read 2 cycles for 1).
__host__ __device__ inline int patch_stride_fwd(int T) { return ((T + 15) / 32) * 32 + 16; }
__host__ __device__ inline int patch_stride_bwd(int T) { return (T + 15) & ~15; }
// (workgroups per pair: every tile its own up to 2048 workgroups a launch, beyond that a workgroup walks
// several tiles -- its one atomic on the pair's float is what the launch waits for at 32 pairs x 256 tiles)
constexpr int kPatchWorkgroupsWanted = 2048, kPatchWorkgroupsPerPairMin = 32;

template <int P>
__global__ __launch_bounds__(kPatchTile *kPatchTile) void ncc_patch_fwd_kernel(
    const float *__restrict__ x1, long x1_stride, const float *__restrict__ x2, int H, int W, int p_rt,
    float eps, float *__restrict__ ncc_out, float4 *__restrict__ coef) {
    const int p = P > 0 ? P : p_rt;
    extern __shared__ float patch_lds[];
    const int T = kPatchTile + p - 1, S = patch_stride_fwd(T);
    float *ta = patch_lds, *tb = patch_lds + T * S;
    __shared__ float red[kPatchTile * kPatchTile / 64];
    const int b = blockIdx.z, hw = H - p + 1, ww = W - p + 1;
    const int ntx = (ww + kPatchTile - 1) / kPatchTile, nty = (hw + kPatchTile - 1) / kPatchTile;
    const float *a = x1 + b * x1_stride, *m = x2 + (long)b * H * W;
    const int ly = threadIdx.x / kPatchTile, lx = threadIdx.x % kPatchTile;
    float v = 0.f;
    for (int tile = blockIdx.x; tile < ntx * nty; tile += gridDim.x) {
        const int y0 = (tile / ntx) * kPatchTile, x0 = (tile % ntx) * kPatchTile;
        for (int i = threadIdx.x; i < T * T; i += kPatchTile * kPatchTile) {
            const int r = i / T, c = i - r * T;
            const int y = y0 + r, x = x0 + c;
            const bool in = y < H && x < W;
            ta[r * S + c] = in ? a[(long)y * W + x] : 0.f;
            tb[r * S + c] = in ? m[(long)y * W + x] : 0.f;
        }
        __syncthreads();
        const int wy = y0 + ly, wx = x0 + lx;
        if (wy < hw && wx < ww) {
            float c[4];
            const float *pa = ta + ly * S + lx, *pb = tb + ly * S + lx;
            v += ncc_patch_window<P>([&](int y, int x) { return pa[y * S + x]; },
                                     [&](int y, int x) { return pb[y * S + x]; }, p, eps, c);
            if (coef) coef[((long)b * hw + wy) * ww + wx] = make_float4(c[0], c[1], c[2], c[3]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kPatchTile * kPatchTile / 64; ++w) t += red[w];
        unsafeAtomicAdd(ncc_out + b, t / ((float)hw * (float)ww));
    }
}

// Backward: a workgroup takes 16 x 16 pixels of one pair and stages the coefficients of the
// (16 + p - 1)^2 windows that hold any of them (zeros outside the window grid); a thread adds up
// its pixel's p^2 windows.
template <int P>
__global__ __launch_bounds__(kPatchTile *kPatchTile) void ncc_patch_bwd_kernel(
    const float *__restrict__ x1, long x1_stride, const float *__restrict__ x2,
    const float4 *__restrict__ coef, const float *__restrict__ g_out, int g_stride, int H, int W, int p_rt,
    float *__restrict__ g_x2) {
    const int p = P > 0 ? P : p_rt;
    extern __shared__ float4 coef_lds[];
    const int T = kPatchTile + p - 1, S = patch_stride_bwd(T);
    const int b = blockIdx.z, hw = H - p + 1, ww = W - p + 1;
    const int y0 = blockIdx.y * kPatchTile, x0 = blockIdx.x * kPatchTile;
    for (int i = threadIdx.x; i < T * T; i += kPatchTile * kPatchTile) {
        const int r = i / T, c = i - r * T;
        const int wy = y0 - (p - 1) + r, wx = x0 - (p - 1) + c;
        const bool in = wy >= 0 && wx >= 0 && wy < hw && wx < ww;
        coef_lds[r * S + c] = in ? coef[((long)b * hw + wy) * ww + wx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int ly = threadIdx.x / kPatchTile, lx = threadIdx.x % kPatchTile;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) return;
    // (tile coordinates: window (wy, wx) sits at (wy - y0 + p - 1, wx - x0 + p - 1))
    const float4 *c0 = coef_lds + (p - 1 - y0) * S + (p - 1 - x0);
    const float a = x1[b * x1_stride + (long)y * W + x], m = x2[((long)b * H + y) * W + x];
    const float s = ncc_patch_pixel_grad<P>(
        [&](int wy, int wx, int k) {
            const float4 c = c0[wy * S + wx];
            return k == 0 ? c.x : (k == 1 ? c.y : (k == 2 ? c.z : c.w));
        },
        y, x, p, a, m);
    const float g = g_out[b * g_stride] / ((float)hw * (float)ww * (float)(p * p));
    g_x2[((long)b * H + y) * W + x] = g * s;
}

// ------------------------------------------------- Sobel pair (gradient NCC)
// out (B, 2, H, W) = {Gx, Gy} * img (B, H, W), zero padding (sobel_core.h); one pixel per thread
__global__ __launch_bounds__(kBlock) void sobel_fwd_kernel(const float *__restrict__ img, int H,
                                                           int W, float *__restrict__ out) {
    const int b = blockIdx.y, n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= H * W) return;
    const int i = n / W, j = n - i * W;
    float gx, gy;
    sobel_pixel(img + (long)b * H * W, H, W, i, j, gx, gy);
    out[((long)b * 2) * H * W + n] = gx;
    out[((long)b * 2 + 1) * H * W + n] = gy;
}

__global__ __launch_bounds__(kBlock) void sobel_bwd_kernel(const float *__restrict__ g, int H, int W,
                                                           float *__restrict__ g_img) {
    const int b = blockIdx.y, n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= H * W) return;
    const int i = n / W, j = n - i * W;
    const float *gx = g + ((long)b * 2) * H * W;
    g_img[(long)b * H * W + n] = sobel_pixel_adjoint(gx, gx + (long)H * W, H, W, i, j);
}

// ------------------------------------------- Gaussian blur + Sobel pair (gradient NCC, sigma > 0)
// The reference's Sobel module blurs first (metrics.py:88-92, torchvision's gaussian_blur: blur_core.h) -- through
// torch that is a reflect pad, a k x k depthwise convolution and its backward, 0.43 of the 0.63 ms of a 32-image
// gradient-NCC call (profiles/r06/gradient_ncc.txt).  Here one launch each way: a 32 x 32 tile of the output, the
// blur as two separable passes through LDS over the tile + its Sobel halo, then the 3 x 3 pair (sobel_core.h).
// The blurred image is zero outside the image for the Sobel's zero padding, like the reference's.
constexpr int kBlurTile = 32, kBlurBlock = 256;

// K > 0: the number of taps as a compile-time constant (7 = the reference's default sigma = 1: the tile sizes
// and with them every index division become constants, the tap loops unroll), else `k`.
template <int K>
__global__ __launch_bounds__(kBlurBlock) void blur_sobel_fwd_kernel(
    const float *__restrict__ img, long img_stride, int H, int W, const float *__restrict__ taps_g, int k_rt,
    float *__restrict__ out) {
    extern __shared__ float lds[];
    const int k = K > 0 ? K : k_rt;
    const int r = k >> 1, BH = kBlurTile + 2, IN = BH + 2 * r;
    float *taps = lds, *in = taps + 32, *tmp = in + IN * IN, *bl = tmp + IN * BH;
    const int b = blockIdx.z, y0 = blockIdx.y * kBlurTile, x0 = blockIdx.x * kBlurTile;
    const float *src = img + (long)b * img_stride;
    if (threadIdx.x < 32) taps[threadIdx.x] = (int)threadIdx.x < k ? taps_g[threadIdx.x] : 0.f;
    for (int i = threadIdx.x; i < IN * IN; i += kBlurBlock) {
        const int yy = i / IN, xx = i - yy * IN;
        in[i] = src[(long)blur_reflect(y0 - 1 - r + yy, H) * W + blur_reflect(x0 - 1 - r + xx, W)];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < IN * BH; i += kBlurBlock) {  // along x
        const int yy = i / BH, c = i - yy * BH;
        float s = 0.f;
        for (int t = 0; t < k; ++t) s = fmaf(taps[t], in[yy * IN + c + t], s);
        tmp[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BH * BH; i += kBlurBlock) {  // along y
        const int rr = i / BH, c = i - rr * BH;
        const int y = y0 - 1 + rr, x = x0 - 1 + c;
        float s = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W)
            for (int t = 0; t < k; ++t) s = fmaf(taps[t], tmp[(rr + t) * BH + c], s);
        bl[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBlurTile * kBlurTile; i += kBlurBlock) {
        const int ly = i / kBlurTile, lx = i - ly * kBlurTile;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        float gx, gy;
        const float *c = bl + (ly + 1) * BH + lx + 1;  // (the tile carries its own halo: no bounds)
        sobel_of([&](int di, int dj) { return c[di * BH + dj]; }, gx, gy);
        out[((long)b * 2 * H + y) * W + x] = gx;
        out[(((long)b * 2 + 1) * H + y) * W + x] = gy;
    }
}

// The adjoint: the Sobel pair's over the tile + r, then the blur's two passes with the reflect padding folded
// back into the weights (blur_adjoint_weight).
template <int K>
__global__ __launch_bounds__(kBlurBlock) void blur_sobel_bwd_kernel(
    const float *__restrict__ g, int H, int W, const float *__restrict__ taps_g, int k_rt,
    float *__restrict__ g_img) {
    extern __shared__ float lds[];
    const int k = K > 0 ? K : k_rt;
    const int r = k >> 1, GB = kBlurTile + 2 * r, GG = GB + 2;
    float *taps = lds, *gx = taps + 32, *gy = gx + GG * GG, *gb = gy + GG * GG, *tmp = gb + GB * GB;
    const int b = blockIdx.z, y0 = blockIdx.y * kBlurTile, x0 = blockIdx.x * kBlurTile;
    const float *sx = g + (long)b * 2 * H * W, *sy = sx + (long)H * W;
    if (threadIdx.x < 32) taps[threadIdx.x] = (int)threadIdx.x < k ? taps_g[threadIdx.x] : 0.f;
    for (int i = threadIdx.x; i < GG * GG; i += kBlurBlock) {
        const int yy = i / GG, xx = i - yy * GG;
        const int y = y0 - r - 1 + yy, x = x0 - r - 1 + xx;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        gx[i] = in ? sx[(long)y * W + x] : 0.f;
        gy[i] = in ? sy[(long)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < GB * GB; i += kBlurBlock) {
        const int yy = i / GB, xx = i - yy * GB;
        const int y = y0 - r + yy, x = x0 - r + xx;
        const float *cx = gx + (yy + 1) * GG + xx + 1, *cy = gy + (yy + 1) * GG + xx + 1;
        gb[i] = (y >= 0 && y < H && x >= 0 && x < W)
                    ? sobel_adjoint_of([&](int di, int dj) { return cx[di * GG + dj]; },
                                       [&](int di, int dj) { return cy[di * GG + dj]; })
                    : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBlurTile * GB; i += kBlurBlock) {  // along y
        const int li = i / GB, c = i - li * GB;
        const int y = y0 + li;
        float s = 0.f;
        if (y < H) {
            if (blur_adjoint_plain(y, H, r))
                for (int d = -r; d <= r; ++d) s = fmaf(taps[r - d], gb[(li + r + d) * GB + c], s);
            else
                for (int d = -r; d <= r; ++d)
                    s = fmaf(blur_adjoint_weight(taps, k, y, y + d, H), gb[(li + r + d) * GB + c], s);
        }
        tmp[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBlurTile * kBlurTile; i += kBlurBlock) {  // along x
        const int li = i / kBlurTile, lj = i - li * kBlurTile;
        const int y = y0 + li, x = x0 + lj;
        if (y >= H || x >= W) continue;
        float s = 0.f;
        if (blur_adjoint_plain(x, W, r))
            for (int d = -r; d <= r; ++d) s = fmaf(taps[r - d], tmp[li * GB + lj + r + d], s);
        else
            for (int d = -r; d <= r; ++d)
                s = fmaf(blur_adjoint_weight(taps, k, x, x + d, W), tmp[li * GB + lj + r + d], s);
        g_img[((long)b * H + y) * W + x] = s;
    }
}

// One Adam step of the two pose parameter groups of a registration (reference
// notebooks/tutorials/registration.ipynb:240-316: torch.optim.Adam over {rotation: lr_rot},
// {translation: lr_xyz}) in ONE launch.  torch's fused Adam is two launches per parameter group
// (step counters, update): four launches of ~5 us for 6 numbers in a 0.19 ms iteration.  The
// update is torch/optim/adam.py's (no weight decay, no amsgrad):
//   step += 1;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;
//   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)        (maximize: g = -g)
// One workgroup: the counters are read by every thread before any is written.
constexpr int kAdamBlock = 256;
__global__ __launch_bounds__(kAdamBlock) void pose_adam_kernel(
    float *__restrict__ rot, float *__restrict__ xyz, const float *__restrict__ g_rot,
    const float *__restrict__ g_xyz, float *__restrict__ m_rot, float *__restrict__ v_rot,
    float *__restrict__ m_xyz, float *__restrict__ v_xyz, float *__restrict__ step_rot,
    float *__restrict__ step_xyz, int B, float lr_rot, float lr_xyz, float beta1, float beta2, float eps,
    int maximize) {
    const float steps[2] = {step_rot[0] + 1.f, step_xyz[0] + 1.f};
    __syncthreads();
    float ss[2], rb2[2];  // lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ss[k] = (k ? lr_xyz : lr_rot) / (1.f - powf(beta1, steps[k]));
        rb2[k] = 1.f / sqrtf(1.f - powf(beta2, steps[k]));
    }
    const int n = 3 * B;
    for (int i = threadIdx.x; i < 2 * n; i += kAdamBlock) {
        const int k = i >= n, j = k ? i - n : i;
        float *p = k ? xyz : rot, *m = k ? m_xyz : m_rot, *v = k ? v_xyz : v_rot;
        float g = (k ? g_xyz : g_rot)[j];
        g = maximize ? -g : g;
        const float mj = fmaf(g - m[j], 1.f - beta1, m[j]);
        const float vj = fmaf(beta2, v[j], (1.f - beta2) * g * g);
        m[j] = mj;
        v[j] = vj;
        p[j] -= ss[k] * mj / fmaf(sqrtf(vj), rb2[k], eps);
    }
    if (threadIdx.x == 0) {
        step_rot[0] = steps[0];
        step_xyz[0] = steps[1];
    }
}

}  // namespace

extern "C" {

int ddrr_sobel_forward(const float *img, int B, int H, int W, float *out, void *stream) {
    if (!img || !out) return fail(-1, "null pointer");
    if (B < 0 || H < 1 || W < 1) return fail(-1, "bad batch / image size");
    if (B == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 images per call");
    hipLaunchKernelGGL(sobel_fwd_kernel, dim3((H * W + kBlock - 1) / kBlock, B), dim3(kBlock), 0,
                       (hipStream_t)stream, img, H, W, out);
    return finish("ddrr_sobel_forward");
}

int ddrr_sobel_backward(const float *g_out, int B, int H, int W, float *g_img, void *stream) {
    if (!g_out || !g_img) return fail(-1, "null pointer");
    if (B < 0 || H < 1 || W < 1) return fail(-1, "bad batch / image size");
    if (B == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 images per call");
    hipLaunchKernelGGL(sobel_bwd_kernel, dim3((H * W + kBlock - 1) / kBlock, B), dim3(kBlock), 0,
                       (hipStream_t)stream, g_out, H, W, g_img);
    return finish("ddrr_sobel_backward");
}

static int blur_args_ok(int B, int H, int W, int k) {
    if (B < 0 || H < 1 || W < 1) return fail(-1, "bad batch / image size");
    if (k < 1 || k > kBlurMaxTaps || !(k & 1)) return fail(-1, "the blur takes an odd number of taps, 1 ... 31");
    if ((k >> 1) >= H || (k >> 1) >= W) return fail(-1, "reflect padding needs k // 2 < min(H, W)");
    if (B > 65535) return fail(-1, "at most 65535 images per call");
    return 0;
}

int ddrr_blur_sobel_forward(const float *img, long img_stride, int B, int H, int W, const float *taps, int k,
                            float *out, void *stream) {
    if (!img || !taps || !out) return fail(-1, "null pointer");
    if (int rc = blur_args_ok(B, H, W, k)) return rc;
    if (img_stride != 0 && img_stride != (long)H * W) return fail(-1, "img_stride must be H W, or 0 for one image");
    if (B == 0) return 0;
    const int BH = kBlurTile + 2, IN = BH + 2 * (k >> 1);
    const dim3 grid((W + kBlurTile - 1) / kBlurTile, (H + kBlurTile - 1) / kBlurTile, B);
    const size_t lds = sizeof(float) * (32 + IN * IN + IN * BH + BH * BH);
    if (k == 7)
        hipLaunchKernelGGL(blur_sobel_fwd_kernel<7>, grid, dim3(kBlurBlock), lds, (hipStream_t)stream, img,
                           img_stride, H, W, taps, k, out);
    else
        hipLaunchKernelGGL(blur_sobel_fwd_kernel<0>, grid, dim3(kBlurBlock), lds, (hipStream_t)stream, img,
                           img_stride, H, W, taps, k, out);
    return finish("ddrr_blur_sobel_forward");
}

int ddrr_blur_sobel_backward(const float *g_out, int B, int H, int W, const float *taps, int k, float *g_img,
                             void *stream) {
    if (!g_out || !taps || !g_img) return fail(-1, "null pointer");
    if (int rc = blur_args_ok(B, H, W, k)) return rc;
    if (B == 0) return 0;
    const int GB = kBlurTile + 2 * (k >> 1), GG = GB + 2;
    const dim3 grid((W + kBlurTile - 1) / kBlurTile, (H + kBlurTile - 1) / kBlurTile, B);
    const size_t lds = sizeof(float) * (32 + 2 * GG * GG + GB * GB + kBlurTile * GB);
    if (k == 7)
        hipLaunchKernelGGL(blur_sobel_bwd_kernel<7>, grid, dim3(kBlurBlock), lds, (hipStream_t)stream, g_out, H, W,
                           taps, k, g_img);
    else
        hipLaunchKernelGGL(blur_sobel_bwd_kernel<0>, grid, dim3(kBlurBlock), lds, (hipStream_t)stream, g_out, H, W,
                           taps, k, g_img);
    return finish("ddrr_blur_sobel_backward");
}

int ddrr_raygen_forward(const float *Mw, const float *Ainv, const float *P, int B, int N,
                        float *source_v, float *target_v, float *img, void *stream) {
    if (!Mw || !Ainv || !P || !source_v || !target_v || !img) return fail(-1, "null pointer");
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (B == 0 || N == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 poses per call");
    const dim3 grid((N + kBlock - 1) / kBlock, B), block(kBlock);
    hipLaunchKernelGGL(raygen_fwd_kernel, grid, block, 0, (hipStream_t)stream, Mw, Ainv, P, N,
                       source_v, target_v, img);
    return finish("ddrr_raygen_forward");
}

int ddrr_siddon_backward_pose(const float *aux, int aux_layout, const float *grad_out,
                              const float *source_v, const float *target_v, const float *img,
                              const float *Mw, const float *Ainv, const float *P, int B, int N,
                              float eps, int with_img_path, float *gMw, void *stream) {
    if (!aux || !grad_out || !source_v || !target_v || !img || !Mw || !Ainv || !P || !gMw)
        return fail(-1, "null pointer");
    if (aux_layout != DDRR_AUX_INTERLEAVED && aux_layout != DDRR_AUX_BLOCKED &&
        aux_layout != DDRR_AUX_PACKED)
        return fail(-1, "bad aux_layout");
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (B == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 poses per call");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(gMw, 0, sizeof(float) * 12 * (size_t)B, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync");
    if (N == 0) return 0;
    const dim3 grid((N + kPoseRaysPerBlock - 1) / kPoseRaysPerBlock, B), block(kBlock);
    hipLaunchKernelGGL(siddon_bwd_pose_kernel, grid, block, 0, st, aux, aux_layout, grad_out,
                       source_v, target_v, img, Mw, Ainv, P, B, N, eps, with_img_path, gMw);
    return finish("ddrr_siddon_backward_pose");
}

int ddrr_ncc_forward(const float *x1, long x1_stride, const float *x2, int B, int N, float eps,
                     float *out, float *stats, void *stream) {
    if (!x1 || !x2 || !out || !stats) return fail(-1, "null pointer");
    if (B < 0 || N < 1) return fail(-1, "bad batch / image size");
    if (B == 0) return 0;
    hipLaunchKernelGGL(ncc_fwd_kernel, dim3(B), dim3(kNccThreads), 0, (hipStream_t)stream, x1,
                       x1_stride, x2, N, eps, out, stats);
    return finish("ddrr_ncc_forward");
}

int ddrr_ncc_backward(const float *x1, long x1_stride, const float *x2, const float *stats,
                      const float *g_out, int g_stride, int B, int N, float *g_x1, float *g_x2,
                      void *stream) {
    if (!x1 || !x2 || !stats || !g_out) return fail(-1, "null pointer");
    if (g_stride != 0 && g_stride != 1) return fail(-1, "g_stride must be 1, or 0 for one value shared by the batch");
    if (g_x1 && x1_stride == 0) return fail(-1, "a shared x1 gets no gradient here");
    if (B < 0 || N < 1) return fail(-1, "bad batch / image size");
    if (B == 0 || B > 65535) return B == 0 ? 0 : fail(-1, "at most 65535 pairs per call");
    hipLaunchKernelGGL(ncc_bwd_kernel, dim3((N + kBlock - 1) / kBlock, B), dim3(kBlock), 0,
                       (hipStream_t)stream, x1, x1_stride, x2, stats, g_out, g_stride, N, g_x1, g_x2);
    return finish("ddrr_ncc_backward");
}

long ddrr_siddon_ncc_workspace_bytes(int B) {
    return B < 1 ? 0 : (long)B * (5 * sizeof(double) + 12 * sizeof(float) + 2 * sizeof(int)) +
                           (long)(sizeof(double) + 2 * sizeof(int));
}

static int check_axes(int a0, int a1, int a2) {
    if (a0 < 0 || a0 > 2 || a1 < 0 || a1 > 2 || a2 < 0 || a2 > 2 || a1 == a0 || a1 == a2)
        return fail(-1, "invalid Euler convention");
    return 0;
}

int ddrr_pose_raygen_forward(const float *rot, const float *xyz, int a0, int a1, int a2,
                             const float *reorient34, const float *Ainv, const float *P, int B, int N,
                             float *Mw, float *source_v, float *target_v, float *img, float *clear,
                             long clear_floats, void *clear_launch_ws, void *stream) {
    if (!rot || !xyz || !reorient34 || !Ainv || !P || !Mw || !source_v || !target_v || !img)
        return fail(-1, "null pointer");
    if (int rc = check_axes(a0, a1, a2)) return rc;
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (clear_floats < 0 || (clear_floats > 0 && !clear)) return fail(-1, "clear_floats without a buffer");
    if ((reinterpret_cast<uintptr_t>(clear) & 15) || (reinterpret_cast<uintptr_t>(clear_launch_ws) & 15))
        return fail(-1, "the buffers to clear must be 16-byte aligned");
    if ((clear_floats > 0 || clear_launch_ws) && (B == 0 || N == 0))
        return fail(-1, "nothing is launched for an empty batch: clear the buffers yourself");
    if (B == 0 || N == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 poses per call");
    const dim3 grid((N + kBlock - 1) / kBlock, B), block(kBlock);
    hipLaunchKernelGGL(pose_raygen_fwd_kernel, grid, block, 0, (hipStream_t)stream, rot, xyz, a0, a1,
                       a2, reorient34, Ainv, P, N, Mw, source_v, target_v, img, clear_floats > 0 ? clear : nullptr,
                       clear_floats, reinterpret_cast<int *>(clear_launch_ws));
    return finish("ddrr_pose_raygen_forward");
}

int ddrr_siddon_ncc_forward(const float *aux, const float *img, const float *x1, long x1_stride, int B,
                            int N, float eps, void *ws, float *ncc, float *stats, float *out,
                            float *ncc_sum, void *stream) {
    if (!aux || !img || !x1 || !ws || !ncc || !stats) return fail(-1, "null pointer");
    if (x1_stride != 0 && x1_stride != N) return fail(-1, "x1_stride must be N, or 0 for a shared image");
    if (B < 0 || N < 1) return fail(-1, "bad batch / image size");
    if (B == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 pairs per call");
    const dim3 block(kBlock);
    if ((long)B * ((N + 4095) / 4096) >= kStepWorkgroupsWanted)
        hipLaunchKernelGGL(siddon_ncc_fwd_kernel<4>, dim3((N + 4095) / 4096, B), block, 0, (hipStream_t)stream,
                           aux, img, x1, x1_stride, B, N, eps, ws, ncc, stats, out, ncc_sum);
    else
        hipLaunchKernelGGL(siddon_ncc_fwd_kernel<1>, dim3((N + 1023) / 1024, B), block, 0, (hipStream_t)stream,
                           aux, img, x1, x1_stride, B, N, eps, ws, ncc, stats, out, ncc_sum);
    return finish("ddrr_siddon_ncc_forward");
}

int ddrr_siddon_ncc_backward_pose(const float *aux, const float *img, const float *x1, long x1_stride,
                                  const float *stats, const float *g_out, int g_stride,
                                  const float *source_v, const float *target_v, const float *Mw,
                                  const float *Ainv, const float *P, const float *rot, const float *xyz,
                                  int a0, int a1, int a2, const float *reorient34, int B, int N,
                                  float eps, int with_img_path, void *ws, float *g_rot, float *g_xyz,
                                  void *stream) {
    if (!aux || !img || !x1 || !stats || !g_out || !source_v || !target_v || !Mw || !Ainv || !P ||
        !rot || !xyz || !reorient34 || !ws || !g_rot || !g_xyz)
        return fail(-1, "null pointer");
    if (int rc = check_axes(a0, a1, a2)) return rc;
    if (x1_stride != 0 && x1_stride != N) return fail(-1, "x1_stride must be N, or 0 for a shared image");
    if (g_stride != 0 && g_stride != 1) return fail(-1, "g_stride must be 1, or 0 for one value shared by the batch");
    if (B < 0 || N < 1) return fail(-1, "bad batch / image size");
    if (B == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 poses per call");
    const dim3 block(kBlock);
    if ((long)B * ((N + 2047) / 2048) >= kStepWorkgroupsWanted)
        hipLaunchKernelGGL(siddon_ncc_bwd_pose_kernel<2048>, dim3((N + 2047) / 2048, B), block, 0,
                           (hipStream_t)stream, aux, img, x1, x1_stride, stats, g_out, g_stride, source_v,
                           target_v, Mw, Ainv, P, rot, xyz, a0, a1, a2, reorient34, B, N, eps, with_img_path,
                           ws, g_rot, g_xyz);
    else
        hipLaunchKernelGGL(siddon_ncc_bwd_pose_kernel<1024>, dim3((N + 1023) / 1024, B), block, 0,
                           (hipStream_t)stream, aux, img, x1, x1_stride, stats, g_out, g_stride, source_v,
                           target_v, Mw, Ainv, P, rot, xyz, a0, a1, a2, reorient34, B, N, eps, with_img_path,
                           ws, g_rot, g_xyz);
    return finish("ddrr_siddon_ncc_backward_pose");
}

int ddrr_siddon_backward_pose_euler(const float *aux, const float *grad_out, const float *source_v,
                                    const float *Mw, const float *Ainv, const float *P, const float *rot,
                                    const float *xyz, int a0, int a1, int a2, const float *reorient34, int B,
                                    int N, float eps, int with_img_path, void *ws, float *g_rot, float *g_xyz,
                                    void *stream) {
    if (!aux || !grad_out || !source_v || !Mw || !Ainv || !P || !rot || !xyz || !reorient34 || !ws || !g_rot ||
        !g_xyz)
        return fail(-1, "null pointer");
    if (int rc = check_axes(a0, a1, a2)) return rc;
    if (B < 0 || N < 1) return fail(-1, "bad batch / image size");
    if (B == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 poses per call");
    const dim3 block(kBlock);
    const float *none = nullptr;
    if ((long)B * ((N + 2047) / 2048) >= kStepWorkgroupsWanted)
        hipLaunchKernelGGL((siddon_ncc_bwd_pose_kernel<2048, true>), dim3((N + 2047) / 2048, B), block, 0,
                           (hipStream_t)stream, aux, none, grad_out, (long)N, none, none, 0, source_v, none, Mw,
                           Ainv, P, rot, xyz, a0, a1, a2, reorient34, B, N, eps, with_img_path, ws, g_rot, g_xyz);
    else
        hipLaunchKernelGGL((siddon_ncc_bwd_pose_kernel<1024, true>), dim3((N + 1023) / 1024, B), block, 0,
                           (hipStream_t)stream, aux, none, grad_out, (long)N, none, none, 0, source_v, none, Mw,
                           Ainv, P, rot, xyz, a0, a1, a2, reorient34, B, N, eps, with_img_path, ws, g_rot, g_xyz);
    return finish("ddrr_siddon_backward_pose_euler");
}

int ddrr_ncc_patch_forward(const float *x1, long x1_stride, const float *x2, int B, int H, int W, int p,
                           float eps, float *out, float *coef, void *stream) {
    if (!x1 || !x2 || !out) return fail(-1, "null pointer");
    if (B < 0 || H < 1 || W < 1) return fail(-1, "bad batch / image size");
    if (p < 1 || p > H || p > W || p > 64) return fail(-1, "patch_size must be 1 ... min(H, W, 64)");
    if (x1_stride != 0 && x1_stride != (long)H * W) return fail(-1, "x1_stride must be H W, or 0 for a shared image");
    if (B == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 pairs per call");
    const int hw = H - p + 1, ww = W - p + 1, T = kPatchTile + p - 1;
    if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)B, (hipStream_t)stream) != hipSuccess)
        return fail(-1, "hipMemsetAsync");
    const int tiles = ((ww + kPatchTile - 1) / kPatchTile) * ((hw + kPatchTile - 1) / kPatchTile);
    const int per_pair = std::min(tiles, std::max(kPatchWorkgroupsPerPairMin, kPatchWorkgroupsWanted / B));
    const dim3 grid(per_pair, 1, B);
    // (the window sizes the reference's notebooks use, as compile-time constants: 13 -- metrics.ipynb:94 --,
    // 9 -- 05_metrics.ipynb:291 --, and their neighbours; any other size takes the run-time loop)
#define DDRR_PATCH_FWD(P_)                                                                                     \
    hipLaunchKernelGGL(ncc_patch_fwd_kernel<P_>, grid, dim3(kPatchTile * kPatchTile),                   \
                       2 * T * patch_stride_fwd(T) * sizeof(float),                                            \
                       (hipStream_t)stream, x1, x1_stride, x2, H, W, p, eps, out, reinterpret_cast<float4 *>(coef))
    switch (p) {
        case 5: DDRR_PATCH_FWD(5); break;
        case 7: DDRR_PATCH_FWD(7); break;
        case 9: DDRR_PATCH_FWD(9); break;
        case 11: DDRR_PATCH_FWD(11); break;
        case 13: DDRR_PATCH_FWD(13); break;
        default: DDRR_PATCH_FWD(0); break;
    }
#undef DDRR_PATCH_FWD
    return finish("ddrr_ncc_patch_forward");
}

int ddrr_ncc_patch_backward(const float *x1, long x1_stride, const float *x2, const float *coef,
                            const float *g_out, int g_stride, int B, int H, int W, int p, float *g_x2,
                            void *stream) {
    if (!x1 || !x2 || !coef || !g_out || !g_x2) return fail(-1, "null pointer");
    if (B < 0 || H < 1 || W < 1) return fail(-1, "bad batch / image size");
    if (p < 1 || p > H || p > W || p > 64) return fail(-1, "patch_size must be 1 ... min(H, W, 64)");
    if (x1_stride != 0 && x1_stride != (long)H * W) return fail(-1, "x1_stride must be H W, or 0 for a shared image");
    if (g_stride != 0 && g_stride != 1) return fail(-1, "g_stride must be 1, or 0 for one value shared by the batch");
    if (B == 0) return 0;
    if (B > 65535) return fail(-1, "at most 65535 pairs per call");
    if ((reinterpret_cast<uintptr_t>(coef) & 15) != 0) return fail(-1, "coef must be 16-byte aligned");
    const int T = kPatchTile + p - 1;
    const dim3 grid((W + kPatchTile - 1) / kPatchTile, (H + kPatchTile - 1) / kPatchTile, B);
#define DDRR_PATCH_BWD(P_)                                                                                  \
    hipLaunchKernelGGL(ncc_patch_bwd_kernel<P_>, grid, dim3(kPatchTile * kPatchTile),          \
                       T * patch_stride_bwd(T) * sizeof(float4),                                            \
                       (hipStream_t)stream, x1, x1_stride, x2, reinterpret_cast<const float4 *>(coef), g_out,  \
                       g_stride, H, W, p, g_x2)
    // (windows past 48: more than the 64 KB of LDS a launch may ask for unannounced)
    const size_t lds_bytes = (size_t)T * patch_stride_bwd(T) * sizeof(float4);
    if (lds_bytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ncc_patch_bwd_kernel<0>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute");
    }
    switch (p) {
        case 5: DDRR_PATCH_BWD(5); break;
        case 7: DDRR_PATCH_BWD(7); break;
        case 9: DDRR_PATCH_BWD(9); break;
        case 11: DDRR_PATCH_BWD(11); break;
        case 13: DDRR_PATCH_BWD(13); break;
        default: DDRR_PATCH_BWD(0); break;
    }
#undef DDRR_PATCH_BWD
    return finish("ddrr_ncc_patch_backward");
}

int ddrr_pose_adam_step(float *rot, float *xyz, const float *g_rot, const float *g_xyz, float *m_rot,
                        float *v_rot, float *m_xyz, float *v_xyz, float *step_rot, float *step_xyz, int B,
                        float lr_rot, float lr_xyz, float beta1, float beta2, float eps, int maximize,
                        void *stream) {
    if (!rot || !xyz || !g_rot || !g_xyz || !m_rot || !v_rot || !m_xyz || !v_xyz || !step_rot || !step_xyz)
        return fail(-1, "null pointer");
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f) || !(lr_rot >= 0.f) ||
        !(lr_xyz >= 0.f))
        return fail(-1, "invalid Adam hyper-parameters");
    if (B <= 0) return B == 0 ? 0 : fail(-1, "negative batch");
    hipLaunchKernelGGL(pose_adam_kernel, dim3(1), dim3(kAdamBlock), 0, (hipStream_t)stream, rot, xyz, g_rot,
                       g_xyz, m_rot, v_rot, m_xyz, v_xyz, step_rot, step_xyz, B, lr_rot, lr_xyz, beta1, beta2,
                       eps, maximize);
    return finish("ddrr_pose_adam_step");
}

int ddrr_pose_euler_forward(const float *rot, const float *xyz, int a0, int a1, int a2,
                            const float *reorient34, int B, float *Mw, void *stream) {
    if (!rot || !xyz || !reorient34 || !Mw) return fail(-1, "null pointer");
    if (a0 < 0 || a0 > 2 || a1 < 0 || a1 > 2 || a2 < 0 || a2 > 2 || a1 == a0 || a1 == a2)
        return fail(-1, "invalid Euler convention");
    if (B <= 0) return B == 0 ? 0 : fail(-1, "negative batch");
    hipLaunchKernelGGL(pose_euler_fwd_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, rot, xyz, a0, a1, a2, reorient34, B, Mw);
    return finish("ddrr_pose_euler_forward");
}

int ddrr_pose_euler_backward(const float *rot, const float *xyz, int a0, int a1, int a2,
                             const float *reorient34, const float *gMw, int B, float *g_rot,
                             float *g_xyz, void *stream) {
    if (!rot || !xyz || !reorient34 || !gMw || !g_rot || !g_xyz) return fail(-1, "null pointer");
    if (a0 < 0 || a0 > 2 || a1 < 0 || a1 > 2 || a2 < 0 || a2 > 2 || a1 == a0 || a1 == a2)
        return fail(-1, "invalid Euler convention");
    if (B <= 0) return B == 0 ? 0 : fail(-1, "negative batch");
    hipLaunchKernelGGL(pose_euler_bwd_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, rot, xyz, a0, a1, a2, reorient34, gMw, B, g_rot, g_xyz);
    return finish("ddrr_pose_euler_backward");
}

}  // extern "C"
