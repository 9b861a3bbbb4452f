// valu_rates.hip -- issue-rate micro-benchmark for the instructions the brick walk is built from
// (gfx950).  Standalone: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
//
// Every case runs the same instruction REP times per loop iteration on independent registers,
// 1024 threads per workgroup (4 waves per SIMD, the brick kernel's occupancy), one workgroup per
// CU.  Reported: shader cycles (s_memtime) per wave-instruction per SIMD, i.e. the issue cost the
// DESIGN.md instruction budgets are written in.  Also checks what `clamp` does on v_pk_fma_f32.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kIters = 100000;
constexpr int kRep = 16;  // independent instructions per iteration

enum Case {
    FMA,
    PK_FMA,
    PK_ADD,
    PK_MUL,
    PK_FMA_CLAMP,
    MIN3,
    CMP_CNDMASK,
    CVT,
    ADD,
    LDS_READ,
    LDS_ADD_RTN,
    LDS_ADD,
    FMA_DEP,
    PK_FMA_DEP,
    MIX_WALK,
    FMA_SAME,
    MUL,
    MAX,
    MED3,
    FLOOR,
    RCP,
    N_CASES
};
const char *kNames[N_CASES] = {"v_fma_f32",
                               "v_pk_fma_f32",
                               "v_pk_add_f32",
                               "v_pk_mul_f32",
                               "v_pk_fma_f32 clamp",
                               "v_min3_f32",
                               "v_cmp_le + v_cndmask (pair)",
                               "v_cvt_i32_f32",
                               "v_add_f32",
                               "ds_read_b32 (no conflicts)",
                               "ds_add_rtn_u32 (64 addresses)",
                               "ds_add_u32 (64 addresses)",
                               "v_fma_f32 dependent chain",
                               "v_pk_fma_f32 dependent chain",
                               "mix: 2 min3 + 12 pk + 2 cvt + 2 ds_read",
                               "v_fma_f32 x, x, x, c (two distinct sources)",
                               "v_mul_f32",
                               "v_max_f32",
                               "v_med3_f32",
                               "v_floor_f32",
                               "v_rcp_f32"};

template <int C>
__global__ __launch_bounds__(1024) void rate_kernel(unsigned long long *cycles, float *sink,
                                                    float seed) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 1024) lds[i] = (float)i;
    __syncthreads();
    float a[kRep];
    v2f p[kRep];
#pragma unroll
    for (int i = 0; i < kRep; ++i) {
        a[i] = seed + (float)(i + tid);
        p[i] = v2f{seed + (float)i, seed - (float)tid};
    }
    const float m = 1.0000001f, c = 1e-9f;
    const v2f pm = {m, m}, pc = {c, c};
    unsigned laddr = (unsigned)(tid & 1023) * 4u;
    unsigned ldsbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)lds;
    laddr += ldsbase;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kRep; ++i) {
            if (C == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (C == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (C == FMA_SAME) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (C == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (C == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (C == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (C == FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            if (C == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (C == PK_FMA)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if (C == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (C == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (C == PK_FMA_CLAMP)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 clamp" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if (C == MIN3)
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % kRep]));
            if (C == CMP_CNDMASK)
                asm volatile("v_cmp_le_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc"
                             : "+v"(a[i])
                             : "v"(m), "v"(c)
                             : "vcc");
            if (C == CVT) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            if (C == LDS_READ) {
                float r;
                asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(laddr));
                a[i] = r;
            }
            if (C == LDS_ADD_RTN) {
                unsigned r;
                unsigned one = 1u;
                asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(r) : "v"(laddr), "v"(one));
                a[i] = __uint_as_float(r);
            }
            if (C == LDS_ADD) {
                unsigned one = 1u;
                asm volatile("ds_add_u32 %0, %1" : : "v"(laddr), "v"(one));
            }
        }
        if (C == FMA_DEP) {
#pragma unroll
            for (int i = 0; i < kRep; ++i)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
        }
        if (C == PK_FMA_DEP) {
#pragma unroll
            for (int i = 0; i < kRep; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pm), "v"(pc));
        }
        if (C == MIX_WALK) {
            // the instruction mix of one step of two rays per lane (18 wave-instructions)
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(a[1]), "v"(a[2]));
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(a[4]), "v"(a[5]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 clamp" : "+v"(p[i + 4]) : "v"(pm), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i + 8]) : "v"(pm), "v"(pc));
            }
            asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[6]));
            asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[7]));
            float r0, r1;
            asm volatile("ds_read_b32 %0, %1" : "=v"(r0) : "v"(laddr));
            asm volatile("ds_read_b32 %0, %1" : "=v"(r1) : "v"(laddr));
            a[8] = r0;
            a[9] = r1;
        }
        if (C == LDS_READ || C == LDS_ADD_RTN || C == LDS_ADD || C == MIX_WALK)
            asm volatile("s_waitcnt lgkmcnt(0)");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && blockIdx.x == 0) cycles[gridDim.x * 16] = r1 - r0;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kRep; ++i) s += a[i] + p[i].x + p[i].y;
    sink[blockIdx.x * 1024 + tid] = s;
    if ((tid & 63) == 0) cycles[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}

template <int C>
void run(int n_cu, unsigned long long *d_cycles, float *d_sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<C>, dim3(n_cu), dim3(1024), 0, 0, d_cycles, d_sink, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<C>, dim3(n_cu), dim3(1024), 0, 0, d_cycles, d_sink, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(n_cu * 16 + 1);
    CHECK(hipMemcpy(h.data(), d_cycles, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < n_cu * 16; ++i) mean += (double)h[i];
    mean /= n_cu * 16;
    const double real_ns = (double)h[n_cu * 16] * 10.0;  // s_memrealtime: 100 MHz
    int per_iter = kRep;
    if (C == CMP_CNDMASK) per_iter = 2 * kRep;
    if (C == MIX_WALK) per_iter = 18;
    // 4 waves share a SIMD: cycles per wave-instruction per SIMD = wave time / (4 * instructions)
    const double cyc = mean / ((double)kIters * per_iter * 4);
    printf("%-44s %6.2f s_memtime ticks = %6.3f ns per wave-instruction per SIMD (kernel %.2f ms; "
           "s_memtime %.3f GHz by s_memrealtime)\n",
           kNames[C], cyc, ms * 1e6 / ((double)kIters * per_iter * 4), ms, mean / real_ns);
}

__global__ void clamp_check(float *out) {
    // what `clamp` does to v_pk_fma_f32: inputs chosen to land below 0, inside and above 1, NaN
    v2f x = {-3.0f, 0.25f}, y = {7.0f, __builtin_nanf("")};
    const v2f one = {1.0f, 1.0f}, zero = {0.0f, 0.0f};
    v2f r0, r1;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r0) : "v"(x), "v"(one), "v"(zero));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r1) : "v"(y), "v"(one), "v"(zero));
    out[0] = r0.x;
    out[1] = r0.y;
    out[2] = r1.x;
    out[3] = r1.y;
    // the select the walk needs: t = clamp(1 - delta * 2^126), delta = 0, 1 ulp, denormal
    const float big = -0x1p126f;
    v2f d0 = {0.0f, 0x1p-25f}, d1 = {0x1p-149f, 0x1p-130f};
    v2f bigv = {big, big};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r0) : "v"(d0), "v"(bigv), "v"(one));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r1) : "v"(d1), "v"(bigv), "v"(one));
    out[4] = r0.x;
    out[5] = r0.y;
    out[6] = r1.x;
    out[7] = r1.y;
    // op_sel broadcast: both halves of the result use the LOW half of src1
    v2f q = {2.0f, 3.0f}, w = {10.0f, 20.0f}, r2;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r2) : "v"(q), "v"(w));
    out[8] = r2.x;
    out[9] = r2.y;
    // neg modifiers on a packed add: a - b
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r2) : "v"(q), "v"(w));
    out[10] = r2.x;
    out[11] = r2.y;
}

int main() {
    int dev = 0, n_cu = 0, clk = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev));
    printf("device %d: %d CUs, clock attribute %.2f GHz; 1024 threads per CU, %d x %d instructions\n", dev,
           n_cu, clk * 1e-6, kIters, kRep);
    unsigned long long *d_cycles;
    float *d_sink;
    CHECK(hipMalloc(&d_cycles, ((size_t)n_cu * 16 + 1) * 8));
    CHECK(hipMalloc(&d_sink, (size_t)n_cu * 1024 * 4));
    run<FMA>(n_cu, d_cycles, d_sink);
    run<ADD>(n_cu, d_cycles, d_sink);
    run<PK_FMA>(n_cu, d_cycles, d_sink);
    run<PK_ADD>(n_cu, d_cycles, d_sink);
    run<PK_MUL>(n_cu, d_cycles, d_sink);
    run<PK_FMA_CLAMP>(n_cu, d_cycles, d_sink);
    run<MIN3>(n_cu, d_cycles, d_sink);
    run<CMP_CNDMASK>(n_cu, d_cycles, d_sink);
    run<CVT>(n_cu, d_cycles, d_sink);
    run<FMA_DEP>(n_cu, d_cycles, d_sink);
    run<PK_FMA_DEP>(n_cu, d_cycles, d_sink);
    run<LDS_READ>(n_cu, d_cycles, d_sink);
    run<LDS_ADD_RTN>(n_cu, d_cycles, d_sink);
    run<LDS_ADD>(n_cu, d_cycles, d_sink);
    run<MIX_WALK>(n_cu, d_cycles, d_sink);
    run<FMA_SAME>(n_cu, d_cycles, d_sink);
    run<MUL>(n_cu, d_cycles, d_sink);
    run<MAX>(n_cu, d_cycles, d_sink);
    run<MED3>(n_cu, d_cycles, d_sink);
    run<FLOOR>(n_cu, d_cycles, d_sink);
    run<RCP>(n_cu, d_cycles, d_sink);
    float *d_out, h[12];
    CHECK(hipMalloc(&d_out, sizeof(h)));
    hipLaunchKernelGGL(clamp_check, dim3(1), dim3(64), 0, 0, d_out);
    CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    printf("clamp(v_pk_fma_f32): -3 -> %g, 0.25 -> %g, 7 -> %g, NaN -> %g\n", h[0], h[1], h[2], h[3]);
    printf("t = clamp(1 - delta * 2^126): delta 0 -> %g, 2^-25 -> %g, 2^-149 -> %g, 2^-130 -> %g\n", h[4],
           h[5], h[6], h[7]);
    printf("op_sel_hi:[1,0] on {2,3} * {10,20} -> {%g, %g} (broadcast of the low half: {20, 30})\n", h[8],
           h[9]);
    printf("neg_lo/neg_hi on {2,3} + (-{10,20}) -> {%g, %g} (expected {-8, -17})\n", h[10], h[11]);
    return 0;
}
