// valu_latency.hip -- dependent-issue latency on gfx950: one wave per SIMD (256 threads per CU)
// runs a chain in which every instruction reads the previous one's result; also 2 and 4
// interleaved independent chains (ILP) and the LDS read -> use round trip.
//   hipcc --offload-arch=gfx950 -O3 valu_latency.hip -o valu_latency && ./valu_latency
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int kIters = 40000;
enum { FMA1, FMA2, FMA4, MIN3_1, MIN3_2, CMPCND_1, MIX_WALK_1, MIX_WALK_2, LDS_CHAIN, N_CASES };
const char *kNames[N_CASES] = {"v_fma_f32, 1 chain",
                               "v_fma_f32, 2 interleaved chains",
                               "v_fma_f32, 4 interleaved chains",
                               "v_min3_f32, 1 chain",
                               "v_min3_f32, 2 interleaved chains",
                               "v_cmp + v_cndmask, 1 chain (per instruction)",
                               "walk-like chain min3 > sub > fma clamp > fma > fma (per instruction)",
                               "same, 2 interleaved chains",
                               "ds_read_b32 > address of the next read (per read)"};

template <int C>
__global__ __launch_bounds__(1024) void lat_kernel(float *sink, float seed, int n_threads) {
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (unsigned)(((i * 37 + 11) & 1023) * 4);
    __syncthreads();
    float a = seed + threadIdx.x, b = seed * 2, c = seed * 3, d = seed * 4;
    const float m = 1.0000001f, k = 1e-9f, nbig = -0x1p126f;
    unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned *)lds +
                    (threadIdx.x & 1023) * 4u;
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned *)lds;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (C == FMA1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));
            if (C == FMA2) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(k));
            }
            if (C == FMA4) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(k));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(m), "v"(k));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(m), "v"(k));
            }
            if (C == MIN3_1) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));
            if (C == MIN3_2) {
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(k));
            }
            if (C == CMPCND_1)
                asm volatile("v_cmp_le_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc"
                             : "+v"(a) : "v"(m), "v"(k) : "vcc");
            if (C == MIX_WALK_1 || C == MIX_WALK_2) {
                asm volatile(
                    "v_min3_f32 %0, %0, %1, %2\n\tv_sub_f32 %0, %0, %1\n\t"
                    "v_fma_f32 %0, %0, %3, %1 clamp\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                    "v_fma_f32 %0, %0, %1, %2"
                    : "+v"(a) : "v"(m), "v"(k), "v"(nbig));
            }
            if (C == MIX_WALK_2) {
                asm volatile(
                    "v_min3_f32 %0, %0, %1, %2\n\tv_sub_f32 %0, %0, %1\n\t"
                    "v_fma_f32 %0, %0, %3, %1 clamp\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                    "v_fma_f32 %0, %0, %1, %2"
                    : "+v"(b) : "v"(m), "v"(k), "v"(nbig));
            }
            if (C == LDS_CHAIN) {
                unsigned nx;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, %2"
                             : "=&v"(nx) : "v"(addr), "v"(base));
                addr = nx;
            }
        }
    }
    sink[blockIdx.x * 1024 + threadIdx.x] = a + b + c + d + (float)addr;
}

template <int C>
void run(int n_cu, float *d_sink, int threads) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(lat_kernel<C>, dim3(n_cu), dim3(threads), 0, 0, d_sink, 1.0f, threads);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(lat_kernel<C>, dim3(n_cu), dim3(threads), 0, 0, d_sink, 1.0f, threads);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    int per = 8;
    if (C == FMA2 || C == MIN3_2) per = 16;
    if (C == FMA4) per = 32;
    if (C == CMPCND_1) per = 16;
    if (C == MIX_WALK_1) per = 40;
    if (C == MIX_WALK_2) per = 80;
    const int waves_per_simd = threads / 256;
    // time one wave needs per instruction it issues
    const double ns = ms * 1e6 / ((double)kIters * per);
    printf("%-74s %d wave(s)/SIMD: %6.3f ns per instruction of a wave  (%5.3f ns per SIMD issue)\n",
           kNames[C], waves_per_simd, ns, ns / waves_per_simd);
}

template <int C>
void run_all(int n_cu, float *d_sink) {
    run<C>(n_cu, d_sink, 256);
    run<C>(n_cu, d_sink, 512);
    run<C>(n_cu, d_sink, 1024);
    if constexpr (C + 1 < N_CASES) run_all<C + 1>(n_cu, d_sink);
}

int main() {
    int dev = 0, n_cu = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    float *d_sink;
    CHECK(hipMalloc(&d_sink, (size_t)n_cu * 1024 * 4));
    run_all<0>(n_cu, d_sink);
    return 0;
}
