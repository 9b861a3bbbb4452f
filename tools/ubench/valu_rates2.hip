// valu_rates2.hip -- second pass of the gfx950 issue-rate micro-benchmark: explicit registers, so
// that VGPR bank placement (bank = index % 4) is controlled, and more opcodes.  One workgroup of
// 1024 threads per CU (4 waves per SIMD); reports ns per wave-instruction per SIMD (wall clock,
// kernels of several ms) and the ratio to v_add_f32.
//   hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o valu_rates2 && ./valu_rates2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int kIters = 60000;

// 16 instructions per block.  D(i): destination register of instruction i; the sources are
// fixed registers v120.. so that their banks are known: v120 bank 0, v121 bank 1, v122 bank 2.
// "nc" (no conflict) blocks use destinations in bank 3 and 0/1/2 as needed.
#define CLOB                                                                                       \
    "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122",    \
        "v123", "v124", "v125", "v126", "v127", "v60", "v61", "v62", "v63", "v64", "v65", "v66",   \
        "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
        "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "vcc", \
        "s40", "s41", "s42", "s43"

#define REP16(OP, A)                                                                             \
    OP(100, A) OP(101, A) OP(102, A) OP(103, A) OP(104, A) OP(105, A) OP(106, A) OP(107, A)      \
        OP(108, A) OP(109, A) OP(110, A) OP(111, A) OP(112, A) OP(113, A) OP(114, A) OP(115, A)
// destinations all in bank 3 (v63, v67, ...): sources v120 (bank 0), v121 (1), v122 (2)
#define REP16B3(OP, A)                                                                       \
    OP(63, A) OP(67, A) OP(71, A) OP(75, A) OP(79, A) OP(83, A) OP(87, A) OP(91, A) OP(103, A) \
        OP(107, A) OP(111, A) OP(115, A) OP(119, A) OP(123, A) OP(127, A) OP(99, A)

#define S1(x) #x
#define S(x) S1(x)
#define OP_ADD(d, A) "v_add_f32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_SUB(d, A) "v_sub_f32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_MUL(d, A) "v_mul_f32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_FMAC(d, A) "v_fmac_f32 v" S(d) ", v120, v121\n\t"
#define OP_FMA3(d, A) "v_fma_f32 v" S(d) ", v" S(d) ", v120, v121\n\t"
#define OP_FMA_OUT(d, A) "v_fma_f32 v" S(d) ", v120, v121, v122\n\t"
#define OP_MIN(d, A) "v_min_f32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_MAX(d, A) "v_max_f32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_MIN3(d, A) "v_min3_f32 v" S(d) ", v" S(d) ", v120, v121\n\t"
#define OP_CMP(d, A) "v_cmp_le_f32 vcc, v" S(d) ", v120\n\t"
#define OP_CMP_S(d, A) "v_cmp_le_f32 s[40:41], v" S(d) ", v120\n\t"
#define OP_CND(d, A) "v_cndmask_b32 v" S(d) ", v" S(d) ", v120, vcc\n\t"
#define OP_CND_S(d, A) "v_cndmask_b32 v" S(d) ", v" S(d) ", v120, s[42:43]\n\t"
#define OP_MOV(d, A) "v_mov_b32 v" S(d) ", v120\n\t"
#define OP_AND(d, A) "v_and_b32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_ADDU(d, A) "v_add_u32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_ADDC(d, A) "v_addc_co_u32 v" S(d) ", vcc, v" S(d) ", v120, vcc\n\t"
#define OP_LSHLADD(d, A) "v_lshl_add_u32 v" S(d) ", v" S(d) ", 2, v120\n\t"
#define OP_MAD_U24(d, A) "v_mad_u32_u24 v" S(d) ", v" S(d) ", v120, v121\n\t"
#define OP_CVT_I(d, A) "v_cvt_i32_f32 v" S(d) ", v" S(d) "\n\t"
#define OP_CVT_F(d, A) "v_cvt_f32_i32 v" S(d) ", v" S(d) "\n\t"
#define OP_FLOOR(d, A) "v_floor_f32 v" S(d) ", v" S(d) "\n\t"
#define OP_RNDNE(d, A) "v_rndne_f32 v" S(d) ", v" S(d) "\n\t"
#define OP_MED3(d, A) "v_med3_f32 v" S(d) ", v" S(d) ", v120, v121\n\t"
#define OP_ADD_CLAMP(d, A) "v_add_f32 v" S(d) ", v" S(d) ", v120 clamp\n\t"
#define OP_FMA_CLAMP(d, A) "v_fma_f32 v" S(d) ", v" S(d) ", v120, v121 clamp\n\t"
#define OP_MUL_CLAMP(d, A) "v_mul_f32 v" S(d) ", v" S(d) ", v120 clamp\n\t"
#define OP_DPP(d, A) "v_mov_b32_dpp v" S(d) ", v120 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define OP_MAXNEG(d, A) "v_max_f32 v" S(d) ", v" S(d) ", -v" S(d) "\n\t"
#define OP_SUBREV_ABS(d, A) "v_sub_f32 v" S(d) ", |v" S(d) "|, v120\n\t"
#define OP_FMA_SGPR(d, A) "v_fma_f32 v" S(d) ", v" S(d) ", s40, v121\n\t"
#define OP_CMPX(d, A) "v_cmp_le_f32 vcc, v" S(d) ", v120\n\tv_cndmask_b32 v" S(d) ", v" S(d) ", v121, vcc\n\t"
#define OP_MINMAX_E64(d, A) "v_min_f32_e64 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_FMA_INL(d, A) "v_fma_f32 v" S(d) ", v" S(d) ", 4.0, v121\n\t"
#define OP_FMA_DEN(d, A) "v_fma_f32 v" S(d) ", v124, v125, v126\n\t"
#define OP_ADD_SGPR(d, A) "v_add_f32 v" S(d) ", s40, v" S(d) "\n\t"
#define OP_MUL_LIT(d, A) "v_mul_f32 v" S(d) ", 0x3f8ccccd, v" S(d) "\n\t"
#define OP_LSHL(d, A) "v_lshlrev_b32 v" S(d) ", 2, v" S(d) "\n\t"
#define OP_OR(d, A) "v_or_b32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_XOR(d, A) "v_xor_b32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_SUBU(d, A) "v_sub_u32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_BFE(d, A) "v_bfe_u32 v" S(d) ", v" S(d) ", 3, 5\n\t"
#define OP_MAX3(d, A) "v_max3_f32 v" S(d) ", v" S(d) ", v120, v121\n\t"
#define OP_FMA_NEG(d, A) "v_fma_f32 v" S(d) ", -v" S(d) ", |v120|, v121\n\t"
#define OP_ADD_DPP(d, A) "v_add_f32_dpp v" S(d) ", v120, v" S(d) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define OP_MADAK(d, A) "v_fmaak_f32 v" S(d) ", v" S(d) ", v120, 0x3f8ccccd\n\t"
#define OP_MUL_LEGACY(d, A) "v_mul_legacy_f32 v" S(d) ", v" S(d) ", v120\n\t"
#define OP_PKFMA(d, A) "v_pk_fma_f32 v[" S(d) ":" S(d) "+1], v[" S(d) ":" S(d) "+1], v[120:121], v[122:123]\n\t"

enum {
    C_ADD,
    C_ADD_B3,
    C_SUB,
    C_MUL,
    C_FMAC,
    C_FMA3,
    C_FMA3_B3,
    C_FMA_OUT_B3,
    C_MIN,
    C_MAX,
    C_MIN3,
    C_MIN3_B3,
    C_CMP,
    C_CMP_S,
    C_CND,
    C_CND_S,
    C_MOV,
    C_AND,
    C_ADDU,
    C_ADDC,
    C_LSHLADD,
    C_MAD_U24,
    C_CVT_I,
    C_CVT_F,
    C_FLOOR,
    C_RNDNE,
    C_MED3,
    C_ADD_CLAMP,
    C_FMA_CLAMP,
    C_MUL_CLAMP,
    C_DPP,
    C_FMA_SGPR,
    C_CMPX,
    C_FMA_INL,
    C_FMA_DEN,
    C_ADD_SGPR,
    C_MUL_LIT,
    C_LSHL,
    C_OR,
    C_XOR,
    C_SUBU,
    C_BFE,
    C_MAX3,
    C_FMA_NEG,
    C_ADD_DPP,
    C_MADAK,
    N_CASES
};

const char *kNames[N_CASES] = {"v_add_f32 d,d,v120",
                               "v_add_f32 (dst bank 3, src bank 0)",
                               "v_sub_f32",
                               "v_mul_f32",
                               "v_fmac_f32 d, v120, v121",
                               "v_fma_f32 d,d,v120,v121 (dst all banks)",
                               "v_fma_f32 d,d,v120,v121 (dst bank 3: no conflict)",
                               "v_fma_f32 d,v120,v121,v122 (banks 0,1,2)",
                               "v_min_f32",
                               "v_max_f32",
                               "v_min3_f32 (dst all banks)",
                               "v_min3_f32 (dst bank 3)",
                               "v_cmp_le_f32 vcc",
                               "v_cmp_le_f32 sgpr pair",
                               "v_cndmask_b32 vcc",
                               "v_cndmask_b32 sgpr pair",
                               "v_mov_b32",
                               "v_and_b32",
                               "v_add_u32",
                               "v_addc_co_u32",
                               "v_lshl_add_u32",
                               "v_mad_u32_u24",
                               "v_cvt_i32_f32",
                               "v_cvt_f32_i32",
                               "v_floor_f32",
                               "v_rndne_f32",
                               "v_med3_f32",
                               "v_add_f32 clamp",
                               "v_fma_f32 clamp",
                               "v_mul_f32 clamp",
                               "v_mov_b32_dpp quad_perm",
                               "v_fma_f32 d,d,s40,v121",
                               "v_cmp + v_cndmask alternating (per pair)",
                               "v_fma_f32 d,d,4.0,v121 (inline constant)",
                               "v_fma_f32 on denormal operands (k * 2^-149 stride)",
                               "v_add_f32 d,s40,d (SGPR operand, VOP2)",
                               "v_mul_f32 d,literal,d",
                               "v_lshlrev_b32",
                               "v_or_b32",
                               "v_xor_b32",
                               "v_sub_u32",
                               "v_bfe_u32",
                               "v_max3_f32",
                               "v_fma_f32 with neg/abs modifiers",
                               "v_add_f32_dpp quad_perm",
                               "v_fmaak_f32 (literal addend)"};

template <int C>
__global__ __launch_bounds__(1024) void rate_kernel(float *sink, float seed) {
    // initialise the registers the blocks use
    asm volatile(
        "v_mov_b32 v120, %0\n\tv_mov_b32 v121, %0\n\tv_mov_b32 v122, %0\n\tv_mov_b32 v123, %0\n\t"
        "s_mov_b64 s[42:43], 0x5555\n\ts_mov_b32 s40, 0x3f800000\n\t"
        "v_mov_b32 v124, 0x41880000\n\tv_mov_b32 v125, 132\n\tv_mov_b32 v126, 4096\n\t"
        :
        : "v"(seed)
        : CLOB);
    for (int it = 0; it < kIters; ++it) {
        if (C == C_ADD) asm volatile(REP16(OP_ADD, 0) : : : CLOB);
        if (C == C_ADD_B3) asm volatile(REP16B3(OP_ADD, 0) : : : CLOB);
        if (C == C_SUB) asm volatile(REP16(OP_SUB, 0) : : : CLOB);
        if (C == C_MUL) asm volatile(REP16(OP_MUL, 0) : : : CLOB);
        if (C == C_FMAC) asm volatile(REP16(OP_FMAC, 0) : : : CLOB);
        if (C == C_FMA3) asm volatile(REP16(OP_FMA3, 0) : : : CLOB);
        if (C == C_FMA3_B3) asm volatile(REP16B3(OP_FMA3, 0) : : : CLOB);
        if (C == C_FMA_OUT_B3) asm volatile(REP16B3(OP_FMA_OUT, 0) : : : CLOB);
        if (C == C_MIN) asm volatile(REP16(OP_MIN, 0) : : : CLOB);
        if (C == C_MAX) asm volatile(REP16(OP_MAX, 0) : : : CLOB);
        if (C == C_MIN3) asm volatile(REP16(OP_MIN3, 0) : : : CLOB);
        if (C == C_MIN3_B3) asm volatile(REP16B3(OP_MIN3, 0) : : : CLOB);
        if (C == C_CMP) asm volatile(REP16(OP_CMP, 0) : : : CLOB);
        if (C == C_CMP_S) asm volatile(REP16(OP_CMP_S, 0) : : : CLOB);
        if (C == C_CND) asm volatile(REP16(OP_CND, 0) : : : CLOB);
        if (C == C_CND_S) asm volatile(REP16(OP_CND_S, 0) : : : CLOB);
        if (C == C_MOV) asm volatile(REP16(OP_MOV, 0) : : : CLOB);
        if (C == C_AND) asm volatile(REP16(OP_AND, 0) : : : CLOB);
        if (C == C_ADDU) asm volatile(REP16(OP_ADDU, 0) : : : CLOB);
        if (C == C_ADDC) asm volatile(REP16(OP_ADDC, 0) : : : CLOB);
        if (C == C_LSHLADD) asm volatile(REP16(OP_LSHLADD, 0) : : : CLOB);
        if (C == C_MAD_U24) asm volatile(REP16(OP_MAD_U24, 0) : : : CLOB);
        if (C == C_CVT_I) asm volatile(REP16(OP_CVT_I, 0) : : : CLOB);
        if (C == C_CVT_F) asm volatile(REP16(OP_CVT_F, 0) : : : CLOB);
        if (C == C_FLOOR) asm volatile(REP16(OP_FLOOR, 0) : : : CLOB);
        if (C == C_RNDNE) asm volatile(REP16(OP_RNDNE, 0) : : : CLOB);
        if (C == C_MED3) asm volatile(REP16(OP_MED3, 0) : : : CLOB);
        if (C == C_ADD_CLAMP) asm volatile(REP16(OP_ADD_CLAMP, 0) : : : CLOB);
        if (C == C_FMA_CLAMP) asm volatile(REP16B3(OP_FMA_CLAMP, 0) : : : CLOB);
        if (C == C_MUL_CLAMP) asm volatile(REP16(OP_MUL_CLAMP, 0) : : : CLOB);
        if (C == C_DPP) asm volatile(REP16(OP_DPP, 0) : : : CLOB);
        if (C == C_FMA_SGPR) asm volatile(REP16(OP_FMA_SGPR, 0) : : : CLOB);
        if (C == C_CMPX) asm volatile(REP16(OP_CMPX, 0) : : : CLOB);
        if (C == C_FMA_INL) asm volatile(REP16(OP_FMA_INL, 0) : : : CLOB);
        if (C == C_FMA_DEN) asm volatile(REP16(OP_FMA_DEN, 0) : : : CLOB);
        if (C == C_ADD_SGPR) asm volatile(REP16(OP_ADD_SGPR, 0) : : : CLOB);
        if (C == C_MUL_LIT) asm volatile(REP16(OP_MUL_LIT, 0) : : : CLOB);
        if (C == C_LSHL) asm volatile(REP16(OP_LSHL, 0) : : : CLOB);
        if (C == C_OR) asm volatile(REP16(OP_OR, 0) : : : CLOB);
        if (C == C_XOR) asm volatile(REP16(OP_XOR, 0) : : : CLOB);
        if (C == C_SUBU) asm volatile(REP16(OP_SUBU, 0) : : : CLOB);
        if (C == C_BFE) asm volatile(REP16(OP_BFE, 0) : : : CLOB);
        if (C == C_MAX3) asm volatile(REP16(OP_MAX3, 0) : : : CLOB);
        if (C == C_FMA_NEG) asm volatile(REP16(OP_FMA_NEG, 0) : : : CLOB);
        if (C == C_ADD_DPP) asm volatile(REP16(OP_ADD_DPP, 0) : : : CLOB);
        if (C == C_MADAK) asm volatile(REP16(OP_MADAK, 0) : : : CLOB);
    }
    float r;
    asm volatile("v_mov_b32 %0, v100" : "=v"(r) : : CLOB);
    sink[blockIdx.x * 1024 + threadIdx.x] = r;
}

double g_add_ns = 0;

template <int C>
void run(int n_cu, float *d_sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<C>, dim3(n_cu), dim3(1024), 0, 0, d_sink, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<C>, dim3(n_cu), dim3(1024), 0, 0, d_sink, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int per_iter = (C == C_CMPX) ? 32 : 16;
    const double ns = ms * 1e6 / ((double)kIters * per_iter * 4);
    if (C == C_ADD) g_add_ns = ns;
    printf("%-52s %6.3f ns per wave-instruction per SIMD  = %4.2f x v_add_f32   (kernel %.2f ms)\n",
           kNames[C], ns, ns / g_add_ns, ms);
}

template <int C>
void run_all(int n_cu, float *d_sink) {
    run<C>(n_cu, d_sink);
    if constexpr (C + 1 < N_CASES) run_all<C + 1>(n_cu, d_sink);
}

int main() {
    int dev = 0, n_cu = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    float *d_sink;
    CHECK(hipMalloc(&d_sink, (size_t)n_cu * 1024 * 4));
    printf("%d CUs, 1024 threads per CU (4 waves per SIMD), %d x 16 instructions per wave\n", n_cu, kIters);
    run_all<0>(n_cu, d_sink);
    return 0;
}
