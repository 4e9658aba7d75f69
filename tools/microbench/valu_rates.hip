// Issue-rate micro-benchmark for the gfx950 VALU instructions the SA kernels lean on.  One workgroup of
// 64*W threads (W waves on one CU); each wave runs REP x 8 independent copies of one instruction between two
// s_memtime reads.  Prints cycles per instruction per wave and per SIMD.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP 512

#define BENCH8(NAME, ASM)                                                                     \
    __global__ void NAME(unsigned long long *out, float seed) {                               \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4,         \
              a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;                                    \
        float b0 = seed * 2, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, \
              b6 = b0 + 6, b7 = b0 + 7;                                                       \
        __syncthreads();                                                                      \
        unsigned long long t0 = __builtin_readcyclecounter();                                 \
        for (int i = 0; i < REP; ++i) {                                                       \
            asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),   \
                         "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),         \
                         "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)::"vcc", "s20", "s21");       \
        }                                                                                     \
        unsigned long long t1 = __builtin_readcyclecounter();                                 \
        __syncthreads();                                                                      \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                         \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 12345.f) out[63] = 1; \
    }

// 8 independent instructions; operands %0..%7 = a, %8..%15 = b
#define I8(op) op(0, 8) op(1, 9) op(2, 10) op(3, 11) op(4, 12) op(5, 13) op(6, 14) op(7, 15)
#define S(x) #x
#define FMA(a, b) "v_fma_f32 %" S(a) ", %" S(a) ", %" S(b) ", %" S(b) "\n"
#define MAX(a, b) "v_max_f32 %" S(a) ", %" S(a) ", %" S(b) "\n"
#define MAX3(a, b) "v_max3_f32 %" S(a) ", %" S(a) ", %" S(b) ", %" S(b) "\n"
#define BFE(a, b) "v_bfe_u32 %" S(a) ", %" S(b) ", 16, 1\n"
#define ADD3(a, b) "v_add3_u32 %" S(a) ", %" S(a) ", %" S(b) ", %" S(b) "\n"
#define PERM(a, b) "v_perm_b32 %" S(a) ", %" S(a) ", %" S(b) ", %" S(b) "\n"
#define AND(a, b) "v_and_b32 %" S(a) ", 0xffff0000, %" S(b) "\n"
#define CVTBF(a, b) "v_cvt_pk_bf16_f32 %" S(a) ", %" S(a) ", %" S(b) "\n"
#define CVTF16(a, b) "v_cvt_pkrtz_f16_f32 %" S(a) ", %" S(a) ", %" S(b) "\n"
#define CMPS(a, b) "v_cmp_gt_f32 s[20:21], %" S(a) ", %" S(b) "\n"
#define CNDM(a, b) "v_cndmask_b32 %" S(a) ", %" S(a) ", %" S(b) ", vcc\n"
#define DPP(a, b) "v_max_f32_dpp %" S(a) ", %" S(b) ", %" S(b) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define DEPFMA(a, b) "v_fma_f32 %0, %0, %8, %8\n"
#define DEPDPP(a, b) "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\ns_nop 1\n"

BENCH8(k_fma, I8(FMA))
BENCH8(k_max, I8(MAX))
BENCH8(k_max3, I8(MAX3))
BENCH8(k_bfe, I8(BFE))
BENCH8(k_add3, I8(ADD3))
BENCH8(k_perm, I8(PERM))
BENCH8(k_and, I8(AND))
BENCH8(k_cvtbf, I8(CVTBF))
BENCH8(k_cvtf16, I8(CVTF16))
BENCH8(k_cmps, I8(CMPS))
BENCH8(k_cndm, I8(CNDM))
BENCH8(k_dpp, I8(DPP))
#define CNDS(a, b) "v_cndmask_b32_e64 %" S(a) ", %" S(a) ", %" S(b) ", s[20:21]\n"
#define FMAS(a, b) "v_fma_f32 %" S(a) ", %" S(a) ", s20, %" S(b) "\n"
#define SUBS(a, b) "v_sub_f32 %" S(a) ", s20, %" S(b) "\n"
#define SUBV(a, b) "v_sub_f32 %" S(a) ", %" S(a) ", %" S(b) "\n"
#define MIN(a, b) "v_min_f32 %" S(a) ", %" S(a) ", %" S(b) "\n"
#define MULV(a, b) "v_mul_f32 %" S(a) ", %" S(a) ", %" S(b) "\n"
#define FMAC(a, b) "v_fmac_f32 %" S(a) ", %" S(b) ", %" S(b) "\n"
#define RDL(a, b) "v_readlane_b32 s20, %" S(a) ", 3\n"
#define SALU(a, b) "s_bitcmp1_b64 s[20:21], 5\ns_cselect_b32 s20, 7, s20\n"
#define LSHL(a, b) "v_lshlrev_b32 %" S(a) ", 16, %" S(b) "\n"
BENCH8(k_cnds, I8(CNDS))
BENCH8(k_fmas, I8(FMAS))
BENCH8(k_subs, I8(SUBS))
BENCH8(k_subv, I8(SUBV))
BENCH8(k_min, I8(MIN))
BENCH8(k_mulv, I8(MULV))
BENCH8(k_fmac, I8(FMAC))
BENCH8(k_rdl, I8(RDL))
BENCH8(k_salu, I8(SALU))
BENCH8(k_lshl, I8(LSHL))
BENCH8(k_depfma, I8(DEPFMA))
BENCH8(k_depdpp, I8(DEPDPP))

// packed fp32: operands are register pairs
#define BENCHPK(NAME, ASM)                                                                    \
    __global__ void NAME(unsigned long long *out, float seed) {                               \
        typedef float f2 __attribute__((ext_vector_type(2)));                                 \
        f2 a0 = {seed, seed + 1}, a1 = a0 + 2.f, a2 = a0 + 4.f, a3 = a0 + 6.f, a4 = a0 + 8.f, a5 = a0 + 10.f, \
           a6 = a0 + 12.f, a7 = a0 + 14.f, b0 = a0 * 0.5f;                                    \
        __syncthreads();                                                                      \
        unsigned long long t0 = __builtin_readcyclecounter();                                 \
        for (int i = 0; i < REP; ++i) {                                                       \
            asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),   \
                         "+v"(a6), "+v"(a7), "+v"(b0));                                       \
        }                                                                                     \
        unsigned long long t1 = __builtin_readcyclecounter();                                 \
        __syncthreads();                                                                      \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                         \
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0;                                    \
        if (s[0] + s[1] == 12345.f) out[63] = 1;                                              \
    }
#define P8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define PKFMA(a) "v_pk_fma_f32 %" S(a) ", %" S(a) ", %8, %8\n"
#define PKADD(a) "v_pk_add_f32 %" S(a) ", %" S(a) ", %8\n"
#define PKMUL(a) "v_pk_mul_f32 %" S(a) ", %" S(a) ", %8\n"
#define PKADDNEG(a) "v_pk_add_f32 %" S(a) ", %" S(a) ", %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
BENCHPK(k_pkfma, P8(PKFMA))
BENCHPK(k_pkadd, P8(PKADD))
BENCHPK(k_pkmul, P8(PKMUL))
BENCHPK(k_pkaddneg, P8(PKADDNEG))

typedef void (*kern_t)(unsigned long long *, float);
struct Entry { const char *name; kern_t k; int per_iter; };

int main() {
    Entry es[] = {{"v_fma_f32", k_fma, 8}, {"v_max_f32", k_max, 8}, {"v_max3_f32", k_max3, 8},
                  {"v_bfe_u32", k_bfe, 8}, {"v_add3_u32", k_add3, 8}, {"v_perm_b32", k_perm, 8},
                  {"v_and_b32", k_and, 8}, {"v_cvt_pk_bf16_f32", k_cvtbf, 8}, {"v_cvt_pkrtz_f16_f32", k_cvtf16, 8},
                  {"v_cmp_gt_f32 -> sgpr", k_cmps, 8}, {"v_cndmask_b32", k_cndm, 8}, {"v_max_f32_dpp", k_dpp, 8},
                  {"v_cndmask_b32_e64 sgpr mask", k_cnds, 8}, {"v_fma_f32 sgpr src", k_fmas, 8}, {"v_sub_f32 sgpr src", k_subs, 8},
                  {"v_sub_f32", k_subv, 8}, {"v_min_f32", k_min, 8}, {"v_mul_f32", k_mulv, 8}, {"v_fmac_f32", k_fmac, 8},
                  {"v_lshlrev_b32", k_lshl, 8},
                  {"dependent v_fma_f32", k_depfma, 8}, {"dependent dpp max + s_nop 1", k_depdpp, 8},
                  {"v_pk_fma_f32", k_pkfma, 8}, {"v_pk_add_f32", k_pkadd, 8}, {"v_pk_mul_f32", k_pkmul, 8},
                  {"v_pk_add_f32 op_sel/neg", k_pkaddneg, 8}};
    unsigned long long *d, h[64];
    hipMalloc(&d, 64 * 8);
    printf("cycles (s_memtime) per instruction per SIMD\n%-30s %10s %10s %10s %10s\n", "instruction", "1 wave", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD"); fflush(stdout);
    for (auto &e : es) {
        double r[4];
        int cfg[4] = {64, 256, 512, 1024};
        for (int c = 0; c < 4; ++c) {
            hipMemset(d, 0, 64 * 8);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(cfg[c]), 0, 0, d, 1.0f);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(cfg[c]), 0, 0, d, 1.0f);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0;
            for (int w = 0; w < cfg[c] / 64; ++w) if (h[w] > mx) mx = h[w];
            // per SIMD: waves/4 waves share a SIMD
            int waves_per_simd = cfg[c] / 64 >= 4 ? cfg[c] / 64 / 4 : 1;
            r[c] = (double)mx / ((double)REP * e.per_iter * waves_per_simd);
        }
        printf("%-30s %10.2f %10.2f %10.2f %10.2f\n", e.name, r[0], r[1], r[2], r[3]); fflush(stdout);
    }
    return 0;
}
