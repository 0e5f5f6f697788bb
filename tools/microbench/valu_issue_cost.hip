// Dev microbenchmark: issue cost (cycles per wave-instruction) of the VALU ops the attention element-wise stages use,
// one wave per SIMD and two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_issue_cost.hip -o valu && ./valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, asmtext)                                                                              \
    __global__ void k_##name(unsigned long long* out, unsigned seed) {                                   \
        unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x1234567u, a2 = a0 + 77u, a3 = a0 * 3u; \
        unsigned b0 = a0 >> 3, b1 = a1 >> 5, b2 = a2 >> 7, b3 = a3 >> 9;                                 \
        unsigned long long t0, t1;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                      \
        for (int it = 0; it < 64; ++it) {                                                                \
            asm volatile(REP16(asmtext) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)::"vcc"); \
        }                                                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                      \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                \
        if (a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 == 0x12345) out[1000] = 1;                             \
    }

// each body = 4 independent instructions (one per register set)
BODY(mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %7\n")
BODY(mul_u24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %5\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %7\n")
BODY(mad_u24, "v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %5, %2\n v_mad_u32_u24 %2, %2, %6, %3\n v_mad_u32_u24 %3, %3, %7, %0\n")
BODY(mul_hi, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %5\n v_mul_hi_u32 %2, %2, %6\n v_mul_hi_u32 %3, %3, %7\n")
BODY(xor_, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7\n")
BODY(lshr, "v_lshrrev_b32 %0, 3, %4\n v_lshrrev_b32 %1, 5, %5\n v_lshrrev_b32 %2, 7, %6\n v_lshrrev_b32 %3, 9, %7\n")
BODY(xad, "v_xad_u32 %0, %0, %4, %1\n v_xad_u32 %1, %1, %5, %2\n v_xad_u32 %2, %2, %6, %3\n v_xad_u32 %3, %3, %7, %0\n")
BODY(bfe, "v_bfe_u32 %0, %4, 8, 8\n v_bfe_u32 %1, %5, 16, 8\n v_bfe_u32 %2, %6, 0, 8\n v_bfe_u32 %3, %7, 24, 8\n")
BODY(alignbit, "v_alignbit_b32 %0, %0, %4, 15\n v_alignbit_b32 %1, %1, %5, 15\n v_alignbit_b32 %2, %2, %6, 15\n v_alignbit_b32 %3, %3, %7, 15\n")
BODY(perm, "v_perm_b32 %0, %0, %4, %1\n v_perm_b32 %1, %1, %5, %2\n v_perm_b32 %2, %2, %6, %3\n v_perm_b32 %3, %3, %7, %0\n")
BODY(cmp_cnd, "v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u32 vcc, %2, %6\n v_cndmask_b32 %3, %3, %7, vcc\n")
BODY(cmp_sdwa, "v_cmp_lt_u32_sdwa vcc, %0, %4 src0_sel:BYTE_1 src1_sel:DWORD\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u32_sdwa vcc, %2, %6 src0_sel:BYTE_2 src1_sel:DWORD\n v_cndmask_b32 %3, %3, %7, vcc\n")
BODY(exp, "v_exp_f32 %0, %4\n v_exp_f32 %1, %5\n v_exp_f32 %2, %6\n v_exp_f32 %3, %7\n")
BODY(fma, "v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %5, %2\n v_fma_f32 %2, %2, %6, %3\n v_fma_f32 %3, %3, %7, %0\n")
BODY(mul_f32, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7\n")
BODY(cvt_pk, "v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4\n")
BODY(mov_dpp, "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
BODY(xor_dpp, "v_xor_b32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %1, %5, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %2, %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %3, %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
BODY(and_or, "v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %5, %2\n v_and_or_b32 %2, %2, %6, %3\n v_and_or_b32 %3, %3, %7, %0\n")
BODY(lshl_add, "v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 5, %5\n v_lshl_add_u32 %2, %2, 7, %6\n v_lshl_add_u32 %3, %3, 9, %7\n")
BODY(add3, "v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %5, %2\n v_add3_u32 %2, %2, %6, %3\n v_add3_u32 %3, %3, %7, %0\n")
BODY(cndmask_imm, "v_cndmask_b32 %0, 0, %4, vcc\n v_cndmask_b32 %1, 0, %5, vcc\n v_cndmask_b32 %2, 0, %6, vcc\n v_cndmask_b32 %3, 0, %7, vcc\n")

typedef void (*kern_t)(unsigned long long*, unsigned);
struct Case { const char* name; kern_t k; };

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8192 * 8);
    std::vector<Case> cases = {
        {"v_mul_lo_u32", k_mul_lo}, {"v_mul_u32_u24", k_mul_u24}, {"v_mad_u32_u24", k_mad_u24}, {"v_mul_hi_u32", k_mul_hi}, {"v_xor_b32", k_xor_},
        {"v_lshrrev_b32", k_lshr}, {"v_xad_u32", k_xad}, {"v_bfe_u32", k_bfe}, {"v_alignbit_b32", k_alignbit}, {"v_perm_b32", k_perm},
        {"v_cmp+v_cndmask", k_cmp_cnd}, {"v_cmp_sdwa+cndmask", k_cmp_sdwa}, {"v_exp_f32", k_exp}, {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul_f32},
        {"v_cvt_pk_bf16_f32", k_cvt_pk}, {"v_mov_b32_dpp", k_mov_dpp}, {"v_xor_b32_dpp", k_xor_dpp}, {"v_and_or_b32", k_and_or}, {"v_lshl_add_u32", k_lshl_add},
        {"v_add3_u32", k_add3}, {"v_cndmask(0,x)", k_cndmask_imm},
    };
    for (int waves : {4, 8, 16}) {  // per workgroup of one CU: 1, 2, 4 waves per SIMD
        printf("---- %d waves per CU (%d per SIMD), cycles per wave-instruction (s_memtime ticks / instructions)\n", waves, waves / 4);
        for (auto& c : cases) {
            hipMemset(d, 0, 8192 * 8);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(c.k, dim3(1), dim3(64 * waves), 0, 0, d, 5u + r);
            hipDeviceSynchronize();
            unsigned long long h[16];
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double mx = 0;
            for (int w = 0; w < waves; ++w) mx = h[w] > mx ? (double)h[w] : mx;
            printf("%-22s %7.2f\n", c.name, mx / (64.0 * 16 * 4));
        }
    }
    return 0;
}
