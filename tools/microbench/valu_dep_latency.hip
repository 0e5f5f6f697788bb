// Dev microbenchmark: cycles per vector instruction of ONE wave per SIMD when every instruction depends on the one before it (a chain) against four
// independent chains interleaved (tools/microbench/valu_issue_cost.hip measures the latter).  hipcc --offload-arch=gfx950 -O3 valu_dep_latency.hip -o valu_dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, asmtext)                                                                              \
    __global__ void k_##name(unsigned long long* out, unsigned seed) {                                   \
        unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x1234567u, a2 = a0 + 77u, a3 = a0 * 3u; \
        unsigned b0 = a0 >> 3;                                                                           \
        unsigned long long t0, t1;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                      \
        for (int it = 0; it < 64; ++it) {                                                                \
            asm volatile(REP16(asmtext) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0)::"vcc");      \
        }                                                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                      \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                \
        if (a0 + a1 + a2 + a3 + b0 == 0x12345) out[1000] = 1;                                            \
    }
// 4 instructions per body
BODY(xor_dep, "v_xor_b32 %0, %0, %4\n v_xor_b32 %0, %0, %4\n v_xor_b32 %0, %0, %4\n v_xor_b32 %0, %0, %4\n")
BODY(xor_ind, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n")
BODY(xor_2ch, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n")
BODY(fma_dep, "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n")
BODY(fma_ind, "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n")
BODY(mul_lo_dep, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %0, %0, %4\n")
BODY(exp_dep, "v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n")
BODY(exp_ind, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
BODY(sdwa_dep, "v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
BODY(cmp_cnd_dep, "v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n")
BODY(mix, "v_fma_f32 %0, %0, %4, %4\n v_exp_f32 %0, %0\n v_add_f32 %1, %0, %1\n v_cndmask_b32 %2, 0, %0, vcc\n")
typedef void (*kern_t)(unsigned long long*, unsigned);
struct Case { const char* name; kern_t k; };
int main() {
    unsigned long long* d;
    hipMalloc(&d, 8192 * 8);
    Case cases[] = {{"v_xor dependent chain", k_xor_dep}, {"v_xor two chains", k_xor_2ch}, {"v_xor four chains", k_xor_ind}, {"v_fma dependent", k_fma_dep}, {"v_fma four chains", k_fma_ind},
                    {"v_mul_lo_u32 dependent", k_mul_lo_dep}, {"v_exp dependent", k_exp_dep}, {"v_exp four chains", k_exp_ind}, {"v_xor_sdwa dependent", k_sdwa_dep},
                    {"v_cmp -> v_cndmask dependent", k_cmp_cnd_dep}, {"fma -> exp -> add, cndmask (softmax element)", k_mix}};
    for (int waves : {4, 8}) {
        printf("---- %d waves per CU (%d per SIMD), cycles per wave-instruction\n", waves, waves / 4);
        for (auto& c : cases) {
            hipMemset(d, 0, 8192 * 8);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(c.k, dim3(1), dim3(64 * waves), 0, 0, d, 5u + r);
            hipDeviceSynchronize();
            unsigned long long h[16];
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double mx = 0;
            for (int w = 0; w < waves; ++w) mx = h[w] > mx ? (double)h[w] : mx;
            printf("%-48s %7.2f\n", c.name, mx / (64.0 * 16 * 4));
        }
    }
    return 0;
}
