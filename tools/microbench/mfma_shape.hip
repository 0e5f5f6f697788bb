// Dev microbenchmark (round 4): does the MFMA SHAPE change what the power-limited chip sustains? v_mfma_f32_32x32x16_bf16 against
// v_mfma_f32_16x16x32_bf16, operands in registers (R) or both re-read from LDS with ds_read_b128 (L), one wave per SIMD, all CUs, random data.
//   hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape && ./mfma_shape [out.txt]
// MI355X_MICROARCH.md reports 1.12-1.15 x the FLOP/s for the 16x16x32 form at equal cycles per FLOP (it moves half the accumulator registers
// per FLOP); this measures it on the box at hand.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int ROWS = 168, IMG = 32 * ROWS;

template <bool SMALL, bool LDS>
__global__ void __launch_bounds__(256) k(unsigned long long* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
    for (int i = threadIdx.x; i < 2 * IMG; i += blockDim.x) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        lds[i] = (__bf16)(((int)(s & 0xFFFF) - 32768) * (1.0f / 32768.f));
    }
    __syncthreads();
    const __bf16* A = lds + (lane & 31) * ROWS + 8 * (lane >> 5);
    const __bf16* B = lds + IMG + (lane & 31) * ROWS + 8 * (lane >> 5);
    bf16x8 fa[4], fb[4];
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 8; ++j) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            fa[q][j] = (__bf16)(((int)(s & 0xFFFF) - 32768) * (1.0f / 32768.f));
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            fb[q][j] = (__bf16)(((int)(s & 0xFFFF) - 32768) * (1.0f / 32768.f));
        }
    if (LDS) for (int q = 0; q < 3; ++q) { fa[q] = *(const bf16x8*)(A + 16 * q); fb[q] = *(const bf16x8*)(B + 16 * q); }
    f32x16 acc[2];
    f32x4 acs[8];
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 4; ++r) acs[a][r] = 0.f;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 20; ++m) {
            const int nx = (m + 3) & 3, col = 16 * ((m + 3) % 10);
            if (LDS) { fa[nx] = *(const bf16x8*)(A + col); fb[nx] = *(const bf16x8*)(B + col); }
            if (SMALL) {  // the same 32 KFLOP as one 32x32x16: two 16x16x32 (each reads its own operand pair in the L form: twice the LDS bytes per FLOP)
                acs[(2 * m) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m & 3], fb[m & 3], acs[(2 * m) & 7], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (LDS) { fa[nx] = *(const bf16x8*)(B + col); fb[nx] = *(const bf16x8*)(A + col); }
                acs[(2 * m + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[(m + 1) & 3], fa[(m + 1) & 3], acs[(2 * m + 1) & 7], 0, 0, 0);
            } else {
                acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m & 3], fb[m & 3], acc[m & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[0][r] + acc[1][r];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 4; ++r) sum += acs[a][r];
    if (sum == 12345.678f) sink[0] = sum;
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

static FILE* g_out = nullptr;
template <bool SMALL, bool LDS>
void run(const char* name, unsigned long long* d, float* sink, int ncu) {
    const int iters = 8000;  // x 20 x 32 KFLOP per wave: ~10 ms per launch
    const size_t shm = 150 * 1024;
    hipFuncSetAttribute((const void*)k<SMALL, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float last = 0;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<SMALL, LDS>), dim3(ncu), dim3(256), shm, 0, d, sink, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1);
    }
    std::vector<unsigned long long> h((size_t)ncu * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < ncu * 4; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double units = 20.0 * iters;  // 32-KFLOP units per wave
    char line[256];
    snprintf(line, sizeof line, "%-44s %6.2f cycles per 32 KFLOP and SIMD   clock %.3f GHz   %8.1f TFLOP/s (last of 5 launches, %.2f ms)\n", name,
             cyc / (ncu * 4) / units, cyc / (rt * 10.0), (double)ncu * 4 * units * 32768.0 / (last * 1e-3) * 1e-12, last);
    fputs(line, stdout);
    if (g_out) fputs(line, g_out);
}
int main(int argc, char** argv) {
    if (argc > 1) g_out = fopen(argv[1], "w");
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    unsigned long long* d; float* sink;
    hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 8); hipMalloc(&sink, 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<false, false>("32x32x16, operands in registers", d, sink, p.multiProcessorCount);
        run<true, false>("16x16x32 x 2, operands in registers", d, sink, p.multiProcessorCount);
        run<false, true>("32x32x16, A and B from LDS (2 KB / 32 KFLOP)", d, sink, p.multiProcessorCount);
        run<true, true>("16x16x32 x 2, A and B from LDS (4 KB / 32 KFLOP)", d, sink, p.multiProcessorCount);
    }
    if (g_out) fclose(g_out);
    return 0;
}
