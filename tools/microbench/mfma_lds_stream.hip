// Dev microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 for the operand patterns the attention kernels use.
//   hipcc --offload-arch=gfx950 -O3 mfma_lds_stream.hip -o mfma_stream && ./mfma_stream
// Variants: accumulator chains (1 dependent chain / 2 / 4 independent), A operand from registers or streamed from LDS
// (ds_read_b128, look-ahead 3, conflict-free 336-B row stride), 1 or 2 waves per SIMD (4 / 8 waves per workgroup, one CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, bool LDS>
__global__ void k(unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[32 * 168 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32 * 168 * 4; i += blockDim.x) tile[i] = (__bf16)(0.001f * (i & 63));
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 b;
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)(0.01f * (lane + j));
    const __bf16* base = tile + (lane & 31) * 168 + 8 * (lane >> 5) + (wave & 3) * 32 * 168;
    bf16x8 fr[4];
    for (int j = 0; j < 8; ++j) fr[0][j] = fr[1][j] = fr[2][j] = fr[3][j] = (__bf16)(0.02f * j);
    if (LDS) { fr[0] = *(const bf16x8*)(base); fr[1] = *(const bf16x8*)(base + 16); fr[2] = *(const bf16x8*)(base + 32); }
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    constexpr int N = 640;  // MFMAs
#pragma unroll 1
    for (int it = 0; it < N / 20; ++it) {
#pragma unroll
        for (int m = 0; m < 20; ++m) {
            if (LDS) fr[(m + 3) & 3] = *(const bf16x8*)(base + 16 * ((m + 3) % 10));
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m & 3], b, acc[m % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0) out[wave] = t1 - t0;
}

template <int NACC, bool LDS>
void run(const char* name, unsigned long long* d, float* sink) {
    for (int waves : {4, 8}) {
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NACC, LDS>), dim3(1), dim3(64 * waves), 0, 0, d, sink);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[w] > mx ? (double)h[w] : mx;
        printf("%-44s %d waves/SIMD: %6.1f cycles per MFMA per wave, %6.1f per MFMA per SIMD\n", name, waves / 4, mx / 640.0, mx / 640.0 / (waves / 4));
    }
}

int main() {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 64);
    hipMalloc(&sink, 4);
    run<1, false>("1 chain, A in registers", d, sink);
    run<2, false>("2 accumulators, A in registers", d, sink);
    run<4, false>("4 accumulators, A in registers", d, sink);
    run<1, true>("1 chain, A streamed from LDS (LA 3)", d, sink);
    run<2, true>("2 accumulators, A streamed from LDS (LA 3)", d, sink);
    run<4, true>("4 accumulators, A streamed from LDS (LA 3)", d, sink);
    return 0;
}
