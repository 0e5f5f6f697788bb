// v1t_amd - measurement aid: the bf16 MFMA rate and the shader clock THIS device sustains (gfx950).
// bench.py runs it in its un-timed set-up and reports roofline.peak_measured / frac_of_measured next to the nominal 2.5 PFLOP/s
// ("against both datasheet and measured peak"). No reference counterpart; not on any compute path.
#include <vector>
#include "common.h"
#include "../../include/v1t_amd.h"

// v_mfma_f32_32x32x16_bf16 back to back, operands in registers, random bit patterns of moderate magnitude (the clock the chip
// holds depends on the data: zeros run faster than the values a training step multiplies), two accumulator chains per wave,
// WAVES waves per workgroup (64 * WAVES threads), one workgroup per CU and launch round.
__global__ __launch_bounds__(512) void mfma_peak_kernel(unsigned long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned s = 0x9E3779B9u * (blockIdx.x * 512 + threadIdx.x + 1);
    bf16x8 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            a[q][j] = (bf16_t)(((int)(s & 0xFFFF) - 32768) * (1.0f / 32768.f));
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            b[q][j] = (bf16_t)(((int)(s & 0xFFFF) - 32768) * (1.0f / 32768.f));
        }
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m & 1] = mfma32(a[m & 3], b[(m >> 1) & 3], acc[m & 1]);
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[0][r] + acc[1][r];
    if (sum == 12345.678f) sink[0] = sum;  // keeps the chains alive
    if (lane == 0) {
        out[2 * (blockIdx.x * 8 + (threadIdx.x >> 6))] = t1 - t0;
        out[2 * (blockIdx.x * 8 + (threadIdx.x >> 6)) + 1] = r1 - r0;
    }
}

extern "C" int v1t_mfma_peak_probe(int iters, int waves_per_simd, double* tflops, double* ghz, double* cycles_per_mfma, void* stream) {
    if (iters <= 0 || (waves_per_simd != 1 && waves_per_simd != 2) || !tflops || !ghz || !cycles_per_mfma) return V1T_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return V1T_ERR_LAUNCH;
    const int waves = 4 * waves_per_simd;
    unsigned long long* d = nullptr;
    float* sink = nullptr;
    if (hipMalloc(&d, (size_t)ncu * 16 * 8) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return V1T_ERR_WORKSPACE;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // The chip is power-managed: the first milliseconds of matrix load still run near 2.4 GHz, then the clock settles (~1.5-1.6 GHz).
    // The LAST of four back-to-back launches is reported (callers pass iters for >= 10 ms per launch): the sustained rate, not the burst.
    float last = 0.f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(mfma_peak_kernel, dim3(ncu), dim3(64 * waves), 0, s, d, sink, iters);
        hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) { hipFree(d); hipFree(sink); return V1T_ERR_LAUNCH; }
        hipEventElapsedTime(&last, e0, e1);
    }
    std::vector<unsigned long long> h((size_t)ncu * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0, cmax = 0;
    for (int b = 0; b < ncu; ++b)
        for (int w = 0; w < waves; ++w) {
            const double c = (double)h[2 * (b * 8 + w)];
            cyc += c; rt += (double)h[2 * (b * 8 + w) + 1];
            cmax = c > cmax ? c : cmax;
        }
    const double mf = 16.0 * iters;
    *cycles_per_mfma = cmax / mf / waves_per_simd;  // per SIMD: the last wave to finish bounds the pipe's busy time
    *ghz = cyc / (rt * 10.0);                        // s_memrealtime ticks are 10 ns
    *tflops = (double)ncu * waves * mf * 32768.0 / (last * 1e-3) * 1e-12;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    hipFree(sink);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
