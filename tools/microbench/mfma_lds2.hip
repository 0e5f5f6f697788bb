// Dev microbenchmark (round 4): what does one CU's LDS deliver beside a saturated matrix pipe?
//   hipcc --offload-arch=gfx950 -O3 mfma_lds2.hip -o mfma_lds2 && ./mfma_lds2 [out.txt]
// v_mfma_f32_32x32x16_bf16 chains whose operands come from
//   R0  registers only                              (the pure-MFMA peak and the clock the chip holds under it)
//   L1  A = one ds_read_b128 per MFMA               (what the round-2/3 attention kernels do: one streamed operand)
//   L2  A and B = two ds_read_b128 per MFMA         (both operands streamed)
//   LT  A = ds_read_b128, B = 2 x ds_read_b64_tr_b16 (one operand read transposed)
//   L3  three ds_read_b128 per MFMA                 (the guide's saturation point: 48 cycles per gap)
//   L4  four ds_read_b128 per MFMA
// at 1 and 2 waves per SIMD, on one CU and on every CU (one workgroup per CU, forced by the LDS size). Random operands
// (DVFS answers to data), look-ahead 3 fragments, conflict-free images. Prints cycles per MFMA per SIMD, LDS bytes per
// clock and CU, the shader clock (s_memtime against the 100 MHz s_memrealtime) and TFLOP/s from the wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define DEVFN __device__ __forceinline__
DEVFN bf16x4 tr_read(const __bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}
DEVFN bf16x8 tr_frag(const __bf16* img, int stride, int k0, int lane) {
    const int g = lane >> 4, i = lane & 15, h = g >> 1, q = i >> 2, p = i & 3;
    const __bf16* a = img + (k0 + 4 * h + q) * stride + 16 * (g & 1) + 4 * p;
    bf16x4 lo = tr_read(a), hi = tr_read(a + 8 * stride);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

enum { R0 = 0, L1 = 1, L2 = 2, LT = 3, L3 = 4, L4 = 5 };
constexpr int ROWS = 168;     // elements per image row (336 B: conflict-free for ds_read_b128 over 32 rows)
constexpr int TRS = 160;      // elements per row of the transposed image (80 dwords = 16 mod 64: conflict-free tr reads)
constexpr int IMG = 32 * ROWS;  // one 32-row image per wave and operand

template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned long long* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // images: per wave A, B, C, D (row-major 32 x 160 (+8 pad)); the transposed image for LT reuses B's region as [k = 160][m = 32..]
    unsigned s = 0x9E3779B9u * (blockIdx.x * 512 + threadIdx.x + 1);
    for (int i = threadIdx.x; i < 4 * IMG; i += blockDim.x) {  // the images are shared by all waves: only addresses WITHIN one wave-instruction can conflict or broadcast
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        lds[i] = (__bf16)(((int)(s & 0xFFFF) - 32768) * (1.0f / 32768.f));
    }
    __syncthreads();
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const __bf16* A = lds + 0 * IMG + (lane & 31) * ROWS + 8 * (lane >> 5);
    const __bf16* B = lds + 1 * IMG + (lane & 31) * ROWS + 8 * (lane >> 5);
    const __bf16* C = lds + 2 * IMG + (lane & 31) * ROWS + 8 * (lane >> 5);
    const __bf16* D = lds + 3 * IMG + (lane & 31) * ROWS + 8 * (lane >> 5);
    const __bf16* T = lds + 1 * IMG;  // 32 x 160 region read as [k = 32][m stride 160] by the tr reads
    bf16x8 fa[4], fb[4], fc[4], fd[4];
    for (int j = 0; j < 8; ++j)
        for (int q = 0; q < 4; ++q) {
            fa[q][j] = (__bf16)(0.01f * (lane + j + q) - 0.3f);
            fb[q][j] = (__bf16)(0.3f - 0.01f * (lane + j + q));
            fc[q][j] = fa[q][j];
            fd[q][j] = fb[q][j];
        }
    if (MODE >= L1) for (int q = 0; q < 3; ++q) fa[q] = *(const bf16x8*)(A + 16 * q);
    if (MODE == L2 || MODE >= L3) for (int q = 0; q < 3; ++q) fb[q] = *(const bf16x8*)(B + 16 * q);
    if (MODE == LT) for (int q = 0; q < 3; ++q) fb[q] = tr_frag(T, TRS, 0, lane);
    if (MODE >= L3) for (int q = 0; q < 3; ++q) fc[q] = *(const bf16x8*)(C + 16 * q);
    if (MODE >= L4) for (int q = 0; q < 3; ++q) fd[q] = *(const bf16x8*)(D + 16 * q);
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 20; ++m) {
            const int nx = (m + 3) & 3, col = 16 * ((m + 3) % 10);
            if (MODE >= L1) fa[nx] = *(const bf16x8*)(A + col);
            if (MODE == L2 || MODE >= L3) fb[nx] = *(const bf16x8*)(B + col);
            if (MODE == LT) fb[nx] = tr_frag(T, TRS, 16 * ((m + 3) & 1), lane);
            if (MODE >= L3) fc[nx] = *(const bf16x8*)(C + col);
            if (MODE >= L4) fd[nx] = *(const bf16x8*)(D + col);
            bf16x8 a = fa[m & 3], b = fb[m & 3];
            if (MODE >= L3) asm volatile("" ::"v"(fc[m & 3]));  // the extra fragments stay whole 16-B reads: used, at no instruction
            if (MODE >= L4) asm volatile("" ::"v"(fd[m & 3]));
            acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float sum = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 16; ++r) sum += acc[a][r];
    if (sum == 12345.678f) sink[0] = sum;
    if (lane == 0) {
        out[(blockIdx.x * 8 + wave) * 2 + 0] = t1 - t0;
        out[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
    }
}

static FILE* g_out = nullptr;
static void emit(const char* s) {
    fputs(s, stdout);
    if (g_out) fputs(s, g_out);
}

template <int MODE>
void run(const char* name, int lds_bytes_per_mfma, unsigned long long* d, float* sink, int ncu) {
    char line[512];
    for (int grid : {1, ncu}) {
        for (int waves : {4, 8}) {
            const int iters = 1600;  // x 20 MFMAs = 32 000 MFMAs per wave: >= 1 M cycles, ~0.5 ms
            const size_t shm = 150 * 1024;  // one workgroup per CU
            hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            float best = 1e30f;
            for (int r = 0; r < 4; ++r) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64 * waves), shm, 0, d, sink, iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (r > 0 && ms < best) best = ms;
            }
            std::vector<unsigned long long> h((size_t)grid * 16);
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0, cmax = 0;
            int n = 0;
            for (int b = 0; b < grid; ++b)
                for (int w = 0; w < waves; ++w) {
                    const double c = (double)h[(b * 8 + w) * 2], t = (double)h[(b * 8 + w) * 2 + 1];
                    cyc += c; rt += t; cmax = c > cmax ? c : cmax; ++n;
                }
            cyc /= n; rt /= n;
            const double mf = 20.0 * iters, wps = waves / 4.0;
            // per SIMD: the LAST wave to finish bounds the pipe's busy time (with two waves the older one is served first and finishes at
            // half time, so the mean over waves would read 24 cycles per MFMA on a 32-cycle pipe)
            const double cpm_simd = cmax / mf / wps;               // cycles per MFMA per SIMD
            const double bclk = lds_bytes_per_mfma * 4.0 / cpm_simd;  // 4 SIMDs
            const double ghz = cyc / (rt * 10.0) ;                 // realtime ticks are 10 ns
            const double tflops = (double)grid * waves * mf * 32768.0 / (best * 1e-3) * 1e-12;
            snprintf(line, sizeof line,
                     "%-34s %3d CU %d w/SIMD: %6.2f cyc/MFMA/SIMD (mean wave %6.2f)  LDS %6.1f B/clk/CU  clock %.3f GHz  %8.1f TFLOP/s wall (%.3f ms)\n",
                     name, grid, waves / 4, cpm_simd, cyc / mf / wps, bclk, ghz, tflops, best);
            emit(line);
        }
    }
}

int main(int argc, char** argv) {
    if (argc > 1) g_out = fopen(argv[1], "w");
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    char line[256];
    snprintf(line, sizeof line, "# %s, %d CUs, clockRate %d kHz; v_mfma_f32_32x32x16_bf16 = 32768 FLOP; nominal peak 2.5 PFLOP/s\n", p.name, ncu, p.clockRate);
    emit(line);
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, (size_t)ncu * 16 * 8);
    hipMalloc(&sink, 4);
    run<R0>("R0 operands in registers", 0, d, sink, ncu);
    run<L1>("L1 A: 1 ds_read_b128", 1024, d, sink, ncu);
    run<L2>("L2 A,B: 2 ds_read_b128", 2048, d, sink, ncu);
    run<LT>("LT A: b128, B: 2 ds_read_b64_tr_b16", 2048, d, sink, ncu);
    run<L3>("L3 3 ds_read_b128", 3072, d, sink, ncu);
    run<L4>("L4 4 ds_read_b128", 4096, d, sink, ncu);
    if (g_out) fclose(g_out);
    return 0;
}
