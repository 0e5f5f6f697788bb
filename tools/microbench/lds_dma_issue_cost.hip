// microbench: issue cost (wave cycles) of one LDS-DMA piece in different forms, every CU busy (2 WG x 4 waves / CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NP 6
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const uint4* g, unsigned long long* out, float* sink) {
    __shared__ uint4 lds[2][4 * NP * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4* src = g + (blockIdx.x % 64) * 4096 + wave * NP * 64 + lane;
    unsigned long long tot = 0;
    uint4 r[NP];
    float acc = 0.f;
    for (int it = 0; it < 64; ++it) {
        uint4* dstb = &lds[it & 1][wave * NP * 64];
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)dstb);
        unsigned long long t0, t1;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0" : "=s"(t0)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 0) {  // save/restore m0 per piece
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src + p * 64), "s"(dst + p * 1024) : "memory");
            }
        } else if constexpr (MODE == 1) {  // m0 per piece, no restore
#pragma unroll
            for (int p = 0; p < NP; ++p)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src + p * 64), "s"(dst + p * 1024) : "memory", "m0");
        } else if constexpr (MODE == 2) {  // m0 once per 4 pieces, immediate offsets
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst) : "memory", "m0");
            asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(src) : "memory");
            asm volatile("global_load_lds_dwordx4 %0, off offset:1024" :: "v"(src) : "memory");
            asm volatile("global_load_lds_dwordx4 %0, off offset:2048" :: "v"(src) : "memory");
            asm volatile("global_load_lds_dwordx4 %0, off offset:3072" :: "v"(src) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst + 4096) : "memory", "m0");
            asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(src + 256) : "memory");
            asm volatile("global_load_lds_dwordx4 %0, off offset:1024" :: "v"(src + 256) : "memory");
        } else if constexpr (MODE == 3) {  // plain loads to registers
#pragma unroll
            for (int p = 0; p < NP; ++p) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[p]) : "v"(src + p * 64) : "memory");
        } else if constexpr (MODE == 4) {  // dword pieces (256 B / instr) x NP
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst) : "memory", "m0");
#pragma unroll
            for (int p = 0; p < NP; ++p) asm volatile("global_load_lds_dword %0, off" :: "v"((const unsigned*)src + p * 64) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0" : "=s"(t1)::"memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tot += t1 - t0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (MODE == 3) {
#pragma unroll
            for (int p = 0; p < NP; ++p) dstb[p * 64 + lane] = r[p];
        }
        __syncthreads();
        // some LDS reads + VALU to mimic a consumer
        for (int j = 0; j < 8; ++j) acc += __uint_as_float(lds[it & 1][(lane + 64 * j + 17 * wave) % (4 * NP * 64)].x);
        __syncthreads();
    }
    if (lane == 0) out[blockIdx.x * 4 + wave] = tot;
    if (acc == 123.456f) sink[0] = acc;
}
int main() {
    uint4* g; unsigned long long* o; float* sink;
    hipMalloc(&g, 64 * 4096 * 16 + 65536); hipMalloc(&o, 512 * 4 * 8); hipMalloc(&sink, 4);
    hipMemset(g, 0, 64 * 4096 * 16 + 65536);
    std::vector<unsigned long long> h(2048);
    auto run = [&](auto kern, const char* name) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, g, o, sink); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), o, 2048 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        printf("%-40s %.1f cycles per piece-issue (6 per iter)\n", name, s / 2048 / 64 / NP);
    };
    run(k<0>, "m0 save/restore per piece");
    run(k<1>, "m0 per piece, no restore");
    run(k<2>, "m0 per 4 pieces + imm offsets");
    run(k<3>, "global_load_dwordx4 -> VGPR");
    run(k<4>, "lds dword pieces (256B)");
    return 0;
}
