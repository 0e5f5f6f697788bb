#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* g, float* out) {
    __shared__ float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -1.f;
    __syncthreads();
    unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g + threadIdx.x * 4), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory", "m0");
    asm volatile("global_load_lds_dwordx4 %0, off offset:2048" :: "v"(g + threadIdx.x * 4) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *g, *o;
    hipMalloc(&g, 4096 * 4); hipMalloc(&o, 2048 * 4);
    hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(g, o);
    std::vector<float> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    for (int i : {0, 1, 255, 256, 511, 512, 513, 767, 768, 1023, 1024}) printf("lds[%d]=%g\n", i, r[i]);
    return 0;
}
