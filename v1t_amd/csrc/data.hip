// Input pipeline on the device: the whole recording of a mouse lives in HBM as packed
// [trials][...] arrays (the Sensorium train tier is 7 x 4500 x 147 KB = 4.6 GB; the MI355X has 288 GB), and a batch is
// ONE gather + standardise pass per field instead of 4 .npy reads per trial on the host (data.py:138-153, 419-434).
#include <hip/hip_runtime.h>

#include "../../include/v1t_amd.h"

namespace {

// out[b][e] = ((src[index[b]][e] - sub[e % nsub]) / div[e % ndiv]) * mul[e % nmul]   (MiceDataset.transform_*,
// data.py:341-403: image (x - mean) / std, response x * precision, behaviour x / std, pupil centre (x - mean) / std),
// IEEE division as numpy does it. gray_c > 1: out has E / gray_c elements per trial, the mean over the gray_c channel
// planes of the transformed values (color2gray, data.py:338-339). One thread per output element, coalesced both ways.
template <typename T>
__global__ __launch_bounds__(256) void gather_transform_kernel(const T* src, const int* index, int B, long long E, const float* sub, long long nsub,
                                                               const float* dv, long long ndiv, const float* mul, long long nmul, int gray_c, float* out) {
    const long long EO = gray_c > 1 ? E / gray_c : E;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * EO) return;
    const int b = (int)(i / EO);
    const long long p = i % EO;
    const T* row = src + (long long)index[b] * E;
    auto one = [&](long long e) {
        float v = (float)row[e];
        if (sub) v = __fsub_rn(v, sub[e % nsub]);
        if (dv) v = __fdiv_rn(v, dv[e % ndiv]);
        if (mul) v = __fmul_rn(v, mul[e % nmul]);
        return v;
    };
    if (gray_c > 1) {
        float s = 0.f;
        for (int c = 0; c < gray_c; ++c) s = __fadd_rn(s, one((long long)c * EO + p));
        out[i] = __fdiv_rn(s, (float)gray_c);
    } else {
        out[i] = one(p);
    }
}

}  // namespace

extern "C" int v1t_gather_transform(const void* src, int src_u8, const int* index, int B, long long E, const float* sub, long long nsub, const float* div,
                                    long long ndiv, const float* mul, long long nmul, int gray_c, float* out, void* stream) {
    if (!src || !index || !out || B < 0 || E <= 0 || (sub && nsub <= 0) || (div && ndiv <= 0) || (mul && nmul <= 0)) return V1T_ERR_ARG;
    if (gray_c > 1 && E % gray_c) return V1T_ERR_ARG;
    if (B == 0) return V1T_OK;
    const long long n = (long long)B * (gray_c > 1 ? E / gray_c : E);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (src_u8)
        hipLaunchKernelGGL(gather_transform_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src, index, B, E, sub, nsub,
                           div, ndiv, mul, nmul, gray_c, out);
    else
        hipLaunchKernelGGL(gather_transform_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, index, B, E, sub, nsub, div, ndiv,
                           mul, nmul, gray_c, out);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
