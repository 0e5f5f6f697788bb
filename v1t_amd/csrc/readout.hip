// v1t_amd — Gaussian2d readout kernels (gfx950). Replaces, for one mouse,
//   F.grid_sample(z, grid, align_corners=True) -> * features -> sum(dim=1) -> + bias
// (reference gaussian2d.py:270-276) without materialising the (B,C,N) sampled tensor.
//
// Layout: the core's residual stream is read token-major as it lies in HBM ([b][cell][c], channel
// stride 1, so a bilinear tap is one contiguous C-float row); the feature weights are stored
// neuron-major [n][c] (row stride FS). One wave per neuron walks all images of the batch with
// lanes spanning channels: every load is a coalesced row, the feature row stays in registers, the
// channel sum is one wave reduction. HBM-bound; algorithmic bytes in DESIGN.md.
#include "readout.h"

namespace {

struct Taps {
    int cell[4];
    float w[4];     // bilinear weight, 0 when the tap is outside the map (zeros padding)
    float ax, ay;
    bool in[4];
};

DEVFN Taps make_taps(float gx, float gy, int W, int H) {
    Taps t;
    const float px = (gx + 1.f) * 0.5f * (float)(W - 1);
    const float py = (gy + 1.f) * 0.5f * (float)(H - 1);
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy;
    t.ax = px - fx;
    t.ay = py - fy;
    const float wx[2] = {1.f - t.ax, t.ax}, wy[2] = {1.f - t.ay, t.ay};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xi = x0 + (k & 1), yi = y0 + (k >> 1);
        t.in[k] = xi >= 0 && xi < W && yi >= 0 && yi < H;
        t.cell[k] = t.in[k] ? yi * W + xi : 0;
        t.w[k] = t.in[k] ? wx[k & 1] * wy[k >> 1] : 0.f;
    }
    return t;
}

template <int NE>
__global__ __launch_bounds__(256) void readout_fwd_kernel(ReadoutArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= a.N) return;
    float f[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        f[i] = (c < a.C) ? a.feat[(size_t)n * a.FS + c] : 0.f;
    }
    const float bias = a.bias ? a.bias[n] : 0.f;
    for (int b = 0; b < a.B; ++b) {
        const float gx = a.grid[((size_t)b * a.N + n) * 2 + 0];
        const float gy = a.grid[((size_t)b * a.N + n) * 2 + 1];
        const Taps t = make_taps(gx, gy, a.W, a.H);
        const float* zb = a.z + (size_t)b * a.zsb;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* zr = zb + (size_t)t.cell[k] * a.zsc;
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int c = lane + 64 * i;
                const float zv = (c < a.C) ? zr[c] : 0.f;
                acc += t.w[k] * f[i] * zv;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) a.out[(size_t)b * a.N + n] = acc + bias;
    }
}

// Backward: G = dL/du (B,N).  dbias[n] += sum_b G;  dfeat[n][c] += sum_b G * S[b][c]   (owned by the
// wave: plain +=);  dz[b][tap][c] += G * w_tap * F[n][c]  (fp32 atomics, 256-B contiguous runs);
// dgrid[b][n] = (dL/dgx, dL/dgy) through the bilinear weights.
template <int NE>
__global__ __launch_bounds__(256) void readout_bwd_kernel(ReadoutArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= a.N) return;
    float f[NE], df[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        f[i] = (c < a.C) ? a.feat[(size_t)n * a.FS + c] : 0.f;
        df[i] = 0.f;
    }
    float gsum = 0.f;
    for (int b = 0; b < a.B; ++b) {
        const float gx = a.grid[((size_t)b * a.N + n) * 2 + 0];
        const float gy = a.grid[((size_t)b * a.N + n) * 2 + 1];
        const float G = a.gout[(size_t)b * a.N + n];
        gsum += G;
        const Taps t = make_taps(gx, gy, a.W, a.H);
        const float* zb = a.z + (size_t)b * a.zsb;
        float* dzb = a.dz ? a.dz + (size_t)b * a.dzsb : nullptr;
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + 64 * i;
            float zv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) zv[k] = (c < a.C && t.in[k]) ? zb[(size_t)t.cell[k] * a.zsc + c] : 0.f;
            df[i] += G * (t.w[0] * zv[0] + t.w[1] * zv[1] + t.w[2] * zv[2] + t.w[3] * zv[3]);
            sx += f[i] * ((1.f - t.ay) * (zv[1] - zv[0]) + t.ay * (zv[3] - zv[2]));
            sy += f[i] * ((1.f - t.ax) * (zv[2] - zv[0]) + t.ax * (zv[3] - zv[1]));
            if (dzb && c < a.C) {
                const float gf = G * f[i];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (t.in[k]) atomicAdd(&dzb[(size_t)t.cell[k] * a.dzsc + c], gf * t.w[k]);
            }
        }
        if (a.dgrid) {
            sx = wave_sum(sx);
            sy = wave_sum(sy);
            if (lane == 0) {
                a.dgrid[((size_t)b * a.N + n) * 2 + 0] = G * sx * 0.5f * (float)(a.W - 1);
                a.dgrid[((size_t)b * a.N + n) * 2 + 1] = G * sy * 0.5f * (float)(a.H - 1);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        if (a.dfeat && c < a.C) a.dfeat[(size_t)n * a.FS + c] += df[i];
    }
    if (a.dbias && lane == 0) a.dbias[n] += gsum;
}

template <int NE>
int launch_t(const ReadoutArgs& a, bool bwd, hipStream_t s) {
    const dim3 grid((a.N + 3) / 4);
    if (bwd) hipLaunchKernelGGL(readout_bwd_kernel<NE>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(readout_fwd_kernel<NE>, grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

int dispatch(const ReadoutArgs& a, bool bwd, hipStream_t s) {
    if (a.C > 256 || a.N <= 0) return a.N <= 0 ? V1T_OK : V1T_ERR_UNSUPPORTED;
    switch ((a.C + 63) / 64) {
        case 1: return launch_t<1>(a, bwd, s);
        case 2: return launch_t<2>(a, bwd, s);
        case 3: return launch_t<3>(a, bwd, s);
        default: return launch_t<4>(a, bwd, s);
    }
}

}  // namespace

int launch_readout_fwd(const ReadoutArgs& a, hipStream_t s) { return dispatch(a, false, s); }
int launch_readout_bwd(const ReadoutArgs& a, hipStream_t s) { return dispatch(a, true, s); }
