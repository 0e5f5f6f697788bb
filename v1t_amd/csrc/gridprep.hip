// v1t_amd — readout grid / core shifter bookkeeping kernels (gfx950). See gridprep.h.
#include "gridprep.h"

namespace {

struct MuOut {
    float mu[2];
    float pre1[GRID_HID];
};

DEVFN void predict_mu(const GridArgs& a, int n, const float* sW0, const float* sb0, const float* sW2, const float* sb2, float (&mu)[2], float (&h)[GRID_HID]) {
    float x[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < a.gd; ++i) x[i] = a.src[(size_t)n * a.gd + i];
    float o0 = sb2[0], o1 = sb2[1];
#pragma unroll
    for (int j = 0; j < GRID_HID; ++j) {
        float p = sb0[j];
        for (int i = 0; i < a.gd; ++i) p += sW0[j * a.gd + i] * x[i];
        h[j] = p;  // pre-activation
        const float e = p > 0.f ? p : expm1f(p);
        o0 += sW2[j] * e;
        o1 += sW2[GRID_HID + j] * e;
    }
    mu[0] = tanhf(o0);
    mu[1] = tanhf(o1);
}

__global__ __launch_bounds__(256) void grid_fwd_kernel(GridArgs a) {
    __shared__ float sW0[GRID_HID * 3], sb0[GRID_HID], sW2[2 * GRID_HID], sb2[2];
    const int tid = threadIdx.x;
    if (a.gd > 0) {
        for (int i = tid; i < GRID_HID * a.gd; i += 256) sW0[i] = a.W0[i];
        for (int i = tid; i < GRID_HID; i += 256) sb0[i] = a.b0[i];
        for (int i = tid; i < 2 * GRID_HID; i += 256) sW2[i] = a.W2[i];
        if (tid < 2) sb2[tid] = a.b2[tid];
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    if (n >= a.N) return;
    float mu[2];
    if (a.gd > 0) {
        float h[GRID_HID];
        predict_mu(a, n, sW0, sb0, sW2, sb2, mu, h);
    } else {
        mu[0] = a.mu_free[2 * n];
        mu[1] = a.mu_free[2 * n + 1];
    }
    const float s00 = a.sigma[4 * n], s01 = a.sigma[4 * n + 1], s10 = a.sigma[4 * n + 2], s11 = a.sigma[4 * n + 3];
    for (int b = 0; b < a.B; ++b) {
        float g0 = mu[0], g1 = mu[1];
        if (a.eps) {
            const float e0 = a.eps[((size_t)b * a.N + n) * 2], e1 = a.eps[((size_t)b * a.N + n) * 2 + 1];
            g0 += s00 * e0 + s01 * e1;  // einsum "ancd,bnid->bnic": g_c = sum_d sigma[n][c][d] eps[d]
            g1 += s10 * e0 + s11 * e1;
        }
        g0 = fminf(fmaxf(g0, -1.f), 1.f);
        g1 = fminf(fmaxf(g1, -1.f), 1.f);
        if (a.shift) {
            g0 += a.shift[2 * b];
            g1 += a.shift[2 * b + 1];
        }
        a.grid[((size_t)b * a.N + n) * 2] = g0;
        a.grid[((size_t)b * a.N + n) * 2 + 1] = g1;
    }
}

__global__ __launch_bounds__(256) void grid_bwd_kernel(GridArgs a) {
    __shared__ float sW0[GRID_HID * 3], sb0[GRID_HID], sW2[2 * GRID_HID], sb2[2];
    __shared__ float sred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.gd > 0) {
        for (int i = tid; i < GRID_HID * a.gd; i += 256) sW0[i] = a.W0[i];
        for (int i = tid; i < GRID_HID; i += 256) sb0[i] = a.b0[i];
        for (int i = tid; i < 2 * GRID_HID; i += 256) sW2[i] = a.W2[i];
        if (tid < 2) sb2[tid] = a.b2[tid];
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    const bool ok = n < a.N;
    float mu[2] = {0.f, 0.f}, h[GRID_HID];
    if (ok) {
        if (a.gd > 0) predict_mu(a, n, sW0, sb0, sW2, sb2, mu, h);
        else { mu[0] = a.mu_free[2 * n]; mu[1] = a.mu_free[2 * n + 1]; }
    }
    float s00 = 0, s01 = 0, s10 = 0, s11 = 0;
    if (ok) { s00 = a.sigma[4 * n]; s01 = a.sigma[4 * n + 1]; s10 = a.sigma[4 * n + 2]; s11 = a.sigma[4 * n + 3]; }
    float dmu0 = 0.f, dmu1 = 0.f, ds00 = 0.f, ds01 = 0.f, ds10 = 0.f, ds11 = 0.f;
    // images in chunks of 8: all loads of a chunk are issued before the first wave reduction / atomic, which the
    // compiler may not move loads across (32 workgroups: this kernel runs at the latency of its own chain)
    for (int b0 = 0; b0 < a.B; b0 += 8) {
        float dd0[8], dd1[8], ee0[8], ee1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u;
            const bool v = ok && b < a.B;
            const size_t i2 = ((size_t)(v ? b : 0) * a.N + (v ? n : 0)) * 2;
            dd0[u] = v ? a.dgrid[i2] : 0.f;
            dd1[u] = v ? a.dgrid[i2 + 1] : 0.f;
            ee0[u] = (v && a.eps) ? a.eps[i2] : 0.f;
            ee1[u] = (v && a.eps) ? a.eps[i2 + 1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u;
            if (b >= a.B) break;
            const float d0 = dd0[u], d1 = dd1[u], e0 = ee0[u], e1 = ee1[u];
            if (a.dshift) {  // d shift[b] = sum_n d grid (the shift is added after the clamp)
                const float t0 = wave_sum(d0), t1 = wave_sum(d1);
                if (lane == 0) {
                    atomicAdd(&a.dshift[2 * b], t0);
                    atomicAdd(&a.dshift[2 * b + 1], t1);
                }
            }
            const float p0 = mu[0] + s00 * e0 + s01 * e1;
            const float p1 = mu[1] + s10 * e0 + s11 * e1;
            // torch.clamp passes the gradient on the closed interval [-1, 1]
            const float g0 = (p0 >= -1.f && p0 <= 1.f) ? d0 : 0.f;
            const float g1 = (p1 >= -1.f && p1 <= 1.f) ? d1 : 0.f;
            dmu0 += g0; dmu1 += g1;
            ds00 += g0 * e0; ds01 += g0 * e1; ds10 += g1 * e0; ds11 += g1 * e1;
        }
    }
    if (ok && a.dsigma) {
        a.dsigma[4 * n] = ds00; a.dsigma[4 * n + 1] = ds01; a.dsigma[4 * n + 2] = ds10; a.dsigma[4 * n + 3] = ds11;
    }
    if (a.gd == 0) {
        if (ok && a.dmu_free) { a.dmu_free[2 * n] = dmu0; a.dmu_free[2 * n + 1] = dmu1; }
        return;
    }
    // grid-predictor backward: per-neuron terms reduced over the wave, one atomic per accumulator per wave
    const float q0 = ok ? dmu0 * (1.f - mu[0] * mu[0]) : 0.f;  // d(pre-tanh)
    const float q1 = ok ? dmu1 * (1.f - mu[1] * mu[1]) : 0.f;
    float x[3] = {0.f, 0.f, 0.f};
    if (ok) for (int i = 0; i < a.gd; ++i) x[i] = a.src[(size_t)n * a.gd + i];
    // per-wave sums go to LDS; one atomic per accumulator per workgroup after the loop
    __shared__ float sacc[4][GRID_HID][6];
    __shared__ float sq[4][2];
    {
        const float t0 = wave_sum(q0), t1 = wave_sum(q1);
        if (lane == 0) { sq[wave][0] = t0; sq[wave][1] = t1; }
    }
#pragma unroll 2
    for (int j = 0; j < GRID_HID; ++j) {
        const float p = ok ? h[j] : 0.f;
        const float e = p > 0.f ? p : expm1f(p);
        const float de = p > 0.f ? 1.f : e + 1.f;  // ELU'
        const float dh = (q0 * sW2[j] + q1 * sW2[GRID_HID + j]) * de;
        const float w20 = wave_sum(q0 * e), w21 = wave_sum(q1 * e), bb = wave_sum(dh);
        float wx[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < a.gd; ++i) wx[i] = wave_sum(dh * x[i]);
        if (lane == 0) {
            sacc[wave][j][0] = w20; sacc[wave][j][1] = w21; sacc[wave][j][2] = bb;
            sacc[wave][j][3] = wx[0]; sacc[wave][j][4] = wx[1]; sacc[wave][j][5] = wx[2];
        }
    }
    __syncthreads();
    if (tid < 2) atomicAdd(&a.db2[tid], sq[0][tid] + sq[1][tid] + sq[2][tid] + sq[3][tid]);
    if (tid < GRID_HID * 6) {
        const int j = tid / 6, c = tid % 6;
        const float v = sacc[0][j][c] + sacc[1][j][c] + sacc[2][j][c] + sacc[3][j][c];
        if (c == 0) atomicAdd(&a.dW2[j], v);
        else if (c == 1) atomicAdd(&a.dW2[GRID_HID + j], v);
        else if (c == 2) atomicAdd(&a.db0[j], v);
        else if (c - 3 < a.gd) atomicAdd(&a.dW0[j * a.gd + c - 3], v);
    }
    (void)sred;
}

// shifter: one workgroup; thread b handles sample b (B <= 1024 -> loop)
__global__ __launch_bounds__(256) void shifter_fwd_kernel(ShifterArgs a) {
    for (int b = threadIdx.x; b < a.B; b += 256) {
        const float x0 = a.pupil[2 * b], x1 = a.pupil[2 * b + 1];
        float h1[5], h2[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) h1[j] = tanhf(a.W0[2 * j] * x0 + a.W0[2 * j + 1] * x1 + a.b0[j]);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float s = a.b2[j];
#pragma unroll
            for (int i = 0; i < 5; ++i) s += a.W2[5 * j + i] * h1[i];
            h2[j] = tanhf(s);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float s = a.b4[c];
#pragma unroll
            for (int i = 0; i < 5; ++i) s += a.W4[5 * c + i] * h2[i];
            a.shift[2 * b + c] = tanhf(s);
        }
    }
}

__global__ __launch_bounds__(256) void shifter_bwd_kernel(ShifterArgs a) {
    __shared__ float acc[64];  // dW0 10, db0 5, dW2 25, db2 5, dW4 10, db4 2 = 57
    if (threadIdx.x < 64) acc[threadIdx.x] = 0.f;
    __syncthreads();
    for (int b = threadIdx.x; b < a.B; b += 256) {
        const float x0 = a.pupil[2 * b], x1 = a.pupil[2 * b + 1];
        float h1[5], h2[5], o[2];
#pragma unroll
        for (int j = 0; j < 5; ++j) h1[j] = tanhf(a.W0[2 * j] * x0 + a.W0[2 * j + 1] * x1 + a.b0[j]);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float s = a.b2[j];
#pragma unroll
            for (int i = 0; i < 5; ++i) s += a.W2[5 * j + i] * h1[i];
            h2[j] = tanhf(s);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float s = a.b4[c];
#pragma unroll
            for (int i = 0; i < 5; ++i) s += a.W4[5 * c + i] * h2[i];
            o[c] = tanhf(s);
        }
        float d3[2], d2[5], d1[5];
#pragma unroll
        for (int c = 0; c < 2; ++c) d3[c] = a.dshift[2 * b + c] * (1.f - o[c] * o[c]);
#pragma unroll
        for (int i = 0; i < 5; ++i) d2[i] = (d3[0] * a.W4[i] + d3[1] * a.W4[5 + i]) * (1.f - h2[i] * h2[i]);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) s += d2[j] * a.W2[5 * j + i];
            d1[i] = s * (1.f - h1[i] * h1[i]);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            atomicAdd(&acc[2 * j], d1[j] * x0);
            atomicAdd(&acc[2 * j + 1], d1[j] * x1);
            atomicAdd(&acc[10 + j], d1[j]);
#pragma unroll
            for (int i = 0; i < 5; ++i) atomicAdd(&acc[15 + 5 * j + i], d2[j] * h1[i]);
            atomicAdd(&acc[40 + j], d2[j]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 5; ++i) atomicAdd(&acc[45 + 5 * c + i], d3[c] * h2[i]);
            atomicAdd(&acc[55 + c], d3[c]);
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 10) a.dW0[t] += acc[t];
    else if (t < 15) a.db0[t - 10] += acc[t];
    else if (t < 40) a.dW2[t - 15] += acc[t];
    else if (t < 45) a.db2[t - 40] += acc[t];
    else if (t < 55) a.dW4[t - 45] += acc[t];
    else if (t < 57) a.db4[t - 55] += acc[t];
}

inline int ok() { return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH; }

}  // namespace

int launch_grid_fwd(const GridArgs& a, hipStream_t s) {
    if (a.N <= 0) return V1T_OK;
    if (a.gd < 0 || a.gd > 3) return V1T_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(grid_fwd_kernel, dim3((a.N + 255) / 256), dim3(256), 0, s, a);
    return ok();
}
int launch_grid_bwd(const GridArgs& a, hipStream_t s) {
    if (a.N <= 0) return V1T_OK;
    if (a.gd < 0 || a.gd > 3) return V1T_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(grid_bwd_kernel, dim3((a.N + 255) / 256), dim3(256), 0, s, a);
    return ok();
}
int launch_shifter_fwd(const ShifterArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(shifter_fwd_kernel, dim3(1), dim3(256), 0, s, a);
    return ok();
}
int launch_shifter_bwd(const ShifterArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(shifter_bwd_kernel, dim3(1), dim3(256), 0, s, a);
    return ok();
}
