// v1t_amd — readout grid / core shifter bookkeeping kernels (gfx950). See gridprep.h.
#include "gridprep.h"

namespace {

// 64 neurons per workgroup (lane = neuron), four waves: wave w evaluates hidden units j = w, w+4, ... of the predictor,
// the partial pre-activations of mu meet in LDS, then wave w writes the images b = w, w+4, ... (a latency-bound kernel:
// the split quarters the per-neuron chain).
struct GridFwdLds {
    float sW0[GRID_HID * 3], sb0[GRID_HID], sW2[2 * GRID_HID], sb2[2];
    float so[4][2][64];
};
DEVFN void grid_fwd_body(const GridArgs& a, int bx, GridFwdLds& L) {
    auto& sW0 = L.sW0; auto& sb0 = L.sb0; auto& sW2 = L.sW2; auto& sb2 = L.sb2; auto& so = L.so;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.gd > 0) {
        for (int i = tid; i < GRID_HID * a.gd; i += 256) sW0[i] = a.W0[i];
        for (int i = tid; i < GRID_HID; i += 256) sb0[i] = a.b0[i];
        for (int i = tid; i < 2 * GRID_HID; i += 256) sW2[i] = a.W2[i];
        if (tid < 2) sb2[tid] = a.b2[tid];
    }
    const int n = bx * 64 + lane;
    const bool ok = n < a.N;
    float x[3] = {0.f, 0.f, 0.f};
    if (ok) for (int i = 0; i < a.gd; ++i) x[i] = a.src[(size_t)n * a.gd + i];
    float s00 = 0, s01 = 0, s10 = 0, s11 = 0;
    if (ok) { s00 = a.sigma[4 * n]; s01 = a.sigma[4 * n + 1]; s10 = a.sigma[4 * n + 2]; s11 = a.sigma[4 * n + 3]; }
    __syncthreads();
    float mu0, mu1;
    if (a.gd > 0) {
        float o0 = 0.f, o1 = 0.f;
        for (int j = wave; j < GRID_HID; j += 4) {
            float p = sb0[j];
            for (int i = 0; i < a.gd; ++i) p += sW0[j * a.gd + i] * x[i];
            const float e = p > 0.f ? p : expm1f(p);
            o0 += sW2[j] * e;
            o1 += sW2[GRID_HID + j] * e;
        }
        so[wave][0][lane] = o0;
        so[wave][1][lane] = o1;
        __syncthreads();
        mu0 = tanhf(sb2[0] + ((so[0][0][lane] + so[1][0][lane]) + (so[2][0][lane] + so[3][0][lane])));
        mu1 = tanhf(sb2[1] + ((so[0][1][lane] + so[1][1][lane]) + (so[2][1][lane] + so[3][1][lane])));
    } else {
        mu0 = ok ? a.mu_free[2 * n] : 0.f;
        mu1 = ok ? a.mu_free[2 * n + 1] : 0.f;
    }
    if (!ok) return;
    for (int b = wave; b < a.B; b += 4) {
        float g0 = mu0, g1 = mu1;
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
        *(float2*)(a.grid + ((size_t)b * a.N + n) * 2) = make_float2(g0, g1);
    }
}
__global__ __launch_bounds__(256) void grid_fwd_kernel(GridArgs a) {
    __shared__ GridFwdLds L;
    grid_fwd_body(a, blockIdx.x, L);
}

// Backward of the grid: 64 neurons per workgroup (lane = neuron), the serial work of a neuron split over the four waves
// so that the kernel's critical path is a quarter of the per-neuron chain (it is a latency-bound kernel: 125 workgroups
// for 8000 neurons): wave w owns hidden units j = w, w+4, ... of the predictor (forward recompute and backward) and
// images b = w, w+4, ...; the pieces meet in LDS twice (mu, then d mu / d sigma).
constexpr int GB_WAVES = 4;
constexpr int GB_SHIFT0 = 192;  // partial row: [0,2) db2, 2 + 6j + {dW2[0][j], dW2[1][j], db0[j], dW0[j][0..2]}, [192, 192 + 2B) d shift
struct GridBwdLds {
    float sW0[GRID_HID * 3], sb0[GRID_HID], sW2[2 * GRID_HID], sb2[2];
    float so[GB_WAVES][2][64];  // partial pre-tanh mu per wave
    float sd[GB_WAVES][6][64];  // partial d mu (2), d sigma (4) per wave
};
DEVFN void grid_bwd_body(const GridArgs& a, int bx, GridBwdLds& L) {
    float* prow = a.part ? a.part + (size_t)bx * a.part_stride : nullptr;
    auto& sW0 = L.sW0; auto& sb0 = L.sb0; auto& sW2 = L.sW2; auto& sb2 = L.sb2; auto& so = L.so; auto& sd = L.sd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.gd > 0) {
        for (int i = tid; i < GRID_HID * a.gd; i += 256) sW0[i] = a.W0[i];
        for (int i = tid; i < GRID_HID; i += 256) sb0[i] = a.b0[i];
        for (int i = tid; i < 2 * GRID_HID; i += 256) sW2[i] = a.W2[i];
        if (tid < 2) sb2[tid] = a.b2[tid];
    }
    const int n = bx * 64 + lane;
    const bool ok = n < a.N;
    // this wave's images: issue the loads first, they are consumed after the mu exchange
    constexpr int MAXI = 8;  // images per wave per pass
    float x[3] = {0.f, 0.f, 0.f};
    if (ok) for (int i = 0; i < a.gd; ++i) x[i] = a.src[(size_t)n * a.gd + i];
    float s00 = 0, s01 = 0, s10 = 0, s11 = 0;
    if (ok) { s00 = a.sigma[4 * n]; s01 = a.sigma[4 * n + 1]; s10 = a.sigma[4 * n + 2]; s11 = a.sigma[4 * n + 3]; }
    __syncthreads();
    // ---- predictor forward, hidden units j = wave, wave + 4, ...
    constexpr int JW = (GRID_HID + GB_WAVES - 1) / GB_WAVES;
    float hp[JW], he[JW];
    float mu0, mu1;
    if (a.gd > 0) {
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int k = 0; k < JW; ++k) {
            const int j = wave + GB_WAVES * k;
            float p = 0.f, e = 0.f;
            if (j < GRID_HID) {
                p = sb0[j];
                for (int i = 0; i < a.gd; ++i) p += sW0[j * a.gd + i] * x[i];
                e = p > 0.f ? p : expm1f(p);
                o0 += sW2[j] * e;
                o1 += sW2[GRID_HID + j] * e;
            }
            hp[k] = p;
            he[k] = e;
        }
        so[wave][0][lane] = o0;
        so[wave][1][lane] = o1;
        __syncthreads();
        // same summation order in every wave -> every wave holds the same mu
        mu0 = tanhf(sb2[0] + ((so[0][0][lane] + so[1][0][lane]) + (so[2][0][lane] + so[3][0][lane])));
        mu1 = tanhf(sb2[1] + ((so[0][1][lane] + so[1][1][lane]) + (so[2][1][lane] + so[3][1][lane])));
    } else {
        mu0 = ok ? a.mu_free[2 * n] : 0.f;
        mu1 = ok ? a.mu_free[2 * n + 1] : 0.f;
    }
    // ---- images b = wave, wave + 4, ...: d shift, d mu, d sigma partials
    float dmu0 = 0.f, dmu1 = 0.f, ds00 = 0.f, ds01 = 0.f, ds10 = 0.f, ds11 = 0.f;
    for (int b0 = wave; b0 < a.B; b0 += GB_WAVES * MAXI) {
        float dd0[MAXI], dd1[MAXI], ee0[MAXI], ee1[MAXI];
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int b = b0 + GB_WAVES * u;
            const bool v = ok && b < a.B;
            const size_t i2 = ((size_t)(v ? b : 0) * a.N + (v ? n : 0)) * 2;
            dd0[u] = v ? a.dgrid[i2] : 0.f;
            dd1[u] = v ? a.dgrid[i2 + 1] : 0.f;
            ee0[u] = (v && a.eps) ? a.eps[i2] : 0.f;
            ee1[u] = (v && a.eps) ? a.eps[i2 + 1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int b = b0 + GB_WAVES * u;
            if (b >= a.B) break;  // wave-uniform
            const float d0 = dd0[u], d1 = dd1[u], e0 = ee0[u], e1 = ee1[u];
            if (a.dshift) {  // d shift[b] = sum_n d grid (the shift is added after the clamp)
                const float t0 = wave_sum(d0), t1 = wave_sum(d1);
                if (lane == 0) {
                    if (prow) {
                        prow[GB_SHIFT0 + 2 * b] = t0;
                        prow[GB_SHIFT0 + 2 * b + 1] = t1;
                    } else {
                        atomicAdd(&a.dshift[2 * b], t0);
                        atomicAdd(&a.dshift[2 * b + 1], t1);
                    }
                }
            }
            const float p0 = mu0 + s00 * e0 + s01 * e1;
            const float p1 = mu1 + s10 * e0 + s11 * e1;
            // torch.clamp passes the gradient on the closed interval [-1, 1]
            const float g0 = (p0 >= -1.f && p0 <= 1.f) ? d0 : 0.f;
            const float g1 = (p1 >= -1.f && p1 <= 1.f) ? d1 : 0.f;
            dmu0 += g0; dmu1 += g1;
            ds00 += g0 * e0; ds01 += g0 * e1; ds10 += g1 * e0; ds11 += g1 * e1;
        }
    }
    sd[wave][0][lane] = dmu0; sd[wave][1][lane] = dmu1;
    sd[wave][2][lane] = ds00; sd[wave][3][lane] = ds01; sd[wave][4][lane] = ds10; sd[wave][5][lane] = ds11;
    __syncthreads();
    auto total = [&](int c) { return (sd[0][c][lane] + sd[1][c][lane]) + (sd[2][c][lane] + sd[3][c][lane]); };
    dmu0 = total(0);
    dmu1 = total(1);
    if (wave == 0 && ok && a.dsigma) {
        a.dsigma[4 * n] = total(2); a.dsigma[4 * n + 1] = total(3); a.dsigma[4 * n + 2] = total(4); a.dsigma[4 * n + 3] = total(5);
    }
    if (a.gd == 0) {
        if (wave == 0 && ok && a.dmu_free) { a.dmu_free[2 * n] = dmu0; a.dmu_free[2 * n + 1] = dmu1; }
        return;
    }
    // ---- predictor backward for this wave's hidden units: per-neuron terms reduced over the wave, one atomic per
    // accumulator per wave (125 workgroups x 4 waves -> ~16 adds per address per wave slot)
    const float q0 = ok ? dmu0 * (1.f - mu0 * mu0) : 0.f;  // d(pre-tanh)
    const float q1 = ok ? dmu1 * (1.f - mu1 * mu1) : 0.f;
    if (wave == 0) {
        const float t0 = wave_sum(q0), t1 = wave_sum(q1);
        if (lane == 0) {
            if (prow) { prow[0] = t0; prow[1] = t1; }
            else { atomicAdd(&a.db2[0], t0); atomicAdd(&a.db2[1], t1); }
        }
    }
#pragma unroll
    for (int k = 0; k < JW; ++k) {
        const int j = wave + GB_WAVES * k;
        if (j >= GRID_HID) break;  // wave-uniform
        const float p = ok ? hp[k] : 0.f, e = ok ? he[k] : 0.f;
        const float de = p > 0.f ? 1.f : e + 1.f;  // ELU'
        const float dh = (q0 * sW2[j] + q1 * sW2[GRID_HID + j]) * de;
        const float w20 = wave_sum(q0 * e), w21 = wave_sum(q1 * e), bb = wave_sum(dh);
        float wx[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < a.gd; ++i) wx[i] = wave_sum(dh * x[i]);
        if (lane == 0) {
            if (prow) {
                float* q = prow + 2 + 6 * j;
                q[0] = w20; q[1] = w21; q[2] = bb; q[3] = wx[0]; q[4] = wx[1]; q[5] = wx[2];
            } else {
                atomicAdd(&a.dW2[j], w20);
                atomicAdd(&a.dW2[GRID_HID + j], w21);
                atomicAdd(&a.db0[j], bb);
                for (int i = 0; i < a.gd; ++i) atomicAdd(&a.dW0[j * a.gd + i], wx[i]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void grid_bwd_kernel(GridArgs a) {
    __shared__ GridBwdLds L;
    grid_bwd_body(a, blockIdx.x, L);
}

// Second stage: column sums of the per-workgroup partial rows, added into the gradients (one writer per element).
DEVFN void grid_bwd_reduce_body(const GridArgs& a, int nrows, int bx, float (&sp)[4][64]) {
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = bx * 64 + lane;
    const int ncols = GB_SHIFT0 + (a.dshift ? 2 * a.B : 0);
    const bool live = c < ncols && !(c >= 2 + 6 * GRID_HID && c < GB_SHIFT0);
    float s0 = 0.f, s1 = 0.f;
    if (live) {
        int r = rg;
        for (; r + 4 < nrows; r += 8) {
            s0 += a.part[(size_t)r * a.part_stride + c];
            s1 += a.part[(size_t)(r + 4) * a.part_stride + c];
        }
        if (r < nrows) s0 += a.part[(size_t)r * a.part_stride + c];
    }
    sp[rg][lane] = s0 + s1;
    __syncthreads();
    if (rg != 0 || !live) return;
    const float v = (sp[0][lane] + sp[1][lane]) + (sp[2][lane] + sp[3][lane]);
    if (c >= GB_SHIFT0) { a.dshift[c - GB_SHIFT0] += v; return; }
    if (a.gd == 0) return;
    if (c < 2) { a.db2[c] += v; return; }
    const int j = (c - 2) / 6, k = (c - 2) % 6;
    if (k == 0) a.dW2[j] += v;
    else if (k == 1) a.dW2[GRID_HID + j] += v;
    else if (k == 2) a.db0[j] += v;
    else if (k - 3 < a.gd) a.dW0[j * a.gd + k - 3] += v;
}
__global__ __launch_bounds__(256) void grid_bwd_reduce_kernel(GridArgs a, int nrows) {
    __shared__ float sp[4][64];
    grid_bwd_reduce_body(a, nrows, blockIdx.x, sp);
}

// shifter: one workgroup; thread b handles sample b (B <= 1024 -> loop)
DEVFN void shifter_fwd_body(const ShifterArgs& a) {
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
__global__ __launch_bounds__(256) void shifter_fwd_kernel(ShifterArgs a) { shifter_fwd_body(a); }

// Per-thread partial sums of the 57 parameter gradients (dW0 10, db0 5, dW2 25, db2 5, dW4 10, db4 2) over the thread's
// samples, one DPP wave reduction per gradient, one LDS add per wave: no same-address atomic chains (the previous version
// issued 57 LDS atomics per sample, all lanes on the same words, and took 33 us for 16 samples).
DEVFN void shifter_bwd_body(const ShifterArgs& a, float (&acc)[4][64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[57];
#pragma unroll
    for (int k = 0; k < 57; ++k) g[k] = 0.f;
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
            g[2 * j] += d1[j] * x0;
            g[2 * j + 1] += d1[j] * x1;
            g[10 + j] += d1[j];
#pragma unroll
            for (int i = 0; i < 5; ++i) g[15 + 5 * j + i] += d2[j] * h1[i];
            g[40 + j] += d2[j];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 5; ++i) g[45 + 5 * c + i] += d3[c] * h2[i];
            g[55 + c] += d3[c];
        }
    }
    const bool active = wave * 64 < a.B;  // wave-uniform: waves without samples skip the reductions
    if (active) {
#pragma unroll
        for (int k = 0; k < 57; ++k) {
            const float v = wave_sum(g[k]);
            if (lane == 0) acc[wave][k] = v;
        }
    } else if (lane < 57) {
        acc[wave][lane] = 0.f;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= 57) return;
    const float v = acc[0][t] + acc[1][t] + acc[2][t] + acc[3][t];
    if (t < 10) a.dW0[t] += v;
    else if (t < 15) a.db0[t - 10] += v;
    else if (t < 40) a.dW2[t - 15] += v;
    else if (t < 45) a.db2[t - 40] += v;
    else if (t < 55) a.dW4[t - 45] += v;
    else a.db4[t - 55] += v;
}
__global__ __launch_bounds__(256) void shifter_bwd_kernel(ShifterArgs a) {
    __shared__ float acc[4][64];
    shifter_bwd_body(a, acc);
}

// ---- several units (mice) per launch (v1t_tails_*, see readout.hip): unit tables by value in the kernel arguments
DEVFN int multi_unit(const int* start, int n, int bid, int& local) {
    int u = 0;
    while (u + 1 < n && bid >= start[u + 1]) ++u;
    local = bid - start[u];
    return u;
}
struct GridMulti {
    GridArgs a[GP_MAX_UNITS];
    int start[GP_MAX_UNITS + 1];
    int n;
};
struct ShifterMulti {
    ShifterArgs a[GP_MAX_UNITS];
    int n;
};
__global__ __launch_bounds__(256) void grid_fwd_multi_kernel(GridMulti m) {
    __shared__ GridFwdLds L;
    int bx;
    const int u = multi_unit(m.start, m.n, blockIdx.x, bx);
    grid_fwd_body(m.a[u], bx, L);
}
__global__ __launch_bounds__(256) void grid_bwd_multi_kernel(GridMulti m) {
    __shared__ GridBwdLds L;
    int bx;
    const int u = multi_unit(m.start, m.n, blockIdx.x, bx);
    grid_bwd_body(m.a[u], bx, L);
}
__global__ __launch_bounds__(256) void grid_bwd_reduce_multi_kernel(GridMulti m) {
    __shared__ float sp[4][64];
    int bx;
    const int u = multi_unit(m.start, m.n, blockIdx.x, bx);
    grid_bwd_reduce_body(m.a[u], (m.a[u].N + 63) / 64, bx, sp);
}
__global__ __launch_bounds__(256) void shifter_fwd_multi_kernel(ShifterMulti m) { shifter_fwd_body(m.a[blockIdx.x]); }
__global__ __launch_bounds__(256) void shifter_bwd_multi_kernel(ShifterMulti m) {
    __shared__ float acc[4][64];
    shifter_bwd_body(m.a[blockIdx.x], acc);
}

inline int ok() { return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH; }

}  // namespace

int launch_grid_fwd(const GridArgs& a, hipStream_t s) {
    if (a.N <= 0) return V1T_OK;
    if (a.gd < 0 || a.gd > 3) return V1T_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(grid_fwd_kernel, dim3((a.N + 63) / 64), dim3(256), 0, s, a);
    return ok();
}
static int gb_stride(int B) { return (GB_SHIFT0 + 2 * B + 31) / 32 * 32; }
size_t grid_bwd_ws_bytes(int B, int N) { return sizeof(float) * (size_t)((N + 63) / 64) * gb_stride(B); }
int launch_grid_bwd(const GridArgs& a0, void* ws, size_t ws_bytes, hipStream_t s) {
    if (a0.N <= 0) return V1T_OK;
    if (a0.gd < 0 || a0.gd > 3) return V1T_ERR_UNSUPPORTED;
    GridArgs a = a0;
    const int nwg = (a.N + 63) / 64;
    a.part = nullptr;
    a.part_stride = 0;
    const bool two_stage = ws && (a.gd > 0 || a.dshift);
    if (two_stage) {
        if (ws_bytes < grid_bwd_ws_bytes(a.B, a.N)) return V1T_ERR_ARG;
        a.part = (float*)ws;
        a.part_stride = gb_stride(a.B);
    }
    hipLaunchKernelGGL(grid_bwd_kernel, dim3(nwg), dim3(256), 0, s, a);
    if (two_stage) {
        const int ncols = GB_SHIFT0 + (a.dshift ? 2 * a.B : 0);
        hipLaunchKernelGGL(grid_bwd_reduce_kernel, dim3((ncols + 63) / 64), dim3(256), 0, s, a, nwg);
    }
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

// ---- standard normal deviates (the readout's position noise eps ~ N(0, I), gaussian2d.py:219-221) from Philox-4x32-10
// (Salmon et al. 2011, the generator family torch's normal_() draws from): key = the full 64-bit seed, counter = (quad index lo,
// quad index hi, stream id, 0), so every (seed, stream) pair is its own 2^64-long stream of independent 128-bit blocks - no
// 32-bit key collapse, no overlapping windows between mice / steps / ranks. One block = 4 uniform words = 2 Box-Muller pairs
// (u1, u2 from DIFFERENT words) = out[4 i .. 4 i + 3]. Stateless: the draw of a step is a function of (seed, stream id) and can
// be replayed.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__global__ __launch_bounds__(256) void normal_fill_kernel(float* out, long long n, uint32_t k0, uint32_t k1, uint32_t stream_id) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // quad index
    if (4 * i >= n) return;
    uint32_t c[4] = {(uint32_t)i, (uint32_t)((unsigned long long)i >> 32), stream_id, 0u};
    philox4x32_10(c, k0, k1);
    float z[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float u1 = ((float)(c[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
        const float u2 = (float)(c[2 * p + 1] >> 8) * (1.0f / 16777216.0f);        // [0, 1)
        const float r = sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        z[2 * p] = r * cs;
        z[2 * p + 1] = r * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * i + e < n) out[4 * i + e] = z[e];
}
struct NormalMulti {
    float* out[GP_MAX_UNITS];
    long long n[GP_MAX_UNITS];
    uint32_t stream_id[GP_MAX_UNITS];
    int start[GP_MAX_UNITS + 1];
    int units;
};
__global__ __launch_bounds__(256) void normal_fill_multi_kernel(NormalMulti m, uint32_t k0, uint32_t k1) {
    int u = 0;
    while (u + 1 < m.units && (int)blockIdx.x >= m.start[u + 1]) ++u;
    const long long n = m.n[u];
    float* out = m.out[u];
    const long long i = (long long)(blockIdx.x - m.start[u]) * 256 + threadIdx.x;  // quad index inside the unit: the same draws as a launch of its own
    if (4 * i >= n) return;
    uint32_t c[4] = {(uint32_t)i, (uint32_t)((unsigned long long)i >> 32), m.stream_id[u], 0u};
    philox4x32_10(c, k0, k1);
    float z[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float u1 = ((float)(c[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);
        const float u2 = (float)(c[2 * p + 1] >> 8) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        z[2 * p] = r * cs;
        z[2 * p + 1] = r * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * i + e < n) out[4 * i + e] = z[e];
}
int launch_normal_fill_multi(float* const* out, const long long* n, const uint32_t* stream_id, int units, uint64_t seed, hipStream_t s) {
    if (units <= 0) return V1T_OK;
    if (units > GP_MAX_UNITS) return V1T_ERR_ARG;
    NormalMulti m{};
    int tot = 0;
    for (int u = 0; u < units; ++u) {
        m.out[u] = out[u]; m.n[u] = n[u]; m.stream_id[u] = stream_id[u];
        m.start[u] = tot;
        tot += (int)(((n[u] + 3) / 4 + 255) / 256);
    }
    m.start[units] = tot;
    m.units = units;
    if (tot == 0) return V1T_OK;
    hipLaunchKernelGGL(normal_fill_multi_kernel, dim3(tot), dim3(256), 0, s, m, (uint32_t)seed, (uint32_t)(seed >> 32));
    return ok();
}
int launch_normal_fill(float* out, long long n, uint64_t seed, uint32_t stream_id, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    const long long quads = (n + 3) / 4;
    hipLaunchKernelGGL(normal_fill_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, out, n, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id);
    return ok();
}
// out[r][0 .. na) = a[r][:], out[r][na .. na + nb) = b[r][:]  (the BehaviorMLP input cat(behaviors, pupil_centers), vit.py:431-432,
// written straight into a batch buffer shared by several mice)
__global__ __launch_bounds__(256) void concat2_kernel(const float* a, int na, const float* b, int nb, int rows, float* out, int ldo) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int w = na + nb;
    if (i >= rows * w) return;
    const int r = i / w, c = i % w;
    out[(size_t)r * ldo + c] = c < na ? a[(size_t)r * na + c] : b[(size_t)r * nb + (c - na)];
}
int launch_concat2(const float* a, int na, const float* b, int nb, int rows, float* out, int ldo, hipStream_t s) {
    const int n = rows * (na + nb);
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(concat2_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, na, b, nb, rows, out, ldo);
    return ok();
}

// ---- multi-unit launchers (one launch per stage over the mice of a training step)
int launch_grid_fwd_multi(const GridArgs* a, int n, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    if (n > GP_MAX_UNITS) return V1T_ERR_ARG;
    GridMulti m{};
    int tot = 0;
    for (int u = 0; u < n; ++u) {
        if (a[u].gd < 0 || a[u].gd > 3) return V1T_ERR_UNSUPPORTED;
        m.a[u] = a[u];
        m.start[u] = tot;
        tot += (a[u].N + 63) / 64;
    }
    m.start[n] = tot;
    m.n = n;
    if (tot == 0) return V1T_OK;
    hipLaunchKernelGGL(grid_fwd_multi_kernel, dim3(tot), dim3(256), 0, s, m);
    return ok();
}
int launch_grid_bwd_multi(const GridArgs* a0, void* const* ws, const size_t* ws_bytes, int n, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    if (n > GP_MAX_UNITS) return V1T_ERR_ARG;
    GridMulti m{}, r{};
    int tot = 0, rtot = 0;
    for (int u = 0; u < n; ++u) {
        GridArgs a = a0[u];
        if (a.gd < 0 || a.gd > 3) return V1T_ERR_UNSUPPORTED;
        if (!ws[u] || ws_bytes[u] < grid_bwd_ws_bytes(a.B, a.N)) return V1T_ERR_WORKSPACE;  // the multi form is the two-stage form
        a.part = (float*)ws[u];
        a.part_stride = gb_stride(a.B);
        m.a[u] = a;
        m.start[u] = tot;
        tot += (a.N + 63) / 64;
        r.a[u] = a;
        r.start[u] = rtot;
        const int ncols = GB_SHIFT0 + (a.dshift ? 2 * a.B : 0);
        rtot += (a.gd > 0 || a.dshift) ? (ncols + 63) / 64 : 0;
    }
    m.start[n] = tot; m.n = n;
    r.start[n] = rtot; r.n = n;
    if (tot) hipLaunchKernelGGL(grid_bwd_multi_kernel, dim3(tot), dim3(256), 0, s, m);
    if (rtot) hipLaunchKernelGGL(grid_bwd_reduce_multi_kernel, dim3(rtot), dim3(256), 0, s, r);
    return ok();
}
int launch_shifter_multi(const ShifterArgs* a, int n, bool bwd, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    if (n > GP_MAX_UNITS) return V1T_ERR_ARG;
    ShifterMulti m{};
    for (int u = 0; u < n; ++u) m.a[u] = a[u];
    m.n = n;
    if (bwd) hipLaunchKernelGGL(shifter_bwd_multi_kernel, dim3(n), dim3(256), 0, s, m);
    else hipLaunchKernelGGL(shifter_fwd_multi_kernel, dim3(n), dim3(256), 0, s, m);
    return ok();
}
