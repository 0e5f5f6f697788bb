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
DEVFN void readout_fwd_body(const ReadoutArgs& a, int bx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = bx * 4 + wave;
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

template <int NE>
__global__ __launch_bounds__(256) void readout_fwd_kernel(ReadoutArgs a) { readout_fwd_body<NE>(a, blockIdx.x); }

// Backward: G = dL/du (B,N).  dbias[n] += sum_b G;  dfeat[n][c] += sum_b G * S[b][c]   (owned by the
// wave: plain +=);  dz[b][tap][c] += G * w_tap * F[n][c]  (fp32 atomics, 256-B contiguous runs);
// dgrid[b][n] = (dL/dgx, dL/dgy) through the bilinear weights.
template <int NE>
DEVFN void readout_bwd_body(const ReadoutArgs& a, int bx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = bx * 4 + wave;
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
            if (dzb && c < a.C) {  // no-scratch form only (launch_readout_bwd without ws)
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
__global__ __launch_bounds__(256) void readout_bwd_kernel(ReadoutArgs a) { readout_bwd_body<NE>(a, blockIdx.x); }

// dz without a flood of global float atomics (4 x C x N x B adds = 317 MB at N = 8000, B = 16: ~250 us at the chip-wide
// 1.3 TB/s atomic rate; LDS float atomics are slower still, ~1 lane per clock per CU): invert the sampling.
//   sort  : workgroup (slice s of the neurons, image b) counting-sorts its taps by cell in LDS (histogram, scan,
//           scatter: integer LDS atomics only) into its region of the scratch: (cell, tap id, bilinear weight),
//           taps outside the map and the padding of the region carry the sentinel cell `cells` and sort last
//   gather: one wave per 64 consecutive sorted taps sums G[b][n] * w * F[n][:] (coalesced feature rows) while the
//           cell stays the same and adds the finished row to dz with ONE atomic row (620 B) per (chunk, cell) run:
//           ~20 MB of atomics for any distribution of the neurons over the map - including the freshly initialised
//           model, whose 8000 neurons all sit on a handful of cells, which per-cell lists could not balance.
constexpr int DZ_SLICES = 4;       // neuron slices per image (parallelism of the sort pass)
constexpr int DZ_MAX_CELLS = 4096; // LDS histogram capacity; larger maps use the atomic form
struct DzSort {
    unsigned* cell;  // [B * DZ_SLICES][R]
    unsigned* id;    // tap id = 4 * (b * N + n) + k
    float* wt;
    int ns, R;       // neurons per slice, region stride (multiple of 64)
};

DEVFN void readout_sort_body(const ReadoutArgs& a, const DzSort& ix, int s, int b, int* hist, int* wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cells = a.H * a.W;
    const int n0 = s * ix.ns, n1 = min(a.N, n0 + ix.ns);
    for (int i = tid; i <= cells; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int n = n0 + tid; n < n1; n += 1024) {
        const size_t i = (size_t)b * a.N + n;
        const Taps t = make_taps(a.grid[2 * i], a.grid[2 * i + 1], a.W, a.H);
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(&hist[t.in[q] ? t.cell[q] : cells], 1);
    }
    __syncthreads();
    // exclusive scan of hist[0..cells]: each thread owns a contiguous run of PER entries
    const int PER = (cells + 1 + 1023) / 1024;
    int run[(DZ_MAX_CELLS + 1 + 1023) / 1024], tot = 0;
#pragma unroll
    for (int e = 0; e < (DZ_MAX_CELLS + 1 + 1023) / 1024; ++e) {
        const int idx = tid * PER + e;
        run[e] = (e < PER && idx <= cells) ? hist[idx] : 0;
        tot += run[e];
    }
    int inc = tot;  // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = inc - tot;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < (DZ_MAX_CELLS + 1 + 1023) / 1024; ++e) {
        const int idx = tid * PER + e;
        if (e < PER && idx <= cells) {
            hist[idx] = base;
            base += run[e];
        }
    }
    __syncthreads();
    const size_t reg = ((size_t)b * DZ_SLICES + s) * ix.R;
    for (int n = n0 + tid; n < n1; n += 1024) {
        const size_t i = (size_t)b * a.N + n;
        const Taps t = make_taps(a.grid[2 * i], a.grid[2 * i + 1], a.W, a.H);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = t.in[q] ? t.cell[q] : cells;
            const int slot = atomicAdd(&hist[c], 1);
            ix.cell[reg + slot] = (unsigned)c;
            ix.id[reg + slot] = 4u * (unsigned)i + q;
            ix.wt[reg + slot] = t.w[q];
        }
    }
    for (int p = 4 * (n1 - n0) + tid; p < ix.R; p += 1024) ix.cell[reg + p] = (unsigned)cells;  // padding
}
__global__ __launch_bounds__(1024) void readout_sort_kernel(ReadoutArgs a, DzSort ix) {
    __shared__ int hist[DZ_MAX_CELLS + 1];
    __shared__ int wsum[16];
    readout_sort_body(a, ix, blockIdx.x, blockIdx.y, hist, wsum);
}

template <int NE>
DEVFN void readout_dz_gather_body(const ReadoutArgs& a, const DzSort& ix, int bx) {
    const int lane = threadIdx.x & 63;
    const size_t chunk = (size_t)bx * 4 + (threadIdx.x >> 6);
    const size_t total = (size_t)a.B * DZ_SLICES * ix.R;
    if (chunk * 64 >= total) return;
    const unsigned cells = a.H * a.W;
    const unsigned rowb = (unsigned)(chunk * 64 / ((size_t)DZ_SLICES * ix.R)) * cells;  // image of this chunk's region
    const unsigned rows = rowb + cells;
    // lane j holds tap j of the chunk; key = global row (image * cells + cell), >= rows for sentinels
    const size_t me = chunk * 64 + lane;
    const unsigned key = rowb + ix.cell[me];
    const unsigned id = key < rows ? ix.id[me] : 0u;
    const float gw = key < rows ? ix.wt[me] * a.gout[id >> 2] : 0.f;
    const unsigned nn = (id >> 2) % (unsigned)a.N;
    float acc[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) acc[j] = 0.f;
    unsigned cur = __builtin_amdgcn_readfirstlane(key);
    auto flush = [&](unsigned row) {
        float* dzr = a.dz + (size_t)(row / cells) * a.dzsb + (size_t)(row % cells) * a.dzsc;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int c = lane + 64 * j;
            if (c < a.C) atomicAdd(&dzr[c], acc[j]);
            acc[j] = 0.f;
        }
    };
    constexpr int RIF = 8;  // feature rows in flight (4 -> 8 rows: 189 -> 178 us per step, round 4)
    for (int t0 = 0; t0 < 64; t0 += RIF) {
        unsigned kq[RIF], nq[RIF], gq[RIF];
        float f[RIF][NE];
#pragma unroll
        for (int u = 0; u < RIF; ++u) {
            kq[u] = __builtin_amdgcn_readlane(key, t0 + u);
            nq[u] = __builtin_amdgcn_readlane(nn, t0 + u);
            gq[u] = __builtin_amdgcn_readlane(__float_as_uint(gw), t0 + u);
        }
        if (kq[0] >= rows) break;  // sentinels and padding sort to the end of the region
#pragma unroll
        for (int u = 0; u < RIF; ++u)
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int c = lane + 64 * j;
                f[u][j] = (c < a.C) ? a.feat[(size_t)nq[u] * a.FS + c] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < RIF; ++u) {
            if (kq[u] >= rows) break;
            if (kq[u] != cur) {
                flush(cur);
                cur = kq[u];
            }
            const float g = __uint_as_float(gq[u]);
#pragma unroll
            for (int j = 0; j < NE; ++j) acc[j] += g * f[u][j];
        }
    }
    if (cur < rows) flush(cur);
}
template <int NE>
__global__ __launch_bounds__(256) void readout_dz_gather_kernel(ReadoutArgs a, DzSort ix) { readout_dz_gather_body<NE>(a, ix, blockIdx.x); }

// ---- the same four kernels over SEVERAL units (mice) in one launch each: a training step runs the tails of all local mice as one launch
// per stage instead of one chain of small launches per mouse (v1t_tails_*). The unit table travels by value in the kernel arguments;
// a workgroup finds its unit from the prefix sums of the units' workgroup counts (wave-uniform scalar loads).
struct ReadoutMulti {
    ReadoutArgs a[TAILS_MAX_UNITS];
    DzSort ix[TAILS_MAX_UNITS];
    int start[TAILS_MAX_UNITS + 1];
    int n;
};
DEVFN int multi_unit(const int* start, int n, int bid, int& local) {
    int u = 0;
    while (u + 1 < n && bid >= start[u + 1]) ++u;
    local = bid - start[u];
    return u;
}
template <int NE>
__global__ __launch_bounds__(256) void readout_fwd_multi_kernel(ReadoutMulti m) {
    int bx;
    const int u = multi_unit(m.start, m.n, blockIdx.x, bx);
    readout_fwd_body<NE>(m.a[u], bx);
}
template <int NE>
__global__ __launch_bounds__(256) void readout_bwd_multi_kernel(ReadoutMulti m) {
    int bx;
    const int u = multi_unit(m.start, m.n, blockIdx.x, bx);
    readout_bwd_body<NE>(m.a[u], bx);
}
__global__ __launch_bounds__(1024) void readout_sort_multi_kernel(ReadoutMulti m) {
    __shared__ int hist[DZ_MAX_CELLS + 1];
    __shared__ int wsum[16];
    int bx;
    const int u = multi_unit(m.start, m.n, blockIdx.x, bx);
    readout_sort_body(m.a[u], m.ix[u], bx % DZ_SLICES, bx / DZ_SLICES, hist, wsum);
}
template <int NE>
__global__ __launch_bounds__(256) void readout_dz_gather_multi_kernel(ReadoutMulti m) {
    int bx;
    const int u = multi_unit(m.start, m.n, blockIdx.x, bx);
    readout_dz_gather_body<NE>(m.a[u], m.ix[u], bx);
}

DzSort sort_plan(void* ws, int B, int N) {
    DzSort d;
    d.ns = (N + DZ_SLICES - 1) / DZ_SLICES;
    d.R = (4 * d.ns + 63) / 64 * 64;
    const size_t arr = (size_t)B * DZ_SLICES * d.R * 4;
    char* p = (char*)ws;
    d.cell = (unsigned*)p; d.id = (unsigned*)(p + arr); d.wt = (float*)(p + 2 * arr);
    return d;
}

template <int NE>
int launch_t(const ReadoutArgs& a, bool bwd, void* ws, size_t ws_bytes, hipStream_t s, int parts = READOUT_BWD_ALL) {
    const dim3 grid((a.N + 3) / 4);
    if (!bwd) {
        hipLaunchKernelGGL(readout_fwd_kernel<NE>, grid, dim3(256), 0, s, a);
    } else if (ws && a.H * a.W <= DZ_MAX_CELLS && ws_bytes >= readout_bwd_ws_bytes(a.B, a.H, a.W, a.N) && (a.dz || parts != READOUT_BWD_ALL)) {
        // the three kernels of the sorted form, individually selectable (the training step sorts the taps while the core is still
        // in its forward - the sort needs the sample positions only - and gathers dz before the parameter gradients, so that the
        // core's backward can start as early as possible)
        const DzSort ix = sort_plan(ws, a.B, a.N);
        if (parts & READOUT_BWD_SORT) hipLaunchKernelGGL(readout_sort_kernel, dim3(DZ_SLICES, a.B), dim3(1024), 0, s, a, ix);
        if (parts & READOUT_BWD_PARAMS) {
            ReadoutArgs a2 = a;
            a2.dz = nullptr;
            hipLaunchKernelGGL(readout_bwd_kernel<NE>, grid, dim3(256), 0, s, a2);
        }
        if ((parts & READOUT_BWD_DZ) && a.dz) {
            const size_t chunks = (size_t)a.B * DZ_SLICES * ix.R / 64;
            hipLaunchKernelGGL(readout_dz_gather_kernel<NE>, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, s, a, ix);
        }
    } else if (parts != READOUT_BWD_ALL) {
        return V1T_ERR_WORKSPACE;  // the split form exists with the sorted workspace only
    } else {
        hipLaunchKernelGGL(readout_bwd_kernel<NE>, grid, dim3(256), 0, s, a);
    }
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

int dispatch(const ReadoutArgs& a, bool bwd, void* ws, size_t ws_bytes, hipStream_t s, int parts = READOUT_BWD_ALL) {
    if (a.C > 256 || a.N <= 0) return a.N <= 0 ? V1T_OK : V1T_ERR_UNSUPPORTED;
    switch ((a.C + 63) / 64) {
        case 1: return launch_t<1>(a, bwd, ws, ws_bytes, s, parts);
        case 2: return launch_t<2>(a, bwd, ws, ws_bytes, s, parts);
        case 3: return launch_t<3>(a, bwd, ws, ws_bytes, s, parts);
        default: return launch_t<4>(a, bwd, ws, ws_bytes, s, parts);
    }
}

template <int NE>
int launch_multi_t(ReadoutMulti& m, int stage, hipStream_t s) {
    int tot = 0;
    for (int u = 0; u < m.n; ++u) {
        const ReadoutArgs& a = m.a[u];
        m.start[u] = tot;
        if (stage == TAILS_SORT) tot += DZ_SLICES * a.B;
        else if (stage == TAILS_DZ) tot += (int)(((size_t)a.B * DZ_SLICES * m.ix[u].R / 64 + 3) / 4);
        else tot += (a.N + 3) / 4;
    }
    m.start[m.n] = tot;
    if (tot == 0) return V1T_OK;
    switch (stage) {
        case TAILS_FWD: hipLaunchKernelGGL(readout_fwd_multi_kernel<NE>, dim3(tot), dim3(256), 0, s, m); break;
        case TAILS_PARAMS: hipLaunchKernelGGL(readout_bwd_multi_kernel<NE>, dim3(tot), dim3(256), 0, s, m); break;
        case TAILS_SORT: hipLaunchKernelGGL(readout_sort_multi_kernel, dim3(tot), dim3(1024), 0, s, m); break;
        case TAILS_DZ: hipLaunchKernelGGL(readout_dz_gather_multi_kernel<NE>, dim3(tot), dim3(256), 0, s, m); break;
        default: return V1T_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

}  // namespace

int launch_readout_multi(const ReadoutArgs* a, void* const* ws, const size_t* ws_bytes, int n, int stage, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    if (n > TAILS_MAX_UNITS) return V1T_ERR_ARG;
    ReadoutMulti m{};
    m.n = n;
    for (int u = 0; u < n; ++u) {
        m.a[u] = a[u];
        if (a[u].C != a[0].C || a[u].C > 256 || a[u].N <= 0) return V1T_ERR_UNSUPPORTED;
        if (stage == TAILS_PARAMS) m.a[u].dz = nullptr;
        if (stage == TAILS_SORT || stage == TAILS_DZ) {
            if (!ws[u] || a[u].H * a[u].W > DZ_MAX_CELLS || ws_bytes[u] < readout_bwd_ws_bytes(a[u].B, a[u].H, a[u].W, a[u].N)) return V1T_ERR_WORKSPACE;
            m.ix[u] = sort_plan(ws[u], a[u].B, a[u].N);
        }
    }
    switch ((a[0].C + 63) / 64) {
        case 1: return launch_multi_t<1>(m, stage, s);
        case 2: return launch_multi_t<2>(m, stage, s);
        case 3: return launch_multi_t<3>(m, stage, s);
        default: return launch_multi_t<4>(m, stage, s);
    }
}

size_t readout_bwd_ws_bytes(int B, int H, int W, int N) {
    (void)H; (void)W;
    const DzSort d = sort_plan(nullptr, B, N);
    return 3 * (size_t)B * DZ_SLICES * d.R * 4 + 256;
}
int launch_readout_fwd(const ReadoutArgs& a, hipStream_t s) { return dispatch(a, false, nullptr, 0, s); }
int launch_readout_bwd(const ReadoutArgs& a, void* ws, size_t ws_bytes, hipStream_t s) { return dispatch(a, true, ws, ws_bytes, s); }
int launch_readout_bwd_parts(const ReadoutArgs& a, void* ws, size_t ws_bytes, int parts, hipStream_t s) { return dispatch(a, true, ws, ws_bytes, s, parts); }
