// v1t_amd — bf16 MFMA GEMM kernels (gfx950). See gemm.h.
//
// gemm_nt tiling: workgroup = 4 waves = 128 rows x (32*NBLK) cols; wave w owns rows 32w..32w+31 and
// all NBLK 32-col MFMA blocks (A fragment read once, reused NBLK times). K-tile 32, LDS rows padded
// by 16 B (80-B stride -> the 16 rows a ds_read_b128 lane group touches land on 16 distinct
// 16-B slots), double-buffered LDS, next tile's global loads issued before the MFMAs and written
// to the other buffer after them (one barrier per K-tile).
#include <cstdlib>

#include "gemm.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LS = BK + 8;  // LDS row stride in elements (80 B)

DEVFN size_t frag_index(const GemmNTArgs& g, int m0, int n0, int nb, int wave, int lane) {
    return ((((size_t)(m0 / BM) * (g.N / 32) + (n0 / 32 + nb)) * 4 + wave) * 64 + lane) * 16;
}

template <int NBLK, int EPI>
DEVFN void gemm_epilogue(const GemmNTArgs& g, f32x16 (&acc)[NBLK], const f32x16 (&resv)[NBLK], int m0, int n0, int wave, int lane) {
    // ---- epilogue: col = n0 + 32nb + (lane&31); row = m0 + 32wave + acc_row(r, lane)
    const int rbase = m0 + 32 * wave;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        const int col = n0 + 32 * nb + (lane & 31);
        float bias = 0.f;
        if constexpr (EPI == EPI_BIAS_RES || EPI == EPI_BIAS_GELU) bias = g.bias ? g.bias[col] : 0.f;
        bf16x8 gp0 = {}, gp1 = {};  // gelu' of this lane's 16 accumulator slots (fragment order)
        if constexpr (EPI == EPI_DGELU) {
            const bf16_t* fp = g.aux + frag_index(g, m0, n0, nb, wave, lane);
            gp0 = *(const bf16x8*)fp;
            gp1 = *(const bf16x8*)(fp + 8);
        }
        float csum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + acc_row(r, lane);
            const bool ok = row < g.M;
            float v = acc[nb][r];
            if constexpr (EPI == EPI_BF16) {
                if (ok) ((bf16_t*)g.C)[(size_t)row * g.ldc + col] = (bf16_t)v;
            } else if constexpr (EPI == EPI_F32) {
                if (ok) ((float*)g.C)[(size_t)row * g.ldc + col] = v;
            } else if constexpr (EPI == EPI_BIAS_RES) {
                v += bias;
                if (g.drop.thresh) v = drop_keep(g.drop.key, row, col, g.drop.thresh) ? v * g.drop.inv_keep : 0.f;
                if (ok) ((float*)g.C)[(size_t)row * g.ldc + col] = resv[nb][r] + v;
            } else if constexpr (EPI == EPI_BIAS_GELU) {
                v += bias;
                // gelu and gelu' share erf / exp: the backward only needs gelu'(v), stored in place of v
                const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
                const float gp = cdf + v * 0.3989422804014327f * __expf(-0.5f * v * v);
                float a = v * cdf;
                if (g.drop.thresh) a = drop_keep(g.drop.key, row, col, g.drop.thresh) ? a * g.drop.inv_keep : 0.f;
                if (r < 8) gp0[r & 7] = (bf16_t)(ok ? gp : 0.f); else gp1[r & 7] = (bf16_t)(ok ? gp : 0.f);
                if (ok) {
                    const bf16_t ah = (bf16_t)a;
                    g.C2[(size_t)row * g.ldc2 + col] = ah;
                    if (g.C2_lo) g.C2_lo[(size_t)row * g.ldc2 + col] = (bf16_t)(a - (float)ah);
                }
            } else if constexpr (EPI == EPI_DGELU) {
                float d = 0.f;
                if (ok) {
                    d = v * (float)(r < 8 ? gp0[r & 7] : gp1[r & 7]);  // aux = gelu'(pre-activation), saved by FC1
                    if (g.drop.thresh) d = drop_keep(g.drop.key, row, col, g.drop.thresh) ? d * g.drop.inv_keep : 0.f;
                    const bf16_t db = (bf16_t)d;
                    ((bf16_t*)g.C)[(size_t)row * g.ldc + col] = db;
                    d = (float)db;
                }
                csum += d;
            }
        }
        if constexpr (EPI == EPI_DGELU) {
            if (g.colsum) {  // only when the weight-gradient GEMM cannot carry the bias column (no pad column)
                csum += __shfl_xor(csum, 32);
                if (lane < 32 && col < g.n_valid) atomicAdd(&g.colsum[col], csum);
            }
        }
        if constexpr (EPI == EPI_BIAS_GELU) {
            bf16_t* fp = (bf16_t*)g.C + frag_index(g, m0, n0, nb, wave, lane);
            *(bf16x8*)fp = gp0;
            *(bf16x8*)(fp + 8) = gp1;
        }
    }
}


template <int NBLK, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNTArgs g) {
    constexpr int BN = 32 * NBLK;
    constexpr int B_CHUNKS = BN * 4;                  // 16-B chunks in a B tile
    constexpr int B_ITERS = (B_CHUNKS + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16_t sA[2][BM * LS];
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][BN * LS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = g.N / BN;
    const int nwg = gridDim.x;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int tile_m = lid / ntn, tile_n = lid % ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = g.K / BK;

    // residual operand of the epilogue: its 16*NBLK narrow loads are issued HERE, in the prologue, so their
    // latency hides under the K loop (issued in the epilogue they cost more than the MFMAs of these skinny GEMMs)
    f32x16 resv[NBLK];
    if constexpr (EPI == EPI_BIAS_RES) {
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 32 * wave + acc_row(r, lane);
                resv[nb][r] = (row < g.M) ? g.res[(size_t)row * g.ldres + n0 + 32 * nb + (lane & 31)] : 0.f;
            }
    }
    u32x4 ra[2], rb[B_ITERS];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            const int gr = m0 + row;
            ra[i] = (gr < g.M) ? *(const u32x4*)(g.A + (size_t)gr * g.lda + k0 + 8 * kc) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            if (c < B_CHUNKS) rb[i] = *(const u32x4*)(g.B + (size_t)(n0 + row) * g.ldb + k0 + 8 * kc);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            *(u32x4*)(&sA[buf][row * LS + 8 * kc]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            if (c < B_CHUNKS) *(u32x4*)(&sB[buf][row * LS + 8 * kc]) = rb[i];
        }
    };

    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    gload(0);
    swrite(0);
    __syncthreads();
    const int frag_off = (lane & 31) * LS + 8 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 a = *(const bf16x8*)(&sA[buf][32 * wave * LS + frag_off + 16 * ks]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const bf16x8 b = *(const bf16x8*)(&sB[buf][32 * nb * LS + frag_off + 16 * ks]);
                acc[nb] = mfma32(a, b, acc[nb]);
            }
        }
        if (kt + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }

    gemm_epilogue<NBLK, EPI>(g, acc, resv, m0, n0, wave, lane);
}


// Split-bf16 variant (see GemmNTArgs): single LDS buffer (hi + lo planes of A and B), next tile's global
// loads in flight during the MFMAs, two barriers per K-tile.
template <int NBLK, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_split_kernel(GemmNTArgs g) {
    constexpr int BN = 32 * NBLK;
    constexpr int B_CHUNKS = BN * 4;
    constexpr int B_ITERS = (B_CHUNKS + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16_t sA[2][BM * LS];   // [plane]
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][BN * LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = g.N / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lid / ntn, tile_n = lid % ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = g.K / BK;

    // residual operand of the epilogue: its 16*NBLK narrow loads are issued HERE, in the prologue, so their
    // latency hides under the K loop (issued in the epilogue they cost more than the MFMAs of these skinny GEMMs)
    f32x16 resv[NBLK];
    if constexpr (EPI == EPI_BIAS_RES) {
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 32 * wave + acc_row(r, lane);
                resv[nb][r] = (row < g.M) ? g.res[(size_t)row * g.ldres + n0 + 32 * nb + (lane & 31)] : 0.f;
            }
    }
    u32x4 ra[2][2], rb[2][B_ITERS];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            const int gr = m0 + row;
            const size_t off = (size_t)gr * g.lda + k0 + 8 * kc;
            ra[0][i] = (gr < g.M) ? *(const u32x4*)(g.A + off) : u32x4{0, 0, 0, 0};
            ra[1][i] = (gr < g.M) ? *(const u32x4*)(g.A_lo + off) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            if (c < B_CHUNKS) {
                const size_t off = (size_t)(n0 + row) * g.ldb + k0 + 8 * kc;
                rb[0][i] = *(const u32x4*)(g.B + off);
                rb[1][i] = *(const u32x4*)(g.B_lo + off);
            }
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
                *(u32x4*)(&sA[pl][row * LS + 8 * kc]) = ra[pl][i];
            }
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) {
                const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
                if (c < B_CHUNKS) *(u32x4*)(&sB[pl][row * LS + 8 * kc]) = rb[pl][i];
            }
        }
    };
    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    gload(0);
    const int frag_off = (lane & 31) * LS + 8 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        swrite();
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 ah = *(const bf16x8*)(&sA[0][32 * wave * LS + frag_off + 16 * ks]);
            const bf16x8 al = *(const bf16x8*)(&sA[1][32 * wave * LS + frag_off + 16 * ks]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const bf16x8 bh = *(const bf16x8*)(&sB[0][32 * nb * LS + frag_off + 16 * ks]);
                const bf16x8 bl = *(const bf16x8*)(&sB[1][32 * nb * LS + frag_off + 16 * ks]);
                acc[nb] = mfma32(al, bh, acc[nb]);
                acc[nb] = mfma32(ah, bl, acc[nb]);
                acc[nb] = mfma32(ah, bh, acc[nb]);
            }
        }
    }
    gemm_epilogue<NBLK, EPI>(g, acc, resv, m0, n0, wave, lane);
}

template <int NBLK>
int launch_nt_n(const GemmNTArgs& a, int epi, hipStream_t s) {
    const int BN = 32 * NBLK;
    const int grid = ((a.M + BM - 1) / BM) * (a.N / BN);
    if (grid <= 0) return V1T_OK;
    if (a.A_lo) {
        if (!a.B_lo) return V1T_ERR_ARG;
        switch (epi) {
            case EPI_BF16: hipLaunchKernelGGL((gemm_nt_split_kernel<NBLK, EPI_BF16>), dim3(grid), dim3(256), 0, s, a); break;
            case EPI_F32: hipLaunchKernelGGL((gemm_nt_split_kernel<NBLK, EPI_F32>), dim3(grid), dim3(256), 0, s, a); break;
            case EPI_BIAS_RES: hipLaunchKernelGGL((gemm_nt_split_kernel<NBLK, EPI_BIAS_RES>), dim3(grid), dim3(256), 0, s, a); break;
            case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_nt_split_kernel<NBLK, EPI_BIAS_GELU>), dim3(grid), dim3(256), 0, s, a); break;
            default: return V1T_ERR_ARG;
        }
        return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    }
    switch (epi) {
        case EPI_BF16: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BF16>), dim3(grid), dim3(256), 0, s, a); break;
        case EPI_F32: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_F32>), dim3(grid), dim3(256), 0, s, a); break;
        case EPI_BIAS_RES: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BIAS_RES>), dim3(grid), dim3(256), 0, s, a); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BIAS_GELU>), dim3(grid), dim3(256), 0, s, a); break;
        case EPI_DGELU: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_DGELU>), dim3(grid), dim3(256), 0, s, a); break;
        default: return V1T_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// gemm_tn: workgroup = 4 waves = 128 Y-columns (output rows) x 32*XBLK X-columns (output cols)
// over one m_chunk of the contraction; m-tile 32 rows, both LDS images stored [m][col] and read
// with ds_read_b64_tr_b16 (row strides are 64 B x odd so the 4 rows of a transposed read hit
// 4 distinct 16-bank groups).
constexpr int TN_YS = 160;  // 128 cols + pad -> 320 B

template <int XBLK>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNArgs g) {
    constexpr int XW = 32 * XBLK;
    constexpr int XS = (XW == 64) ? 96 : (XW == 128 ? 160 : (XW == 160 ? 160 : XW + 32));
    constexpr int X_CHUNKS = 32 * (XW / 8);
    constexpr int X_ITERS = (X_CHUNKS + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16_t sY[2][32 * TN_YS];
    __shared__ __attribute__((aligned(16))) bf16_t sX[2][32 * XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 128, x0 = blockIdx.y * XW;
    const int mb = blockIdx.z * g.m_chunk;
    const int me = min(g.M, mb + g.m_chunk);
    const int nt = (me - mb + 31) / 32;
    const bool wave_on = (n0 + 32 * wave) < g.NY;  // wave-uniform

    u32x4 ry[2], rx[X_ITERS];
    auto gload = [&](int t) {
        const int mt = mb + 32 * t;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 4, cc = c & 15;
            const int m = mt + row, col = n0 + 8 * cc;
            ry[i] = (m < me && col < g.NY) ? *(const u32x4*)(g.Y + (size_t)m * g.ldy + col) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < X_ITERS; ++i) {
            const int c = tid + 256 * i, row = c / (XW / 8), cc = c % (XW / 8);
            const int m = mt + row;
            if (c < X_CHUNKS) rx[i] = (m < me) ? *(const u32x4*)(g.X + (size_t)m * g.ldx + x0 + 8 * cc) : u32x4{0, 0, 0, 0};
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 4, cc = c & 15;
            *(u32x4*)(&sY[buf][row * TN_YS + 8 * cc]) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < X_ITERS; ++i) {
            const int c = tid + 256 * i, row = c / (XW / 8), cc = c % (XW / 8);
            if (c < X_CHUNKS) *(u32x4*)(&sX[buf][row * XS + 8 * cc]) = rx[i];
        }
    };

    f32x16 acc[XBLK];
#pragma unroll
    for (int xb = 0; xb < XBLK; ++xb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xb][r] = 0.f;

    if (nt > 0) {
        gload(0);
        swrite(0);
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload(t + 1);
        if (wave_on) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 a = lds_tr_frag_nat(sY[buf], TN_YS, 16 * s, 32 * wave, lane);
#pragma unroll
                for (int xb = 0; xb < XBLK; ++xb) {
                    const bf16x8 b = lds_tr_frag_nat(sX[buf], XS, 16 * s, 32 * xb, lane);
                    acc[xb] = mfma32(a, b, acc[xb]);
                }
            }
        }
        if (t + 1 < nt) swrite(buf ^ 1);
        __syncthreads();
    }
    if (!wave_on) return;
#pragma unroll
    for (int xb = 0; xb < XBLK; ++xb) {
        const int xc = x0 + 32 * xb + (lane & 31);
        const int xs = xc / g.xseg_pad, xr = xc % g.xseg_pad;
        const bool xok = xr < g.xseg_valid;
        const bool isb = g.dbias != nullptr && xc == g.ones_col;
        const int ncol = xs * g.xseg_valid + xr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int yr = n0 + 32 * wave + acc_row(r, lane);
            const int ys = yr / g.yseg_pad, yy = yr % g.yseg_pad;
            if (yr < g.NY && yy < g.yseg_valid) {
                if (xok) atomicAdd(&g.dW[(size_t)(ys * g.yseg_valid + yy) * g.ldw + ncol], acc[xb][r] * g.alpha);
                else if (isb) atomicAdd(&g.dbias[ys * g.yseg_valid + yy], acc[xb][r] * g.alpha);
            }
        }
    }
}

}  // namespace

int launch_gemm_nt(const GemmNTArgs& a_, int epi, hipStream_t s) {
    GemmNTArgs a = a_;
    static const int dbg = std::getenv("V1T_DBG_GEMM") ? std::atoi(std::getenv("V1T_DBG_GEMM")) : 0;
    a.dbg = dbg;
    if (a.K % BK != 0 || a.N % 32 != 0 || (a.lda % 8) || (a.ldb % 8)) return V1T_ERR_ARG;
    if (a.N % 160 == 0) return launch_nt_n<5>(a, epi, s);
    if (a.N % 128 == 0) return launch_nt_n<4>(a, epi, s);
    if (a.N % 64 == 0) return launch_nt_n<2>(a, epi, s);
    return launch_nt_n<1>(a, epi, s);
}

int launch_gemm_tn(const GemmTNArgs& a, hipStream_t s) {
    if (a.NY % 32 != 0 || a.NX % 32 != 0 || a.m_chunk % 32 != 0 || (a.ldy % 8) || (a.ldx % 8)) return V1T_ERR_ARG;
    if (a.M <= 0) return V1T_OK;
    const int gy = (a.NY + 127) / 128, gz = (a.M + a.m_chunk - 1) / a.m_chunk;
#define TN_LAUNCH(XB)                                                                                       \
    hipLaunchKernelGGL((gemm_tn_kernel<XB>), dim3(gy, a.NX / (32 * XB), gz), dim3(256), 0, s, a)
    if (a.NX % 160 == 0) TN_LAUNCH(5);
    else if (a.NX % 128 == 0) TN_LAUNCH(4);
    else if (a.NX % 64 == 0) TN_LAUNCH(2);
    else TN_LAUNCH(1);
#undef TN_LAUNCH
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
