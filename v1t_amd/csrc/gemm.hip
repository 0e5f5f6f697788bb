// v1t_amd — bf16 MFMA GEMM kernels (gfx950). See gemm.h.
//
// gemm_nt tiling: workgroup = 4 waves = 128 rows x (32*NBLK) cols; wave w owns rows 32w..32w+31 and
// all NBLK 32-col MFMA blocks (A fragment read once, reused NBLK times). K-tile 32, LDS rows padded
// by 16 B (80-B stride -> the 16 rows a ds_read_b128 lane group touches land on 16 distinct
// 16-B slots), double-buffered LDS, next tile's global loads issued before the MFMAs and written
// to the other buffer after them (one barrier per K-tile).
#include <type_traits>
#include <cstdlib>

#include "gemm.h"
#include "elementwise.h"

namespace {

// Stores of planes that only the BACKWARD reads (LayerNorm outputs, gelu', the fp16 activation plane of the fused MLP forward): non-temporal, so
// that 0.4-0.5 GB per block written once do not evict the weight tiles and the residual rows the forward's next kernels read from L2
// (-DV1T_NO_NT_SAVED: plain stores, A/B builds).
template <typename V>
DEVFN void store_saved(V* p, const V& v) {
#ifdef V1T_NO_NT_SAVED
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
constexpr int FRAG_BM = 128;  // row tile of the kernels that read / write fragment-order buffers (NW = 4)

DEVFN size_t frag_index(const GemmNTArgs& g, int m0, int n0, int nb, int wave, int lane) {
    return ((((size_t)(m0 / FRAG_BM) * (g.N / 32) + (n0 / 32 + nb)) * 4 + wave) * 64 + lane) * 16;
}

// STAGED (EPI_BIAS_GELU): the activation is left in `acc` (fp32) for the caller's LDS-staged stores instead of being
// written with 2-byte stores here.
template <int NBLK, int EPI, bool STAGED = false>
DEVFN void gemm_epilogue(const GemmNTArgs& g, f32x16 (&acc)[NBLK], const f32x16 (&resv)[NBLK], int m0, int n0, int wave, int lane) {
    // ---- epilogue: col = n0 + 32nb + (lane&31); row = m0 + 32wave + acc_row(r, lane)
    const int rbase = m0 + 32 * wave;
    int trow[16] = {};  // EPI_PATCH: token index of each accumulator row (one modulo per row, not per element)
    if constexpr (EPI == EPI_PATCH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) trow[r] = (rbase + acc_row(r, lane)) % g.T;
    }
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        const int col = n0 + 32 * nb + (lane & 31);
        float bias = 0.f;
        if constexpr (EPI == EPI_BIAS_RES || EPI == EPI_BIAS_GELU) bias = g.bias ? g.bias[col] : 0.f;
        bf16x8 gp0 = {}, gp1 = {};  // gelu' of this lane's 16 accumulator slots (fragment order)
        if constexpr (EPI == EPI_DGELU) {  // prefetched by the kernel before its last K tile: 8 dwords in resv[nb][0..7]
            f32x4 lo = {resv[nb][0], resv[nb][1], resv[nb][2], resv[nb][3]}, hi = {resv[nb][4], resv[nb][5], resv[nb][6], resv[nb][7]};
            gp0 = __builtin_bit_cast(bf16x8, lo);
            gp1 = __builtin_bit_cast(bf16x8, hi);
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
            } else if constexpr (EPI == EPI_PATCH) {
                const int t = trow[r];
                const bool cok = col < g.n_valid;
                if (ok) {
                    v = cok ? (t == 0 ? g.cls[col] + g.pos[col] : v + g.bias[col] + g.pos[(size_t)t * g.n_valid + col]) : 0.f;
                    if (g.drop.thresh && cok) v = drop_keep(g.drop.key, row, col, g.drop.thresh) ? v * g.drop.inv_keep : 0.f;
                    ((float*)g.C)[(size_t)row * g.ldc + col] = v;
                }
            } else if constexpr (EPI == EPI_BIAS_RES) {
                v += bias;
                if (g.drop.thresh) v = drop_keep(g.drop.key, row, col, g.drop.thresh) ? v * g.drop.inv_keep : 0.f;
                if (g.row_scale && ok) v *= g.row_scale[row / g.T];
                if (ok) ((float*)g.C)[(size_t)row * g.ldc + col] = resv[nb][r] + v;
            } else if constexpr (EPI == EPI_BIAS_GELU) {
                v += bias;
                // gelu and gelu' share one exp: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below the 2^-17 of the
                // hi + lo activation planes) on E = exp(-v^2/2), which is also the pdf factor of gelu'(v); libm's erff
                // alone cost ~35 instructions per element and made this epilogue longer than the K loop
                const float ax = fabsf(v) * 0.70710678118654752f;
                const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
                const float E = __expf(-0.5f * v * v);
                const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
                const float erfa = fmaf(-poly, E, 1.0f);                    // erf(|v| / sqrt 2)
                const float cdf = 0.5f * (1.0f + copysignf(erfa, v));
                const float gp = cdf + v * 0.3989422804014327f * E;
                float a = v * cdf;
                if (g.drop.thresh) a = drop_keep(g.drop.key, row, col, g.drop.thresh) ? a * g.drop.inv_keep : 0.f;
                if (r < 8) gp0[r & 7] = (bf16_t)(ok ? gp : 0.f); else gp1[r & 7] = (bf16_t)(ok ? gp : 0.f);
                if constexpr (STAGED) {
                    acc[nb][r] = a;
                } else if (ok) {
                    const bf16_t ah = (bf16_t)a;
                    if (g.C2) g.C2[(size_t)row * g.ldc2 + col] = ah;
                    if (g.C2_lo) g.C2_lo[(size_t)row * g.ldc2 + col] = aux_plane(a, ah, g.f16);
                }
            } else if constexpr (EPI == EPI_DGELU) {
                float d = 0.f;
                if (ok) {
                    d = v * (float)(r < 8 ? gp0[r & 7] : gp1[r & 7]);  // aux = gelu'(pre-activation), saved by FC1
                    if (g.drop.thresh) d = drop_keep(g.drop.key, row, col, g.drop.thresh) ? d * g.drop.inv_keep : 0.f;
                    const bf16_t db = (bf16_t)d;
                    if constexpr (!STAGED) ((bf16_t*)g.C)[(size_t)row * g.ldc + col] = db;
                    d = (float)db;
                }
                if constexpr (STAGED) acc[nb][r] = d;
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
            if (!g.lean) {  // kernel-uniform: inference forwards skip the gelu' plane (433 MB per block at batch 256)
                bf16_t* fp = (bf16_t*)g.C + frag_index(g, m0, n0, nb, wave, lane);
                store_saved((bf16x8*)fp, gp0);
                store_saved((bf16x8*)(fp + 8), gp1);
            }
        }
    }
}


// C = res + dropout(acc + bias) * row_scale over a workgroup's BM x 32 NBLK tile whose accumulators hold rows 32 wave + acc_row(r, lane), columns
// n0 + 32 nb + (lane & 31) (the proj / FC2 epilogue of gemm_nt_kernel; also the tail of mlp_fwd_kernel):
template <int NBLK, int BM>
DEVFN void epi_bias_res(const GemmNTArgs& g, f32x16 (&acc)[NBLK], int m0, int n0, int wave, int lane) {
    const int rows_here = min(BM, g.M - m0);  // < BM in the last row tile only: rows past M then get an out-of-range offset (reads 0, stores dropped)
    const __amdgpu_buffer_rsrc_t rr = buf_rsrc(g.res ? g.res + (size_t)m0 * g.ldres + n0 : nullptr, g.res ? (uint32_t)rows_here * (uint32_t)g.ldres * 4u : 0u);
    const __amdgpu_buffer_rsrc_t cr = buf_rsrc((float*)g.C + (size_t)m0 * g.ldc + n0, (uint32_t)rows_here * (uint32_t)g.ldc * 4u);
    const uint32_t lrow = (uint32_t)(32 * wave + 4 * (lane >> 5));
    const uint32_t dthr = g.drop.thresh;  // 0 keeps everything (hash >= 0)
    const float dinv = g.drop.thresh ? g.drop.inv_keep : 1.0f;
    const uint32_t vres = (lrow * (uint32_t)g.ldres + (uint32_t)(lane & 31)) * 4u, vout = (lrow * (uint32_t)g.ldc + (uint32_t)(lane & 31)) * 4u;
    auto body = [&](auto ragged_c) __attribute__((always_inline)) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            if (nb) {  // one column block at a time
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            const int col = n0 + 32 * nb + (lane & 31);
            const float bias = g.bias ? g.bias[col] : 0.f;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ar = (r & 3) + 8 * (r >> 2);
                uint32_t vo = vres;
                if (RAGGED && (int)lrow + ar >= rows_here) vo |= BUF_OOB;
                rv[r] = buf_load_f32(rr, vo, (ar * g.ldres + 32 * nb) * 4);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ar = (r & 3) + 8 * (r >> 2);
                const uint32_t grow = (uint32_t)m0 + lrow + (uint32_t)ar;
                const bool keep = drop_hash(g.drop.key, grow, col) >= dthr;
                float v = keep ? (acc[nb][r] + bias) * dinv : 0.f;
                if (g.row_scale) v *= g.row_scale[min((int)grow, g.M - 1) / g.T];
                uint32_t vo = vout;
                if (RAGGED && (int)lrow + ar >= rows_here) vo |= BUF_OOB;
                buf_store_f32(cr, vo, (ar * g.ldc + 32 * nb) * 4, rv[r] + v);
            }
        }
    };
    if (rows_here == BM) body(std::false_type{});  // workgroup-uniform
    else body(std::true_type{});
}

// ---- LDS-DMA ring K loop (round 5): launches with at most ONE workgroup per CU ---------------------------------------------------------
// A launch of <= 256 workgroups (a rank's share of a multi-GPU step, a 16-image launch of the per-mouse loop) runs one 4-wave workgroup per
// CU, i.e. one wave per SIMD: nothing overlaps a workgroup's K loop, and with the register-staged double buffer a CU has ONE K tile of
// operand loads (36 KB) in flight - it then fetches at ~12 B per clock (outstanding bytes / memory latency: DESIGN.md "what bounds"), while
// the same kernel is bandwidth-bound at full-size launches where 2-3 workgroups per CU interleave. Registers cannot hold more tiles (three
// register sets pushed gemm_nt to one wave per SIMD in AGPR form: profiles/r04_gemm_experiments.txt). LDS can: with one workgroup per CU
// the whole 160 KB are this workgroup's, so the operand tiles go global -> LDS by LDS-DMA (no registers at all) into a ring of NS stages of
// [128 + 32 NBLK rows][64 k] bf16 (36 KB at NBLK = 5), NS - 1 stages (108 KB) in flight behind a counted vmcnt, one barrier per 64-wide K
// tile. Layout: rows unpadded (a DMA instruction writes 1 KB = 8 rows linearly); the 16-B chunk c of row r sits at chunk position
// c ^ ((r >> 1) & 7), applied on the GLOBAL side (each lane fetches the chunk its LDS slot holds), so that the MFMA fragment reads
// (16 consecutive rows x 16 B per quarter-wave) touch every bank once. Instruction j = wave + 4 i of a stage covers rows 8 j .. 8 j + 7:
// i < 4 the A rows, i >= 4 the B rows; j & 1 = wave & 1, so a lane fetches the SAME chunk index in every instruction it issues.
constexpr int RING_BK = 64, RING_BM = 128;
template <int NBLK> constexpr int ring_stage_elems() { return (RING_BM + 32 * NBLK) * RING_BK; }
template <int N> DEVFN void wait_vmcnt_imm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// arow(i) = GLOBAL row of A (already clamped into the valid range) that tile row 8 wave + 32 i + (lane >> 3) reads, i = 0..3
template <int NBLK, int NS, bool F16, typename RowFn>
DEVFN void ring_kloop(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ Bm, int ldb, int nk, bf16_t* smem, int wave, int lane,
                      RowFn arow, f32x16 (&acc)[NBLK]) {
    constexpr int PW = 4 + NBLK;                    // DMA instructions per wave and stage
    constexpr int STAGE = ring_stage_elems<NBLK>();  // bf16 elements
    static_assert(NS >= 3 && (NS - 2) * PW < 64, "ring depth");
    const int c = (lane & 7) ^ (4 * (wave & 1) + (lane >> 4));  // the chunk this lane fetches (see above)
    const bf16_t* ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ap[i] = A + (size_t)arow(i) * lda + 8 * c;
    const bf16_t* bp = Bm + (size_t)(8 * wave + (lane >> 3)) * ldb + 8 * c;
    auto issue = [&](int kt) {
        bf16_t* st = smem + (kt % NS) * STAGE + wave * 512;  // instruction j at 1 KB * j = 512 elements * (wave + 4 i)
        const int k0 = kt * RING_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(ap[i] + k0, st + 2048 * i);
#pragma unroll
        for (int i = 0; i < NBLK; ++i) lds_dma16(bp + (size_t)(32 * i) * ldb + k0, st + 2048 * (4 + i));
    };
    // fragment reads: row (lane & 31) of a 32-row block, k chunk 2 ks + (lane >> 5), at chunk position (2 ks + h) ^ ((row >> 1) & 7)
    int foff[RING_BK / 16];
    {
        const int row = lane & 31, h = lane >> 5, sw = (row >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < RING_BK / 16; ++ks) foff[ks] = row * RING_BK + 8 * ((2 * ks + h) ^ sw);
    }
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) issue(t);
    for (int t = 0; t < nk; ++t) {
        const int rem = nk - 1 - t;  // stages behind t that have been issued: min(rem, NS - 2)
        if (rem >= NS - 2) wait_vmcnt_imm<(NS - 2) * PW>();
        else if (NS > 3 && rem == 1) wait_vmcnt_imm<PW>();
        else if (NS > 4 && rem == 2) wait_vmcnt_imm<2 * PW>();
        else wait_vmcnt_imm<0>();
        __syncthreads();  // stage t has landed for every wave, and every wave is done with stage t - 1, whose slot the next issue overwrites
        if (t + NS - 1 < nk) issue(t + NS - 1);
        const bf16_t* sa = smem + (t % NS) * STAGE + 32 * wave * RING_BK;
        const bf16_t* sb = smem + (t % NS) * STAGE + RING_BM * RING_BK;
#pragma unroll
        for (int ks = 0; ks < RING_BK / 16; ++ks) {
            const bf16x8 a = *(const bf16x8*)(sa + foff[ks]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const bf16x8 b = *(const bf16x8*)(sb + 32 * nb * RING_BK + foff[ks]);
                if constexpr (F16) acc[nb] = mfma32h(a, b, acc[nb]);
                else acc[nb] = mfma32(a, b, acc[nb]);
            }
        }
    }
    __syncthreads();  // every wave is done with the operand stages: the epilogues stage their outputs in the same memory
}
constexpr int RING_NS = 4;  // 4 x 36 KB at NBLK = 5

// NW waves per workgroup = 32*NW rows (only 4 is launched: 64-row workgroups measured the same time on the
// single-column-tile GEMMs, which are bound by how they read A, not by workgroups in flight).
template <int NBLK, int EPI, int NW, int BK, bool F16 = false, bool RD = false, int RING = 0>
__global__ __launch_bounds__(64 * NW, ((EPI == EPI_BIAS_GELU && BK == 32 && NBLK <= 4) || ((EPI == EPI_BIAS_RES || EPI == EPI_DGELU || EPI == EPI_BF16 || EPI == EPI_PATCH) && BK == 32 && NW == 4)) ? 3 : 1) void gemm_nt_kernel(GemmNTArgs g) {
    static_assert(!RD || (EPI == EPI_BF16 && NBLK == 5), "row dot: bf16 output, one 160-column tile per head");
    static_assert(RING == 0 || (EPI == EPI_BIAS_RES && NW == 4 && BK == RING_BK), "ring K loop: the proj / FC2 form (nothing peeled around the last K tile)");
    constexpr int BM = 32 * NW, NT = 64 * NW;
    constexpr int LS = BK + 8, KC = BK / 8;  // LDS row stride (16-B pad: 80 / 144 B), 16-B chunks per row
    constexpr int A_ITERS = BM * KC / NT;
    constexpr int BN = 32 * NBLK;
    constexpr int B_CHUNKS = BN * KC;                 // 16-B chunks in a B tile
    constexpr int B_ITERS = (B_CHUNKS + NT - 1) / NT;
    static_assert(NW == 4 || (EPI != EPI_BIAS_GELU && EPI != EPI_DGELU), "fragment-order buffers assume 128-row tiles");
    // one allocation: the operand tiles (sA, sB) and, after the K loop, the bf16 output staging of EPI_BF16
    constexpr int CS = BN + 8;  // staging row stride (elements): 16-B aligned rows, odd multiple of 16 B
    constexpr int SMEM_DB = (2 * BM * LS + 2 * BN * LS) > (BM * CS) ? (2 * BM * LS + 2 * BN * LS) : (BM * CS);
    constexpr int SMEM = RING ? RING * ring_stage_elems<NBLK>() : SMEM_DB;
    __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM];
    bf16_t (*sA)[BM * LS] = (bf16_t (*)[BM * LS])smem;
    bf16_t (*sB)[BN * LS] = (bf16_t (*)[BN * LS])(smem + 2 * BM * LS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = g.N / BN;
    const int nwg = gridDim.x;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int tile_m = lid / ntn, tile_n = lid % ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = g.K / BK;

    // residual operand of the epilogue: its 16*NBLK narrow loads are issued at the start of the LAST K tile (peeled below), so
    // their latency hides under that tile's MFMAs without the 16*NBLK registers being live through the whole K loop next to
    // the staging registers (held from the prologue they pushed the proj / FC2 kernels to 284 VGPRs = one workgroup per CU)
    f32x16 resv[NBLK];
    auto load_res = [&]() {};  // EPI_BIAS_RES fetches the residual block by block in its epilogue (below)
    u32x4 ra[A_ITERS], rb[B_ITERS];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int c = tid + NT * i, row = c / KC, kc = c % KC;
            const int gr = m0 + row;
            ra[i] = (gr < g.M) ? *(const u32x4*)(g.A + (size_t)gr * g.lda + k0 + 8 * kc) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NT * i, row = c / KC, kc = c % KC;
            if (c < B_CHUNKS) rb[i] = *(const u32x4*)(g.B + (size_t)(n0 + row) * g.ldb + k0 + 8 * kc);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int c = tid + NT * i, row = c / KC, kc = c % KC;
            *(u32x4*)(&sA[buf][row * LS + 8 * kc]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NT * i, row = c / KC, kc = c % KC;
            if (c < B_CHUNKS) *(u32x4*)(&sB[buf][row * LS + 8 * kc]) = rb[i];
        }
    };

    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  if constexpr (RING) {
    ring_kloop<NBLK, RING, F16>(g.A, g.lda, g.B + (size_t)n0 * g.ldb, g.ldb, nk, smem, wave, lane,
                                [&](int i) { return min(m0 + 8 * wave + 32 * i + (lane >> 3), g.M - 1); }, acc);
  } else {
    gload(0);
    swrite(0);
    __syncthreads();
    const int frag_off = (lane & 31) * LS + 8 * (lane >> 5);
    auto ktile = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const bf16x8 a = *(const bf16x8*)(&sA[buf][32 * wave * LS + frag_off + 16 * ks]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const bf16x8 b = *(const bf16x8*)(&sB[buf][32 * nb * LS + frag_off + 16 * ks]);
                if constexpr (F16) acc[nb] = mfma32h(a, b, acc[nb]);
                else acc[nb] = mfma32(a, b, acc[nb]);
            }
        }
    };
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int buf = kt & 1;
        gload(kt + 1);
        ktile(buf);
        swrite(buf ^ 1);
        __syncthreads();
    }
    if constexpr (EPI == EPI_BIAS_RES) load_res();
    if constexpr (EPI == EPI_DGELU) {  // saved gelu' fragments (2 x 16 B per block), same early issue
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const bf16_t* fp = g.aux + frag_index(g, m0, n0, nb, wave, lane);
            const f32x4 lo = *(const f32x4*)fp, hi = *(const f32x4*)(fp + 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                resv[nb][j] = lo[j];
                resv[nb][4 + j] = hi[j];
            }
        }
    }
    ktile((nk - 1) & 1);
    __syncthreads();
  }

    // 16-bit outputs go through LDS: a lane holds one column of 16 rows per block, i.e. 2-byte global stores that fill
    // 64 B of two rows per instruction; staged [32 rows][BN] per wave (wave-private, no barrier) and written back as 16-B
    // chunks, an instruction covers 1 KB of consecutive row segments (QKV: 101 MB of output per 16 images; the GEMM was
    // bound by these stores)
    auto staged_store = [&](bf16_t* dst, int ld, auto conv) {
        bf16_t* st = smem + wave * 32 * CS;
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[acc_row(r, lane) * CS + 32 * nb + (lane & 31)] = conv(acc[nb][r]);
        constexpr int CPR = BN / 8;  // 16-B chunks per row
#pragma unroll
        for (int c0 = 0; c0 < 32 * CPR; c0 += 64) {
            const int c = c0 + lane;
            if (c < 32 * CPR) {
                const int row = c / CPR, ch = c % CPR;
                const int grow = m0 + 32 * wave + row;
                if (grow < g.M) *(u32x4*)(dst + (size_t)grow * ld + n0 + 8 * ch) = *(const u32x4*)(st + row * CS + 8 * ch);
            }
        }
    };
    if constexpr (EPI == EPI_BF16 && RD) {
        // staged store + row dot (RowDotArgs): the second matrix's chunks are fetched first (10 x 16 B per lane, in flight while the tile is
        // staged), every 16-B chunk of the rounded output meets its partner on the way out, the 20 chunk sums of a row meet in LDS (float atomics
        // of this wave only), lanes 0-31 write the row constants of the wave's 32 rows
        constexpr int CPR = BN / 8, NIT = (32 * CPR + 63) / 64;
        static_assert(32 * (CPR + 1) * 4 <= 32 * CS * 2, "the chunk sums fit the wave's staging region");
        u32x4 ov[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = 64 * it + lane, row = c / CPR, ch = c % CPR;
            const int grow = m0 + 32 * wave + row;
            ov[it] = (c < 32 * CPR && grow < g.M) ? *(const u32x4*)(g.rd.o + (size_t)grow * g.rd.ldo + n0 + 8 * ch) : u32x4{0, 0, 0, 0};
        }
        __syncthreads();  // every wave is done with the operand tiles
        bf16_t* st = smem + wave * 32 * CS;
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[acc_row(r, lane) * CS + 32 * nb + (lane & 31)] = (bf16_t)acc[nb][r];
        float parts[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            parts[it] = 0.f;
            const int c = 64 * it + lane;
            if (c < 32 * CPR) {
                const int row = c / CPR, ch = c % CPR;
                const int grow = m0 + 32 * wave + row;
                const u32x4 dv = *(const u32x4*)(st + row * CS + 8 * ch);
                if (grow < g.M) *(u32x4*)((bf16_t*)g.C + (size_t)grow * g.ldc + n0 + 8 * ch) = dv;
                const bf16x8 d8 = __builtin_bit_cast(bf16x8, dv), o8 = __builtin_bit_cast(bf16x8, ov[it]);
                float part = 0.f;
                if (g.rd.o_f16) {  // kernel-uniform
                    const f16x8 oh = __builtin_bit_cast(f16x8, ov[it]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) part = fmaf((float)d8[j], (float)oh[j], part);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) part = fmaf((float)d8[j], (float)o8[j], part);
                }
                parts[it] = part;
            }
        }
        // the chunk sums go where the tile was staged (every read of it is behind this wave: its LDS operations execute in order), as
        // [32 rows][CPR + 1] floats: chunk c of the wave sits at c + c / CPR, a row's CPR sums are read by one lane, both without bank conflicts
        float* sp = (float*)st;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = 64 * it + lane;
            if (c < 32 * CPR) sp[c + c / CPR] = parts[it];
        }
        if (lane < 32) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < CPR; ++j) dot += sp[lane * (CPR + 1) + j];
            const int grow = m0 + 32 * wave + lane;
            if (grow < g.M) {
                const int bi = grow / g.rd.T, t = grow - bi * g.rd.T, bh = bi * g.rd.H + tile_n;
                g.rd.ndelta[(size_t)bh * g.rd.TPQ + t] = -dot * g.rd.keep;
                g.rd.nlse[(size_t)bh * g.rd.TPQ + t] = -g.rd.lse2[(size_t)bh * g.rd.T + t];
            }
        }
    } else if constexpr (EPI == EPI_BF16) {
        __syncthreads();  // every wave is done with the operand tiles
        staged_store((bf16_t*)g.C, g.ldc, [](float v) { return (bf16_t)v; });
    } else if constexpr (EPI == EPI_BIAS_GELU) {
        gemm_epilogue<NBLK, EPI, true>(g, acc, resv, m0, n0, wave, lane);  // gelu' (fragment order) written, activation left in acc
        __syncthreads();
        if (g.C2) staged_store(g.C2, g.ldc2, [](float v) { return (bf16_t)v; });
        if (g.C2_lo) {
            const int f16 = g.f16;
            staged_store(g.C2_lo, g.ldc2, [f16](float v) { return aux_plane(v, (bf16_t)v, f16); });
        }
    } else if constexpr (EPI == EPI_DGELU) {
        // the gradient (already rounded to bf16) is left in acc and leaves through the same staged 16-B row stores: element-wise it
        // was 64 two-byte stores per lane, 64 B of two rows per instruction (190 MB per 112-image launch at the default shape)
        gemm_epilogue<NBLK, EPI, true>(g, acc, resv, m0, n0, wave, lane);
        __syncthreads();
        staged_store((bf16_t*)g.C, g.ldc, [](float v) { return (bf16_t)v; });
    } else if constexpr (EPI == EPI_BIAS_RES) {
        // C = res + dropout(acc + bias) * row_scale with the accumulator layout's 4-B accesses through buffer resources over this workgroup's
        // rows: scalar base, ONE per-lane offset register (row 32 wave + 4 (lane >> 5), column lane & 31), the accumulator row (r & 3) + 8 (r >> 2)
        // and the column block in the scalar offset - no 64-bit address pair per access, the residual of one column block (16 loads) in flight
        // at a time instead of all 16 NBLK values held through the last K tile (152 VGPRs -> two workgroups per CU)
        epi_bias_res<NBLK, BM>(g, acc, m0, n0, wave, lane);
    } else {
        gemm_epilogue<NBLK, EPI>(g, acc, resv, m0, n0, wave, lane);
    }
}


// Split-bf16 variant (see GemmNTArgs): single LDS buffer (hi + lo planes of A and B), next tile's global
// loads in flight during the MFMAs, two barriers per K-tile.
template <int NBLK, int EPI, int NW, int BK>
__global__ __launch_bounds__(64 * NW) void gemm_nt_split_kernel(GemmNTArgs g) {
    constexpr int BM = 32 * NW, NT = 64 * NW;
    constexpr int LS = BK + 8, KC = BK / 8;
    constexpr int A_ITERS = BM * KC / NT;
    constexpr int BN = 32 * NBLK;
    constexpr int B_CHUNKS = BN * KC;
    constexpr int B_ITERS = (B_CHUNKS + NT - 1) / NT;
    static_assert(NW == 4 || (EPI != EPI_BIAS_GELU && EPI != EPI_DGELU), "fragment-order buffers assume 128-row tiles");
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
                resv[nb][r] = (g.res && row < g.M) ? g.res[(size_t)row * g.ldres + n0 + 32 * nb + (lane & 31)] : 0.f;
            }
    }
    u32x4 ra[2][A_ITERS], rb[2][B_ITERS];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int c = tid + NT * i, row = c / KC, kc = c % KC;
            const int gr = m0 + row;
            const size_t off = (size_t)gr * g.lda + k0 + 8 * kc;
            ra[0][i] = (gr < g.M) ? *(const u32x4*)(g.A + off) : u32x4{0, 0, 0, 0};
            ra[1][i] = (gr < g.M) ? *(const u32x4*)(g.A_lo + off) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NT * i, row = c / KC, kc = c % KC;
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
            for (int i = 0; i < A_ITERS; ++i) {
                const int c = tid + NT * i, row = c / KC, kc = c % KC;
                *(u32x4*)(&sA[pl][row * LS + 8 * kc]) = ra[pl][i];
            }
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) {
                const int c = tid + NT * i, row = c / KC, kc = c % KC;
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
        for (int ks = 0; ks < BK / 16; ++ks) {
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

// V1T_GEMM_RING=0 (dev, A/B): the register-staged double buffer also for launches of at most one workgroup per CU
static const bool g_gemm_ring = !(dev_env("V1T_GEMM_RING") && !atoi(dev_env("V1T_GEMM_RING")));
template <int NBLK, int NW, int BK>
int launch_nt_nw(const GemmNTArgs& a, int epi, hipStream_t s) {
    const int BN = 32 * NBLK, BMW = 32 * NW;
    const int grid = ((a.M + BMW - 1) / BMW) * (a.N / BN);
    if (grid <= 0) return V1T_OK;
    const dim3 blk(64 * NW);
    if (a.f16) {
        if (a.A_lo || a.B_lo) return V1T_ERR_ARG;
        switch (epi) {
            case EPI_BF16: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BF16, NW, BK, true>), dim3(grid), blk, 0, s, a); break;
            case EPI_F32: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_F32, NW, BK, true>), dim3(grid), blk, 0, s, a); break;
            case EPI_BIAS_RES:
                if constexpr (NBLK == 5 && NW == 4 && BK == RING_BK) {
                    // at most one workgroup per CU: the LDS-DMA ring K loop (three 36-KB stages in flight instead of one register-staged tile)
                    if (g_gemm_ring && grid <= 256 && a.K / BK >= 3) {
                        hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BIAS_RES, NW, BK, true, false, RING_NS>), dim3(grid), blk, 0, s, a);
                        break;
                    }
                }
                hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BIAS_RES, NW, BK, true>), dim3(grid), blk, 0, s, a);
                break;
            case EPI_PATCH: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_PATCH, NW, BK, true>), dim3(grid), blk, 0, s, a); break;
            case EPI_BIAS_GELU:
                if constexpr (NW == 4) { hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BIAS_GELU, 4, BK, true>), dim3(grid), blk, 0, s, a); break; }
                return V1T_ERR_ARG;
            default: return V1T_ERR_ARG;
        }
        return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    }
    if (a.A_lo) {
        if (!a.B_lo) return V1T_ERR_ARG;
        switch (epi) {
            case EPI_BF16: hipLaunchKernelGGL((gemm_nt_split_kernel<NBLK, EPI_BF16, NW, BK>), dim3(grid), blk, 0, s, a); break;
            case EPI_F32: hipLaunchKernelGGL((gemm_nt_split_kernel<NBLK, EPI_F32, NW, BK>), dim3(grid), blk, 0, s, a); break;
            case EPI_BIAS_RES: hipLaunchKernelGGL((gemm_nt_split_kernel<NBLK, EPI_BIAS_RES, NW, BK>), dim3(grid), blk, 0, s, a); break;
            case EPI_PATCH: hipLaunchKernelGGL((gemm_nt_split_kernel<NBLK, EPI_PATCH, NW, BK>), dim3(grid), blk, 0, s, a); break;
            case EPI_BIAS_GELU:
                if constexpr (NW == 4) { hipLaunchKernelGGL((gemm_nt_split_kernel<NBLK, EPI_BIAS_GELU, 4, BK>), dim3(grid), blk, 0, s, a); break; }
                return V1T_ERR_ARG;
            default: return V1T_ERR_ARG;
        }
        return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    }
    if (a.rd.o) {
        if constexpr (NBLK == 5) {
            if (epi == EPI_BF16) {
                hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BF16, NW, BK, false, true>), dim3(grid), blk, 0, s, a);
                return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
            }
        }
        return V1T_ERR_UNSUPPORTED;
    }
    switch (epi) {
        case EPI_BF16: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BF16, NW, BK>), dim3(grid), blk, 0, s, a); break;
        case EPI_F32: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_F32, NW, BK>), dim3(grid), blk, 0, s, a); break;
        case EPI_BIAS_RES: hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BIAS_RES, NW, BK>), dim3(grid), blk, 0, s, a); break;
        case EPI_BIAS_GELU:
            if constexpr (NW == 4) { hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_BIAS_GELU, 4, BK>), dim3(grid), blk, 0, s, a); break; }
            return V1T_ERR_ARG;
        case EPI_DGELU:
            if constexpr (NW == 4) { hipLaunchKernelGGL((gemm_nt_kernel<NBLK, EPI_DGELU, 4, BK>), dim3(grid), blk, 0, s, a); break; }
            return V1T_ERR_ARG;
        default: return V1T_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
template <int NBLK>
int launch_nt_n(const GemmNTArgs& a, int epi, hipStream_t s) {
    // K-tile 64 where K allows: 128 B (a whole line) of every A row per tile instead of 64 B and half the barriers - but
    // its 83 KB of LDS leave one workgroup per CU. Measured on the whole step: 16-image launches (M = 26 k rows) 41.8 ->
    // 41.3 ms with the 64-wide tile, 112-image launches (M = 185 k) 3987 -> 4032 images/s with the 32-wide one (three
    // workgroups per CU hide more latency once there are enough of them): chosen by M.
    // Round 5: "few rows" means AT MOST ONE ROUND of one-workgroup-per-CU launches. A 28-image launch (a rank's share of a 4-GPU step: 362
    // row tiles) is under 65 536 rows too, but at one workgroup per CU it ran as two rounds, 106 of them in the second (proj 63 us against
    // 25 us at 14 images and 110 us at 112): the 32-wide tile, three workgroups per CU, holds all of it at once.
    static const int force_bk = dev_env("V1T_GEMM_BK") ? atoi(dev_env("V1T_GEMM_BK")) : 0;  // dev switch
    const long long tiles = (long long)((a.M + 127) / 128) * (a.N / (32 * NBLK));
    const bool bk64 = a.K % 64 == 0 && force_bk != 32 && (force_bk == 64 || tiles <= 256);
    return bk64 ? launch_nt_nw<NBLK, 4, 64>(a, epi, s) : launch_nt_nw<NBLK, 4, 32>(a, epi, s);
}

// ------------------------------------------------------------------------------------------
// LayerNorm fused into the GEMM that consumes it (LN1 -> QKV, LN2 -> FC1; K = DP <= 160), A-STATIONARY: a workgroup of 8
// waves owns 256 rows. Each wave loads its 32 rows of the fp32 residual stream straight in MFMA A-fragment order (lane = row,
// half rows on the two 32-lane halves: the row statistics are 80 in-lane adds and one exchange with lane ^ 32), normalises
// them in registers, writes the bf16 plane the backward's weight-gradient GEMM reads (+ mean / rstd, + the injected
// residual) and keeps the fp16 A fragments of the whole K extent (DP / 16 x 4 VGPRs) for the rest of the kernel. It then
// walks over ALL column tiles of the weight (128 columns each, [128][DP] fp16 double-buffered in LDS), so the activation is
// read from HBM once instead of once per column tile and never written / re-read as an fp16 plane, and the LayerNorm launch
// disappears. Against ln_fwd + gemm_nt on the default V1T (112 images, 185 k rows): the QKV GEMM fetched 59 MB x 12 column
// tiles + 614 KB of weights x 1448 row tiles through L2 - per CU that is its ~11.7 B / cycle fetch limit (outstanding misses x
// latency), i.e. the old kernel was bound by its re-reads; here a workgroup fetches 160 KB of x and 614 KB of W.
// Epilogues: EPI_BF16 (QKV) and EPI_BIAS_GELU (FC1; gelu' in the fragment order of 128-row tiles, as gemm_nt writes it).
template <int DP, int EPI, int NW, int NBLK, int RPW = 1>
__global__ __launch_bounds__(64 * NW, 2) void ln_gemm_kernel(LnFwdArgs l, GemmNTArgs g, int nsplit) {
    // RPW = 32-row sets per wave: set u of wave w covers rows m0 + 32 NW u + 32 w ..., so a workgroup spans RPW consecutive (32 NW)-row tiles
    constexpr int KS = DP / 16, BN = 32 * NBLK, LS = DP + 8, NTH = 64 * NW, KC = DP / 8, BM = 32 * NW * RPW, BMS = 32 * NW;
    static_assert(RPW == 1 || NW == 4, "row sets are whole 128-row tiles");
    constexpr int B_CHUNKS = BN * KC, B_ITERS = (B_CHUNKS + NTH - 1) / NTH;
    constexpr int CS = BN + 8;
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][BN * LS];
    __shared__ __attribute__((aligned(16))) bf16_t stg[NW][32 * CS];
    __shared__ __attribute__((aligned(16))) float sgb[2][DP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r31 = lane & 31, h2 = lane >> 5;
    // nsplit > 1 (launches with few row tiles: a rank's share of a multi-GPU step): the column tiles of a row tile are dealt
    // over nsplit workgroups; each normalises the rows for itself, the first one writes the LayerNorm outputs
    const int sp = blockIdx.x % nsplit;
    const int m0 = (blockIdx.x / nsplit) * BM;
    const int ntn = g.N / BN;

    u32x4 rb[B_ITERS];
    auto gload = [&](int tn) {
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NTH * i, brow = c / KC, kc = c % KC;
            if (c < B_CHUNKS) rb[i] = *(const u32x4*)(g.B + (size_t)(tn * BN + brow) * g.ldb + 8 * kc);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NTH * i, brow = c / KC, kc = c % KC;
            if (c < B_CHUNKS) *(u32x4*)(&sB[buf][brow * LS + 8 * kc]) = rb[i];
        }
    };
    gload(sp);  // in flight during the LayerNorm
    if (tid < DP) {
        sgb[0][tid] = tid < l.D ? l.gamma[tid] : 0.f;
        sgb[1][tid] = tid < l.D ? l.beta[tid] : 0.f;
    }

    // ---- LayerNorm of this lane's half row (columns 16 ks + 8 h2 + e), one row set after the other
    bf16x8 afrag[RPW][KS];
    bool staged = false;
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
    if (u) {  // keep the row sets' prologues apart (their 80 fp32 values each would be live together)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    const int row = m0 + BMS * u + 32 * wave + r31;
    const bool rok = row < l.rows, wln = rok && sp == 0;
    const int rr = rok ? row : l.rows - 1;
    float xv[KS][8];
    {
        const float* xp = l.x + (size_t)rr * DP + 8 * h2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 a = *(const f32x4*)(xp + 16 * ks), b = *(const f32x4*)(xp + 16 * ks + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[ks][e] = a[e];
                xv[ks][4 + e] = b[e];
            }
        }
        if (l.inject) {
            const float* ip = l.inject + (size_t)(rr / l.T) * DP + 8 * h2;
            float* op = l.xout + (size_t)row * DP + 8 * h2;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f32x4 a = *(const f32x4*)(ip + 16 * ks), b = *(const f32x4*)(ip + 16 * ks + 4);
                f32x4 oa, ob;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    oa[e] = xv[ks][e] += a[e];
                    ob[e] = xv[ks][4 + e] += b[e];
                }
                if (wln) {
                    *(f32x4*)(op + 16 * ks) = oa;
                    *(f32x4*)(op + 16 * ks + 4) = ob;
                }
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (ks >= KS - 2 && 16 * ks + 8 * h2 + e >= l.D) xv[ks][e] = 0.f;  // pad columns (D > DP - 32) stay out of the statistics
            sum += xv[ks][e];
        }
    sum += __shfl_xor(sum, 32);
    const float mean = sum / l.D;
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (ks >= KS - 2 && 16 * ks + 8 * h2 + e >= l.D) ? 0.f : xv[ks][e] - mean;
            q += d * d;
        }
    q += __shfl_xor(q, 32);
    const float rstd = rsqrtf(q / l.D + l.eps);
    const bool wz = wln && !l.lean;  // the LayerNorm plane and its statistics: read by the backward only
    if (wz && h2 == 0) {
        l.mean[row] = mean;
        l.rstd[row] = rstd;
    }
    if (!staged) __syncthreads();  // gamma / beta staged
    staged = true;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c0 = 16 * ks + 8 * h2;
        const f32x4 g0 = *(const f32x4*)&sgb[0][c0], g1 = *(const f32x4*)&sgb[0][c0 + 4];
        const f32x4 b0 = *(const f32x4*)&sgb[1][c0], b1 = *(const f32x4*)&sgb[1][c0 + 4];
        bf16x8 zh;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cc = c0 + e;
            const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
            float z = (xv[ks][e] - mean) * rstd * ga + be;
            if (ks >= KS - 2 && cc >= l.D) z = (cc == l.ones_col) ? 1.f : 0.f;
            zh[e] = (bf16_t)z;
            afrag[u][ks][e] = aux_plane(z, zh[e], 1);
        }
        if (wz) store_saved((bf16x8*)(l.z + (size_t)row * DP + c0), zh);
    }
    }  // row sets

    // ---- all column tiles of the weight against the resident A fragments
    const int wv = wave & 3;  // the epilogue helpers think in 128-row tiles of 4 waves (NW = 4 or 8)
    bf16_t* st = stg[wave];
    auto staged_store = [&](int u, f32x16 (&acc)[NBLK], bf16_t* dst, int ld, int n0, auto conv) {
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[acc_row(r, lane) * CS + 32 * nb + r31] = conv(acc[nb][r]);
        constexpr int CPR = BN / 8;
#pragma unroll
        for (int c0 = 0; c0 < 32 * CPR; c0 += 64) {
            const int c = c0 + lane, crow = c / CPR, ch = c % CPR;
            const int grow = m0 + BMS * u + 32 * wave + crow;
            if (grow < g.M) *(u32x4*)(dst + (size_t)grow * ld + n0 + 8 * ch) = *(const u32x4*)(st + crow * CS + 8 * ch);
        }
    };
    swrite(0);
    __syncthreads();
    const int boff = r31 * LS + 8 * h2;
    for (int tn = sp, it = 0; tn < ntn; tn += nsplit, ++it) {
        const int buf = it & 1, n0 = tn * BN;
        if (tn + nsplit < ntn) gload(tn + nsplit);
        f32x16 acc[RPW][NBLK];
#pragma unroll
        for (int u = 0; u < RPW; ++u)
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][nb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const bf16x8 b = *(const bf16x8*)(&sB[buf][32 * nb * LS + boff + 16 * ks]);  // one LDS read feeds every row set
#pragma unroll
                for (int u = 0; u < RPW; ++u) acc[u][nb] = mfma32h(afrag[u][ks], b, acc[u][nb]);
            }
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            if (u) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            const int m0p = m0 + BMS * u + 128 * (wave >> 2);
            if (m0p >= g.M) {
                // a 128-row tile wholly beyond M (second half of the last workgroup): nothing to store, and the gelu' fragment buffer
                // is only allocated for ceil(M / 128) tiles
            } else if constexpr (EPI == EPI_BF16) {
                staged_store(u, acc[u], (bf16_t*)g.C, g.ldc, n0, [](float v) { return (bf16_t)v; });
            } else {
                f32x16 resv[NBLK];
                gemm_epilogue<NBLK, EPI_BIAS_GELU, true>(g, acc[u], resv, m0p, n0, wv, lane);  // gelu' written (fragment order), activation left in acc
                if (g.C2) staged_store(u, acc[u], g.C2, g.ldc2, n0, [](float v) { return (bf16_t)v; });
                if (g.C2_lo) staged_store(u, acc[u], g.C2_lo, g.ldc2, n0, [](float v) { return aux_plane(v, (bf16_t)v, 1); });
            }
        }
        if (tn + nsplit < ntn) swrite(buf ^ 1);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// The whole MLP branch forward in one launch (vit.py:132-154 behind LN2, :356-358): x_out = x + row_scale * dropout(FC2(dropout(GELU(FC1(LN(x)))))).
// ln_gemm_kernel's FC1 form (4 waves, 128 rows, 64-column weight tiles, the LayerNorm'd rows resident as A fragments) with FC2 folded into its
// column-tile loop: the fp16 activation tile a wave stages in LDS for its 16-B row stores (the plane the backward's dW2 GEMM reads) IS the A operand
// of FC2's K-chunk [64 tn, 64 tn + 64) - the wave reads it back as fragments (row = lane, 8 consecutive k) and accumulates x W2^T over the column
// tiles in 5 more accumulators, against the W2 chunk [DP][64] staged beside the W1 tile. After the last tile the accumulators are FC2's (same
// layout and same K order as gemm_nt's) and leave through its proj / FC2 epilogue (epi_bias_res). The separate FC2 launch, its read of the
// activation plane (190 MB per 112-image launch) and its fill / drain disappear; everything the backward reads is written as before.
// NBLK = 1 (32-column tiles): double-buffered weight tiles 2 x (10.7 + 12.8) KB + the staging 10 KB = 57 KB, two workgroups per CU as
// ln_gemm_kernel has, 24 staging registers, 256 VGPRs. (NBLK = 2: 107 KB, one workgroup per CU: slower at every launch size measured.)
// Per 112-image step against the two launches: 20.88 / 20.94 against 21.02 / 21.02 ms; a 28-image share 6.06 against 6.14; C5 19.68 against 20.04.
template <int DP, int NBLK>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(LnFwdArgs l, GemmNTArgs g, GemmNTArgs g2) {
    constexpr int NW = 4, NB2 = DP / 32;
    constexpr int KS = DP / 16, BN = 32 * NBLK, LS = DP + 8, NTH = 64 * NW, KC = DP / 8, BM = 32 * NW;
    constexpr int CS = BN + 8, L2S = BN + 8, KC2 = BN / 8;
    constexpr int B_CHUNKS = BN * KC, B_ITERS = (B_CHUNKS + NTH - 1) / NTH;
    constexpr int W_CHUNKS = DP * KC2, W_ITERS = (W_CHUNKS + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][BN * LS];
    __shared__ __attribute__((aligned(16))) bf16_t sW2[2][DP * L2S];
    __shared__ __attribute__((aligned(16))) bf16_t stg[NW][32 * CS];
    __shared__ __attribute__((aligned(16))) float sgb[2][DP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r31 = lane & 31, h2 = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int ntn = g.N / BN;

    u32x4 rb[B_ITERS], rw[W_ITERS];
    auto gload = [&](int tn) {
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NTH * i, brow = c / KC, kc = c % KC;
            if (c < B_CHUNKS) rb[i] = *(const u32x4*)(g.B + (size_t)(tn * BN + brow) * g.ldb + 8 * kc);
        }
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int c = tid + NTH * i, n = c / KC2, kc = c % KC2;
            if (c < W_CHUNKS) rw[i] = *(const u32x4*)(g2.B + (size_t)n * g2.ldb + tn * BN + 8 * kc);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NTH * i, brow = c / KC, kc = c % KC;
            if (c < B_CHUNKS) *(u32x4*)(&sB[buf][brow * LS + 8 * kc]) = rb[i];
        }
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int c = tid + NTH * i, n = c / KC2, kc = c % KC2;
            if (c < W_CHUNKS) *(u32x4*)(&sW2[buf][n * L2S + 8 * kc]) = rw[i];
        }
    };
    gload(0);  // in flight during the LayerNorm
    if (tid < DP) {
        sgb[0][tid] = tid < l.D ? l.gamma[tid] : 0.f;
        sgb[1][tid] = tid < l.D ? l.beta[tid] : 0.f;
    }

    // ---- LayerNorm of this lane's half row (columns 16 ks + 8 h2 + e): ln_gemm_kernel's prologue (no injection in front of LN2)
    bf16x8 afrag[KS];
    {
        const int row = m0 + 32 * wave + r31;
        const bool rok = row < l.rows;
        const int rr = rok ? row : l.rows - 1;
        float xv[KS][8];
        const float* xp = l.x + (size_t)rr * DP + 8 * h2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 a = *(const f32x4*)(xp + 16 * ks), b = *(const f32x4*)(xp + 16 * ks + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[ks][e] = a[e];
                xv[ks][4 + e] = b[e];
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (ks >= KS - 2 && 16 * ks + 8 * h2 + e >= l.D) xv[ks][e] = 0.f;  // pad columns (D > DP - 32) stay out of the statistics
                sum += xv[ks][e];
            }
        sum += __shfl_xor(sum, 32);
        const float mean = sum / l.D;
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (ks >= KS - 2 && 16 * ks + 8 * h2 + e >= l.D) ? 0.f : xv[ks][e] - mean;
                q += d * d;
            }
        q += __shfl_xor(q, 32);
        const float rstd = rsqrtf(q / l.D + l.eps);
        const bool wz = rok && !l.lean;  // the LayerNorm plane and its statistics: read by the backward only
        if (wz && h2 == 0) {
            l.mean[row] = mean;
            l.rstd[row] = rstd;
        }
        __syncthreads();  // gamma / beta staged
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = 16 * ks + 8 * h2;
            const f32x4 g0 = *(const f32x4*)&sgb[0][c0], g1 = *(const f32x4*)&sgb[0][c0 + 4];
            const f32x4 b0 = *(const f32x4*)&sgb[1][c0], b1 = *(const f32x4*)&sgb[1][c0 + 4];
            bf16x8 zh;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int cc = c0 + e;
                const float ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
                float z = (xv[ks][e] - mean) * rstd * ga + be;
                if (ks >= KS - 2 && cc >= l.D) z = (cc == l.ones_col) ? 1.f : 0.f;
                zh[e] = (bf16_t)z;
                afrag[ks][e] = aux_plane(z, zh[e], 1);
            }
            if (wz) store_saved((bf16x8*)(l.z + (size_t)row * DP + c0), zh);
        }
    }

    bf16_t* st = stg[wave];
    auto staged_store = [&](f32x16 (&acc)[NBLK], bf16_t* dst, int ld, int n0, auto conv) {
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[acc_row(r, lane) * CS + 32 * nb + r31] = conv(acc[nb][r]);
        constexpr int CPR = BN / 8;
#pragma unroll
        for (int c0 = 0; c0 < 32 * CPR; c0 += 64) {
            const int c = c0 + lane, crow = c / CPR, ch = c % CPR;
            const int grow = m0 + 32 * wave + crow;
            if (grow < g.M) store_saved((u32x4*)(dst + (size_t)grow * ld + n0 + 8 * ch), *(const u32x4*)(st + crow * CS + 8 * ch));  // read by the backward only
        }
    };
    f32x16 acc2[NB2];
#pragma unroll
    for (int d = 0; d < NB2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[d][r] = 0.f;
    swrite(0);
    __syncthreads();
    const int boff = r31 * LS + 8 * h2, woff = r31 * L2S + 8 * h2, aoff = r31 * CS + 8 * h2;
    for (int tn = 0; tn < ntn; ++tn) {
        const int n0 = tn * BN, buf = tn & 1;
        if (tn + 1 < ntn) gload(tn + 1);
        {
            f32x16 acc[NBLK];
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mfma32h(afrag[ks], *(const bf16x8*)(&sB[buf][32 * nb * LS + boff + 16 * ks]), acc[nb]);
            f32x16 resv[NBLK];
            gemm_epilogue<NBLK, EPI_BIAS_GELU, true>(g, acc, resv, m0, n0, wave, lane);  // gelu' written (fragment order), activation left in acc
            if (g.C2) staged_store(acc, g.C2, g.ldc2, n0, [](float v) { return (bf16_t)v; });
            staged_store(acc, g.C2_lo, g.ldc2, n0, [](float v) { return aux_plane(v, (bf16_t)v, 1); });  // last: the staging now holds the fp16 tile
        }
        // ---- FC2 over this K-chunk: A = the staged tile (this wave's own region: its LDS operations execute in order), B = the W2 chunk
#pragma unroll
        for (int ks2 = 0; ks2 < BN / 16; ++ks2) {
            const bf16x8 a2 = *(const bf16x8*)(st + aoff + 16 * ks2);
#pragma unroll
            for (int d = 0; d < NB2; ++d) acc2[d] = mfma32h(a2, *(const bf16x8*)(&sW2[buf][32 * d * L2S + woff + 16 * ks2]), acc2[d]);
        }
        if (tn + 1 < ntn) swrite(buf ^ 1);
        __syncthreads();
    }
    epi_bias_res<NB2, BM>(g2, acc2, m0, 0, wave, lane);
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

// gemm_tn2: the production weight-gradient kernel. Wave tile = (32*YB) output rows x (32*XB) output
// columns with YB*XB = 5 accumulator blocks; two shapes:
//   <1,5>: 4 waves stacked over Y -> workgroup 128 x 160 (dWqkv, dW1: wide Y, X = one padded LN row)
//   <5,1>: 4 waves side by side over X -> workgroup 160 x 128 (dWo, dW2: Y is the 160-wide residual grad)
// so no wave idles on a 160-wide operand. 64 contraction rows per stage, staged by LDS-DMA into two
// [64][160] images per operand (row stride 320 B = 64 B x 5: conflict-free transposed reads), double
// buffered; rows past the end of the chunk are clamped for the fetch and masked in the 1-block operand.
constexpr int TN2_ROWS = 64, TN2_STR = 160, TN2_CPR = TN2_STR / 8;

// NST (round 6): stages of the operand ring. 2 = double buffer (rounds 1-5): ONE 41-KB stage in flight per workgroup while it computes, and the
// 82 KB of LDS allow one workgroup per CU anyway - at the small launches of the per-mouse loop or of a rank's share (20-40 stages per workgroup,
// ~250 workgroups) a stage then costs its DMA latency + transfer. 3 = two stages in flight behind a counted vmcnt (123 KB: one workgroup per CU - the
// double buffer's 81 920 B fit a CU's 160 KB exactly TWICE, which is what full-size launches want; the launcher picks by workgroup count).
template <int YB, int XB, bool XF16 = false, int NST = 3>
__global__ __launch_bounds__(256, (NST == 2 ? 2 : 1)) void gemm_tn2_kernel(GemmTNArgs g) {
    constexpr int WY = (YB == 1) ? 4 : 1, WX = 4 / WY;
    constexpr int YW = 32 * YB * WY, XW = 32 * XB * WX;
    static_assert(YW <= TN2_STR && XW <= TN2_STR, "tile");
    constexpr int NI = TN2_ROWS * TN2_CPR / 64 / 4;  // DMA instructions per wave per operand tile (5)
    __shared__ __attribute__((aligned(16))) bf16_t sY[NST][TN2_ROWS * TN2_STR];
    __shared__ __attribute__((aligned(16))) bf16_t sX[NST][TN2_ROWS * TN2_STR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = (WY == 4) ? wave : 0, wx = (WX == 4) ? wave : 0;
    // XCD-aware tile order: the hardware deals workgroups round-robin over the 8 XCDs in linear (x, y, z) order; after the remap the tiles of
    // one m-chunk - which all read the same rows of the narrow operand - are neighbours inside ONE XCD's share and find those rows in its L2
    // (dWo: the 59 MB of dy were fetched once per X tile, 5 times)
#ifndef V1T_TN2_NOREMAP
    const int nwg = gridDim.x * gridDim.y * gridDim.z;
    const int lid = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nwg);
    const int bx = lid % gridDim.x, by = (lid / gridDim.x) % gridDim.y, bz = lid / (gridDim.x * gridDim.y);
#else
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#endif
    const int n0 = bx * YW, x0 = by * XW;
    const int mb = bz * g.m_chunk;
    const int me = min(g.M, mb + g.m_chunk);
    const int nt = (me - mb + TN2_ROWS - 1) / TN2_ROWS;
    const int yw0 = n0 + 32 * YB * wy, xw0 = x0 + 32 * XB * wx;
    const bool wave_on = yw0 < g.NY && xw0 < g.NX;  // wave-uniform (launcher guarantees whole wave tiles)

    int drow[NI], dyc[NI], dxc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = 64 * (wave + 4 * i) + lane;
        drow[i] = c / TN2_CPR;
        const int cc = c % TN2_CPR;
        dyc[i] = min(n0 + 8 * min(cc, YW / 8 - 1), g.NY - 8);
        dxc[i] = min(x0 + 8 * min(cc, XW / 8 - 1), g.NX - 8);
    }
    // Round 5: whole tiles go through TileDma (common.h) - one instruction per 1-KB piece with a scalar base and loop-invariant lane offsets,
    // four pieces per M0 set-up - instead of one lds_dma16 per piece (64-bit address arithmetic + an M0 save / set / restore each: ~10 x 90 issue
    // cycles per stage and wave against 20 MFMAs = 640). Same LDS image ([64][160], pieces in row order), same row clamp (rows past the
    // m-chunk re-read its last row and are masked in the one-block operand). Tiles that overhang the matrix keep the per-piece form.
    using DmaY = TileDma<YW, TN2_STR, TN2_ROWS, 4>;
    using DmaX = TileDma<XW, TN2_STR, TN2_ROWS, 4>;
    static_assert(DmaY::LDS_ELEMS == TN2_ROWS * TN2_STR && DmaY::NINST == 4 * NI, "the image the per-piece form fills");
#ifdef V1T_TN2_NO_TILEDMA  // A/B builds (V1T_BUILD_LIB)
    const bool whole = false;
#else
    const bool whole = n0 + YW <= g.NY && x0 + XW <= g.NX;  // workgroup-uniform
#endif
    DmaY dmaY;
    DmaX dmaX;
    const bool tn_nt = g.nt != 0;  // kernel-uniform
    dmaY.init(lane, wave, g.ldy);
    dmaX.init(lane, wave, g.ldx);
    auto issue = [&](int t, int buf) {
        const int mt = mb + TN2_ROWS * t;
        if (whole) {
            // the operand whose tile columns belong to THIS workgroup alone is read exactly once (<1, 5>: Y - dqkv / dh; <5, 1>: X - the attention
            // output / activation plane): non-temporal. The narrow operand is re-read by every column tile through the L2 and keeps the default
            // policy. V1T_TN2_NT=0 (dev, A/B): both plain, as in rounds 1-5.
            if (tn_nt && YB == 1) dmaY.template issue<true, true>(g.Y + n0, mt, me, sY[buf]);
            else dmaY.issue(g.Y + n0, mt, me, sY[buf]);
            if (tn_nt && YB != 1) dmaX.template issue<true, true>(g.X + x0, mt, me, sX[buf]);
            else dmaX.issue(g.X + x0, mt, me, sX[buf]);
            return;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const size_t m = (size_t)min(mt + drow[i], me - 1);
            lds_dma16(g.Y + m * g.ldy + dyc[i], &sY[buf][512 * (wave + 4 * i)]);
            lds_dma16(g.X + m * g.ldx + dxc[i], &sX[buf][512 * (wave + 4 * i)]);
        }
    };

    f32x16 acc[YB * XB];
#pragma unroll
    for (int i = 0; i < YB * XB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    auto compute = [&](auto ragged_c, int buf, int valid) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
        for (int s = 0; s < TN2_ROWS / 16; ++s) {
            bf16x8 a[YB], b[XB];
#pragma unroll
            for (int yb = 0; yb < YB; ++yb) a[yb] = lds_tr_frag_nat(sY[buf], TN2_STR, 16 * s, 32 * (YB * wy + yb), lane);
#pragma unroll
            for (int xb = 0; xb < XB; ++xb) b[xb] = lds_tr_frag_nat(sX[buf], TN2_STR, 16 * s, 32 * (XB * wx + xb), lane);
            if constexpr (RAGGED) {
                const int k0 = 16 * s + 8 * (lane >> 5);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k0 + j >= valid) {
                        if constexpr (YB == 1) a[0][j] = 0; else b[0][j] = 0;
                    }
            }
            if constexpr (XF16) {
#pragma unroll
                for (int xb = 0; xb < XB; ++xb) b[xb] = f16_frag_to_bf16(b[xb]);
            }
#pragma unroll
            for (int yb = 0; yb < YB; ++yb)
#pragma unroll
                for (int xb = 0; xb < XB; ++xb) acc[yb * XB + xb] = mfma32(a[yb], b[xb], acc[yb * XB + xb]);
        }
    };

    if constexpr (NST == 2) {
        if (nt > 0) issue(0, 0);
        dma_wait_and_barrier();
        for (int t = 0; t < nt - 1; ++t) {
            const int buf = t & 1;
            issue(t + 1, buf ^ 1);
            if (wave_on) compute(std::false_type{}, buf, TN2_ROWS);
            dma_wait_and_barrier();
        }
    } else {
        // every wave issues exactly 2 NI DMA operations per stage (both forms of `issue`), so "all but the newest stage have landed" is one
        // counted vmcnt; stage t + 2 goes into the buffer stage t - 1 left behind the previous barrier
        static_assert(DmaY::PW == NI && DmaX::PW == NI && DmaX::NINST == 4 * NI, "2 NI operations per stage and wave");
        if (nt > 0) issue(0, 0);
        if (nt > 1) issue(1, 1);
        if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int t = 0; t < nt - 1; ++t) {
            const int nb = buf + 2 >= NST ? buf + 2 - NST : buf + 2;
            if (t + 2 < nt) issue(t + 2, nb);
            if (wave_on) compute(std::false_type{}, buf, TN2_ROWS);
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf = buf + 1 == NST ? 0 : buf + 1;
        }
    }
    if (!wave_on || nt <= 0) return;
    {
        const int valid = me - mb - TN2_ROWS * (nt - 1);
        const int lb = (nt - 1) % NST;
        if (valid < TN2_ROWS) compute(std::true_type{}, lb, valid);
        else compute(std::false_type{}, lb, TN2_ROWS);
    }
    if (g.slab) {
        // partial tile -> slab in accumulator-fragment order (16 floats per lane per block, plain 16-B stores);
        // tn_reduce_kernel sums the chunks and applies the index maps. ~3x cheaper than 41 MB of fp32 atomics.
        float* sl = g.slab + ((((size_t)bz * gridDim.y + by) * gridDim.x + bx) * 4 + wave) * (YB * XB * 1024) + lane * 16;
#pragma unroll
        for (int i = 0; i < YB * XB; ++i)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *(f32x4*)(sl + i * 1024 + 4 * r4) = f32x4{acc[i][4 * r4], acc[i][4 * r4 + 1], acc[i][4 * r4 + 2], acc[i][4 * r4 + 3]};
        return;
    }
#pragma unroll
    for (int yb = 0; yb < YB; ++yb)
#pragma unroll
        for (int xb = 0; xb < XB; ++xb) {
            const f32x16& c = acc[yb * XB + xb];
            const int xc = xw0 + 32 * xb + (lane & 31);
            const int xs = xc / g.xseg_pad, xr = xc % g.xseg_pad;
            const bool xok = xr < g.xseg_valid;
            const bool isb = g.dbias != nullptr && xc == g.ones_col;
            const int ncol = xs * g.xseg_valid + xr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int yr = yw0 + 32 * yb + acc_row(r, lane);
                const int ys = yr / g.yseg_pad, yy = yr % g.yseg_pad;
                if (yy < g.yseg_valid) {
                    if (xok) atomicAdd(&g.dW[(size_t)(ys * g.yseg_valid + yy) * g.ldw + ncol], c[r] * g.alpha);
                    else if (isb) atomicAdd(&g.dbias[ys * g.yseg_valid + yy], c[r] * g.alpha);
                }
            }
        }
}


// Sums the partial tiles gemm_tn2 left in the slab over the m-chunks and adds them into dW / dbias (plain
// read-modify-write: exactly one thread owns each output element). Thread = 4 accumulator registers of one
// lane of one block: consecutive threads read consecutive 16 B of each chunk's slab.
template <int YB, int XB>
DEVFN void tn_reduce_body(const GemmTNArgs& g, int gx, int gy, int gz, int bx_) {
    constexpr int WY = (YB == 1) ? 4 : 1, WX = 4 / WY;
    constexpr int YW = 32 * YB * WY, XW = 32 * XB * WX;
    const int t = bx_ * 256 + threadIdx.x;
    const int total = gx * gy * 4 * YB * XB * 256;
    if (t >= total) return;
    const int r4 = t & 3, lane = (t >> 2) & 63, rest = t >> 8;
    const int blk = rest % (YB * XB), wave = (rest / (YB * XB)) & 3, tile = rest / (YB * XB * 4);
    const int bx = tile % gx, by = tile / gx;
    const int wy = (WY == 4) ? wave : 0, wx = (WX == 4) ? wave : 0;
    const int yw0 = bx * YW + 32 * YB * wy, xw0 = by * XW + 32 * XB * wx;
    if (yw0 >= g.NY || xw0 >= g.NX) return;
    const size_t zs = (size_t)total * 4;
    const float* p = g.slab + (size_t)t * 4;
    f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    int z = 0;
    for (; z + 4 <= gz; z += 4) {
        const f32x4 a = *(const f32x4*)(p + (size_t)z * zs), b = *(const f32x4*)(p + (size_t)(z + 1) * zs);
        const f32x4 c = *(const f32x4*)(p + (size_t)(z + 2) * zs), d = *(const f32x4*)(p + (size_t)(z + 3) * zs);
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; z < gz; ++z) s0 += *(const f32x4*)(p + (size_t)z * zs);
    s0 += s1; s2 += s3; s0 += s2;
    const int yb = blk / XB, xb = blk % XB;
    const int xc = xw0 + 32 * xb + (lane & 31);
    const int xs = xc / g.xseg_pad, xr = xc % g.xseg_pad;
    const bool xok = xr < g.xseg_valid;
    const bool isb = g.dbias != nullptr && xc == g.ones_col;
    const int ncol = xs * g.xseg_valid + xr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yr = yw0 + 32 * yb + acc_row(4 * r4 + i, lane);
        const int ys = yr / g.yseg_pad, yy = yr % g.yseg_pad;
        if (yy < g.yseg_valid) {
            if (xok) g.dW[(size_t)(ys * g.yseg_valid + yy) * g.ldw + ncol] += s0[i] * g.alpha;
            else if (isb) g.dbias[ys * g.yseg_valid + yy] += s0[i] * g.alpha;
        }
    }
}
template <int YB, int XB>
__global__ __launch_bounds__(256) void tn_reduce_kernel(GemmTNArgs g, int gx, int gy, int gz) {
    tn_reduce_body<YB, XB>(g, gx, gy, gz, blockIdx.x);
}
// The slab reductions of several weight-gradient GEMMs in ONE launch (round 5): a group of a block's GEMMs runs back to back on the second
// stream (api.hip, flush_dw), so their reductions can follow as one launch instead of one behind every GEMM - 17 -> 5 launches per backward.
constexpr int TN_MULTI = 4;
struct TnReduceMulti {
    GemmTNArgs g[TN_MULTI];
    int gx[TN_MULTI], gy[TN_MULTI], gz[TN_MULTI], shape[TN_MULTI];  // shape 0: <5, 1>, 1: <1, 5>
    int start[TN_MULTI + 1];                                        // prefix sums of the units' workgroup counts
    int n;
};
__global__ __launch_bounds__(256) void tn_reduce_multi_kernel(TnReduceMulti m) {
    int u = 0;
    while (u + 1 < m.n && (int)blockIdx.x >= m.start[u + 1]) ++u;  // wave-uniform
    const int bx = blockIdx.x - m.start[u];
    if (m.shape[u] == 0) tn_reduce_body<5, 1>(m.g[u], m.gx[u], m.gy[u], m.gz[u], bx);
    else tn_reduce_body<1, 5>(m.g[u], m.gx[u], m.gy[u], m.gz[u], bx);
}


// ------------------------------------------------------------------------------------------
// dX GEMM of a branch's first linear layer fused with the backward of the LayerNorm in front of it (round 3): dz = A . B^T never goes to
// HBM (118 MB written and read back per LayerNorm and 112-image step at the default shape). A 128-row x DP tile holds complete rows, so
// everything ln_bwd_kernel (elementwise.hip) does happens on the accumulators: row statistics over the 32 lanes of a half-wave x NBLK
// blocks (DPP butterfly), G = gin + rstd (dz gamma - mean(dz gamma) - xhat mean(dz gamma xhat)), the column partials for dgamma / dbeta /
// the injection gradient / the next branch's bias (one in-lane sum over a lane's 16 rows per block, then half-waves, waves through LDS,
// one atomic per column and workgroup) and the dropout-backward / bf16 cast of G for the next (earlier) branch, staged through LDS into
// 16-B row stores. Four rows at a time (one accumulator register group): x and gin are read with the accumulator layout's 4-B accesses
// (128 B per half-wave and row), 20 + 20 loads in flight per group, so the live set stays at the accumulators + one row group.
// Tiles are image-aligned (grid = row blocks of an image x images) because the injection gradient is per image.
DEVFN float half32_sum(float v) {  // sum over the 32 lanes of a half-wave, returned to every lane
    v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);  // row_mirror: every lane holds the sum of its 16-lane row
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // rows 0 <-> 1, 2 <-> 3
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// ---- LayerNorm backward on a workgroup's accumulators dz (BM rows x 32 NBLK columns: column = 32 nb + (lane & 31), row = 32 wave + acc_row(r, lane)):
// the tail of gemm_lnbwd_kernel (tiles aligned to an image: row0 = the image's first row, t0 = the tile's first row inside it, Tlim = T, b = image)
// and of mlp_bwd_kernel (global 128-row tiles: row0 = 0, t0 = the tile's first row, Tlim = all rows, no per-image terms). `smem`: >= BM x CS bf16 of
// LDS that every wave is done with.
template <int NBLK, bool NEXT, int BM, int CS>
DEVFN void lnbwd_epilogue(const LnBwdArgs& l, f32x16 (&acc)[NBLK], size_t row0, int t0, int Tlim, int b, bf16_t* smem, int tid, int wave, int lane) {
    constexpr int BN = 32 * NBLK, NW = BM / 32;
    // ---- LayerNorm backward on the accumulators: column = 32 nb + (lane & 31), row = 32 wave + acc_row(r, lane)
    const float snext = l.scale_next ? l.scale_next[b] : 1.f;
    const float invD = 1.0f / (float)l.D;
    float gam[NBLK], adg[NBLK], adb[NBLK], ainj[NBLK], abn[NBLK];
    bool cok[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        const int col = 32 * nb + (lane & 31);
        cok[nb] = col < l.D;
        gam[nb] = cok[nb] ? l.gamma[col] : 0.f;
        adg[nb] = adb[nb] = ainj[nb] = abn[nb] = 0.f;
    }
    if (t0 + BM > Tlim) {  // ragged last tile of the image (workgroup-uniform): its clamped rows contribute nothing
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool rok = t0 + 32 * wave + acc_row(r, lane) < Tlim;
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[nb][r] = rok ? acc[nb][r] : 0.f;
        }
    }
    bf16_t* st = smem + wave * 32 * CS;
    constexpr int RPG = 4;  // accumulator rows per group: 8 NBLK loads in flight per group
    // dropout of the next branch without a branch: threshold 0 keeps everything (hash >= 0), factor 1
    const uint32_t dthr = (NEXT && l.drop_next.thresh) ? l.drop_next.thresh : 0u;
    const float dinv = (NEXT && l.drop_next.thresh) ? l.drop_next.inv_keep : 1.0f;
    // the image's rows as buffers: rows past the image read 0 (x, gin, mean, rstd = 0: G = 0 there) and their stores are dropped
    const uint32_t img_bytes = (uint32_t)Tlim * (uint32_t)l.DP * 4u;
    const __amdgpu_buffer_rsrc_t x_r = buf_rsrc(l.x + row0 * l.DP, img_bytes), gin_r = buf_rsrc(l.gin + row0 * l.DP, img_bytes);
    const __amdgpu_buffer_rsrc_t gout_r = buf_rsrc(l.gout + row0 * l.DP, img_bytes);
    const __amdgpu_buffer_rsrc_t mean_r = buf_rsrc(l.mean + row0, (uint32_t)Tlim * 4u), rstd_r = buf_rsrc(l.rstd + row0, (uint32_t)Tlim * 4u);
#pragma unroll
    for (int rg = 0; rg < 16 / RPG; ++rg) {
        // one row group at a time: without the fence the scheduler hoists every group's 40 loads to the top
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float mean[RPG], rstd[RPG], xh[NBLK][RPG], gi[NBLK][RPG];
        uint32_t trow[RPG], eoff[RPG];  // row inside the image, byte offset of (row, lane & 31) from the image's first row
#pragma unroll
        for (int i = 0; i < RPG; ++i) {
            trow[i] = (uint32_t)(t0 + 32 * wave + acc_row(RPG * rg + i, lane));
            eoff[i] = (trow[i] * (uint32_t)l.DP + (uint32_t)(lane & 31)) * 4u;
            if (trow[i] >= (uint32_t)Tlim) eoff[i] |= BUF_OOB;
            mean[i] = buf_load_f32(mean_r, trow[i] * 4u, 0);
            rstd[i] = buf_load_f32(rstd_r, trow[i] * 4u, 0);
        }
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int i = 0; i < RPG; ++i) {
                xh[nb][i] = buf_load_f32(x_r, eoff[i], 128 * nb);
                gi[nb][i] = buf_load_f32(gin_r, eoff[i], 128 * nb);
            }
        float s1[RPG], s2[RPG];
#pragma unroll
        for (int i = 0; i < RPG; ++i) {
            s1[i] = s2[i] = 0.f;
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const float xc = (xh[nb][i] - mean[i]) * rstd[i];
                xh[nb][i] = cok[nb] ? xc : 0.f;
                const float dy = acc[nb][RPG * rg + i] * gam[nb];  // pad columns: gamma = 0
                s1[i] += dy;
                s2[i] = fmaf(dy, xh[nb][i], s2[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < RPG; ++i) {
            s1[i] = half32_sum(s1[i]) * invD;
            s2[i] = half32_sum(s2[i]) * invD;
        }
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const int col = 32 * nb + (lane & 31);
#pragma unroll
            for (int i = 0; i < RPG; ++i) {
                const float dz = acc[nb][RPG * rg + i];
                const float gfull = fmaf(rstd[i], fmaf(-xh[nb][i], s2[i], dz * gam[nb] - s1[i]), gi[nb][i]);
                const float go = cok[nb] ? gfull : 0.f;  // rows past the image: gi = rstd = 0
                adg[nb] = fmaf(dz, xh[nb][i], adg[nb]);
                adb[nb] += dz;
                ainj[nb] += go;
                buf_store_f32(gout_r, eoff[i], 128 * nb, go);
                if constexpr (NEXT) {
                    const bool keep = drop_hash(l.drop_next.key, (uint32_t)row0 + trow[i], col) >= dthr;
                    const bf16_t vb = (bf16_t)(keep ? go * snext * dinv : 0.f);
                    abn[nb] += (float)vb;
                    st[acc_row(RPG * rg + i, lane) * CS + col] = vb;
                }
            }
        }
        // the column sums must be formed HERE: left alone the compiler sinks all 4 x 16 x NBLK additions behind the loop and keeps
        // (spills) every G and dy value until then
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) asm volatile("" : "+v"(adg[nb]), "+v"(adb[nb]), "+v"(ainj[nb]), "+v"(abn[nb]));
    }
    if constexpr (NEXT) {  // wave-private staging -> 16-B chunks of consecutive row segments
        constexpr int CPR = BN / 8;
#pragma unroll
        for (int c0 = 0; c0 < 32 * CPR; c0 += 64) {
            const int c = c0 + lane;
            if (c < 32 * CPR) {
                const int row = c / CPR, ch = c % CPR;
                const int t = t0 + 32 * wave + row;
                if (t < Tlim) *(u32x4*)(l.dy_next + (row0 + t) * l.DP + 8 * ch) = *(const u32x4*)(st + row * CS + 8 * ch);
            }
        }
    }
    // column partials: the two half-waves hold different rows of the same columns; each wave parks its 4 x BN sums in its own staging
    // region (its read-back above is complete: LDS operations of a wave execute in order)
    float* sred_w = (float*)(smem + wave * 32 * CS);
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        float q[4] = {adg[nb], adb[nb], ainj[nb], abn[nb]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned u = __float_as_uint(q[k]);
            const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            q[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            if (lane < 32) sred_w[k * BN + 32 * nb + lane] = q[k];
        }
    }
    __syncthreads();
    const int c = tid;
    if (c < l.D) {
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float* sw = (const float*)(smem + w * 32 * CS);
            r0 += sw[c];
            r1 += sw[BN + c];
            r2 += sw[2 * BN + c];
            r3 += sw[3 * BN + c];
        }
        atomicAdd(&l.dgamma[c], r0);
        atomicAdd(&l.dbeta[c], r1);
        if (l.dinject) atomicAdd(&l.dinject[(size_t)b * l.DP + c], r2);
        if (NEXT && l.dbias_next) atomicAdd(&l.dbias_next[c], r3);
    }
}

template <int NBLK, bool NEXT, int RING = 0>
__global__ __launch_bounds__(256, RING ? 1 : 2) void gemm_lnbwd_kernel(GemmNTArgs g, LnBwdArgs l) {
    constexpr int NW = 4, BK = 32, BM = 32 * NW, NT = 64 * NW;
    constexpr int LS = BK + 8, KC = BK / 8;
    constexpr int A_ITERS = BM * KC / NT;
    constexpr int BN = 32 * NBLK;
    constexpr int B_CHUNKS = BN * KC;
    constexpr int B_ITERS = (B_CHUNKS + NT - 1) / NT;
    constexpr int CS = BN + 8;  // staging row stride of the bf16 output (elements)
    constexpr int SMEM_DB = (2 * BM * LS + 2 * BN * LS) > (BM * CS) ? (2 * BM * LS + 2 * BN * LS) : (BM * CS);
    constexpr int SMEM = RING ? RING * ring_stage_elems<NBLK>() : SMEM_DB;  // RING: the LDS-DMA ring K loop (launches of at most one workgroup per CU)
    static_assert(!RING || RING * ring_stage_elems<NBLK>() >= BM * CS, "the epilogue's staging fits the ring");
    __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM];
    static_assert(4 * BN * 4 <= 32 * CS * 2, "column partials fit the wave's staging region");
    bf16_t (*sA)[BM * LS] = (bf16_t (*)[BM * LS])smem;
    bf16_t (*sB)[BN * LS] = (bf16_t (*)[BN * LS])(smem + 2 * BM * LS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntb = (l.T + BM - 1) / BM;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / ntb, t0 = (lid % ntb) * BM;
    const size_t row0 = (size_t)b * l.T;
    const int nk = g.K / BK;

    // two K tiles of operand loads in flight (register sets 0 / 1, tile kt in set kt & 1), one tile in LDS ahead of the MFMAs: a workgroup's K loop is a
    // chain of HBM round trips (64 B of each of its 128 rows per tile), and a launch with fewer workgroups than the chip holds has nothing else to cover them
    u32x4 ra[2][A_ITERS], rb[2][B_ITERS];
    auto gload = [&](auto set, int kt) {
        constexpr int S = decltype(set)::value;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int c = tid + NT * i, row = c / KC, kc = c % KC;
            ra[S][i] = *(const u32x4*)(g.A + (row0 + min(t0 + row, l.T - 1)) * g.lda + k0 + 8 * kc);  // rows past the image: clamped, masked below
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {  // no branch around a load: the compiler's vmcnt counts stay exact (lanes past the tile repeat its last chunk)
            const int c = min(tid + NT * i, B_CHUNKS - 1), row = c / KC, kc = c % KC;
            rb[S][i] = *(const u32x4*)(g.B + (size_t)row * g.ldb + k0 + 8 * kc);
        }
    };
    auto swrite = [&](auto set, int buf) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int c = tid + NT * i, row = c / KC, kc = c % KC;
            *(u32x4*)(&sA[buf][row * LS + 8 * kc]) = ra[S][i];
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NT * i, row = c / KC, kc = c % KC;
            if (c < B_CHUNKS) *(u32x4*)(&sB[buf][row * LS + 8 * kc]) = rb[S][i];
        }
    };
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  if constexpr (RING) {
    // rows past the image: clamped for the fetch, masked below
    ring_kloop<NBLK, RING, false>(g.A, g.lda, g.B, g.ldb, g.K / RING_BK, smem, wave, lane,
                                  [&](int i) { return (int)row0 + min(t0 + 8 * wave + 32 * i + (lane >> 3), l.T - 1); }, acc);
  } else {
    gload(set0{}, 0);
    gload(set1{}, min(1, nk - 1));
    swrite(set0{}, 0);
    __syncthreads();
    const int frag_off = (lane & 31) * LS + 8 * (lane >> 5);
    auto ktile = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const bf16x8 a = *(const bf16x8*)(&sA[buf][32 * wave * LS + frag_off + 16 * ks]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const bf16x8 bb = *(const bf16x8*)(&sB[buf][32 * nb * LS + frag_off + 16 * ks]);
                acc[nb] = mfma32(a, bb, acc[nb]);
            }
        }
    };
    // tile kt sits in LDS buffer kt & 1, tile kt + 1 is in flight into set (kt + 1) & 1; past the last tile the loads repeat it (no branch in the loop)
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        gload(set0{}, kt + 2);
        ktile(0);
        swrite(set1{}, 1);
        __syncthreads();
        gload(set1{}, min(kt + 3, nk - 1));
        ktile(1);
        swrite(set0{}, 0);
        __syncthreads();
    }
    if (kt + 1 < nk) {  // an even count: tiles nk - 2 (buffer 0) and nk - 1 (set 1)
        ktile(0);
        swrite(set1{}, 1);
        __syncthreads();
        ktile(1);
    } else {
        ktile(0);
    }
    __syncthreads();  // every wave is done with the operand tiles: the bf16 staging below reuses them
  }

    lnbwd_epilogue<NBLK, NEXT, BM, CS>(l, acc, row0, t0, l.T, b, smem, tid, wave, lane);
}

#ifdef V1T_EXPERIMENTS  // round-5 experiment 26 (V1T_MLP_BWD_FUSE; measured neutral): experiment builds only
// ------------------------------------------------------------------------------------------
// The MLP branch BACKWARD down to the residual-stream gradient in one launch - mlp_fwd_kernel's mirror: dhpre = (dy W2) * gelu' * mask (the dGELU
// GEMM, K = DP: the dy rows resident as A fragments, the hidden units walked in 32-column tiles) with dz = dhpre W1 folded into the tile loop - the
// bf16 dhpre tile a wave stages in LDS for its row stores (the plane the dW1 GEMM reads) is the A operand of the K-chunk [32 tn, 32 tn + 32) of
// the second GEMM, accumulated in NB2 blocks against the W1^T chunk staged beside the W2^T tile - and the LayerNorm backward of LN2 on those
// accumulators at the end (lnbwd_epilogue: gemm_lnbwd_kernel's tail, here over global 128-row tiles - LN2 has no per-image term). Against
// gemm_nt<EPI_DGELU> + gemm_lnbwd_kernel: one launch, dhpre written (for dW1) but not read back (190 MB per 112-image launch).
template <int DP>
__global__ __launch_bounds__(256, 2) void mlp_bwd_kernel(GemmNTArgs g, GemmNTArgs g2, LnBwdArgs l) {
    constexpr int NW = 4, NB2 = DP / 32, KS = DP / 16, BN = 32, LS = DP + 8, NTH = 64 * NW, KC = DP / 8, BM = 32 * NW;
    constexpr int CS = BN + 8, L2S = BN + 8, KC2 = BN / 8, ECS = DP + 8;
    constexpr int B_CHUNKS = BN * KC, B_ITERS = (B_CHUNKS + NTH - 1) / NTH;
    constexpr int W_CHUNKS = DP * KC2, W_ITERS = (W_CHUNKS + NTH - 1) / NTH;
    constexpr int SB = BN * LS, SW = DP * L2S, SS = 32 * CS;
    constexpr int SMEM = (2 * SB + 2 * SW + NW * SS) > BM * ECS ? (2 * SB + 2 * SW + NW * SS) : BM * ECS;
    __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM];
    bf16_t* sB = smem;                 // [2][BN x LS]: W2^T tile (hidden unit rows, K = DP)
    bf16_t* sW = smem + 2 * SB;        // [2][DP x L2S]: W1^T chunk (input-feature rows, K-chunk of 32 hidden units)
    bf16_t* stg = smem + 2 * SB + 2 * SW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r31 = lane & 31, h2 = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int ntn = g.N / BN;

    u32x4 rb[B_ITERS], rw[W_ITERS];
    auto gload = [&](int tn) {
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NTH * i, brow = c / KC, kc = c % KC;
            if (c < B_CHUNKS) rb[i] = *(const u32x4*)(g.B + (size_t)(tn * BN + brow) * g.ldb + 8 * kc);
        }
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int c = tid + NTH * i, n = c / KC2, kc = c % KC2;
            if (c < W_CHUNKS) rw[i] = *(const u32x4*)(g2.B + (size_t)n * g2.ldb + tn * BN + 8 * kc);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int c = tid + NTH * i, brow = c / KC, kc = c % KC;
            if (c < B_CHUNKS) *(u32x4*)(&sB[buf * SB + brow * LS + 8 * kc]) = rb[i];
        }
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int c = tid + NTH * i, n = c / KC2, kc = c % KC2;
            if (c < W_CHUNKS) *(u32x4*)(&sW[buf * SW + n * L2S + 8 * kc]) = rw[i];
        }
    };
    gload(0);
    // the dy rows of this wave as A fragments of the whole K extent (lane = row, half rows on the two 32-lane halves; rows past M: the last row,
    // masked where they leave the workgroup)
    bf16x8 afrag[KS];
    {
        const int row = min(m0 + 32 * wave + r31, g.M - 1);
        const bf16_t* ap = g.A + (size_t)row * g.lda + 8 * h2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afrag[ks] = *(const bf16x8*)(ap + 16 * ks);
    }
    bf16_t* st = stg + wave * SS;
    f32x16 acc2[NB2];
#pragma unroll
    for (int d = 0; d < NB2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[d][r] = 0.f;
    swrite(0);
    __syncthreads();
    const int boff = r31 * LS + 8 * h2, woff = r31 * L2S + 8 * h2, aoff = r31 * CS + 8 * h2;
    for (int tn = 0; tn < ntn; ++tn) {
        const int n0 = tn * BN, buf = tn & 1;
        if (tn + 1 < ntn) gload(tn + 1);
        {
            f32x16 acc[1], resv[1];
            {  // saved gelu' fragments of this (row tile, column block): 2 x 16 B per lane
                const bf16_t* fp = g.aux + frag_index(g, m0, n0, 0, wave, lane);
                const f32x4 lo = *(const f32x4*)fp, hi = *(const f32x4*)(fp + 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    resv[0][j] = lo[j];
                    resv[0][4 + j] = hi[j];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc[0] = mfma32(afrag[ks], *(const bf16x8*)(&sB[buf * SB + boff + 16 * ks]), acc[0]);
            gemm_epilogue<1, EPI_DGELU, true>(g, acc, resv, m0, n0, wave, lane);  // d(pre-activation), rounded to bf16, left in acc; bias column sums
            // staged [32 rows][32] per wave -> 16-B row chunks of the dhpre plane (the dW1 GEMM reads it); the staging is FC1-backward's A operand
#pragma unroll
            for (int r = 0; r < 16; ++r) st[acc_row(r, lane) * CS + r31] = (bf16_t)acc[0][r];
            constexpr int CPR = BN / 8;
#pragma unroll
            for (int c0 = 0; c0 < 32 * CPR; c0 += 64) {
                const int c = c0 + lane, crow = c / CPR, ch = c % CPR;
                const int grow = m0 + 32 * wave + crow;
                if (grow < g.M) *(u32x4*)((bf16_t*)g.C + (size_t)grow * g.ldc + n0 + 8 * ch) = *(const u32x4*)(st + crow * CS + 8 * ch);
            }
        }
#pragma unroll
        for (int ks2 = 0; ks2 < BN / 16; ++ks2) {
            const bf16x8 a2 = *(const bf16x8*)(st + aoff + 16 * ks2);
#pragma unroll
            for (int d = 0; d < NB2; ++d) acc2[d] = mfma32(a2, *(const bf16x8*)(&sW[buf * SW + 32 * d * L2S + woff + 16 * ks2]), acc2[d]);
        }
        if (tn + 1 < ntn) swrite(buf ^ 1);
        __syncthreads();
    }
    // (the last barrier of the loop: every wave is done with the tiles and its staging - the epilogue's staging overlays them)
    lnbwd_epilogue<NB2, true, BM, ECS>(l, acc2, 0, m0, g.M, 0, smem, tid, wave, lane);
}
#endif  // V1T_EXPERIMENTS

}  // namespace


int launch_ln_gemm(const LnFwdArgs& l, const GemmNTArgs& g, int epi, hipStream_t s) {
    if (!g.f16 || g.A_lo || g.B_lo || g.K != l.DP || g.N % 64 != 0 || (g.ldb % 8) || l.D <= l.DP - 32 || l.D > l.DP) return V1T_ERR_UNSUPPORTED;
    if (epi == EPI_BF16 ? (g.ldc % 8) != 0 : (epi != EPI_BIAS_GELU || (g.ldc2 % 8) != 0)) return V1T_ERR_UNSUPPORTED;
    if (g.M != l.rows || (l.inject && !l.xout)) return V1T_ERR_ARG;
    if (l.rows <= 0) return V1T_OK;
    // workgroups: row tiles x nsplit, one workgroup per CU (a rank of an 8-GPU step has 91 row tiles, the single-GPU batch 724)
    // two shapes: 8 waves x 256 rows x 128-column tiles, one workgroup per CU (LDS 157 KB), or 4 waves x 128 rows x 64-column
    // tiles, two independent workgroups per CU (62 KB each) whose MFMA / epilogue / store phases drift apart and overlap
    // 0: 8 waves x 256 rows x 128-column tiles; 1: 4 waves x 128 rows x 64 columns; 2: 4 waves x 2 row sets (256 rows) x 64 columns:
    // a weight tile is fetched and read from LDS once per 256 rows (its fetches through L2 cost the QKV launch 58 of 258 us with one
    // row set: ablation in profiles/r03_gemm_experiments.txt). QKV at full-size launches: 265 -> 250 us; not for the GELU epilogue
    // (its live values + the second row set's accumulators and A fragments do not fit 256 registers: 183 -> 283 us) nor for small
    // launches (fewer, longer workgroups)
    static const int force_shape = dev_env("V1T_LNG_SHAPE") ? atoi(dev_env("V1T_LNG_SHAPE")) : -1;  // dev switch
    const int shape = force_shape >= 0 ? force_shape : ((epi == EPI_BF16 && l.rows >= 512 * 256) ? 2 : 1);
    const int BMr = shape == 1 ? 128 : 256, BNr = shape ? 64 : 128;
    if (g.N % BNr != 0) return V1T_ERR_UNSUPPORTED;
    const int rt = (l.rows + BMr - 1) / BMr, ntn = g.N / BNr;
    static const int force_split = dev_env("V1T_LNG_SPLIT") ? atoi(dev_env("V1T_LNG_SPLIT")) : 0;  // dev switch
    // cost model in column-tile units: rounds of 256 workgroups x (LayerNorm prologue ~1.5 tiles + the workgroup's tiles); measured
    // on one rank's share of 2- / 4- / 8-GPU steps (362 / 181 / 91 row tiles -> 2 / 1 / 2)
    int nsplit = 1;
    float best = 1e30f;
    for (int ns = 1; ns <= ntn; ++ns) {
        const int slots = shape ? 512 : 256;  // resident workgroups
        const float unit = shape == 2 ? 2.0f : 1.0f;  // row sets per workgroup: prologue and column tiles cost twice
        const float cost = (float)((rt * ns + slots - 1) / slots) * unit * ((shape ? 3.0f : 1.5f) + (float)((ntn + ns - 1) / ns));
        if (cost < best - 1e-3f) { best = cost; nsplit = ns; }
    }
    // between one and two workgroups per CU the model's "one round" is lopsided (362 row tiles at a 4-GPU share: 106 CUs hold two workgroups
    // that each walk all 30 column tiles): two column halves per row tile measured 6.17 / 6.16 against 6.21 / 6.23 ms per step (round 5)
    if (shape == 1 && nsplit == 1 && rt > 256 && rt <= 512 && ntn >= 2) nsplit = 2;
    if (force_split > 0) nsplit = std::min(force_split, ntn);
    const dim3 grid(rt * nsplit), blk(shape ? 256 : 512);
#define LNG_CASE(DPV)                                                                                                       \
    case DPV:                                                                                                               \
        if (shape == 2) {                                                                                            \
            if (epi == EPI_BF16) hipLaunchKernelGGL((ln_gemm_kernel<DPV, EPI_BF16, 4, 2, 2>), grid, blk, 0, s, l, g, nsplit); \
            else hipLaunchKernelGGL((ln_gemm_kernel<DPV, EPI_BIAS_GELU, 4, 2, 2>), grid, blk, 0, s, l, g, nsplit);           \
        } else if (shape) {                                                                                                 \
            if (epi == EPI_BF16) hipLaunchKernelGGL((ln_gemm_kernel<DPV, EPI_BF16, 4, 2>), grid, blk, 0, s, l, g, nsplit);   \
            else hipLaunchKernelGGL((ln_gemm_kernel<DPV, EPI_BIAS_GELU, 4, 2>), grid, blk, 0, s, l, g, nsplit);              \
        } else {                                                                                                            \
            if (epi == EPI_BF16) hipLaunchKernelGGL((ln_gemm_kernel<DPV, EPI_BF16, 8, 4>), grid, blk, 0, s, l, g, nsplit);   \
            else hipLaunchKernelGGL((ln_gemm_kernel<DPV, EPI_BIAS_GELU, 8, 4>), grid, blk, 0, s, l, g, nsplit);              \
        }                                                                                                                   \
        break;
    switch (l.DP) {
        LNG_CASE(32) LNG_CASE(64) LNG_CASE(96) LNG_CASE(128) LNG_CASE(160)
        default: return V1T_ERR_UNSUPPORTED;
    }
#undef LNG_CASE
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

// LN2 -> FC1 (+ GELU, dropout) -> FC2 (+ bias, dropout, residual) as one launch (mlp_fwd_kernel). l / g as for launch_ln_gemm(EPI_BIAS_GELU), g2 as for
// launch_gemm_nt(EPI_BIAS_RES) with A = g's fp16 activation plane. V1T_ERR_UNSUPPORTED: use those two.
int launch_mlp_fwd(const LnFwdArgs& l, const GemmNTArgs& g, const GemmNTArgs& g2, hipStream_t s) {
    if (l.DP != 160 || g.K != l.DP || g.M != l.rows || !g.f16 || !g2.f16 || g.A_lo || g.B_lo || g2.A_lo || g2.B_lo || l.inject) return V1T_ERR_UNSUPPORTED;
    if (g.N % 128 != 0 || g2.K != g.N || g2.N != l.DP || g2.M != g.M || !g.C2_lo || g2.A != g.C2_lo || g2.lda != g.ldc2 || (g.ldb % 8) || (g2.ldb % 8)) return V1T_ERR_UNSUPPORTED;
    if (!g2.res || g2.res != l.x || g2.ldres != l.DP || !g2.C || g2.rd.o || g.rd.o) return V1T_ERR_UNSUPPORTED;
    if (g.M <= 0) return V1T_OK;
    // More than one workgroup per CU only: a launch of <= 256 row tiles (a 14-image share of an 8-GPU step, a 16-image launch of the per-mouse loop)
    // is faster as two kernels - ln_gemm deals its column tiles over two workgroups per row tile there and FC2 streams through the LDS-DMA ring
    // (sim 8: 3.55 fused against 3.47 ms; per-mouse loop 28.3 against 27.7; 28 images and up the fused launch wins: profiles/r05_small_launch_experiments.txt #24).
    // V1T_MLP_FUSE=2 (dev, A/B): fused at every size.
    static const bool always = dev_env("V1T_MLP_FUSE") && atoi(dev_env("V1T_MLP_FUSE")) == 2;
    const int tiles = (g.M + 127) / 128;
    if (tiles <= 256 && !always) return V1T_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((mlp_fwd_kernel<160, 1>), dim3(tiles), dim3(256), 0, s, l, g, g2);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

// dGELU GEMM (gd: launch_gemm_nt(EPI_DGELU) arguments) + dz GEMM (gz: A = gd's output plane) + LayerNorm backward (l: as launch_gemm_ln_bwd, with
// dy_next, without per-image terms) as one launch (mlp_bwd_kernel). V1T_ERR_UNSUPPORTED: use launch_gemm_nt + launch_gemm_ln_bwd.
int launch_mlp_bwd(const GemmNTArgs& gd, const GemmNTArgs& gz, const LnBwdArgs& l, hipStream_t s) {
    if (l.DP != 160 || gd.K != l.DP || gz.N != l.DP || gz.K != gd.N || gd.N % 32 != 0 || gd.M != gz.M || gd.M != l.B * l.T) return V1T_ERR_UNSUPPORTED;
    if (gd.A_lo || gd.B_lo || gz.A_lo || gz.B_lo || gd.f16 || gz.f16 || gd.rd.o || gz.rd.o || !gd.aux || !gd.C || gz.A != (const bf16_t*)gd.C || gz.lda != gd.ldc) return V1T_ERR_UNSUPPORTED;
    if ((gd.lda % 8) || (gd.ldb % 8) || (gz.ldb % 8) || (gd.ldc % 8) || l.dinject || l.scale_next || !l.dy_next || !l.gin || !l.gout || !l.dgamma || !l.dbeta || l.D > l.DP) return V1T_ERR_UNSUPPORTED;
    if (gd.M <= 0) return V1T_OK;
    // OFF by default: measured neutral (112 images: 20.70 / 20.88 / 20.80 fused against 20.79 / 20.80 / 20.63 ms per step; 28- and 14-image shares
    // 6.14 / 3.49 against 6.15 / 3.51: profiles/r05_small_launch_experiments.txt #26) - unlike the forward's fold, whose GELU stage hides FC2's
    // MFMAs, both halves here are bound by the same HBM traffic, and the weight-gradient GEMMs of the second stream fill whatever a shorter
    // chain leaves. V1T_MLP_BWD_FUSE=1: above 256 row tiles, 2: at every size (the equality test runs it).
#ifdef V1T_EXPERIMENTS
    static const int mode = dev_env("V1T_MLP_BWD_FUSE") ? atoi(dev_env("V1T_MLP_BWD_FUSE")) : 0;
    const int tiles = (gd.M + 127) / 128;
    if (mode == 0 || (mode != 2 && tiles <= 256)) return V1T_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((mlp_bwd_kernel<160>), dim3(tiles), dim3(256), 0, s, gd, gz, l);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
#else
    (void)s;
    return V1T_ERR_UNSUPPORTED;  // the product runs the two launches (the fused form lives in experiment builds)
#endif
}

int launch_gemm_ln_bwd(const GemmNTArgs& g, const LnBwdArgs& l, hipStream_t s) {
    if (g.N != l.DP || l.DP % 32 != 0 || l.DP > 160 || g.K % 32 != 0 || (g.lda % 8) || (g.ldb % 8) || g.A_lo || g.f16) return V1T_ERR_UNSUPPORTED;
    if (g.M != l.B * l.T || l.D > l.DP || !l.gin || !l.gout || !l.dgamma || !l.dbeta) return V1T_ERR_ARG;
    if (g.M <= 0) return V1T_OK;
    const dim3 grid(((l.T + 127) / 128) * l.B), blk(256);
#define V1T_GLB(NB)                                                                                       \
    case NB:                                                                                              \
        if (l.dy_next) hipLaunchKernelGGL((gemm_lnbwd_kernel<NB, true>), grid, blk, 0, s, g, l);          \
        else hipLaunchKernelGGL((gemm_lnbwd_kernel<NB, false>), grid, blk, 0, s, g, l);                   \
        break;
    // at most one workgroup per CU (a rank's share of a multi-GPU step, a 16-image launch of the per-mouse loop): the LDS-DMA ring K loop
    if (g_gemm_ring && l.DP == 160 && grid.x <= 256 && g.K % RING_BK == 0 && g.K / RING_BK >= 3) {
        if (l.dy_next) hipLaunchKernelGGL((gemm_lnbwd_kernel<5, true, RING_NS>), grid, blk, 0, s, g, l);
        else hipLaunchKernelGGL((gemm_lnbwd_kernel<5, false, RING_NS>), grid, blk, 0, s, g, l);
        return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    }
    switch (l.DP / 32) {
        V1T_GLB(1) V1T_GLB(2) V1T_GLB(3) V1T_GLB(4) V1T_GLB(5)
        default: return V1T_ERR_UNSUPPORTED;
    }
#undef V1T_GLB
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

bool gemm_nt_takes_row_dot(const GemmNTArgs& a, int epi) {
    return epi == EPI_BF16 && a.N % 160 == 0 && !a.f16 && !a.A_lo && !a.B_lo && (a.ldc % 8) == 0 && (a.rd.ldo % 8) == 0 && a.rd.H == a.N / 160;
}

int launch_gemm_nt(const GemmNTArgs& a, int epi, hipStream_t s) {
    if (a.K % 32 != 0 || a.N % 32 != 0 || (a.lda % 8) || (a.ldb % 8)) return V1T_ERR_ARG;
    if (a.rd.o && !gemm_nt_takes_row_dot(a, epi)) return V1T_ERR_UNSUPPORTED;
    if ((a.f16 || a.A_lo) && a.rd.o) return V1T_ERR_UNSUPPORTED;
    if (epi == EPI_BF16 && !a.A_lo && (a.ldc % 8)) return V1T_ERR_ARG;  // 16-B output chunks
    if (epi == EPI_BIAS_GELU && !a.A_lo && (a.ldc2 % 8)) return V1T_ERR_ARG;
    if (epi == EPI_DGELU && (a.ldc % 8)) return V1T_ERR_ARG;
    if (a.N % 160 == 0) return launch_nt_n<5>(a, epi, s);
    if (a.N % 128 == 0) return launch_nt_n<4>(a, epi, s);
    if (a.N % 64 == 0) return launch_nt_n<2>(a, epi, s);
    return launch_nt_n<1>(a, epi, s);
}

size_t gemm_tn_slab_bytes(int M, int NY, int NX, int m_chunk) {
    if (M <= 0 || m_chunk <= 0 || m_chunk % TN2_ROWS != 0) return 0;
    const size_t gz = (size_t)(M + m_chunk - 1) / m_chunk;
    if (NY % 160 == 0 && NX % 128 == 0 && NY <= 320) return gz * (NY / 160) * (NX / 128) * 4 * 5 * 1024 * sizeof(float);
    if (NX % 160 == 0) return gz * ((NY + 127) / 128) * (NX / 160) * 4 * 5 * 1024 * sizeof(float);
    return 0;
}

bool gemm_tn_takes_f16_x(int NY, int NX, int m_chunk) { return m_chunk % TN2_ROWS == 0 && NY % 160 == 0 && NX % 128 == 0 && NY <= 320; }

// `defer`: launch the GEMM only and describe its slab reduction in *defer (the caller runs launch_tn_reduce_group over several of them);
// GEMMs without a slab, or of a shape without one, ignore it.
static int launch_gemm_tn_impl(const GemmTNArgs& a_in, hipStream_t s, TnReduceMulti* defer) {
    static const bool tn_nt = !(dev_env("V1T_TN2_NT") && !atoi(dev_env("V1T_TN2_NT")));
    GemmTNArgs a = a_in;
    a.nt = tn_nt ? 1 : 0;
    if (a.NY % 32 != 0 || a.NX % 32 != 0 || a.m_chunk % 32 != 0 || (a.ldy % 8) || (a.ldx % 8)) return V1T_ERR_ARG;
    if (a.x_f16 && !gemm_tn_takes_f16_x(a.NY, a.NX, a.m_chunk)) return V1T_ERR_UNSUPPORTED;
    if (a.M <= 0) return V1T_OK;
    const int gz = (a.M + a.m_chunk - 1) / a.m_chunk;
    // the 3-stage ring (123 KB) is for launches of at most one workgroup per CU; above that the double buffer's 81 920 B let TWO workgroups share a
    // CU's 160 KB, which hides more latency than a deeper ring (112-image launch, alone: 113 / 110 us per GEMM against 174 / 136 with the ring -
    // profiles/r06_experiments.txt #4). V1T_TN2_NST=2 / 3 (dev, A/B) forces either.
    static const int tn2_force = dev_env("V1T_TN2_NST") ? atoi(dev_env("V1T_TN2_NST")) : 0;
    auto reduce = [&](int shape, int gx, int gy2) {
        if (!a.slab) return;
        if (defer && defer->n < TN_MULTI) {
            const int u = defer->n++;
            defer->g[u] = a; defer->gx[u] = gx; defer->gy[u] = gy2; defer->gz[u] = gz; defer->shape[u] = shape;
            defer->start[u + 1] = defer->start[u] + gx * gy2 * 20;
            return;
        }
        if (shape == 0) hipLaunchKernelGGL((tn_reduce_kernel<5, 1>), dim3(gx * gy2 * 20), dim3(256), 0, s, a, gx, gy2, gz);
        else hipLaunchKernelGGL((tn_reduce_kernel<1, 5>), dim3(gx * gy2 * 20), dim3(256), 0, s, a, gx, gy2, gz);
    };
    if (a.m_chunk % TN2_ROWS == 0 && a.NY % 160 == 0 && a.NX % 128 == 0 && a.NY <= 320) {
        const int gx = a.NY / 160, gy2 = a.NX / 128;
        const int tn2_nst = tn2_force ? tn2_force : (gx * gy2 * gz <= 256 ? 3 : 2);
        if (tn2_nst == 2) {
            if (a.x_f16) hipLaunchKernelGGL((gemm_tn2_kernel<5, 1, true, 2>), dim3(gx, gy2, gz), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_tn2_kernel<5, 1, false, 2>), dim3(gx, gy2, gz), dim3(256), 0, s, a);
        } else {
            if (a.x_f16) hipLaunchKernelGGL((gemm_tn2_kernel<5, 1, true>), dim3(gx, gy2, gz), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_tn2_kernel<5, 1>), dim3(gx, gy2, gz), dim3(256), 0, s, a);
        }
        reduce(0, gx, gy2);
        return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    }
    if (a.m_chunk % TN2_ROWS == 0 && a.NX % 160 == 0) {
        const int gx = (a.NY + 127) / 128, gy2 = a.NX / 160;
        const int tn2_nst = tn2_force ? tn2_force : (gx * gy2 * gz <= 256 ? 3 : 2);
        if (tn2_nst == 2) hipLaunchKernelGGL((gemm_tn2_kernel<1, 5, false, 2>), dim3(gx, gy2, gz), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_tn2_kernel<1, 5>), dim3(gx, gy2, gz), dim3(256), 0, s, a);
        reduce(1, gx, gy2);
        return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    }
    const int gy = (a.NY + 127) / 128;
#define TN_LAUNCH(XB)                                                                                       \
    hipLaunchKernelGGL((gemm_tn_kernel<XB>), dim3(gy, a.NX / (32 * XB), gz), dim3(256), 0, s, a)
    if (a.NX % 160 == 0) TN_LAUNCH(5);
    else if (a.NX % 128 == 0) TN_LAUNCH(4);
    else if (a.NX % 64 == 0) TN_LAUNCH(2);
    else TN_LAUNCH(1);
#undef TN_LAUNCH
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_gemm_tn(const GemmTNArgs& a, hipStream_t s) { return launch_gemm_tn_impl(a, s, nullptr); }
// n GEMMs (distinct slab regions) back to back on `s`, then ONE launch for all their slab reductions
int launch_gemm_tn_group(const GemmTNArgs* a, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += TN_MULTI) {
        TnReduceMulti m{};
        for (int i = i0; i < std::min(n, i0 + TN_MULTI); ++i) {
            const int rc = launch_gemm_tn_impl(a[i], s, &m);
            if (rc) return rc;
        }
        if (m.n > 0) {
            hipLaunchKernelGGL(tn_reduce_multi_kernel, dim3(m.start[m.n]), dim3(256), 0, s, m);
            if (hipGetLastError() != hipSuccess) return V1T_ERR_LAUNCH;
        }
    }
    return V1T_OK;
}
