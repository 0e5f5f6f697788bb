// v1t_amd — bf16 MFMA GEMMs with fused epilogues (gfx950).
//  gemm_nt : C[M][N]  = A[M][K] . B[N][K]^T   (both operands K-contiguous; forward and dX GEMMs,
//            the dX form uses the transposed bf16 weight shadow so it is NT as well)
//  gemm_tn : dW[n][k] += sum_m Y[m][n] . X[m][k]   (weight gradients; both operands are read
//            transposed from LDS with ds_read_b64_tr_b16; fp32 atomics into the gradient arena)
#pragma once
#include "common.h"

enum GemmEpi {
    EPI_BF16 = 0,       // C bf16 = acc
    EPI_F32 = 1,        // C fp32 = acc
    EPI_BIAS_RES = 2,   // C fp32 = res + dropout(acc + bias)                       (proj, FC2)
    EPI_BIAS_GELU = 3,  // v = acc + bias; C bf16 = gelu'(v) in ACCUMULATOR-FRAGMENT order (below); C2 bf16 = dropout(gelu(v)) (FC1)
    EPI_DGELU = 4,      // C bf16 = acc * mask/keep * aux (aux = saved gelu', fragment order)          (dX of FC2)
    EPI_PATCH = 5,      // C fp32 = dropout(acc + bias + pos[row % T]); rows with row % T == 0 are the class token:
                        // dropout(cls + pos[0]) (their A rows are zero)                               (patch embedding)
    // Fragment order: the 16 accumulator values a lane holds for one 32x32 block are stored contiguously,
    // index (((tile_m * (N/32) + col_block) * 4 + wave) * 64 + lane) * 16 + reg. Producer (FC1) and consumer
    // (dX of FC2) share M, N and the 128-row tile, so both sides move 2 x 16 B per lane per block instead of 16
    // scattered 2-byte accesses (the 2-byte loads alone cost 34 us of that 86 us kernel).
};

// EPI_BF16 with 160-column tiles only: the dot product of every (row, column tile) of the bf16-ROUNDED output with the same segment of a second
// 16-bit matrix, i.e. the attention backward's row constants delta = rowsum(dO * O) per head while dO = dy . Wo leaves the GEMM (the separate
// pass re-read dO and O: 0.47 GB per 112-image launch). Written negated / scaled, next to the negated log-sum-exp, into the padded row-constant
// arrays the dK/dV kernel reads (attention.h): index (b * H + column tile) * TPQ + t for row b * T + t. Pad rows (t >= T) are not touched.
struct RowDotArgs {
    const bf16_t* o; int ldo; int o_f16;  // [M][N]: bf16, or fp16 when o_f16; nullptr = off
    const float* lse2;                    // [B][H][T]
    float* nlse; float* ndelta;           // [B * H][TPQ]
    int T, TPQ, H;
    float keep;                           // ndelta = -keep * dot
};

struct GemmNTArgs {
    const bf16_t* A; int lda;
    const bf16_t* B; int ldb;
    // split-bf16 ("bf16x3") mode when A_lo != nullptr: A = A + A_lo, B = B + B_lo (each a bf16 plane of the
    // same layout), acc += A.B + A_lo.B + A.B_lo  -> ~2^-17 relative operand precision at 3 MFMAs per step.
    // Used by the four forward linear layers, whose bf16 operand rounding otherwise dominates the error
    // of the predicted responses (DESIGN.md "Numerics").
    const bf16_t* A_lo; const bf16_t* B_lo;
    bf16_t* C2_lo;                // EPI_BIAS_GELU: second plane of the activation output (bf16 residual, or fp16 when f16)
    int f16;                      // A and B hold fp16 bit patterns: one fp16 MFMA per step (A_lo / B_lo unused)
    int M, N, K;  // M = valid rows (guarded); N % (32*NBLK) == 0; K % 32 == 0
    void* C; int ldc;
    const float* bias;            // [N] fp32 (padded with zeros) or nullptr
    const float* res; int ldres;  // fp32 residual
    bf16_t* C2; int ldc2;
    const bf16_t* aux; int ldaux;
    float* colsum; int n_valid;   // fp32 atomics, natural column index < n_valid
    DropCfg drop;
    const float* pos; const float* cls; int T;  // EPI_PATCH: natural [T][n_valid] position table, [n_valid] class token
    const float* row_scale;       // EPI_BIAS_RES: per-image factor on the branch (stochastic depth), index row / T, or nullptr
    RowDotArgs rd;                // EPI_BF16, N % 160 == 0: see RowDotArgs
    int lean;                     // EPI_BIAS_GELU, inference: gelu' (C, read by the backward only) is not written
};
bool gemm_nt_takes_row_dot(const GemmNTArgs& a, int epi);

struct GemmTNArgs {
    const bf16_t* Y; int ldy;  // [M][NY]
    const bf16_t* X; int ldx;  // [M][NX]
    int M, NY, NX;             // NY % 32 == 0, NX % (32*XBLK) == 0
    float* dW; int ldw;        // natural (unpadded) row-major fp32 matrix, atomicAdd
    int yseg_pad, yseg_valid;  // padded row index i -> natural (i / pad) * valid + i % pad, valid iff i % pad < valid
    int xseg_pad, xseg_valid;
    int m_chunk;               // contraction rows per workgroup (multiple of 32)
    float alpha;
    // bias gradient for free: if X carries a column of ones at padded index `ones_col` (the LayerNorm kernel
    // writes it into a pad column), output column ones_col = column sums of Y -> atomicAdd into dbias
    float* dbias; int ones_col;
    // optional fp32 scratch of gemm_tn_slab_bytes() bytes: partial tiles go there with plain stores and a
    // reduce kernel adds them into dW (float atomics run at ~1.3 TB/s chip-wide and bound this GEMM otherwise)
    float* slab;
    // X holds fp16 values (the forward's second plane of the attention output / GELU output): converted to bf16 fragment by fragment.
    // Only where gemm_tn_takes_f16_x() says so (the 160 x 128 workgroup shape, one X block per wave).
    int x_f16;
    int nt;  // set by the launcher: the once-read operand of gemm_tn2 through the non-temporal policy
};
bool gemm_tn_takes_f16_x(int NY, int NX, int m_chunk);
size_t gemm_tn_slab_bytes(int M, int NY, int NX, int m_chunk);

int launch_gemm_nt(const GemmNTArgs& a, int epi, hipStream_t s);
int launch_gemm_tn(const GemmTNArgs& a, hipStream_t s);
int launch_gemm_tn_group(const GemmTNArgs* a, int n, hipStream_t s);  // the GEMMs back to back, their slab reductions as ONE launch
// LayerNorm(l) fused into C = LN(x) . B^T (fp16 operands, K = l.DP <= 160, N % 128 == 0; epi = EPI_BF16 | EPI_BIAS_GELU): g.A is
// not read, l.z (bf16 plane), l.mean, l.rstd and l.xout are written, l.z_lo is not. V1T_ERR_UNSUPPORTED: use ln_fwd + gemm_nt.
struct LnFwdArgs;
int launch_ln_gemm(const LnFwdArgs& l, const GemmNTArgs& g, int epi, hipStream_t s);
// LN -> FC1 (EPI_BIAS_GELU) -> FC2 (EPI_BIAS_RES) in one launch; g2.A must be g's fp16 activation plane. V1T_ERR_UNSUPPORTED (other shapes, launches of
// at most 256 row tiles): launch the two.
int launch_mlp_fwd(const LnFwdArgs& l, const GemmNTArgs& g, const GemmNTArgs& g2, hipStream_t s);
// dz = A . B^T (bf16 operands, N = l.DP <= 160) consumed in registers by the LayerNorm backward `l` describes (l.dz is not read; same
// outputs as launch_gemm_nt(EPI_F32) + launch_ln_bwd). V1T_ERR_UNSUPPORTED: use those two.
struct LnBwdArgs;
int launch_gemm_ln_bwd(const GemmNTArgs& g, const LnBwdArgs& l, hipStream_t s);
// dGELU GEMM + dz GEMM + LayerNorm backward of the MLP branch in one launch (gz.A = gd.C). V1T_ERR_UNSUPPORTED: launch_gemm_nt(gd, EPI_DGELU) + launch_gemm_ln_bwd(gz, l).
int launch_mlp_bwd(const GemmNTArgs& gd, const GemmNTArgs& gz, const LnBwdArgs& l, hipStream_t s);
