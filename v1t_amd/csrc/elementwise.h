// v1t_amd — HBM-bound kernels around the MFMA products (gfx950): weight packing, patch embedding,
// LayerNorm fwd/bwd fused with the BehaviorMLP injection and the next branch's dropout/cast,
// BehaviorMLP, fused L1 + AdamW, ELU1 + Poisson loss.
#pragma once
#include "common.h"

// One entry of the weight-shadow pack table: dst (bf16 or fp32, padded / optionally transposed)
// <- fp32 master (natural shape). Index maps: padded index i -> natural (i / pad) * valid + i % pad,
// zero where i % pad >= valid.
struct PackDesc {
    long long src_off;  // floats into the parameter arena
    long long dst_off;  // BYTES into the shadow arena
    int src_ld;         // natural leading dimension (columns of the natural matrix)
    int drows, dcols;   // padded dst shape
    int rseg_pad, rseg_valid, cseg_pad, cseg_valid;  // maps in SOURCE orientation (rows, cols of the natural matrix)
    int transpose;      // dst[r][c] = src[c][r]
    int out_f32;
    int lo_plane;       // 16-bit outputs: 1 = the residual bf16(v - float(bf16(v))) (split-bf16 low plane), 2 = fp16(v)
};
int launch_pack(const float* params, void* shadow, const PackDesc* d_desc, int ndesc, hipStream_t s);

struct PatchArgs {
    const float* img;  // [B][C][IH][IW]
    int B, C, IH, IW, P, stride, NH, NW;  // NH x NW patch grid; L = NH*NW; T = L + 1
    int D, DP;
    const float* W;    // [D][C*P*P]
    const float* bias; // [D]
    const float* cls;  // [D]
    const float* pos;  // [T][D]
    float* x;          // [B*T][DP] fp32 out (fwd) / grad in (bwd)
    DropCfg drop;
    // backward accumulators (fp32, natural shapes)
    float* dW; float* dbias; float* dcls; float* dpos;
};
// CCT conv tokenizer (reference core/cct.py:30-104): Conv2d(C -> D, kernel P, stride, zero padding `pad`, no bias) as unfold + MFMA
// GEMM, then ReLU -> MaxPool2d(kernel 3, stride 2, padding 1) -> "b c h w -> b (h w) c" -> + position table -> dropout.
struct ConvTokArgs {
    const float* img;        // [B][C][IH][IW]
    int B, C, IH, IW, P, stride, pad;
    int CH, CW;              // conv output grid
    int NH, NW;              // pooled grid (tokens per image = NH * NW)
    int D, DP;
    const float* conv;       // [B*CH*CW][DP] fp32 conv output (pool forward)
    const float* pos;        // [NH*NW][D] or nullptr
    float* x;                // forward: tokens out [B*NH*NW][DP] (pad columns 0); backward: gradient wrt the tokens (in)
    unsigned char* idx;      // [B*NH*NW][DP]: window element 0..8 (3 * dy + dx) that holds the maximum, 255 = ReLU inactive (no gradient)
    DropCfg drop;
};
// U[(b*CH + cy)*CW + cx][j] = img[b][c][cy*stride - pad + kh][cx*stride - pad + kw] (0 outside the image), j = (c*P + kh)*P + kw;
// columns >= C*P*P are zero. hi (bf16) and optional second plane (fp16 / bf16 residual) like launch_patch_unfold.
int launch_conv_unfold(const ConvTokArgs& a, bf16_t* u_hi, bf16_t* u_lo, int lo_f16, int ldu, hipStream_t s);
int launch_cct_pool_fwd(const ConvTokArgs& a, hipStream_t s);
// gd[(b*CH + cy)*CW + cx][d] (bf16) = sum over the pool windows whose maximum sits at (cy, cx): dropout_bwd(a.x[b][window][d])
int launch_cct_pool_bwd(const ConvTokArgs& a, bf16_t* gd, hipStream_t s);

int launch_patch_embed_fwd(const PatchArgs& a, hipStream_t s);
int launch_patch_embed_bwd(const PatchArgs& a, hipStream_t s);
// MFMA form of the patch embedding (C*P*P % 32 == 0): the unfolded patches as a bf16 matrix U [B*T][ldu]
// (row b*T is the class token: zeros; columns >= C*P*P: a 1 in column C*P*P of every patch row, then zeros - the
// ones column hands the bias gradient to the weight-gradient GEMM), hi and optional lo plane.
int launch_patch_unfold(const PatchArgs& a, bf16_t* u_hi, bf16_t* u_lo, int lo_f16, int ldu, hipStream_t s);
// backward prologue: dpos / dcls sums over the batch and gd = bf16(dropout_bwd(g)) [B*T][DP] for the dW GEMM
int launch_patch_bwd_pos_cast(const PatchArgs& a, bf16_t* gd, hipStream_t s);
// Patch modes 2 / 3 (vit.py:83-100): the unfolded patches in fp32 (they go through a LayerNorm over the patch before
// the projection). mode 2 = Shifted Patch Tokenization: channels = image + its 4 diagonal shifts by P/2 with zero
// padding (PatchShifting vit.py:15-38, single-channel input), patch width (C+4)*P*P. Class-token rows and pad columns 0.
int launch_patch_unfold_f32(const PatchArgs& a, int spt, float* u, int ldu, hipStream_t s);
// same backward prologue for modes 2 / 3: gd rows of the class token are ZERO (the projection never sees that row) and
// the masked gradient is also kept in fp32 (gdf, may be NULL) for the LayerNorm behind the projection (mode 3)
int launch_patch_bwd_pos_cast_nocls(const PatchArgs& a, bf16_t* gd, float* gdf, hipStream_t s);
// mode 3 epilogue: x0 = dropout(LN(y) * gamma + beta + pos) for patch rows, dropout(cls + pos[0]) for class-token rows
struct PatchLn2Args {
    const float* y; float* x0; float* mean; float* rstd;
    const float* gamma; const float* beta; const float* pos; const float* cls;
    int rows, T, D, DP; float eps; DropCfg drop;
};
int launch_patch_ln2_finish(const PatchLn2Args& a, hipStream_t s);
// LayerNorm parameter gradients only: dgamma[c] += sum_r dz[r][c] * xhat[r][c], dbeta[c] += sum_r dz[r][c]
int launch_ln_param_grad(const float* dz, int lddz, const float* x, int ldx, const float* mean, const float* rstd, int rows, int D,
                         float* dgamma, float* dbeta, hipStream_t s);

// Gradient with respect to the core input (analysis path; the reference gets it from autograd through vit.py:66-72, 122-129):
// dU = g . W in fp32 (g: fp32 rows through `drop`, class-token rows zero when cls; or bf16 rows gb), the input gradient of the
// LayerNorm over the patch (modes 2 / 3), and col2im back onto the image (stride, zero padding, SPT's shifted channels).
int launch_patch_du(const float* gf, const bf16_t* gb, int ldg, DropCfg drop, int T, int cls, const float* W, int D, int PD, long long rows, float* du,
                    hipStream_t s);
int launch_patch_ln_bwd_rows(const float* dz, int lddz, const float* u, int ldu, const float* mean, const float* rstd, const float* gamma, long long rows,
                             int PD, float* out, hipStream_t s);
int launch_patch_col2im(const float* du, int PD, int B, int C, int IH, int IW, int P, int stride, int pad, int GH, int GW, int rows_per_image, int row0,
                        int spt, float* dx, hipStream_t s);
int launch_resize_bilinear_bwd(const float* dout, float* din, int planes, int IH, int IW, int OH, int OW, hipStream_t s);

struct LnFwdArgs {
    const float* x;      // [rows][DP]
    const float* inject; // [B][DP] or nullptr: x += inject[b] (written to xout)
    float* xout;         // [rows][DP] (may alias x when inject == nullptr -> not written)
    const float* gamma; const float* beta;  // [D] natural
    bf16_t* z;           // [rows][DP]
    bf16_t* z_lo;        // [rows][DP] second plane or nullptr: bf16 residual z - float(bf16(z)), or fp16(z) when lo_f16
    int lo_f16;
    float* mean; float* rstd;  // [rows]
    int rows, T, D, DP;
    float eps;
    int ones_col;        // >= D: z[:, ones_col] = 1 (bias-gradient column for the weight-gradient GEMM), < 0: none
    int lean;            // inference: the fused LayerNorm + GEMM kernel keeps z in registers only (z / mean / rstd feed nothing but the backward)
};
int launch_ln_fwd(const LnFwdArgs& a, hipStream_t s);

struct LnBwdArgs {
    const float* dz;     // [rows][DP] grad wrt LN output (fp32)
    const float* x;      // LN input
    const float* mean; const float* rstd;
    const float* gamma;  // [D]
    const float* gin;    // [rows][DP] residual-stream grad flowing past the LN, or nullptr (= 0)
    float* gout;         // [rows][DP] = gin + dx
    float* dgamma; float* dbeta;  // [D] atomics
    float* dinject;      // [B][DP] atomics or nullptr: sum over the image's tokens of gout
    bf16_t* dy_next;     // [rows][DP] bf16 = dropout_bwd(gout) for the next (earlier) branch, or nullptr
    float* dbias_next;   // [D] atomics: column sums of dy_next, or nullptr
    DropCfg drop_next;
    const float* scale_next;  // [B] or nullptr: stochastic-depth factor of the next branch (multiplies dy_next)
    int B, T, D, DP;
};
int launch_ln_bwd(const LnBwdArgs& a, hipStream_t s);

struct CastArgs {
    const float* g;      // [rows][DP]
    bf16_t* dy;          // [rows][DP]
    float* dbias;        // [D] atomics or nullptr
    DropCfg drop;
    int rows, D, DP;
    const float* scale; int T;  // [rows / T] stochastic-depth factor of the branch, or nullptr
};
int launch_drop_cast(const CastArgs& a, hipStream_t s);

struct BmlpArgs {
    const float* v;      // [B][IN]
    int B, IN, J, D, DP; // J = D/2 hidden
    const float* W1; const float* b1;  // [J][IN], [J] (b may be nullptr)
    const float* W3; const float* b3;  // [D][J], [D]
    float* hid;          // [B][J]  tanh(hidden), saved
    float* out;          // [B][DP] tanh(out), saved (pad = 0)
    const float* dout;   // [B][DP]
    float* dW1; float* db1; float* dW3; float* db3;
};
constexpr int BMLP_MAX_BLOCKS = 16;
struct BmlpBatch {
    BmlpArgs blk[BMLP_MAX_BLOCKS];
    int n;
};
int launch_bmlp_fwd(const BmlpArgs& a, hipStream_t s);
int launch_bmlp_fwd_multi(const BmlpBatch& bb, hipStream_t s);  // all blocks' B-MLPs in one launch
int launch_bmlp_bwd(const BmlpBatch& bb, hipStream_t s);  // all blocks' B-MLPs in one launch

struct AdamArgs {
    float* p; float* g; float* m; float* v;
    long long n;
    float lr, beta1, beta2, eps, weight_decay;
    float bc1, bc2;      // 1 - beta^step
    float l1;            // adds l1 * sign(p) to the gradient (folded L1 regulariser), 0 = off
    int zero_grad;       // write 0 to g after use
};
int launch_adamw(const AdamArgs& a, hipStream_t s);
int launch_l1_sum(const float* p, long long n, float scale, float* out_accum, hipStream_t s);
int launch_l1_grad(const float* p, float* g, long long n, float scale, hipStream_t s);
int launch_l1_grad_dev(const float* p, float* g, long long n, float scale, const float* gscale, hipStream_t s);

struct LossArgs {
    const float* u;      // [B][N] readout pre-activation
    const float* y;      // [B][N] targets
    float* yhat;         // [B][N] or nullptr
    float* du;           // [B][N] or nullptr: dLoss/du * gscale
    float* loss;         // scalar accumulator (atomic) or nullptr
    float* loss_total;   // second accumulator shared by several units (atomic) or nullptr
    long long n;
    float loss_scale;    // sqrt(ds_size / batch)
    float gscale;        // upstream gradient of the loss (1 for plain training)
};
int launch_elu1_poisson(const LossArgs& a, hipStream_t s);
int launch_poisson_loss(const float* yp, const float* yt, long long n, float eps, float scale, float* dy, float* loss, hipStream_t s);
int launch_elu1_bwd(const float* u, const float* y, const float* g, long long n, float* du, hipStream_t s);
constexpr int LOSS_MAX_UNITS = 8, ADAM_MAX_RANGES = 24;
int launch_elu1_poisson_multi(const LossArgs* a, int n, hipStream_t s);  // n <= LOSS_MAX_UNITS units in one launch
int launch_adamw_multi(const AdamArgs* a, int n, hipStream_t s);         // any n: one launch per ADAM_MAX_RANGES pieces
int launch_fill_zero(void* p, long long bytes, hipStream_t s);

int launch_dropout_mask(uint8_t* out, long long rows, long long cols, DropCfg d, hipStream_t s);
int launch_attn_dropout_mask(uint8_t* out, long long rows, long long T, AttnDrop d, hipStream_t s);
int launch_crop_nearest(const float* in, int B, int C, int IH, int IW, const float* grid, const float* shifts, float* out, int OH, int OW,
                        hipStream_t s);
int launch_resize_bilinear(const float* in, float* out, int planes, int IH, int IW, int OH, int OW, hipStream_t s);
constexpr int INPUTS_MAX_UNITS = 8;
// every unit's images resized / copied into one batch buffer + cat(behaviors, pupil centres) rows, one launch per INPUTS_MAX_UNITS units
int launch_inputs_multi(const float* const* img, const float* const* beh, const float* const* pup, const int* n_images, int n, int C, int IH, int IW,
                        float* out, int OH, int OW, float* beh_out, int na, int nb, hipStream_t s);
