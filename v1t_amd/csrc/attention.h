// v1t_amd — fused multi-head self-attention (flash-style, bf16 MFMA, fp32 online softmax), gfx950.
// Replaces Attention.scaled_dot_product_attention (reference vit.py:253-265): head dim = emb dim
// (vit.py:218), scores scaled by the `scale` buffer (or per-head LSA parameter with the diagonal
// masked, vit.py:235-261), softmax, dropout on P, P.V. The (B,H,T,T) score tensor is never
// materialised; the backward recomputes P from Q,K and the saved log2-sum-exp.
#pragma once
#include "common.h"

struct AttnArgs {
    const bf16_t* qkv; int ldqkv;  // [rows][3*H*DP]: q | k | v, each H heads of DP (zero padded) columns
    bf16_t* o; int ldo;            // [rows][H*DP]
    bf16_t* o_lo;                  // forward only: second plane of o or nullptr: bf16 residual o - float(bf16(o)), or fp16(o) when lo_f16
    int lo_f16;
    float* lse2;                   // [B][H][T]  log2-domain: max + log2(sum)
    int B, H, T;
    const float* scale;            // device: 1 value, or H values when scale_per_head
    int scale_per_head;
    int mask_diag;                 // LSA: exclude key == query
    AttnDrop adrop;                // P dropout (2x2-block hash, 8-bit rate; common.h)
    // backward only
    const bf16_t* dO; int lddo;    // [rows][H*DP]
    const float* delta;            // [B][H][T] keep_prob * rowsum(dO * O)
    bf16_t* dqkv; int lddqkv;      // [rows][3*H*DP]
    float* dscale;                 // [H] fp32 atomics or nullptr (LSA scale gradient)
    // optional scratch of attn_ds_bytes(): the dK/dV kernel writes dS' = P (dP - delta) there, tile-major
    // [B*H][128-query tile][64-key tile][128][64] bf16 with ldds = attn_ds_ld(T) key columns, and dQ = dS' . K becomes one
    // streaming GEMM instead of a second recomputation of S and dP (5 products instead of 7)
    bf16_t* ds; int ldds;
};
inline int attn_ds_ld(int T) { return (T + 127) / 128 * 128; }
inline size_t attn_ds_bytes(int B, int H, int T) { return (size_t)B * H * ((T + 127) / 128 * 128) * attn_ds_ld(T) * 2; }

int launch_attn_fwd(const AttnArgs& a, int DP, hipStream_t s);
int launch_attn_delta(const AttnArgs& a, int DP, float* delta, hipStream_t s);
int launch_attn_bwd(const AttnArgs& a, int DP, hipStream_t s);  // dq + dkv kernels

// attention rollout building blocks (reference utils/attention_rollout.py:92-122)
int launch_rollout_headmax(const AttnArgs& a, int DP, float* A, int TP, float* rowsum, hipStream_t s);
int launch_rollout_vecmat(const float* A, const float* rowsum, const float* v, float* u, int B, int T, int TP, hipStream_t s);
