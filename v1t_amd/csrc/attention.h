// v1t_amd — fused multi-head self-attention (flash-style, bf16 MFMA, fp32 online softmax), gfx950.
// Replaces Attention.scaled_dot_product_attention (reference vit.py:253-265): head dim = emb dim
// (vit.py:218), scores scaled by the `scale` buffer (or per-head LSA parameter with the diagonal
// masked, vit.py:235-261), softmax, dropout on P, P.V. The (B,H,T,T) score tensor is never
// materialised; the backward recomputes P from Q,K and the saved log2-sum-exp.
#pragma once
#include "common.h"

struct AttnArgs {
    const bf16_t* qkv; int ldqkv;  // [rows][3*H*DP]: q | k | v, each H heads of DP (zero padded) columns
    bf16_t* o; int ldo;            // [rows][H*DP]; forward: may be nullptr when only the second plane is wanted; backward: fp16 values when o_f16
    int o_f16;
    bf16_t* o_lo;                  // forward only: second plane of o or nullptr: bf16 residual o - float(bf16(o)), or fp16(o) when lo_f16
    int lo_f16;
    float* lse2;                   // [B][H][T]  log2-domain: max + log2(sum)
    int B, H, T;
    const float* scale;            // device: 1 value, or H values when scale_per_head
    int scale_per_head;
    int mask_diag;                 // LSA: exclude key == query
    AttnDrop adrop;                // P dropout (2x2-block hash, 8-bit rate; common.h)
    // forward only, set by its launcher: the queries of an (image, head) are covered by fwd_full 256-query row blocks followed by fwd_half
    // 128-query ones (half blocks: waves 4-7 hold no query and only stage K / V); fwd_lpt: full blocks dispatched before half blocks inside
    // each XCD's chunk of the grid
    int fwd_full, fwd_half, fwd_lpt;
    // backward only
    const bf16_t* dO; int lddo;    // [rows][H*DP]
    const float* delta;            // [B][H][T] keep_prob * rowsum(dO * O)
    bf16_t* dqkv; int lddqkv;      // [rows][3*H*DP]
    float* dscale;                 // [H] fp32 atomics or nullptr (LSA scale gradient)
    // optional scratch of attn_ds_bytes(): the dK/dV kernel writes the bf16 dS' = P (dP - delta) there and dQ = dS' . K becomes
    // one streaming GEMM instead of a second recomputation of S and dP (5 MFMA products instead of 7). Layout (attention.hip,
    // "producer / consumer backward"): [B*H][32-query block][32-key block][k-step 2][lane 64][8] bf16 - a 32 x 32 block is the
    // 2 KB the consumer wave holds as its two B-operand fragments (key on the lane), stored with two 16-B-per-lane instructions.
    // Behind it: the padded per-row constants [B*H][TPq] fp32 each, nlse = -lse2 (log2 domain: the dK/dV producers pre-multiply K by
    // scale log2 e, so S' = (c K) Q^T + nlse is the exponent itself) and ndelta = -keep_prob * rowsum(dO * O) (pad rows:
    // nlse = -1e30, so P = 0 there), written by attn_delta2_kernel.
    bf16_t* ds; int ldds;
    int ds_nt;  // dQ GEMM: read dS' with the non-temporal policy (set by the launcher)
};
__host__ __device__ inline int attn_ds_ld(int T) { return (T + 127) / 128 * 128; }             // key columns covered (128-key workgroups)
__host__ __device__ inline int attn_ds_tpq(int T) { return (T + 31) / 32 * 32; }               // query rows covered (32-query blocks)
__host__ __device__ inline size_t attn_ds_elems(int B, int H, int T) { return (size_t)B * H * attn_ds_tpq(T) * attn_ds_ld(T); }
__host__ __device__ inline size_t attn_rc_floats(int B, int H, int T) { return (size_t)B * H * attn_ds_tpq(T); }
inline size_t attn_ds_bytes(int B, int H, int T) { return attn_ds_elems(B, H, T) * 2 + 2 * attn_rc_floats(B, H, T) * 4; }

int launch_attn_fwd(const AttnArgs& a, int DP, hipStream_t s);
int launch_attn_delta(const AttnArgs& a, int DP, float* delta, hipStream_t s);
// pad rows (t >= T) of the row-constant arrays behind a.ds only (nlse = -1e30: P = 0 there; ndelta = 0): for callers that get the real rows
// from the dO GEMM's epilogue (gemm.h, RowDotArgs) instead of launch_attn_delta
int launch_attn_rc_pad(const AttnArgs& a, hipStream_t s);
int launch_attn_bwd(const AttnArgs& a, int DP, hipStream_t s);  // dq + dkv kernels

// attention rollout building blocks (reference utils/attention_rollout.py:92-122)
int launch_rollout_headmax(const AttnArgs& a, int DP, float* A, int TP, float* rowsum, int q_rows, hipStream_t s);  // q_rows > 0: the first q_rows query rows only
int launch_rollout_vecmat(const float* A, const float* rowsum, const float* v, float* u, int B, int T, int TP, hipStream_t s);
int launch_rollout_matmul(const float* A, const float* rowsum, const float* Xin, float* Xout, int B, int T, int TP, hipStream_t s);
