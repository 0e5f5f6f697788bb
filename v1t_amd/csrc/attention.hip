// v1t_amd — fused attention kernels (gfx950). See attention.h.
//
// All products are "swapped" so that the softmax axis bookkeeping is lane-local:
//   forward (attn_fwd_kernel; 8 waves x 32 queries share each 64-key K/V stage):
//             S^T[key][q] = K . Q^T   (A = K tile rows from LDS, B = Q fragments in registers)
//             q sits on the lane, the 32 keys of a tile in 16 registers x 2 lane halves, so the
//             row max/sum is 15 in-lane ops + one cross-half exchange;
//             O^T[d][q] += V^T . P^T   (A = V tile read TRANSPOSED with ds_read_b64_tr_b16,
//             B = the S^T accumulator converted to bf16 in place — no LDS round trip).
//   backward, default (head dim >= 128, no LSA diagonal): 5 MFMA products.
//             attn_bwd_dkv2_kernel: producer / consumer wave roles, key on the lane: S = Q . K^T, dP = dO . V^T (producers),
//             dV^T += dO^T . P, dK^T += Q^T . dS' (consumers); the bf16 dS' = P (dP - delta) is also stored to HBM and
//             attn_bwd_dq2_kernel computes dQ = dS' . K as a streaming GEMM over it. A single-kernel backward would need
//             fp32 atomics for dQ: at T=1654, dh=160 those bytes (B*H*T*160*4 per 128-key block, 3 GB per 112-image launch)
//             exceed the chip's ~1.3 TB/s atomic rate by far.
//   backward, LSA (diagonal mask + learnable per-head scale) and small head dims: attn_bwd_dq_body + attn_bwd_dkv_body
//             (4 waves, one wave per SIMD, 7 products: S and dP are recomputed in the dQ body), fused in one launch.
//
// The element-wise work between the MFMAs shares the SIMD's vector issue port with them (an MFMA holds it 8 of its 32
// cycles; a lone wave issues one vector instruction per ~4.5 cycles whatever its kind: tools/microbench/valu_issue_cost.hip),
// so it is kept branch-free and minimal: dropout / LSA-diagonal / tail masking are template parameters (no runtime
// flags in the loop), only the last tile carries bounds checks, the softmax scale is folded into the
// exp2 argument, 1/keep and the score scale are folded into the epilogue, and the dropout mask costs one 32-bit hash per
// 2x2 (query, key) block (common.h).
#include <type_traits>

#include <cstdlib>
#include "attention.h"

// Timing-only ablations compile pieces of kernels out and return GARBAGE (V1T_F3_* of the 16-wave forward experiment, V1T_B2_* of the dK/dV
// kernel, V1T_DEV_ABLATION_* of round 5): they are accepted only together with -DV1T_DEV_ABLATION, which v1t_amd/build.py refuses to pass
// when it builds the product library (only `V1T_BUILD_LIB=libv1t_amd_<experiment>.so` builds may carry it).
#if (defined(V1T_F3_NODMA) || defined(V1T_F3_NOLDS) || defined(V1T_F3_STUB) || defined(V1T_B2_NODMA) || defined(V1T_B2_NOFRAG) || defined(V1T_B2_NOVALU) || \
     defined(V1T_B2_NOSTORE) || defined(V1T_B2_NOKEEP) || defined(V1T_DEV_ABLATION_HM_NOSTORE)) && !defined(V1T_DEV_ABLATION)
#error "timing-only ablation macros need -DV1T_DEV_ABLATION (experiment libraries only: V1T_BUILD_LIB=... python -m v1t_amd.build)"
#endif
#ifdef V1T_KPROF
// dev-only in-kernel timeline (tools/kprof.py): s_memtime stamps of one workgroup's waves, KP_T0 <= tile < KP_T0 + KP_NT
#define KP_T0 8
#define KP_NT 8
#define KP_NP 16
__device__ unsigned long long g_kprof[8 * KP_NT * KP_NP];  // up to 8 waves (forward kernel)
#define KP_DECL unsigned long long kp_t[KP_NP] = {}
#define KP_STAMP(i)                                            \
    do {                                                       \
        __builtin_amdgcn_sched_barrier(0);                     \
        asm volatile("s_memtime %0" : "=s"(kp_t[i])::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)
#define KP_FLUSH(kt, wave, lane)                                                                      \
    do {                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        if (blockIdx.x == KP_BLOCK && (kt) >= KP_T0 && (kt) < KP_T0 + KP_NT && (lane) == 0)           \
            for (int i_ = 0; i_ < KP_NP; ++i_) g_kprof[((wave) * KP_NT + (kt) - KP_T0) * KP_NP + i_] = kp_t[i_]; \
    } while (0)
#define KP_BLOCK 300
#else
#define KP_DECL
#define KP_STAMP(i)
#define KP_FLUSH(kt, wave, lane)
#endif

#ifdef V1T_KSUM
// dev-only: per-segment cycle sums over a whole workgroup (no per-step stores): KS_MARK(k) adds the cycles since the previous
// mark to segment k; tools/ksum.py prints them for one workgroup
__device__ unsigned long long g_ksum[8 * 8 + 2];
#define KS_DECL unsigned long long ks_prev = __builtin_amdgcn_s_memtime(), ks_sum[9] = {}; const unsigned long long ks_c0 = ks_prev, ks_r0 = __builtin_amdgcn_s_memrealtime()
#define KS_MARK_(k)                                                  \
    do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                           \
        const unsigned long long ks_now = __builtin_amdgcn_s_memtime(); \
        ks_sum[k] += ks_now - ks_prev;                               \
        ks_prev = ks_now;                                            \
        __builtin_amdgcn_sched_barrier(0);                           \
    } while (0)
#ifdef V1T_KSUM_PRO  // prologue / epilogue breakdown: the steps' marks all go to segment 7, KSP_MARK(k) to segment k
#define KS_MARK(k) KS_MARK_(8)
#define KSP_MARK(k) KS_MARK_(k)
#else
#define KS_MARK(k) KS_MARK_(k)
#define KSP_MARK(k)
#endif
#define KS_END(blk, wave, lane)                                                                                   \
    do {                                                                                                          \
        if (blockIdx.x == (blk) && (lane) == 0) {                                                                 \
            for (int i_ = 0; i_ < 8; ++i_) g_ksum[(wave) * 8 + i_] = ks_sum[i_];                                  \
            if ((wave) == 0) {                                                                                    \
                g_ksum[64] = __builtin_amdgcn_s_memtime() - ks_c0;                                                \
                g_ksum[65] = __builtin_amdgcn_s_memrealtime() - ks_r0;                                            \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#else
#define KS_DECL
#define KS_MARK(k)
#define KSP_MARK(k)
#define KS_END(blk, wave, lane)
#endif

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 6.0f;  // defer-max threshold in log2 units (P <= 2^6 stays exact enough in bf16)

template <int DP>
struct Geo {
    static constexpr int KS = DP / 16;                           // k-steps over the head dim
    static constexpr int DB = DP / 32;                           // 32-wide d blocks
    static constexpr int RSTR = DP + 8;                          // row-read stride (16-B pad)
    static constexpr int TSTR = ((DP / 32) % 2 == 1) ? DP : DP + 32;  // transposed-read-only stride (64 B x odd)
    static constexpr int CHUNKS = 32 * DP / 8;                   // 16-B chunks of a 32-row tile
    static constexpr int ITERS = (CHUNKS + 255) / 256;
};

// (TileDma - global -> LDS tile staging by LDS-DMA, one instruction per 1-KB piece - lives in common.h: the weight-gradient GEMM uses it too)

// Single-instruction helpers: plain fmaxf() on MFMA outputs makes hipcc emit a canonicalising v_max(x, x) per operand,
// and element-wise (bf16_t) casts of accumulator values come out as one v_cvt_pk per ELEMENT plus v_perm to pair them.
// HAZARD: hipcc does not see inline asm as a reader of MFMA results and inserts no wait states for it (the hardware
// does not interlock: the max of a just-finished S tile came out different from run to run). The max helpers are
// volatile so that they stay behind mfma_result_fence(), which every use on fresh MFMA output must be preceded by.
DEVFN void mfma_result_fence() { asm volatile("s_nop 11" ::: "memory"); }  // 12 wait states: 8-pass XDL write -> VALU read
DEVFN float vmax3(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
DEVFN float vmax2(float a, float b) {
    float r;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// acc_to_b() (common.h) with explicit PACKED conversions: a 2-vector fptrunc selects v_cvt_pk_bf16_f32 directly. (Not
// inline asm: hipcc inserts no wait states between an asm-written VGPR and the MFMA that reads it as an operand - the
// dV of a 64-wide head came out NaN from run to run that way.)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) bf16_t bf16x2_t;
DEVFN bf16x8 acc_to_b_pk(const f32x16& x, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2_t v = {x[8 * s + 2 * j], x[8 * s + 2 * j + 1]};
        const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
        r[2 * j] = b[0];
        r[2 * j + 1] = b[1];
    }
    return r;
}
// max over the two half-waves (lane l and l + 32 hold the same query): one v_permlane32_swap instead of a ds_bpermute
DEVFN float half_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// s_waitcnt vmcnt(n) for a wave-uniform n (the immediate must be a literal)
DEVFN void wait_vmcnt_dyn(int n) {
    switch (n) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
// raw barrier: waits for this wave's LDS operations only (a __syncthreads() would also drain the LDS-DMA queue)
DEVFN void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
DEVFN void zero16(f32x16& x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = 0.f;
}

// lane part of the transposed-fragment address (see lds_tr_frag in common.h): (4h + q) * stride + 16*(g&1) + 4p
DEVFN int tr_lane_off(int lane, int stride) {
    const int g = lane >> 4, i = lane & 15;
    return (4 * (g >> 1) + (i >> 2)) * stride + 16 * (g & 1) + 4 * (i & 3);
}
// A operand, 32 m x 16 k, from an image stored [k][m]; k order matches acc_to_b(). `p` = img + tr_lane_off.
template <int STR>
DEVFN bf16x8 tr_frag(const bf16_t* p, int k0, int m0) {
    const bf16x4 lo = lds_tr_read(p + k0 * STR + m0);
    const bf16x4 hi = lds_tr_read(p + (k0 + 8) * STR + m0);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// Row-permuted tile images (dK/dV kernel): a 32-row tile whose rows are read BOTH as 16-B row chunks (ds_read_b128: wants a row
// stride of 16 B x odd, RSTR) and transposed (ds_read_b64_tr_b16, a 16-lane group takes 4 consecutive rows x 32 B: with RSTR's
// 84-bank stride rows 0 and 3 of a group share 4 banks - a quarter of all LDS cycles of that kernel were bank conflicts). No row
// stride serves both, but the 4 x 4 index transpose inside each 16-row block does: logical row 4 a + b sits at position a + 4 b,
// so the 4 rows of a transposed read are 4 positions apart (bank starts 0 / 16 / 32 / 48) and the 8 / 16 consecutive rows of a
// row read still land on 16 distinct residues.
DEVFN int perm_row(int r) { return (r & 16) | ((r & 3) << 2) | ((r >> 2) & 3); }
// transposed-fragment lane offset / fragment read for such an image (rows k0 + 8 e + 4 h + q -> positions k0 + 2 e + h + 4 q)
DEVFN int tr_lane_off_perm(int lane, int stride) {
    const int g = lane >> 4, i = lane & 15;
    return ((g >> 1) + 4 * (i >> 2)) * stride + 16 * (g & 1) + 4 * (i & 3);
}
template <int STR>
DEVFN bf16x8 tr_frag_perm(const bf16_t* p, int k0, int m0) {
    const bf16x4 lo = lds_tr_read(p + k0 * STR + m0);
    const bf16x4 hi = lds_tr_read(p + (k0 + 2) * STR + m0);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// keep-bits of the 4 consecutive elements (r&3 = 0..3) of one accumulator group: the lane's varying
// coordinate runs over 2 consecutive 2x2 blocks (words w0: elements 0,1; w1: elements 2,3); sh_even /
// sh_odd select the byte for an even / odd value of the varying coordinate (they encode the parity of
// the lane's fixed coordinate).
DEVFN void drop4(uint32_t x0, uint32_t step, uint32_t sh_even, uint32_t sh_odd, uint32_t thr, bool (&keep)[4]) {
    const uint32_t w0 = mix1(x0), w1 = mix1(x0 + step);
    keep[0] = __builtin_amdgcn_ubfe(w0, sh_even, 8u) >= thr;
    keep[1] = __builtin_amdgcn_ubfe(w0, sh_odd, 8u) >= thr;
    keep[2] = __builtin_amdgcn_ubfe(w1, sh_even, 8u) >= thr;
    keep[3] = __builtin_amdgcn_ubfe(w1, sh_odd, 8u) >= thr;
}

// 1-D grid over (row block, head, image), XCD-aware: the row blocks of one (image, head) get consecutive
// logical ids and each XCD owns a contiguous chunk of ids, so a head's K/V (or Q/dO) stays in ONE L2.
DEVFN void decode_block(const AttnArgs& a, int bid, int nblk, int& rb, int& h, int& b, int rows_per_block = 128) {
    const int nrb = (a.T + rows_per_block - 1) / rows_per_block;
    const int lid = xcd_remap(bid, nblk);
    rb = lid % nrb;
    const int bh = lid / nrb;
    h = bh % a.H;
    b = bh / a.H;
}

// Forward kernel: the 1654 queries of an (image, head) as F full row blocks (256 queries, 8 waves) followed by Hh half blocks (128 queries:
// waves 4-7 hold no query and only stage K / V, so the block takes ~2/3 of a full one's time). The launcher (choose_fwd_split) picks (F, Hh):
// by default the natural cover (T = 1654: 6 full + the 118-query remainder as one half block); for small launches the cover whose
// list-scheduled makespan on the 32 CUs of an XCD is shortest (14 images: 4 full + 5 half per (image, head) = 1.67 rounds instead of 2).
// With `lpt` the full blocks of an XCD's chunk of logical ids are dispatched before its half blocks (longest processing time first): the
// hardware hands out workgroups in blockIdx order as CUs free up, so the tail of the launch is then made of the short workgroups.
// Logical id = (image * H + head) * (F + Hh) + block; XCD-aware as decode_block (each XCD owns a contiguous chunk of ids).
DEVFN void decode_fwd(const AttnArgs& a, int bid, int nblk, int& q0, int& q_end, int& h, int& b) {
    const int F = a.fwd_full, Hh = a.fwd_half, nrb = F + Hh;
    int bh, i;
    if (!a.fwd_lpt || Hh == 0 || F == 0) {
        const int lid = xcd_remap(bid, nblk);
        bh = lid / nrb;
        i = lid - bh * nrb;
    } else {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;  // xcd_remap: this XCD owns logical ids [base, base + n)
        const int n = q + (xcd < r ? 1 : 0);
        auto fulls_below = [&](int x) { return (x / nrb) * F + min(x % nrb, F); };  // full blocks among the logical ids [0, x)
        const int f0 = fulls_below(base), nfull = fulls_below(base + n) - f0;
        if (k < nfull) {
            const int m = f0 + k;  // the m-th full block of the launch
            bh = m / F;
            i = m - bh * F;
        } else {
            const int j = (base - f0) + (k - nfull);  // the j-th half block of the launch
            bh = j / Hh;
            i = F + (j - bh * Hh);
        }
    }
    h = bh % a.H;
    b = bh / a.H;
    q0 = i < F ? 256 * i : 256 * F + 128 * (i - F);
    q_end = min(a.T, q0 + (i < F ? 256 : 128));
}

// ------------------------------------------------------------------------------------------
constexpr int FWD_WAVES = 8;  // 256 queries per workgroup share each K/V tile: half the LDS-DMA pieces per wave of a 4-wave workgroup
template <int DP, bool DROP, bool DIAG, bool STAGGER = false>
__global__ __launch_bounds__(64 * FWD_WAVES, 1) void attn_fwd_kernel(AttnArgs a) {
    using G = Geo<DP>;
    // K/V are staged 64 keys at a time (one barrier and one burst of LDS-DMA pieces per 64 keys) and consumed as two
    // 32-key tiles
    // (UNIFORM: every wave issues three K and three V pieces per stage; with the per-wave piece counts - 21 K and 20 V pieces over 8 waves -
    // and the ragged check the issue was a chain of 13 wave-uniform branches in the middle of part_b's softmax stretch)
#ifdef V1T_FWD_DMA_BRANCHY  // dev (A/B): the round-4 form
    constexpr bool DMA_UNI = false;
#else
    constexpr bool DMA_UNI = true;
#endif
    using DmaK = TileDma<DP, G::RSTR, 64, FWD_WAVES, DMA_UNI>;
    using DmaV = TileDma<DP, G::TSTR, 64, FWD_WAVES, DMA_UNI>;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][DmaK::LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) bf16_t sV[2][DmaV::LDS_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int q0, q_end, h, b;
    decode_fwd(a, blockIdx.x, gridDim.x, q0, q_end, h, b);
    const int q = q0 + 32 * wave + (lane & 31);
    const int h2 = lane >> 5;
    const int HD = a.H * DP;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    const bf16_t* kbase = qkv_b + HD + h * DP;
    const bf16_t* vbase = qkv_b + 2 * HD + h * DP;
    const float c = a.scale[a.scale_per_head ? h : 0] * LOG2E;
    DmaK dmaK;
    DmaV dmaV;
    dmaK.init(lane, wave, a.ldqkv);
    dmaV.init(lane, wave, a.ldqkv);

    // the first K / V stage goes out BEFORE the Q fragments: the fragment loads (16 B per lane, two lanes per row) keep the
    // workgroup's memory path busy for thousands of cycles and everything issued behind them waits (dK/dV kernel, same finding)
    dmaK.issue(kbase, 0, a.T, sK[0]);
    dmaV.issue(vbase, 0, a.T, sV[0]);
    bf16x8 qf[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        u32x4 t = (q < q_end) ? *(const u32x4*)(qkv_b + (size_t)q * a.ldqkv + h * DP + 16 * ks + 8 * h2) : u32x4{0, 0, 0, 0};
        qf[ks] = *(bf16x8*)&t;
    }
    f32x16 o[G::DB];
#pragma unroll
    for (int d = 0; d < G::DB; ++d) zero16(o[d]);
    float m2 = NEG_BIG, lsum = 0.f;  // running max in the scaled log2 domain; per-lane partial sum

    // dropout: lane-fixed coordinate = q (row), varying = key
    const uint32_t T2 = (uint32_t)(a.T + 1) >> 1;
    const uint32_t dbase = a.adrop.key + ((uint32_t)(b * a.H + h) * T2 + ((uint32_t)q >> 1)) * ADROP_K1 + (uint32_t)(2 * h2) * ADROP_K2;
    const uint32_t sh_even = 16 * (q & 1), sh_odd = sh_even + 8;
    const uint32_t tile_bh = (uint32_t)(b * a.H + h), tile_nqb = (uint32_t)(a.T + 31) >> 5, tile_qb = (uint32_t)(q0 >> 5) + (uint32_t)wave;  // wave-uniform
    (void)sh_odd; (void)tile_bh; (void)tile_nqb; (void)tile_qb;

    const int koff = (lane & 31) * G::RSTR + 8 * h2;
    const int voff = tr_lane_off(lane, G::TSTR);
    const int nt = (a.T + 31) / 32;

    // A wave whose 32 queries all lie beyond T (T = 1654: waves 4-7 of every (image, head)'s 7th workgroup) only stages its share of
    // K / V and keeps the barriers: its SIMD partner then has the matrix pipe and the issue port to itself, the ragged workgroup
    // finishes in ~2/3 of the time and the grid (12.25 rounds of equal workgroups = 13) desynchronises into ~12.
    const bool dead_wave = q0 + 32 * wave >= q_end;  // wave-uniform
    KP_DECL;
    // A 32-key tile in two parts. part_a: S^T = K Q^T (all K fragments in flight, then the chain) and the transposed V fragments,
    // issued behind the chain's MFMAs so that they land while the softmax runs; everything it reads from LDS is in registers when
    // it returns. part_b: softmax, dropout, O^T += V^T P^T from those registers (no LDS access), and - for one part_b per stage -
    // the LDS-DMA of stage `next_stage` (-1: none) in its VALU-only stretch.
    auto part_a = [&](auto tail_tag, int kt, int buf, f32x16& s, bf16x8 (&vfr)[2 * G::DB]) __attribute__((always_inline)) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        KP_STAMP(0);
        zero16(s);
        const bf16_t* kp = &sK[buf][32 * (kt & 1) * G::RSTR + koff];
        const bf16_t* vp = &sV[buf][32 * (kt & 1) * G::TSTR + voff];
        bf16x8 kfr[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) kfr[ks] = *(const bf16x8*)(kp + 16 * ks);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) s = mfma32(kfr[ks], qf[ks], s);
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            vfr[2 * d] = tr_frag<G::TSTR>(vp, 0, 32 * d);
            vfr[2 * d + 1] = tr_frag<G::TSTR>(vp, 16, 32 * d);
        }
        // issue order: every K read first (one LDS latency for the whole chain), then one MFMA + the
        // transposed V reads that fit behind it
        __builtin_amdgcn_sched_group_barrier(0x100, G::KS, 0);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * G::DB / G::KS, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        KP_STAMP(1);
        if constexpr (TAIL || DIAG) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * kt + acc_row(r, lane);
                bool dead = false;
                if constexpr (TAIL) dead = key >= a.T;
                if constexpr (DIAG) dead = dead || key == q;
                s[r] = dead ? NEG_BIG : s[r];
            }
        }
    };
    auto part_b = [&](auto rag_tag, int kt, int next_stage, f32x16& s, const bf16x8 (&vfr)[2 * G::DB]) __attribute__((always_inline)) {
        constexpr bool MAY_RAG = decltype(rag_tag)::value;  // may the stage issued here reach beyond T (only the last one can)
        mfma_result_fence();
        float pmax = vmax3(s[0], s[1], s[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) pmax = vmax3(pmax, s[r], s[r + 1]);
        pmax = half_max(vmax2(pmax, s[15])) * c;
        KP_STAMP(2);
        if (!__all(pmax <= m2 + RESCALE_THR)) {
            const float mn = fmaxf(m2, pmax);
            const float alpha = fast_exp2(m2 - mn);
#pragma unroll
            for (int d = 0; d < G::DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            lsum *= alpha;
            m2 = mn;
        }
        const float negm = -m2;
        // the next stage's LDS-DMA is issued HERE, in the VALU-only stretch: a piece costs its wave 100-185 cycles of
        // issue while ds_reads are in flight (tile start) but 25-60 when the LDS is quiet (kprof timeline)
        if (next_stage >= 0) {  // wave-uniform
            dmaK.template issue<MAY_RAG>(kbase, 64 * next_stage, a.T, sK[next_stage & 1]);
            dmaV.template issue<MAY_RAG>(vbase, 64 * next_stage, a.T, sV[next_stage & 1]);
        }
        KP_STAMP(3);
        uint32_t thr8v = 0;  // byte threshold of this 32 x 32 tile (common.h: dithered per tile, scalar arithmetic)
        if constexpr (DROP) thr8v = attn_tile_thresh(a.adrop, tile_bh, tile_nqb, tile_qb, (uint32_t)kt);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t wq[2] = {0u, 0u};
            // dropout decisions by a byte compare (v_cmp_ge_u32_sdwa + select: 2 instructions per element, a shift per word) instead of bit-field
            // extract + compare + select; the same decisions bit for bit (round 4: forward 979 -> 962-967 us, profiles/r04_attn_experiments.txt #6)
            if constexpr (DROP) {  // the two words of this group, shifted so that this query's decisions are bytes 0 (even key) and 1 (odd key)
                const uint32_t x0 = dbase + (uint32_t)(16 * kt + 4 * g) * ADROP_K2;
                wq[0] = mix1(x0) >> sh_even;
                wq[1] = mix1(x0 + ADROP_K2) >> sh_even;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = fast_exp2(fmaf(s[4 * g + j], c, negm));
                lsum += p;
                if constexpr (DROP) {
                    float pd;
                    if ((j & 1) == 0)
                        asm("v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(pd) : "v"(wq[j >> 1]), "s"(thr8v), "v"(p) : "vcc");
                    else
                        asm("v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(pd) : "v"(wq[j >> 1]), "s"(thr8v), "v"(p) : "vcc");
                    s[4 * g + j] = pd;
                } else {
                    s[4 * g + j] = p;
                }
            }
        }
        const bf16x8 p0 = acc_to_b_pk(s, 0), p1 = acc_to_b_pk(s, 1);
        KP_STAMP(4);
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            o[d] = mfma32(vfr[2 * d], p0, o[d]);
            o[d] = mfma32(vfr[2 * d + 1], p1, o[d]);
        }
#ifdef V1T_KPROF
        KP_STAMP(5);
        KP_FLUSH(kt, wave, lane);
#endif
    };
    auto tile = [&](auto tail_tag, auto rag_tag, int kt, int buf, int next_stage) __attribute__((always_inline)) {
        f32x16 s;
        bf16x8 vfr[2 * G::DB];
        part_a(tail_tag, kt, buf, s, vfr);
        part_b(rag_tag, kt, next_stage, s, vfr);
    };
    constexpr std::true_type RAG{};
    constexpr std::false_type NORAG{};

    touch(qf);
    touch(c);
    dma_wait_and_barrier();
    const int ns = (nt + 1) / 2;  // 64-key stages
    if (dead_wave) {  // its own loop (a branch inside the tile costs the live path 55 spilled registers): same stages, same barriers
        for (int st = 0; st < ns - 1; ++st) {
            dmaK.issue(kbase, 64 * (st + 1), a.T, sK[(st & 1) ^ 1]);
            dmaV.issue(vbase, 64 * (st + 1), a.T, sV[(st & 1) ^ 1]);
            dma_wait_and_barrier();
        }
        return;
    }
    if (STAGGER && wave >= FWD_WAVES / 2 && ns > 1) {
        // The SIMD partners (waves w and w + 4) run the same program between the same barriers: left alone they do their MFMA chains
        // at the same time and their softmax at the same time, so the matrix pipe idles while both are in the VALU stretch. The
        // second-dispatched half therefore runs its stages ROTATED: the barrier falls behind part_a of the stage's second tile
        // (every LDS read of the stage done, S and the V fragments in registers), part_b of that tile runs after the barrier.
        // Same work per stage, same barriers, same two buffers - only the phase differs, by the length of a softmax + P.V.
        f32x16 sc;
        bf16x8 vc[2 * G::DB];
        tile(std::false_type{}, RAG, 0, 0, 1);
        part_a(std::false_type{}, 1, 0, sc, vc);
        dma_wait_and_barrier();
        for (int st = 1; st < ns - 1; ++st) {
            const int buf = st & 1;
            part_b(RAG, 2 * st - 1, st + 1, sc, vc);
            tile(std::false_type{}, RAG, 2 * st, buf, -1);
            part_a(std::false_type{}, 2 * st + 1, buf, sc, vc);
            dma_wait_and_barrier();
        }
        part_b(RAG, 2 * (ns - 1) - 1, -1, sc, vc);
    } else {
        // only the LAST stage can reach beyond T (64 (ns - 1) < T): the steady loop issues stages 1 .. ns - 2 without the ragged check, the
        // peeled last iteration issues stage ns - 1 with it
        auto stage = [&](auto rag_tag, int st) __attribute__((always_inline)) {
            const int buf = st & 1;
            tile(std::false_type{}, rag_tag, 2 * st, buf, st + 1);
            tile(std::false_type{}, rag_tag, 2 * st + 1, buf, -1);
            dma_wait_and_barrier();
            KP_STAMP(7);
        };
        if constexpr (DMA_UNI) {
            for (int st = 0; st < ns - 2; ++st) stage(NORAG, st);
            if (ns >= 2) stage(RAG, ns - 2);
        } else {
            for (int st = 0; st < ns - 1; ++st) stage(RAG, st);
        }
    }
    tile(std::true_type{}, RAG, 2 * (ns - 1), (ns - 1) & 1, -1);
    if (2 * (ns - 1) + 1 < nt) tile(std::true_type{}, RAG, 2 * (ns - 1) + 1, (ns - 1) & 1, -1);

    const float ltot = lsum + __shfl_xor(lsum, 32);
    const float inv = (DROP ? a.adrop.inv_keep : 1.0f) / ltot;
    if (q < q_end) {
        if (h2 == 0) a.lse2[((size_t)b * a.H + h) * a.T + q] = m2 + log2f(ltot);
        const size_t orow = ((size_t)b * a.T + q) * a.ldo + h * DP;
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                bf16x4 w, wl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = o[d][4 * rq + j] * inv;
                    w[j] = (bf16_t)v;
                    wl[j] = aux_plane(v, w[j], a.lo_f16);
                }
                if (a.o) *(bf16x4*)(a.o + orow + 32 * d + 8 * rq + 4 * h2) = w;
                if (a.o_lo) *(bf16x4*)(a.o_lo + orow + 32 * d + 8 * rq + 4 * h2) = wl;
            }
    }
}

#ifdef V1T_EXPERIMENTS  // round-3 experiment kernel (V1T_ATTN_FWD_V3): experiment builds only, never in the product library
// ------------------------------------------------------------------------------------------
// Forward with FOUR waves per SIMD (round 3). Why: in the kernels above a wave runs S -> softmax -> P.V one after the other, ~260
// instructions per 32-key tile at ~10 cycles each (in-kernel timeline, tools/kprof.py: MFMA chains, LDS and exp latencies are exposed
// because a SIMD holds only two such waves - each needs ~230 registers), and the matrix pipe is busy 47 % of the time. Here the tile
// is cut between two waves of <= 128 registers: waves 0-7 ("S-waves", 32 queries each: Q fragments, S^T = K Q^T, the online softmax,
// dropout; they never touch V or O) hand the bf16 P^T fragments - already the B operand of the second product - through LDS to
// waves 8-15 ("PV-waves", SIMD partners w + 8: O^T += V^T P^T, 80 accumulator registers, all K / V staging, the output). A SIMD then
// holds two S-waves and two PV-waves whose stalls overlap. One barrier per 32-key tile; the PV-waves work one tile behind.
// K / V tiles of 32 keys in a 5-deep ring (tile t in slot t % 5: K is read in interval t - 1 - the S-waves run their chain one tile
// ahead of their softmax - V in interval t + 1, the LDS-DMA of tile t + 3 is issued in interval t into the slot tile t - 2 left in
// interval t - 1). The running maximum lives in the S-waves; when it
// moves (deferred: RESCALE_THR) they publish the factor per query and a flag, and the PV-wave scales O before it adds the tile.
constexpr int F3_PAIRS = 8, F3_NBUF = 5;
template <int DP, bool DROP>
__global__ __launch_bounds__(128 * F3_PAIRS, 4) void attn_fwd3_kernel(AttnArgs a) {
    using G = Geo<DP>;
    using DmaK = TileDma<DP, G::RSTR, 32, F3_PAIRS>;
    using DmaV = TileDma<DP, G::TSTR, 32, F3_PAIRS>;
    __shared__ __attribute__((aligned(16))) bf16_t sK[F3_NBUF][DmaK::LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) bf16_t sV[F3_NBUF][DmaV::LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) u32x4 sP[2][F3_PAIRS][2][64];  // [tile parity][pair][k-step][lane]
    __shared__ float sAlpha[2][F3_PAIRS][64];                              // rescale factor of the lane's query
    __shared__ int sFlag[2][F3_PAIRS];                                     // 1: the factors of this tile are to be applied
    __shared__ float sInv[F3_PAIRS][64];                                   // epilogue: (1 / keep) / row sum
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave & (F3_PAIRS - 1);      // pair: S-wave pr and PV-wave pr + 8 own queries [32 pr, 32 pr + 32) of the block
    const bool pv = wave >= F3_PAIRS;          // wave-uniform role
    int rb, h, b;
    decode_block(a, blockIdx.x, gridDim.x, rb, h, b, 32 * F3_PAIRS);
    const int q = rb * (32 * F3_PAIRS) + 32 * pr + (lane & 31);
    const int h2 = lane >> 5;
    const int HD = a.H * DP;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    const bf16_t* kbase = qkv_b + HD + h * DP;
    const bf16_t* vbase = qkv_b + 2 * HD + h * DP;
    const int nt = (a.T + 31) / 32;
    const bool dead_pair = rb * (32 * F3_PAIRS) + 32 * pr >= a.T;  // all 32 queries beyond T: stage and keep the barriers only

    if (pv) {
        // ------------------------------------------------------------------ PV-wave
        DmaK dmaK;
        DmaV dmaV;
        dmaK.init(lane, pr, a.ldqkv);
        dmaV.init(lane, pr, a.ldqkv);
        // vector-memory operations this wave issues per tile (counted waits)
        const int nops = min(DmaK::PW, max(0, DmaK::NINST - pr * DmaK::PW)) + min(DmaV::PW, max(0, DmaV::NINST - pr * DmaV::PW));
        auto stage = [&](int t) {
            dmaK.issue(kbase, 32 * t, a.T, sK[t % F3_NBUF]);
            dmaV.issue(vbase, 32 * t, a.T, sV[t % F3_NBUF]);
        };
        stage(0);
        if (nt > 1) stage(1);
        if (nt > 2) stage(2);
        f32x16 o[G::DB];
#pragma unroll
        for (int d = 0; d < G::DB; ++d) zero16(o[d]);
        const int voff = tr_lane_off(lane, G::TSTR);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // tiles 0 .. 2 in LDS
        for (int i = 0; i <= nt; ++i) {  // interval i: the S-waves finish tile i (and issue the chain of tile i + 1), this wave works on tile i - 1
#ifndef V1T_F3_NODMA  // dev ablation: without it the tiles of the prologue are re-used (garbage results, timing only)
            if (i + 3 < nt) stage(i + 3);
#endif
            if (i >= 1 && !dead_pair) {
                const int j = i - 1, par = j & 1;
                const u32x4 pa = sP[par][pr][0][lane], pb = sP[par][pr][1][lane];
                const int flag = __builtin_amdgcn_readfirstlane(sFlag[par][pr]);
                if (flag) {  // the running maximum moved in tile j: O of the tiles before it is scaled first
                    const float al = sAlpha[par][pr][lane];
#pragma unroll
                    for (int d = 0; d < G::DB; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[d][r] *= al;
                }
                const bf16_t* vp = &sV[j % F3_NBUF][voff];
                const bf16x8 p0 = __builtin_bit_cast(bf16x8, pa), p1 = __builtin_bit_cast(bf16x8, pb);
#ifdef V1T_F3_NOLDS  // dev ablation: operands from registers instead of LDS fragments (garbage results, timing only)
                (void)vp;
#pragma unroll
                for (int d = 0; d < G::DB; ++d) {
                    o[d] = mfma32(p1, p0, o[d]);
                    o[d] = mfma32(p0, p1, o[d]);
                }
#else
#pragma unroll
                for (int d = 0; d < G::DB; ++d) {
                    o[d] = mfma32(tr_frag<G::TSTR>(vp, 0, 32 * d), p0, o[d]);
                    o[d] = mfma32(tr_frag<G::TSTR>(vp, 16, 32 * d), p1, o[d]);
                }
#endif
            }
            // everything issued before this interval has landed (tile i + 2, whose K the S-waves read in the next interval, among
            // it); this interval's own operations (tile i + 3) may stay in flight
#ifndef V1T_F3_NODMA
            if (i + 3 < nt) wait_vmcnt_dyn(nops);
            else
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
        if (q < a.T) {
            const float inv = sInv[pr][lane];
            const size_t orow = ((size_t)b * a.T + q) * a.ldo + h * DP;
#pragma unroll
            for (int d = 0; d < G::DB; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    bf16x4 w, wl;
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const float v = o[d][4 * rq + jx] * inv;
                        w[jx] = (bf16_t)v;
                        wl[jx] = aux_plane(v, w[jx], a.lo_f16);
                    }
                    if (a.o) *(bf16x4*)(a.o + orow + 32 * d + 8 * rq + 4 * h2) = w;
                    if (a.o_lo) *(bf16x4*)(a.o_lo + orow + 32 * d + 8 * rq + 4 * h2) = wl;
                }
        }
        return;
    }

    // ---------------------------------------------------------------------- S-wave
    // Software-pipelined over the tiles: interval i issues the chain S^T(i + 1) = K(i + 1) Q^T one MFMA per slot (its K fragment read
    // LA slots ahead) with a slice of tile i's softmax (exp2, row sum, dropout, packing) in each slot, fenced by sched_barrier(0) -
    // the wave issues in order, so its own MFMAs and its vector work only overlap if they alternate in the instruction stream.
    const float c = a.scale[a.scale_per_head ? h : 0] * LOG2E;
    bf16x8 qf[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        u32x4 t = (q < a.T) ? *(const u32x4*)(qkv_b + (size_t)q * a.ldqkv + h * DP + 16 * ks + 8 * h2) : u32x4{0, 0, 0, 0};
        qf[ks] = *(bf16x8*)&t;
    }
    float m2 = NEG_BIG, lsum = 0.f;
    const uint32_t T2 = (uint32_t)(a.T + 1) >> 1;
    const uint32_t dbase = a.adrop.key + ((uint32_t)(b * a.H + h) * T2 + ((uint32_t)q >> 1)) * ADROP_K1 + (uint32_t)(2 * h2) * ADROP_K2;
    const uint32_t sh_even = 16 * (q & 1), sh_odd = sh_even + 8;
    const int koff = (lane & 31) * G::RSTR + 8 * h2;
    touch(qf);
    touch(c);
    __builtin_amdgcn_s_barrier();  // tiles 0 .. 2 in LDS (the PV-waves waited for them)
    if (dead_pair) {
        for (int i = 0; i <= nt; ++i) lds_barrier();
        return;
    }
    auto mask_tail = [&](int t, f32x16& s) {  // keys beyond T (last tile only)
        if (32 * t + 32 > a.T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (32 * t + acc_row(r, lane) >= a.T) ? NEG_BIG : s[r];
        }
    };
    uint32_t w0 = 0, w1 = 0;
    auto element = [&](int t, f32x16& s, int r, float negm) {  // element r of tile t: score -> P (dropped), row sum
        const int g = r >> 2, jx = r & 3;
        if constexpr (DROP) {
            if (jx == 0) {
                const uint32_t x0 = dbase + (uint32_t)(16 * t + 4 * g) * ADROP_K2;
                w0 = mix1(x0);
                w1 = mix1(x0 + ADROP_K2);
            }
        }
#ifdef V1T_F3_STUB  // dev ablation: no element-wise work at all (what is left is MFMA + LDS + staging + barriers)
        (void)g; (void)negm;
        asm volatile("" : "+v"(s[r]));
        return;
#endif
        const float p = fast_exp2(fmaf(s[r], c, negm));
        lsum += p;
        if constexpr (DROP) {
            const bool keep = __builtin_amdgcn_ubfe(jx < 2 ? w0 : w1, (jx & 1) ? sh_odd : sh_even, 8u) >= attn_tile_thresh(a.adrop, (uint32_t)(b * a.H + h), (uint32_t)(a.T + 31) >> 5, (uint32_t)q >> 5, (uint32_t)t);
            s[r] = keep ? p : 0.f;
        } else {
            s[r] = p;
        }
    };
    f32x16 sA, sB;
    {  // tile 0's chain, un-pipelined
        zero16(sA);
        const bf16_t* kp = &sK[0][koff];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) sA = mfma32(*(const bf16x8*)(kp + 16 * ks), qf[ks], sA);
    }
    constexpr int LA = 3;
    auto interval = [&](int i, f32x16& s, f32x16& s2) {
        const int par = i & 1;
        mask_tail(i, s);
        mfma_result_fence();
        float pmax = vmax3(s[0], s[1], s[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) pmax = vmax3(pmax, s[r], s[r + 1]);
        pmax = half_max(vmax2(pmax, s[15])) * c;
        int flag = 0;
        if (!__all(pmax <= m2 + RESCALE_THR)) {
            const float mn = fmaxf(m2, pmax);
            const float alpha = fast_exp2(m2 - mn);
            lsum *= alpha;
            m2 = mn;
            sAlpha[par][pr][lane] = alpha;
            flag = 1;
        }
        if (lane == 0) sFlag[par][pr] = flag;
        const float negm = -m2;
        const bool next = i + 1 < nt;  // wave-uniform
        unsigned ka = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&sK[(i + 1) % F3_NBUF][koff];
        asm volatile("" : "+v"(ka));  // opaque base: fragment addresses stay immediates
        auto frag = [&](int m) { return *(const __attribute__((address_space(3))) bf16x8*)(uintptr_t)(ka + 32u * (unsigned)m); };
        bf16x8 fr[G::KS];
        if (next) {
#pragma unroll
            for (int m = 0; m < LA; ++m) fr[m] = frag(m);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s2[r] = 0.f;
        bf16x8 n0, n1;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < G::KS; ++m) {
            if (next) {
#ifdef V1T_F3_NOLDS
                s2 = mfma32(qf[(m + 1) % G::KS], qf[m], s2);
#else
                if (m + LA < G::KS) fr[m + LA] = frag(m + LA);
                s2 = mfma32(fr[m], qf[m], s2);
#endif
            }
#pragma unroll
            for (int it = m * 18 / G::KS; it < (m + 1) * 18 / G::KS; ++it) {  // 16 element slices + 2 packing slices over the slots
                if (it < 16) element(i, s, it, negm);
                else if (it == 16) n0 = acc_to_b_pk(s, 0);
                else n1 = acc_to_b_pk(s, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        sP[par][pr][0][lane] = __builtin_bit_cast(u32x4, n0);
        sP[par][pr][1][lane] = __builtin_bit_cast(u32x4, n1);
        lds_barrier();
    };
    for (int i = 0; i < nt; i += 2) {
        interval(i, sA, sB);
        if (i + 1 < nt) interval(i + 1, sB, sA);
    }
    {  // interval nt: the PV-waves finish the last tile; they read the normalisation behind the last barrier
        const float ltot = lsum + __shfl_xor(lsum, 32);
        sInv[pr][lane] = (DROP ? a.adrop.inv_keep : 1.0f) / ltot;
        if (q < a.T && h2 == 0) a.lse2[((size_t)b * a.H + h) * a.T + q] = m2 + log2f(ltot);
        lds_barrier();
    }
}

#endif  // V1T_EXPERIMENTS

// delta[b][h][t] = keep_prob * sum_d dO * O. 16 lanes per (row, head) segment: a load instruction reads 256 contiguous
// bytes of each of its 4 segments (one thread per segment read 16 B at a 320-B stride per lane: 3.4 TB/s).
template <int DP>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs a, float* delta) {
    constexpr int CH = DP / 8;  // 16-B chunks per segment
    const int sub = threadIdx.x & 15;
    const long long seg = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const long long total = (long long)a.B * a.T * a.H;
    const bool ok = seg < total;
    const int h = ok ? (int)(seg % a.H) : 0;
    const long long row = ok ? seg / a.H : 0;
    const bf16_t* po = a.o + (size_t)row * a.ldo + h * DP;
    const bf16_t* pd = a.dO + (size_t)row * a.lddo + h * DP;
    float acc = 0.f;
    if (ok) {
#pragma unroll
        for (int c = sub; c < CH; c += 16) {
            const bf16x8 x = *(const bf16x8*)(po + 8 * c);
            const bf16x8 y = *(const bf16x8*)(pd + 8 * c);
            if (a.o_f16) {  // kernel-uniform
                const f16x8 xh = __builtin_bit_cast(f16x8, x);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)xh[j] * (float)y[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)x[j] * (float)y[j];
            }
        }
    }
    acc += __shfl_xor(acc, 8);
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 1);
    if (ok && sub == 0) {
        const long long b_ = row / a.T, t = row % a.T;
        delta[((size_t)b_ * a.H + h) * a.T + t] = acc * a.adrop.keep_prob;  // 1/keep is folded into the backward epilogues
    }
}

// ------------------------------------------------------------------------------------------
// DP >= 128: Q + dO fragments (2*DP/4 VGPRs) + the dQ accumulator (DP/2) + S/dP/staging exceed 256
// registers, so that shape runs one wave per SIMD with the 512-register budget (no spills).
constexpr int BWD_TR = 64;  // rows per staged tile of the backward kernels
template <int DP>
struct BwdLds {  // one layout for both backward bodies, so that they can share a launch
    static constexpr int TE = TileDma<DP, Geo<DP>::RSTR, BWD_TR>::LDS_ELEMS;
    bf16_t a[2][TE];      // dQ body: K tiles;  dK/dV body: Q tiles
    bf16_t b[2][TE];      // dQ body: V tiles;  dK/dV body: dO tiles
    float l[2][BWD_TR];   // dK/dV body: lse2 of the tile's queries
    float d[2][BWD_TR];   // dK/dV body: delta * keep_prob
};

template <int DP, bool DROP, bool DIAG>
DEVFN void attn_bwd_dq_body(const AttnArgs& a, int bid, int nblk, BwdLds<DP>& lds) {
    using G = Geo<DP>;
    // 64-key tiles processed as two independent 32-key halves: with one wave per SIMD the element-wise
    // work of one half is issued while the MFMAs of the other half execute (the matrix pipe is asynchronous)
    constexpr int TR = BWD_TR;
    auto& sK = lds.a;
    auto& sV = lds.b;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rb, h, b;
    decode_block(a, bid, nblk, rb, h, b);
    const int q = rb * 128 + 32 * wave + (lane & 31);
    const int h2 = lane >> 5;
    const int HD = a.H * DP;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    const bf16_t* kbase = qkv_b + HD + h * DP;
    const bf16_t* vbase = qkv_b + 2 * HD + h * DP;
    TileDma<DP, G::RSTR, TR> dma;
    dma.init(lane, wave, a.ldqkv);
    const float sc = a.scale[a.scale_per_head ? h : 0];
    const float c = sc * LOG2E;
    const bool qok = q < a.T;

    bf16x8 qf[G::KS], dof[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        u32x4 t = qok ? *(const u32x4*)(qkv_b + (size_t)q * a.ldqkv + h * DP + 16 * ks + 8 * h2) : u32x4{0, 0, 0, 0};
        qf[ks] = *(bf16x8*)&t;
        u32x4 u = qok ? *(const u32x4*)(a.dO + ((size_t)b * a.T + q) * a.lddo + h * DP + 16 * ks + 8 * h2) : u32x4{0, 0, 0, 0};
        dof[ks] = *(bf16x8*)&u;
    }
    const size_t sidx = ((size_t)b * a.H + h) * a.T + (qok ? q : 0);
    const float neglse = -a.lse2[sidx];
    // 1/keep is folded into the epilogue: dS' = P * (keep ? dP : 0  -  delta * keep_prob)
    const float dl = a.delta[sidx];  // already x keep_prob (delta kernel)
    f32x16 dq[G::DB];
#pragma unroll
    for (int d = 0; d < G::DB; ++d) zero16(dq[d]);
    float dsc = 0.f;
    const uint32_t T2 = (uint32_t)(a.T + 1) >> 1;
    const uint32_t dbase = a.adrop.key + ((uint32_t)(b * a.H + h) * T2 + ((uint32_t)q >> 1)) * ADROP_K1 + (uint32_t)(2 * h2) * ADROP_K2;
    const uint32_t sh_even = 16 * (q & 1), sh_odd = sh_even + 8;

    const int koff = (lane & 31) * G::RSTR + 8 * h2;
    const int toff = tr_lane_off(lane, G::RSTR);
    const int nt = (a.T + TR - 1) / TR;

    auto phase1 = [&](f32x16& s, f32x16& dp, int hf, int buf) {
        zero16(s);
        zero16(dp);
        const bf16_t* kp = &sK[buf][32 * hf * G::RSTR + koff];
        const bf16_t* vp = &sV[buf][32 * hf * G::RSTR + koff];
        bf16x8 kfr[G::KS], vfr[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) kfr[ks] = *(const bf16x8*)(kp + 16 * ks);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) vfr[ks] = *(const bf16x8*)(vp + 16 * ks);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) s = mfma32(kfr[ks], qf[ks], s);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) dp = mfma32(vfr[ks], dof[ks], dp);
    };
    // element-wise stage of one 32-key half: S^T, dP^T -> dS'^T (in s)
    auto softmax_half = [&](auto tail_tag, f32x16& s, const f32x16& dp, int kt, int hf) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bool keep[4] = {true, true, true, true};
            if constexpr (DROP)
                drop4(dbase + (uint32_t)(32 * kt + 16 * hf + 4 * g) * ADROP_K2, ADROP_K2, sh_even, sh_odd, attn_tile_thresh(a.adrop, (uint32_t)(b * a.H + h), (uint32_t)(a.T + 31) >> 5, (uint32_t)q >> 5, (uint32_t)(2 * kt + hf)), keep);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                float p = fast_exp2(fmaf(s[r], c, neglse));
                if constexpr (TAIL || DIAG) {
                    const int key = TR * kt + 32 * hf + acc_row(r, lane);
                    bool dead = false;
                    if constexpr (TAIL) dead = key >= a.T;
                    if constexpr (DIAG) dead = dead || key == q;
                    p = dead ? 0.f : p;
                }
                const float gg = keep[j] ? dp[r] : 0.f;
                const float ds = p * (gg - dl);
                if constexpr (DIAG) dsc = fmaf(ds, s[r], dsc);
                s[r] = ds;
            }
        }
    };
    auto tr_load = [&](bf16x8 (&tfr)[2 * G::DB], int hf, int buf) {
        const bf16_t* tp = &sK[buf][32 * hf * G::RSTR + toff];
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            tfr[2 * d] = tr_frag<G::RSTR>(tp, 0, 32 * d);
            tfr[2 * d + 1] = tr_frag<G::RSTR>(tp, 16, 32 * d);
        }
    };
    auto phase3 = [&](const bf16x8 (&tfr)[2 * G::DB], const f32x16& s) {
        const bf16x8 b0 = acc_to_b(s, 0), b1 = acc_to_b(s, 1);
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            dq[d] = mfma32(tfr[2 * d], b0, dq[d]);
            dq[d] = mfma32(tfr[2 * d + 1], b1, dq[d]);
        }
    };
    // same MFMA / VALU interleave as the dK/dV kernel (see there)
    auto tile = [&](auto tail_tag, int kt, int buf) {
        f32x16 s0, dp0, s1, dp1;
        bf16x8 tfr[2 * G::DB];
        phase1(s0, dp0, 0, buf);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * G::KS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * G::KS, 0);
        __builtin_amdgcn_sched_barrier(0);
        phase1(s1, dp1, 1, buf);
        softmax_half(tail_tag, s0, dp0, kt, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * G::KS, 1);
#pragma unroll
        for (int i = 0; i < 2 * G::KS; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        tr_load(tfr, 0, buf);
        phase3(tfr, s0);
        softmax_half(tail_tag, s1, dp1, kt, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * G::DB, 2);
#pragma unroll
        for (int i = 0; i < 2 * G::DB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
            __builtin_amdgcn_sched_group_barrier(0x002, 16, 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        tr_load(tfr, 1, buf);
        phase3(tfr, s1);
        __builtin_amdgcn_sched_barrier(0);
    };

    dma.issue(kbase, 0, a.T, sK[0]);
    dma.issue(vbase, 0, a.T, sV[0]);
    touch(qf);
    touch(dof);
    touch(neglse);
    touch(dl);
    touch(c);
    dma_wait_and_barrier();
    for (int kt = 0; kt < nt - 1; ++kt) {
        const int buf = kt & 1;
        dma.issue(kbase, TR * (kt + 1), a.T, sK[buf ^ 1]);
        dma.issue(vbase, TR * (kt + 1), a.T, sV[buf ^ 1]);
        tile(std::false_type{}, kt, buf);
        dma_wait_and_barrier();
    }
    tile(std::true_type{}, nt - 1, (nt - 1) & 1);

    const float kfac = DROP ? a.adrop.inv_keep : 1.0f;
    if constexpr (DIAG) {
        if (a.dscale) {
            const float tot = wave_sum(qok ? dsc : 0.f) * kfac;
            if (lane == 0) atomicAdd(&a.dscale[a.scale_per_head ? h : 0], tot);
        }
    }
    if (qok) {
        const float f = sc * kfac;
        bf16_t* orow = a.dqkv + ((size_t)b * a.T + q) * a.lddqkv + h * DP;
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                bf16x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (bf16_t)(dq[d][4 * rq + j] * f);
                *(bf16x4*)(orow + 32 * d + 8 * rq + 4 * h2) = w;
            }
    }
}

// ------------------------------------------------------------------------------------------
template <int DP, bool DROP, bool DIAG>
DEVFN void attn_bwd_dkv_body(const AttnArgs& a, int bid, int nblk, BwdLds<DP>& lds) {
    using G = Geo<DP>;
    constexpr int TR = BWD_TR;  // 64-query tiles = two independent 32-query halves (see the dQ kernel)
    auto& sQ = lds.a;
    auto& sD = lds.b;
    auto& sL = lds.l;
    auto& sDl = lds.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rb, h, b;
    decode_block(a, bid, nblk, rb, h, b);
    const int key = rb * 128 + 32 * wave + (lane & 31);
    TileDma<DP, G::RSTR, TR> dma, dmaD;
    dma.init(lane, wave, a.ldqkv);
    dmaD.init(lane, wave, a.lddo);
    const int h2 = lane >> 5;
    const int HD = a.H * DP;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    const bf16_t* qbase = qkv_b + h * DP;
    const bf16_t* dobase = a.dO + (size_t)b * a.T * a.lddo + h * DP;
    const float* lbase = a.lse2 + ((size_t)b * a.H + h) * a.T;
    const float* dbase_ = a.delta + ((size_t)b * a.H + h) * a.T;
    const float sc = a.scale[a.scale_per_head ? h : 0];
    const float c = sc * LOG2E;
    const bool kok = key < a.T;

    bf16x8 kf[G::KS], vf[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        const bf16_t* rowp = qkv_b + (size_t)key * a.ldqkv + h * DP + 16 * ks + 8 * h2;
        u32x4 t = kok ? *(const u32x4*)(rowp + HD) : u32x4{0, 0, 0, 0};
        kf[ks] = *(bf16x8*)&t;
        u32x4 u = kok ? *(const u32x4*)(rowp + 2 * HD) : u32x4{0, 0, 0, 0};
        vf[ks] = *(bf16x8*)&u;
    }
    f32x16 dk[G::DB], dv[G::DB];
#pragma unroll
    for (int d = 0; d < G::DB; ++d) {
        zero16(dk[d]);
        zero16(dv[d]);
    }
    // dropout: lane-fixed coordinate = key (column), varying = query row
    const uint32_t T2 = (uint32_t)(a.T + 1) >> 1;
    const uint32_t dbase = a.adrop.key + ((uint32_t)(b * a.H + h) * T2 + (uint32_t)(2 * h2)) * ADROP_K1 + ((uint32_t)key >> 1) * ADROP_K2;
    const uint32_t sh_even = 8 * (key & 1), sh_odd = sh_even + 16;

    const int roff = (lane & 31) * G::RSTR + 8 * h2;
    const int toff = tr_lane_off(lane, G::RSTR);
    const int nt = (a.T + TR - 1) / TR;
    // lse2 and delta (pre-scaled by keep_prob in the delta kernel) of the tile's 64 queries go to LDS by DMA too
    // (an ordinary load inside the loop would make hipcc drain the whole vmcnt queue at its first use)
    auto stage = [&](int t, int buf) {
        dma.issue(qbase, TR * t, a.T, sQ[buf]);
        dmaD.issue(dobase, TR * t, a.T, sD[buf]);
        const int qq = min(TR * t + lane, a.T - 1);
        if (wave == 0) lds_dma4(lbase + qq, sL[buf]);
        if (wave == 1) lds_dma4(dbase_ + qq, sDl[buf]);
    };
    // element-wise stage of one 32-query half: S, dP accumulators -> dropped P (dp) and dS' (s), in place
    auto softmax_half = [&](auto tail_tag, f32x16& s, f32x16& dp, int qt, int hf, int buf) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 nl = *(const f32x4*)(&sL[buf][32 * hf + 8 * g + 4 * h2]);
            const f32x4 dl = *(const f32x4*)(&sDl[buf][32 * hf + 8 * g + 4 * h2]);
            bool keep[4] = {true, true, true, true};
            if constexpr (DROP)
                drop4(dbase + (uint32_t)(32 * qt + 16 * hf + 4 * g) * ADROP_K1, ADROP_K1, sh_even, sh_odd, attn_tile_thresh(a.adrop, (uint32_t)(b * a.H + h), (uint32_t)(a.T + 31) >> 5, (uint32_t)(2 * qt + hf), (uint32_t)key >> 5), keep);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                float p = fast_exp2(fmaf(s[r], c, -nl[j]));
                if constexpr (TAIL || DIAG) {
                    const int qq = TR * qt + 32 * hf + acc_row(r, lane);
                    bool dead = false;
                    if constexpr (TAIL) dead = qq >= a.T;
                    if constexpr (DIAG) dead = dead || qq == key;
                    p = dead ? 0.f : p;
                }
                const float gg = keep[j] ? dp[r] : 0.f;
                dp[r] = keep[j] ? p : 0.f;     // dropped P (x 1/keep in the epilogue) -> dV
                s[r] = p * (gg - dl[j]);        // dS' (x scale/keep in the epilogue)   -> dK
            }
        }
    };
    auto phase1 = [&](f32x16& s, f32x16& dp, int hf, int buf) {
        zero16(s);
        zero16(dp);
        const bf16_t* qp = &sQ[buf][32 * hf * G::RSTR + roff];
        const bf16_t* dop = &sD[buf][32 * hf * G::RSTR + roff];
        bf16x8 qfr[G::KS], dfr[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qfr[ks] = *(const bf16x8*)(qp + 16 * ks);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) dfr[ks] = *(const bf16x8*)(dop + 16 * ks);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) s = mfma32(qfr[ks], kf[ks], s);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) dp = mfma32(dfr[ks], vf[ks], dp);
    };
    auto tr_load_d = [&](bf16x8 (&tdf)[2 * G::DB], int hf, int buf) {
        const bf16_t* td = &sD[buf][32 * hf * G::RSTR + toff];
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            tdf[2 * d] = tr_frag<G::RSTR>(td, 0, 32 * d);
            tdf[2 * d + 1] = tr_frag<G::RSTR>(td, 16, 32 * d);
        }
    };
    auto tr_load_q = [&](bf16x8 (&tqf)[2 * G::DB], int hf, int buf) {
        const bf16_t* tq = &sQ[buf][32 * hf * G::RSTR + toff];
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            tqf[2 * d] = tr_frag<G::RSTR>(tq, 0, 32 * d);
            tqf[2 * d + 1] = tr_frag<G::RSTR>(tq, 16, 32 * d);
        }
    };
    auto phase3_v = [&](const bf16x8 (&tdf)[2 * G::DB], const f32x16& dp) {
        const bf16x8 p0 = acc_to_b_pk(dp, 0), p1 = acc_to_b_pk(dp, 1);
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            dv[d] = mfma32(tdf[2 * d], p0, dv[d]);
            dv[d] = mfma32(tdf[2 * d + 1], p1, dv[d]);
        }
    };
    auto phase3_k = [&](auto tail_tag, const bf16x8 (&tqf)[2 * G::DB], const f32x16& s, int qt, int hf) {
        const bf16x8 s0 = acc_to_b_pk(s, 0), s1 = acc_to_b_pk(s, 1);
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            dk[d] = mfma32(tqf[2 * d], s0, dk[d]);
            dk[d] = mfma32(tqf[2 * d + 1], s1, dk[d]);
        }
    };
    // One wave per SIMD issues in order, and an MFMA keeps the issue port only 8 of its 32 cycles: the
    // element-wise work of one half is therefore INTERLEAVED, instruction by instruction, with the MFMAs of
    // the other half (sched_group_barrier: 1 MFMA + a few VALU), instead of alternating MFMA-only and VALU-only
    // stretches. Fragment loads are batched so that the live set stays inside the 512-register budget.
    auto tile = [&](auto tail_tag, int qt, int buf) {
        f32x16 s0, dp0, s1, dp1;
        bf16x8 tdf[2 * G::DB], tqf[2 * G::DB];
        phase1(s0, dp0, 0, buf);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * G::KS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * G::KS, 0);
        __builtin_amdgcn_sched_barrier(0);
        // group A: phase 1 of half 1 (MFMA) interleaved with the element-wise stage of half 0 (VALU)
        phase1(s1, dp1, 1, buf);
        softmax_half(tail_tag, s0, dp0, qt, 0, buf);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * G::KS + 8, 1);
#pragma unroll
        for (int i = 0; i < 2 * G::KS; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x002, 9, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // group B: phase 3 of half 0 (MFMA) interleaved with the element-wise stage of half 1 (VALU)
        tr_load_d(tdf, 0, buf);
        tr_load_q(tqf, 0, buf);
        phase3_v(tdf, dp0);
        phase3_k(tail_tag, tqf, s0, qt, 0);
        softmax_half(tail_tag, s1, dp1, qt, 1, buf);
        __builtin_amdgcn_sched_group_barrier(0x100, 8 * G::DB + 8, 2);
#pragma unroll
        for (int i = 0; i < 4 * G::DB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
            __builtin_amdgcn_sched_group_barrier(0x002, 9, 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        tr_load_d(tdf, 1, buf);
        tr_load_q(tqf, 1, buf);
        phase3_v(tdf, dp1);
        phase3_k(tail_tag, tqf, s1, qt, 1);
        __builtin_amdgcn_sched_barrier(0);
    };

    stage(0, 0);
    touch(kf);
    touch(vf);
    touch(c);
    dma_wait_and_barrier();
    for (int qt = 0; qt < nt - 1; ++qt) {
        const int buf = qt & 1;
        stage(qt + 1, buf ^ 1);
        tile(std::false_type{}, qt, buf);
        dma_wait_and_barrier();
    }
    tile(std::true_type{}, nt - 1, (nt - 1) & 1);

    if (kok) {
        const float kfac = DROP ? a.adrop.inv_keep : 1.0f;
        const float fk = sc * kfac;
        bf16_t* orow = a.dqkv + ((size_t)b * a.T + key) * a.lddqkv + h * DP;
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int rq4 = 0; rq4 < 4; ++rq4) {
                bf16x4 wk, wv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wk[j] = (bf16_t)(dk[d][4 * rq4 + j] * fk);
                    wv[j] = (bf16_t)(dv[d][4 * rq4 + j] * kfac);
                }
                *(bf16x4*)(orow + HD + 32 * d + 8 * rq4 + 4 * h2) = wk;
                *(bf16x4*)(orow + 2 * HD + 32 * d + 8 * rq4 + 4 * h2) = wv;
            }
    }
}

// How the forward kernel covers the queries of an (image, head): `full` 256-query row blocks + `half` 128-query ones (decode_fwd).
// Natural cover: floor(T / 256) full blocks and the remainder as one more full block, or as ONE half block when it is at most 128 queries
// (T = 1654: 6 + 1). Small launches (at most 1152 workgroups that way) may do better with fewer full and more half blocks: the launch is a few
// rounds of one-workgroup-per-CU workgroups and ends when the last CU does; candidates (F, ceil((T - 256 F) / 128)) are list-scheduled,
// longest first, on the 32 CUs of the fullest XCD with a half block at 0.67 of a full one (measured: the ragged block of round 3), and the
// shortest makespan wins when it beats the natural cover by 5 % (ties: fewer half blocks - each stages the whole K / V for half the queries).
// 14 images x 4 heads: natural 6 + 1 = two rounds; 4 + 5 = 1.67 (attn_fwd 138 -> 128 us per launch, profiles/r05_small_launch_experiments.txt);
// 16 and 28 images keep the natural cover (16: 2.0 either way; 28: 3.0 with LPT order against 3.34). V1T_FWD_SPLIT=0 (dev): natural cover.
static void choose_fwd_split_uncached(int B, int H, int T, int& F, int& Hh, int& lpt);
static void choose_fwd_split(int B, int H, int T, int& F, int& Hh, int& lpt) {
    // one decision per launch shape and thread (the list scheduling below costs the host ~40 us, four launches per step)
    thread_local int key[3] = {-1, -1, -1}, val[3] = {0, 0, 0};
    if (key[0] != B || key[1] != H || key[2] != T) {
        choose_fwd_split_uncached(B, H, T, val[0], val[1], val[2]);
        key[0] = B; key[1] = H; key[2] = T;
    }
    F = val[0]; Hh = val[1]; lpt = val[2];
}
static void choose_fwd_split_uncached(int B, int H, int T, int& F, int& Hh, int& lpt) {
    const int rem = T % 256;
    const int F0 = T / 256 + (rem > 128 ? 1 : 0), H0 = (rem >= 1 && rem <= 128) ? 1 : 0;
    F = F0; Hh = H0;
    const int nblk0 = B * H * (F0 + H0);
    static const int lpt_max = dev_env("V1T_FWD_LPT_MAX") ? atoi(dev_env("V1T_FWD_LPT_MAX")) : 1152;  // dev (A/B)
    lpt = (H0 && nblk0 >= 640 && nblk0 <= lpt_max) ? 1 : 0;  // measured (round 5): helps a 28-image launch, hurts at 14 and at 112 images
    static const bool allow = !(dev_env("V1T_FWD_SPLIT") && !atoi(dev_env("V1T_FWD_SPLIT")));
    if (!allow || nblk0 > lpt_max || T <= 256) return;
    const int nbh = (B * H + 7) / 8;  // (image, head) pairs of the fullest XCD
    auto makespan = [&](int f, int hh) {
        float cu[32];
        for (float& x : cu) x = 0.f;
        auto put = [&](int n, float cost) {
            for (int i = 0; i < n; ++i) {
                int m = 0;
                for (int c = 1; c < 32; ++c) if (cu[c] < cu[m]) m = c;
                cu[m] += cost;
            }
        };
        put(nbh * f, 1.0f);
        put(nbh * hh, 0.67f);
        float ms = 0.f;
        for (float x : cu) ms = std::max(ms, x);
        return ms;
    };
    float best = makespan(F0, H0);
    for (int f = F0 - 1; f >= std::max(0, F0 - 4); --f) {
        const int hh = (T - 256 * f + 127) / 128;
        const float ms = makespan(f, hh);
        if (ms < 0.95f * best) { best = ms; F = f; Hh = hh; lpt = 1; }
    }
}

template <int DP, bool DROP, bool DIAG>
int launch_fwd_t(const AttnArgs& a_in, hipStream_t s) {
    AttnArgs a = a_in;
    dim3 grid(((a.T + 32 * FWD_WAVES - 1) / (32 * FWD_WAVES)) * a.H * a.B);  // the natural cover (the experiment kernels below)
    choose_fwd_split(a.B, a.H, a.T, a.fwd_full, a.fwd_half, a.fwd_lpt);
    const dim3 grid_main((a.fwd_full + a.fwd_half) * a.H * a.B);
    prof_begin(PROF_ATTN_FWD, s);
#ifdef V1T_EXPERIMENTS
    static const bool v3 = dev_env("V1T_ATTN_FWD_V3") != nullptr;  // dev: the 16-wave S / PV role kernel
    if constexpr (DP >= 128 && !DIAG) {
        if (v3) {
            hipLaunchKernelGGL((attn_fwd3_kernel<DP, DROP>), grid, dim3(128 * F3_PAIRS), 0, s, a);
            prof_end(PROF_ATTN_FWD, s);
            return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
        }
    }
    static const bool stagger = dev_env("V1T_ATTN_FWD_STAGGER") != nullptr;  // dev: A/B of the rotated second half
    if constexpr (DP >= 128 && !DIAG) {
        if (stagger) {
            hipLaunchKernelGGL((attn_fwd_kernel<DP, DROP, DIAG, true>), grid_main, dim3(64 * FWD_WAVES), 0, s, a);
            prof_end(PROF_ATTN_FWD, s);
            return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
        }
    }
#else
    (void)grid;
#endif
    hipLaunchKernelGGL((attn_fwd_kernel<DP, DROP, DIAG>), grid_main, dim3(64 * FWD_WAVES), 0, s, a);
    prof_end(PROF_ATTN_FWD, s);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
template <int DP, bool DROP, bool DIAG>
__global__ __launch_bounds__(256, (DP >= 128 ? 1 : 2)) void attn_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) BwdLds<DP> lds;
    attn_bwd_dq_body<DP, DROP, DIAG>(a, blockIdx.x, gridDim.x, lds);
}
template <int DP, bool DROP, bool DIAG>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) BwdLds<DP> lds;
    attn_bwd_dkv_body<DP, DROP, DIAG>(a, blockIdx.x, gridDim.x, lds);
}
// Both backward bodies in ONE launch (DP >= 128, where both run one workgroup per CU): n = 832 workgroups of each
// on 256 CUs are 3.25 rounds, i.e. 4 per kernel (8 in all, the last of each a quarter full); together they are
// 6.5 rounds of mixed length, the longer dK/dV workgroups dispatched first. nblk8 = n rounded up to 8 keeps
// blockIdx % 8 (the XCD) equal to the body-local id % 8 that xcd_remap() assumes.
template <int DP, bool DROP, bool DIAG>
__global__ __launch_bounds__(256, 1) void attn_bwd_fused_kernel(AttnArgs a, int nblk, int nblk8) {
    __shared__ __attribute__((aligned(16))) BwdLds<DP> lds;
    const int bid = blockIdx.x;
    if (bid < nblk8) {
        if (bid < nblk) attn_bwd_dkv_body<DP, DROP, DIAG>(a, bid, nblk, lds);
    } else {
        attn_bwd_dq_body<DP, DROP, DIAG>(a, bid - nblk8, nblk, lds);
    }
}

// ------------------------------------------------------------------------------------------
// Producer / consumer backward (default path: DP >= 128, no LSA diagonal): dK / dV + materialised dS', then dQ = dS' . K.
//
// Why: in the one-wave-per-SIMD kernels above every MFMA of a 512-register kernel is selected in AGPR form, so each
// S / dP element is moved AGPR -> VGPR (and P / dS' operands back) around the element-wise stage: of ~1100 vector
// instructions per 64 x 32 tile only ~570 are arithmetic (ISA count), and a lone wave issues one vector instruction per
// ~4.5 cycles whatever its kind (tools/microbench/valu_issue_cost.hip) - 35 % matrix-pipe occupancy. Here a workgroup is
// 8 waves with <= 256 registers each (two per SIMD, all MFMAs in VGPR form, no accumulator copies), split by ROLE:
//   waves 0-3 ("producers"): 32 keys each; K / V fragments stay in registers; per 32-query block S = Q K^T and dP = dO V^T
//       (20 MFMAs, key on the lane, accumulators initialised with the row constants -lse2/c and -delta so that
//       P = exp2(c S') needs no subtraction), the element-wise stage, and the bf16 P / dS' fragments - exactly the
//       B operands of the next two products - handed to the partner wave through LDS (lane l writes 4 x 16 B, lane l of
//       the partner reads them back: no transposition);
//   waves 4-7 ("consumers", SIMD partners of 0-3): dV^T += dO^T P, dK^T += Q^T dS' (20 MFMAs, 160 accumulator
//       registers), the dS' block stored to HBM as two 16-B-per-lane instructions (the old kernel: 16 two-byte stores),
//       and its half of the LDS-DMA staging (Q / dO tiles of 32 rows into a 5-slot ring, three tiles ahead, counted vmcnt).
// One barrier per 32-query block; the consumer works one block behind the producer. The matrix pipe of a SIMD then sees
// 40 MFMAs per block from two waves whose element-wise / LDS phases overlap the partner's MFMAs.
constexpr int B2_SLOTS = 5;  // tiles live at step i: i - 1 (consumer), i (waits for the consumer), i + 1 (producer), i + 2 (landing), i + 3 (being issued)
template <int DP>
struct Bwd2Lds {
    using Dma = TileDma<DP, Geo<DP>::RSTR, 32, 4>;
    bf16_t q[B2_SLOTS][Dma::LDS_ELEMS];
    bf16_t d[B2_SLOTS][Dma::LDS_ELEMS];
    float rc[B2_SLOTS][64];   // [0, 32): -lse2 of the slot's queries, [32, 64): -keep_prob * delta
    u32x4 hand[2][4][4][64];  // [block parity][pair][P k-step 0, P k-step 1, dS' k-step 0, dS' k-step 1][lane]
};

// row constants for the kernel below, padded to 32-query blocks: pad rows get nlse = -1e30 (P = 0), ndelta = 0.
// 16 lanes per (row, head) segment, as attn_delta_kernel.
template <int DP>
__global__ __launch_bounds__(256) void attn_delta2_kernel(AttnArgs a, float* nlse, float* ndelta, int TPQ) {
    constexpr int CH = DP / 8;
    const int sub = threadIdx.x & 15;
    const long long seg = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const long long total = (long long)a.B * TPQ * a.H;
    const bool ok = seg < total;
    const int h = ok ? (int)(seg % a.H) : 0;
    const long long rowp = ok ? seg / a.H : 0;  // padded row index b * TPQ + t
    const long long b_ = rowp / TPQ;
    const int t = (int)(rowp % TPQ);
    const bool real = ok && t < a.T;
    const long long row = b_ * a.T + (real ? t : 0);
    const bf16_t* po = a.o + (size_t)row * a.ldo + h * DP;
    const bf16_t* pd = a.dO + (size_t)row * a.lddo + h * DP;
    float acc = 0.f;
    if (real) {
#pragma unroll
        for (int c = sub; c < CH; c += 16) {
            const bf16x8 x = *(const bf16x8*)(po + 8 * c);
            const bf16x8 y = *(const bf16x8*)(pd + 8 * c);
            if (a.o_f16) {  // kernel-uniform
                const f16x8 xh = __builtin_bit_cast(f16x8, x);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)xh[j] * (float)y[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)x[j] * (float)y[j];
            }
        }
    }
    acc += __shfl_xor(acc, 8);
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 1);
    if (ok && sub == 0) {
        const size_t o = ((size_t)b_ * a.H + h) * TPQ + t;
        nlse[o] = real ? -a.lse2[((size_t)b_ * a.H + h) * a.T + t] : NEG_BIG;
        ndelta[o] = real ? -acc * a.adrop.keep_prob : 0.f;
    }
}

__global__ __launch_bounds__(256) void attn_rc_pad_kernel(float* nlse, float* ndelta, int BH, int T, int TPQ) {
    const int npad = TPQ - T;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BH * npad) return;
    const size_t o = (size_t)(i / npad) * TPQ + T + i % npad;
    nlse[o] = NEG_BIG;
    ndelta[o] = 0.f;
}

DEVFN int b2_slot(int t) { return (t + 3) - B2_SLOTS * ((t + 3) / B2_SLOTS); }  // tiles 0, 1 in slots 3, 4: slots 0-2 hold K / V images during the prologue

#ifdef V1T_KCLK
__device__ unsigned long long g_kclk[4];  // dev: shader cycles / 100 MHz ticks of one workgroup of the kernel below
#endif
template <int DP, bool DROP>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv2_kernel(AttnArgs a) {
    using G = Geo<DP>;
    using Dma = typename Bwd2Lds<DP>::Dma;
    static_assert(Dma::PW == 3 && Dma::NINST == 11, "three DMA pieces per wave and tile, the eleventh half empty: head dim 160 (launch_bwd_t routes the others elsewhere)");
    __shared__ __attribute__((aligned(16))) Bwd2Lds<DP> lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef V1T_KCLK
    const unsigned long long kclk_c0 = __builtin_amdgcn_s_memtime(), kclk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    KS_DECL;
    const int pw = wave & 3;  // pair: producer pw and consumer pw + 4 own keys [32 pw, 32 pw + 32) of the 128-key block
    int rb, h, b;
    decode_block(a, blockIdx.x, gridDim.x, rb, h, b);
    const int key = rb * 128 + 32 * pw + (lane & 31);
    const int h2 = lane >> 5;
    const int HD = a.H * DP;
    const bool kok = key < a.T;
    const int nq = (a.T + 31) / 32;        // 32-query blocks
    const int TPQ = 32 * nq;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    const size_t bh = (size_t)b * a.H + h;

    // ---- LDS-DMA staging of the 32-row Q / dO tiles, one piece (1 KiB, one instruction) per call. Wave pair pw owns pieces
    // 3 pw .. 3 pw + 2 of each tile (TileDma's split): the consumer stages its Q pieces, the producer its dO pieces, so both
    // roles carry half of the issue cost (~100 cycles a piece). The third piece of pair 3 does not exist (10.5 pieces per
    // tile): its consumer stages the row constants instead, its producer repeats piece 1 - every wave issues exactly 3
    // operations per tile and the end-of-step vmcnt is one immediate.
    const bool last = pw == 3;  // wave-uniform
    auto lane_off = [&](int i, int ld, int max_row) {  // byte offset of this lane's 16 B of piece i inside the tile's rows
        const int p = 64 * (pw * Dma::PW + i) + lane;
        const int r = min(perm_row(min(p / Dma::CPR, 31)), max_row), cc = min(p % Dma::CPR, DP / 8 - 1);  // LDS position -> the row it holds
        return (unsigned)((r * ld + 8 * cc) * 2);
    };
    const bf16_t* const img = wave < 4 ? a.dO + (size_t)b * a.T * a.lddo + h * DP : qkv_b + h * DP;  // this wave's tile source
    const int ld = wave < 4 ? a.lddo : a.ldqkv;
    unsigned voff[3];  // (biased: see stage())
#pragma unroll
    for (int i = 0; i < 3; ++i) voff[i] = lane_off(i, ld, 31) + 2048u - 1024u * i;
    const float* rcg = (const float*)(a.ds + attn_ds_elems(a.B, a.H, a.T)) + bh * TPQ;  // nlse; ndelta B*H*TPQ floats behind it
    const unsigned rc_voff = (unsigned)(((lane & 31) + (h2 ? attn_rc_floats(a.B, a.H, a.T) : 0)) * 4);
    // LDS destinations as 32-bit byte addresses computed from one base (a generic-pointer form costs a null check and 64-bit arithmetic per
    // operation - for lds.rc[slot] even a 64-bit division - in scalar instructions, which take the wave's issue slots like vector ones); the
    // two or three pieces of a tile share ONE M0 set-up: the instruction offset advances the LDS and the global address alike, so piece i
    // carries offset 1024 i and a lane offset of (its own) - 1024 i + DBIAS against a base lowered by DBIAS
    constexpr unsigned DBIAS = 2048, SLOT_BYTES = Dma::LDS_ELEMS * 2;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&lds);
    const unsigned tile0 = lds_base + (unsigned)(wave < 4 ? offsetof(Bwd2Lds<DP>, d) : offsetof(Bwd2Lds<DP>, q)) + 1024u * (3 * pw);  // slot 0, this wave's first piece
    const unsigned rc0 = lds_base + (unsigned)offsetof(Bwd2Lds<DP>, rc);
    auto stage = [&](int t) {  // this wave's three operations of tile t
        const unsigned slot = (unsigned)b2_slot(t);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(tile0 + slot * SLOT_BYTES);
        const char* src = (const char*)(img + (size_t)32 * t * ld) - DBIAS;
        unsigned v[3] = {voff[0], voff[1], voff[2]};
        if (32 * t + 32 > a.T) {
            asm volatile("; ragged tile: rows beyond T are clamped to T - 1 (finite data; P = 0 there)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 3; ++i) v[i] = lane_off(i, ld, a.T - 1 - 32 * t) + DBIAS - 1024u * i;
        }
        unsigned keep;
        if (!last) {
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\tglobal_load_lds_dwordx4 %3, %5 offset:1024\n\t"
                         "global_load_lds_dwordx4 %4, %5 offset:2048\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(m0v), "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(src) : "memory");
        } else if (wave < 4) {  // pair 3 has no third piece: its producer repeats piece 1 (every wave issues exactly three operations per tile)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\tglobal_load_lds_dwordx4 %3, %4 offset:1024\n\t"
                         "global_load_lds_dwordx4 %3, %4 offset:1024\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(m0v), "v"(v[0]), "v"(v[1]), "s"(src) : "memory");
        } else {  // ... its consumer stages the row constants: lanes 0-31: -lse2 -> rc[0..31], lanes 32-63: -keep_prob delta -> rc[32..63]
            const unsigned rcv = __builtin_amdgcn_readfirstlane(rc0 + slot * 256u);
            const float* rsrc = rcg + 32 * t;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\tglobal_load_lds_dwordx4 %3, %4 offset:1024\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %6, %7\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(m0v), "v"(v[0]), "v"(v[1]), "s"(src), "s"(rcv), "v"(rc_voff), "s"(rsrc) : "memory");
        }
    };

    // ---- prologue: the workgroup's K / V rows (128 keys) as eight 32-row tile images in ring slots 0-2 and the hand-off
    // area (pair p's keys -> q[p] = K, d[p] = V; pair 3: hand), all unused until tiles 2 .. arrive, from which the producers read their MFMA fragments. Whole rows through registers: consumer p + 4
    // loads K tile p, producer p V tile p, ten 16-B-per-lane loads of consecutive chunks each (full lines), then ten 16-B LDS
    // writes. History: (1) the fragments straight from HBM (16 B per lane, two lanes per row, 80 instructions per workgroup) held
    // the vector memory path for 10 k cycles with every other first access of the workgroup queued behind them - 18 k of a
    // 155 k-cycle workgroup passed before the first step; (2) LDS-DMA like the Q / dO tiles: 14 k, bound by that path's ~11 B per
    // cycle and CU (140 KB with tiles 0-2); (3) this: the loads use the ordinary path while the DMA path carries tiles 0-2.
    // Keys beyond T are clamped to row T - 1: finite data, their P / dS' columns are masked where they leave the workgroup.
    constexpr int KV_IT = (32 * (DP / 8) + 63) / 64;
    u32x4 kvreg[KV_IT];
    auto load_kv = [&]() {
        const bf16_t* src = qkv_b + h * DP + (wave < 4 ? 2 * HD : HD);
#pragma unroll
        for (int it = 0; it < KV_IT; ++it) {
            const int p = min(64 * it + lane, 32 * (DP / 8) - 1);
            const int r = min(rb * 128 + 32 * pw + p / (DP / 8), a.T - 1), cc = p % (DP / 8);
            kvreg[it] = *(const u32x4*)(src + (size_t)r * a.ldqkv + 8 * cc);
        }
    };
    static_assert(sizeof(lds.hand) >= 2 * 32 * G::RSTR * sizeof(bf16_t), "two tile images fit the hand-off area");
    auto kv_image = [&](bool v) -> bf16_t* { return last ? (bf16_t*)&lds.hand[0][0][0][0] + (v ? 32 * G::RSTR : 0) : (v ? lds.d[pw] : lds.q[pw]); };
    auto store_kv = [&]() {
        bf16_t* tile = kv_image(wave < 4);
#pragma unroll
        for (int it = 0; it < KV_IT; ++it) {
            const int p = 64 * it + lane;
            if (p < 32 * (DP / 8)) *(u32x4*)(tile + (p / (DP / 8)) * G::RSTR + 8 * (p % (DP / 8))) = kvreg[it];
        }
    };

    // ---- dropout of P (common.h): the lane's fixed coordinate is its key (column), the varying one the query row. A hash
    // word serves a 2 x 2 block of (query, key); the lane's two decisions of a word sit in the bytes key & 1 and 2 + (key & 1),
    // so after a shift by 8 (key & 1) both are decided by ONE 9-bit SWAR compare: ((w & 0x00FF00FF) | 0x01000100) - thr * 0x00010001
    // has bit 8 / bit 24 set iff byte 0 / byte 2 >= thr (keep): 8 words per block (row pairs (8 g + 4 h2 + 2 u, + 1), word index
    // 2 g + u), a decision then costs a 1-bit v_bfe_i32 and a v_and. The PRODUCER hashes them itself, one word in the slot that first
    // needs it. (Round 2 had the consumers do it and pass the words through LDS; since then the consumers have become the
    // critical path of a step - 2640 against 2000 cycles of own work - and the exchange cost 1.4 % of the backward.)
    const uint32_t T2 = (uint32_t)(a.T + 1) >> 1;
    const uint32_t dbase = a.adrop.key + ((uint32_t)bh * T2 + (uint32_t)(2 * h2)) * ADROP_K1 + ((uint32_t)key >> 1) * ADROP_K2;
    const uint32_t dshift = 8 * (key & 1);
    // the raw word shifted so that this key's two decisions are bytes 0 (even row) and 2 (odd row); the compare reads the byte itself
    // (v_cmp_ge_u32_sdwa): compare + select per element. (Rounds 2-3: a 9-bit SWAR preparation per word - and / or / sub - then a 1-bit v_bfe_i32
    // and a v_and per element; same decisions bit for bit, 24 vector instructions fewer per 32 x 32 block: backward pair 2519 / 2530 -> 2480 / 2484 us,
    // profiles/r04_attn_experiments.txt #6.)
    const uint32_t tile_nqb = (uint32_t)nq, tile_kb = (uint32_t)(rb * 4 + pw);  // per-tile byte threshold (common.h): wave-uniform, scalar arithmetic
    auto keep_word = [&](int blk, int wi) { return mix1(dbase + (uint32_t)(16 * blk + 4 * (wi >> 1) + (wi & 1)) * ADROP_K1) >> dshift; };

    if (wave < 4) {
        // ------------------------------------------------------------------ producer
        // (static priorities - s_setprio for either role - change nothing measurable here: A/B within 1 %)
        const float sc = a.scale[a.scale_per_head ? h : 0];
        const float c = sc * LOG2E;
        bf16x8 kf[G::KS], vf[G::KS];
        load_kv();
        stage(0);
        if (nq > 1) stage(1);
        KSP_MARK(0);
        store_kv();  // (waits for the loads only: the compiler counts its own; the DMA operations behind them stay in flight)
        KSP_MARK(1);
        lds_barrier();  // K / V images written
        KSP_MARK(2);
        const int roff_lin = (lane & 31) * G::RSTR + 8 * h2;            // K / V images: rows in order
        const int roff = perm_row(lane & 31) * G::RSTR + 8 * h2;        // Q / dO tiles: rows permuted (perm_row)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            kf[ks] = *(const bf16x8*)(kv_image(false) + roff_lin + 16 * ks);
            vf[ks] = *(const bf16x8*)(kv_image(true) + roff_lin + 16 * ks);
        }
        // Rounds 2-5 pre-multiplied K by c = scale log2(e) here (P = exp2(S') then costs no multiply per element). That re-rounds c k to
        // bf16: 2^-9 relative PER TERM of the exponent, i.e. an absolute error of ~1e-3 |c S| / sqrt(terms) on it - harmless while scores are
        // O(1), but the forward exponentiates c (q . k) with q . k exact in fp32, so at the scores of TRAINED weights (|c S| = 30-140 in log2
        // units) the recomputed P disagreed with the forward's by 3-10 % per element (round 6: gradients of the trained-regime golden at
        // 1.4-1.8 x the bound with dropout on, the extreme-score kernel test at 6 x). K stays as the QKV GEMM left it, the accumulator
        // starts from -lse2 / c and P = exp2(c S'): two multiplies per element in the producers, who wait ~600 cycles at every barrier anyway.
        const float inv_c = 1.0f / c;
        touch(kf);
        touch(vf);
        KSP_MARK(3);
        lds_barrier();  // every producer holds its fragments: slots 0-2 are free for tiles 2 ..
        if (nq > 2) stage(2);
        KSP_MARK(4);
        // Accumulators of S' = c Q K^T - lse2 and dP (- keep_prob delta) start from the row constants of the query block
        // (register r <-> query row 8 (r >> 2) + 4 h2 + (r & 3))
        auto init_rows = [&](int slot, f32x16& s, f32x16& dp, float (&nd)[16]) {
            const float* rc = lds.rc[slot];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 nl = *(const f32x4*)(rc + 8 * g + 4 * h2);
                const f32x4 dl = *(const f32x4*)(rc + 32 + 8 * g + 4 * h2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[4 * g + j] = nl[j] * inv_c;
                    if constexpr (DROP) {
                        nd[4 * g + j] = dl[j];
                        dp[4 * g + j] = 0.f;
                    } else {
                        dp[4 * g + j] = dl[j];
                    }
                }
            }
        };
        // One step = 2 KS "slots", each ONE MFMA of block i + 1's chains (S' over the head dimension, then dP) plus a slice of
        // block i's element-wise stage, fenced by sched_barrier(0): the wave issues in order and an MFMA holds the issue port
        // 8 of its 32 cycles, so the slice's vector instructions fill the rest. (sched_group_barrier pipelines did not
        // survive hipcc's scheduler here: it put the 20 MFMAs first and the ~190 vector instructions behind them.)
        // 20 slices dealt over the slots: 16 elements (hash words of a 4-row group with its first element, exp2, dropout, dS')
        // and 4 packed operand fragments to the partner wave. The chains' operand fragments are read LA slots ahead.
        constexpr int NSLOT = 2 * G::KS, LA = 3;
        KP_DECL;
        auto step = [&](auto issued_tag, int i, f32x16& s, f32x16& dp, float (&nd)[16], f32x16& s2, f32x16& dp2, float (&nd2)[16]) {
            constexpr bool ISSUED = decltype(issued_tag)::value;
            KP_STAMP(0);
            KS_MARK(0);
#ifndef V1T_B2_NODMA  // dev ablations (timing only, garbage results): V1T_B2_NODMA / NOSTORE / NOVALU / NOKEEP / NOFRAG
            if constexpr (ISSUED) stage(i + 3);
#endif
            KS_MARK(1);
            const int slot = b2_slot(i + 1);  // block nq does not exist: a stale tile, results never used
            init_rows(slot, s2, dp2, nd2);
            // per-step base addresses kept opaque: otherwise the 2 KS fragment addresses are hoisted out of the loop as 2 KS
            // registers and re-based with one vector instruction each per step instead of being immediates
            unsigned qa = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&lds.q[slot][roff];
            unsigned da = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&lds.d[slot][roff];
            asm volatile("" : "+v"(qa), "+v"(da));
            auto frag = [&](int m) {
#ifdef V1T_B2_NOFRAG
                return kf[(m + 1) % G::KS];
#endif
                const unsigned ad = (m < G::KS ? qa : da) + 32u * (unsigned)(m % G::KS);
                return *(const __attribute__((address_space(3))) bf16x8*)(uintptr_t)ad;
            };
            bf16x8 fr[NSLOT];
#pragma unroll
            for (int m = 0; m < LA; ++m) fr[m] = frag(m);
            u32x4 kw[2] = {};
            uint32_t dthr8 = 0;
            if constexpr (DROP) dthr8 = attn_tile_thresh(a.adrop, (uint32_t)bh, tile_nqb, (uint32_t)i, tile_kb);
            u32x4* hb = lds.hand[i & 1][pw][0];
            KP_STAMP(1);
            KS_MARK(2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NSLOT; ++m) {
                if (m == NSLOT / 2) { KP_STAMP(2); }
                if (m + LA < NSLOT) fr[m + LA] = frag(m + LA);
                if (m < G::KS) s2 = mfma32(fr[m], kf[m], s2);
                else dp2 = mfma32(fr[m], vf[m - G::KS], dp2);
#pragma unroll
                for (int it = m * 20 / NSLOT; it < (m + 1) * 20 / NSLOT; ++it) {  // 16 element slices + 4 fragment slices over the slots
                    if (it < 16) {
                        const int r = it, g = it >> 2, j = it & 3;
#ifdef V1T_B2_NOVALU
                        (void)g; (void)j;
                        asm volatile("" : "+v"(s[r]), "+v"(dp[r]));
                        continue;
#endif
                        const float p = fast_exp2(s[r] * c);
                        if constexpr (DROP) {
                            const int wi = 2 * g + (j >> 1);  // keep word of this row pair; bit 8 / 24: even / odd row
                            // the producer hashes its own keep words (one per row pair, in the slot that first needs it): the
                            // consumers are the critical path of a step (ablation: their 8 hash words cost the kernel 7 %) while
                            // the producers wait ~600 cycles at every barrier
                            if ((j & 1) == 0) kw[wi >> 2][wi & 3] = keep_word(i, wi);
                            float pd;
                            if ((j & 1) == 0)
                                asm("v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(pd) : "v"(kw[wi >> 2][wi & 3]), "s"(dthr8), "v"(p) : "vcc");
                            else
                                asm("v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(pd) : "v"(kw[wi >> 2][wi & 3]), "s"(dthr8), "v"(p) : "vcc");
                            dp[r] = fmaf(pd, dp[r], p * nd[r]);
                            s[r] = pd;
                        } else {
                            dp[r] = p * dp[r];
                            s[r] = p;
                        }
                    } else {
                        const int f = it - 16;  // P k-step 0, P k-step 1, dS' k-step 0, dS' k-step 1
                        hb[64 * f + lane] = __builtin_bit_cast(u32x4, acc_to_b_pk(f < 2 ? s : dp, f & 1));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            KP_STAMP(3);
            KS_MARK(3);
            // everything this wave staged before this step has landed (the tile read in step i + 1 among it)
            if constexpr (ISSUED) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            KP_STAMP(4);
            KS_MARK(4);
            lds_barrier();
            KP_STAMP(5);
            KS_MARK(5);
            KP_FLUSH(i, wave, lane);
        };
        // block 0's chains (no element-wise stage to overlap with yet)
        auto chains0 = [&](f32x16& s, f32x16& dp, float (&nd)[16]) {
            init_rows(b2_slot(0), s, dp, nd);
            const bf16_t* qp = &lds.q[b2_slot(0)][roff];
            const bf16_t* dop = &lds.d[b2_slot(0)][roff];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) s = mfma32(*(const bf16x8*)(qp + 16 * ks), kf[ks], s);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) dp = mfma32(*(const bf16x8*)(dop + 16 * ks), vf[ks], dp);
        };
        f32x16 sA, dpA, sB, dpB;
        float ndA[16], ndB[16];
        chains0(sA, dpA, ndA);  // block 0's chains while tiles 1 and 2 land
        KSP_MARK(5);
        if (nq > 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // tile 1 (step 0 reads it); tile 2 is waited for in step 0
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        KSP_MARK(6);
        KS_MARK(6);
        {
            int i = 0;
            for (; i + 4 < nq; i += 2) {  // steady state: both steps stage a tile
                step(std::true_type{}, i, sA, dpA, ndA, sB, dpB, ndB);
                step(std::true_type{}, i + 1, sB, dpB, ndB, sA, dpA, ndA);
            }
            for (; i < nq; i += 2) {
                if (i + 3 < nq) step(std::true_type{}, i, sA, dpA, ndA, sB, dpB, ndB);
                else step(std::false_type{}, i, sA, dpA, ndA, sB, dpB, ndB);
                if (i + 1 < nq) step(std::false_type{}, i + 1, sB, dpB, ndB, sA, dpA, ndA);
            }
        }
        lds_barrier();  // the consumers' last step
        KS_MARK(7);
        KS_END(gridDim.x / 2 + 88, wave, lane);
#ifdef V1T_KCLK
        if (blockIdx.x == 3000 && wave == 0 && lane == 0) {
            g_kclk[0] = __builtin_amdgcn_s_memtime() - kclk_c0;
            g_kclk[1] = __builtin_amdgcn_s_memrealtime() - kclk_r0;
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumer
    load_kv();
    stage(0);
    if (nq > 1) stage(1);
    KSP_MARK(0);
    f32x16 dk[G::DB], dv[G::DB];
#pragma unroll
    for (int d = 0; d < G::DB; ++d) {
        zero16(dk[d]);
        zero16(dv[d]);
    }
    const int toff = tr_lane_off_perm(lane, G::RSTR);
    const int nkb = a.ldds / 32;  // 32-key blocks per query block
    bf16_t* ds_wave = a.ds + ((bh * nq) * nkb + (size_t)(rb * 4 + pw)) * 1024 + lane * 8;
    const unsigned kmask = kok ? 0xFFFFFFFFu : 0u;  // keys beyond T: zeros (the dQ GEMM multiplies them with clamped K rows)
    const bool ktail = rb * 128 + 32 * pw + 32 > a.T;
    store_kv();
    KSP_MARK(1);
    lds_barrier();  // K / V images written
    KSP_MARK(2);
    lds_barrier();  // the producers hold their K / V fragments: slots 0-2 are free
    if (nq > 2) stage(2);
    KSP_MARK(3);
    if (nq > 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    KSP_MARK(6);
    KS_MARK(6);
    // step 0: nothing to consume yet
    if (nq > 3) stage(3);
    KSP_MARK(4);
    if (nq > 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    KSP_MARK(5);
    KP_DECL;
    constexpr int NSLOT = 4 * G::DB, LA = 3;  // one MFMA per slot: dV over (d block, k-step), then dK
    // step i >= 1: consume block j = i - 1 (dS' to HBM, dV^T += dO^T P, dK^T += Q^T dS'; operand k order = acc_to_b_pk order) and,
    // if ISSUED, stage tile i + 3. The hand-off reads go first so that their latency hides behind the DMA issue.
    auto cstep = [&](auto issued_tag, int i) {
        constexpr bool ISSUED = decltype(issued_tag)::value;
        KP_STAMP(0);
        KS_MARK(0);
        const int j = i - 1, slot = b2_slot(j);
        const u32x4* hb = lds.hand[j & 1][pw][0];
        const u32x4 p0 = hb[lane], p1 = hb[64 + lane];
        const u32x4 s0 = hb[128 + lane], s1 = hb[192 + lane];
#ifndef V1T_B2_NODMA
        if constexpr (ISSUED) stage(i + 3);
#endif
        KP_STAMP(1);
        KS_MARK(1);
        unsigned ta = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&lds.d[slot][toff];
        unsigned qa = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&lds.q[slot][toff];
        asm volatile("" : "+v"(ta), "+v"(qa));
        auto frag = [&](int m) {
#ifdef V1T_B2_NOFRAG
            return __builtin_bit_cast(bf16x8, (m & 1) ? p1 : s0);
#endif
            const bf16_t* p = (const bf16_t*)(const __attribute__((address_space(3))) bf16_t*)(uintptr_t)(m < 2 * G::DB ? ta : qa);
            return tr_frag_perm<G::RSTR>(p, 16 * (m & 1), 32 * ((m % (2 * G::DB)) >> 1));
        };
        bf16x8 fr[NSLOT];
#pragma unroll
        for (int m = 0; m < LA; ++m) fr[m] = frag(m);
        bf16_t* dst = ds_wave + (size_t)j * nkb * 1024;
        u32x4 m0 = s0, m1 = s1;
        if (ktail) {  // wave-uniform: only the last key block of an (image, head) has keys beyond T
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m0[e] &= kmask;
                m1[e] &= kmask;
            }
        }
#ifndef V1T_B2_NOSTORE
        // non-temporal: 2.5 GB per launch, written once and read by the NEXT kernel - keeping the lines in L2 only evicts the K / V / Q / dO
        // rows the workgroups of this XCD re-read (round 4 A/B, tools/ab_many.py: backward 2557 -> 2514 us per 112-image launch, same checksum)
        __builtin_nontemporal_store(m0, (u32x4*)dst);
        __builtin_nontemporal_store(m1, (u32x4*)(dst + 512));
#else
        asm volatile("" ::"v"(m0), "v"(m1), "v"(dst));
#endif
        KP_STAMP(2);
        KS_MARK(2);
        const bf16x8 P0 = __builtin_bit_cast(bf16x8, p0), P1 = __builtin_bit_cast(bf16x8, p1);
        const bf16x8 S0 = __builtin_bit_cast(bf16x8, s0), S1 = __builtin_bit_cast(bf16x8, s1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NSLOT; ++m) {
            if (m + LA < NSLOT) fr[m + LA] = frag(m + LA);
            const int d = (m % (2 * G::DB)) >> 1;
            if (m < 2 * G::DB) dv[d] = mfma32(fr[m], (m & 1) ? P1 : P0, dv[d]);
            else dk[d] = mfma32(fr[m], (m & 1) ? S1 : S0, dk[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // everything issued before this step has landed (the tile the producers read in step i + 1 among it); this step's
        // own three DMA operations and its two dS' stores may stay in flight
        KP_STAMP(3);
        KS_MARK(3);
        if constexpr (ISSUED) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KP_STAMP(4);
        KS_MARK(4);
        lds_barrier();
        KP_STAMP(5);
        KS_MARK(5);
        KP_FLUSH(i, wave, lane);
    };
    {
        int i = 1;
        for (; i + 3 < nq; ++i) cstep(std::true_type{}, i);
        for (; i <= nq; ++i) cstep(std::false_type{}, i);
    }
    KS_MARK(0);
    // ---- epilogue: dK / dV rows leave as whole 320-B rows. The accumulators hold, per key (lane), 4 consecutive head dims per
    // register group: stored directly that is 8 B per lane over 32 rows per instruction, 160 instructions a wave whose ~20 k
    // partial-line writes drain for ~15 k cycles into the NEXT workgroup's prologue (its first loads queue behind them). After
    // the last barrier nobody reads the tile ring any more: the wave transposes through its own two slots and writes 16-B
    // chunks, 3.2 rows per instruction.
    {
        const float sc = a.scale[a.scale_per_head ? h : 0];
        const float kfac = DROP ? a.adrop.inv_keep : 1.0f;
        const float fk = sc * kfac;
        bf16_t* const tk = lds.q[pw];
        bf16_t* const tv = lds.d[pw];
        const int wo = (lane & 31) * G::RSTR + 4 * h2;
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int rq4 = 0; rq4 < 4; ++rq4) {
                bf16x4 wk, wv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wk[j] = (bf16_t)(dk[d][4 * rq4 + j] * fk);
                    wv[j] = (bf16_t)(dv[d][4 * rq4 + j] * kfac);
                }
                *(bf16x4*)(tk + wo + 32 * d + 8 * rq4) = wk;
                *(bf16x4*)(tv + wo + 32 * d + 8 * rq4) = wv;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no barrier
        constexpr int CPRD = DP / 8;  // 16-B chunks per row
        const int key0 = rb * 128 + 32 * pw;
        bf16_t* const obase = a.dqkv + ((size_t)b * a.T + key0) * a.lddqkv + h * DP;
#pragma unroll
        for (int it = 0; it < (32 * CPRD + 63) / 64; ++it) {
            const int pch = 64 * it + lane;
            const int row = pch / CPRD, ch = pch - row * CPRD;
            if (pch < 32 * CPRD && key0 + row < a.T) {
                const u32x4 xk = *(const u32x4*)(tk + row * G::RSTR + 8 * ch);
                const u32x4 xv = *(const u32x4*)(tv + row * G::RSTR + 8 * ch);
                bf16_t* o = obase + (size_t)row * a.lddqkv + 8 * ch;
                *(u32x4*)(o + HD) = xk;
                *(u32x4*)(o + 2 * HD) = xv;
            }
        }
    }
    KSP_MARK(7);
    KS_MARK(7);
    KS_END(gridDim.x / 2 + 88, wave, lane);
}

// dQ = dS' . K over the materialised dS' (layout: attention.h). Workgroup = 8 waves = 8 query blocks (256 queries) of one
// (image, head); a wave owns one 32-query block: its dS' blocks (2 KB each, key on the lane as the dK/dV kernel stored them)
// arrive by LDS-DMA into a wave-private region and are read TRANSPOSED as the A operand (32 queries x 16 keys, natural key
// order), K tiles are shared by the workgroup and read transposed as the B operand. Bound by reading dS' once from HBM
// (2 B per (query, key)) through the LDS-DMA path: 32-key stages in a 3-deep ring behind a counted vmcnt. Accumulator rows are queries in the permuted
// order the stored blocks imply (see the epilogue).
// DEEP (round 5): separate ring depths for the two operands - FOUR dS' stages (three in flight, 96 KB) and three K stages - instead of one
// 3-deep ring of (dS' + K) stages (two in flight: 64 KB of dS' + 21 KB of K). The kernel streams 2.48 GB of dS' per 112-image launch and a CU
// fetches what it keeps in flight per memory latency (~11.7 B per clock with 85 KB outstanding): the K tile is a quarter of a stage and comes out
// of L2, so its depth buys nothing, while a fourth dS' stage fits the 160 KB exactly (4 x 32 KB + 3 x 10 KB = 161 792 B). Issue order per
// step: K(st + 2), then dS'(st + 3); the counted vmcnt in front of the barrier lets dS'(st + 1), K(st + 1) and dS'(st + 2) fly.
template <int DP, bool DEEP = false>
__global__ __launch_bounds__(512, 1) void attn_bwd_dq2_kernel(AttnArgs a) {
    using G = Geo<DP>;
    // QPW query blocks per wave: a workgroup covers 8 x QPW x 32 = 512 queries, so a head's K tiles are streamed by 4 workgroups
    // instead of 7 (at 256 queries K re-reads from L2 were 40 % of the bytes the LDS-DMA path moved, and that path - ~24 GB/s per
    // CU - is what bounds this kernel)
    constexpr int KT = 32, NBUF = 3, QPW = 2;  // 3 x 43 KB of LDS: two stages in flight while one is consumed
    constexpr int NBS = DEEP ? 4 : NBUF, NBK = NBUF;  // ring depths of the dS' blocks / the K tiles
    using DmaK = TileDma<DP, G::TSTR, KT, 8>;  // read transposed only: the 64 B x odd row stride (RSTR's 336 B: 2-way conflicts on a quarter of the banks)
    __shared__ __attribute__((aligned(16))) bf16_t sS[NBS][8][QPW][1024];
    __shared__ __attribute__((aligned(16))) bf16_t sK[NBK][DmaK::LDS_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rb, h, b;
    decode_block(a, blockIdx.x, gridDim.x, rb, h, b, 256 * QPW);
    const int nq = (a.T + 31) / 32, nkb = a.ldds / 32;
    const size_t bh = (size_t)b * a.H + h;
    DmaK dmaK;
    dmaK.init(lane, wave, a.ldqkv);
    const bf16_t* kbase = a.qkv + (size_t)b * a.T * a.ldqkv + a.H * DP + h * DP;
    // this wave's query blocks: rb * 8 QPW + wave + 8 u (interleaved, so that the ragged last workgroup keeps every wave busy)
    int qb[QPW];
    const char* sbase[QPW];
    int nact = 0;
#pragma unroll
    for (int u = 0; u < QPW; ++u) {
        qb[u] = rb * 8 * QPW + wave + 8 * u;
        if (qb[u] < nq) nact = u + 1;  // wave-uniform; blocks are active in order of u
        sbase[u] = (const char*)(a.ds + ((bh * nq + min(qb[u], nq - 1)) * nkb) * 1024);
    }
    // a block is 2 KB contiguous ([k-step 2][half 2][key 32][8]): the instruction offset advances source and LDS alike. The LDS image
    // keeps the upper half of each k-step ROTATED by 4 keys (slot j <- key (j + 4) & 31, by the lane's source offset): a transposed
    // read takes 4 keys x 16 B from both halves, which sit 512 B = a whole number of bank rounds apart - 2-way conflicts on every
    // A read; after the rotation the two 64-B groups are 16 banks apart
    const unsigned svoff = (unsigned)((lane < 32 ? lane : 32 + ((lane + 4) & 31)) * 16);
    f32x16 dq[QPW][G::DB];
#pragma unroll
    for (int u = 0; u < QPW; ++u)
#pragma unroll
        for (int d = 0; d < G::DB; ++d) zero16(dq[u][d]);
    // A operand (32 queries x 16 keys, natural key order) read transposed from a stored block: chunk s = [h 2][key 32][8] holds
    // for key k, half h the queries 16 s + 8 (j >> 2) + 4 h + (j & 3), j = 0..7. The 16-lane group gi = lane >> 4 takes chunk
    // s = gi & 1 (A rows m = 16 s + i) and keys 8 (gi >> 1) + {0..3} (second read: + 4); lane 4 q' + p of the group supplies
    // the address of key row q', columns 4 p .. 4 p + 3 = elements 4 (p & 1) .. of half p >> 1; lane i receives column i.
    const int gi = lane >> 4, li = lane & 15;
    const int ahalf = (li & 3) >> 1, akey = 8 * (gi >> 1) + (li >> 2);
    int aoffs[4];  // element offsets for key offsets 0, 4, 16, 20 (k-step kk, first / second read)
#pragma unroll
    for (int c = 0; c < 4; ++c)
        aoffs[c] = (gi & 1) * 512 + ahalf * 256 + ((akey + 16 * (c >> 1) + 4 * (c & 1) - 4 * ahalf) & 31) * 8 + (li & 1) * 4;
    const int nst = nkb;  // 32-key stages
    const bool ds_nt = a.ds_nt != 0;  // kernel-uniform
    // vector-memory operations this wave issues per stage: its own dS' blocks (2 pieces each) + its share of the K tile
    const int nk_ops = min(DmaK::PW, max(0, DmaK::NINST - wave * DmaK::PW)), ns_ops = 2 * nact;
    const int nops = ns_ops + nk_ops;
    auto stage_s = [&](int st) {
        const int buf = st % NBS;
#pragma unroll
        for (int u = 0; u < QPW; ++u)
            if (u < nact) {
                const unsigned l0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&sS[buf][wave][u][0]);
                // dS' is read exactly once: the non-temporal policy (streams past the L2's / Infinity Cache's LRU; tools/microbench/hbm_stream: LDS-DMA reads
                // 7.1 against 6.4 TB/s) - round 6; V1T_DQ2_NT=0 (dev, A/B): the plain policy of rounds 2-5
                if (ds_nt) {
                    TileDma<DP, G::RSTR>::template group<2, true>(sbase[u] + (size_t)st * 2048, l0, svoff, svoff, 0, 0);
                } else {
                    TileDma<DP, G::RSTR>::template group<2>(sbase[u] + (size_t)st * 2048, l0, svoff, svoff, 0, 0);
                }
            }
    };
    auto stage_k = [&](int st) { dmaK.issue(kbase, KT * st, a.T, sK[st % NBK]); };
    auto stage = [&](int st) {
        stage_s(st);
        stage_k(st);
    };
    if constexpr (DEEP) {  // issue sequence S(0) K(0) S(1) K(1) S(2) | K(2) S(3) | K(3) S(4) | ...
        stage(0);
        if (nst > 1) stage(1);
        if (nst > 2) stage_s(2);
    } else {
        stage(0);
        if (nst > 1) stage(1);
    }
    for (int st = 0; st < nst; ++st) {
        const int buf = st % NBS, kbuf = st % NBK;
        if constexpr (DEEP) {
            // everything up to K(st) has landed for this wave; behind it in the queue: dS'(st + 1), K(st + 1), dS'(st + 2) may fly
            const int fly = (st + 1 < nst ? nops : 0) + (st + 2 < nst ? ns_ops : 0);
            wait_vmcnt_dyn(fly);
            __builtin_amdgcn_s_barrier();  // also: every wave is past stage st - 1, whose dS' / K slots the next DMAs overwrite
            if (st + 2 < nst) stage_k(st + 2);
            if (st + 3 < nst) stage_s(st + 3);
        } else {
            // stage st has landed for this wave (the younger stage may fly), then for the workgroup
            const int younger = min(NBUF - 2, nst - 1 - st);
            wait_vmcnt_dyn(younger * nops);
            __builtin_amdgcn_s_barrier();  // also: every wave is past stage st - 1, whose buffer the next DMA overwrites
            if (st + NBUF - 1 < nst) stage(st + NBUF - 1);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 kfr[G::DB];
#pragma unroll
            for (int d = 0; d < G::DB; ++d) kfr[d] = lds_tr_frag_nat(sK[kbuf], G::TSTR, 16 * kk, 32 * d, lane);
#pragma unroll
            for (int u = 0; u < QPW; ++u)
                if (u < nact) {
                    const bf16_t* ab = &sS[buf][wave][u][0];
                    const bf16x4 lo = lds_tr_read(ab + aoffs[2 * kk]), hi = lds_tr_read(ab + aoffs[2 * kk + 1]);
                    bf16x8 afr;
                    afr[0] = lo[0]; afr[1] = lo[1]; afr[2] = lo[2]; afr[3] = lo[3];
                    afr[4] = hi[0]; afr[5] = hi[1]; afr[6] = hi[2]; afr[7] = hi[3];
#pragma unroll
                    for (int d = 0; d < G::DB; ++d) dq[u][d] = mfma32(afr, kfr[d], dq[u][d]);
                }
        }
    }
    const float f = a.scale[a.scale_per_head ? h : 0] * (a.adrop.thresh16 ? a.adrop.inv_keep : 1.0f);
    const int dcol = lane & 31;
#pragma unroll
    for (int u = 0; u < QPW; ++u)
        if (u < nact) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = acc_row(r, lane);          // A-operand row
                const int i = m & 15;                    // position inside the stored 16-query row of chunk m >> 4
                const int q = 32 * qb[u] + 16 * (m >> 4) + 8 * ((i & 7) >> 2) + 4 * (i >> 3) + (i & 3);
                if (q < a.T) {
                    bf16_t* orow = a.dqkv + ((size_t)b * a.T + q) * a.lddqkv + h * DP + dcol;
#pragma unroll
                    for (int d = 0; d < G::DB; ++d) orow[32 * d] = (bf16_t)(dq[u][d][r] * f);
                }
            }
        }
}

template <int DP, bool DROP, bool DIAG>
int launch_bwd_t(const AttnArgs& a_in, hipStream_t s) {
    const AttnArgs& a = a_in;
    const int n = ((a.T + 127) / 128) * a.H * a.B;
    // the producer / consumer pair is laid out for head dim 160 (10.5 DMA pieces per 32-row tile dealt 3 / 3 / 3 / 2 over the wave pairs, the
    // row constants riding in pair 3's free piece): at head dim 128 pair 3 would issue pieces that do not exist - the first test at that
    // size (round 4, tools/attn_probe.py) returned garbage gradients - so every other head dim takes the recompute kernels below
    if constexpr (DP == 160 && !DIAG) {
        if (a.ds) {
            if (a.ldds != attn_ds_ld(a.T)) return V1T_ERR_ARG;
            prof_begin(PROF_ATTN_DKV, s);
            hipLaunchKernelGGL((attn_bwd_dkv2_kernel<DP, DROP>), dim3(n), dim3(512), 0, s, a);
            prof_end(PROF_ATTN_DKV, s);
            prof_begin(PROF_ATTN_DQ, s);
            static const bool dq_deep = !(dev_env("V1T_DQ2_DEEP") && !atoi(dev_env("V1T_DQ2_DEEP")));  // dev (A/B): 0 = the one 3-deep ring
            static const bool dq_nt = !(dev_env("V1T_DQ2_NT") && !atoi(dev_env("V1T_DQ2_NT")));
            AttnArgs a = a_in;
            a.ds_nt = dq_nt ? 1 : 0;
            if (dq_deep) hipLaunchKernelGGL((attn_bwd_dq2_kernel<DP, true>), dim3(((a.T + 511) / 512) * a.H * a.B), dim3(512), 0, s, a);
            else hipLaunchKernelGGL((attn_bwd_dq2_kernel<DP, false>), dim3(((a.T + 511) / 512) * a.H * a.B), dim3(512), 0, s, a);
            prof_end(PROF_ATTN_DQ, s);
            return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
        }
    }
    static const bool split = dev_env("V1T_ATTN_BWD_SPLIT") != nullptr;  // dev switch: time the two bodies separately
    if (DP >= 128 && !split) {
        const int n8 = (n + 7) / 8 * 8;
        prof_begin(PROF_ATTN_DKV, s);
        hipLaunchKernelGGL((attn_bwd_fused_kernel<DP, DROP, DIAG>), dim3(n8 + n), dim3(256), 0, s, a, n, n8);
        prof_end(PROF_ATTN_DKV, s);
        return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    }
    dim3 grid(n);
    prof_begin(PROF_ATTN_DQ, s);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DP, DROP, DIAG>), grid, dim3(256), 0, s, a);
    prof_end(PROF_ATTN_DQ, s);
    prof_begin(PROF_ATTN_DKV, s);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DP, DROP, DIAG>), grid, dim3(256), 0, s, a);
    prof_end(PROF_ATTN_DKV, s);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
template <int DP>
int launch_delta_t(const AttnArgs& a, float* delta, hipStream_t s) {
    if (DP == 160 && !a.mask_diag && a.ds) {  // producer / consumer backward: padded row constants behind dS'
        const int TPQ = attn_ds_tpq(a.T);
        float* nlse = (float*)(a.ds + attn_ds_elems(a.B, a.H, a.T));
        const long long total = (long long)a.B * TPQ * a.H;
        hipLaunchKernelGGL((attn_delta2_kernel<DP>), dim3((unsigned)((total * 16 + 255) / 256)), dim3(256), 0, s, a, nlse, nlse + attn_rc_floats(a.B, a.H, a.T), TPQ);
        return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    }
    const long long total = (long long)a.B * a.T * a.H;
    hipLaunchKernelGGL((attn_delta_kernel<DP>), dim3((unsigned)((total * 16 + 255) / 256)), dim3(256), 0, s, a, delta);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

template <int DP>
int fwd_flags(const AttnArgs& a, hipStream_t s) {
    const bool drop = a.adrop.thresh16 != 0, diag = a.mask_diag != 0;
    if (drop) return diag ? launch_fwd_t<DP, true, true>(a, s) : launch_fwd_t<DP, true, false>(a, s);
    return diag ? launch_fwd_t<DP, false, true>(a, s) : launch_fwd_t<DP, false, false>(a, s);
}
template <int DP>
int bwd_flags(const AttnArgs& a, hipStream_t s) {
    const bool drop = a.adrop.thresh16 != 0, diag = a.mask_diag != 0;
    if (drop) return diag ? launch_bwd_t<DP, true, true>(a, s) : launch_bwd_t<DP, true, false>(a, s);
    return diag ? launch_bwd_t<DP, false, true>(a, s) : launch_bwd_t<DP, false, false>(a, s);
}


// ------------------------------------------------------------------------------------------
// Attention rollout support (reference utils/attention_rollout.py:92-122): A[b][i][j] = max over heads of
// the softmax probabilities P_h[i][j] = exp2(S_h * c_h - lse2_h[i]) (recomputed from the saved q, k and
// log2-sum-exp, never materialising the (B,H,T,T) tensor the reference's forward hooks stack), plus the row
// sums rs[i] = sum_j A[i][j] + 1 of the identity-augmented matrix. One wave = 32 query rows, all heads' Q
// fragments in registers (one wave per SIMD), K tiles of all heads staged by LDS-DMA.
// PH (per head): instead of the head-max, every head's probabilities P[b][h][q][:] (B, H, T, TP) - the tensor the reference's
// Recorder stacks (attention_rollout.py:31-36, 76); rowsum is not written.
template <int DP, int HH, bool PH = false>
__global__ __launch_bounds__(256, 1) void rollout_headmax_kernel(AttnArgs a, float* A, int TP, float* rowsum) {
    using G = Geo<DP>;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][HH][TileDma<DP, G::RSTR>::LDS_ELEMS];
    // head-max tile of a wave (32 queries x 32 keys fp32) on its way out: a lane holds 16 keys of ONE query (4 x 16 B at a row stride of TP
    // floats), so stored from the registers an instruction scatters 64 16-B pieces over 32 rows - partial lines for the 2.8 GB this kernel
    // writes per 256-image launch. Staged through a wave-private block instead (row stride 36 floats: conflict-free 16-B writes and reads)
    // and written as 8 lanes x 16 B = one 128-B row segment per query row (round 5)
    constexpr int HS = 36;
    __shared__ __attribute__((aligned(16))) float sOut[PH ? 1 : 4][PH ? 4 : 32 * HS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int q = blockIdx.x * 128 + 32 * wave + (lane & 31);
    const int h2 = lane >> 5;
    const int HD = HH * DP;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    TileDma<DP, G::RSTR> dma;
    dma.init(lane, wave, a.ldqkv);
    const bool qok = q < a.T;
    bf16x8 qf[HH][G::KS];
    float cs[HH], nl[HH];
#pragma unroll
    for (int h = 0; h < HH; ++h) {
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            u32x4 t = qok ? *(const u32x4*)(qkv_b + (size_t)q * a.ldqkv + h * DP + 16 * ks + 8 * h2) : u32x4{0, 0, 0, 0};
            qf[h][ks] = *(bf16x8*)&t;
        }
        cs[h] = a.scale[a.scale_per_head ? h : 0] * LOG2E;
        nl[h] = -a.lse2[((size_t)b * HH + h) * a.T + (qok ? q : 0)];
    }
    const int koff = (lane & 31) * G::RSTR + 8 * h2;
    const int nt = (a.T + 31) / 32;
    float rs = 0.f;
    auto stage = [&](int t, int buf) {
#pragma unroll
        for (int h = 0; h < HH; ++h) dma.issue(qkv_b + HD + h * DP, 32 * t, a.T, sK[buf][h]);
    };
    stage(0, 0);
#pragma unroll
    for (int h = 0; h < HH; ++h) {
        touch(qf[h]);
        touch(cs[h]);
        touch(nl[h]);
    }
    dma_wait_and_barrier();
    for (int kt = 0; kt < nt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nt) stage(kt + 1, buf ^ 1);
        f32x16 amax;
#pragma unroll
        for (int h = 0; h < HH; ++h) {
            f32x16 s;
            zero16(s);
            const bf16_t* kp = &sK[buf][h][koff];
            bf16x8 kfr[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) kfr[ks] = *(const bf16x8*)(kp + 16 * ks);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) s = mfma32(kfr[ks], qf[h][ks], s);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(fmaf(s[r], cs[h], nl[h]));
                amax[r] = (h == 0 || PH) ? p : fmaxf(amax[r], p);
            }
            if constexpr (PH) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int key0 = 32 * kt + 8 * g + 4 * h2;
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (key0 + j >= a.T || (a.mask_diag && key0 + j == q)) ? 0.f : amax[4 * g + j];
                    if (qok && key0 < TP) *(f32x4*)(A + (((size_t)b * HH + h) * a.T + q) * TP + key0) = o;
                }
            }
        }
        if constexpr (PH) {
            dma_wait_and_barrier();
            continue;
        }
        if constexpr (!PH) {
            float* so = sOut[wave];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = 32 * kt + 8 * g + 4 * h2;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = key0 + j;
                    const bool dead = key >= a.T || (a.mask_diag && key == q);
                    o[j] = dead ? 0.f : amax[4 * g + j];
                    rs += o[j];
                }
                *(f32x4*)(so + (lane & 31) * HS + 8 * g + 4 * h2) = o;
            }
            // (LDS operations of one wave execute in order: no barrier between the wave's own writes and reads)
            const int qw0 = blockIdx.x * 128 + 32 * wave;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + (lane >> 3), kc = 32 * kt + 4 * (lane & 7);
                const f32x4 o = *(const f32x4*)(so + row * HS + 4 * (lane & 7));
                if (qw0 + row < a.T && kc < TP) *(f32x4*)(A + ((size_t)b * a.T + qw0 + row) * TP + kc) = o;
            }
        }
        dma_wait_and_barrier();
    }
    rs += __shfl_xor(rs, 32);
    if (!PH && qok && h2 == 0) rowsum[(size_t)b * a.T + q] = rs + 1.0f;
}

// The head-max map with TWO waves per SIMD (round 5). The 4-wave kernel above keeps the Q fragments of all HH heads in registers (330 VGPRs
// at head dim 160 and 4 heads): one wave per SIMD, whose 40 MFMAs, 64 exponentials, LDS-DMA issue and stores per 32-key tile run one after
// the other (~5800 cycles per tile, 1.96 ms per 256-image launch, with the matrix pipe ~22 % busy). Here a workgroup still owns 128 query
// rows but has 8 waves: waves 0-3 ("group 0") compute heads [0, HH/2), waves 4-7 heads [HH/2, HH) of the SAME 32 queries each - half the Q
// fragments per wave, so two waves fit a SIMD and one group's exponentials overlap the other's MFMAs. Group 1 hands its partial maximum
// (16 floats per lane) to its group-0 partner through a double-buffered LDS block; group 0 finishes tile kt - 1 (maximum, masking, row sum,
// staged 128-B row stores) at the start of iteration kt, behind the barrier that published it. K tiles of all heads are staged by all 8
// waves. Same results bit for bit (max is exact and order-free; the row sums add the same values in the same order).
template <int DP, int HH>
__global__ __launch_bounds__(512, 1) void rollout_headmax8_kernel(AttnArgs a, float* A, int TP, float* rowsum) {
    static_assert(HH % 2 == 0, "two head groups");
    using G = Geo<DP>;
    // K tiles are staged by group 1's four waves ALONE: group 0 writes the finished tiles to HBM, and on this target loads and stores
    // pending together retire vmcnt out of order, so a wave that does both can only wait with vmcnt(0) - i.e. for its STORES' round trip to
    // HBM - at every tile end (44 % of the wave cycles were waits, SQ counters profiles/r05_pmc_headmax.txt). Split this way group 1's
    // vmcnt(0) covers LDS-DMA only and group 0 never waits for memory: a store's data has left the registers when it is issued.
    using Dma = TileDma<DP, G::RSTR, 32, 4>;
    constexpr int HG = HH / 2, HS = 36;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][HH][Dma::LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) float sX[2][4][64 * 16];  // group 1 -> group 0: [tile parity][query wave][lane][16]
    __shared__ __attribute__((aligned(16))) float sOut[4][32 * HS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int b = blockIdx.y;
    const int q = blockIdx.x * 128 + 32 * w4 + (lane & 31);
    const int h2 = lane >> 5;
    const int HD = HH * DP;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    Dma dma;
    dma.init(lane, w4, a.ldqkv);
    const bool qok = q < a.T;
    bf16x8 qf[HG][G::KS];
    float cs[HG], nl[HG];
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
        const int h = grp * HG + hh;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            u32x4 t = qok ? *(const u32x4*)(qkv_b + (size_t)q * a.ldqkv + h * DP + 16 * ks + 8 * h2) : u32x4{0, 0, 0, 0};
            qf[hh][ks] = *(bf16x8*)&t;
        }
        cs[hh] = a.scale[a.scale_per_head ? h : 0] * LOG2E;
        nl[hh] = -a.lse2[((size_t)b * HH + h) * a.T + (qok ? q : 0)];
    }
    const int koff = (lane & 31) * G::RSTR + 8 * h2;
    const int nt = (a.T + 31) / 32;
    float rs = 0.f;
    auto stage = [&](int t, int buf) {
        if (grp == 1) {  // wave-uniform
#pragma unroll
            for (int h = 0; h < HH; ++h) dma.issue(qkv_b + HD + h * DP, 32 * t, a.T, sK[buf][h]);
        }
    };
    auto tile_end = [&]() {
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this group's LDS-DMA of the next tile has landed
        __syncthreads();
    };
    // group 0: combine its own maximum of tile kt with group 1's, mask, add to the row sums, write the tile out as 128-B row segments
    f32x16 mine;
    zero16(mine);
    auto finish = [&](int kt) {
        const float* px = &sX[kt & 1][w4][lane * 16];
        float* so = sOut[w4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 other = *(const f32x4*)(px + 4 * g);
            const int key0 = 32 * kt + 8 * g + 4 * h2;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = key0 + j;
                const bool dead = key >= a.T || (a.mask_diag && key == q);
                o[j] = dead ? 0.f : fmaxf(mine[4 * g + j], other[j]);
                rs += o[j];
            }
            *(f32x4*)(so + (lane & 31) * HS + 8 * g + 4 * h2) = o;
        }
        const int qw0 = blockIdx.x * 128 + 32 * w4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + (lane >> 3), kc = 32 * kt + 4 * (lane & 7);
            const f32x4 o = *(const f32x4*)(so + row * HS + 4 * (lane & 7));
#ifdef V1T_DEV_ABLATION_HM_NOSTORE  // timing-only ablation (garbage results): build.py refuses V1T_DEV_ABLATION* for the product library
            if (qw0 + row < a.T && kc < TP && o[0] == 12345.f) *(f32x4*)(A + ((size_t)b * a.T + qw0 + row) * TP + kc) = o;
#else
            if (qw0 + row < a.T && kc < TP) *(f32x4*)(A + ((size_t)b * a.T + qw0 + row) * TP + kc) = o;
#endif
        }
    };
    stage(0, 0);
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
        touch(qf[hh]);
        touch(cs[hh]);
        touch(nl[hh]);
    }
    tile_end();
    for (int kt = 0; kt < nt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nt) stage(kt + 1, buf ^ 1);
        if (grp == 0 && kt > 0) finish(kt - 1);  // published by the barrier that ended iteration kt - 1 (wave-uniform branch)
        f32x16 amax;
#pragma unroll
        for (int hh = 0; hh < HG; ++hh) {
            const int h = grp * HG + hh;
            f32x16 sacc;
            zero16(sacc);
            const bf16_t* kp = &sK[buf][h][koff];
            bf16x8 kfr[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) kfr[ks] = *(const bf16x8*)(kp + 16 * ks);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) sacc = mfma32(kfr[ks], qf[hh][ks], sacc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(fmaf(sacc[r], cs[hh], nl[hh]));
                amax[r] = hh == 0 ? p : fmaxf(amax[r], p);
            }
        }
        if (grp == 1) {
            float* px = &sX[kt & 1][w4][lane * 16];
#pragma unroll
            for (int g = 0; g < 4; ++g) *(f32x4*)(px + 4 * g) = f32x4{amax[4 * g], amax[4 * g + 1], amax[4 * g + 2], amax[4 * g + 3]};
        } else {
            mine = amax;
        }
        tile_end();
    }
    if (grp == 0) {
        finish(nt - 1);
        rs += __shfl_xor(rs, 32);
        if (qok && h2 == 0) rowsum[(size_t)b * a.T + q] = rs + 1.0f;
    }
}

// u[b][j] = sum_i w_i * (A[b][i][j] + delta_ij),  w_i = v[b][i] / rs[b][i]   (v == nullptr: v = e_0)
__global__ __launch_bounds__(256) void rollout_vecmat_kernel(const float* A, const float* rowsum, const float* v, float* u, int T, int TP) {
    extern __shared__ __attribute__((aligned(16))) float sw[];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < T; i += 256) {
        // v == nullptr (e_0): only row 0 and its row sum are read - the head-max kernel may have been run for the first rows alone
        sw[i] = v ? v[(size_t)b * T + i] / rowsum[(size_t)b * T + i] : (i == 0 ? 1.f / rowsum[(size_t)b * T] : 0.f);
    }
    __syncthreads();
    const int j0 = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (j0 >= TP) return;
    const float* Ab = A + (size_t)b * T * TP + j0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int iend = v ? T : 1;  // e_0 picks row 0 only
    for (int i = 0; i < iend; ++i) {
        const f32x4 x = *(const f32x4*)(Ab + (size_t)i * TP);
        const float w = sw[i];
        acc[0] += w * x[0]; acc[1] += w * x[1]; acc[2] += w * x[2]; acc[3] += w * x[3];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j0 + j < T) u[(size_t)b * T + j0 + j] = acc[j] + sw[j0 + j];
}

// One step of the FULL rollout chain, result <- A_hat . result with A_hat = (A + I) / rowsum (attention_rollout.py:107-117),
// carried as X = result^T so that both operands are K-contiguous and the output is the next step's left operand without a
// transpose:  Xout[n][i] = sum_j Xin[n][j] * (A[i][j] + [i == j]) / rowsum[i];  Xin == nullptr is the identity (first block).
// fp32 in, fp32 out, on the MFMAs in split-bf16 (x = hi + lo, hi.hi + lo.hi + hi.lo: ~2^-17 relative per product - the reference
// multiplies in fp32): operands are converted while they are staged (no bf16 planes in HBM); tiling below. 2 T^3 flops per image and
// step (9 GFLOP at T = 1654), three MFMA products each; only row 0 of the final product is used downstream, which is why
// v1t_rollout_vecmat (2 T^2) is the default - this entry exists because the reference's algorithm is the matrix chain.
constexpr int RM_BK = 32, RM_LS = RM_BK + 8;
// Workgroup = 8 waves = a 256 x 256 tile of one image's X . A_hat^T, waves 4 (rows) x 2 (columns), each 64 x 128 = 2 x 4 MFMA blocks:
// an LDS fragment feeds 4 (X) or 2 (A_hat) blocks x 3 split products, 12 fragment reads per 24 MFMAs. (The first version, 128 x 128
// tiles of 4 waves, fetched 287 MB per image and step through L2 for 22 MB of operands and ran at the ~6.4 TB/s that moves; this tile
// halves the traffic.) K tiles of 32: fp32 operands loaded to registers one tile ahead, split into bf16 hi + lo planes while they are
// written to LDS (single buffer: 80 KB, two barriers per K tile).
__global__ __launch_bounds__(512, 2) void rollout_matmul_kernel(const float* A, const float* rowsum, const float* Xin, float* Xout, int T, int TP) {
    constexpr int BM = 256, BN = 256;
    __shared__ __attribute__((aligned(16))) bf16_t sm[4][BM * RM_LS];  // X hi, X lo, A hi, A lo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;  // this wave: rows 64 wr .., columns 128 wc ..
    const int b = blockIdx.z, n0 = blockIdx.y * BM, i0 = blockIdx.x * BN;
    const float* Ab = A + (size_t)b * T * TP;
    const float* Xb = Xin ? Xin + (size_t)b * T * TP : nullptr;
    const float* rs = rowsum + (size_t)b * T;
    const int nk = (T + RM_BK - 1) / RM_BK;

    f32x4 rx[4], ra[4];  // 256 rows x 32 k = 2048 chunks of 4 floats per operand, 4 per thread
    float rinv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid + 512 * i) >> 3;
        rinv[i] = (i0 + r < T) ? 1.0f / rs[i0 + r] : 0.f;
    }
    // interior: the workgroup's rows are all < T (both operands); then a K tile that ends inside T and does not meet the
    // diagonal of the A_hat rows needs no per-element masks (they were ~2/3 of the staging instructions)
    const bool rows_in = n0 + BM <= T && i0 + BN <= T && Xb != nullptr;
    auto gload = [&](int kt) {
        const int k0 = kt * RM_BK;
        if (rows_in && k0 + RM_BK <= T && (k0 + RM_BK <= i0 || k0 >= i0 + BN)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 512 * i, r = c >> 3, k = k0 + 4 * (c & 7);
                rx[i] = *(const f32x4*)(Xb + (size_t)(n0 + r) * TP + k);
                const f32x4 y = *(const f32x4*)(Ab + (size_t)(i0 + r) * TP + k);
                ra[i] = f32x4{y[0] * rinv[i], y[1] * rinv[i], y[2] * rinv[i], y[3] * rinv[i]};
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 512 * i, r = c >> 3, k = kt * RM_BK + 4 * (c & 7);
            const int n = n0 + r, ii = i0 + r;
            f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
            if (Xb) {
                if (n < T && k < TP) x = *(const f32x4*)(Xb + (size_t)n * TP + k);
            }
            if (ii < T && k < TP) y = *(const f32x4*)(Ab + (size_t)ii * TP + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool kin = k + e < T;
                if (!Xb) x[e] = (n == k + e) ? 1.f : 0.f;
                x[e] = (kin && n < T) ? x[e] : 0.f;
                y[e] = (kin && ii < T) ? (y[e] + (ii == k + e ? 1.f : 0.f)) * rinv[i] : 0.f;
            }
            rx[i] = x;
            ra[i] = y;
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 512 * i, r = c >> 3, off = r * RM_LS + 4 * (c & 7);
            bf16x4 xh, xl, ah, al;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[e] = (bf16_t)rx[i][e];
                xl[e] = (bf16_t)(rx[i][e] - (float)xh[e]);
                ah[e] = (bf16_t)ra[i][e];
                al[e] = (bf16_t)(ra[i][e] - (float)ah[e]);
            }
            *(bf16x4*)&sm[0][off] = xh;
            *(bf16x4*)&sm[1][off] = xl;
            *(bf16x4*)&sm[2][off] = ah;
            *(bf16x4*)&sm[3][off] = al;
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) zero16(acc[rb][nb]);
    gload(0);
    const int foff = (lane & 31) * RM_LS + 8 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        swrite();
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < RM_BK / 16; ++ks) {
            bf16x8 xh[2], xl[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                xh[rb] = *(const bf16x8*)&sm[0][(64 * wr + 32 * rb) * RM_LS + foff + 16 * ks];
                xl[rb] = *(const bf16x8*)&sm[1][(64 * wr + 32 * rb) * RM_LS + foff + 16 * ks];
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const bf16x8 ah = *(const bf16x8*)&sm[2][(128 * wc + 32 * nb) * RM_LS + foff + 16 * ks];
                const bf16x8 al = *(const bf16x8*)&sm[3][(128 * wc + 32 * nb) * RM_LS + foff + 16 * ks];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    acc[rb][nb] = mfma32(xl[rb], ah, acc[rb][nb]);
                    acc[rb][nb] = mfma32(xh[rb], al, acc[rb][nb]);
                    acc[rb][nb] = mfma32(xh[rb], ah, acc[rb][nb]);
                }
            }
        }
        __syncthreads();  // every wave is done with this K tile's planes
    }
    float* Ob = Xout + (size_t)b * T * TP;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int col = i0 + 128 * wc + 32 * nb + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + 64 * wr + 32 * rb + acc_row(r, lane);
                if (row < T && col < TP) Ob[(size_t)row * TP + col] = col < T ? acc[rb][nb][r] : 0.f;
            }
        }
}

template <int DP>
int launch_headmax_t(const AttnArgs& a, float* A, int TP, float* rowsum, int q_rows, hipStream_t s) {
    // q_rows > 0: only the first q_rows query rows (rounded up to whole 128-query workgroups) - the row chain's first step reads row 0 alone
    dim3 grid(((q_rows > 0 ? std::min(q_rows, a.T) : a.T) + 127) / 128, a.B);
    if (!rowsum) {  // per-head probabilities
        switch (a.H) {
            case 1: hipLaunchKernelGGL((rollout_headmax_kernel<DP, 1, true>), grid, dim3(256), 0, s, a, A, TP, rowsum); break;
            case 2: hipLaunchKernelGGL((rollout_headmax_kernel<DP, 2, true>), grid, dim3(256), 0, s, a, A, TP, rowsum); break;
            case 3: hipLaunchKernelGGL((rollout_headmax_kernel<DP, 3, true>), grid, dim3(256), 0, s, a, A, TP, rowsum); break;
            case 4: hipLaunchKernelGGL((rollout_headmax_kernel<DP, 4, true>), grid, dim3(256), 0, s, a, A, TP, rowsum); break;
            default: return V1T_ERR_UNSUPPORTED;
        }
        return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    }
    // V1T_HEADMAX8=0 (dev, A/B): the 4-wave kernel also where the 8-wave one applies (an even head count at head dims >= 128, where the
    // 4-wave kernel cannot fit two waves on a SIMD)
    static const bool hm8 = !(dev_env("V1T_HEADMAX8") && !atoi(dev_env("V1T_HEADMAX8")));
    if constexpr (DP >= 128) {
        if (hm8 && (a.H == 2 || a.H == 4)) {
            if (a.H == 2) hipLaunchKernelGGL((rollout_headmax8_kernel<DP, 2>), grid, dim3(512), 0, s, a, A, TP, rowsum);
            else hipLaunchKernelGGL((rollout_headmax8_kernel<DP, 4>), grid, dim3(512), 0, s, a, A, TP, rowsum);
            return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
        }
    }
    switch (a.H) {
        case 1: hipLaunchKernelGGL((rollout_headmax_kernel<DP, 1>), grid, dim3(256), 0, s, a, A, TP, rowsum); break;
        case 2: hipLaunchKernelGGL((rollout_headmax_kernel<DP, 2>), grid, dim3(256), 0, s, a, A, TP, rowsum); break;
        case 3: hipLaunchKernelGGL((rollout_headmax_kernel<DP, 3>), grid, dim3(256), 0, s, a, A, TP, rowsum); break;
        case 4: hipLaunchKernelGGL((rollout_headmax_kernel<DP, 4>), grid, dim3(256), 0, s, a, A, TP, rowsum); break;
        default: return V1T_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

}  // namespace

#define DP_DISPATCH(FN, ...)                           \
    switch (DP) {                                      \
        case 32: return FN<32>(__VA_ARGS__);           \
        case 64: return FN<64>(__VA_ARGS__);           \
        case 96: return FN<96>(__VA_ARGS__);           \
        case 128: return FN<128>(__VA_ARGS__);         \
        case 160: return FN<160>(__VA_ARGS__);         \
        default: return V1T_ERR_UNSUPPORTED;           \
    }

int launch_attn_fwd(const AttnArgs& a, int DP, hipStream_t s) { DP_DISPATCH(fwd_flags, a, s) }
int launch_attn_delta(const AttnArgs& a, int DP, float* delta, hipStream_t s) { DP_DISPATCH(launch_delta_t, a, delta, s) }
int launch_attn_rc_pad(const AttnArgs& a, hipStream_t s) {
    if (!a.ds) return V1T_ERR_ARG;
    const int TPQ = attn_ds_tpq(a.T), n = a.B * a.H * (TPQ - a.T);
    if (n <= 0) return V1T_OK;
    float* nlse = (float*)(a.ds + attn_ds_elems(a.B, a.H, a.T));
    hipLaunchKernelGGL(attn_rc_pad_kernel, dim3((n + 255) / 256), dim3(256), 0, s, nlse, nlse + attn_rc_floats(a.B, a.H, a.T), a.B * a.H, a.T, TPQ);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_attn_bwd(const AttnArgs& a, int DP, hipStream_t s) { DP_DISPATCH(bwd_flags, a, s) }

int launch_rollout_headmax(const AttnArgs& a, int DP, float* A, int TP, float* rowsum, int q_rows, hipStream_t s) {
    if (a.H * (DP > 96 ? 4 : 2) > 16) return V1T_ERR_UNSUPPORTED;  // Q fragments of all heads must fit the register file
    DP_DISPATCH(launch_headmax_t, a, A, TP, rowsum, q_rows, s)
}
int launch_rollout_matmul(const float* A, const float* rowsum, const float* Xin, float* Xout, int B, int T, int TP, hipStream_t s) {
    if (TP % 4 != 0 || TP < T || B > 65535) return V1T_ERR_ARG;
    dim3 grid((T + 255) / 256, (T + 255) / 256, B);
    prof_begin(PROF_ROLLOUT_MM, s);
    hipLaunchKernelGGL(rollout_matmul_kernel, grid, dim3(512), 0, s, A, rowsum, Xin, Xout, T, TP);
    prof_end(PROF_ROLLOUT_MM, s);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_rollout_vecmat(const float* A, const float* rowsum, const float* v, float* u, int B, int T, int TP, hipStream_t s) {
    dim3 grid((TP / 4 + 255) / 256, B);
    hipLaunchKernelGGL(rollout_vecmat_kernel, grid, dim3(256), sizeof(float) * T, s, A, rowsum, v, u, T, TP);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}

#ifdef V1T_KSUM
extern "C" int v1t_ksum_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ksum), sizeof(unsigned long long) * 66) == hipSuccess ? 0 : -1;
}
#endif
#ifdef V1T_KCLK
extern "C" int v1t_kclk_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kclk), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : -1;
}
#endif
#ifdef V1T_KPROF
extern "C" int v1t_kprof_read(unsigned long long* out, int n) {
    const int total = 8 * KP_NT * KP_NP;
    if (n < total) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kprof), sizeof(unsigned long long) * total) != hipSuccess) return -2;
    return total;
}
#endif
