// v1t_amd — C-ABI implementation: plan (dims, arena / shadow / workspace layouts) and the launch
// sequences of the ViT core forward/backward. See include/v1t_amd.h for the contract.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/v1t_amd.h"
#include "attention.h"
#include "elementwise.h"
#include "gemm.h"
#include "gridprep.h"
#include "readout.h"

namespace {

struct TensorInfo {
    std::string name;
    long long off;
    int ndim;
    long long shape[4];
    bool is_param;
    long long numel() const {
        long long n = 1;
        for (int i = 0; i < ndim; ++i) n *= shape[i];
        return n;
    }
};

struct BmlpOff {
    long long w1, b1, w3, b3;
};
struct BlockOff {
    long long ln1w, ln1b, qkv, proj, projb, scale, ln2w, ln2b, fc1, fc1b, fc2, fc2b;
    std::vector<BmlpOff> bmlp;
    // shadow (bytes)
    long long s_qkv, s_qkv_t, s_proj, s_proj_t, s_fc1, s_fc1_t, s_fc2, s_fc2_t, s_projb, s_fc1b, s_fc2b;
    long long s_qkv_lo, s_proj_lo, s_fc1_lo, s_fc2_lo;  // low planes of the forward weights (split-bf16)
};

inline long long align_up(long long x, long long a) { return (x + a - 1) / a * a; }

}  // namespace

struct v1t_vit {
    v1t_vit_config c;
    int C, IH, IW, P, S, NH, NW, L, T, D, DP, H, HD, HDP, M, MP, NB, IN, J, PD;
    int HE, HEP;      // head dim (unpadded / padded to 32): emb_dim for the ViT core (vit.py:218), emb_dim / heads^2 for CCT (cct.py:110-127)
    int cls;          // class tokens in front of the patches: 1 (ViT) or 0 (CCT)
    bool cct;         // core_kind 1
    int CH, CW, RCI;  // CCT: conv output grid (before the max pool), rows of the unfolded-patch matrix per image (CH * CW; ViT: T)
    long long o_cpos; // CCT: fixed position table (buffer) or -1
    int gh, gw;
    bool inject;
    int nbmlp;
    std::vector<TensorInfo> tensors;
    long long arena_floats, param_floats;
    long long o_cls, o_pos, o_pw, o_pb;
    long long o_pln_w, o_pln_b, o_pln2_w, o_pln2_b;  // patch modes 2/3: LayerNorm over the patch, mode 3: LayerNorm over D (-1: absent)
    long long s_pw_t;               // patch modes 2/3: transposed hi plane [PD][DP] (dU = gd . W)
    long long s_pw, s_pw_lo, s_pb;  // patch weights: bf16 hi / lo planes [DP][PD], fp32 bias [DP] (-1: VALU patch kernels)
    int PDX;                        // row stride of the unfolded-patch matrix U (PD + ones column, padded to 128)
    std::vector<BlockOff> blk;
    long long shadow_bytes;
    std::vector<PackDesc> pack;
    mutable PackDesc* d_pack;  // uploaded lazily by the first v1t_vit_pack (create works without a GPU)
    // backward of SMALL launches (a rank's share of a multi-GPU step): the four weight-gradient GEMMs of a block run on a second stream beside
    // the dX GEMMs of the same gradients (created by the first such backward)
    mutable hipStream_t dw_stream = nullptr;
    mutable hipEvent_t dw_ready[4] = {}, dw_done[4] = {};

    long long add(const std::string& name, std::initializer_list<long long> shape, bool is_param, long long& cursor) {
        TensorInfo t;
        t.name = name;
        t.off = cursor;
        t.ndim = (int)shape.size();
        int i = 0;
        for (auto s : shape) t.shape[i++] = s;
        for (; i < 4; ++i) t.shape[i] = 1;
        t.is_param = is_param;
        cursor += t.numel();
        tensors.push_back(t);
        return t.off;
    }
};

namespace {

// -------------------------------------------------------------------------- workspace layout
struct WsLayout {
    long long x0, beta, hid, u_hi, u_lo;
    long long u32, pmean1, prstd1, py, pmean2, prstd2;  // patch modes 2/3 (0 bytes otherwise)
    long long cconv, cidx;                              // CCT tokenizer: conv output fp32 [B*CH*CW][DP], arg-max of the pool windows u8 [B*L][DP]
    // per block
    long long blk_stride, xa, xm, xo, z1, qkv, o, lse2, mean1, rstd1, z2, mean2, rstd2, hpre, hact;
    long long z1_lo, o_lo, z2_lo, hact_lo;  // low planes (forward-only consumers)
    long long total;
};

// V1T_NOSPLIT=<mask> (dev, numerics ablation): bit 0/1/2/3 runs QKV / proj / FC1 / FC2 of the forward in plain bf16
static const int g_nosplit = dev_env("V1T_NOSPLIT") ? atoi(dev_env("V1T_NOSPLIT")) : 0;
// Operand format of the forward linear layers (patch projection, QKV, proj, FC1, FC2): fp16 planes + one fp16 MFMA per
// step (default; 0.28 of the 1e-3 parity bound on the default V1T), or V1T_FWD_BF16X3=1 (dev, numerics ablation): bf16
// hi + lo planes and three MFMAs (0.07 of the bound, 2.4x the GEMM time). The second plane of every forward activation /
// weight holds the one or the other; the bf16 "hi" planes are what the backward reads (except the
// attention output / GELU output in fp16 mode: x16_attn_out / x16_gelu_out below).
static const int g_fwd_f16 = (dev_env("V1T_FWD_BF16X3") && atoi(dev_env("V1T_FWD_BF16X3"))) ? 0 : 1;
static bool x16_attn_out(const v1t_vit* h, long long R);
static bool x16_gelu_out(const v1t_vit* h, long long R);

WsLayout ws_layout(const v1t_vit* h, int B, bool save) {
    WsLayout w;
    const long long R = (long long)B * h->T;
    long long cur = 0;
    auto take = [&](long long bytes) {
        const long long o = cur;
        cur = align_up(cur + bytes, 256);
        return o;
    };
    w.x0 = take(R * h->DP * 4);
    w.beta = take((long long)h->NB * B * h->DP * 4);
    w.hid = take((long long)h->NB * B * std::max(h->J, 1) * 4);
    const long long RU = (long long)B * h->RCI;  // rows of the unfolded-patch matrix
    w.u_hi = take(h->s_pw >= 0 ? RU * h->PDX * 2 : 0);
    w.u_lo = take(h->s_pw >= 0 ? RU * h->PDX * 2 : 0);
    w.cconv = take(h->cct ? RU * h->DP * 4 : 0);
    w.cidx = take(h->cct ? R * h->DP : 0);
    const bool pln = h->c.patch_mode >= 2, pln2 = h->c.patch_mode == 3;
    w.u32 = take(pln ? R * h->PDX * 4 : 0);
    w.pmean1 = take(pln ? R * 4 : 0);
    w.prstd1 = take(pln ? R * 4 : 0);
    w.py = take(pln2 ? R * h->DP * 4 : 0);
    w.pmean2 = take(pln2 ? R * 4 : 0);
    w.prstd2 = take(pln2 ? R * 4 : 0);
    const long long b0 = cur;
    w.xa = take(R * h->DP * 4) - b0;
    w.xm = take(R * h->DP * 4) - b0;
    w.xo = take(R * h->DP * 4) - b0;
    w.z1 = take(R * h->DP * 2) - b0;
    w.qkv = take(R * 3 * h->HDP * 2) - b0;
    w.o = take(x16_attn_out(h, R) ? 0 : R * h->HDP * 2) - b0;  // no bf16 plane where the backward reads the fp16 one
    w.lse2 = take((long long)B * h->H * h->T * 4) - b0;
    w.mean1 = take(R * 4) - b0;
    w.rstd1 = take(R * 4) - b0;
    w.z2 = take(R * h->DP * 2) - b0;
    w.mean2 = take(R * 4) - b0;
    w.rstd2 = take(R * 4) - b0;
    w.hpre = take((R + 127) / 128 * 128 * h->MP * 2) - b0;  // gelu' in accumulator-fragment order (full 128-row tiles)
    w.hact = take(x16_gelu_out(h, R) ? 0 : R * h->MP * 2) - b0;
    w.z1_lo = take(R * h->DP * 2) - b0;
    w.o_lo = take(R * h->HDP * 2) - b0;
    w.z2_lo = take(R * h->DP * 2) - b0;
    w.hact_lo = take(R * h->MP * 2) - b0;
    w.blk_stride = cur - b0;
    w.total = b0 + (save ? (long long)h->NB : 1LL) * w.blk_stride;
    w.xa += b0; w.xm += b0; w.xo += b0; w.z1 += b0; w.qkv += b0; w.o += b0; w.lse2 += b0;
    w.mean1 += b0; w.rstd1 += b0; w.z2 += b0; w.mean2 += b0; w.rstd2 += b0; w.hpre += b0; w.hact += b0;
    w.z1_lo += b0; w.o_lo += b0; w.z2_lo += b0; w.hact_lo += b0;
    return w;
}

// Attention backward with the materialised dS' (attention.h): the dK/dV kernel stores dS' tile-major and dQ = dS' . K is a
// streaming GEMM, instead of the fused kernel whose dQ body recomputes S and dP (7 MFMA products -> 5). Standalone at the
// default shape it is a draw (dK/dV + store 378 us + GEMM 120 us = 498 us against 497 us fused, dropout on; 470 against 490
// without dropout); inside the training step it is 1.6 % of the whole step faster (3062 against 3015 images/s, twice),
// so it is the default. V1T_ATTN_BWD_DS=0 (dev) selects the fused recompute kernel, which LSA (mask_diag) always uses.
static const int g_attn_ds = (dev_env("V1T_ATTN_BWD_DS") && !atoi(dev_env("V1T_ATTN_BWD_DS"))) ? 0 : 1;
struct ScratchLayout {
    long long G, dy, dy2, dy3, dhpre, dz, dO, delta, dqkv, dqkv2, dbeta, slab, pu, pgd, pdu, ds, total;
};
// contraction rows per workgroup of the weight-gradient GEMMs: aim at >= ~512 workgroups at full-size launches. Launches under 65 536 rows (a
// rank's share of a multi-GPU step, the 16-image launches of the per-mouse loop) run beside the dX GEMMs on the second stream and are bound by
// their fixed parts - prologue, partial-tile stores and the slab reduction, which reads one partial tile per m-chunk - not by idle CUs: 256
// there (half the m-chunks, half the slab traffic). A/B in one call at a 14-image share: 512: 3.67 / 3.69 ms per step, 256: 3.60 / 3.57,
// 128: 3.62 / 3.56, 64: 3.61 / 3.68; the 7 x 16-image loop: 28.8 -> 28.3 ms (profiles/r05_small_launch_experiments.txt)
int tn_mchunk(long long R, int tiles) {
    static const int forced = dev_env("V1T_TN_WGS") ? atoi(dev_env("V1T_TN_WGS")) : 0;  // dev switch
    const int target = forced > 0 ? forced : (R < 65536 ? 256 : 512);
    const int want = std::max(1, target / std::max(tiles, 1));
    const int mc = (int)round_up((R + want - 1) / want, 64);
    return std::max(mc, 128);
}
struct TnPlan { int mc_fc2, mc_fc1, mc_proj, mc_qkv, mc_patch; size_t slab; };
TnPlan tn_plan(const v1t_vit* h, long long R) {
    TnPlan p;
    const int DP = h->DP, MP = h->MP, HDP = h->HDP;
    p.mc_fc2 = tn_mchunk(R, ((DP + 127) / 128) * (MP / 128 > 0 ? MP / 128 : 1));
    p.mc_fc1 = tn_mchunk(R, (MP + 127) / 128);
    p.mc_proj = tn_mchunk(R, ((DP + 127) / 128) * h->H);
    p.mc_qkv = tn_mchunk(R, (3 * HDP + 127) / 128);
    const long long RU = R / h->T * h->RCI;
    p.mc_patch = tn_mchunk(RU, ((DP + 159) / 160) * (h->PDX / 128));
    p.slab = std::max(std::max(gemm_tn_slab_bytes((int)R, DP, MP, p.mc_fc2), gemm_tn_slab_bytes((int)R, MP, DP, p.mc_fc1)),
                      std::max(gemm_tn_slab_bytes((int)R, DP, HDP, p.mc_proj), gemm_tn_slab_bytes((int)R, 3 * HDP, DP, p.mc_qkv)));
    if (h->s_pw >= 0) p.slab = std::max(p.slab, gemm_tn_slab_bytes((int)RU, DP, h->PDX, p.mc_patch));
    return p;
}

// The attention output and the GELU output exist as two 16-bit planes: bf16 (the X operand of the weight-gradient GEMMs dWo / dW2 and of
// the row constants) and fp16 (the A operand of the forward's projection / FC2 GEMM). Where the weight-gradient kernel can convert its X
// fragments (gemm_tn_takes_f16_x) the bf16 plane is never written: the backward reads the fp16 one (237 + 190 MB less per block and
// 112-image step at the default shape).
static const bool g_keep_bf16 = dev_env("V1T_KEEP_BF16_PLANES") != nullptr;  // dev (A/B): write and read the bf16 planes as before
static bool x16_attn_out(const v1t_vit* h, long long R) {
    return !g_keep_bf16 && g_fwd_f16 && !(g_nosplit & 2) && gemm_tn_takes_f16_x(h->DP, h->HDP, tn_plan(h, R).mc_proj);
}
static bool x16_gelu_out(const v1t_vit* h, long long R) {
    return !g_keep_bf16 && g_fwd_f16 && !(g_nosplit & 8) && gemm_tn_takes_f16_x(h->DP, h->MP, tn_plan(h, R).mc_fc2);
}

// Weight-gradient GEMMs on a second stream beside the main stream's kernels (backward, below): for launches under 262 144 rows, or as
// V1T_DW_SIDE=0 / 1 forces (dev). One place decides for the scratch layout (four slab regions, second dqkv) and for the backward.
// Round 5, with the hand-over once per block (same call, on / off): 14 images 3.46-3.49 vs 3.63-3.64 ms per step, 28 images 6.05-6.15 vs
// 6.20-6.22, 56 images 11.08-11.23 vs 11.29-11.38, 112 images 20.79-20.89 vs 20.99-21.00 and, on the final build, 20.69 / 20.74 vs 20.96 / 21.01
// (bench.py). The HBM-bound weight-gradient GEMMs of a block then share the chip with the compute-bound dK/dV kernel of the next one, whose
// launches take 1.83 instead of 1.61 ms live: at 112 images the bench line's dominant-kernel time includes that sharing (it says so:
// roofline.shares_gpu_with) - the step is 1.3 % faster for it.
static bool dw_side_for(long long R) {
    static const int dw_force = std::getenv("V1T_DW_SIDE") ? atoi(std::getenv("V1T_DW_SIDE")) : -1;
    return dw_force >= 0 ? dw_force > 0 : R < 262144;
}

ScratchLayout scratch_layout(const v1t_vit* h, int B) {
    ScratchLayout s;
    const long long R = (long long)B * h->T;
    long long cur = 0;
    auto take = [&](long long bytes) {
        const long long o = cur;
        cur = align_up(cur + bytes, 256);
        return o;
    };
    s.G = take(R * h->DP * 4);
    s.dy = take(R * h->DP * 2);
    s.dy2 = take(R * h->DP * 2);  // the attention branch's output gradient (dyp, backward below)
    s.dy3 = take(R * h->DP * 2);  // the MLP branch's output gradient of odd blocks (dy alternates by block parity, backward below)
    s.dhpre = take(R * h->MP * 2);
    s.dz = take(R * h->DP * 4);
    s.dO = take(R * h->HDP * 2);
    s.delta = take((long long)B * h->H * h->T * 4);
    s.dqkv = take(R * 3 * h->HDP * 2);
    s.dqkv2 = take(dw_side_for(R) ? R * 3 * h->HDP * 2 : 0);  // odd blocks' dqkv where the weight-gradient GEMMs run a block behind (second stream)
    s.dbeta = take((long long)h->NB * B * h->DP * 4);
    // one region per weight-gradient GEMM of a block where they run beside the dX GEMMs (second stream), else one region they share in turn
    s.slab = take((dw_side_for(R) ? 4 : 1) * (((long long)tn_plan(h, R).slab + 255) / 256 * 256));
    const long long RU = (long long)B * h->RCI;
    s.pu = take(h->s_pw >= 0 ? RU * h->PDX * 2 : 0);
    s.pgd = take(h->s_pw >= 0 ? RU * h->DP * 2 : 0);
    s.pdu = take(h->c.patch_mode >= 2 ? R * h->PD * 4 : 0);
    s.ds = take(g_attn_ds && !h->c.use_lsa ? (long long)attn_ds_bytes(B, h->H, h->T) : 0);  // materialised dS' of one block
    s.total = cur;
    return s;
}

DropCfg make_drop(bool training, float p, uint64_t seed, uint32_t stream) {
    DropCfg d;
    d.key = drop_key(seed, stream);
    d.thresh = 0;
    d.inv_keep = 1.f;
    if (training && p > 0.f) {
        double t = std::floor((double)p * 4294967296.0 + 0.5);
        if (t > 4294967295.0) t = 4294967295.0;
        d.thresh = (uint32_t)t;
        if (d.thresh == 0) d.thresh = 1;
        d.inv_keep = (float)(1.0 / (1.0 - (double)p));
    }
    return d;
}

// LayerNorm followed by the GEMM that reads it: one fused A-stationary launch where the shape allows (gemm.h, launch_ln_gemm;
// fp16 operands, K = DP <= 160, N % 128 == 0), else the two kernels. V1T_LN_FUSE=0 (dev): always the two kernels.
static const int g_ln_fuse = (dev_env("V1T_LN_FUSE") && !atoi(dev_env("V1T_LN_FUSE"))) ? 0 : 1;
static const int g_mlp_fuse = (dev_env("V1T_MLP_FUSE") && !atoi(dev_env("V1T_MLP_FUSE"))) ? 0 : 1;
static inline int ln_then_gemm(const LnFwdArgs& l, const GemmNTArgs& g, int epi, hipStream_t s) {
    if (g_ln_fuse) {
        const int rc = launch_ln_gemm(l, g, epi, s);
        if (rc != V1T_ERR_UNSUPPORTED) return rc;
    }
    const int rc = launch_ln_fwd(l, s);
    return rc != V1T_OK ? rc : launch_gemm_nt(g, epi, s);
}
static inline void fwd_operands(GemmNTArgs& g) {
    if (!g_fwd_f16) return;
    g.f16 = 1;  // the producers wrote fp16 into the second planes (EPI_BIAS_GELU writes C2_lo the same way)
    if (g.A_lo && g.B_lo) { g.A = g.A_lo; g.B = g.B_lo; }
    else g.f16 = 0;  // V1T_NOSPLIT ablation: plain bf16
    g.A_lo = g.B_lo = nullptr;
}
// V1T_DEBUG_SYNC=1: print the launch about to be made and synchronise after it (fault localisation).
static const bool g_debug_sync = std::getenv("V1T_DEBUG_SYNC") != nullptr;
 AttnDrop make_adrop(bool training, float p, uint64_t seed, uint32_t stream) {
    AttnDrop d;
    d.key = drop_key(seed, stream);
    d.thresh16 = d.thresh8 = d.frac8 = 0;
    d.inv_keep = 1.f;
    d.keep_prob = 1.f;
    if (training && p > 0.f) {
        // rate = t / 65536 (byte decisions against a per-tile dithered threshold, common.h); a rate below 2^-17 rounds to "no dropout at this
        // site", above 1 - 2^-17 to 65535 / 65536
        const int t = std::min((int)std::floor((double)p * 65536.0 + 0.5), 65535);
        if (t > 0) {
            d.thresh16 = (uint32_t)t;
            d.thresh8 = (uint32_t)t >> 8;
            d.frac8 = (uint32_t)t & 255u;
            d.inv_keep = (float)(65536.0 / (double)(65536 - t));
            d.keep_prob = (float)((double)(65536 - t) / 65536.0);
        }
    }
    return d;
}

#define CHECK(x)                                                                      \
    do {                                                                              \
        if (g_debug_sync) { std::fprintf(stderr, "[v1t] %s:%d %s\n", __func__, __LINE__, #x); std::fflush(stderr); } \
        const int _e = (x);                                                           \
        if (_e != V1T_OK) return _e;                                                  \
        if (g_debug_sync && hipDeviceSynchronize() != hipSuccess) { std::fprintf(stderr, "[v1t] FAILED after %s\n", #x); return V1T_ERR_LAUNCH; } \
    } while (0)

int find_shape(int n, int* h, int* w) {
    int d1 = (int)std::ceil(std::sqrt((double)n));
    while (d1 > 0 && n % d1 != 0) --d1;
    *h = d1;
    *w = n / d1;
    return 0;
}

}  // namespace

// ---- per-launch event timing of one kernel class
namespace {
struct Prof {
    int cls = -1;
    std::vector<hipEvent_t> ev;  // start/stop pairs
    int used = 0;
} g_prof;
}  // namespace
void prof_begin(int cls, hipStream_t s) {
    if (cls != g_prof.cls || g_prof.used + 2 > (int)g_prof.ev.size()) return;
    (void)hipEventRecord(g_prof.ev[g_prof.used], s);
}
void prof_end(int cls, hipStream_t s) {
    if (cls != g_prof.cls || g_prof.used + 2 > (int)g_prof.ev.size()) return;
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s);
    g_prof.used += 2;
}

extern "C" {

int v1t_profile_enable(int kernel_class, int max_launches) {
    for (auto e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.used = 0;
    g_prof.cls = kernel_class;
    if (kernel_class < 0) return V1T_OK;
    g_prof.ev.resize(2 * (size_t)std::max(max_launches, 0));
    for (auto& e : g_prof.ev)
        if (hipEventCreate(&e) != hipSuccess) return V1T_ERR_LAUNCH;
    return V1T_OK;
}
int v1t_profile_read(int* launches, double* total_ms) {
    double tot = 0.0;
    for (int i = 0; i + 1 < g_prof.used; i += 2) {
        if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) return V1T_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess) return V1T_ERR_LAUNCH;
        tot += ms;
    }
    if (launches) *launches = g_prof.used / 2;
    if (total_ms) *total_ms = tot;
    g_prof.used = 0;
    return V1T_OK;
}

int v1t_abi_version(void) { return 1; }

const char* v1t_error_string(int code) {
    switch (code) {
        case V1T_OK: return "ok";
        case V1T_ERR_ARG: return "invalid argument";
        case V1T_ERR_UNSUPPORTED: return "configuration not supported by the gfx950 kernels";
        case V1T_ERR_LAUNCH: return "HIP kernel launch failed";
        case V1T_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

int v1t_vit_create(const v1t_vit_config* cfg, v1t_vit** out) {
    if (!cfg || !out) return V1T_ERR_ARG;
    const bool cct = cfg->core_kind == 1;
    if (cfg->core_kind != 0 && cfg->core_kind != 1) return V1T_ERR_UNSUPPORTED;
    if (!cct && (cfg->patch_mode < 0 || cfg->patch_mode > 3)) return V1T_ERR_UNSUPPORTED;
    if (!cct && cfg->patch_mode == 2 && cfg->in_channels != 1) return V1T_ERR_UNSUPPORTED;  // vit.py:84 sizes the SPT patch as (c + 4) * P^2: single-channel only
    if (cfg->patch_stride < 1 || cfg->patch_stride > cfg->patch_size) return V1T_ERR_ARG;
    if (cfg->behavior_mode != 0 && cfg->behavior_mode != 2 && cfg->behavior_mode != 3 && cfg->behavior_mode != 4) return V1T_ERR_ARG;
    if (cct && (cfg->behavior_mode == 2 || cfg->conv_pad < 0 || cfg->pos_mode < 0 || cfg->pos_mode > 1)) return V1T_ERR_ARG;  // core.py:27-28: mode 2 is ViT-only
    v1t_vit* h = new v1t_vit();
    h->c = *cfg;
    if (cct) { h->c.patch_mode = 0; h->c.use_lsa = 0; h->c.use_bias = 1; }
    cfg = &h->c;
    h->cct = cct;
    h->cls = cct ? 0 : 1;
    h->C = cfg->in_channels; h->IH = cfg->in_h; h->IW = cfg->in_w; h->P = cfg->patch_size; h->S = cfg->patch_stride;
    h->CH = h->CW = 0;
    if (cct) {  // Conv2d(k = P, stride = S, padding = conv_pad) then MaxPool2d(kernel 3, stride 2, padding 1) (cct.py:46-56)
        h->CH = (h->IH + 2 * cfg->conv_pad - h->P) / h->S + 1;
        h->CW = (h->IW + 2 * cfg->conv_pad - h->P) / h->S + 1;
        if (h->CH < 1 || h->CW < 1) { delete h; return V1T_ERR_ARG; }
        h->NH = (h->CH + 2 - 3) / 2 + 1;
        h->NW = (h->CW + 2 - 3) / 2 + 1;
    } else {
        h->NH = (h->IH - h->P) / h->S + 1;
        h->NW = (h->IW - h->P) / h->S + 1;
    }
    if (h->NH < 1 || h->NW < 1) { delete h; return V1T_ERR_ARG; }
    h->L = h->NH * h->NW; h->T = h->L + h->cls;
    h->RCI = cct ? h->CH * h->CW : h->T;
    h->D = cfg->emb_dim; h->DP = round_up(h->D, 32);
    h->H = cfg->num_heads;
    if (cct) {  // inner = emb_dim // heads is the WHOLE qkv width of one of q / k / v; it is then cut into `heads` heads (cct.py:110-127)
        const int inner = h->D / h->H;
        if (h->H < 1 || inner < h->H || inner % h->H != 0) { delete h; return V1T_ERR_ARG; }  // the reference's assert (cct.py:111-113)
        h->HE = inner / h->H;
    } else {
        h->HE = h->D;
    }
    h->HEP = round_up(h->HE, 32);
    h->HD = h->H * h->HE; h->HDP = h->H * h->HEP;
    h->M = cfg->mlp_dim; h->MP = round_up(h->M, 32);
    h->NB = cfg->num_blocks;
    h->inject = cfg->behavior_mode == 2 || cfg->behavior_mode == 3 || cfg->behavior_mode == 4;
    h->IN = cfg->behavior_mode == 2 ? 3 : 5;
    h->J = h->D / 2;
    h->PD = (cfg->patch_mode == 2 ? h->C + 4 : h->C) * h->P * h->P;
    h->nbmlp = cfg->behavior_mode == 4 ? std::max(cfg->num_mice, 1) : 1;
    if (h->DP > 160 || (h->DP != 32 && h->DP != 64 && h->DP != 96 && h->DP != 128 && h->DP != 160)) { delete h; return V1T_ERR_UNSUPPORTED; }
    if (h->HEP > 160) { delete h; return V1T_ERR_UNSUPPORTED; }
    find_shape(h->L, &h->gh, &h->gw);

    // ---- parameter arena (natural shapes, reference state-dict names)
    long long cur = 0;
    const bool bias = cfg->use_bias != 0;
    h->o_cls = h->o_pos = h->o_pb = h->o_cpos = -1;
    h->o_pln_w = h->o_pln_b = h->o_pln2_w = h->o_pln2_b = -1;
    if (cct) {
        h->o_pw = h->add("tokenizer.conv2d.weight", {h->D, h->C, h->P, h->P}, true, cur);
    } else {
        h->o_cls = h->add("patch_embedding.cls_token", {1, 1, h->D}, true, cur);
        h->o_pos = h->add("patch_embedding.pos_embedding", {h->T, h->D}, true, cur);
    }
    if (cct) {
    } else if (cfg->patch_mode == 0) {
        h->o_pw = h->add("patch_embedding.projection.2.weight", {h->D, h->PD}, true, cur);
        h->o_pb = h->add("patch_embedding.projection.2.bias", {h->D}, true, cur);
    } else if (cfg->patch_mode == 2) {  // PatchShifting, Unfold, Rearrange, LayerNorm(patch), Linear (vit.py:83-91)
        h->o_pln_w = h->add("patch_embedding.projection.3.weight", {h->PD}, true, cur);
        h->o_pln_b = h->add("patch_embedding.projection.3.bias", {h->PD}, true, cur);
        h->o_pw = h->add("patch_embedding.projection.4.weight", {h->D, h->PD}, true, cur);
        h->o_pb = h->add("patch_embedding.projection.4.bias", {h->D}, true, cur);
    } else if (cfg->patch_mode == 3) {  // Unfold, Rearrange, LayerNorm(patch), Linear, LayerNorm(D) (vit.py:92-100)
        h->o_pln_w = h->add("patch_embedding.projection.2.weight", {h->PD}, true, cur);
        h->o_pln_b = h->add("patch_embedding.projection.2.bias", {h->PD}, true, cur);
        h->o_pw = h->add("patch_embedding.projection.3.weight", {h->D, h->PD}, true, cur);
        h->o_pb = h->add("patch_embedding.projection.3.bias", {h->D}, true, cur);
        h->o_pln2_w = h->add("patch_embedding.projection.4.weight", {h->D}, true, cur);
        h->o_pln2_b = h->add("patch_embedding.projection.4.bias", {h->D}, true, cur);
    } else {
        h->o_pw = h->add("patch_embedding.projection.0.weight", {h->D, h->C, h->P, h->P}, true, cur);
        h->o_pb = h->add("patch_embedding.projection.0.bias", {h->D}, true, cur);
    }
    h->blk.resize(h->NB);
    for (int k = 0; k < h->NB; ++k) {
        BlockOff& b = h->blk[k];
        const std::string p = "transformer.blocks." + std::to_string(k) + ".";
        const std::string mlp = cct ? "mlp." : "mlp.model.";  // cct.py:161-168: the Sequential itself is the attribute
        b.ln1w = h->add(p + "mha.layer_norm.weight", {h->D}, true, cur);
        b.ln1b = h->add(p + "mha.layer_norm.bias", {h->D}, true, cur);
        b.qkv = h->add(p + (cct ? "mha.qkv.weight" : "mha.to_qkv.weight"), {3LL * h->HD, h->D}, true, cur);
        b.proj = h->add(p + "mha.projection.0.weight", {h->D, h->HD}, true, cur);
        b.projb = bias ? h->add(p + "mha.projection.0.bias", {h->D}, true, cur) : -1;
        b.scale = cfg->use_lsa ? h->add(p + "mha.scale", {h->H}, true, cur) : -1;
        b.ln2w = h->add(p + mlp + "0.weight", {h->D}, true, cur);
        b.ln2b = h->add(p + mlp + "0.bias", {h->D}, true, cur);
        b.fc1 = h->add(p + mlp + "1.weight", {h->M, h->D}, true, cur);
        b.fc1b = bias ? h->add(p + mlp + "1.bias", {h->M}, true, cur) : -1;
        b.fc2 = h->add(p + mlp + "4.weight", {h->D, h->M}, true, cur);
        b.fc2b = bias ? h->add(p + mlp + "4.bias", {h->D}, true, cur) : -1;
        if (h->inject) {
            b.bmlp.resize(h->nbmlp);
            for (int m = 0; m < h->nbmlp; ++m) {
                const std::string q = p + (cct ? "b_mlp.models." : "b-mlp.models.") + (cfg->behavior_mode == 4 ? "@" + std::to_string(m) : std::string("share")) + ".";
                b.bmlp[m].w1 = h->add(q + "0.weight", {h->J, h->IN}, true, cur);
                b.bmlp[m].b1 = bias ? h->add(q + "0.bias", {h->J}, true, cur) : -1;
                b.bmlp[m].w3 = h->add(q + "3.weight", {h->D, h->J}, true, cur);
                b.bmlp[m].b3 = bias ? h->add(q + "3.bias", {h->D}, true, cur) : -1;
            }
        }
    }
    h->param_floats = cur;
    if (!cfg->use_lsa)
        for (int k = 0; k < h->NB; ++k)
            h->blk[k].scale = h->add("transformer.blocks." + std::to_string(k) + ".mha.scale", {}, false, cur);
    if (cct && cfg->pos_mode == 1) h->o_cpos = h->add("tokenizer.pos_embedding", {1, h->L, h->D}, false, cur);
    h->arena_floats = cur;

    // ---- shadow layout + pack table
    long long sc = 0;
    auto stake = [&](long long bytes) {
        const long long o = sc;
        sc = align_up(sc + bytes, 256);
        return o;
    };
    auto desc = [&](long long src, int src_ld, long long dst, int drows, int dcols, int rp, int rv, int cp, int cv, int tr, int f32) {
        PackDesc d;
        d.src_off = src; d.dst_off = dst; d.src_ld = src_ld; d.drows = drows; d.dcols = dcols;
        d.rseg_pad = rp; d.rseg_valid = rv; d.cseg_pad = cp; d.cseg_valid = cv; d.transpose = tr; d.out_f32 = f32 & 1; d.lo_plane = (f32 & 4) ? 1 : ((f32 >> 1) & 1 ? (g_fwd_f16 ? 2 : 1) : 0);  // 4: always the bf16 residual (split-bf16 operand)
        h->pack.push_back(d);
    };
    for (int k = 0; k < h->NB; ++k) {
        BlockOff& b = h->blk[k];
        const int DP = h->DP, D = h->D, MP = h->MP, M = h->M, HDP = h->HDP;
        const int HEP = h->HEP, HE = h->HE;  // head segments of the qkv rows / proj columns
        b.s_qkv = stake(3LL * HDP * DP * 2);   desc(b.qkv, D, b.s_qkv, 3 * HDP, DP, HEP, HE, DP, D, 0, 0);
        b.s_qkv_t = stake(3LL * HDP * DP * 2); desc(b.qkv, D, b.s_qkv_t, DP, 3 * HDP, HEP, HE, DP, D, 1, 0);
        b.s_proj = stake((long long)DP * HDP * 2);   desc(b.proj, h->HD, b.s_proj, DP, HDP, DP, D, HEP, HE, 0, 0);
        b.s_proj_t = stake((long long)DP * HDP * 2); desc(b.proj, h->HD, b.s_proj_t, HDP, DP, DP, D, HEP, HE, 1, 0);
        b.s_fc1 = stake((long long)MP * DP * 2);   desc(b.fc1, D, b.s_fc1, MP, DP, MP, M, DP, D, 0, 0);
        b.s_fc1_t = stake((long long)MP * DP * 2); desc(b.fc1, D, b.s_fc1_t, DP, MP, MP, M, DP, D, 1, 0);
        b.s_fc2 = stake((long long)DP * MP * 2);   desc(b.fc2, M, b.s_fc2, DP, MP, DP, D, MP, M, 0, 0);
        b.s_fc2_t = stake((long long)DP * MP * 2); desc(b.fc2, M, b.s_fc2_t, MP, DP, DP, D, MP, M, 1, 0);
        b.s_qkv_lo = stake(3LL * HDP * DP * 2);       desc(b.qkv, D, b.s_qkv_lo, 3 * HDP, DP, HEP, HE, DP, D, 0, 2);
        b.s_proj_lo = stake((long long)DP * HDP * 2); desc(b.proj, h->HD, b.s_proj_lo, DP, HDP, DP, D, HEP, HE, 0, 2);
        b.s_fc1_lo = stake((long long)MP * DP * 2);   desc(b.fc1, D, b.s_fc1_lo, MP, DP, MP, M, DP, D, 0, 2);
        b.s_fc2_lo = stake((long long)DP * MP * 2);   desc(b.fc2, M, b.s_fc2_lo, DP, MP, DP, D, MP, M, 0, 2);
        b.s_projb = b.s_fc1b = b.s_fc2b = -1;
        if (bias) {
            b.s_projb = stake(DP * 4); desc(b.projb, D, b.s_projb, 1, DP, 1, 1, DP, D, 0, 1);
            b.s_fc1b = stake(MP * 4);  desc(b.fc1b, M, b.s_fc1b, 1, MP, 1, 1, MP, M, 0, 1);
            b.s_fc2b = stake(DP * 4);  desc(b.fc2b, D, b.s_fc2b, 1, DP, 1, 1, DP, D, 0, 1);
        }
    }
    h->s_pw = h->s_pw_lo = h->s_pb = h->s_pw_t = -1;
    h->PDX = (h->PD + 1 + 127) / 128 * 128;
    if (cfg->patch_mode >= 2 && (h->PD % 32 != 0 || h->PDX > 384)) { delete h; return V1T_ERR_UNSUPPORTED; }
    if (cfg->patch_mode >= 2) {
        h->s_pw_t = stake((long long)h->PD * h->DP * 2); desc(h->o_pw, h->PD, h->s_pw_t, h->PD, h->DP, h->DP, h->D, h->PD, h->PD, 1, 0);
    }
    if (cct && h->PD % 32 != 0) { delete h; return V1T_ERR_UNSUPPORTED; }  // the conv tokenizer exists as the MFMA GEMM over unfolded patches only
    if (h->PD % 32 == 0) {  // MFMA patch embedding (both patch modes store the weight as [D][C*P*P])
        h->s_pw = stake((long long)h->DP * h->PD * 2);    desc(h->o_pw, h->PD, h->s_pw, h->DP, h->PD, h->DP, h->D, h->PD, h->PD, 0, 0);
        // (CCT: the conv runs in split-bf16 - three MFMA products, ~2^-17 - whatever the other linears use: the arg-max of the
        // max pool behind it decides where gradients go, and near-ties flip 30x more often at fp16 operand precision; it is 0.1 %
        // of the model's FLOPs)
        h->s_pw_lo = stake((long long)h->DP * h->PD * 2); desc(h->o_pw, h->PD, h->s_pw_lo, h->DP, h->PD, h->DP, h->D, h->PD, h->PD, 0, cct ? 4 : 2);
        if (h->o_pb >= 0) { h->s_pb = stake(h->DP * 4);   desc(h->o_pb, h->D, h->s_pb, 1, h->DP, 1, 1, h->DP, h->D, 0, 1); }
    }
    h->shadow_bytes = std::max<long long>(sc, 256);
    h->d_pack = nullptr;
    *out = h;
    return V1T_OK;
}

void v1t_vit_destroy(v1t_vit* h) {
    if (!h) return;
    if (h->d_pack) (void)hipFree(h->d_pack);
    for (int i = 0; i < 4; ++i) {
        if (h->dw_ready[i]) (void)hipEventDestroy(h->dw_ready[i]);
        if (h->dw_done[i]) (void)hipEventDestroy(h->dw_done[i]);
    }
    if (h->dw_stream) (void)hipStreamDestroy(h->dw_stream);
    delete h;
}

long long v1t_vit_arena_floats(const v1t_vit* h) { return h->arena_floats; }
long long v1t_vit_param_floats(const v1t_vit* h) { return h->param_floats; }
int v1t_vit_num_tensors(const v1t_vit* h) { return (int)h->tensors.size(); }
int v1t_vit_tensor_info(const v1t_vit* h, int idx, char* name, int name_cap, long long* offset, int* ndim, long long* shape4, int* is_param) {
    if (idx < 0 || idx >= (int)h->tensors.size()) return V1T_ERR_ARG;
    const TensorInfo& t = h->tensors[idx];
    if (name && name_cap > 0) {
        std::strncpy(name, t.name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (offset) *offset = t.off;
    if (ndim) *ndim = t.ndim;
    if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
    if (is_param) *is_param = t.is_param ? 1 : 0;
    return V1T_OK;
}
int v1t_vit_tokens(const v1t_vit* h) { return h->T; }
int v1t_vit_cls_tokens(const v1t_vit* h) { return h->cls; }
int v1t_vit_padded_dim(const v1t_vit* h) { return h->DP; }
int v1t_vit_grid_h(const v1t_vit* h) { return h->gh; }
int v1t_vit_grid_w(const v1t_vit* h) { return h->gw; }
long long v1t_vit_shadow_bytes(const v1t_vit* h) { return h->shadow_bytes; }
long long v1t_vit_workspace_bytes(const v1t_vit* h, int batch, int save) { return ws_layout(h, batch, save != 0).total; }
long long v1t_vit_scratch_bytes(const v1t_vit* h, int batch) { return scratch_layout(h, batch).total; }

long long v1t_vit_workspace_offset(const v1t_vit* h, int batch, int save, const char* name, int block) {
    const WsLayout w = ws_layout(h, batch, save != 0);
    const long long bo = (save ? block : 0) * w.blk_stride;
    const std::string n(name);
    if (n == "x0") return w.x0;
    if (n == "beta") return w.beta + (long long)block * batch * h->DP * 4;
    if (n == "xa") return w.xa + bo;
    if (n == "xm") return w.xm + bo;
    if (n == "xo") return w.xo + bo;
    if (n == "z1") return w.z1 + bo;
    if (n == "qkv") return w.qkv + bo;
    if (n == "o") return w.o + bo;
    if (n == "lse2") return w.lse2 + bo;
    if (n == "z2") return w.z2 + bo;
    if (n == "hpre") return w.hpre + bo;
    if (n == "hact") return w.hact + bo;
    return -1;
}

int v1t_vit_pack(const v1t_vit* h, const float* arena, void* shadow, void* stream) {
    if (!h || !arena || !shadow) return V1T_ERR_ARG;
    if (!h->d_pack && !h->pack.empty()) {  // one-time upload of the pack table (setup path, never inside a graph capture)
        PackDesc* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(PackDesc) * h->pack.size()) != hipSuccess) return V1T_ERR_LAUNCH;
        if (hipMemcpy(d, h->pack.data(), sizeof(PackDesc) * h->pack.size(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d);
            return V1T_ERR_LAUNCH;
        }
        h->d_pack = d;
    }
    return launch_pack(arena, shadow, h->d_pack, (int)h->pack.size(), (hipStream_t)stream);
}

int v1t_vit_forward(const v1t_vit* h, const float* arena, const void* shadow, const float* images, const float* behaviors,
                    int mouse_idx, int B, void* workspace, long long ws_bytes, int save, int training, uint64_t seed,
                    const float* path_scale, float* out, void* stream) {
    if (!h || !arena || !shadow || !images || !workspace || !out || B <= 0) return V1T_ERR_ARG;
    if (h->inject && !behaviors) return V1T_ERR_ARG;
    const WsLayout w = ws_layout(h, B, save != 0);
    if (ws_bytes < w.total) return V1T_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const char* sh = (const char*)shadow;
    const int R = B * h->T, DP = h->DP, D = h->D, HDP = h->HDP, MP = h->MP;
    const bool train = training != 0;
    const int bm = (h->c.behavior_mode == 4) ? mouse_idx : 0;
    if (bm < 0 || bm >= h->nbmlp) return V1T_ERR_ARG;
    // save: 0 = inference (blocks share one workspace region), 1 = everything the backward reads, 2 = inference that keeps every block's
    // qkv and log-sum-exp (attention rollout / probabilities). 0 and 2 are LEAN: the planes only the backward reads - LayerNorm outputs and
    // statistics, gelu' - are not written by the fused LayerNorm + GEMM kernels (1.1 GB per block at batch 256; round 5)
    const int lean = (save != 1 && !train) ? 1 : 0;

    // patch embedding -> x0
    PatchArgs pa{};
    pa.img = images; pa.B = B; pa.C = h->C; pa.IH = h->IH; pa.IW = h->IW; pa.P = h->P; pa.stride = h->S; pa.NH = h->NH; pa.NW = h->NW;
    pa.D = D; pa.DP = DP; pa.W = arena + h->o_pw; pa.bias = arena + h->o_pb; pa.cls = arena + h->o_cls; pa.pos = arena + h->o_pos;
    float* xcur = (h->NB == 0) ? out : (float*)(ws + w.x0);
    pa.x = xcur;
    pa.drop = make_drop(train, h->c.p_dropout, seed, 0xFFFFu);
    if (h->cct) {
        // conv tokenizer (cct.py:46-104): zero-padded unfold -> fp16-operand MFMA GEMM against the conv weight [D][C*P*P] -> ReLU,
        // 3 x 3 / stride 2 max pool, + position table, dropout in one pass (the arg-max of every window is kept for the backward)
        bf16_t* u_hi = (bf16_t*)(ws + w.u_hi);
        bf16_t* u_lo = (bf16_t*)(ws + w.u_lo);
        ConvTokArgs ct{};
        ct.img = images; ct.B = B; ct.C = h->C; ct.IH = h->IH; ct.IW = h->IW; ct.P = h->P; ct.stride = h->S; ct.pad = h->c.conv_pad;
        ct.CH = h->CH; ct.CW = h->CW; ct.NH = h->NH; ct.NW = h->NW; ct.D = D; ct.DP = DP;
        CHECK(launch_conv_unfold(ct, u_hi, u_lo, 0, h->PDX, s));  // hi + bf16 residual planes
        GemmNTArgs g{};
        g.A = u_hi; g.A_lo = u_lo; g.lda = h->PDX; g.B = (const bf16_t*)(sh + h->s_pw); g.B_lo = (const bf16_t*)(sh + h->s_pw_lo); g.ldb = h->PD;
        g.M = B * h->RCI; g.N = DP; g.K = h->PD; g.ldc = DP; g.C = (float*)(ws + w.cconv);
        CHECK(launch_gemm_nt(g, EPI_F32, s));
        ct.conv = (const float*)(ws + w.cconv); ct.pos = h->o_cpos >= 0 ? arena + h->o_cpos : nullptr; ct.x = xcur; ct.idx = (unsigned char*)(ws + w.cidx);
        ct.drop = pa.drop;
        CHECK(launch_cct_pool_fwd(ct, s));
    } else if (h->s_pw >= 0) {  // unfold -> split-bf16 MFMA GEMM with the bias / position / class-token / dropout epilogue
        bf16_t* u_hi = (bf16_t*)(ws + w.u_hi);
        bf16_t* u_lo = (bf16_t*)(ws + w.u_lo);
        const int pm = h->c.patch_mode;
        if (pm >= 2) {  // fp32 patches (mode 2: + 4 diagonal shifts) -> LayerNorm over the patch -> bf16 hi / lo planes
            float* u32 = (float*)(ws + w.u32);
            CHECK(launch_patch_unfold_f32(pa, pm == 2, u32, h->PDX, s));
            LnFwdArgs l{};
            l.x = u32; l.gamma = arena + h->o_pln_w; l.beta = arena + h->o_pln_b; l.z = u_hi; l.z_lo = u_lo; l.lo_f16 = g_fwd_f16;
            l.mean = (float*)(ws + w.pmean1); l.rstd = (float*)(ws + w.prstd1);
            l.rows = R; l.T = h->T; l.D = h->PD; l.DP = h->PDX; l.eps = 1e-5f; l.ones_col = h->PD;
            CHECK(launch_ln_fwd(l, s));
        } else {
            CHECK(launch_patch_unfold(pa, u_hi, u_lo, g_fwd_f16, h->PDX, s));
        }
        GemmNTArgs g{};
        g.A = u_hi; g.A_lo = u_lo; g.lda = h->PDX; g.B = (const bf16_t*)(sh + h->s_pw); g.B_lo = (const bf16_t*)(sh + h->s_pw_lo); g.ldb = h->PD;
        fwd_operands(g);
        g.M = R; g.N = DP; g.K = h->PD; g.ldc = DP;
        g.bias = (const float*)(sh + h->s_pb);
        if (pm == 3) {  // projection output -> LayerNorm(D) -> + pos / class token -> dropout (vit.py:92-100, 122-128)
            float* py = (float*)(ws + w.py);
            g.C = py;
            CHECK(launch_gemm_nt(g, EPI_BIAS_RES, s));  // no residual, no dropout: y = acc + bias
            PatchLn2Args f{};
            f.y = py; f.x0 = xcur; f.mean = (float*)(ws + w.pmean2); f.rstd = (float*)(ws + w.prstd2);
            f.gamma = arena + h->o_pln2_w; f.beta = arena + h->o_pln2_b; f.pos = arena + h->o_pos; f.cls = arena + h->o_cls;
            f.rows = R; f.T = h->T; f.D = D; f.DP = DP; f.eps = 1e-5f; f.drop = pa.drop;
            CHECK(launch_patch_ln2_finish(f, s));
        } else {
            g.C = xcur;
            g.pos = arena + h->o_pos; g.cls = arena + h->o_cls; g.T = h->T; g.n_valid = D;
            g.drop = pa.drop;
            CHECK(launch_gemm_nt(g, EPI_PATCH, s));
        }
    } else {
        CHECK(launch_patch_embed_fwd(pa, s));
    }

    if (h->inject)
        for (int k0 = 0; k0 < h->NB; k0 += BMLP_MAX_BLOCKS) {  // every block's BehaviorMLP in one launch
            BmlpBatch bb{};
            bb.n = std::min(BMLP_MAX_BLOCKS, h->NB - k0);
            for (int i = 0; i < bb.n; ++i) {
                const int k = k0 + i;
                const BmlpOff& bo = h->blk[k].bmlp[bm];
                BmlpArgs& ba = bb.blk[i];
                ba.v = behaviors; ba.B = B; ba.IN = h->IN; ba.J = h->J; ba.D = D; ba.DP = DP;
                ba.W1 = arena + bo.w1; ba.b1 = bo.b1 >= 0 ? arena + bo.b1 : nullptr;
                ba.W3 = arena + bo.w3; ba.b3 = bo.b3 >= 0 ? arena + bo.b3 : nullptr;
                ba.hid = (float*)(ws + w.hid) + (size_t)k * B * h->J;
                ba.out = (float*)(ws + w.beta) + (size_t)k * B * DP;
            }
            CHECK(launch_bmlp_fwd_multi(bb, s));
        }

    const bool x16o = x16_attn_out(h, R), x16a = x16_gelu_out(h, R);  // only the fp16 planes of o / gelu(h) are written
    for (int k = 0; k < h->NB; ++k) {
        const BlockOff& b = h->blk[k];
        char* wb = ws + (save ? k : 0) * w.blk_stride;
        float* xa = h->inject ? (float*)(wb + w.xa) : xcur;
        float* xm = (float*)(wb + w.xm);
        float* xo = (k == h->NB - 1) ? out : (float*)(wb + w.xo);
        bf16_t* z1 = (bf16_t*)(wb + w.z1);
        bf16_t* qkv = (bf16_t*)(wb + w.qkv);
        bf16_t* o = (bf16_t*)(wb + w.o);
        bf16_t* z2 = (bf16_t*)(wb + w.z2);
        bf16_t* hpre = (bf16_t*)(wb + w.hpre);
        bf16_t* hact = (bf16_t*)(wb + w.hact);

        LnFwdArgs l1{};
        l1.x = xcur; l1.inject = h->inject ? (float*)(ws + w.beta) + (size_t)k * B * DP : nullptr; l1.xout = xa;
        l1.gamma = arena + b.ln1w; l1.beta = arena + b.ln1b; l1.z = z1; l1.z_lo = (bf16_t*)(wb + w.z1_lo); l1.lo_f16 = g_fwd_f16;
        l1.mean = (float*)(wb + w.mean1); l1.rstd = (float*)(wb + w.rstd1);
        l1.rows = R; l1.T = h->T; l1.D = D; l1.DP = DP; l1.eps = h->c.ln_eps; l1.ones_col = -1; l1.lean = lean;
        GemmNTArgs g{};
        g.A = z1; g.lda = DP; g.B = (const bf16_t*)(sh + b.s_qkv); g.ldb = DP; g.M = R; g.N = 3 * HDP; g.K = DP; g.C = qkv; g.ldc = 3 * HDP;
        g.A_lo = (const bf16_t*)(wb + w.z1_lo); g.B_lo = (const bf16_t*)(sh + b.s_qkv_lo);
        if (g_nosplit & 1) g.A_lo = g.B_lo = nullptr;
        fwd_operands(g);
        CHECK(ln_then_gemm(l1, g, EPI_BF16, s));

        AttnArgs at{};
        at.qkv = qkv; at.ldqkv = 3 * HDP; at.o = x16o ? nullptr : o; at.ldo = HDP; at.o_lo = (bf16_t*)(wb + w.o_lo); at.lo_f16 = g_fwd_f16; at.lse2 = (float*)(wb + w.lse2);
        at.B = B; at.H = h->H; at.T = h->T; at.scale = arena + b.scale; at.scale_per_head = h->c.use_lsa ? 1 : 0; at.mask_diag = h->c.use_lsa ? 1 : 0;
        at.adrop = make_adrop(train, h->c.t_dropout, seed, 8 * k + 0);
        CHECK(launch_attn_fwd(at, h->HEP, s));

        g = GemmNTArgs{};
        g.A = o; g.lda = HDP; g.B = (const bf16_t*)(sh + b.s_proj); g.ldb = HDP; g.M = R; g.N = DP; g.K = HDP; g.C = xm; g.ldc = DP;
        g.A_lo = (const bf16_t*)(wb + w.o_lo); g.B_lo = (const bf16_t*)(sh + b.s_proj_lo);
        if (g_nosplit & 2) g.A_lo = g.B_lo = nullptr;
        fwd_operands(g);
        g.bias = b.s_projb >= 0 ? (const float*)(sh + b.s_projb) : nullptr; g.res = xa; g.ldres = DP;
        g.drop = make_drop(train, h->c.t_dropout, seed, 8 * k + 1);
        g.row_scale = path_scale ? path_scale + (size_t)(2 * k + 0) * B : nullptr; g.T = h->T;
        CHECK(launch_gemm_nt(g, EPI_BIAS_RES, s));

        LnFwdArgs l2{};
        l2.x = xm; l2.inject = nullptr; l2.xout = nullptr; l2.gamma = arena + b.ln2w; l2.beta = arena + b.ln2b; l2.z = z2; l2.z_lo = (bf16_t*)(wb + w.z2_lo); l2.lo_f16 = g_fwd_f16;
        l2.mean = (float*)(wb + w.mean2); l2.rstd = (float*)(wb + w.rstd2);
        l2.rows = R; l2.T = h->T; l2.D = D; l2.DP = DP; l2.eps = h->c.ln_eps; l2.lean = lean;
        l2.ones_col = (DP > D && h->blk[k].fc1b >= 0) ? DP - 1 : -1;  // d(fc1 bias) comes out of the dW1 GEMM
        g = GemmNTArgs{};
        g.A = z2; g.lda = DP; g.B = (const bf16_t*)(sh + b.s_fc1); g.ldb = DP; g.M = R; g.N = MP; g.K = DP; g.C = hpre; g.ldc = MP;
        g.A_lo = (const bf16_t*)(wb + w.z2_lo); g.B_lo = (const bf16_t*)(sh + b.s_fc1_lo); g.C2_lo = (bf16_t*)(wb + w.hact_lo);
        if (g_nosplit & 4) g.A_lo = g.B_lo = nullptr;
        fwd_operands(g);
        g.C2 = x16a ? nullptr : hact; g.ldc2 = MP; g.bias = b.s_fc1b >= 0 ? (const float*)(sh + b.s_fc1b) : nullptr;
        g.drop = make_drop(train, h->c.t_dropout, seed, 8 * k + 2);
        g.lean = lean;

        GemmNTArgs g2{};
        g2.A = hact; g2.lda = MP; g2.B = (const bf16_t*)(sh + b.s_fc2); g2.ldb = MP; g2.M = R; g2.N = DP; g2.K = MP; g2.C = xo; g2.ldc = DP;
        g2.A_lo = (const bf16_t*)(wb + w.hact_lo); g2.B_lo = (const bf16_t*)(sh + b.s_fc2_lo);
        if (g_nosplit & 8) g2.A_lo = g2.B_lo = nullptr;
        fwd_operands(g2);
        g2.bias = b.s_fc2b >= 0 ? (const float*)(sh + b.s_fc2b) : nullptr; g2.res = xm; g2.ldres = DP;
        g2.drop = make_drop(train, h->c.t_dropout, seed, 8 * k + 3);
        g2.row_scale = path_scale ? path_scale + (size_t)(2 * k + 1) * B : nullptr; g2.T = h->T;
        // the whole MLP branch as one launch where the shape allows (gemm.h, launch_mlp_fwd: fp16 operands, DP = 160, more than 256 row tiles);
        // V1T_MLP_FUSE=0 (dev, A/B): never, 2: at every size
        int rc_mlp = V1T_ERR_UNSUPPORTED;
        if (g_mlp_fuse && g_ln_fuse) rc_mlp = launch_mlp_fwd(l2, g, g2, s);
        if (rc_mlp == V1T_ERR_UNSUPPORTED) {
            CHECK(ln_then_gemm(l2, g, EPI_BIAS_GELU, s));
            CHECK(launch_gemm_nt(g2, EPI_BIAS_RES, s));
        } else {
            CHECK(rc_mlp);
        }
        xcur = xo;
    }
    return V1T_OK;
}

// dz = dY . W (fp32, g.C) followed by the backward of the LayerNorm in front of that linear layer: one kernel where the shape allows
// (gemm.h, launch_gemm_ln_bwd: dz stays in the accumulators), else the GEMM and ln_bwd_kernel. V1T_LNBWD_UNFUSED=1 (dev, A/B): always two.
static const bool g_lnbwd_unfused = dev_env("V1T_LNBWD_UNFUSED") != nullptr;
static const bool g_delta_unfused = dev_env("V1T_DELTA_UNFUSED") != nullptr;  // dev (A/B): attn_delta2_kernel instead of the dO GEMM's row-dot epilogue
static int dx_then_ln_bwd(const GemmNTArgs& g, const LnBwdArgs& lb, hipStream_t s) {
    if (!g_lnbwd_unfused) {
        const int rc = launch_gemm_ln_bwd(g, lb, s);
        if (rc != V1T_ERR_UNSUPPORTED) return rc;
    }
    const int rc = launch_gemm_nt(g, EPI_F32, s);
    return rc != V1T_OK ? rc : launch_ln_bwd(lb, s);
}

int v1t_vit_backward(const v1t_vit* h, const float* arena, const void* shadow, const float* images, const float* behaviors,
                     int mouse_idx, int B, const void* workspace, void* scratch, long long scratch_bytes, int training,
                     uint64_t seed, const float* path_scale, const float* gout, float* grads, void* stream) {
    return v1t_vit_backward_events(h, arena, shadow, images, behaviors, mouse_idx, B, workspace, scratch, scratch_bytes, training, seed, path_scale,
                                   gout, grads, nullptr, stream);
}
int v1t_vit_backward_events(const v1t_vit* h, const float* arena, const void* shadow, const float* images, const float* behaviors,
                            int mouse_idx, int B, const void* workspace, void* scratch, long long scratch_bytes, int training,
                            uint64_t seed, const float* path_scale, const float* gout, float* grads, void* const* block_done,
                            void* stream) {
    return v1t_vit_backward_input(h, arena, shadow, images, behaviors, mouse_idx, B, workspace, scratch, scratch_bytes, training, seed, path_scale,
                                  gout, grads, block_done, nullptr, stream);
}
// fp32 [rows of the unfolded-patch matrix][PD] behind the backward's own scratch: dU of the input gradient
static long long input_grad_bytes(const v1t_vit* h, int B) { return align_up((long long)B * h->RCI * h->PD * 4, 256); }
long long v1t_vit_scratch_bytes_input(const v1t_vit* h, int batch) {
    if (!h || batch <= 0) return 0;
    return scratch_layout(h, batch).total + input_grad_bytes(h, batch);
}
int v1t_vit_backward_input(const v1t_vit* h, const float* arena, const void* shadow, const float* images, const float* behaviors,
                           int mouse_idx, int B, const void* workspace, void* scratch, long long scratch_bytes, int training,
                           uint64_t seed, const float* path_scale, const float* gout, float* grads, void* const* block_done,
                           float* dimages, void* stream) {
    if (!h || !arena || !shadow || !images || !workspace || !scratch || !gout || !grads || B <= 0) return V1T_ERR_ARG;
    const WsLayout w = ws_layout(h, B, true);
    const ScratchLayout sl = scratch_layout(h, B);
    if (scratch_bytes < sl.total + (dimages ? input_grad_bytes(h, B) : 0)) return V1T_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const char* ws = (const char*)workspace;
    char* sc = (char*)scratch;
    const char* sh = (const char*)shadow;
    const int R = B * h->T, DP = h->DP, D = h->D, HDP = h->HDP, MP = h->MP, M = h->M;
    const bool train = training != 0;
    const int bm = (h->c.behavior_mode == 4) ? mouse_idx : 0;
    if (bm < 0 || bm >= h->nbmlp) return V1T_ERR_ARG;

    float* G = (float*)(sc + sl.G);
    // dy: gradient of a block's FC2 output (the MLP branch), dyp: gradient of its projection output (the attention branch). Two buffers, so
    // that the kernel that writes the one (LN2 backward -> dyp, LN1 backward -> dy of the block before) never has to wait for the weight-
    // gradient GEMM that still reads the other on the second stream: with ONE buffer the main stream idled 25-36 us per block in front of
    // the LN2 backward at a rank's share of an 8-GPU step (join of dW2, launched three short kernels earlier)
    // round 5, second-stream mode: dy alternates between two buffers by block parity and dqkv likewise, so that a block's four weight-gradient
    // GEMMs can be handed to the second stream as ONE group (one event each way per block instead of four: every hipEventRecord in the main
    // stream's queue cost ~11 us of dispatch gap at a 14-image share, 16 of them per backward)
    bf16_t* dy_par[2] = {(bf16_t*)(sc + sl.dy), (bf16_t*)(sc + sl.dy3)};
    bf16_t* dyp = (bf16_t*)(sc + sl.dy2);
    bf16_t* dhpre = (bf16_t*)(sc + sl.dhpre);
    float* dz = (float*)(sc + sl.dz);
    bf16_t* dO = (bf16_t*)(sc + sl.dO);
    float* delta = (float*)(sc + sl.delta);
    bf16_t* dqkv_par[2] = {(bf16_t*)(sc + sl.dqkv), (bf16_t*)(sc + (dw_side_for(R) ? sl.dqkv2 : sl.dqkv))};
    float* dbeta = (float*)(sc + sl.dbeta);
    if (h->inject) CHECK(launch_fill_zero(dbeta, (long long)h->NB * B * DP * 4, s));  // (a kernel of the library, not a runtime fill: nothing foreign in the step's trace)
    const TnPlan tp = tn_plan(h, R);
    float* slab = tp.slab ? (float*)(sc + sl.slab) : nullptr;
    const size_t slab_stride = ((tp.slab + 255) / 256 * 256) / sizeof(float);
    // Weight-gradient GEMMs beside the dX GEMMs (second stream): dW = dY^T X and dX = dY W only share their input dY. At full-size launches the
    // two would fight for HBM (a side-stream experiment with the slab reductions alone lost 0.4 %, profiles/r04_gemm_experiments.txt); at a
    // rank's share of a multi-GPU step (23 k rows: ~182 workgroups per launch on 256 CUs, latency-bound) the second stream fills idle CUs.
    // V1T_DW_SIDE=0 / 1 forces it off / on (dev).
    const bool dw_side = slab && dw_side_for(R);
    if (dw_side && !h->dw_stream) {
        // V1T_DW_PRIO=low (dev, A/B): the second stream at the lowest queue priority, so that its workgroups only take CUs the main stream's
        // kernels leave idle
        static const bool dw_low = dev_env("V1T_DW_PRIO") && std::string(dev_env("V1T_DW_PRIO")) == "low";
        int lo = 0, hi = 0;
        if (dw_low && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) {
            if (hipStreamCreateWithPriority(&h->dw_stream, hipStreamNonBlocking, lo) != hipSuccess) return V1T_ERR_LAUNCH;
        } else if (hipStreamCreateWithFlags(&h->dw_stream, hipStreamNonBlocking) != hipSuccess) return V1T_ERR_LAUNCH;
        for (int i = 0; i < 4; ++i)
            if (hipEventCreateWithFlags(&h->dw_ready[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->dw_done[i], hipEventDisableTiming) != hipSuccess) return V1T_ERR_LAUNCH;
    }
    // Second-stream mode: GEMM j of a block (0 dW2, 1 dW1, 2 dWo, 3 dWqkv) is QUEUED by launch_dw and handed over by flush_dw(g) as a group -
    // behind one event recorded on `s` - with the block's other weight-gradient GEMMs; join_dw(g) makes `s` wait for group g (parity
    // events). Group k = {dWqkv of block k + 1, dW2 / dW1 / dWo of block k}, flushed behind block k's LN2 backward and joined at the start of
    // block k - 1: every buffer a group reads stays untouched until then (dy and dqkv alternate by block parity, dyp and dhpre are rewritten
    // only after that join). Single-stream mode (full-size launches): launch_dw launches at once on `s`, the rest are no-ops.
    std::vector<GemmTNArgs> dw_queue;
    auto launch_dw = [&](GemmTNArgs& t, int j) -> int {
        t.slab = slab ? slab + (dw_side ? (size_t)j * slab_stride : 0) : nullptr;
        if (!dw_side) return launch_gemm_tn(t, s);
        dw_queue.push_back(t);
        return V1T_OK;
    };
    bool dw_pending[2] = {false, false};
    auto flush_dw = [&](int g) -> int {
        if (!dw_side || dw_queue.empty()) return V1T_OK;
        const int e = g & 1;
        if (hipEventRecord(h->dw_ready[e], s) != hipSuccess || hipStreamWaitEvent(h->dw_stream, h->dw_ready[e], 0) != hipSuccess) return V1T_ERR_LAUNCH;
        {
            const int rc = launch_gemm_tn_group(dw_queue.data(), (int)dw_queue.size(), h->dw_stream);  // + ONE launch for the group's slab reductions
            if (rc) return rc;
        }
        dw_queue.clear();
        dw_pending[e] = true;
        return hipEventRecord(h->dw_done[e], h->dw_stream) == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
    };
    int join_err = V1T_OK;
    auto join_dw = [&](int g) {
        const int e = g & 1;
        if (dw_side && dw_pending[e]) {
            if (hipStreamWaitEvent(s, h->dw_done[e], 0) != hipSuccess) join_err = V1T_ERR_LAUNCH;
            dw_pending[e] = false;
        }
    };
    // a failed launch in mid-block must not leave second-stream work un-joined behind `s` (the caller may free or reuse the scratch)
    struct DwGuard {
        decltype(join_dw)& join;
        ~DwGuard() { join(0); join(1); }
    } dw_guard{join_dw};
    const bool x16o = x16_attn_out(h, R), x16a = x16_gelu_out(h, R);  // the forward left only the fp16 planes of o / gelu(h)

    const float* gin = gout;
    if (h->NB > 0) {
        // entry of the last block's MLP branch: dy = dropout_bwd(gout) (bf16), db2 += colsum
        const BlockOff& b = h->blk[h->NB - 1];
        CastArgs ca{};
        ca.g = gout; ca.dy = dy_par[(h->NB - 1) & 1]; ca.dbias = b.fc2b >= 0 ? grads + b.fc2b : nullptr;
        ca.drop = make_drop(train, h->c.t_dropout, seed, 8 * (h->NB - 1) + 3);
        ca.scale = path_scale ? path_scale + (size_t)(2 * (h->NB - 1) + 1) * B : nullptr; ca.T = h->T;
        ca.rows = R; ca.D = D; ca.DP = DP;
        CHECK(launch_drop_cast(ca, s));
    }
    for (int k = h->NB - 1; k >= 0; --k) {
        const BlockOff& b = h->blk[k];
        const char* wb = ws + (long long)k * w.blk_stride;
        const float* xa = h->inject ? (const float*)(wb + w.xa) : (k == 0 ? (const float*)(ws + w.x0) : (const float*)(ws + (long long)(k - 1) * w.blk_stride + w.xo));
        const float* xm = (const float*)(wb + w.xm);
        const bf16_t* z1 = (const bf16_t*)(wb + w.z1);
        const bf16_t* qkv = (const bf16_t*)(wb + w.qkv);
        const bf16_t* o = (const bf16_t*)(wb + w.o);
        const bf16_t* z2 = (const bf16_t*)(wb + w.z2);
        const bf16_t* hpre = (const bf16_t*)(wb + w.hpre);
        const bf16_t* hact = (const bf16_t*)(wb + w.hact);
        bf16_t* dy = dy_par[k & 1];          // gradient of this block's FC2 output (written by the block above, or by the entry cast)
        bf16_t* dy_prev = dy_par[(k + 1) & 1];  // ... of block k - 1's, written by this block's LN1 backward
        bf16_t* dqkv = dqkv_par[k & 1];
        join_dw(k + 1);  // group k + 1 (read dhpre, dyp, this parity's dqkv ... of the blocks above): complete before this block rewrites them
        if (block_done && k + 2 < h->NB && block_done[k + 2]) {  // block k + 2's gradients: its own group and, for dWqkv, group k + 1 - both joined
            if (hipEventRecord((hipEvent_t)block_done[k + 2], s) != hipSuccess) return V1T_ERR_LAUNCH;
        }

        // ---- MLP branch: dW2 += dy^T hact
        GemmTNArgs t{};
        t.Y = dy; t.ldy = DP; t.X = x16a ? (const bf16_t*)(wb + w.hact_lo) : hact; t.x_f16 = x16a; t.ldx = MP; t.M = R; t.NY = DP; t.NX = MP; t.dW = grads + b.fc2; t.ldw = M;
        t.yseg_pad = DP; t.yseg_valid = D; t.xseg_pad = MP; t.xseg_valid = M; t.alpha = 1.f;
        t.m_chunk = tp.mc_fc2;
        CHECK(launch_dw(t, 0));  // reads dy, hact
        // d_hpre = (dy . W2) * mask * gelu'(hpre); db1 += colsum
        GemmNTArgs g{};
        g.A = dy; g.lda = DP; g.B = (const bf16_t*)(sh + b.s_fc2_t); g.ldb = DP; g.M = R; g.N = MP; g.K = DP; g.C = dhpre; g.ldc = MP;
        g.aux = hpre; g.ldaux = MP;
        g.colsum = (b.fc1b >= 0 && DP == D) ? grads + b.fc1b : nullptr; g.n_valid = M;  // else: ones column in dW1
        g.drop = make_drop(train, h->c.t_dropout, seed, 8 * k + 2);
        // dz2 = d_hpre . W1, then the LN2 backward: G = gin + dx; dy = dropout_bwd(G) for the projection output; dbo += colsum
        GemmNTArgs gz{};
        gz.A = dhpre; gz.lda = MP; gz.B = (const bf16_t*)(sh + b.s_fc1_t); gz.ldb = MP; gz.M = R; gz.N = DP; gz.K = MP; gz.C = dz; gz.ldc = DP;
        LnBwdArgs lb{};
        lb.dz = dz; lb.x = xm; lb.mean = (const float*)(wb + w.mean2); lb.rstd = (const float*)(wb + w.rstd2); lb.gamma = arena + b.ln2w;
        lb.gin = gin; lb.gout = G; lb.dgamma = grads + b.ln2w; lb.dbeta = grads + b.ln2b; lb.dinject = nullptr;
        lb.dy_next = dyp; lb.dbias_next = b.projb >= 0 ? grads + b.projb : nullptr;
        lb.drop_next = make_drop(train, h->c.t_dropout, seed, 8 * k + 1);
        lb.scale_next = path_scale ? path_scale + (size_t)(2 * k + 0) * B : nullptr;
        lb.B = B; lb.T = h->T; lb.D = D; lb.DP = DP;
        // the dGELU GEMM, the dz GEMM and the LN2 backward as one launch where the shape allows (gemm.h, launch_mlp_bwd: DP = 160, more than 256
        // row tiles, no stochastic depth) - off unless V1T_MLP_BWD_FUSE=1 / 2 (dev: measured neutral)
        const int rc_mb = g_lnbwd_unfused ? V1T_ERR_UNSUPPORTED : launch_mlp_bwd(g, gz, lb, s);
        if (rc_mb == V1T_ERR_UNSUPPORTED) CHECK(launch_gemm_nt(g, EPI_DGELU, s));
        else CHECK(rc_mb);
        // dW1 += d_hpre^T z2
        t = GemmTNArgs{};
        t.Y = dhpre; t.ldy = MP; t.X = z2; t.ldx = DP; t.M = R; t.NY = MP; t.NX = DP; t.dW = grads + b.fc1; t.ldw = D;
        t.yseg_pad = MP; t.yseg_valid = M; t.xseg_pad = DP; t.xseg_valid = D; t.alpha = 1.f;
        if (DP > D && b.fc1b >= 0) { t.dbias = grads + b.fc1b; t.ones_col = DP - 1; }  // z2[:, DP-1] == 1 (LN kernel)
        t.m_chunk = tp.mc_fc1;
        CHECK(launch_dw(t, 1));  // reads dhpre, z2
        if (rc_mb == V1T_ERR_UNSUPPORTED) CHECK(dx_then_ln_bwd(gz, lb, s));
        gin = G;

        // ---- attention branch: dWo += dy^T o
        t = GemmTNArgs{};
        t.Y = dyp; t.ldy = DP; t.X = x16o ? (const bf16_t*)(wb + w.o_lo) : o; t.x_f16 = x16o; t.ldx = HDP; t.M = R; t.NY = DP; t.NX = HDP; t.dW = grads + b.proj; t.ldw = h->HD;
        t.yseg_pad = DP; t.yseg_valid = D; t.xseg_pad = h->HEP; t.xseg_valid = h->HE; t.alpha = 1.f;
        t.m_chunk = tp.mc_proj;
        CHECK(launch_dw(t, 2));  // reads dyp, o
        // group k: dWqkv of block k + 1 (queued there), dW2, dW1, dWo of this block - one event each way. (Round 6, experiment 1: handing the group
        // over BEHIND the dO GEMM, so that it starts beside the MFMA-bound dK/dV kernel instead of beside that HBM-bound GEMM: the dO GEMM
        // 250 -> 142 us, the dK/dV kernel 1.90 -> 2.00 ms, the step 20.78 vs 20.77 ms - the work is conserved. profiles/r06_experiments.txt)
        CHECK(flush_dw(k));
        // dO = dy . Wo
        g = GemmNTArgs{};
        g.A = dyp; g.lda = DP; g.B = (const bf16_t*)(sh + b.s_proj_t); g.ldb = DP; g.M = R; g.N = HDP; g.K = DP; g.C = dO; g.ldc = HDP;
        AttnArgs at{};
        at.qkv = qkv; at.ldqkv = 3 * HDP; at.o = x16o ? (bf16_t*)(wb + w.o_lo) : (bf16_t*)o; at.o_f16 = x16o; at.ldo = HDP; at.lse2 = (float*)(wb + w.lse2);
        at.B = B; at.H = h->H; at.T = h->T; at.scale = arena + b.scale; at.scale_per_head = h->c.use_lsa ? 1 : 0; at.mask_diag = h->c.use_lsa ? 1 : 0;
        at.adrop = make_adrop(train, h->c.t_dropout, seed, 8 * k + 0);
        at.dO = dO; at.lddo = HDP; at.delta = delta; at.dqkv = dqkv; at.lddqkv = 3 * HDP;
        at.dscale = h->c.use_lsa ? grads + b.scale : nullptr;
        if (g_attn_ds && !h->c.use_lsa) { at.ds = (bf16_t*)(sc + sl.ds); at.ldds = attn_ds_ld(h->T); }
        // the row constants of the producer / consumer backward (delta = rowsum(dO * O) per head, -lse) leave the dO GEMM's epilogue where its
        // column tile is one head (RowDotArgs, gemm.h); else the separate pass over dO and O
        bool rowdot = false;
        if (at.ds && h->HEP == 160 && !g_delta_unfused) {
            float* nlse = (float*)(at.ds + attn_ds_elems(B, h->H, h->T));
            g.rd.o = at.o; g.rd.ldo = HDP; g.rd.o_f16 = at.o_f16; g.rd.lse2 = at.lse2; g.rd.nlse = nlse; g.rd.ndelta = nlse + attn_rc_floats(B, h->H, h->T);
            g.rd.T = h->T; g.rd.TPQ = attn_ds_tpq(h->T); g.rd.H = h->H; g.rd.keep = at.adrop.keep_prob;
            rowdot = gemm_nt_takes_row_dot(g, EPI_BF16);
            if (!rowdot) g.rd = RowDotArgs{};
        }
        CHECK(launch_gemm_nt(g, EPI_BF16, s));
        if (!rowdot) CHECK(launch_attn_delta(at, h->HEP, delta, s));
        else if (k == h->NB - 1) CHECK(launch_attn_rc_pad(at, s));  // the pad rows are constants and only this call writes them: once per backward
        CHECK(launch_attn_bwd(at, h->HEP, s));
        // dWqkv += dqkv^T z1
        t = GemmTNArgs{};
        t.Y = dqkv; t.ldy = 3 * HDP; t.X = z1; t.ldx = DP; t.M = R; t.NY = 3 * HDP; t.NX = DP; t.dW = grads + b.qkv; t.ldw = D;
        t.yseg_pad = h->HEP; t.yseg_valid = h->HE; t.xseg_pad = DP; t.xseg_valid = D; t.alpha = 1.f;
        t.m_chunk = tp.mc_qkv;
        CHECK(launch_dw(t, 3));  // reads dqkv, z1: queued for the next block's group (the last block's: flushed behind the loop)
        // dz1 = dqkv . Wqkv
        g = GemmNTArgs{};
        g.A = dqkv; g.lda = 3 * HDP; g.B = (const bf16_t*)(sh + b.s_qkv_t); g.ldb = 3 * HDP; g.M = R; g.N = DP; g.K = 3 * HDP; g.C = dz; g.ldc = DP;
        // LN1 backward: G = G + dx; d(beta_k) = token sums; dy for block k-1's FC2 output
        lb = LnBwdArgs{};
        lb.dz = dz; lb.x = xa; lb.mean = (const float*)(wb + w.mean1); lb.rstd = (const float*)(wb + w.rstd1); lb.gamma = arena + b.ln1w;
        lb.gin = G; lb.gout = G; lb.dgamma = grads + b.ln1w; lb.dbeta = grads + b.ln1b;
        lb.dinject = h->inject ? dbeta + (size_t)k * B * DP : nullptr;
        if (k > 0) {
            lb.dy_next = dy_prev;
            lb.dbias_next = h->blk[k - 1].fc2b >= 0 ? grads + h->blk[k - 1].fc2b : nullptr;
            lb.drop_next = make_drop(train, h->c.t_dropout, seed, 8 * (k - 1) + 3);
            lb.scale_next = path_scale ? path_scale + (size_t)(2 * (k - 1) + 1) * B : nullptr;
        }
        lb.B = B; lb.T = h->T; lb.D = D; lb.DP = DP;
        CHECK(dx_then_ln_bwd(g, lb, s));
        // every gradient of block k's attention / MLP parameters is now enqueued (its BehaviorMLP's follow at the end); single-stream mode:
        // they are complete on `s` here. Second-stream mode: the block's completion event is recorded once its groups are joined (above)
        if (!dw_side && block_done && block_done[k]) {
            if (hipEventRecord((hipEvent_t)block_done[k], s) != hipSuccess) return V1T_ERR_LAUNCH;
        }
    }
    // ---- BehaviorMLP backward (all blocks in one launch): needs the injection gradients, complete behind block 0's LN1 backward
    auto bmlp_backward = [&](hipStream_t st) -> int {
        if (!h->inject) return V1T_OK;
        for (int k0 = 0; k0 < h->NB; k0 += BMLP_MAX_BLOCKS) {
            BmlpBatch bb{};
            bb.n = std::min(BMLP_MAX_BLOCKS, h->NB - k0);
            for (int i = 0; i < bb.n; ++i) {
                const int k = k0 + i;
                const BmlpOff& bo = h->blk[k].bmlp[bm];
                BmlpArgs& ba = bb.blk[i];
                ba.v = behaviors; ba.B = B; ba.IN = h->IN; ba.J = h->J; ba.D = D; ba.DP = DP;
                ba.W1 = arena + bo.w1; ba.W3 = arena + bo.w3;
                ba.hid = (float*)(ws + w.hid) + (size_t)k * B * h->J;
                ba.out = (float*)(ws + w.beta) + (size_t)k * B * DP;
                ba.dout = dbeta + (size_t)k * B * DP;
                ba.dW1 = grads + bo.w1; ba.db1 = bo.b1 >= 0 ? grads + bo.b1 : nullptr;
                ba.dW3 = grads + bo.w3; ba.db3 = bo.b3 >= 0 ? grads + bo.b3 : nullptr;
            }
            const int rc = launch_bmlp_bwd(bb, st);
            if (rc) return rc;
        }
        return V1T_OK;
    };
    // Second-stream mode: what is left in the queue - dWqkv of block 0 - goes out as a last group ("group -1": parity 1, whose events group 1
    // released at the start of block 0) TOGETHER with the BehaviorMLP backward, and both run beside the patch-embedding backward below instead
    // of in front of / behind it (68 us of main-stream idle + 23 us at a 14-image share). Group 0 is joined first: the patch GEMM reuses slab
    // region 0.
    const bool tail_side = dw_side && !dw_queue.empty();
    if (tail_side) {
        const int e = 1;
        if (hipEventRecord(h->dw_ready[e], s) != hipSuccess || hipStreamWaitEvent(h->dw_stream, h->dw_ready[e], 0) != hipSuccess) return V1T_ERR_LAUNCH;
        CHECK(launch_gemm_tn_group(dw_queue.data(), (int)dw_queue.size(), h->dw_stream));
        dw_queue.clear();
        CHECK(bmlp_backward(h->dw_stream));
        dw_pending[e] = true;
        if (hipEventRecord(h->dw_done[e], h->dw_stream) != hipSuccess) return V1T_ERR_LAUNCH;
    }
    join_dw(0);
    if (join_err) return join_err;
    // block 1's gradients are complete here (its dWqkv rode in group 0, its dW2 / dW1 / dWo in group 1, joined at the start of block 0): its
    // exchange can start beside the patch-embedding backward; only block 0 has to wait for the tail group
    if (dw_side && block_done && h->NB > 1 && block_done[1] && hipEventRecord((hipEvent_t)block_done[1], s) != hipSuccess) return V1T_ERR_LAUNCH;
    // ---- patch embedding backward (gin = grad wrt x0)
    PatchArgs pa{};
    pa.img = images; pa.B = B; pa.C = h->C; pa.IH = h->IH; pa.IW = h->IW; pa.P = h->P; pa.stride = h->S; pa.NH = h->NH; pa.NW = h->NW;
    pa.D = D; pa.DP = DP; pa.x = (float*)gin;
    pa.drop = make_drop(train, h->c.p_dropout, seed, 0xFFFFu);
    pa.dW = grads + h->o_pw; pa.dbias = grads + h->o_pb; pa.dcls = grads + h->o_cls; pa.dpos = grads + h->o_pos;
    if (h->cct) {
        // conv tokenizer backward: route d x0 through the dropout mask, the saved arg-max of every pool window and the ReLU into the
        // conv output grid (bf16), re-unfold the images and dW += d conv^T . U  (no input gradient: the cropper samples nearest)
        bf16_t* gd = (bf16_t*)(sc + sl.pgd);
        bf16_t* u = (bf16_t*)(sc + sl.pu);
        ConvTokArgs ct{};
        ct.img = images; ct.B = B; ct.C = h->C; ct.IH = h->IH; ct.IW = h->IW; ct.P = h->P; ct.stride = h->S; ct.pad = h->c.conv_pad;
        ct.CH = h->CH; ct.CW = h->CW; ct.NH = h->NH; ct.NW = h->NW; ct.D = D; ct.DP = DP;
        ct.x = (float*)gin; ct.idx = (unsigned char*)(ws + w.cidx); ct.drop = pa.drop;
        CHECK(launch_cct_pool_bwd(ct, gd, s));
        CHECK(launch_conv_unfold(ct, u, nullptr, 0, h->PDX, s));
        GemmTNArgs t{};
        const int RU = B * h->RCI;
        t.Y = gd; t.ldy = DP; t.X = u; t.ldx = h->PDX; t.M = RU; t.NY = DP; t.NX = h->PDX; t.dW = grads + h->o_pw; t.ldw = h->PD;
        t.yseg_pad = DP; t.yseg_valid = D; t.xseg_pad = h->PDX; t.xseg_valid = h->PD; t.alpha = 1.f;
        t.m_chunk = tp.mc_patch; t.slab = slab;
        CHECK(launch_gemm_tn(t, s));
    } else if (h->s_pw >= 0 && h->c.patch_mode >= 2) {
        // modes 2 / 3: [LayerNorm(D) backward] -> dW (+ dbias) against the normalised patches of the forward -> dU -> LayerNorm(patch) parameter gradients
        const char* ws = (const char*)workspace;
        bf16_t* gd = (bf16_t*)(sc + sl.pgd);
        const bool ln2 = h->c.patch_mode == 3;
        float* gdf = ln2 ? (float*)(sc + sl.dz) : nullptr;  // fp32 scratch, free at this point
        CHECK(launch_patch_bwd_pos_cast_nocls(pa, gd, gdf, s));
        if (ln2) {
            LnBwdArgs lb{};
            lb.dz = gdf; lb.x = (const float*)(ws + w.py); lb.mean = (const float*)(ws + w.pmean2); lb.rstd = (const float*)(ws + w.prstd2);
            lb.gamma = arena + h->o_pln2_w; lb.gin = nullptr; lb.gout = (float*)(sc + sl.G);
            lb.dgamma = grads + h->o_pln2_w; lb.dbeta = grads + h->o_pln2_b; lb.dy_next = gd;
            lb.B = B; lb.T = h->T; lb.D = D; lb.DP = DP;
            CHECK(launch_ln_bwd(lb, s));
        }
        GemmTNArgs t{};
        t.Y = gd; t.ldy = DP; t.X = (const bf16_t*)(ws + w.u_hi); t.ldx = h->PDX; t.M = R; t.NY = DP; t.NX = h->PDX; t.dW = grads + h->o_pw; t.ldw = h->PD;
        t.yseg_pad = DP; t.yseg_valid = D; t.xseg_pad = h->PDX; t.xseg_valid = h->PD; t.alpha = 1.f;
        t.dbias = grads + h->o_pb; t.ones_col = h->PD;
        t.m_chunk = tp.mc_patch; t.slab = slab;
        CHECK(launch_gemm_tn(t, s));
        float* du = (float*)(sc + sl.pdu);
        GemmNTArgs g{};
        g.A = gd; g.lda = DP; g.B = (const bf16_t*)(sh + h->s_pw_t); g.ldb = DP; g.M = R; g.N = h->PD; g.K = DP; g.C = du; g.ldc = h->PD;
        CHECK(launch_gemm_nt(g, EPI_F32, s));
        CHECK(launch_ln_param_grad(du, h->PD, (const float*)(ws + w.u32), h->PDX, (const float*)(ws + w.pmean1), (const float*)(ws + w.prstd1), R, h->PD,
                                   grads + h->o_pln_w, grads + h->o_pln_b, s));
    } else if (h->s_pw >= 0) {  // dpos / dcls + bf16 gradient, then dW (+ dbias through the ones column) against the unfolded patches the forward left
        // in the workspace (its bf16 plane; round 2 unfolded the images a second time here: 65 us per step)
        bf16_t* gd = (bf16_t*)(sc + sl.pgd);
        const bf16_t* u = (const bf16_t*)((const char*)workspace + w.u_hi);
        CHECK(launch_patch_bwd_pos_cast(pa, gd, s));
        GemmTNArgs t{};
        t.Y = gd; t.ldy = DP; t.X = u; t.ldx = h->PDX; t.M = R; t.NY = DP; t.NX = h->PDX; t.dW = grads + h->o_pw; t.ldw = h->PD;
        t.yseg_pad = DP; t.yseg_valid = D; t.xseg_pad = h->PDX; t.xseg_valid = h->PD; t.alpha = 1.f;
        t.dbias = grads + h->o_pb; t.ones_col = h->PD;
        t.m_chunk = tp.mc_patch; t.slab = slab;
        CHECK(launch_gemm_tn(t, s));
    } else {
        CHECK(launch_patch_embed_bwd(pa, s));
    }
    if (!tail_side) CHECK(bmlp_backward(s));
    if (dimages) {
        // gradient with respect to the core input (vit.py:66-72, 122-129 under autograd; cct.py:30-104): dU = d x0 . W in fp32 against the fp32
        // master weight, [the LayerNorm over the patch], col2im. An analysis path (MEIs, saliency): training never asks for it.
        float* dU = (float*)(sc + sl.total);
        const float* Wp = arena + h->o_pw;  // [D][PD] (mode 1 / CCT: the conv weight [D][C][P][P], the same layout)
        if (h->cct) {
            const long long RU = (long long)B * h->RCI;
            CHECK(launch_patch_du(nullptr, (const bf16_t*)(sc + sl.pgd), DP, pa.drop, 0, 0, Wp, D, h->PD, RU, dU, s));
            CHECK(launch_patch_col2im(dU, h->PD, B, h->C, h->IH, h->IW, h->P, h->S, h->c.conv_pad, h->CH, h->CW, h->RCI, 0, 0, dimages, s));
        } else if (h->c.patch_mode >= 2) {
            const char* wsb = (const char*)workspace;
            CHECK(launch_patch_ln_bwd_rows((const float*)(sc + sl.pdu), h->PD, (const float*)(wsb + w.u32), h->PDX, (const float*)(wsb + w.pmean1),
                                           (const float*)(wsb + w.prstd1), arena + h->o_pln_w, R, h->PD, dU, s));
            CHECK(launch_patch_col2im(dU, h->PD, B, h->C, h->IH, h->IW, h->P, h->S, 0, h->NH, h->NW, h->T, 1, h->c.patch_mode == 2, dimages, s));
        } else {
            CHECK(launch_patch_du((const float*)gin, nullptr, DP, pa.drop, h->T, 1, Wp, D, h->PD, R, dU, s));
            CHECK(launch_patch_col2im(dU, h->PD, B, h->C, h->IH, h->IW, h->P, h->S, 0, h->NH, h->NW, h->T, 1, 0, dimages, s));
        }
    }
    join_dw(1);  // everything of the second stream is behind `s` from here on
    if (join_err) return join_err;
    if (dw_side && block_done && block_done[0] && hipEventRecord((hipEvent_t)block_done[0], s) != hipSuccess)  // block 0: its dWqkv is in the tail group
        return V1T_ERR_LAUNCH;
    return V1T_OK;
}

int v1t_dropout_mask(uint64_t seed, uint32_t stream_id, float p, long long rows, long long cols, uint8_t* out, void* stream) {
    if (!out || rows <= 0 || cols <= 0) return V1T_ERR_ARG;
    if (stream_id != 0xFFFFu && stream_id % 8 == 0)  // attention-P stream: rows = B*H*T, cols = T
        return launch_attn_dropout_mask(out, rows, cols, make_adrop(true, p, seed, stream_id), (hipStream_t)stream);
    return launch_dropout_mask(out, rows, cols, make_drop(true, p, seed, stream_id), (hipStream_t)stream);
}

int v1t_vit_backward_second_stream(const v1t_vit* h, int batch) {
    if (!h || batch <= 0) return 0;
    const long long R = (long long)batch * h->T;
    return (tn_plan(h, R).slab && dw_side_for(R)) ? 1 : 0;
}

float v1t_attention_dropout_rate(float p) {
    const AttnDrop d = make_adrop(p > 0.f, p, 0, 0);
    return (float)((double)d.thresh16 / 65536.0);
}

int v1t_gaussian2d_forward(const float* z, long long zsb, long long zsc, int B, int C, int H, int W, int N, const float* grid,
                           const float* feat, int FS, const float* bias, float* out, void* stream) {
    if (!z || !grid || !feat || !out) return V1T_ERR_ARG;
    ReadoutArgs a{};
    a.z = z; a.zsb = zsb; a.zsc = zsc; a.B = B; a.C = C; a.H = H; a.W = W; a.N = N; a.grid = grid; a.feat = feat; a.FS = FS; a.bias = bias; a.out = out;
    return launch_readout_fwd(a, (hipStream_t)stream);
}

int v1t_gaussian2d_backward_ws(const float* z, long long zsb, long long zsc, int B, int C, int H, int W, int N, const float* grid,
                               const float* feat, int FS, const float* gout, float* dz, long long dzsb, long long dzsc,
                               float* dgrid, float* dfeat, float* dbias, void* ws, long long ws_bytes, void* stream) {
    if (!z || !grid || !feat || !gout || ws_bytes < 0) return V1T_ERR_ARG;
    ReadoutArgs a{};
    a.z = z; a.zsb = zsb; a.zsc = zsc; a.B = B; a.C = C; a.H = H; a.W = W; a.N = N; a.grid = grid; a.feat = feat; a.FS = FS;
    a.gout = gout; a.dz = dz; a.dzsb = dzsb; a.dzsc = dzsc; a.dgrid = dgrid; a.dfeat = dfeat; a.dbias = dbias;
    return launch_readout_bwd(a, ws, (size_t)ws_bytes, (hipStream_t)stream);
}
int v1t_gaussian2d_backward_parts(const float* z, long long zsb, long long zsc, int B, int C, int H, int W, int N, const float* grid,
                                  const float* feat, int FS, const float* gout, float* dz, long long dzsb, long long dzsc,
                                  float* dgrid, float* dfeat, float* dbias, void* ws, long long ws_bytes, int parts, void* stream) {
    if (!grid || !ws || ws_bytes <= 0 || parts <= 0 || parts > READOUT_BWD_ALL) return V1T_ERR_ARG;
    if ((parts & (READOUT_BWD_PARAMS | READOUT_BWD_DZ)) && (!z || !feat || !gout)) return V1T_ERR_ARG;
    ReadoutArgs a{};
    a.z = z; a.zsb = zsb; a.zsc = zsc; a.B = B; a.C = C; a.H = H; a.W = W; a.N = N; a.grid = grid; a.feat = feat; a.FS = FS;
    a.gout = gout; a.dz = dz; a.dzsb = dzsb; a.dzsc = dzsc; a.dgrid = dgrid; a.dfeat = dfeat; a.dbias = dbias;
    return launch_readout_bwd_parts(a, ws, (size_t)ws_bytes, parts, (hipStream_t)stream);
}
int v1t_gaussian2d_backward(const float* z, long long zsb, long long zsc, int B, int C, int H, int W, int N, const float* grid,
                            const float* feat, int FS, const float* gout, float* dz, long long dzsb, long long dzsc,
                            float* dgrid, float* dfeat, float* dbias, void* stream) {
    return v1t_gaussian2d_backward_ws(z, zsb, zsc, B, C, H, W, N, grid, feat, FS, gout, dz, dzsb, dzsc, dgrid, dfeat, dbias, nullptr, 0, stream);
}
long long v1t_gaussian2d_backward_ws_bytes(int B, int H, int W, int N) { return (long long)readout_bwd_ws_bytes(B, H, W, N); }

int v1t_crop_nearest(const float* in, int B, int C, int IH, int IW, const float* grid, const float* shifts, float* out, int OH, int OW,
                     void* stream) {
    if (!in || !grid || !out || B < 0 || C <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return V1T_ERR_ARG;
    return launch_crop_nearest(in, B, C, IH, IW, grid, shifts, out, OH, OW, (hipStream_t)stream);
}
int v1t_resize_bilinear(const float* in, int planes, int IH, int IW, float* out, int OH, int OW, void* stream) {
    if (!in || !out || planes < 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return V1T_ERR_ARG;
    return launch_resize_bilinear(in, out, planes, IH, IW, OH, OW, (hipStream_t)stream);
}
int v1t_resize_bilinear_backward(const float* dout, int planes, int IH, int IW, float* din, int OH, int OW, void* stream) {
    if (!dout || !din || planes < 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return V1T_ERR_ARG;
    return launch_resize_bilinear_bwd(dout, din, planes, IH, IW, OH, OW, (hipStream_t)stream);
}
int v1t_readout_grid_forward(int B, int N, int gd, const float* src, const float* W0, const float* b0, const float* W2, const float* b2,
                             const float* mu_free, const float* sigma, const float* eps, const float* shift, float* grid, void* stream) {
    if (!sigma || !grid || (gd > 0 && (!src || !W0 || !b0 || !W2 || !b2)) || (gd == 0 && !mu_free)) return V1T_ERR_ARG;
    GridArgs a{};
    a.B = B; a.N = N; a.gd = gd; a.src = src; a.W0 = W0; a.b0 = b0; a.W2 = W2; a.b2 = b2; a.mu_free = mu_free; a.sigma = sigma;
    a.eps = eps; a.shift = shift; a.grid = grid;
    return launch_grid_fwd(a, (hipStream_t)stream);
}
int v1t_readout_grid_backward(int B, int N, int gd, const float* src, const float* W0, const float* b0, const float* W2, const float* b2,
                              const float* mu_free, const float* sigma, const float* eps, const float* dgrid, float* dW0, float* db0,
                              float* dW2, float* db2, float* dmu_free, float* dsigma, float* dshift, void* stream) {
    return v1t_readout_grid_backward_ws(B, N, gd, src, W0, b0, W2, b2, mu_free, sigma, eps, dgrid, dW0, db0, dW2, db2, dmu_free, dsigma, dshift,
                                        nullptr, 0, stream);
}
long long v1t_readout_grid_backward_ws_bytes(int B, int N) { return (long long)grid_bwd_ws_bytes(B, N); }
int v1t_readout_grid_backward_ws(int B, int N, int gd, const float* src, const float* W0, const float* b0, const float* W2, const float* b2,
                                 const float* mu_free, const float* sigma, const float* eps, const float* dgrid, float* dW0, float* db0,
                                 float* dW2, float* db2, float* dmu_free, float* dsigma, float* dshift, void* ws, long long ws_bytes,
                                 void* stream) {
    if (!sigma || !dgrid || (gd > 0 && (!src || !W0 || !b0 || !W2 || !b2 || !dW0 || !db0 || !dW2 || !db2)) || (gd == 0 && !mu_free)) return V1T_ERR_ARG;
    GridArgs a{};
    a.B = B; a.N = N; a.gd = gd; a.src = src; a.W0 = W0; a.b0 = b0; a.W2 = W2; a.b2 = b2; a.mu_free = mu_free; a.sigma = sigma;
    a.eps = eps; a.dgrid = dgrid; a.dW0 = dW0; a.db0 = db0; a.dW2 = dW2; a.db2 = db2; a.dmu_free = dmu_free; a.dsigma = dsigma; a.dshift = dshift;
    return launch_grid_bwd(a, ws, (size_t)(ws_bytes < 0 ? 0 : ws_bytes), (hipStream_t)stream);
}
int v1t_normal_fill(float* out, long long n, uint64_t seed, uint32_t stream_id, void* stream) {
    if (!out || n < 0) return V1T_ERR_ARG;
    return launch_normal_fill(out, n, seed, stream_id, (hipStream_t)stream);
}
int v1t_concat2(const float* a, int na, const float* b, int nb, int rows, float* out, int ldo, void* stream) {
    if (!out || rows < 0 || na < 0 || nb < 0 || (na && !a) || (nb && !b) || ldo < na + nb) return V1T_ERR_ARG;
    return launch_concat2(a, na, b, nb, rows, out, ldo, (hipStream_t)stream);
}
int v1t_core_shifter_forward(int B, const float* pupil, const float* W0, const float* b0, const float* W2, const float* b2, const float* W4,
                             const float* b4, float* shift, void* stream) {
    if (!pupil || !W0 || !b0 || !W2 || !b2 || !W4 || !b4 || !shift) return V1T_ERR_ARG;
    ShifterArgs a{};
    a.B = B; a.pupil = pupil; a.W0 = W0; a.b0 = b0; a.W2 = W2; a.b2 = b2; a.W4 = W4; a.b4 = b4; a.shift = shift;
    return launch_shifter_fwd(a, (hipStream_t)stream);
}
int v1t_core_shifter_backward(int B, const float* pupil, const float* W0, const float* b0, const float* W2, const float* b2, const float* W4,
                              const float* b4, const float* dshift, float* dW0, float* db0, float* dW2, float* db2, float* dW4, float* db4,
                              void* stream) {
    if (!pupil || !W0 || !b0 || !W2 || !b2 || !W4 || !b4 || !dshift || !dW0 || !db0 || !dW2 || !db2 || !dW4 || !db4) return V1T_ERR_ARG;
    ShifterArgs a{};
    a.B = B; a.pupil = pupil; a.W0 = W0; a.b0 = b0; a.W2 = W2; a.b2 = b2; a.W4 = W4; a.b4 = b4; a.dshift = dshift;
    a.dW0 = dW0; a.db0 = db0; a.dW2 = dW2; a.db2 = db2; a.dW4 = dW4; a.db4 = db4;
    return launch_shifter_bwd(a, (hipStream_t)stream);
}

int v1t_elu1_poisson(const float* u, const float* y, long long n, float loss_scale, float gscale, float* yhat, float* du,
                     float* loss, void* stream) {
    if (!u || n < 0) return V1T_ERR_ARG;
    LossArgs a{};
    a.u = u; a.y = y; a.yhat = yhat; a.du = du; a.loss = loss; a.n = n; a.loss_scale = loss_scale; a.gscale = gscale;
    return launch_elu1_poisson(a, (hipStream_t)stream);
}

int v1t_poisson_loss(const float* y_pred, const float* y_true, long long n, float eps, float loss_scale, float* dy, float* loss, void* stream) {
    if (!y_pred || !y_true || !loss || n < 0) return V1T_ERR_ARG;
    return launch_poisson_loss(y_pred, y_true, n, eps, loss_scale, dy, loss, (hipStream_t)stream);
}
int v1t_elu1_backward(const float* u, const float* y, const float* g, long long n, float* du, void* stream) {
    if (!u || !y || !g || !du || n < 0) return V1T_ERR_ARG;
    return launch_elu1_bwd(u, y, g, n, du, (hipStream_t)stream);
}

int v1t_adamw_step(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float l1, int zero_grad, void* stream) {
    if (!p || !g || !m || !v || step < 1) return V1T_ERR_ARG;
    AdamArgs a{};
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bc1 = (float)(1.0 - std::pow((double)beta1, step));
    a.bc2 = (float)(1.0 - std::pow((double)beta2, step));
    a.l1 = l1; a.zero_grad = zero_grad;
    return launch_adamw(a, (hipStream_t)stream);
}
int v1t_l1_sum(const float* p, long long n, float scale, float* out, void* stream) { return launch_l1_sum(p, n, scale, out, (hipStream_t)stream); }
int v1t_l1_grad(const float* p, float* g, long long n, float scale, void* stream) { return launch_l1_grad(p, g, n, scale, (hipStream_t)stream); }
int v1t_l1_grad_dev(const float* p, float* g, long long n, float scale, const float* gscale, void* stream) {
    if (!gscale) return V1T_ERR_ARG;
    return launch_l1_grad_dev(p, g, n, scale, gscale, (hipStream_t)stream);
}

int v1t_gemm_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* C, int ldc, int out_f32, void* stream) {
    GemmNTArgs g{};
    g.A = (const bf16_t*)A; g.lda = lda; g.B = (const bf16_t*)B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.C = C; g.ldc = ldc;
    return launch_gemm_nt(g, out_f32 ? EPI_F32 : EPI_BF16, (hipStream_t)stream);
}
int v1t_gemm_tn(const void* Y, int ldy, const void* X, int ldx, int M, int NY, int NX, float* dW, int ldw, int m_chunk, void* stream) {
    GemmTNArgs t{};
    t.Y = (const bf16_t*)Y; t.ldy = ldy; t.X = (const bf16_t*)X; t.ldx = ldx; t.M = M; t.NY = NY; t.NX = NX; t.dW = dW; t.ldw = ldw;
    t.yseg_pad = NY; t.yseg_valid = NY; t.xseg_pad = NX; t.xseg_valid = NX; t.m_chunk = m_chunk; t.alpha = 1.f;
    return launch_gemm_tn(t, (hipStream_t)stream);
}
long long v1t_gemm_tn_slab_bytes(int M, int NY, int NX, int m_chunk) { return (long long)gemm_tn_slab_bytes(M, NY, NX, m_chunk); }
int v1t_gemm_tn_slab(const void* Y, int ldy, const void* X, int ldx, int M, int NY, int NX, float* dW, int ldw, int m_chunk,
                     float* slab, long long slab_bytes, void* stream) {
    GemmTNArgs t{};
    t.Y = (const bf16_t*)Y; t.ldy = ldy; t.X = (const bf16_t*)X; t.ldx = ldx; t.M = M; t.NY = NY; t.NX = NX; t.dW = dW; t.ldw = ldw;
    t.yseg_pad = NY; t.yseg_valid = NY; t.xseg_pad = NX; t.xseg_valid = NX; t.m_chunk = m_chunk; t.alpha = 1.f;
    const long long need = (long long)gemm_tn_slab_bytes(M, NY, NX, m_chunk);
    if (need > 0 && (!slab || slab_bytes < need)) return V1T_ERR_ARG;
    t.slab = need > 0 ? slab : nullptr;
    return launch_gemm_tn(t, (hipStream_t)stream);
}
int v1t_attention_forward(const void* qkv, int B, int H, int T, int DP, const float* scale, int scale_per_head, int mask_diag,
                          float dropout_p, uint64_t seed, uint32_t stream_id, void* o, float* lse2, void* stream) {
    AttnArgs a{};
    a.qkv = (const bf16_t*)qkv; a.ldqkv = 3 * H * DP; a.o = (bf16_t*)o; a.ldo = H * DP; a.lse2 = lse2; a.B = B; a.H = H; a.T = T;
    a.scale = scale; a.scale_per_head = scale_per_head; a.mask_diag = mask_diag;
    a.adrop = make_adrop(dropout_p > 0.f, dropout_p, seed, stream_id);
    return launch_attn_fwd(a, DP, (hipStream_t)stream);
}
int v1t_attention_backward(const void* qkv, const void* o, const void* dO, const float* lse2, int B, int H, int T, int DP,
                           const float* scale, int scale_per_head, int mask_diag, float dropout_p, uint64_t seed,
                           uint32_t stream_id, float* delta_ws, void* dqkv, float* dscale, void* stream) {
    return v1t_attention_backward_ws(qkv, o, dO, lse2, B, H, T, DP, scale, scale_per_head, mask_diag, dropout_p, seed, stream_id, delta_ws, dqkv,
                                     dscale, nullptr, 0, stream);
}
// The planes the ViT core itself uses: the attention output as ONE fp16 plane (2^-12), which the backward's row constants
// delta = rowsum(dO o O) read - eight times finer than the bf16 plane of v1t_attention_forward / _backward_ws.
int v1t_attention_forward_f16o(const void* qkv, int B, int H, int T, int DP, const float* scale, int scale_per_head, int mask_diag,
                               float dropout_p, uint64_t seed, uint32_t stream_id, void* o_f16, float* lse2, void* stream) {
    if (!qkv || !scale || !o_f16 || !lse2) return V1T_ERR_ARG;
    AttnArgs a{};
    a.qkv = (const bf16_t*)qkv; a.ldqkv = 3 * H * DP; a.o = nullptr; a.o_lo = (bf16_t*)o_f16; a.lo_f16 = 1; a.ldo = H * DP; a.lse2 = lse2; a.B = B; a.H = H; a.T = T;
    a.scale = scale; a.scale_per_head = scale_per_head; a.mask_diag = mask_diag;
    a.adrop = make_adrop(dropout_p > 0.f, dropout_p, seed, stream_id);
    return launch_attn_fwd(a, DP, (hipStream_t)stream);
}
int v1t_attention_backward_ws_f16o(const void* qkv, const void* o_f16, const void* dO, const float* lse2, int B, int H, int T, int DP,
                                   const float* scale, int scale_per_head, int mask_diag, float dropout_p, uint64_t seed,
                                   uint32_t stream_id, float* delta_ws, void* dqkv, float* dscale, void* ds_ws, long long ds_bytes,
                                   void* stream) {
    if (ds_ws && ds_bytes < (long long)attn_ds_bytes(B, H, T)) return V1T_ERR_WORKSPACE;
    AttnArgs a{};
    if (ds_ws) { a.ds = (bf16_t*)ds_ws; a.ldds = attn_ds_ld(T); }
    a.qkv = (const bf16_t*)qkv; a.ldqkv = 3 * H * DP; a.o = (bf16_t*)o_f16; a.o_f16 = 1; a.ldo = H * DP; a.lse2 = (float*)lse2; a.B = B; a.H = H; a.T = T;
    a.scale = scale; a.scale_per_head = scale_per_head; a.mask_diag = mask_diag;
    a.adrop = make_adrop(dropout_p > 0.f, dropout_p, seed, stream_id);
    a.dO = (const bf16_t*)dO; a.lddo = H * DP; a.delta = delta_ws; a.dqkv = (bf16_t*)dqkv; a.lddqkv = 3 * H * DP; a.dscale = dscale;
    CHECK(launch_attn_delta(a, DP, delta_ws, (hipStream_t)stream));
    return launch_attn_bwd(a, DP, (hipStream_t)stream);
}
long long v1t_attention_backward_ws_bytes(int B, int H, int T) { return (long long)attn_ds_bytes(B, H, T); }
int v1t_attention_backward_ws(const void* qkv, const void* o, const void* dO, const float* lse2, int B, int H, int T, int DP,
                              const float* scale, int scale_per_head, int mask_diag, float dropout_p, uint64_t seed,
                              uint32_t stream_id, float* delta_ws, void* dqkv, float* dscale, void* ds_ws, long long ds_bytes,
                              void* stream) {
    if (ds_ws && ds_bytes < (long long)attn_ds_bytes(B, H, T)) return V1T_ERR_WORKSPACE;
    AttnArgs a{};
    if (ds_ws) { a.ds = (bf16_t*)ds_ws; a.ldds = attn_ds_ld(T); }
    a.qkv = (const bf16_t*)qkv; a.ldqkv = 3 * H * DP; a.o = (bf16_t*)o; a.ldo = H * DP; a.lse2 = (float*)lse2; a.B = B; a.H = H; a.T = T;
    a.scale = scale; a.scale_per_head = scale_per_head; a.mask_diag = mask_diag;
    a.adrop = make_adrop(dropout_p > 0.f, dropout_p, seed, stream_id);
    a.dO = (const bf16_t*)dO; a.lddo = H * DP; a.delta = delta_ws; a.dqkv = (bf16_t*)dqkv; a.lddqkv = 3 * H * DP; a.dscale = dscale;
    CHECK(launch_attn_delta(a, DP, delta_ws, (hipStream_t)stream));
    return launch_attn_bwd(a, DP, (hipStream_t)stream);
}

int v1t_rollout_headmax(const void* qkv, const float* lse2, int B, int H, int T, int DP, const float* scale, int scale_per_head,
                        int mask_diag, float* A, int TP, float* rowsum, void* stream) {
    if (!qkv || !lse2 || !scale || !A || !rowsum || TP < T || TP % 4) return V1T_ERR_ARG;
    AttnArgs a{};
    a.qkv = (const bf16_t*)qkv; a.ldqkv = 3 * H * DP; a.lse2 = (float*)lse2; a.B = B; a.H = H; a.T = T;
    a.scale = scale; a.scale_per_head = scale_per_head; a.mask_diag = mask_diag;
    return launch_rollout_headmax(a, DP, A, TP, rowsum, 0, (hipStream_t)stream);
}
int v1t_rollout_headmax_rows(const void* qkv, const float* lse2, int B, int H, int T, int DP, const float* scale, int scale_per_head,
                             int mask_diag, float* A, int TP, float* rowsum, int q_rows, void* stream) {
    if (!qkv || !lse2 || !scale || !A || !rowsum || TP < T || TP % 4 || q_rows <= 0) return V1T_ERR_ARG;
    AttnArgs a{};
    a.qkv = (const bf16_t*)qkv; a.ldqkv = 3 * H * DP; a.lse2 = (float*)lse2; a.B = B; a.H = H; a.T = T;
    a.scale = scale; a.scale_per_head = scale_per_head; a.mask_diag = mask_diag;
    return launch_rollout_headmax(a, DP, A, TP, rowsum, q_rows, (hipStream_t)stream);
}
int v1t_attention_probs(const void* qkv, const float* lse2, int B, int H, int T, int DP, const float* scale, int scale_per_head,
                        int mask_diag, float* P, int TP, void* stream) {
    if (!qkv || !lse2 || !scale || !P || TP < T || TP % 4) return V1T_ERR_ARG;
    AttnArgs a{};
    a.qkv = (const bf16_t*)qkv; a.ldqkv = 3 * H * DP; a.lse2 = (float*)lse2; a.B = B; a.H = H; a.T = T;
    a.scale = scale; a.scale_per_head = scale_per_head; a.mask_diag = mask_diag;
    return launch_rollout_headmax(a, DP, P, TP, nullptr, 0, (hipStream_t)stream);
}
int v1t_rollout_matmul(const float* A, const float* rowsum, const float* Xin, float* Xout, int B, int T, int TP, void* stream) {
    if (!A || !rowsum || !Xout || Xout == Xin || B <= 0 || T <= 0) return V1T_ERR_ARG;
    return launch_rollout_matmul(A, rowsum, Xin, Xout, B, T, TP, (hipStream_t)stream);
}
int v1t_rollout_vecmat(const float* A, const float* rowsum, const float* v, float* u, int B, int T, int TP, void* stream) {
    if (!A || !rowsum || !u || TP < T || TP % 4) return V1T_ERR_ARG;
    return launch_rollout_vecmat(A, rowsum, v, u, B, T, TP, (hipStream_t)stream);
}

int v1t_layernorm_forward(const float* x, const float* inject, float* xout, const float* gamma, const float* beta, void* z,
                          float* mean, float* rstd, int B, int T, int D, int DP, float eps, void* stream) {
    LnFwdArgs a{};
    a.x = x; a.inject = inject; a.xout = xout; a.gamma = gamma; a.beta = beta; a.z = (bf16_t*)z; a.mean = mean; a.rstd = rstd;
    a.rows = B * T; a.T = T; a.D = D; a.DP = DP; a.eps = eps; a.ones_col = -1;
    return launch_ln_fwd(a, (hipStream_t)stream);
}
int v1t_layernorm_backward(const float* dz, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* gin, float* gout, float* dgamma, float* dbeta, float* dinject, void* dy_next,
                           float* dbias_next, int B, int T, int D, int DP, void* stream) {
    LnBwdArgs a{};
    a.dz = dz; a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.gin = gin; a.gout = gout; a.dgamma = dgamma; a.dbeta = dbeta;
    a.dinject = dinject; a.dy_next = (bf16_t*)dy_next; a.dbias_next = dbias_next; a.B = B; a.T = T; a.D = D; a.DP = DP;
    return launch_ln_bwd(a, (hipStream_t)stream);
}

}  // extern "C"
