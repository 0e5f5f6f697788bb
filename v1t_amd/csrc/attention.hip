// v1t_amd — fused attention kernels (gfx950). See attention.h.
//
// Common geometry: workgroup = 4 waves, one (batch, head); a wave owns 32 rows (queries in the
// forward and dQ kernels, keys in the dK/dV kernel) and walks the other axis in tiles of 32.
// All products are "swapped" so that the softmax axis bookkeeping is lane-local:
//   forward : S^T[key][q] = K . Q^T   (A = K tile rows from LDS, B = Q fragments in registers)
//             q sits on the lane, the 32 keys of a tile in 16 registers x 2 lane halves, so the
//             row max/sum is 15 in-lane ops + one cross-half exchange;
//             O^T[d][q] += V^T . P^T   (A = V tile read TRANSPOSED with ds_read_b64_tr_b16,
//             B = the S^T accumulator converted to bf16 in place — no LDS round trip).
//   dQ      : same orientation; dQ^T[d][q] += K^T . dS^T.
//   dK/dV   : S[q][key] = Q . K^T (key on the lane), dV^T += dO^T . P, dK^T += Q^T . dS.
// The backward is split into a dQ kernel and a dK/dV kernel (7 MFMA products instead of 5): a
// single-kernel backward needs fp32 atomics for dQ, and at T=1654, dh=160 those atomic bytes
// (B*H*T*160*4 per 128-key block) exceed the chip's ~1.3 TB/s atomic rate by far.
#include "attention.h"

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

// stage a 32-row x DP tile: rows t0.. of image `b`, zero-filled beyond T
template <int DP>
DEVFN void tile_gload(u32x4 (&r)[Geo<DP>::ITERS], const bf16_t* base, int ld, int t0, int T, int tid) {
#pragma unroll
    for (int i = 0; i < Geo<DP>::ITERS; ++i) {
        const int c = tid + 256 * i, row = c / (DP / 8), cc = c % (DP / 8);
        if (c < Geo<DP>::CHUNKS)
            r[i] = (t0 + row < T) ? *(const u32x4*)(base + (size_t)(t0 + row) * ld + 8 * cc) : u32x4{0, 0, 0, 0};
    }
}
template <int DP, int STR>
DEVFN void tile_swrite(const u32x4 (&r)[Geo<DP>::ITERS], bf16_t* s, int tid) {
#pragma unroll
    for (int i = 0; i < Geo<DP>::ITERS; ++i) {
        const int c = tid + 256 * i, row = c / (DP / 8), cc = c % (DP / 8);
        if (c < Geo<DP>::CHUNKS) *(u32x4*)(s + row * STR + 8 * cc) = r[i];
    }
}

DEVFN void zero16(f32x16& x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = 0.f;
}

// ------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
    using G = Geo<DP>;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][32 * G::RSTR];
    __shared__ __attribute__((aligned(16))) bf16_t sV[2][32 * G::TSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * 128 + 32 * wave + (lane & 31);
    const int h2 = lane >> 5;
    const int HD = a.H * DP;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    const bf16_t* kbase = qkv_b + HD + h * DP;
    const bf16_t* vbase = qkv_b + 2 * HD + h * DP;
    const float c = a.scale[a.scale_per_head ? h : 0] * LOG2E;

    bf16x8 qf[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        u32x4 t = (q < a.T) ? *(const u32x4*)(qkv_b + (size_t)q * a.ldqkv + h * DP + 16 * ks + 8 * h2) : u32x4{0, 0, 0, 0};
        qf[ks] = *(bf16x8*)&t;
    }
    f32x16 o[G::DB];
#pragma unroll
    for (int d = 0; d < G::DB; ++d) zero16(o[d]);
    float m = NEG_BIG, lsum = 0.f;
    const uint32_t drow = (uint32_t)((b * a.H + h) * a.T + q) * 0x9E3779B1u + a.drop.key;

    const int nt = (a.T + 31) / 32;
    u32x4 rk[G::ITERS], rv[G::ITERS];
    tile_gload<DP>(rk, kbase, a.ldqkv, 0, a.T, tid);
    tile_gload<DP>(rv, vbase, a.ldqkv, 0, a.T, tid);
    tile_swrite<DP, G::RSTR>(rk, sK[0], tid);
    tile_swrite<DP, G::TSTR>(rv, sV[0], tid);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nt) {
            tile_gload<DP>(rk, kbase, a.ldqkv, 32 * (kt + 1), a.T, tid);
            tile_gload<DP>(rv, vbase, a.ldqkv, 32 * (kt + 1), a.T, tid);
        }
        f32x16 s;
        zero16(s);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(&sK[buf][(lane & 31) * G::RSTR + 16 * ks + 8 * h2]);
            s = mfma32(kf, qf[ks], s);
        }
        float pmax = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * kt + acc_row(r, lane);
            float t = s[r] * c;
            if (key >= a.T || (a.mask_diag && key == q)) t = NEG_BIG;
            s[r] = t;
            pmax = fmaxf(pmax, t);
        }
        pmax = fmaxf(pmax, __shfl_xor(pmax, 32));
        if (!__all(pmax <= m + RESCALE_THR)) {
            const float mn = fmaxf(m, pmax);
            const float alpha = fast_exp2(m - mn);
#pragma unroll
            for (int d = 0; d < G::DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            lsum *= alpha;
            m = mn;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = fast_exp2(s[r] - m);
            lsum += p;
            if (a.drop.thresh) {
                const uint32_t key = 32 * kt + acc_row(r, lane);
                p = (mix32(drow + key * 0x85EBCA77u) >= a.drop.thresh) ? p * a.drop.inv_keep : 0.f;
            }
            s[r] = p;
        }
        const bf16x8 p0 = acc_to_b(s, 0), p1 = acc_to_b(s, 1);
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            const bf16x8 v0 = lds_tr_frag(sV[buf], G::TSTR, 0, 32 * d, lane);
            o[d] = mfma32(v0, p0, o[d]);
            const bf16x8 v1 = lds_tr_frag(sV[buf], G::TSTR, 16, 32 * d, lane);
            o[d] = mfma32(v1, p1, o[d]);
        }
        if (kt + 1 < nt) {
            tile_swrite<DP, G::RSTR>(rk, sK[buf ^ 1], tid);
            tile_swrite<DP, G::TSTR>(rv, sV[buf ^ 1], tid);
        }
        __syncthreads();
    }
    const float ltot = lsum + __shfl_xor(lsum, 32);
    const float inv = 1.0f / ltot;
    if (q < a.T) {
        if (h2 == 0) a.lse2[((size_t)b * a.H + h) * a.T + q] = m + log2f(ltot);
        bf16_t* orow = a.o + ((size_t)b * a.T + q) * a.ldo + h * DP;
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                bf16x4 w, wl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = o[d][4 * rq + j] * inv;
                    w[j] = (bf16_t)v;
                    wl[j] = (bf16_t)(v - (float)w[j]);
                }
                *(bf16x4*)(orow + 32 * d + 8 * rq + 4 * h2) = w;
                if (a.o_lo) *(bf16x4*)(a.o_lo + (orow - a.o) + 32 * d + 8 * rq + 4 * h2) = wl;
            }
    }
}

// delta[b][h][t] = sum_d dO * O
template <int DP>
__global__ void attn_delta_kernel(AttnArgs a, float* delta) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = a.B * a.T * a.H;
    if (idx >= total) return;
    const int h = idx % a.H, row = idx / a.H;
    const int b = row / a.T, t = row % a.T;
    const bf16_t* po = a.o + (size_t)row * a.ldo + h * DP;
    const bf16_t* pd = a.dO + (size_t)row * a.lddo + h * DP;
    float acc = 0.f;
#pragma unroll 4
    for (int c = 0; c < DP / 8; ++c) {
        const bf16x8 x = *(const bf16x8*)(po + 8 * c);
        const bf16x8 y = *(const bf16x8*)(pd + 8 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (float)x[j] * (float)y[j];
    }
    delta[((size_t)b * a.H + h) * a.T + t] = acc;
}

// ------------------------------------------------------------------------------------------
// DP >= 128: Q + dO fragments (2*DP/4 VGPRs) + the dQ accumulator (DP/2) + S/dP/staging exceed 256
// registers, so that shape runs one wave per SIMD with the 512-register budget (no spills).
template <int DP>
__global__ __launch_bounds__(256, (DP >= 128 ? 1 : 2)) void attn_bwd_dq_kernel(AttnArgs a) {
    using G = Geo<DP>;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][32 * G::RSTR];
    __shared__ __attribute__((aligned(16))) bf16_t sV[2][32 * G::RSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * 128 + 32 * wave + (lane & 31);
    const int h2 = lane >> 5;
    const int HD = a.H * DP;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    const bf16_t* kbase = qkv_b + HD + h * DP;
    const bf16_t* vbase = qkv_b + 2 * HD + h * DP;
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
    const float lse = a.lse2[sidx], dl = a.delta[sidx];
    f32x16 dq[G::DB];
#pragma unroll
    for (int d = 0; d < G::DB; ++d) zero16(dq[d]);
    float dsc = 0.f;
    const uint32_t drow = (uint32_t)((b * a.H + h) * a.T + q) * 0x9E3779B1u + a.drop.key;

    const int nt = (a.T + 31) / 32;
    u32x4 rk[G::ITERS], rv[G::ITERS];
    tile_gload<DP>(rk, kbase, a.ldqkv, 0, a.T, tid);
    tile_gload<DP>(rv, vbase, a.ldqkv, 0, a.T, tid);
    tile_swrite<DP, G::RSTR>(rk, sK[0], tid);
    tile_swrite<DP, G::RSTR>(rv, sV[0], tid);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nt) {
            tile_gload<DP>(rk, kbase, a.ldqkv, 32 * (kt + 1), a.T, tid);
            tile_gload<DP>(rv, vbase, a.ldqkv, 32 * (kt + 1), a.T, tid);
        }
        f32x16 s, dp;
        zero16(s);
        zero16(dp);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const int off = (lane & 31) * G::RSTR + 16 * ks + 8 * h2;
            const bf16x8 kf = *(const bf16x8*)(&sK[buf][off]);
            s = mfma32(kf, qf[ks], s);
            const bf16x8 vf = *(const bf16x8*)(&sV[buf][off]);
            dp = mfma32(vf, dof[ks], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * kt + acc_row(r, lane);
            const bool ok = qok && key < a.T && !(a.mask_diag && key == q);
            const float p = ok ? fast_exp2(s[r] * c - lse) : 0.f;
            float g = dp[r];
            if (a.drop.thresh) g = (mix32(drow + (uint32_t)key * 0x85EBCA77u) >= a.drop.thresh) ? g * a.drop.inv_keep : 0.f;
            const float ds = p * (g - dl);
            dsc += ds * s[r];
            s[r] = ds;
        }
        const bf16x8 b0 = acc_to_b(s, 0), b1 = acc_to_b(s, 1);
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            const bf16x8 k0 = lds_tr_frag(sK[buf], G::RSTR, 0, 32 * d, lane);
            dq[d] = mfma32(k0, b0, dq[d]);
            const bf16x8 k1 = lds_tr_frag(sK[buf], G::RSTR, 16, 32 * d, lane);
            dq[d] = mfma32(k1, b1, dq[d]);
        }
        if (kt + 1 < nt) {
            tile_swrite<DP, G::RSTR>(rk, sK[buf ^ 1], tid);
            tile_swrite<DP, G::RSTR>(rv, sV[buf ^ 1], tid);
        }
        __syncthreads();
    }
    if (a.dscale) {
        const float tot = wave_sum(dsc);
        if (lane == 0) atomicAdd(&a.dscale[a.scale_per_head ? h : 0], tot);
    }
    if (qok) {
        bf16_t* orow = a.dqkv + ((size_t)b * a.T + q) * a.lddqkv + h * DP;
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                bf16x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (bf16_t)(dq[d][4 * rq + j] * sc);
                *(bf16x4*)(orow + 32 * d + 8 * rq + 4 * h2) = w;
            }
    }
}

// ------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(AttnArgs a) {
    using G = Geo<DP>;
    __shared__ __attribute__((aligned(16))) bf16_t sQ[2][32 * G::RSTR];
    __shared__ __attribute__((aligned(16))) bf16_t sD[2][32 * G::RSTR];
    __shared__ __attribute__((aligned(16))) float sL[2][32];
    __shared__ __attribute__((aligned(16))) float sDl[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int key = blockIdx.x * 128 + 32 * wave + (lane & 31);
    const int h2 = lane >> 5;
    const int HD = a.H * DP;
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.T * a.ldqkv;
    const bf16_t* qbase = qkv_b + h * DP;
    const bf16_t* dobase = a.dO + (size_t)b * a.T * a.lddo + h * DP;
    const float* lbase = a.lse2 + ((size_t)b * a.H + h) * a.T;
    const float* dbase = a.delta + ((size_t)b * a.H + h) * a.T;
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
    const uint32_t dcol = (uint32_t)key * 0x85EBCA77u + a.drop.key;
    const uint32_t drow0 = (uint32_t)((b * a.H + h) * a.T);

    const int nt = (a.T + 31) / 32;
    u32x4 rq[G::ITERS], rd[G::ITERS];
    float rl = 0.f, rdl = 0.f;
    auto gload = [&](int t) {
        tile_gload<DP>(rq, qbase, a.ldqkv, 32 * t, a.T, tid);
        tile_gload<DP>(rd, dobase, a.lddo, 32 * t, a.T, tid);
        if (tid < 32) {
            const int qq = 32 * t + tid;
            rl = (qq < a.T) ? lbase[qq] : 0.f;
            rdl = (qq < a.T) ? dbase[qq] : 0.f;
        }
    };
    auto swrite = [&](int buf) {
        tile_swrite<DP, G::RSTR>(rq, sQ[buf], tid);
        tile_swrite<DP, G::RSTR>(rd, sD[buf], tid);
        if (tid < 32) {
            sL[buf][tid] = rl;
            sDl[buf][tid] = rdl;
        }
    };
    gload(0);
    swrite(0);
    __syncthreads();
    for (int qt = 0; qt < nt; ++qt) {
        const int buf = qt & 1;
        if (qt + 1 < nt) gload(qt + 1);
        f32x16 s, dp;
        zero16(s);
        zero16(dp);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const int off = (lane & 31) * G::RSTR + 16 * ks + 8 * h2;
            const bf16x8 qa = *(const bf16x8*)(&sQ[buf][off]);
            s = mfma32(qa, kf[ks], s);
            const bf16x8 da = *(const bf16x8*)(&sD[buf][off]);
            dp = mfma32(da, vf[ks], dp);
        }
        // rows of s / dp = queries 32qt + acc_row(r, lane); column = this lane's key
        f32x4 lse4[4], dl4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            lse4[g] = *(const f32x4*)(&sL[buf][8 * g + 4 * h2]);
            dl4[g] = *(const f32x4*)(&sDl[buf][8 * g + 4 * h2]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = 32 * qt + acc_row(r, lane);
            const bool ok = kok && qq < a.T && !(a.mask_diag && key == qq);
            const float p = ok ? fast_exp2(s[r] * c - lse4[r >> 2][r & 3]) : 0.f;
            float pd = p, g = dp[r];
            if (a.drop.thresh) {
                const bool keep = mix32(dcol + (drow0 + (uint32_t)qq) * 0x9E3779B1u) >= a.drop.thresh;
                pd = keep ? p * a.drop.inv_keep : 0.f;
                g = keep ? g * a.drop.inv_keep : 0.f;
            }
            dp[r] = pd;                             // dropped P   -> dV
            s[r] = p * (g - dl4[r >> 2][r & 3]);    // dS (unscaled) -> dK
        }
        const bf16x8 p0 = acc_to_b(dp, 0), p1 = acc_to_b(dp, 1);
        const bf16x8 s0 = acc_to_b(s, 0), s1 = acc_to_b(s, 1);
#pragma unroll
        for (int d = 0; d < G::DB; ++d) {
            const bf16x8 do0 = lds_tr_frag(sD[buf], G::RSTR, 0, 32 * d, lane);
            dv[d] = mfma32(do0, p0, dv[d]);
            const bf16x8 do1 = lds_tr_frag(sD[buf], G::RSTR, 16, 32 * d, lane);
            dv[d] = mfma32(do1, p1, dv[d]);
            const bf16x8 q0 = lds_tr_frag(sQ[buf], G::RSTR, 0, 32 * d, lane);
            dk[d] = mfma32(q0, s0, dk[d]);
            const bf16x8 q1 = lds_tr_frag(sQ[buf], G::RSTR, 16, 32 * d, lane);
            dk[d] = mfma32(q1, s1, dk[d]);
        }
        if (qt + 1 < nt) swrite(buf ^ 1);
        __syncthreads();
    }
    if (kok) {
        bf16_t* orow = a.dqkv + ((size_t)b * a.T + key) * a.lddqkv + h * DP;
#pragma unroll
        for (int d = 0; d < G::DB; ++d)
#pragma unroll
            for (int rq4 = 0; rq4 < 4; ++rq4) {
                bf16x4 wk, wv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wk[j] = (bf16_t)(dk[d][4 * rq4 + j] * sc);
                    wv[j] = (bf16_t)(dv[d][4 * rq4 + j]);
                }
                *(bf16x4*)(orow + HD + 32 * d + 8 * rq4 + 4 * h2) = wk;
                *(bf16x4*)(orow + 2 * HD + 32 * d + 8 * rq4 + 4 * h2) = wv;
            }
    }
}

template <int DP>
int launch_fwd_t(const AttnArgs& a, hipStream_t s) {
    dim3 grid((a.T + 127) / 128, a.H, a.B);
    prof_begin(PROF_ATTN_FWD, s);
    hipLaunchKernelGGL((attn_fwd_kernel<DP>), grid, dim3(256), 0, s, a);
    prof_end(PROF_ATTN_FWD, s);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
template <int DP>
int launch_bwd_t(const AttnArgs& a, hipStream_t s) {
    dim3 grid((a.T + 127) / 128, a.H, a.B);
    prof_begin(PROF_ATTN_DQ, s);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DP>), grid, dim3(256), 0, s, a);
    prof_end(PROF_ATTN_DQ, s);
    prof_begin(PROF_ATTN_DKV, s);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DP>), grid, dim3(256), 0, s, a);
    prof_end(PROF_ATTN_DKV, s);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
template <int DP>
int launch_delta_t(const AttnArgs& a, float* delta, hipStream_t s) {
    const int total = a.B * a.T * a.H;
    hipLaunchKernelGGL((attn_delta_kernel<DP>), dim3((total + 255) / 256), dim3(256), 0, s, a, delta);
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

int launch_attn_fwd(const AttnArgs& a, int DP, hipStream_t s) { DP_DISPATCH(launch_fwd_t, a, s) }
int launch_attn_delta(const AttnArgs& a, int DP, float* delta, hipStream_t s) { DP_DISPATCH(launch_delta_t, a, delta, s) }
int launch_attn_bwd(const AttnArgs& a, int DP, hipStream_t s) { DP_DISPATCH(launch_bwd_t, a, s) }
