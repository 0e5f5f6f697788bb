// v1t_amd — device-side helpers shared by all gfx950 kernels.
// Wave = 64 lanes; MFMA = v_mfma_f32_32x32x16_bf16 (A 32x16, B 16x32, C/D 32x32 fp32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

// Development switches (A/B experiments, ablations, the round-1..5 experiment kernels) exist ONLY in experiment builds
// (`V1T_BUILD_LIB=libv1t_amd_exp.so V1T_HIPCC_EXTRA=-DV1T_EXPERIMENTS python -m v1t_amd.build`, loaded with V1T_LIB): in the product
// library dev_env() is a constant nullptr, every switch folds to its measured default and the experiment kernels are not compiled, so
// the product has no untested configurations. The product reads two documented environment options through
// std::getenv directly: V1T_DW_SIDE (api.hip) and V1T_DEBUG_SYNC.
#ifdef V1T_EXPERIMENTS
inline const char* dev_env(const char* name) { return std::getenv(name); }
#else
constexpr const char* dev_env(const char*) { return nullptr; }
#endif

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define V1T_WAVE 64
#define DEVFN __device__ __forceinline__

// ---- error codes returned by the C-ABI (mapped to RuntimeError by the Python shim) ----
#define V1T_OK 0
#define V1T_ERR_ARG -1
#define V1T_ERR_UNSUPPORTED -2
#define V1T_ERR_LAUNCH -3
#define V1T_ERR_WORKSPACE -4

// ---- MFMA 32x32x16 bf16 -------------------------------------------------------------
// Lane l (r = l&31, h = l>>5):  A[row r][k = 8h + j],  B[k = 8h + j][col r],  j = 0..7
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5), reg = 0..15.
DEVFN f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// fp16 operands (same rate, 11-bit significand): the forward linear layers, whose operand rounding dominates the error of
// the predicted responses (DESIGN.md "Numerics"). The planes travel in bf16_t-typed buffers as raw 16-bit patterns.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// a fragment whose 16-bit lanes hold fp16 values (the forward's second planes) as bf16 operand: 8 conversions + 4 packed roundings
DEVFN bf16x8 f16_frag_to_bf16(bf16x8 x) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const f16x8 h = __builtin_bit_cast(f16x8, x);
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2_ v = {(float)h[2 * j], (float)h[2 * j + 1]};
        const bf16x2_ b = __builtin_convertvector(v, bf16x2_);
        r[2 * j] = b[0];
        r[2 * j + 1] = b[1];
    }
    return r;
}
DEVFN f32x16 mfma32h(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// Second plane of a forward activation / weight next to its bf16 plane `hi` (which the backward uses): the fp16 value
// (saturated) when f16, else the bf16 residual v - hi of the split-bf16 ("bf16x3") scheme.
DEVFN bf16_t aux_plane(float v, bf16_t hi, int f16) {
    if (f16) return __builtin_bit_cast(bf16_t, (f16_t)fminf(fmaxf(v, -65504.f), 65504.f));
    return (bf16_t)(v - (float)hi);
}
DEVFN int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// An accumulator tile X (rows in registers, column on the lane) as the B operand of the next
// MFMA that sums over X's ROW index: k-step s (0,1) takes registers 8s..8s+7; element j of lane
// half h is row 16s + 8(j>>2) + 4h + (j&3) of X, so the A operand must use the same k order.
DEVFN bf16x8 acc_to_b(const f32x16& x, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)x[8 * s + j];
    return r;
}

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements is delivered
// column-major: lane 4q+p of the group supplies the address of row q, cols 4p..4p+3; lane i of the
// group receives column i, rows 0..3. EXEC must be all ones.
DEVFN bf16x4 lds_tr_read(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}

// A operand (32 rows m x 16 k) read TRANSPOSED from an LDS image stored [k][m] (row = k index,
// m contiguous), in the permuted k order that matches acc_to_b():
//   element j of lane (m = l&31, h = l>>5)  <-  img[k0 + 16s + 8(j>>2) + 4h + (j&3)][m0 + m]
// `stride` in elements. k0 already includes 16*s.
DEVFN bf16x8 lds_tr_frag(const bf16_t* img, int stride, int k0, int m0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int h = g >> 1, q = i >> 2, p = i & 3;
    const bf16_t* a = img + (k0 + 4 * h + q) * stride + m0 + 16 * (g & 1) + 4 * p;
    bf16x4 lo = lds_tr_read(a);
    bf16x4 hi = lds_tr_read(a + 8 * stride);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// Same transposed read but in NATURAL k order (element j <- img[k0 + 8h + j][m0 + m]); used when
// the other operand is also read from memory in natural order (TN GEMM).
DEVFN bf16x8 lds_tr_frag_nat(const bf16_t* img, int stride, int k0, int m0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int h = g >> 1, q = i >> 2, p = i & 3;
    const bf16_t* a = img + (k0 + 8 * h + q) * stride + m0 + 16 * (g & 1) + 4 * p;
    bf16x4 lo = lds_tr_read(a);
    bf16x4 hi = lds_tr_read(a + 4 * stride);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// ---- raw buffer access ------------------------------------------------------------------
// A buffer resource over [p, p + bytes): loads / stores take a 32-bit per-lane byte offset (one address register instead of a 64-bit
// pair per access, the base sits in scalar registers) and the hardware range-checks offset + immediate against `bytes`: an
// out-of-range load returns 0, an out-of-range store is dropped - masked rows cost no branch (offset | BUF_OOB).
constexpr uint32_t BUF_OOB = 0x80000000u;
DEVFN __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
DEVFN float buf_load_f32(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
DEVFN void buf_store_f32(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, soff, 0);
}

// ---- LDS-DMA (global_load_lds) --------------------------------------------------------
// One LDS-DMA wave-instruction: 16 B per lane from each lane's own global address to lds_base + 16*lane.
// Issued through inline asm ON PURPOSE: for the builtin, hipcc cannot tell which LDS buffer a pending DMA
// writes and drains vmcnt(0) in front of the next ds_read of ANY buffer, which serialises the prefetch of
// tile t+1 with the compute of tile t. The asm form is invisible to its wait-count bookkeeping; dma_wait()
// (vmcnt(0)) in front of the tile-end barrier orders it. M0 is saved/restored inside the same statement.
DEVFN void lds_dma16(const void* gsrc, const void* lds_dst_wave_uniform) {
    unsigned keep;
    const unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds_dst_wave_uniform;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
}
DEVFN void lds_dma4(const void* gsrc, const void* lds_dst_wave_uniform) {
    unsigned keep;
    const unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds_dst_wave_uniform;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
}
// Pin the wait for prologue loads in front of the main loop: otherwise hipcc places its vmcnt(0) at their
// first use INSIDE the loop, where it would also drain the (invisible to it) DMA queue on every iteration.
template <int N>
DEVFN void touch(const bf16x8 (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" ::"v"(x[i]));
}
DEVFN void touch(float x) { asm volatile("" ::"v"(x)); }

DEVFN void dma_wait_and_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// Global -> LDS tile staging by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
// The LDS image of a [ROWS][STR] bf16 tile is a linear array of 16-B chunks (STR/8 per row, the last
// (STR-DP)/8 of each row are padding); one wave-instruction ("piece") fills 64 consecutive chunks (1 KiB) with
// each lane's own global source address, so the row padding costs nothing but a dummy fetch.
// Issue cost is what matters (in-kernel s_memtime timeline, tools/kprof.py): with per-piece 64-bit address
// arithmetic, M0 juggling and EXEC masking a piece cost its wave ~150 cycles, 850 per 32-key tile of the forward
// kernel. Here a piece is ONE instruction: wave w owns PW consecutive pieces, the tile's row-0 address is a
// scalar base (saddr form), the lane's byte offset is loop-invariant, and pieces 1..3 of a group of four reuse
// the group's M0 through the instruction offset (it advances the LDS and the global address alike, so the lane
// offset carries 3072 - imm and the base is biased by -3072). The last piece of a tile may be partial: its
// surplus lanes re-fetch the last row into slack behind the tile (size buffers with LDS_ELEMS).
// Rows beyond T are clamped to row T-1 (finite data; every consumer masks them): only the ragged last tile
// takes that path, recomputing its offsets.
// UNIFORM: EVERY wave issues all of its PW pieces, no branch on which of them exist (a wave-uniform branch chain per issue otherwise:
// `cnt` depends on the wave) - pieces beyond the tile re-fetch the tile's last chunk into slack behind it (LDS_ELEMS grows to NWAVES x PW
// pieces). For issues that sit inside a hand-scheduled stretch, where every branch splits the compiler's scheduling region.
template <int DP, int STR, int ROWS = 32, int NWAVES = 4, bool UNIFORM = false>
struct TileDma {
    static constexpr int CPR = STR / 8;                 // chunks per LDS row
    static constexpr int NCH = ROWS * CPR;              // chunks per tile
    static constexpr int NINST = (NCH + 63) / 64;       // pieces per tile
    static constexpr int PW = (NINST + NWAVES - 1) / NWAVES;  // pieces per wave
    static constexpr int NG = (PW + 3) / 4;             // M0 groups per wave
    static constexpr int LDS_ELEMS = (UNIFORM ? NWAVES * PW : NINST) * 512;  // tile + slack of the partial last piece (UNIFORM: of the pieces beyond the tile)
    static constexpr int BIAS = 3072;
    unsigned voff[PW];
    int wave, lane, ld;
    DEVFN unsigned lane_off(int i, int max_row) const {
        const int p = 64 * (wave * PW + i) + lane;
        const int r = min(p / CPR, max_row), cc = min(p % CPR, DP / 8 - 1);
        return (unsigned)((r * ld + 8 * cc) * 2 + BIAS - 1024 * (i & 3));
    }
    DEVFN void init(int lane_, int wave_uniform, int ld_elems) {
        wave = wave_uniform; lane = lane_; ld = ld_elems;
#pragma unroll
        for (int i = 0; i < PW; ++i) voff[i] = lane_off(i, ROWS - 1);
    }
    // NTP: the non-temporal policy on every piece (an operand that is read exactly once: it streams past the L2's / Infinity Cache's LRU;
    // tools/microbench/hbm_stream: LDS-DMA reads 7.1 against 6.4 TB/s)
    template <int CNT, bool NTP = false>
    DEVFN static void group(const char* base, unsigned m0v, unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
        unsigned keep;
#define V1T_DMA_GROUP(NTS)                                                                                                                              \
        if constexpr (CNT == 4)                                                                                                                         \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6" NTS "\n\tglobal_load_lds_dwordx4 %3, %6 offset:1024" NTS "\n\t" \
                         "global_load_lds_dwordx4 %4, %6 offset:2048" NTS "\n\tglobal_load_lds_dwordx4 %5, %6 offset:3072" NTS "\n\ts_mov_b32 m0, %0"         \
                         : "=&s"(keep) : "s"(m0v), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(base) : "memory");                                             \
        else if constexpr (CNT == 3)                                                                                                                    \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5" NTS "\n\tglobal_load_lds_dwordx4 %3, %5 offset:1024" NTS "\n\t" \
                         "global_load_lds_dwordx4 %4, %5 offset:2048" NTS "\n\ts_mov_b32 m0, %0"                                                        \
                         : "=&s"(keep) : "s"(m0v), "v"(v0), "v"(v1), "v"(v2), "s"(base) : "memory");                                                      \
        else if constexpr (CNT == 2)                                                                                                                    \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4" NTS "\n\tglobal_load_lds_dwordx4 %3, %4 offset:1024" NTS "\n\t" \
                         "s_mov_b32 m0, %0"                                                                                                             \
                         : "=&s"(keep) : "s"(m0v), "v"(v0), "v"(v1), "s"(base) : "memory");                                                               \
        else if constexpr (CNT == 1)                                                                                                                    \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" NTS "\n\ts_mov_b32 m0, %0"                \
                         : "=&s"(keep) : "s"(m0v), "v"(v0), "s"(base) : "memory");
        if constexpr (NTP) { V1T_DMA_GROUP(" nt") } else { V1T_DMA_GROUP("") }
#undef V1T_DMA_GROUP
    }
    // img: element (row 0, col 0) of this (image, head) slice; t0: first row of the tile; lds: tile base (LDS_ELEMS elements)
    // MAY_RAG = false: the caller knows that the tile lies wholly inside T (no ragged check in the issue)
    template <bool MAY_RAG = true, bool NTP = false>
    DEVFN void issue(const bf16_t* img, int t0_, int T, bf16_t* lds) const {
        // a tile that lies WHOLLY beyond T (the dQ GEMM walks the 128-key padded width of dS': up to three such 32-key tiles when T % 128 <= 96 -
        // first met at T = 34 114, tests/test_gpu_longseq.py; T = 1654 has none) is fetched as T - 1 repeated: without the clamp the row clamp below
        // turned negative and the 32-bit lane offset, which the hardware zero-extends, pointed 4 GB behind the tile
        const int t0 = min(t0_, T - 1);
        // (wave-uniform by construction; the explicit readfirstlanes keep the "s" asm operands in SGPRs where hipcc's uniformity
        // analysis gives up - a tile index that comes out of a loop rotated per wave half)
        const unsigned long long bb = (unsigned long long)(uintptr_t)((const char*)(img + (size_t)t0 * ld) - BIAS);
        const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(bb >> 32)), b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bb);
        const char* base = (const char*)(uintptr_t)(((unsigned long long)b_hi << 32) | (unsigned long long)b_lo);
        const unsigned l0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds);
        const bool ragged = MAY_RAG && t0 + ROWS > T;  // wave-uniform
        unsigned v[PW];
#pragma unroll
        for (int i = 0; i < PW; ++i) v[i] = voff[i];
        if (ragged) {
            asm volatile("; ragged tile: clamp rows" ::: "memory");  // keeps this a branch (no if-conversion into the hot path)
#pragma unroll
            for (int i = 0; i < PW; ++i) v[i] = lane_off(i, T - 1 - t0);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int n0 = wave * PW + 4 * g;                       // wave-uniform
            const int cnt = UNIFORM ? min(4, PW - 4 * g) : min(min(4, PW - 4 * g), NINST - n0);  // pieces of this group that exist (UNIFORM: compile-time)
            const unsigned m0v = __builtin_amdgcn_readfirstlane(l0 + 1024u * (unsigned)n0);
            auto at = [&](int k) { return v[4 * g + k < PW ? 4 * g + k : PW - 1]; };
            if (cnt >= 4) group<4, NTP>(base, m0v, at(0), at(1), at(2), at(3));
            else if (cnt == 3) group<3, NTP>(base, m0v, at(0), at(1), at(2), 0);
            else if (cnt == 2) group<2, NTP>(base, m0v, at(0), at(1), 0, 0);
            else if (cnt == 1) group<1, NTP>(base, m0v, at(0), 0, 0, 0);
        }
    }
};

// ---- counter-based dropout ------------------------------------------------------------
// keep(seed, stream, row, col): stateless 32-bit mix; the SAME function is evaluated by the
// forward and the backward kernels and is exported through the C-ABI (v1t_dropout_mask) so that
// the CPU oracle can replay the exact mask. thresh = round(p * 2^32); keep iff hash >= thresh.
DEVFN uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t drop_key(uint64_t seed, uint32_t stream) {
    uint32_t x = (uint32_t)seed ^ (uint32_t)(seed >> 32) * 0x9E3779B9u;
    x ^= stream * 0x85EBCA6Bu + 0x165667B1u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
DEVFN uint32_t drop_hash(uint32_t key, uint32_t row, uint32_t col) {
    return mix32(key + row * 0x9E3779B1u + col * 0x85EBCA77u);
}
DEVFN bool drop_keep(uint32_t key, uint32_t row, uint32_t col, uint32_t thresh) {
    return drop_hash(key, row, col) >= thresh;
}
struct DropCfg {
    uint32_t key;     // drop_key(seed, stream)
    uint32_t thresh;  // 0 => dropout disabled
    float inv_keep;   // 1 / (1 - p)
};

// Attention-P dropout: B*H*T*T decisions per block and pass, evaluated inside three MFMA kernels, so the
// mask must be cheap. One 32-bit word serves a 2x2 block of (query, key): word(bh, q>>1, k>>1) =
// mix1(key + (bh*T2 + (q>>1))*K1 + (k>>1)*K2), T2 = (T+1)/2; element (q,k) keeps iff
// byte[2*(q&1) + (k&1)] >= thr. Byte decisions alone quantise the rate to 1/256 (rounds 1-4: 0.2544 -> 65/256 =
// 0.2539); since round 5 the threshold is dithered per 32 x 32 tile of (query, key): with the rate as
// thresh16 / 65536 a tile compares against thr = (thresh16 + f) >> 8, f = the top byte of the Weyl sum
// key + TILE + (bh*nqb + (q>>5))*K1 + (k>>5)*K2 - i.e. against thresh8 + 1 in a fraction frac8 / 256 of the tiles
// (equidistributed over the rows and columns of tiles: an additive golden-ratio recurrence, lower variance than
// a hash; `key` moves the pattern every step, block and (image, head)) and against thresh8 = thresh16 >> 8 in
// the others. Five scalar instructions per tile on wave-uniform values. An element's keep probability (over
// the keys) is then exactly 1 - thresh16/65536 (0.2544 -> 0.25439453, 2e-5 relative) at the cost of the byte
// scheme; elements of a tile share the choice, which correlates two of them by
// frac (1 - frac) / 65536 / (p (1 - p)), frac = frac8 / 256: 9e-6 at the default rate. 1/keep uses the 16-bit
// rate, so the estimator stays unbiased. (Tried in round 5, profiles/r05_small_launch_experiments.txt #21:
// 16-bit decisions, one word per query and key pair - dK/dV kernel + 5 %, + 3 % with DPP-shared words; a hashed
// per-tile dither behind a wave-uniform branch - forward + 5 %; branch-free - step + 0.5 %.)
// mix1 is one xorshift-multiply-xorshift round (byte uniformity / neighbour correlation checked).
DEVFN uint32_t mix1(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
    return x;
}
#define ADROP_K1 0x9E3779B1u
#define ADROP_K2 0x85EBCA77u
#define ADROP_TILE 0x68E31DA4u  // key offset of the per-tile threshold words
struct AttnDrop {
    uint32_t key;       // drop_key(seed, stream)
    uint32_t thresh16;  // 0 => disabled; rate = thresh16 / 65536
    uint32_t thresh8;   // thresh16 >> 8: a tile keeps iff byte >= thresh8 (+ 1 in a dithered tile)
    uint32_t frac8;     // thresh16 & 255: the fraction of dithered tiles, in 1/256
    float inv_keep;     // 65536 / (65536 - thresh16)
    float keep_prob;    // (65536 - thresh16) / 65536
};
// byte threshold of the 32 x 32 tile (qblk, kblk) = (q >> 5, k >> 5) of (image, head) bh; nqb = ceil(T / 32). Branch-free on purpose: a
// wave-uniform branch here splits the softmax stretch of the attention kernels into basic blocks (forward + 5 %).
DEVFN uint32_t attn_tile_thresh(const AttnDrop& d, uint32_t bh, uint32_t nqb, uint32_t qblk, uint32_t kblk) {
    const uint32_t f = (d.key + ADROP_TILE + (bh * nqb + qblk) * ADROP_K1 + kblk * ADROP_K2) >> 24;
    return (d.thresh16 + f) >> 8;
}
DEVFN bool attn_drop_keep(const AttnDrop& d, uint32_t bh, uint32_t T, uint32_t q, uint32_t k) {
    const uint32_t T2 = (T + 1) >> 1, nqb = (T + 31) >> 5;
    const uint32_t w = mix1(d.key + (bh * T2 + (q >> 1)) * ADROP_K1 + (k >> 1) * ADROP_K2);
    return ((w >> (8 * (2 * (q & 1) + (k & 1)))) & 0xFFu) >= attn_tile_thresh(d, bh, nqb, q >> 5, k >> 5);
}

DEVFN float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // raw v_exp_f32
// ---- small math --------------------------------------------------------------------------
DEVFN float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
DEVFN float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// Sum over the 64 lanes, returned to every lane. DPP adds (quad swaps, half-row and row mirrors, row broadcasts)
// leave the total in lane 63, one v_readlane hands it back: 6 VALU + 1, no LDS traffic. The __shfl_xor butterfly
// it replaces compiles to six ds_bpermute round trips (~100 cycles each on an otherwise idle wave).
// EXEC must be all ones (every caller reduces with inactive lanes contributing zeros).
template <int CTRL, int ROW_MASK>
DEVFN float dpp_add(float v) {
    const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(x);
}
DEVFN float wave_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);  // row_mirror: every lane holds its row's sum
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
DEVFN float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware bijective block-id remap (8 XCDs; blocks b and b+8 share an XCD's L2): gives each XCD
// a contiguous chunk of the logical tile space so neighbouring tiles reuse operands in one L2.
DEVFN int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- optional per-launch hipEvent timing of one kernel class (see v1t_profile_enable) ----
enum ProfClass { PROF_ATTN_FWD = 0, PROF_ATTN_DQ = 1, PROF_ATTN_DKV = 2, PROF_GEMM_NT = 3, PROF_GEMM_TN = 4, PROF_READOUT_FWD = 5, PROF_READOUT_BWD = 6, PROF_ROLLOUT_MM = 7 };
void prof_begin(int cls, hipStream_t s);
void prof_end(int cls, hipStream_t s);
