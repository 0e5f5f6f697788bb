// v1t_amd — O(N) bookkeeping around the readout sample positions, fused into four small kernels:
//   readout grid:  mu_n = tanh(W2 . ELU(W0 . src_n + b0) + b2)  (or the free parameter _mu),
//                  grid[b][n] = clamp(sigma_n . eps[b][n] + mu_n, -1, 1) + shift[b]
//                  (reference gaussian2d.py:188-235, 265-268, init_grid_predictor :113-131) and its backward;
//   core shifter:  shift[b] = tanh(W4 . tanh(W2 . tanh(W0 . pupil[b] + b0) + b2) + b4)
//                  (reference core_shifter.py:24-40 with num_layers = 3, model.py:86-92) and its backward.
#pragma once
#include "common.h"

constexpr int GRID_HID = 30;  // hidden width of the grid predictor (gaussian2d.py:105)

struct GridArgs {
    int B, N, gd;             // gd = grid_predictor_dim (2 or 3); gd == 0: free parameter mu_free
    const float* src;         // (N, gd) normalised cortical coordinates
    const float* W0; const float* b0;  // (30, gd), (30)
    const float* W2; const float* b2;  // (2, 30), (2)
    const float* mu_free;     // (N, 2) when gd == 0
    const float* sigma;       // (N, 2, 2)
    const float* eps;         // (B, N, 2) or nullptr (eval: grid = clamp(mu))
    const float* shift;       // (B, 2) or nullptr
    float* grid;              // (B, N, 2) out
    // backward
    const float* dgrid;       // (B, N, 2)
    float* dW0; float* db0; float* dW2; float* db2;  // atomics (zero-initialised by the caller)
    float* dmu_free;          // (N, 2) overwritten (gd == 0)
    float* dsigma;            // (N, 2, 2) overwritten, or nullptr
    float* dshift;            // (B, 2) atomics (zero-initialised), or nullptr
    float* part;              // workspace [workgroups][part_stride] for the two-stage reduction, or nullptr (atomics)
    int part_stride;
};
int launch_grid_fwd(const GridArgs& a, hipStream_t s);
// ws (grid_bwd_ws_bytes, may be nullptr): per-workgroup partial sums of the predictor gradients and d shift, summed by a
// second small kernel. Without it every workgroup adds into the ~7 cache lines of those gradients with float atomics,
// which the L2 serialises per line (measured: ~17 ns per same-line atomic, 69 us for 8000 neurons x 16 images).
size_t grid_bwd_ws_bytes(int B, int N);
int launch_grid_bwd(const GridArgs& a, void* ws, size_t ws_bytes, hipStream_t s);

struct ShifterArgs {
    int B;
    const float* pupil;       // (B, 2)
    const float* W0; const float* b0;  // (5, 2), (5)
    const float* W2; const float* b2;  // (5, 5), (5)
    const float* W4; const float* b4;  // (2, 5), (2)
    float* shift;             // (B, 2) out
    const float* dshift;      // (B, 2)
    float* dW0; float* db0; float* dW2; float* db2; float* dW4; float* db4;  // overwritten
};
int launch_shifter_fwd(const ShifterArgs& a, hipStream_t s);
int launch_shifter_bwd(const ShifterArgs& a, hipStream_t s);
int launch_normal_fill(float* out, long long n, uint64_t seed, uint32_t stream_id, hipStream_t s);
// one launch per stage over n <= GP_MAX_UNITS units (the mice of a training step); the backward form needs every unit's workspace
constexpr int GP_MAX_UNITS = 8;
int launch_grid_fwd_multi(const GridArgs* a, int n, hipStream_t s);
int launch_grid_bwd_multi(const GridArgs* a, void* const* ws, const size_t* ws_bytes, int n, hipStream_t s);
int launch_shifter_multi(const ShifterArgs* a, int n, bool bwd, hipStream_t s);
int launch_normal_fill_multi(float* const* out, const long long* n, const uint32_t* stream_id, int units, uint64_t seed, hipStream_t s);
int launch_concat2(const float* a, int na, const float* b, int nb, int rows, float* out, int ldo, hipStream_t s);
