// v1t_amd — Gaussian2d readout (see readout.hip).
#pragma once
#include "common.h"

struct ReadoutArgs {
    const float* z;        // core map: element (b, cell, c) at z[b*zsb + cell*zsc + c]; cell = y*W + x
    long long zsb, zsc;
    int B, C, H, W, N;
    const float* grid;     // [B][N][2] (x, y) in [-1,1] (+shift)
    const float* feat;     // [N][FS] neuron-major feature weights
    int FS;
    const float* bias;     // [N] or nullptr
    float* out;            // [B][N]
    // backward
    const float* gout;     // [B][N]
    float* dz;             // same addressing as z (dzsb, dzsc), fp32 atomics, or nullptr
    long long dzsb, dzsc;
    float* dgrid;          // [B][N][2] (overwritten) or nullptr
    float* dfeat;          // [N][FS] (+=) or nullptr
    float* dbias;          // [N] (+=) or nullptr
};

int launch_readout_fwd(const ReadoutArgs& a, hipStream_t s);
// ws: optional scratch of readout_bwd_ws_bytes() bytes; with it dz is gathered through per-cell lists (no float atomics)
size_t readout_bwd_ws_bytes(int B, int H, int W, int N);
int launch_readout_bwd(const ReadoutArgs& a, void* ws, size_t ws_bytes, hipStream_t s);
// the sorted form's kernels one by one: parts = bit mask of the values below (SORT needs grid only; DZ needs SORT's scratch and gout)
enum { READOUT_BWD_SORT = 1, READOUT_BWD_PARAMS = 2, READOUT_BWD_DZ = 4, READOUT_BWD_ALL = 7 };
int launch_readout_bwd_parts(const ReadoutArgs& a, void* ws, size_t ws_bytes, int parts, hipStream_t s);
// one launch over n <= TAILS_MAX_UNITS units (the mice of a training step): stage = forward, tap sort, dz gather or parameter gradients
constexpr int TAILS_MAX_UNITS = 8;
enum { TAILS_FWD = 0, TAILS_SORT = 1, TAILS_DZ = 2, TAILS_PARAMS = 3 };
int launch_readout_multi(const ReadoutArgs* a, void* const* ws, const size_t* ws_bytes, int n, int stage, hipStream_t s);
