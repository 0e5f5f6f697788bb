// Streaming validation / evaluation metrics on the device.
//
// The reference stacks every prediction of an epoch on the host (train.py:24-25,186 `vstack(...).cpu()`, utils.py:93
// `predictions.cpu()`) and computes the metrics there. Here each micro-batch's (B, N) predictions are folded into
// per-neuron fp64 moments as they leave the readout, and only the final per-neuron vectors ever leave HBM:
//   compute_metrics (train.py:29-39): msse (losses.py:25-29), poisson_loss (:32-40), correlation(dim=0) (:43-58)
//   Metrics (metrics.py:65-142): single-trial correlation, correlation to the repeat average, FEV / FEVe.
// All kernels are HBM-streaming (one read of pred and target per element, 8 B/element algorithmic) with one thread per
// neuron column, so a row is one coalesced 256 B-per-wave read; moments are fp64 so that the one-pass raw-moment form
// of the variance is at least as accurate as the reference's two-pass fp32.
#include <hip/hip_runtime.h>

#include "../../include/v1t_amd.h"

namespace v1t {
namespace {

constexpr int MT = 256;

__device__ inline double block_sum(double v, double* lds) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < MT / 64; ++i) r += lds[i];
    __syncthreads();
    return r;  // valid on thread 0
}

// acc[5][N] += (Σp, Σt, Σp², Σt², Σp·t) over this block's rows; scal[0] += Σ(t − p)², scal[1] += Σ(p − t·log(p + eps)).
__global__ __launch_bounds__(MT) void metrics_accumulate_kernel(const float* pred, const float* tgt, int B, int N, float eps, double* acc,
                                                                double* scal) {
    __shared__ double lds[MT / 64];
    const int n = blockIdx.x * MT + threadIdx.x;
    double sp = 0, st = 0, spp = 0, stt = 0, spt = 0, se = 0, spl = 0;
    if (n < N) {
        auto fold = [&](float pf, float tf) {
            const double p = pf, t = tf;
            sp += p;
            st += t;
            spp += p * p;
            stt += t * t;
            spt += p * t;
            const float d = tf - pf;  // the reference's element arithmetic is fp32 (losses.py:27,38)
            se += (double)(d * d);
            spl += (double)(pf - tf * logf(pf + eps));
        };
        for (int b = blockIdx.y; b < B; b += gridDim.y) fold(pred[(size_t)b * N + n], tgt[(size_t)b * N + n]);
        atomicAdd(&acc[n], sp);
        atomicAdd(&acc[(size_t)N + n], st);
        atomicAdd(&acc[(size_t)2 * N + n], spp);
        atomicAdd(&acc[(size_t)3 * N + n], stt);
        atomicAdd(&acc[(size_t)4 * N + n], spt);
    }
    const double e = block_sum(se, lds);
    const double l = block_sum(spl, lds);
    if (threadIdx.x == 0) {
        atomicAdd(&scal[0], e);
        atomicAdd(&scal[1], l);
    }
}

// losses.py:43-58 over dim 0 from the moments: corr = (E[pt] − mp·mt) / ((sp + eps)(st + eps)), population std.
__device__ inline double corr_from_moments(double sp, double st, double spp, double stt, double spt, double cnt, double eps) {
    const double mp = sp / cnt, mt = st / cnt;
    const double vp = fmax(spp / cnt - mp * mp, 0.0), vt = fmax(stt / cnt - mt * mt, 0.0);
    return (spt / cnt - mp * mt) / ((sqrt(vp) + eps) * (sqrt(vt) + eps));
}

__global__ __launch_bounds__(MT) void metrics_correlation_kernel(const double* acc, double count, int N, float eps, float* corr, double* mean_out) {
    __shared__ double lds[MT / 64];
    const int n = blockIdx.x * MT + threadIdx.x;
    double c = 0.0;
    if (n < N) {
        c = corr_from_moments(acc[n], acc[(size_t)N + n], acc[(size_t)2 * N + n], acc[(size_t)3 * N + n], acc[(size_t)4 * N + n], count, (double)eps);
        if (corr) corr[n] = (float)c;
    }
    const double s = block_sum(c, lds);
    if (threadIdx.x == 0 && mean_out) atomicAdd(mean_out, s / (double)N);
}

// Repeated presentations (metrics.py:41-58): group[b] = index of trial b's image. gacc[3][G][N] += (Σt, Σt², Σp) per
// (image, neuron); sqerr[N] += Σ(t − p)². A thread owns a neuron column of its row chunk; rows of one image collide only
// across chunks (fp64 atomics).
__global__ __launch_bounds__(MT) void metrics_group_accumulate_kernel(const float* pred, const float* tgt, const int* group, int B, int N, int G,
                                                                      double* gacc, double* sqerr) {
    const int n = blockIdx.x * MT + threadIdx.x;
    if (n >= N) return;
    const size_t plane = (size_t)G * N;
    double se = 0.0;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const int g = group[b];
        if (g < 0 || g >= G) continue;
        const float pf = pred[(size_t)b * N + n], tf = tgt[(size_t)b * N + n];
        const double t = tf;
        atomicAdd(&gacc[(size_t)g * N + n], t);
        atomicAdd(&gacc[plane + (size_t)g * N + n], t * t);
        atomicAdd(&gacc[2 * plane + (size_t)g * N + n], (double)pf);
        const float d = tf - pf;
        se += (double)(d * d);
    }
    atomicAdd(&sqerr[n], se);
}

// metrics.py:77-142 per neuron: correlation between the per-image mean response and mean prediction (:77-93), and
// FEV / FEVe (:95-127): noise = mean over images of var(repeats, ddof=1), total = var(all trials, ddof=1),
// fev = (total − noise) / total, feve = 1 − (mean (t − p)² − noise) / (total − noise).
__global__ __launch_bounds__(MT) void metrics_group_finalize_kernel(const double* gacc, const int* gcount, const double* sqerr, int G, int N, float eps,
                                                                    float* corr_avg, float* fev, float* feve) {
    const int n = blockIdx.x * MT + threadIdx.x;
    if (n >= N) return;
    const size_t plane = (size_t)G * N;
    double st = 0, stt = 0, noise = 0, total_cnt = 0;
    double mr = 0, mp = 0, mrr = 0, mpp = 0, mrp = 0;
    int used = 0;
    for (int g = 0; g < G; ++g) {
        const double c = (double)gcount[g];
        if (c <= 0) continue;
        const double a = gacc[(size_t)g * N + n], a2 = gacc[plane + (size_t)g * N + n], p = gacc[2 * plane + (size_t)g * N + n];
        const double r = a / c, q = p / c;
        mr += r;
        mp += q;
        mrr += r * r;
        mpp += q * q;
        mrp += r * q;
        noise += (a2 - c * r * r) / (c - 1.0);  // one repeat -> 0/0 = nan, as np.var(ddof=1) gives
        st += a;
        stt += a2;
        total_cnt += c;
        ++used;
    }
    const double ng = (double)used;
    if (corr_avg) corr_avg[n] = (float)corr_from_moments(mr, mp, mrr, mpp, mrp, ng, (double)eps);
    const double mean = st / total_cnt;
    const double total = (stt - total_cnt * mean * mean) / (total_cnt - 1.0);
    noise /= ng;
    const double pv = sqerr[n] / total_cnt;
    if (fev) fev[n] = (float)((total - noise) / total);
    if (feve) feve[n] = (float)(1.0 - (pv - noise) / (total - noise));
}

inline int ok() { return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH; }
inline int row_chunks(int B) { return B <= 4 ? 1 : (B + 3) / 4 > 64 ? 64 : (B + 3) / 4; }

}  // namespace
}  // namespace v1t

using namespace v1t;

extern "C" {

int v1t_metrics_accumulate(const float* pred, const float* target, int B, int N, float eps, double* acc, double* scal, void* stream) {
    if (!pred || !target || !acc || !scal || B < 0 || N <= 0) return V1T_ERR_ARG;
    if (B == 0) return V1T_OK;
    hipLaunchKernelGGL(metrics_accumulate_kernel, dim3((N + MT - 1) / MT, row_chunks(B)), dim3(MT), 0, (hipStream_t)stream, pred, target, B, N, eps, acc,
                       scal);
    return ok();
}

int v1t_metrics_correlation(const double* acc, long long count, int N, float eps, float* corr, double* mean_out, void* stream) {
    if (!acc || count <= 0 || N <= 0 || (!corr && !mean_out)) return V1T_ERR_ARG;
    hipLaunchKernelGGL(metrics_correlation_kernel, dim3((N + MT - 1) / MT), dim3(MT), 0, (hipStream_t)stream, acc, (double)count, N, eps, corr, mean_out);
    return ok();
}

int v1t_metrics_group_accumulate(const float* pred, const float* target, const int* group, int B, int N, int G, double* gacc, double* sqerr,
                                 void* stream) {
    if (!pred || !target || !group || !gacc || !sqerr || B < 0 || N <= 0 || G <= 0) return V1T_ERR_ARG;
    if (B == 0) return V1T_OK;
    hipLaunchKernelGGL(metrics_group_accumulate_kernel, dim3((N + MT - 1) / MT, row_chunks(B)), dim3(MT), 0, (hipStream_t)stream, pred, target, group, B, N,
                       G, gacc, sqerr);
    return ok();
}

int v1t_metrics_group_finalize(const double* gacc, const int* gcount, const double* sqerr, int G, int N, float eps, float* corr_avg, float* fev,
                               float* feve, void* stream) {
    if (!gacc || !gcount || !sqerr || G <= 0 || N <= 0) return V1T_ERR_ARG;
    hipLaunchKernelGGL(metrics_group_finalize_kernel, dim3((N + MT - 1) / MT), dim3(MT), 0, (hipStream_t)stream, gacc, gcount, sqerr, G, N, eps, corr_avg,
                       fev, feve);
    return ok();
}

}  // extern "C"
