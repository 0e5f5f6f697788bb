// v1t_amd — HBM-bound kernels (gfx950). See elementwise.h.
#include "elementwise.h"

namespace {

// ------------------------------------------------------------------------------------------
__global__ void pack_kernel(const float* __restrict__ params, char* shadow, const PackDesc* descs) {
    const PackDesc d = descs[blockIdx.y];
    const long long total = (long long)d.drows * d.dcols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / d.dcols), c = (int)(i % d.dcols);
        const int pr = d.transpose ? c : r, pc = d.transpose ? r : c;
        const int rs = pr / d.rseg_pad, rr = pr % d.rseg_pad;
        const int cs = pc / d.cseg_pad, cr = pc % d.cseg_pad;
        float v = 0.f;
        if (rr < d.rseg_valid && cr < d.cseg_valid)
            v = params[d.src_off + (long long)(rs * d.rseg_valid + rr) * d.src_ld + cs * d.cseg_valid + cr];
        if (d.out_f32) ((float*)(shadow + d.dst_off))[i] = v;
        else if (d.lo_plane) ((bf16_t*)(shadow + d.dst_off))[i] = aux_plane(v, (bf16_t)v, d.lo_plane == 2);
        else ((bf16_t*)(shadow + d.dst_off))[i] = (bf16_t)v;
    }
}

// ------------------------------------------------------------------------------------------
// Patch embedding (reference vit.py:66-72,122-128): x[b][0] = cls + pos[0];
// x[b][1+l] = U[b][l] . Wp^T + bp + pos[1+l], U = unfold(k=P, s=stride) in (c,kh,kw) order; dropout.
// fp32 VALU: 0.03 GFLOP/image, HBM/LDS-bound. Workgroup = 64 patches of one image; thread = output
// channel d; the image (C*IH*IW floats) and Wp^T ([j][d], conflict-free across d) live in LDS and
// patch pixels are LDS broadcasts.
constexpr int PCHUNK = 64;

__global__ __launch_bounds__(256) void patch_fwd_kernel(PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int PD = a.C * a.P * a.P, L = a.NH * a.NW, T = L + 1;
    float* sImg = smem;
    float* sW = smem + a.C * a.IH * a.IW;
    const int b = blockIdx.y, l0 = blockIdx.x * PCHUNK, tid = threadIdx.x;
    const float* img = a.img + (size_t)b * a.C * a.IH * a.IW;
    for (int i = tid; i < a.C * a.IH * a.IW; i += 256) sImg[i] = img[i];
    for (int i = tid; i < PD * a.D; i += 256) {
        const int d = i / PD, j = i % PD;  // coalesced read of W[d][j]
        sW[j * a.D + d] = a.W[i];
    }
    __syncthreads();
    const int d = tid;
    if (d >= a.DP) return;
    const bool dval = d < a.D;
    if (blockIdx.x == 0) {
        const int row = b * T;
        float v = dval ? a.cls[d] + a.pos[d] : 0.f;
        if (a.drop.thresh && dval) v = drop_keep(a.drop.key, row, d, a.drop.thresh) ? v * a.drop.inv_keep : 0.f;
        a.x[(size_t)row * a.DP + d] = v;
    }
    const int np = min(PCHUNK, L - l0);
    const float bias = dval ? a.bias[d] : 0.f;
    for (int p0 = 0; p0 < np; p0 += 8) {
        float acc[8];
        int base[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u] = 0.f;
            const int l = min(l0 + p0 + u, L - 1);
            base[u] = (l / a.NW) * a.stride * a.IW + (l % a.NW) * a.stride;
        }
        if (dval) {
            int j = 0;
            for (int c = 0; c < a.C; ++c)
                for (int kh = 0; kh < a.P; ++kh)
                    for (int kw = 0; kw < a.P; ++kw, ++j) {
                        const float w = sW[j * a.D + d];
                        const int off = (c * a.IH + kh) * a.IW + kw;
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc[u] += w * sImg[base[u] + off];
                    }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int l = l0 + p0 + u;
            if (l < L && p0 + u < np) {
                const int row = b * T + 1 + l;
                float v = dval ? acc[u] + bias + a.pos[(size_t)(1 + l) * a.D + d] : 0.f;
                if (a.drop.thresh && dval) v = drop_keep(a.drop.key, row, d, a.drop.thresh) ? v * a.drop.inv_keep : 0.f;
                a.x[(size_t)row * a.DP + d] = v;
            }
        }
    }
}

// dpos[t][d] += sum_b gd[b][t][d]; dcls[d] += sum_b gd[b][0][d]   (gd = g * mask / keep)
__global__ void patch_bwd_pos_kernel(PatchArgs a) {
    const int L = a.NH * a.NW, T = L + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * a.D) return;
    const int t = idx / a.D, d = idx % a.D;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) {
        const int row = b * T + t;
        float g = a.x[(size_t)row * a.DP + d];
        if (a.drop.thresh) g = drop_keep(a.drop.key, row, d, a.drop.thresh) ? g * a.drop.inv_keep : 0.f;
        s += g;
    }
    a.dpos[idx] += s;
    if (t == 0) a.dcls[d] += s;
}

// U[b*T + t][j]: thread = one 16-B chunk (8 columns) of one row
__global__ __launch_bounds__(256) void patch_unfold_kernel(PatchArgs a, bf16_t* u_hi, bf16_t* u_lo, int lo_f16, int ldu) {
    const int PD = a.C * a.P * a.P, PP = a.P * a.P, L = a.NH * a.NW, T = L + 1, cpr = ldu / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * T * cpr) return;
    const int jc = (int)(idx % cpr);
    const long long row = idx / cpr;
    const int t = (int)(row % T), b = (int)(row / T);
    bf16x8 hi = {}, lo = {};
    if (t > 0) {
        const int l = t - 1, y0 = (l / a.NW) * a.stride, x0 = (l % a.NW) * a.stride;
        const float* img = a.img + (size_t)b * a.C * a.IH * a.IW;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 8 * jc + e;
            float v = 0.f;
            if (j < PD) {
                const int c = j / PP, kh = (j % PP) / a.P, kw = j % a.P;
                v = img[((size_t)c * a.IH + y0 + kh) * a.IW + x0 + kw];
            } else if (j == PD) {
                v = 1.f;
            }
            hi[e] = (bf16_t)v;
            lo[e] = aux_plane(v, hi[e], lo_f16);
        }
    }
    *(bf16x8*)(u_hi + row * ldu + 8 * jc) = hi;
    if (u_lo) *(bf16x8*)(u_lo + row * ldu + 8 * jc) = lo;
}

// ---- gradient with respect to the core input (elementwise.h, launch_patch_input_grad): the reference's core is plain autograd, so
// d response / d image comes for free there (vit.py:66-72, 122-129); gradient-based analyses of a trained model (MEIs, saliency) need it.
// Not on the training path (the cropper samples nearest), so plain fp32 VALU from the fp32 residual-stream gradient.
// dU[r][j] = sum_d g[r][d] W[d][j], W the fp32 master [D][PD]; g = dropout_bwd(gf) (fp32 rows, class-token rows skipped) or gb (bf16 rows)
__global__ __launch_bounds__(256) void patch_du_kernel(const float* __restrict__ gf, const bf16_t* __restrict__ gb, int ldg, DropCfg drop, int T, int cls,
                                                       const float* __restrict__ W, int D, int PD, long long rows, float* __restrict__ du) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * PD) return;
    const long long row = i / PD;
    const int j = (int)(i % PD);
    float acc = 0.f;
    if (!(cls && row % T == 0)) {
        for (int d = 0; d < D; ++d) {
            float g;
            if (gf) {
                g = gf[row * ldg + d];
                if (drop.thresh) g = drop_keep(drop.key, (uint32_t)row, (uint32_t)d, drop.thresh) ? g * drop.inv_keep : 0.f;
            } else {
                g = (float)gb[row * ldg + d];
            }
            acc = fmaf(g, W[(size_t)d * PD + j], acc);
        }
    }
    du[i] = acc;
}
// input gradient of the LayerNorm over the patch (patch modes 2 / 3, vit.py:88, 97): one wave per row,
// out = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dz * gamma, xhat = (u - mean) * rstd
__global__ __launch_bounds__(256) void patch_ln_bwd_rows_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ u, int ldu,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, long long rows, int PD, float* __restrict__ out) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int j = lane; j < PD; j += 64) {
        const float g = dz[row * lddz + j] * gamma[j], xh = (u[row * ldu + j] - mu) * rs;
        s1 += g; s2 += g * xh;
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const float m1 = s1 / (float)PD, m2 = s2 / (float)PD;
    for (int j = lane; j < PD; j += 64) {
        const float g = dz[row * lddz + j] * gamma[j], xh = (u[row * ldu + j] - mu) * rs;
        out[row * PD + j] = rs * (g - m1 - xh * m2);
    }
}
// col2im: dx[b][c][y][x] = sum over the (patch, kh, kw) that read the pixel of dU[row(patch)][(c*P + kh)*P + kw]; zero padding `pad` (CCT conv
// tokenizer), stride, and - spt - the four diagonal half-patch shifts of channel 0 that Shifted Patch Tokenization appends as channels C..C+3
__global__ __launch_bounds__(256) void patch_col2im_kernel(const float* __restrict__ du, int PD, int B, int C, int IH, int IW, int P, int stride, int pad,
                                                           int GH, int GW, int rows_per_image, int row0, int spt, float* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * C * IH * IW) return;
    const int x = (int)(i % IW), y = (int)((i / IW) % IH), c = (int)((i / ((long long)IW * IH)) % C), b = (int)(i / ((long long)IW * IH * C));
    const float* base = du + ((size_t)b * rows_per_image + row0) * PD;
    float acc = 0.f;
    auto gather = [&](int chan, int ys, int xs) {  // ys / xs = y0 + kh / x0 + kw of the patch element that read this pixel
        for (int kh = 0; kh < P; ++kh) {
            const int yy = ys + pad - kh;
            if (yy < 0 || yy % stride != 0 || yy / stride >= GH) continue;
            for (int kw = 0; kw < P; ++kw) {
                const int xx = xs + pad - kw;
                if (xx < 0 || xx % stride != 0 || xx / stride >= GW) continue;
                acc += base[((size_t)(yy / stride) * GW + xx / stride) * PD + (chan * P + kh) * P + kw];
            }
        }
    };
    gather(c, y, x);
    if (spt && c == 0) {
        const int sh = P / 2;
        for (int q = 0; q < 4; ++q) gather(C + q, y - ((q >> 1) ? sh : -sh), x - ((q & 1) ? sh : -sh));
    }
    dx[i] = acc;
}
// adjoint of resize_bilinear_kernel: din[pl][tap] += weight * dout[pl][y][x] (din zeroed by the caller)
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int planes, int IH, int IW, int OH, int OW) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)planes * OH * OW) return;
    const int x = (int)(i % OW), y = (int)((i / OW) % OH);
    const long long pl = i / ((long long)OW * OH);
    const float sy = fmaxf(((float)y + 0.5f) * ((float)IH / (float)OH) - 0.5f, 0.f);
    const float sx = fmaxf(((float)x + 0.5f) * ((float)IW / (float)OW) - 0.5f, 0.f);
    const int y0 = min((int)sy, IH - 1), x0 = min((int)sx, IW - 1);
    const int y1 = min(y0 + 1, IH - 1), x1 = min(x0 + 1, IW - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    float* p = din + pl * IH * IW;
    const float g = dout[i];
    atomicAdd(p + (size_t)y0 * IW + x0, g * (1.f - fx) * (1.f - fy));
    atomicAdd(p + (size_t)y0 * IW + x1, g * fx * (1.f - fy));
    atomicAdd(p + (size_t)y1 * IW + x0, g * (1.f - fx) * fy);
    atomicAdd(p + (size_t)y1 * IW + x1, g * fx * fy);
}

// ---- CCT conv tokenizer (elementwise.h, ConvTokArgs)
__global__ __launch_bounds__(256) void conv_unfold_kernel(ConvTokArgs a, bf16_t* u_hi, bf16_t* u_lo, int lo_f16, int ldu) {
    const int PD = a.C * a.P * a.P, PP = a.P * a.P, cpr = ldu / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long rows = (long long)a.B * a.CH * a.CW;
    if (idx >= rows * cpr) return;
    const int jc = (int)(idx % cpr);
    const long long row = idx / cpr;
    const int cx = (int)(row % a.CW), cy = (int)((row / a.CW) % a.CH), b = (int)(row / ((long long)a.CW * a.CH));
    const int y0 = cy * a.stride - a.pad, x0 = cx * a.stride - a.pad;
    const float* img = a.img + (size_t)b * a.C * a.IH * a.IW;
    bf16x8 hi = {}, lo = {};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = 8 * jc + e;
        float v = 0.f;
        if (j < PD) {
            const int c = j / PP, y = y0 + (j % PP) / a.P, x = x0 + j % a.P;
            if (y >= 0 && y < a.IH && x >= 0 && x < a.IW) v = img[((size_t)c * a.IH + y) * a.IW + x];
        }
        hi[e] = (bf16_t)v;
        lo[e] = aux_plane(v, hi[e], lo_f16);
    }
    *(bf16x8*)(u_hi + row * ldu + 8 * jc) = hi;
    if (u_lo) *(bf16x8*)(u_lo + row * ldu + 8 * jc) = lo;
}
// one thread per (token row, column): max over the 3 x 3 window (stride 2, padding 1: out-of-grid elements do not take part, as
// MaxPool2d pads with -inf) of relu(conv), first maximum in row-major scan order like ATen's max_pool2d
__global__ __launch_bounds__(256) void cct_pool_fwd_kernel(ConvTokArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int L = a.NH * a.NW;
    if (i >= (long long)a.B * L * a.DP) return;
    const int d = (int)(i % a.DP);
    const long long row = i / a.DP;
    const int l = (int)(row % L), b = (int)(row / L);
    float out = 0.f;
    unsigned char arg = 255;
    if (d < a.D) {
        const int ny = l / a.NW, nx = l % a.NW;
        float best = -3.0e38f;
        int bi = -1;
        for (int dy = 0; dy < 3; ++dy) {
            const int cy = 2 * ny - 1 + dy;
            if (cy < 0 || cy >= a.CH) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int cx = 2 * nx - 1 + dx;
                if (cx < 0 || cx >= a.CW) continue;
                const float v = fmaxf(a.conv[(((size_t)b * a.CH + cy) * a.CW + cx) * a.DP + d], 0.f);  // ReLU first (cct.py:88-89)
                if (v > best) { best = v; bi = 3 * dy + dx; }
            }
        }
        // a maximum of 0 means every element of the window is clipped by the ReLU: no gradient passes
        arg = (best > 0.f) ? (unsigned char)bi : (unsigned char)255;
        out = best + (a.pos ? a.pos[(size_t)l * a.D + d] : 0.f);
        if (a.drop.thresh) out = drop_keep(a.drop.key, (uint32_t)row, (uint32_t)d, a.drop.thresh) ? out * a.drop.inv_keep : 0.f;
    }
    a.x[row * a.DP + d] = out;
    a.idx[row * a.DP + d] = arg;
}
// one thread per (conv position, column): the <= 4 pool windows that contain the position
__global__ __launch_bounds__(256) void cct_pool_bwd_kernel(ConvTokArgs a, bf16_t* gd) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long rows = (long long)a.B * a.CH * a.CW;
    if (i >= rows * a.DP) return;
    const int d = (int)(i % a.DP);
    const long long row = i / a.DP;
    const int cx = (int)(row % a.CW), cy = (int)((row / a.CW) % a.CH), b = (int)(row / ((long long)a.CW * a.CH));
    const int L = a.NH * a.NW;
    float g = 0.f;
    if (d < a.D) {
        for (int ny = cy / 2; ny <= (cy + 1) / 2; ++ny) {          // windows with 2 ny - 1 <= cy <= 2 ny + 1
            if (ny >= a.NH) continue;
            const int dy = cy - (2 * ny - 1);
            for (int nx = cx / 2; nx <= (cx + 1) / 2; ++nx) {
                if (nx >= a.NW) continue;
                const int dx = cx - (2 * nx - 1);
                const long long trow = (long long)b * L + ny * a.NW + nx;
                if (a.idx[trow * a.DP + d] != (unsigned char)(3 * dy + dx)) continue;
                float t = a.x[trow * a.DP + d];
                if (a.drop.thresh) t = drop_keep(a.drop.key, (uint32_t)trow, (uint32_t)d, a.drop.thresh) ? t * a.drop.inv_keep : 0.f;
                g += t;
            }
        }
    }
    gd[row * a.DP + d] = (bf16_t)g;
}

// dpos / dcls as patch_bwd_pos_kernel, plus the bf16 copy of the masked gradient (pad columns zero)
__global__ void patch_bwd_pos_cast_kernel(PatchArgs a, bf16_t* gd) {
    const int L = a.NH * a.NW, T = L + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * a.DP) return;
    const int t = idx / a.DP, d = idx % a.DP;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) {
        const int row = b * T + t;
        float g = 0.f;
        if (d < a.D) {
            g = a.x[(size_t)row * a.DP + d];
            if (a.drop.thresh) g = drop_keep(a.drop.key, row, d, a.drop.thresh) ? g * a.drop.inv_keep : 0.f;
        }
        gd[(size_t)row * a.DP + d] = (bf16_t)g;
        s += g;
    }
    if (d < a.D) {
        a.dpos[(size_t)t * a.D + d] += s;
        if (t == 0) a.dcls[d] += s;
    }
}

// fp32 unfold for patch modes 2 / 3; thread = one 16-B chunk (4 columns) of one row
__global__ __launch_bounds__(256) void patch_unfold_f32_kernel(PatchArgs a, int spt, float* u, int ldu) {
    const int PP = a.P * a.P, L = a.NH * a.NW, T = L + 1, cpr = ldu / 4;
    const int CH = spt ? a.C + 4 : a.C, PD = CH * PP, sh = a.P / 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * T * cpr) return;
    const int jc = (int)(idx % cpr);
    const long long row = idx / cpr;
    const int t = (int)(row % T), b = (int)(row / T);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t > 0) {
        const int l = t - 1, y0 = (l / a.NW) * a.stride, x0 = (l % a.NW) * a.stride;
        const float* img = a.img + (size_t)b * a.C * a.IH * a.IW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * jc + e;
            if (j < PD) {
                const int c = j / PP, kh = (j % PP) / a.P, kw = j % a.P;
                int yy = y0 + kh, xx = x0 + kw, ch = c;
                if (spt && c >= a.C) {  // shifted copies of channel 0: left-upper, right-upper, left-bottom, right-bottom
                    const int q = c - a.C;  // padded[..., yy + (q>>1 ? 2sh : 0), xx + (q&1 ? 2sh : 0)] with pad sh on every side
                    yy += ((q >> 1) ? sh : -sh);
                    xx += ((q & 1) ? sh : -sh);
                    ch = 0;
                }
                if (yy >= 0 && yy < a.IH && xx >= 0 && xx < a.IW) v[e] = img[((size_t)ch * a.IH + yy) * a.IW + xx];
            }
        }
    }
    *(f32x4*)(u + row * ldu + 4 * jc) = v;
}

__global__ void patch_bwd_pos_cast_nocls_kernel(PatchArgs a, bf16_t* gd, float* gdf) {
    const int L = a.NH * a.NW, T = L + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * a.DP) return;
    const int t = idx / a.DP, d = idx % a.DP;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) {
        const int row = b * T + t;
        float g = 0.f;
        if (d < a.D) {
            g = a.x[(size_t)row * a.DP + d];
            if (a.drop.thresh) g = drop_keep(a.drop.key, row, d, a.drop.thresh) ? g * a.drop.inv_keep : 0.f;
        }
        s += g;
        const float gp = t == 0 ? 0.f : g;  // the class-token row bypasses the projection
        gd[(size_t)row * a.DP + d] = (bf16_t)gp;
        if (gdf) gdf[(size_t)row * a.DP + d] = gp;
    }
    if (d < a.D) {
        a.dpos[(size_t)t * a.D + d] += s;
        if (t == 0) a.dcls[d] += s;
    }
}

// one wave per row (mode 3 only, not a hot path)
__global__ __launch_bounds__(256) void patch_ln2_finish_kernel(PatchLn2Args a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int t = row % a.T;
    float v[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < a.D) ? a.y[(size_t)row * a.DP + c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / a.D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < a.D) ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / a.D + a.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c >= a.DP) continue;
        float o = 0.f;
        if (c < a.D) {
            o = (t == 0) ? a.cls[c] + a.pos[c] : (v[i] - mean) * rstd * a.gamma[c] + a.beta[c] + a.pos[(size_t)t * a.D + c];
            if (a.drop.thresh) o = drop_keep(a.drop.key, row, c, a.drop.thresh) ? o * a.drop.inv_keep : 0.f;
        }
        a.x0[(size_t)row * a.DP + c] = o;
    }
    if (lane == 0) {
        a.mean[row] = mean;
        a.rstd[row] = rstd;
    }
}

// thread = column, workgroup = 64 rows (not a hot path)
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* dz, int lddz, const float* x, int ldx, const float* mean, const float* rstd,
                                                            int rows, int D, float* dgamma, float* dbeta) {
    const int r0 = blockIdx.x * 64;
    for (int c = threadIdx.x; c < D; c += 256) {
        float sg = 0.f, sb = 0.f;
        for (int r = r0; r < min(r0 + 64, rows); ++r) {
            const float g = dz[(size_t)r * lddz + c];
            sg += g * (x[(size_t)r * ldx + c] - mean[r]) * rstd[r];
            sb += g;
        }
        atomicAdd(&dgamma[c], sg);
        atomicAdd(&dbeta[c], sb);
    }
}

// dWp[d][j] += sum_{b,l} gd[b][1+l][d] * U[b][l][j]; dbp[d] += sum gd.  Workgroup = 64 patches of one
// image, thread = d, 16 j's at a time in registers; results are transposed through LDS so each
// atomic wave-instruction covers 64-B runs instead of 64 different rows.
__global__ __launch_bounds__(256) void patch_bwd_w_kernel(PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int PD = a.C * a.P * a.P, L = a.NH * a.NW, T = L + 1;
    float* sImg = smem;
    float* sT = smem + a.C * a.IH * a.IW;  // [D][16]
    const int b = blockIdx.y, l0 = blockIdx.x * PCHUNK, tid = threadIdx.x;
    const float* img = a.img + (size_t)b * a.C * a.IH * a.IW;
    for (int i = tid; i < a.C * a.IH * a.IW; i += 256) sImg[i] = img[i];
    __syncthreads();
    const int d = tid;
    const bool dval = d < a.D;
    const int np = min(PCHUNK, L - l0);
    const int PP = a.P * a.P;
    for (int j0 = 0; j0 < PD; j0 += 16) {
        float acc[16];
        int off[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            acc[jj] = 0.f;
            const int j = j0 + jj, c = j / PP, kh = (j % PP) / a.P, kw = j % a.P;
            off[jj] = (c * a.IH + kh) * a.IW + kw;
        }
        float bsum = 0.f;
        if (dval) {
            for (int p = 0; p < np; ++p) {
                const int l = l0 + p, row = b * T + 1 + l;
                float g = a.x[(size_t)row * a.DP + d];
                if (a.drop.thresh) g = drop_keep(a.drop.key, row, d, a.drop.thresh) ? g * a.drop.inv_keep : 0.f;
                bsum += g;
                const int base = (l / a.NW) * a.stride * a.IW + (l % a.NW) * a.stride;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) acc[jj] += g * sImg[base + off[jj]];
            }
            if (j0 == 0) atomicAdd(&a.dbias[d], bsum);
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) sT[d * 16 + jj] = acc[jj];
        }
        __syncthreads();
        for (int e = tid; e < a.D * 16; e += 256) atomicAdd(&a.dW[(size_t)(e >> 4) * PD + j0 + (e & 15)], sT[e]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm kernels. Layout for both: a wave works on 8 rows at once, 8 lanes per row; lane j of a row owns the 16-B
// chunks j, j + 8, ... (CPL = DP / 32 of them), so one load instruction reads a 128-B line of each of 8 rows and every
// access is 16 B (fp32) or 8 B (bf16). (One wave per row with lanes striding single elements issued 4-B loads and 2-B
// stores: 2.2-2.4 TB/s.) Row statistics are 3 DPP adds over the 8 lanes of a row.
DEVFN float row8_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    return dpp_add<0x141, 0xF>(v);  // row_half_mirror: lanes i <-> 7 - i of each group of 8
}
typedef __attribute__((ext_vector_type(4))) bf16_t bf16x4_t;

// LayerNorm forward fused with the BehaviorMLP injection x += beta[b] (vit.py:356-359).
constexpr int LNF_RPW = 2;  // 8-row groups per wave: all their loads are issued before the first reduction (bytes in flight)
template <int CPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 7;
    f32x4 v[LNF_RPW][CPL];
    int rows[LNF_RPW];
    bool roks[LNF_RPW];
#pragma unroll
    for (int it = 0; it < LNF_RPW; ++it) {
        const int row = ((blockIdx.x * 4 + wave) * LNF_RPW + it) * 8 + (lane >> 3);
        const bool rok = row < a.rows;
        const int rr = rok ? row : a.rows - 1;
        const int b = rr / a.T;
        rows[it] = row;
        roks[it] = rok;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = 4 * (j + 8 * k);
            v[it][k] = *(const f32x4*)(a.x + (size_t)rr * a.DP + c);
            if (a.inject) {
                v[it][k] += *(const f32x4*)(a.inject + (size_t)b * a.DP + c);
                if (rok) *(f32x4*)(a.xout + (size_t)row * a.DP + c) = v[it][k];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < LNF_RPW; ++it) {
        const int row = rows[it];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (4 * (j + 8 * k) + e >= a.D) v[it][k][e] = 0.f;  // pad columns are zero in x; keep them out of the statistics regardless
                s += v[it][k][e];
            }
        const float mean = row8_sum(s) / a.D;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = (4 * (j + 8 * k) + e < a.D) ? v[it][k][e] - mean : 0.f;
                q += d * d;
            }
        const float rstd = rsqrtf(row8_sum(q) / a.D + a.eps);
        if (!roks[it]) continue;  // the row reductions above need every lane
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = 4 * (j + 8 * k);
            bf16x4_t zh, zl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cc = c + e;
                const float z = (cc < a.D) ? (v[it][k][e] - mean) * rstd * a.gamma[cc] + a.beta[cc] : (cc == a.ones_col ? 1.f : 0.f);
                zh[e] = (bf16_t)z;
                zl[e] = aux_plane(z, zh[e], a.lo_f16);
            }
            *(bf16x4_t*)(a.z + (size_t)row * a.DP + c) = zh;
            if (a.z_lo) *(bf16x4_t*)(a.z_lo + (size_t)row * a.DP + c) = zl;
        }
        if (j == 0) {
            a.mean[row] = mean;
            a.rstd[row] = rstd;
        }
    }
}

// LayerNorm backward + residual add + (optional) token-sum for the injection gradient +
// (optional) dropout-backward/cast of the result for the next branch and its bias gradient.
// Workgroup = LNB_ROWS rows of one image (LNB_ROWS / 32 8-row steps per wave); column partials stay in registers across rows and
// are summed over the 8 row slots of a wave with lane swaps, over the waves through LDS, one atomic per column per WG.
constexpr int LNB_ROWS = 128;  // 64: 29.0 us, 96: 28.8, 128: 25.5, 256: 31.5 (default shape; fewer column-reduction tails vs. workgroups in flight)
DEVFN float rowslot_sum(float v) {  // sum over the 8 row slots (lane bits 3, 4, 5); valid in lanes 0..7
    v = dpp_add<0x128, 0xF>(v);  // row_ror:8 -> lane l += lane l ^ 8
    {
        const unsigned u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // rows 0<->1, 2<->3
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const unsigned u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // halves
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}
template <int CPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs a) {
    __shared__ float sred[4][4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 7, slot = lane >> 3;
    const int b = blockIdx.y, t0 = blockIdx.x * LNB_ROWS;
    const float snext = a.scale_next ? a.scale_next[b] : 1.f;
    f32x4 gam[CPL], adg[CPL], adb[CPL], ainj[CPL], abn[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (j + 8 * k) + e;
            gam[k][e] = (c < a.D) ? a.gamma[c] : 0.f;
            adg[k][e] = adb[k][e] = ainj[k][e] = abn[k][e] = 0.f;
        }
    }
#pragma unroll 1
    for (int it = 0; it < LNB_ROWS / 32; ++it) {
        const int t = t0 + 8 * (wave + 4 * it) + slot;
        if (t0 + 8 * (wave + 4 * it) >= a.T) break;  // wave-uniform
        const bool rok = t < a.T;
        const int row = b * a.T + (rok ? t : a.T - 1);
        const float mean = a.mean[row], rstd = a.rstd[row];
        f32x4 dz[CPL], xh[CPL], gi[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const size_t o = (size_t)row * a.DP + 4 * (j + 8 * k);
            dz[k] = *(const f32x4*)(a.dz + o);
            xh[k] = *(const f32x4*)(a.x + o);
            gi[k] = a.gin ? *(const f32x4*)(a.gin + o) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool cok = 4 * (j + 8 * k) + e < a.D;
                xh[k][e] = cok ? (xh[k][e] - mean) * rstd : 0.f;
                dz[k][e] = cok ? dz[k][e] : 0.f;
                const float dy = dz[k][e] * gam[k][e];
                s1 += dy;
                s2 += dy * xh[k][e];
            }
        s1 = row8_sum(s1) / a.D;
        s2 = row8_sum(s2) / a.D;
        if (!rok) continue;  // the row reductions above need every lane
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = 4 * (j + 8 * k);
            f32x4 go;
            bf16x4_t vb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool cok = c + e < a.D;
                go[e] = cok ? gi[k][e] + rstd * (dz[k][e] * gam[k][e] - s1 - xh[k][e] * s2) : 0.f;
                adg[k][e] += dz[k][e] * xh[k][e];
                adb[k][e] += dz[k][e];
                ainj[k][e] += go[e];
                float v = go[e] * snext;
                if (a.dy_next && a.drop_next.thresh && cok)
                    v = drop_keep(a.drop_next.key, row, c + e, a.drop_next.thresh) ? v * a.drop_next.inv_keep : 0.f;
                vb[e] = (bf16_t)v;
                abn[k][e] += (float)vb[e];
            }
            *(f32x4*)(a.gout + (size_t)row * a.DP + c) = go;
            if (a.dy_next) *(bf16x4_t*)(a.dy_next + (size_t)row * a.DP + c) = vb;
        }
    }
    // EXEC is whole again here (the loop's `continue` only masks lanes inside an iteration)
#pragma unroll
    for (int k = 0; k < CPL; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r0 = rowslot_sum(adg[k][e]), r1 = rowslot_sum(adb[k][e]);
            const float r2 = rowslot_sum(ainj[k][e]), r3 = rowslot_sum(abn[k][e]);
            if (slot == 0) {
                const int c = 4 * (j + 8 * k) + e;
                sred[wave][0][c] = r0;
                sred[wave][1][c] = r1;
                sred[wave][2][c] = r2;
                sred[wave][3][c] = r3;
            }
        }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < a.D) {
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            r0 += sred[w][0][c];
            r1 += sred[w][1][c];
            r2 += sred[w][2][c];
            r3 += sred[w][3][c];
        }
        atomicAdd(&a.dgamma[c], r0);
        atomicAdd(&a.dbeta[c], r1);
        if (a.dinject) atomicAdd(&a.dinject[(size_t)b * a.DP + c], r2);
        if (a.dy_next && a.dbias_next) atomicAdd(&a.dbias_next[c], r3);
    }
}

// Entry of the backward: dy = bf16(dropout_bwd(g) * branch scale), dbias += column sums of dy. Same 8-rows-per-wave,
// 16-B-per-lane layout as the LayerNorm kernels.
template <int CPL>
__global__ __launch_bounds__(256) void drop_cast_kernel(CastArgs a) {
    __shared__ float sred[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 7, slot = lane >> 3;
    f32x4 acc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a workgroup walks several 128-row tiles with its column sums in registers: the bias gradient is ONE atomic per column and workgroup,
    // and with a workgroup per tile (1447 at 112 images) the 1447 atomics on each of 155 addresses were a serial chain longer than the
    // kernel's memory time (107 us for 177 MB)
    const int ntiles = (a.rows + LNB_ROWS - 1) / LNB_ROWS;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * LNB_ROWS;
    f32x4 g[LNB_ROWS / 32][CPL];
#pragma unroll
    for (int it = 0; it < LNB_ROWS / 32; ++it) {
        const int row = min(r0 + 8 * (wave + 4 * it) + slot, a.rows - 1);
#pragma unroll
        for (int k = 0; k < CPL; ++k) g[it][k] = *(const f32x4*)(a.g + (size_t)row * a.DP + 4 * (j + 8 * k));
    }
#pragma unroll
    for (int it = 0; it < LNB_ROWS / 32; ++it) {
        const int row = r0 + 8 * (wave + 4 * it) + slot;
        if (row >= a.rows) continue;
        const float sc = a.scale ? a.scale[row / a.T] : 1.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = 4 * (j + 8 * k);
            bf16x4_t vb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = (c + e < a.D) ? g[it][k][e] * sc : 0.f;
                if (a.drop.thresh && c + e < a.D) v = drop_keep(a.drop.key, row, c + e, a.drop.thresh) ? v * a.drop.inv_keep : 0.f;
                vb[e] = (bf16_t)v;
                acc[k][e] += (float)vb[e];
            }
            *(bf16x4_t*)(a.dy + (size_t)row * a.DP + c) = vb;
        }
    }
    }  // tiles
    // EXEC is whole again here
#pragma unroll
    for (int k = 0; k < CPL; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r = rowslot_sum(acc[k][e]);
            if (slot == 0) sred[wave][4 * (j + 8 * k) + e] = r;
        }
    __syncthreads();
    const int c = threadIdx.x;
    if (a.dbias && c < a.D) atomicAdd(&a.dbias[c], sred[0][c] + sred[1][c] + sred[2][c] + sred[3][c]);
}

// ------------------------------------------------------------------------------------------
// BehaviorMLP (vit.py:157-202): out = tanh(W3 . tanh(W1 . v + b1) + b3). One workgroup per sample.
DEVFN void bmlp_fwd_body(const BmlpArgs& a, float (&sv)[8], float (&sh)[256]) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < a.IN) sv[tid] = a.v[b * a.IN + tid];
    __syncthreads();
    for (int j = tid; j < a.J; j += 256) {
        float s = a.b1 ? a.b1[j] : 0.f;
        for (int i = 0; i < a.IN; ++i) s += a.W1[j * a.IN + i] * sv[i];
        const float h = tanhf(s);
        sh[j] = h;
        a.hid[(size_t)b * a.J + j] = h;
    }
    __syncthreads();
    for (int d = tid; d < a.DP; d += 256) {
        float o = 0.f;
        if (d < a.D) {
            float s = a.b3 ? a.b3[d] : 0.f;
            for (int j = 0; j < a.J; ++j) s += a.W3[(size_t)d * a.J + j] * sh[j];
            o = tanhf(s);
        }
        a.out[(size_t)b * a.DP + d] = o;
    }
}
__global__ __launch_bounds__(256) void bmlp_fwd_kernel(BmlpArgs a) {
    __shared__ float sv[8];
    __shared__ float sh[256];
    bmlp_fwd_body(a, sv, sh);
}
// every block's BehaviorMLP in one launch (they only depend on the behaviour rows): grid (B, blocks) - four ~6-us launches at the head of
// every forward were 22 us of a 14-image rank's 3.7-ms step
__global__ __launch_bounds__(256) void bmlp_fwd_multi_kernel(BmlpBatch bb) {
    __shared__ float sv[8];
    __shared__ float sh[256];
    bmlp_fwd_body(bb.blk[blockIdx.y], sv, sh);
}

// BehaviorMLP backward for all blocks in one launch: grid (BMLP_SPLIT, NB). Every workgroup recomputes the
// tiny intermediates (dpre2 (B,D), dpre1 (B,J)) in LDS and owns a 1/BMLP_SPLIT slice of each gradient, so
// every output element has exactly one writer -> plain += into the gradient arena.
constexpr int BMLP_SPLIT = 64;
__global__ __launch_bounds__(256) void bmlp_bwd_kernel(BmlpBatch bb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const BmlpArgs& a = bb.blk[blockIdx.y];
    float* s2 = smem;               // [B][D]
    float* s1 = smem + a.B * a.D;   // [B][J]
    const int tid = threadIdx.x, part = blockIdx.x, nparts = gridDim.x;
    for (int e = tid; e < a.B * a.D; e += 256) {
        const int b = e / a.D, d = e % a.D;
        const float o = a.out[(size_t)b * a.DP + d];
        s2[e] = a.dout[(size_t)b * a.DP + d] * (1.f - o * o);
    }
    __syncthreads();
    for (int e = part * 256 + tid; e < a.D * a.J; e += 256 * nparts) {
        const int d = e / a.J, j = e % a.J;
        float s = 0.f;
        for (int b = 0; b < a.B; ++b) s += s2[b * a.D + d] * a.hid[(size_t)b * a.J + j];
        a.dW3[e] += s;
    }
    if (a.db3)
        for (int d = part * 256 + tid; d < a.D; d += 256 * nparts) {
            float s = 0.f;
            for (int b = 0; b < a.B; ++b) s += s2[b * a.D + d];
            a.db3[d] += s;
        }
    // d pre1 only for the hidden units this workgroup owns (its slice of dW1 / db1 needs no others): (b, j) elements,
    // four lanes per element splitting the sum over D, combined by two butterfly steps.
    const int jn = (a.J + nparts - 1) / nparts, j0 = part * jn, j1 = min(a.J, j0 + jn);
    const int ne = a.B * max(j1 - j0, 0);
    for (int q0 = 0; q0 < ne * 4; q0 += 256) {  // uniform trip count: the shuffles need every lane
        const int q = q0 + tid, e = q >> 2, sub = q & 3;
        const bool v = e < ne;
        const int b = v ? e / (j1 - j0) : 0, j = v ? j0 + e % (j1 - j0) : 0;
        float s = 0.f;
        if (v)
            for (int d = sub; d < a.D; d += 4) s += s2[b * a.D + d] * a.W3[(size_t)d * a.J + j];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (v && sub == 0) {
            const float h = a.hid[(size_t)b * a.J + j];
            s1[b * jn + (j - j0)] = s * (1.f - h * h);
        }
    }
    __syncthreads();
    for (int e = tid; e < (j1 - j0) * a.IN; e += 256) {
        const int jl = e / a.IN, i = e % a.IN;
        float s = 0.f;
        for (int b = 0; b < a.B; ++b) s += s1[b * jn + jl] * a.v[b * a.IN + i];
        a.dW1[(size_t)(j0 + jl) * a.IN + i] += s;
    }
    if (a.db1)
        for (int jl = tid; jl < j1 - j0; jl += 256) {
            float s = 0.f;
            for (int b = 0; b < a.B; ++b) s += s1[b * jn + jl];
            a.db1[j0 + jl] += s;
        }
}

// ------------------------------------------------------------------------------------------
// Fused L1-sign + AdamW (torch.optim.AdamW semantics, train.py:216-223) over a flat fp32 arena.
DEVFN void adamw_body(const AdamArgs& a, int bx, int nbx) {
    const float step_size = a.lr / a.bc1;
    const float inv_sqrt_bc2 = rsqrtf(a.bc2);
    for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < a.n; i += (long long)nbx * blockDim.x) {
        float p = a.p[i], g = a.g[i];
        if (a.l1 != 0.f) g += a.l1 * ((p > 0.f) ? 1.f : ((p < 0.f) ? -1.f : 0.f));
        if (a.weight_decay != 0.f) p *= 1.f - a.lr * a.weight_decay;
        const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
        const float denom = sqrtf(v) * inv_sqrt_bc2 + a.eps;
        a.p[i] = p - step_size * (m / denom);
        a.m[i] = m;
        a.v[i] = v;
        if (a.zero_grad) a.g[i] = 0.f;
    }
}
__global__ void adamw_kernel(AdamArgs a) { adamw_body(a, blockIdx.x, gridDim.x); }
// several (arena range, learning rate, L1 coefficient) pieces in one launch: the per-mouse arenas of a training step
struct AdamMulti {
    AdamArgs a[ADAM_MAX_RANGES];
    int start[ADAM_MAX_RANGES + 1];
    int n;
};
__global__ void adamw_multi_kernel(AdamMulti m) {
    int u = 0;
    while (u + 1 < m.n && (int)blockIdx.x >= m.start[u + 1]) ++u;
    adamw_body(m.a[u], blockIdx.x - m.start[u], m.start[u + 1] - m.start[u]);
}

__global__ void l1_sum_kernel(const float* __restrict__ p, long long n, float scale, float* out) {
    __shared__ float sred[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += fabsf(p[i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (sred[0] + sred[1] + sred[2] + sred[3]) * scale);
}

__global__ void l1_grad_kernel(const float* __restrict__ p, float* g, long long n, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = p[i];
        g[i] += scale * ((x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f));
    }
}

// the same with the upstream gradient read on the device (an autograd node's grad_output: no host round trip in the backward)
__global__ void l1_grad_dev_kernel(const float* __restrict__ p, float* g, long long n, float scale, const float* __restrict__ gscale) {
    const float sc = scale * gscale[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = p[i];
        g[i] += sc * ((x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f));
    }
}

// ELU1 (models/utils.py:109-118) + Poisson loss (losses.py:153-166) forward and dLoss/du.
DEVFN void elu1_poisson_body(const LossArgs& a, int bx, int nbx, float (&sred)[4]) {
    const float eps = 1.1920928955078125e-07f;
    float ls = 0.f;
    for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < a.n; i += (long long)nbx * blockDim.x) {
        const float u = a.u[i];
        const float yh = (u > 0.f) ? u + 1.0f : expm1f(u) + 1.0f;
        if (a.yhat) a.yhat[i] = yh;
        if (a.y) {
            const float yt = a.y[i] + eps, yp = yh + eps;
            ls += yp - yt * logf(yp);
            if (a.du) {
                const float dyh = (u > 0.f) ? 1.f : expf(u);
                a.du[i] = a.gscale * a.loss_scale * (1.f - yt / yp) * dyh;
            }
        }
    }
    if (a.loss) {
        ls = wave_sum(ls);
        if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = ls;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float v = (sred[0] + sred[1] + sred[2] + sred[3]) * a.loss_scale;
            atomicAdd(a.loss, v);
            if (a.loss_total) atomicAdd(a.loss_total, v);  // the step's loss over all units (no host-side sum of the per-unit scalars)
        }
    }
}
__global__ void elu1_poisson_kernel(LossArgs a) {
    __shared__ float sred[4];
    elu1_poisson_body(a, blockIdx.x, gridDim.x, sred);
}
struct LossMulti {
    LossArgs a[LOSS_MAX_UNITS];
    int start[LOSS_MAX_UNITS + 1];
    int n;
};
__global__ void elu1_poisson_multi_kernel(LossMulti m) {
    __shared__ float sred[4];
    int u = 0;
    while (u + 1 < m.n && (int)blockIdx.x >= m.start[u + 1]) ++u;
    elu1_poisson_body(m.a[u], blockIdx.x - m.start[u], m.start[u + 1] - m.start[u], sred);
}
// PoissonLoss on the model's output (losses.py:141-166 + scale_ds :114-119): the reference criterion's call - add eps to both, sum(y_pred -
// y_true log y_pred), x sqrt(ds_size / batch) - as ONE launch that also leaves dLoss/dy_pred (the reference's graph: add, add, log, mul, sub,
// sum, mul = 7 launches forward and as many backward on a (16, 8000) tensor, in a loop whose small launches the host can barely keep fed).
__global__ void poisson_loss_kernel(const float* __restrict__ yp_, const float* __restrict__ yt_, long long n, float eps, float scale, float* dy, float* loss) {
    __shared__ float sred[4];
    float ls = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float yt = yt_[i] + eps, yp = yp_[i] + eps;
        ls += yp - yt * logf(yp);
        if (dy) dy[i] = scale * (1.f - yt / yp);
    }
    ls = wave_sum(ls);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = ls;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (sred[0] + sred[1] + sred[2] + sred[3]) * scale);
}
// backward of ELU + 1 (models/utils.py:109-118): du = g (u > 0 ? 1 : exp(u) = y) - compare, ones, where, mul as one launch
__global__ void elu1_bwd_kernel(const float* __restrict__ u, const float* __restrict__ y, const float* __restrict__ g, long long n, float* du) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) du[i] = g[i] * (u[i] > 0.f ? 1.f : y[i]);
}
// zero fill (the step's token-gradient buffer and small accumulators): 16 B per lane, tail by bytes
__global__ __launch_bounds__(256) void fill_zero_kernel(char* p, long long bytes) {
    const long long n16 = bytes >> 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) ((u32x4*)p)[i] = u32x4{0u, 0u, 0u, 0u};
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) p[(n16 << 4) + threadIdx.x] = 0;
}

__global__ void dropout_mask_kernel(uint8_t* out, long long rows, long long cols, DropCfg d) {
    const long long total = rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / cols), c = (uint32_t)(i % cols);
        out[i] = (d.thresh == 0 || drop_keep(d.key, r, c, d.thresh)) ? 1 : 0;
    }
}

__global__ void attn_dropout_mask_kernel(uint8_t* out, long long rows, long long T, AttnDrop d) {
    const long long total = rows * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / T), k = (uint32_t)(i % T);
        const uint32_t bh = row / (uint32_t)T, q = row % (uint32_t)T;
        out[i] = (d.thresh16 == 0 || attn_drop_keep(d, bh, (uint32_t)T, q, k)) ? 1 : 0;
    }
}

// ImageCropper's resize (image_cropper.py:96-99,134-135): torchvision Resize(antialias=False) on a tensor = bilinear with
// half-pixel centres (align_corners=False), source index clamped at 0, taps clamped at the edge. HBM-bound: one thread
// per output pixel (144x256 -> 36x64 reads a 2x2 block per pixel).
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* in, float* out, int planes, int IH, int IW, int OH, int OW) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)planes * OH * OW) return;
    const int x = (int)(i % OW), y = (int)((i / OW) % OH);
    const long long pl = i / ((long long)OW * OH);
    const float sy = fmaxf(((float)y + 0.5f) * ((float)IH / (float)OH) - 0.5f, 0.f);
    const float sx = fmaxf(((float)x + 0.5f) * ((float)IW / (float)OW) - 0.5f, 0.f);
    const int y0 = min((int)sy, IH - 1), x0 = min((int)sx, IW - 1);
    const int y1 = min(y0 + 1, IH - 1), x1 = min(x0 + 1, IW - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    const float* p = in + pl * IH * IW;
    const float top = p[(size_t)y0 * IW + x0] * (1.f - fx) + p[(size_t)y0 * IW + x1] * fx;
    const float bot = p[(size_t)y1 * IW + x0] * (1.f - fx) + p[(size_t)y1 * IW + x1] * fx;
    out[i] = top * (1.f - fy) + bot * fy;
}

// The step's inputs in ONE launch: every unit's images resized (or copied when IH == OH and IW == OW) into its slice of the shared batch buffer,
// and its BehaviorMLP input cat(behaviors, pupil_centers) (vit.py:431-432) into the shared (B, nb) buffer - what v1t_resize_bilinear + v1t_concat2
// do per mouse (14 launches of ~7 us per step on the main stream). The same arithmetic per pixel as resize_bilinear_kernel.
struct InputsMulti {
    const float* img[INPUTS_MAX_UNITS];
    const float* beh[INPUTS_MAX_UNITS];
    const float* pup[INPUTS_MAX_UNITS];
    int planes0[INPUTS_MAX_UNITS + 1];  // first output plane (image x channel) of each unit
    int img0[INPUTS_MAX_UNITS + 1];     // first image of each unit
    int n;
};
__global__ __launch_bounds__(256) void inputs_multi_kernel(InputsMulti m, float* out, int IH, int IW, int OH, int OW, float* beh_out, int na, int nb) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long npix = (long long)m.planes0[m.n] * OH * OW;
    if (i < npix) {
        const int x = (int)(i % OW), y = (int)((i / OW) % OH);
        const int pl = (int)(i / ((long long)OW * OH));
        int u = 0;
        while (u + 1 < m.n && pl >= m.planes0[u + 1]) ++u;
        const float* p = m.img[u] + (size_t)(pl - m.planes0[u]) * IH * IW;
        if (IH == OH && IW == OW) {
            out[i] = p[(size_t)y * IW + x];
        } else {
            const float sy = fmaxf(((float)y + 0.5f) * ((float)IH / (float)OH) - 0.5f, 0.f);
            const float sx = fmaxf(((float)x + 0.5f) * ((float)IW / (float)OW) - 0.5f, 0.f);
            const int y0 = min((int)sy, IH - 1), x0 = min((int)sx, IW - 1);
            const int y1 = min(y0 + 1, IH - 1), x1 = min(x0 + 1, IW - 1);
            const float fy = sy - (float)y0, fx = sx - (float)x0;
            const float top = p[(size_t)y0 * IW + x0] * (1.f - fx) + p[(size_t)y0 * IW + x1] * fx;
            const float bot = p[(size_t)y1 * IW + x0] * (1.f - fx) + p[(size_t)y1 * IW + x1] * fx;
            out[i] = top * (1.f - fy) + bot * fy;
        }
        return;
    }
    const long long j = i - npix;
    const int w = na + nb;
    if (!beh_out || j >= (long long)m.img0[m.n] * w) return;
    const int r = (int)(j / w), c = (int)(j % w);
    int u = 0;
    while (u + 1 < m.n && r >= m.img0[u + 1]) ++u;
    const int rl = r - m.img0[u];
    beh_out[j] = c < na ? m.beh[u][(size_t)rl * na + c] : m.pup[u][(size_t)rl * nb + (c - na)];
}

// ImageCropper's crop (image_cropper.py:101-110,126-133): F.grid_sample(mode="nearest", align_corners=True, zeros
// padding) over grid[oy][ox] = (x, y) in [-crop, crop] plus an optional per-image (x, y) shift. Source pixel =
// rint(((g + 1) / 2) * (size - 1)) in fp32 (round half to even, as ATen's nearbyint), outside the image -> 0.
// One thread per output pixel; reads are a strided gather of the (L2-resident) image, writes coalesced.
__global__ __launch_bounds__(256) void crop_nearest_kernel(const float* in, int B, int C, int IH, int IW, const float* grid, const float* shifts,
                                                           float* out, int OH, int OW) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * C * OH * OW) return;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const long long pl = i / ((long long)OW * OH);
    const int b = (int)(pl / C);
    float gx = grid[((size_t)oy * OW + ox) * 2], gy = grid[((size_t)oy * OW + ox) * 2 + 1];
    if (shifts) {
        gx = __fadd_rn(gx, shifts[b * 2]);
        gy = __fadd_rn(gy, shifts[b * 2 + 1]);
    }
    const float fx = rintf(__fmul_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f), (float)(IW - 1)));
    const float fy = rintf(__fmul_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f), (float)(IH - 1)));
    float v = 0.f;
    if (fx >= 0.f && fx <= (float)(IW - 1) && fy >= 0.f && fy <= (float)(IH - 1)) v = in[pl * IH * IW + (size_t)fy * IW + (size_t)fx];
    out[i] = v;
}

inline int ok() { return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH; }
inline int nblocks(long long n, int cap = 2048) { return (int)std::min<long long>((n + 255) / 256, cap); }

}  // namespace

int launch_pack(const float* params, void* shadow, const PackDesc* d_desc, int ndesc, hipStream_t s) {
    if (ndesc <= 0) return V1T_OK;
    hipLaunchKernelGGL(pack_kernel, dim3(64, ndesc), dim3(256), 0, s, params, (char*)shadow, d_desc);
    return ok();
}

static size_t patch_smem_fwd(const PatchArgs& a) { return sizeof(float) * ((size_t)a.C * a.IH * a.IW + (size_t)a.C * a.P * a.P * a.D); }
static size_t patch_smem_bwd(const PatchArgs& a) { return sizeof(float) * ((size_t)a.C * a.IH * a.IW + (size_t)a.D * 16); }

int launch_patch_embed_fwd(const PatchArgs& a, hipStream_t s) {
    const size_t smem = patch_smem_fwd(a);
    if (a.DP > 256 || smem > 160 * 1024) return V1T_ERR_UNSUPPORTED;
    static size_t configured = 0;
    if (smem > configured) {
        if (hipFuncSetAttribute((const void*)patch_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return V1T_ERR_LAUNCH;
        configured = smem;
    }
    const int L = a.NH * a.NW;
    hipLaunchKernelGGL(patch_fwd_kernel, dim3((L + PCHUNK - 1) / PCHUNK, a.B), dim3(256), smem, s, a);
    return ok();
}

int launch_patch_embed_bwd(const PatchArgs& a, hipStream_t s) {
    const int PD = a.C * a.P * a.P, L = a.NH * a.NW, T = L + 1;
    const size_t smem = patch_smem_bwd(a);
    if (a.DP > 256 || PD % 16 != 0 || smem > 160 * 1024) return V1T_ERR_UNSUPPORTED;
    static size_t configured = 0;
    if (smem > configured) {
        if (hipFuncSetAttribute((const void*)patch_bwd_w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return V1T_ERR_LAUNCH;
        configured = smem;
    }
    hipLaunchKernelGGL(patch_bwd_pos_kernel, dim3((T * a.D + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(patch_bwd_w_kernel, dim3((L + PCHUNK - 1) / PCHUNK, a.B), dim3(256), smem, s, a);
    return ok();
}

int launch_ln_fwd(const LnFwdArgs& a, hipStream_t s) {
    if (a.DP > 384 || a.DP % 32 != 0) return V1T_ERR_UNSUPPORTED;
    const dim3 grid((a.rows + 32 * LNF_RPW - 1) / (32 * LNF_RPW));
#define LN_FWD_CASE(N) case N: hipLaunchKernelGGL(ln_fwd_kernel<N>, grid, dim3(256), 0, s, a); break;
    switch (a.DP / 32) {
        LN_FWD_CASE(1) LN_FWD_CASE(2) LN_FWD_CASE(3) LN_FWD_CASE(4) LN_FWD_CASE(5) LN_FWD_CASE(6)
        LN_FWD_CASE(7) LN_FWD_CASE(8) LN_FWD_CASE(9) LN_FWD_CASE(10) LN_FWD_CASE(11) LN_FWD_CASE(12)
        default: return V1T_ERR_UNSUPPORTED;
    }
#undef LN_FWD_CASE
    return ok();
}

int launch_ln_bwd(const LnBwdArgs& a, hipStream_t s) {
    if (a.DP > 256 || a.DP % 32 != 0) return V1T_ERR_UNSUPPORTED;
    const dim3 grid((a.T + LNB_ROWS - 1) / LNB_ROWS, a.B);
    switch (a.DP / 32) {
        case 1: hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(ln_bwd_kernel<3>, grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL(ln_bwd_kernel<5>, grid, dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL(ln_bwd_kernel<6>, grid, dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL(ln_bwd_kernel<7>, grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, dim3(256), 0, s, a); break;
    }
    return ok();
}

int launch_drop_cast(const CastArgs& a, hipStream_t s) {
    if (a.DP > 256 || a.DP % 32) return V1T_ERR_UNSUPPORTED;
    const dim3 grid(std::min((a.rows + LNB_ROWS - 1) / LNB_ROWS, 512));  // <= two workgroups per CU, each over several tiles
    switch (a.DP / 32) {
#define V1T_DC(C) case C: hipLaunchKernelGGL(drop_cast_kernel<C>, grid, dim3(256), 0, s, a); break;
        V1T_DC(1) V1T_DC(2) V1T_DC(3) V1T_DC(4) V1T_DC(5) V1T_DC(6) V1T_DC(7) V1T_DC(8)
#undef V1T_DC
        default: return V1T_ERR_UNSUPPORTED;
    }
    return ok();
}

int launch_bmlp_fwd(const BmlpArgs& a, hipStream_t s) {
    if (a.IN > 8 || a.J > 256) return V1T_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bmlp_fwd_kernel, dim3(a.B), dim3(256), 0, s, a);
    return ok();
}

int launch_bmlp_fwd_multi(const BmlpBatch& bb, hipStream_t s) {
    if (bb.n <= 0) return V1T_OK;
    if (bb.n > BMLP_MAX_BLOCKS || bb.blk[0].IN > 8 || bb.blk[0].J > 256) return V1T_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bmlp_fwd_multi_kernel, dim3(bb.blk[0].B, bb.n), dim3(256), 0, s, bb);
    return ok();
}

int launch_bmlp_bwd(const BmlpBatch& bb0, hipStream_t s) {
    if (bb0.n <= 0) return V1T_OK;
    if (bb0.n > BMLP_MAX_BLOCKS) return V1T_ERR_UNSUPPORTED;
    // the kernel keeps d(pre-activations) of its samples in LDS: batches larger than fit in 64 KB go in several launches
    // (the parameter gradients accumulate with +=, one writer per element)
    const int per = (int)std::max<size_t>(1, (64 * 1024) / (sizeof(float) * (size_t)(bb0.blk[0].D + bb0.blk[0].J)));
    const int B = bb0.blk[0].B;
    for (int b0 = 0; b0 < B; b0 += per) {
        BmlpBatch bb = bb0;
        const int nb = std::min(per, B - b0);
        for (int k = 0; k < bb.n; ++k) {
            BmlpArgs& a = bb.blk[k];
            a.B = nb;
            a.v += (size_t)b0 * a.IN;
            a.hid += (size_t)b0 * a.J;
            a.out += (size_t)b0 * a.DP;
            a.dout += (size_t)b0 * a.DP;
        }
        const size_t smem = sizeof(float) * (size_t)nb * (bb.blk[0].D + bb.blk[0].J);
        hipLaunchKernelGGL(bmlp_bwd_kernel, dim3(BMLP_SPLIT, bb.n), dim3(256), smem, s, bb);
    }
    return ok();
}

int launch_adamw(const AdamArgs& a, hipStream_t s) {
    if (a.n <= 0) return V1T_OK;
    hipLaunchKernelGGL(adamw_kernel, dim3(nblocks(a.n, 4096)), dim3(256), 0, s, a);
    return ok();
}
int launch_l1_sum(const float* p, long long n, float scale, float* out, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(l1_sum_kernel, dim3(nblocks(n, 1024)), dim3(256), 0, s, p, n, scale, out);
    return ok();
}
int launch_l1_grad(const float* p, float* g, long long n, float scale, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(l1_grad_kernel, dim3(nblocks(n, 4096)), dim3(256), 0, s, p, g, n, scale);
    return ok();
}
int launch_l1_grad_dev(const float* p, float* g, long long n, float scale, const float* gscale, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(l1_grad_dev_kernel, dim3(nblocks(n, 4096)), dim3(256), 0, s, p, g, n, scale, gscale);
    return ok();
}
int launch_adamw_multi(const AdamArgs* a, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += ADAM_MAX_RANGES) {
        AdamMulti m{};
        int tot = 0, k = 0;
        for (int i = i0; i < n && i < i0 + ADAM_MAX_RANGES; ++i) {
            if (a[i].n <= 0) continue;
            m.a[k] = a[i];
            m.start[k] = tot;
            tot += nblocks(a[i].n, 4096);
            ++k;
        }
        m.start[k] = tot;
        m.n = k;
        if (tot) hipLaunchKernelGGL(adamw_multi_kernel, dim3(tot), dim3(256), 0, s, m);
    }
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_elu1_poisson_multi(const LossArgs* a, int n, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    if (n > LOSS_MAX_UNITS) return V1T_ERR_ARG;
    LossMulti m{};
    int tot = 0;
    for (int u = 0; u < n; ++u) {
        m.a[u] = a[u];
        m.start[u] = tot;
        // few workgroups per unit: every workgroup ends in two float atomics, and the units' loss scalars and the total share ONE cache line
        // - same-line atomics serialise in the L2 at ~17 ns each (3500 workgroups made this 1 M-element kernel take 93 us)
        tot += a[u].n > 0 ? nblocks(a[u].n, 32) : 0;
    }
    m.start[n] = tot;
    m.n = n;
    if (tot) hipLaunchKernelGGL(elu1_poisson_multi_kernel, dim3(tot), dim3(256), 0, s, m);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_fill_zero(void* p, long long bytes, hipStream_t s) {
    if (bytes <= 0) return V1T_OK;
    const long long n16 = bytes >> 4;
    const int grid = (int)std::min<long long>(std::max<long long>((n16 + 255) / 256, 1), 4096);
    hipLaunchKernelGGL(fill_zero_kernel, dim3(grid), dim3(256), 0, s, (char*)p, bytes);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_elu1_poisson(const LossArgs& a, hipStream_t s) {
    if (a.n <= 0) return V1T_OK;
    hipLaunchKernelGGL(elu1_poisson_kernel, dim3(nblocks(a.n, 1024)), dim3(256), 0, s, a);
    return ok();
}
int launch_poisson_loss(const float* yp, const float* yt, long long n, float eps, float scale, float* dy, float* loss, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(poisson_loss_kernel, dim3(nblocks(n, 1024)), dim3(256), 0, s, yp, yt, n, eps, scale, dy, loss);
    return ok();
}
int launch_elu1_bwd(const float* u, const float* y, const float* g, long long n, float* du, hipStream_t s) {
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(elu1_bwd_kernel, dim3(nblocks(n, 1024)), dim3(256), 0, s, u, y, g, n, du);
    return ok();
}
int launch_attn_dropout_mask(uint8_t* out, long long rows, long long T, AttnDrop d, hipStream_t s) {
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(nblocks(rows * T, 4096)), dim3(256), 0, s, out, rows, T, d);
    return ok();
}
int launch_dropout_mask(uint8_t* out, long long rows, long long cols, DropCfg d, hipStream_t s) {
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(nblocks(rows * cols, 4096)), dim3(256), 0, s, out, rows, cols, d);
    return ok();
}

int launch_patch_unfold(const PatchArgs& a, bf16_t* u_hi, bf16_t* u_lo, int lo_f16, int ldu, hipStream_t s) {
    const long long n = (long long)a.B * (a.NH * a.NW + 1) * (ldu / 8);
    hipLaunchKernelGGL(patch_unfold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, u_hi, u_lo, lo_f16, ldu);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_conv_unfold(const ConvTokArgs& a, bf16_t* u_hi, bf16_t* u_lo, int lo_f16, int ldu, hipStream_t s) {
    const long long n = (long long)a.B * a.CH * a.CW * (ldu / 8);
    hipLaunchKernelGGL(conv_unfold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, u_hi, u_lo, lo_f16, ldu);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_cct_pool_fwd(const ConvTokArgs& a, hipStream_t s) {
    const long long n = (long long)a.B * a.NH * a.NW * a.DP;
    hipLaunchKernelGGL(cct_pool_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_cct_pool_bwd(const ConvTokArgs& a, bf16_t* gd, hipStream_t s) {
    const long long n = (long long)a.B * a.CH * a.CW * a.DP;
    hipLaunchKernelGGL(cct_pool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, gd);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_patch_bwd_pos_cast(const PatchArgs& a, bf16_t* gd, hipStream_t s) {
    const int T = a.NH * a.NW + 1;
    hipLaunchKernelGGL(patch_bwd_pos_cast_kernel, dim3((T * a.DP + 255) / 256), dim3(256), 0, s, a, gd);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_resize_bilinear(const float* in, float* out, int planes, int IH, int IW, int OH, int OW, hipStream_t s) {
    const long long n = (long long)planes * OH * OW;
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, planes, IH, IW, OH, OW);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_resize_bilinear_bwd(const float* dout, float* din, int planes, int IH, int IW, int OH, int OW, hipStream_t s) {
    const long long n = (long long)planes * OH * OW;
    if (n <= 0) return V1T_OK;
    const int rc = launch_fill_zero(din, (long long)planes * IH * IW * 4, s);
    if (rc) return rc;
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dout, din, planes, IH, IW, OH, OW);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_patch_du(const float* gf, const bf16_t* gb, int ldg, DropCfg drop, int T, int cls, const float* W, int D, int PD, long long rows, float* du,
                    hipStream_t s) {
    const long long n = rows * PD;
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(patch_du_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gf, gb, ldg, drop, T, cls, W, D, PD, rows, du);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_patch_ln_bwd_rows(const float* dz, int lddz, const float* u, int ldu, const float* mean, const float* rstd, const float* gamma, long long rows,
                             int PD, float* out, hipStream_t s) {
    if (rows <= 0) return V1T_OK;
    hipLaunchKernelGGL(patch_ln_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dz, lddz, u, ldu, mean, rstd, gamma, rows, PD, out);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_patch_col2im(const float* du, int PD, int B, int C, int IH, int IW, int P, int stride, int pad, int GH, int GW, int rows_per_image, int row0,
                        int spt, float* dx, hipStream_t s) {
    const long long n = (long long)B * C * IH * IW;
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(patch_col2im_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, du, PD, B, C, IH, IW, P, stride, pad, GH, GW, rows_per_image,
                       row0, spt, dx);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_inputs_multi(const float* const* img, const float* const* beh, const float* const* pup, const int* n_images, int n, int C, int IH, int IW,
                        float* out, int OH, int OW, float* beh_out, int na, int nb, hipStream_t s) {
    int i0 = 0;
    for (int c0 = 0; c0 < n; c0 += INPUTS_MAX_UNITS) {
        const int cn = std::min(INPUTS_MAX_UNITS, n - c0);
        InputsMulti m{};
        int pl = 0, im = 0;
        for (int u = 0; u < cn; ++u) {
            m.img[u] = img[c0 + u]; m.beh[u] = beh ? beh[c0 + u] : nullptr; m.pup[u] = pup ? pup[c0 + u] : nullptr;
            m.planes0[u] = pl; m.img0[u] = im;
            pl += n_images[c0 + u] * C; im += n_images[c0 + u];
        }
        m.planes0[cn] = pl; m.img0[cn] = im; m.n = cn;
        const long long tot = (long long)pl * OH * OW + (beh_out ? (long long)im * (na + nb) : 0);
        if (tot > 0)
            hipLaunchKernelGGL(inputs_multi_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, m, out + (size_t)i0 * C * OH * OW, IH, IW, OH, OW,
                               beh_out ? beh_out + (size_t)i0 * (na + nb) : nullptr, na, nb);
        i0 += im;
    }
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_crop_nearest(const float* in, int B, int C, int IH, int IW, const float* grid, const float* shifts, float* out, int OH, int OW,
                        hipStream_t s) {
    const long long n = (long long)B * C * OH * OW;
    if (n <= 0) return V1T_OK;
    hipLaunchKernelGGL(crop_nearest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, B, C, IH, IW, grid, shifts, out, OH, OW);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_patch_unfold_f32(const PatchArgs& a, int spt, float* u, int ldu, hipStream_t s) {
    const long long n = (long long)a.B * (a.NH * a.NW + 1) * (ldu / 4);
    hipLaunchKernelGGL(patch_unfold_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, spt, u, ldu);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_patch_bwd_pos_cast_nocls(const PatchArgs& a, bf16_t* gd, float* gdf, hipStream_t s) {
    const int T = a.NH * a.NW + 1;
    hipLaunchKernelGGL(patch_bwd_pos_cast_nocls_kernel, dim3((T * a.DP + 255) / 256), dim3(256), 0, s, a, gd, gdf);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_patch_ln2_finish(const PatchLn2Args& a, hipStream_t s) {
    if (a.DP > 256) return V1T_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(patch_ln2_finish_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
int launch_ln_param_grad(const float* dz, int lddz, const float* x, int ldx, const float* mean, const float* rstd, int rows, int D,
                         float* dgamma, float* dbeta, hipStream_t s) {
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3((rows + 63) / 64), dim3(256), 0, s, dz, lddz, x, ldx, mean, rstd, rows, D, dgamma, dbeta);
    return hipGetLastError() == hipSuccess ? V1T_OK : V1T_ERR_LAUNCH;
}
