/*
 * v1t_amd — C-ABI of the MI355X-native V1T hot path (libv1t_amd.so, gfx950).
 *
 * The reference (bryanlimy/V1T) has no FFI: its hot path sits behind two Python class
 * registries, `register("vit")` (src/v1t/models/core/core.py:8-16, used at core/vit.py:365) and
 * `register("gaussian2d")` (src/v1t/models/readout/readout.py:10-18, used at
 * readout/gaussian2d.py:13). The entry points below are what a ctypes binding behind those two
 * registry entries calls (INTEGRATION.md shows the stub); each cites the reference code it replaces.
 *
 * Conventions: every pointer is a DEVICE pointer borrowed for the call, except v1t_vit_config, name/shape out-parameters and the TABLES of the
 * multi-unit entry points (v1t_tail_unit / v1t_adam_range arrays, the pointer and count arrays of v1t_inputs_multi), which are host memory read
 * during the call (their contents travel in the kernel arguments; the device pointers inside them are borrowed like any other). No device
 * memory is allocated inside a launch function (workspaces are sized by the *_bytes queries and passed in; the measurement aid
 * v1t_mfma_peak_probe is the one exception); all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*);
 * v1t_vit_backward(_events) additionally owns one internal stream per plan (created by its first small launch, destroyed with the plan) on which
 * the weight-gradient GEMMs of launches below 262 144 rows run, joined to `stream` by events before the call returns its last launch.
 * Return value 0 = ok, negative = error code below (the Python shim raises RuntimeError, which the reference's OOM probe utils/utils.py:460
 * relies on). Thread-safe for distinct handles; calls on ONE handle must be serialised by the caller (the plan owns the second stream and
 * the events of its backward, created on first use on the device that is current then: two backward calls on the same plan from two threads
 * or two streams would race on them).
 */
#ifndef V1T_AMD_H
#define V1T_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define V1T_OK 0
#define V1T_ERR_ARG -1
#define V1T_ERR_UNSUPPORTED -2
#define V1T_ERR_LAUNCH -3
#define V1T_ERR_WORKSPACE -4

int v1t_abi_version(void);
const char* v1t_error_string(int code);

/* ---------------------------------------------------------------- ViT core (core/vit.py:365-436) */
typedef struct v1t_vit_config {
    int in_channels, in_h, in_w;      /* core input shape = ImageCropper.output_shape (model.py:78) */
    int patch_size, patch_stride, patch_mode; /* vit.py:384-391; patch_mode 0 (unfold+linear), 1 (conv), 2 (SPT), 3 (dual PatchNorm): all native */
    int emb_dim, num_heads, mlp_dim, num_blocks; /* vit.py:392-398; head dim = emb_dim (vit.py:218) */
    int behavior_mode;                /* 0 none, 2/3 shared B-MLP, 4 per-mouse B-MLP (vit.py:157-202) */
    int num_mice;                     /* B-MLP instances when behavior_mode == 4, else ignored */
    int use_lsa, use_bias;            /* vit.py:235-251; --disable_bias => use_bias = 0 */
    float p_dropout, t_dropout;       /* vit.py:390,398 */
    float ln_eps;                     /* nn.LayerNorm default 1e-5 */
    /* core_kind 1 = the CCT core (core/cct.py:247-317) on the same plan: conv tokenizer Conv2d(k = patch_size, stride, padding =
     * conv_pad, no bias) -> ReLU -> MaxPool2d(3, 2, 1) -> tokens (no class token) + fixed position table (pos_mode 1: the "sine"
     * buffer, cct.py:17-27; 0: none) -> dropout (cct.py:30-104); attention with qkv width 3 * emb_dim / num_heads and head dim
     * emb_dim / num_heads^2 (cct.py:107-143; the scale buffer holds (emb_dim / num_heads)^-0.5); biases always on; state-dict
     * names of cct.py ("tokenizer.conv2d.weight", "transformer.blocks.<k>.mha.qkv.weight", "...mlp.1.weight", "...b_mlp...").
     * patch_mode / use_lsa / use_bias are ignored. 0 = the ViT core above. */
    int core_kind, conv_pad, pos_mode;
} v1t_vit_config;

typedef struct v1t_vit v1t_vit;       /* opaque plan: dims, arena/shadow/workspace layout, pack table */

int v1t_vit_create(const v1t_vit_config* cfg, v1t_vit** out);
void v1t_vit_destroy(v1t_vit* h);

/* Parameter arena: ONE flat fp32 buffer; tensors keep the reference's natural shapes and state-dict
 * names so checkpoints stay compatible. Entries [0, param_count) are
 * nn.Parameters (what ViTCore.regularizer vit.py:419-421 and the optimizer see); entries after that
 * are buffers (`mha.scale` without LSA). `name` is relative to the core module
 * ("transformer.blocks.0.mha.to_qkv.weight"); per-mouse B-MLPs use "models.@<i>". */
long long v1t_vit_arena_floats(const v1t_vit* h);
long long v1t_vit_param_floats(const v1t_vit* h);
int v1t_vit_num_tensors(const v1t_vit* h);
int v1t_vit_tensor_info(const v1t_vit* h, int idx, char* name, int name_cap, long long* offset,
                        int* ndim, long long* shape4, int* is_param);
int v1t_vit_tokens(const v1t_vit* h);      /* T = patches + class tokens */
int v1t_vit_cls_tokens(const v1t_vit* h);  /* 1 (ViT: token 0 is the class token) or 0 (CCT) */
int v1t_vit_padded_dim(const v1t_vit* h);  /* DP: row stride of the token-major output */
int v1t_vit_grid_h(const v1t_vit* h);      /* latent (h, w) = find_shape(patches), vit.py:411-417 / cct.py:293-299 */
int v1t_vit_grid_w(const v1t_vit* h);

long long v1t_vit_shadow_bytes(const v1t_vit* h);                 /* bf16 padded/transposed weight shadow */
long long v1t_vit_workspace_bytes(const v1t_vit* h, int batch, int save_for_backward);
long long v1t_vit_scratch_bytes(const v1t_vit* h, int batch);     /* backward-only scratch */
/* byte offset of a named intermediate inside the forward workspace (tests / rollout): names
 * "x0","xa","xm","z1","qkv","o","lse2","z2","hpre","hact","beta"; returns <0 if unknown. "o" / "hact" are the bf16 planes of
 * the attention output / GELU output: zero-sized (never written) in plans whose backward reads the forward's fp16 planes */
long long v1t_vit_workspace_offset(const v1t_vit* h, int batch, int save_for_backward, const char* name, int block);

/* refresh the shadow from the fp32 arena (after an optimizer step / load_state_dict) */
int v1t_vit_pack(const v1t_vit* h, const float* arena, void* shadow, void* stream);

/* ViTCore.forward vit.py:423-436. images (B,C,H,W) fp32; behaviors (B,IN) fp32 already
 * concatenated with pupil centers for behavior_mode 3/4 (vit.py:431-432), NULL for mode 0.
 * out: token-major residual stream (B, T, DP) fp32, CLS at t = 0, columns >= emb_dim are 0; the
 * reference's (B, C', h, w) output is the strided view out[:, 1:, :emb_dim] (vit.py:434-435).
 * training != 0 enables the three dropouts with the counter-based mask keyed by `seed`.
 * save_for_backward: 1 = keep everything v1t_vit_backward reads (one workspace region per block); 0 = inference, the blocks share one
 * region; 2 = inference that keeps every block's qkv and log-sum-exp (v1t_rollout_headmax / v1t_attention_probs). 0 and 2 do not write
 * the planes only the backward reads (LayerNorm outputs and statistics, gelu'); workspace sizes: v1t_vit_workspace_bytes(h, batch, mode). */
int v1t_vit_forward(const v1t_vit* h, const float* arena, const void* shadow, const float* images,
                    const float* behaviors, int mouse_idx, int batch, void* workspace,
                    long long workspace_bytes, int save_for_backward, int training, uint64_t seed,
                    const float* path_scale, float* out, void* stream);
/* path_scale: stochastic depth (DropPath, models/utils.py:121-141; vit.py:360-361) - NULL (= identity: eval, or
 * drop_path 0) or device floats [num_blocks][2][batch]: the factor floor(keep + U) / keep of sample b for the
 * {attention, MLP} branch of each block, drawn by the caller (one U[0,1) per sample and branch, as the reference does).
 * The backward must be given the same array. */

/* Backward of the above: gout (B,T,DP) fp32 = dL/d out; accumulates (+=) into grads, a flat fp32
 * buffer with the arena's layout (gradient accumulation over mice, train.py:97-111, is free). */
int v1t_vit_backward(const v1t_vit* h, const float* arena, const void* shadow, const float* images,
                     const float* behaviors, int mouse_idx, int batch, const void* workspace,
                     void* scratch, long long scratch_bytes, int training, uint64_t seed,
                     const float* path_scale, const float* gout, float* grads, void* stream);
/* Same, recording hipEvent block_done[k] (num_blocks entries, each may be NULL) on `stream` as soon as every gradient of
 * block k's attention / MLP parameters (state-dict keys core.transformer.blocks.k.{mha,mlp}.*) is complete - blocks finish
 * last to first - so that the data-parallel exchange of that block's slice of the gradient arena (reference train.py:97-111
 * accumulates the same sums on one device) can start while the earlier blocks are still in their backward. The BehaviorMLP
 * and patch-embedding gradients are complete when the call's work is. */
int v1t_vit_backward_events(const v1t_vit* h, const float* arena, const void* shadow, const float* images,
                            const float* behaviors, int mouse_idx, int batch, const void* workspace,
                            void* scratch, long long scratch_bytes, int training, uint64_t seed,
                            const float* path_scale, const float* gout, float* grads, void* const* block_done,
                            void* stream);

/* Same, and - when dimages is not NULL - also writes (=, not +=) the gradient with respect to the core input, dimages (B, C, H, W) fp32:
 * the reference's core is plain autograd, so d response / d image comes with `.backward()` there (Unfold + Linear vit.py:66-72, Conv2d
 * :73-82, the patch LayerNorms :83-100, cls / pos / dropout :122-129; conv tokenizer cct.py:30-104) and gradient-based analyses of a
 * trained model (MEIs, saliency) rely on it. d x0 . W in fp32 against the fp32 master weight, [input gradient of the LayerNorm over the
 * patch], col2im. scratch must then hold v1t_vit_scratch_bytes_input(h, batch) bytes. */
long long v1t_vit_scratch_bytes_input(const v1t_vit* h, int batch);
int v1t_vit_backward_input(const v1t_vit* h, const float* arena, const void* shadow, const float* images,
                           const float* behaviors, int mouse_idx, int batch, const void* workspace,
                           void* scratch, long long scratch_bytes, int training, uint64_t seed,
                           const float* path_scale, const float* gout, float* grads, void* const* block_done,
                           float* dimages, void* stream);

/* 1 when v1t_vit_backward* of `batch` images hands its weight-gradient GEMMs (dW = dY^T X of the four linear layers of a block) to a second
 * stream that runs them beside the next block's dX / attention kernels (launches under 262 144 token rows), 0 when everything runs on
 * `stream`. The gradients are complete on `stream` when the call's work is, either way. */
int v1t_vit_backward_second_stream(const v1t_vit* h, int batch);

/* keep-mask of one dropout stream, for replaying the exact mask in a CPU check.
 * stream ids: 8*block + {0: attention P (rows B*H*T, cols T), 1: proj out, 2: fc1 out, 3: fc2 out}
 * (rows B*T, cols = feature index), 0xFFFF: patch embedding. out: rows*cols bytes (1 = keep). */
int v1t_dropout_mask(uint64_t seed, uint32_t stream_id, float p, long long rows, long long cols,
                     uint8_t* out, void* stream);
/* The attention-P dropout (vit.py:263) is evaluated B*H*T*T times inside three MFMA kernels; its mask uses one 32-bit hash per
 * 2x2 (query,key) block with byte decisions against a threshold that is dithered per 32x32 tile (thresh8 + 1 in a fraction
 * frac8/256 of the tiles, thresh8 in the others; csrc/common.h), so an element's drop probability is round(65536 p)/65536
 * (0.2544 -> 0.25439453, 2e-5 relative; p < 2^-17 -> 0 = no dropout at this site, p > 65535/65536 -> 65535/65536) and 1/(1-rate)
 * uses that rate. (Rounds 1-4: a fixed byte threshold, round(256 p)/256 = 0.25390625.) This returns the effective rate. */
float v1t_attention_dropout_rate(float p);

/* ------------------------------------------------ Gaussian2d readout (readout/gaussian2d.py:237-278) */
/* z: core map, element (b, cell, c) at z[b*zsb + cell*zsc + c] (cell = y*W + x, channel stride 1);
 * grid (B,N,2) sample positions (x,y) incl. shifts (gaussian2d.py:265-268); feat (N,FS) neuron-major
 * feature weights (the (1,C,1,N) parameter viewed transposed); bias (N) or NULL; out (B,N). */
int v1t_gaussian2d_forward(const float* z, long long zsb, long long zsc, int B, int C, int H, int W,
                           int N, const float* grid, const float* feat, int FS, const float* bias,
                           float* out, void* stream);
/* gout (B,N). dz (same addressing as z, += via atomics; must be zero-initialised by the caller),
 * dgrid (B,N,2) overwritten, dfeat (N,FS) +=, dbias (N) +=. Any of dz/dgrid/dfeat/dbias may be NULL. */
int v1t_gaussian2d_backward(const float* z, long long zsb, long long zsc, int B, int C, int H, int W,
                            int N, const float* grid, const float* feat, int FS, const float* gout,
                            float* dz, long long dzsb, long long dzsc, float* dgrid, float* dfeat,
                            float* dbias, void* stream);
/* Same with `ws` = v1t_gaussian2d_backward_ws_bytes() bytes of scratch: the taps are counting-sorted by cell (per image)
 * into the scratch and dz is summed per run of equal cells, one atomic row per run (~20 MB) instead of 4*C*N*B
 * float atomics (317 MB at N = 8000, B = 16). The form the training path uses. */
long long v1t_gaussian2d_backward_ws_bytes(int B, int H, int W, int N);
int v1t_gaussian2d_backward_ws(const float* z, long long zsb, long long zsc, int B, int C, int H, int W,
                               int N, const float* grid, const float* feat, int FS, const float* gout,
                               float* dz, long long dzsb, long long dzsc, float* dgrid, float* dfeat,
                               float* dbias, void* ws, long long ws_bytes, void* stream);
/* The three kernels of the form above one by one, `parts` = bit mask: 1 sort the taps (needs grid only - the training step runs
 * it while the core is still in its forward), 2 parameter gradients (dgrid, dfeat, dbias), 4 gather dz from the sorted taps
 * (needs part 1's scratch and gout; z / dz addressing as above). 7 == v1t_gaussian2d_backward_ws. */
int v1t_gaussian2d_backward_parts(const float* z, long long zsb, long long zsc, int B, int C, int H, int W,
                                  int N, const float* grid, const float* feat, int FS, const float* gout,
                                  float* dz, long long dzsb, long long dzsc, float* dgrid, float* dfeat,
                                  float* dbias, void* ws, long long ws_bytes, int parts, void* stream);

/* Readout sample positions (gaussian2d.py:188-235, 265-268): mu from the grid predictor
 * (Linear(gd,30) -> ELU -> Linear(30,2) -> tanh on the normalised cortical coordinates src (N,gd); gd == 0:
 * free parameter mu_free (N,2)), grid[b][n] = clamp(sigma_n . eps[b][n] + mu_n, -1, 1) + shift[b].
 * eps NULL = eval (grid = clamp(mu)); shift NULL = no shifter. */
int v1t_readout_grid_forward(int B, int N, int gd, const float* src, const float* W0, const float* b0,
                             const float* W2, const float* b2, const float* mu_free, const float* sigma,
                             const float* eps, const float* shift, float* grid, void* stream);
/* dgrid (B,N,2) -> dW0 (30,gd), db0 (30), dW2 (2,30), db2 (2) (+= via atomics, zero them first),
 * dmu_free (N,2) / dsigma (N,2,2) overwritten, dshift (B,2) += (zero it first). NULL outputs are skipped. */
int v1t_readout_grid_backward(int B, int N, int gd, const float* src, const float* W0, const float* b0,
                              const float* W2, const float* b2, const float* mu_free, const float* sigma,
                              const float* eps, const float* dgrid, float* dW0, float* db0, float* dW2,
                              float* db2, float* dmu_free, float* dsigma, float* dshift, void* stream);
/* Same with a caller-owned workspace (v1t_readout_grid_backward_ws_bytes): per-workgroup partial sums + a second
 * reduction kernel instead of float atomics on the few cache lines of dW0/db0/dW2/db2/dshift (deterministic, 4x faster).
 * The outputs are still accumulated into (+=). ws NULL = the atomics path above. */
long long v1t_readout_grid_backward_ws_bytes(int B, int N);
int v1t_readout_grid_backward_ws(int B, int N, int gd, const float* src, const float* W0, const float* b0,
                                 const float* W2, const float* b2, const float* mu_free, const float* sigma,
                                 const float* eps, const float* dgrid, float* dW0, float* db0, float* dW2,
                                 float* db2, float* dmu_free, float* dsigma, float* dshift, void* ws,
                                 long long ws_bytes, void* stream);
/* The readout's position noise (gaussian2d.py:219-221: `norm = mu.new(...).normal_()`): n standard normal deviates from
 * Philox-4x32-10 keyed by the full 64-bit seed with counter (index, stream_id): every (seed, stream_id) is its own stream of
 * independent 128-bit blocks, Box-Muller pairs from different words; stateless (a step's draw can be replayed), one launch. */
int v1t_normal_fill(float* out, long long n, uint64_t seed, uint32_t stream_id, void* stream);
/* out[r][0..na) = a[r][:], out[r][na..na+nb) = b[r][:], out rows ldo floats apart: the BehaviorMLP input
 * torch.cat((behaviors, pupil_centers), dim=-1) (vit.py:431-432) written straight into a shared batch buffer. */
int v1t_concat2(const float* a, int na, const float* b, int nb, int rows, float* out, int ldo, void* stream);
/* CoreShifter MLP 2->5->5->2, tanh after every layer (core_shifter.py:24-40; model.py:86-92) */
int v1t_core_shifter_forward(int B, const float* pupil, const float* W0, const float* b0, const float* W2,
                             const float* b2, const float* W4, const float* b4, float* shift, void* stream);
/* the six parameter gradients are ACCUMULATED (+=): hand in the gradient arena views, or zeroed buffers */
int v1t_core_shifter_backward(int B, const float* pupil, const float* W0, const float* b0, const float* W2,
                              const float* b2, const float* W4, const float* b4, const float* dshift,
                              float* dW0, float* db0, float* dW2, float* db2, float* dW4, float* db4,
                              void* stream);

/* ImageCropper crop (image_cropper.py:101-110,126-133): F.grid_sample(mode="nearest", align_corners=True) of
 * in[B][C][IH][IW] over grid[OH][OW][2] = (x, y) in [-1, 1] (the module's `grid` buffer), moved per image by
 * shifts[B][2] (the ImageShifter output, image_cropper.py:41-47; NULL = none). Outside the image -> 0. */
int v1t_crop_nearest(const float* in, int B, int C, int IH, int IW, const float* grid, const float* shifts, float* out,
                     int OH, int OW, void* stream);

/* ImageCropper resize (image_cropper.py:96-99,134-135): torchvision Resize(antialias=False) = bilinear, half-pixel
 * centres, on `planes` = B*C images of IH x IW -> OH x OW (144x256 -> 36x64 for Sensorium). */
int v1t_resize_bilinear(const float* in, int planes, int IH, int IW, float* out, int OH, int OW, void* stream);
/* its adjoint (autograd of F.interpolate in the reference): din (planes, IH, IW) = sum over the output pixels of weight * dout (written, not +=) */
int v1t_resize_bilinear_backward(const float* dout, int planes, int IH, int IW, float* din, int OH, int OW, void* stream);

/* ELU1 (models/utils.py:109-118) + PoissonLoss (losses.py:153-166, scale_ds :114-119).
 * yhat/du/loss may be NULL; y may be NULL (inference: only yhat). loss is += (zero it first). */
int v1t_elu1_poisson(const float* u, const float* y, long long n, float loss_scale, float gscale,
                     float* yhat, float* du, float* loss, void* stream);

/* PoissonLoss as the reference's criterion calls it, on the model's output (losses.py:141-166: add eps to targets and predictions,
 * sum(y_pred - y_true log y_pred); scale_ds :114-119 = loss_scale): loss += the scaled sum (zero it first); dy (may be NULL) =
 * dLoss/dy_pred = loss_scale (1 - (y_true + eps) / (y_pred + eps)). One launch for the criterion's seven. */
int v1t_poisson_loss(const float* y_pred, const float* y_true, long long n, float eps, float loss_scale, float* dy, float* loss,
                     void* stream);
/* Backward of ELU + 1 (models/utils.py:109-118) from its input u and output y: du = g * (u > 0 ? 1 : y). */
int v1t_elu1_backward(const float* u, const float* y, const float* g, long long n, float* du, void* stream);

/* ------------------------------------------------------------------ input pipeline (packed per-mouse store in HBM) */
/* MiceDataset.__getitem__ (data.py:419-434) for a whole batch: gather trials index[B] from a packed [trials][E] array
 * (fp32, or uint8 when src_u8) and apply the dataset transform (data.py:341-403)
 *   out[b][e] = ((src[index[b]][e] - sub[e % nsub]) / div[e % ndiv]) * mul[e % nmul]      (NULL array = step skipped)
 * gray_c > 1: out[b] has E / gray_c elements, the mean over the gray_c channel planes (color2gray, data.py:338-339). */
int v1t_gather_transform(const void* src, int src_u8, const int* index, int B, long long E, const float* sub,
                         long long nsub, const float* div, long long ndiv, const float* mul, long long nmul,
                         int gray_c, float* out, void* stream);

/* ------------------------------------------------------- validation / evaluation metrics (streaming, fp64 moments) */
/* compute_metrics (train.py:29-39) without stacking predictions on the host (train.py:24-25,186): fold one micro-batch
 * pred/target[B][N] into acc[5][N] += (sum p, sum t, sum p^2, sum t^2, sum p*t) per neuron and
 * scal[0] += msse = sum (t-p)^2 (losses.py:25-29), scal[1] += poisson_loss = sum (p - t*log(p+eps)) (losses.py:32-40).
 * acc / scal are caller-owned, zeroed before the first call. */
int v1t_metrics_accumulate(const float* pred, const float* target, int B, int N, float eps, double* acc, double* scal,
                           void* stream);
/* losses.correlation(y_pred, y_true, dim=0) (losses.py:43-58, eps 1e-8) from the moments of `count` trials:
 * corr[N] (may be NULL) and *mean_out += mean over neurons (may be NULL). */
int v1t_metrics_correlation(const double* acc, long long count, int N, float eps, float* corr, double* mean_out,
                            void* stream);
/* Metrics.split_responses (metrics.py:41-58) as streaming sums: group[b] in [0, G) is the image index of trial b;
 * gacc[3][G][N] += (sum t, sum t^2, sum p) per (image, neuron), sqerr[N] += sum (t-p)^2. */
int v1t_metrics_group_accumulate(const float* pred, const float* target, const int* group, int B, int N, int G,
                                 double* gacc, double* sqerr, void* stream);
/* Metrics.correlation_to_average (metrics.py:77-93) and Metrics._fev (metrics.py:95-127) per neuron from those sums
 * and gcount[G] trials per image; any of the three outputs may be NULL. */
int v1t_metrics_group_finalize(const double* gacc, const int* gcount, const double* sqerr, int G, int N, float eps,
                               float* corr_avg, float* fev, float* feve, void* stream);

/* ------------------------------------------------------------------ optimiser-side HBM kernels */
/* torch.optim.AdamW step (train.py:216-223) over a flat arena, with the L1 regulariser's gradient
 * l1 * sign(p) folded in (vit.py:419-421, gaussian2d.py:99-100) and optional fused zero_grad. */
int v1t_adamw_step(float* p, float* g, float* m, float* v, long long n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float l1, int zero_grad,
                   void* stream);
int v1t_l1_sum(const float* p, long long n, float scale, float* out_accum, void* stream);   /* out += scale*sum|p| */
int v1t_l1_grad(const float* p, float* g, long long n, float scale, void* stream);          /* g += scale*sign(p) */
/* the same with the upstream gradient of the regulariser's autograd node (train.py:71: (micro / batch) * model.regularizer) read on the
 * device: g += scale * gscale[0] * sign(p) - the backward of the L1 term without a host round trip */
int v1t_l1_grad_dev(const float* p, float* g, long long n, float scale, const float* gscale, void* stream);

/* ------------------------------------------------------------- building blocks (parity tests) */
/* C[M][N] (bf16 or fp32) = A[M][K] . B[N][K]^T, bf16 inputs, fp32 accumulate */
int v1t_gemm_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* C, int ldc,
                int out_f32, void* stream);
/* dW[n][k] (fp32, +=) = sum_m Y[m][n] X[m][k], bf16 inputs */
int v1t_gemm_tn(const void* Y, int ldy, const void* X, int ldx, int M, int NY, int NX, float* dW,
                int ldw, int m_chunk, void* stream);
/* same, with the per-chunk partial tiles staged in `slab` (v1t_gemm_tn_slab_bytes bytes of scratch; 0 bytes =
 * this shape has no slab path) and summed by a reduce kernel instead of fp32 atomics: the form the ViT
 * backward uses, deterministic for a given m_chunk */
long long v1t_gemm_tn_slab_bytes(int M, int NY, int NX, int m_chunk);
int v1t_gemm_tn_slab(const void* Y, int ldy, const void* X, int ldx, int M, int NY, int NX, float* dW,
                     int ldw, int m_chunk, float* slab, long long slab_bytes, void* stream);
/* qkv (B*T, 3*H*DP) bf16 -> o (B*T, H*DP) bf16, lse2 (B,H,T) */
int v1t_attention_forward(const void* qkv, int B, int H, int T, int DP, const float* scale,
                          int scale_per_head, int mask_diag, float dropout_p, uint64_t seed,
                          uint32_t stream_id, void* o, float* lse2, void* stream);
/* the same pair over the planes the ViT core itself uses: the attention output as ONE fp16 plane (vit.py:264-265's `out`, 2^-12 instead
 * of bf16's 2^-9), which the backward's row constants delta = rowsum(dO o O) read (kernel-level tests of the production configuration) */
int v1t_attention_forward_f16o(const void* qkv, int B, int H, int T, int DP, const float* scale, int scale_per_head, int mask_diag,
                               float dropout_p, uint64_t seed, uint32_t stream_id, void* o_f16, float* lse2, void* stream);
int v1t_attention_backward_ws_f16o(const void* qkv, const void* o_f16, const void* dO, const float* lse2, int B, int H, int T, int DP,
                                   const float* scale, int scale_per_head, int mask_diag, float dropout_p, uint64_t seed,
                                   uint32_t stream_id, float* delta_ws, void* dqkv, float* dscale, void* ds_ws, long long ds_bytes,
                                   void* stream);
int v1t_attention_backward(const void* qkv, const void* o, const void* dO, const float* lse2, int B,
                           int H, int T, int DP, const float* scale, int scale_per_head, int mask_diag,
                           float dropout_p, uint64_t seed, uint32_t stream_id, float* delta_ws,
                           void* dqkv, float* dscale, void* stream);
/* Same with a caller-owned scratch of v1t_attention_backward_ws_bytes(): the dK/dV kernel materialises
 * dS' = P (dP - delta) ([B*H][T][roundup(T,128)] bf16) and dQ = dS' . K runs as one streaming GEMM over it, instead of
 * recomputing S and dP a second time (5 MFMA products instead of 7). Not used for mask_diag (LSA). NULL = recompute. */
long long v1t_attention_backward_ws_bytes(int B, int H, int T);
int v1t_attention_backward_ws(const void* qkv, const void* o, const void* dO, const float* lse2, int B, int H, int T,
                              int DP, const float* scale, int scale_per_head, int mask_diag, float dropout_p,
                              uint64_t seed, uint32_t stream_id, float* delta_ws, void* dqkv, float* dscale,
                              void* ds_ws, long long ds_bytes, void* stream);

/* ------------------------------------------- attention rollout (utils/attention_rollout.py:92-133) */
/* Head-max of the softmax probabilities of one block, recomputed from that block's saved qkv
 * (B*T, 3*H*DP bf16) and log2-sum-exp (B,H,T) instead of the (B,H,T,T) tensors the reference's
 * forward hooks clone (attention_rollout.py:28-36): A (B, T, TP) fp32 with TP = T rounded up to 4
 * (= max over heads, attention_rollout.py:105), rowsum (B,T) = sum_j A[i][j] + 1 (the row sums of A + I,
 * :109-111). */
int v1t_rollout_headmax(const void* qkv, const float* lse2, int B, int H, int T, int DP,
                        const float* scale, int scale_per_head, int mask_diag, float* A, int TP,
                        float* rowsum, void* stream);
/* The same for the first q_rows query rows only (whole 128-row workgroups: rows beyond the last one of them are not written): the row
 * chain starts from v = e_0, so its first step (the LAST block, attention_rollout.py:113-118) reads row 0 of that block's matrix alone. */
int v1t_rollout_headmax_rows(const void* qkv, const float* lse2, int B, int H, int T, int DP,
                             const float* scale, int scale_per_head, int mask_diag, float* A, int TP,
                             float* rowsum, int q_rows, void* stream);
/* Per-head softmax probabilities of one block, P (B, H, T, TP) fp32 (pad columns zero): what the reference's Recorder hooks
 * capture from every block's `attend` module (attention_rollout.py:31-36, stacked to (B, L, H, T, T) at :76), recomputed from
 * the saved qkv and log2-sum-exp like v1t_rollout_headmax. 43.8 MB per image and block at the default size: for a bounded
 * number of images (the "emit_probs" switch of SURVEY.md 8b next to "emit_headmax"). */
int v1t_attention_probs(const void* qkv, const float* lse2, int B, int H, int T, int DP,
                        const float* scale, int scale_per_head, int mask_diag, float* P, int TP,
                        void* stream);
/* One step of the rollout chain restricted to row 0 (only J[-1, 0, 1:] is used, :118):
 * u = v . ((A + I) / rowsum); v == NULL means v = e_0 (first step = row 0 of the last block). */
int v1t_rollout_vecmat(const float* A, const float* rowsum, const float* v, float* u, int B, int T,
                       int TP, void* stream);
/* One step of the FULL matrix chain the reference multiplies out, result = A_hat @ result with A_hat = (A + I) / rowsum
 * (attention_rollout.py:107-117), carried transposed (X = result^T, (B, T, TP) fp32 like A) so that the output is the next
 * step's input:  Xout[n][i] = sum_j Xin[n][j] * A_hat[i][j];  Xin == NULL is the identity (first block). Blocks in the
 * reference's order, first to last; the heat vector J[-1][0, 1:] (:118) is column 0 of the last X: Xout[1:, 0]. Split-bf16
 * MFMA products (~2^-17 relative), fp32 accumulate; 2 T^3 flops per image and step. Xout must not alias Xin. */
int v1t_rollout_matmul(const float* A, const float* rowsum, const float* Xin, float* Xout, int B, int T,
                       int TP, void* stream);

/* Live kernel timing (bench.py roofline): when enabled, every launch of the selected kernel class is
 * bracketed by hipEvents on its own stream. class ids: 0 attention fwd, 1 attention bwd dQ,
 * 2 attention bwd dK/dV, 3 gemm_nt, 4 gemm_tn, 5 readout fwd, 6 readout bwd, 7 rollout matmul. */
int v1t_profile_enable(int kernel_class, int max_launches);   /* kernel_class < 0 disables */
int v1t_profile_read(int* launches, double* total_ms);        /* synchronises the recorded events */

/* ------------------------------------------------------------------ the per-mouse tails of a training step, one launch per stage
 * The reference's step loops over mice (train.py:97-111): per mouse the core shifter MLP (core_shifter.py:24-40, model.py:86-92), the
 * readout's sample positions mu / sigma . eps / clamp / + shift (gaussian2d.py:188-235, 265-268), the Gaussian2d readout
 * (gaussian2d.py:270-276), ELU + 1 (models/utils.py:109-118), the Poisson loss with its sqrt(ds_size / batch) scale (losses.py:141-166,
 * train.py:64-72) and their backward. A `v1t_tail_unit` names one local mouse-batch ("unit"): its n_images images sit at image_offset in
 * the shared token buffer of the core (one core pass over all local mice). Each entry point below runs ONE launch per kernel stage over
 * all units (tables of <= 8 units per launch) instead of one chain of ~11 small launches per mouse; the arithmetic per unit is that of
 * v1t_core_shifter_*, v1t_normal_fill, v1t_readout_grid_*, v1t_gaussian2d_*_parts and v1t_elu1_poisson. */
typedef struct v1t_tail_unit {
    int n_images, n_neurons;      /* images of this unit, neurons of its mouse */
    int image_offset;             /* first image of the unit in the shared (B, T, DP) token / token-gradient buffers */
    int grid_dim;                 /* grid predictor input dim 2 / 3; 0: free parameter `mu` */
    unsigned int eps_stream;      /* Philox stream id of this unit's position noise (v1t_normal_fill) */
    int fill_eps;                 /* 1: draw eps in v1t_tails_prepare; 0: `eps` already holds the noise to use (replayed draws) */
    float loss_scale;             /* sqrt(ds_size / batch size) */
    int feat_stride;              /* floats per neuron row of the neuron-major feature storage */
    const float* pupil;           /* (n, 2) pupil centres: the core shifter's input (NULL without a shifter) */
    const float* response;        /* (n, N) targets */
    const float* sp[6];           /* core shifter W0, b0, W2, b2, W4, b4 */
    float* dsp[6];                /* ... their gradients (+=) */
    const float* src;             /* (N, grid_dim) normalised cortical coordinates */
    const float* gp[4];           /* grid predictor W0, b0, W2, b2 */
    float* dgp[4];                /* ... their gradients (+=) */
    const float* mu;  float* dmu; /* free-parameter positions (grid_dim == 0) and gradient */
    const float* sigma; float* dsigma;
    const float* feat; float* dfeat;   /* [N][feat_stride] features and gradient (+=) */
    const float* bias; float* dbias;   /* [N] or NULL */
    float* shift; float* dshift;  /* (n, 2) scratch: shifter output / its gradient (dshift zero on entry of v1t_tails_backward); NULL: no shifter */
    float* eps; float* grid; float* dgrid;   /* (n, N, 2) scratch */
    float* u; float* yhat; float* du;        /* (n, N) scratch: readout pre-activation, prediction (may be NULL), dLoss/du */
    float* loss;                  /* this unit's loss accumulator (+=, atomic: zero it per step) */
    void* rws; long long rws_bytes;   /* v1t_gaussian2d_backward_ws_bytes(n, gh, gw, N) */
    void* gws; long long gws_bytes;   /* v1t_readout_grid_backward_ws_bytes(n, N) */
} v1t_tail_unit;
/* stage 1, needs nothing from the core: shifter forward, position noise (units with fill_eps), sample positions, tap sort */
int v1t_tails_prepare(const v1t_tail_unit* units, int n_units, unsigned long long eps_seed, int gh, int gw, void* stream);
/* stage 2, behind the core forward: readout -> ELU1 + Poisson (du, per-unit loss, loss_total += every unit's loss; NULL: skip) ->
 * dz gathered into dtokens (+=, zero it per step: v1t_fill_zero). tokens / dtokens: element (image b, cell, channel c) at
 * [b * zsb + cell * zsc + c], pointing at the first cell (class-token row skipped). */
int v1t_tails_forward(const v1t_tail_unit* units, int n_units, const float* tokens, float* dtokens, long long zsb, long long zsc, int C, int gh,
                      int gw, float* loss_total, void* stream);
/* stage 3, may overlap the core backward: d features / d bias / d grid, grid-predictor (or mu) and sigma gradients, d shift, shifter gradients */
int v1t_tails_backward(const v1t_tail_unit* units, int n_units, const float* tokens, long long zsb, long long zsc, int C, int gh, int gw,
                       void* stream);
/* AdamW (+ folded L1) over several arena ranges in one launch per 24 ranges: what v1t_adamw_step does per range */
typedef struct v1t_adam_range {
    float* p; float* g; float* m; float* v;
    long long n;
    float lr, l1;
    int step;   /* 1-based step count of the arena the range belongs to */
    int pad_;
} v1t_adam_range;
int v1t_adamw_multi(const v1t_adam_range* ranges, int n, float beta1, float beta2, float eps, float weight_decay, int zero_grad, void* stream);
/* zero `bytes` bytes at a 16-byte aligned device address (the step's token-gradient buffer and loss / d shift accumulators) */
int v1t_fill_zero(void* p, long long bytes, void* stream);

/* The step's inputs in one launch: unit i's n_images[i] images (C, IH, IW) -> ImageCropper's bilinear resize to (OH, OW) (image_cropper.py:96-99; a plain
 * copy when the sizes are equal) into consecutive slices of `out`, and - when beh_out is given - the BehaviorMLP input rows cat(behaviors (na),
 * pupil_centers (nb)) (vit.py:431-432) into consecutive rows of beh_out (row stride na + nb; nb = 0: behaviours only). What v1t_resize_bilinear and
 * v1t_concat2 do per mouse. */
int v1t_inputs_multi(const float* const* images, const float* const* behaviors, const float* const* pupil_centers, const int* n_images, int n_units,
                     int C, int IH, int IW, float* out, int OH, int OW, float* beh_out, int na, int nb, void* stream);

/* Measurement aid (no reference counterpart; SURVEY.md 8d asks for the roofline fraction against the datasheet AND the measured peak):
 * runs v_mfma_f32_32x32x16_bf16 back to back on every CU (register operands, random data, `waves_per_simd` 1 or 2, 16 * iters MFMAs
 * per wave), four launches back to back, and returns for the LAST one (the sustained rate - the chip's power management needs some
 * milliseconds of load to settle: pass iters for >= 10 ms per launch) the TFLOP/s from its wall time, the shader clock it held
 * (s_memtime / s_memrealtime) and the cycles per MFMA and SIMD. Allocates and frees its own few KB; synchronises `stream`. */
int v1t_mfma_peak_probe(int iters, int waves_per_simd, double* tflops, double* ghz, double* cycles_per_mfma, void* stream);

/* LayerNorm (vit.py:220,145) forward: z bf16 (rows, DP) = LN(x (+ inject[b])) ; backward: see
 * csrc/elementwise.h LnBwdArgs (gout = gin + dLN; optional token-sum, next-branch cast + bias colsum) */
int v1t_layernorm_forward(const float* x, const float* inject, float* xout, const float* gamma,
                          const float* beta, void* z, float* mean, float* rstd, int B, int T, int D,
                          int DP, float eps, void* stream);
int v1t_layernorm_backward(const float* dz, const float* x, const float* mean, const float* rstd,
                           const float* gamma, const float* gin, float* gout, float* dgamma,
                           float* dbeta, float* dinject, void* dy_next, float* dbias_next, int B,
                           int T, int D, int DP, void* stream);

#ifdef __cplusplus
}
#endif
#endif
