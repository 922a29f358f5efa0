/*
 * skd.h -- C ABI of libskd_hip.so: the MI355X (gfx950) kernels behind the
 * distillation-step hot path of irfanICMLL/structure_knowledge_distillation.
 *
 * Conventions (kept from the reference's native boundary, libs/src/bn.h:7-19 and
 * libs/src/lib_cffi.cpp:1-2,36-168):
 *   - every entry returns int: 1 = success, 0 = failure (libs/src/bn.cu:245-249;
 *     the Python side turns 0 into RuntimeError like libs/functions.py:13-16);
 *   - raw DEVICE pointers + sizes only, no framework types; a NULL pointer means
 *     "optional tensor absent" (lib_cffi.cpp:62-63 / `weight != 0` in bn.cu:153);
 *   - the caller owns all memory, outputs are pre-sized, kernels never allocate
 *     (the legacy skd_bn_* entries keep one grow-only scratch per device, see below);
 *   - dweight / dbias are ACCUMULATED into (+=), caller zero-fills (bn.cu:217-229);
 *   - asynchronous on the caller's stream (`stream` is a hipStream_t passed as void*;
 *     NULL = the default stream) -- replaces THCState_getCurrentStream, lib_cffi.cpp:37;
 *   - stateless / re-entrant: one host thread per GPU with that GPU current.
 * All tensors are contiguous fp32 NCHW unless stated; N = batch, C = channels,
 * S = product of the spatial dims (lib_cffi.cpp:24-34).
 */
#ifndef SKD_H_
#define SKD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *skd_stream_t; /* hipStream_t */

/* activation codes for the fused entries (libs/functions.py:7-10) */
#define SKD_ACT_NONE 0
#define SKD_ACT_LEAKY_RELU 1
#define SKD_ACT_ELU 2
/* forward-only (inference): the nn.ReLU that follows BatchNorm2d in networks/pspnet_combine.py:36,68,72,
 * fused into the normalise pass.  Not invertible from the output, so the backward entries reject it. */
#define SKD_ACT_RELU 3

/* library / build identification: returns e.g. 950 for gfx950 */
int skd_abi_version(void);
int skd_target_arch(void);

/* ------------------------------------------------------------------------------------
 * 1. Drop-in replacements, one per reference export (same argument lists).
 * ---------------------------------------------------------------------------------- */
/* replaces _bn_mean_var_cuda, libs/src/bn.h:7 (kernel bn.cu:125-138) */
int skd_bn_mean_var(int N, int C, int S, const float *x, float *mean, float *var, skd_stream_t stream);
/* replaces _bn_forward_cuda, bn.h:8-9 (kernel bn.cu:140-165); y and z may alias x */
int skd_bn_forward(int N, int C, int S, const float *x, const float *mean, const float *var,
                   const float *weight, const float *bias, float *y, float *z, float eps,
                   skd_stream_t stream);
/* replaces _bn_edz_eydz_cuda, bn.h:10-11 (kernel bn.cu:167-184) */
int skd_bn_edz_eydz(int N, int C, int S, const float *z, const float *dz, const float *weight,
                    const float *bias, float *edz, float *eydz, float eps, skd_stream_t stream);
/* replaces _bn_backward_cuda, bn.h:12-14 (kernel bn.cu:186-232); dx/dweight/dbias may be NULL */
int skd_bn_backward(int N, int C, int S, const float *dz, const float *z, const float *var,
                    const float *weight, const float *bias, const float *edz, const float *eydz,
                    float *dx, float *dweight, float *dbias, float eps, skd_stream_t stream);
/* replace _leaky_relu_cuda / _leaky_relu_backward_cuda / _elu_* , bn.h:15-19 (bn.cu:302-377);
 * here N is the flat element count */
int skd_leaky_relu(int64_t N, float *x, float slope, skd_stream_t stream);
int skd_leaky_relu_backward(int64_t N, const float *x, float *dx, float slope, skd_stream_t stream);
int skd_elu(int64_t N, float *x, skd_stream_t stream);
int skd_elu_backward(int64_t N, const float *x, float *dx, skd_stream_t stream);
int skd_elu_inv(int64_t N, float *x, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 2. Fused InPlace-ABN path (what libs/functions.py:70-162 does in 3-6 launches + torch ops).
 *    Workspace: skd_abn_workspace_floats(N,C,S) floats, owned by the caller.
 * ---------------------------------------------------------------------------------- */
int64_t skd_abn_workspace_floats(int N, int C, int S);

/* training forward, single rank: one-pass shifted statistics -> mean/var (+ running-stat update,
 * functions.py:90-91, n = N*S*replicas) -> in-place normalise + affine(|w|+eps) + activation.
 * running_mean / running_var may be NULL (no update). mean/var [C] receive the batch statistics. */
int skd_abn_forward_train(int N, int C, int S, float *x, const float *weight, const float *bias,
                          float *running_mean, float *running_var, float *mean, float *var,
                          float momentum, float eps, int activation, float slope,
                          float *workspace, skd_stream_t stream);
/* split form for cross-rank synchronised statistics (functions.py:183-209):
 *   stats   : per-rank mean/var only (no running update, no normalisation)
 *   apply   : normalise with externally supplied (combined) mean/var; also the eval-mode forward
 *             (mean = running_mean, var = running_var, functions.py:92-93) */
int skd_abn_stats(int N, int C, int S, const float *x, float *mean, float *var, float *workspace,
                  skd_stream_t stream);
int skd_abn_apply(int N, int C, int S, float *x, const float *mean, const float *var,
                  const float *weight, const float *bias, float eps, int activation, float slope,
                  skd_stream_t stream);
/* x <- act(bn(x) + residual): the tail of a residual block, `out = bn(conv(..)); out = out + residual;
 * relu(out)` (networks/pspnet_combine.py:37-43, 78-82), in ONE in-place pass (12 B/element instead of
 * 8 + 12 + 8).  mean/var as for skd_abn_apply; residual has the shape of x. */
int skd_abn_apply_residual(int N, int C, int S, float *x, const float *residual, const float *mean,
                           const float *var, const float *weight, const float *bias, float eps,
                           int activation, float slope, skd_stream_t stream);
/* Training-time BN -> (+ residual) -> ReLU as ONE op, out of place (networks/pspnet_combine.py:36-43, 68-82:
 * BatchNorm2d = InPlace-ABN(activation='none'), then nn.ReLU, at block tails `out + residual` first).
 * The convolution output x is left untouched and is what backward reads (y is recomputed from x, mean, var);
 * `out` = act(bn(x) [+ residual]) is what the next layer consumes; the ReLU mask is `out > 0`.
 *   apply_to / forward_train_to : generic out-of-place forms of skd_abn_apply / skd_abn_forward_train
 *                                 (residual may be NULL; activation may be SKD_ACT_RELU)
 *   relu_backward_reduce        : edz = mean(dz), eydz = mean(y*dz) with dz = dout * (out > 0)
 *   relu_backward_dx            : dx = (dz - edz - y*eydz) * gamma * invStd ; dres = dz (may be NULL);
 *                                 dweight / dbias accumulated (+=) like skd_bn_backward */
int skd_abn_apply_to(int N, int C, int S, const float *x, const float *residual, float *out,
                     const float *mean, const float *var, const float *weight, const float *bias, float eps,
                     int activation, float slope, skd_stream_t stream);
int skd_abn_forward_train_to(int N, int C, int S, const float *x, const float *residual, float *out,
                             const float *weight, const float *bias, float *running_mean,
                             float *running_var, float *mean, float *var, float momentum, float eps,
                             int activation, float slope, float *workspace, skd_stream_t stream);
int skd_abn_relu_backward_reduce(int N, int C, int S, const float *x, const float *out, const float *dout,
                                 const float *mean, const float *var, float *edz, float *eydz, float eps,
                                 float *workspace, skd_stream_t stream);
int skd_abn_relu_backward_dx(int N, int C, int S, const float *x, const float *out, const float *dout,
                             const float *mean, const float *var, const float *weight, const float *edz,
                             const float *eydz, float *dx, float *dres, float *dweight, float *dbias,
                             float eps, skd_stream_t stream);
/* inference form for channels-last (NHWC) tensors: x is (rows = N*H*W, C) row-major, C % 4 == 0;
 * x <- act(bn(x) [+ residual]) in place (residual may be NULL, same layout).  Lets the frozen teacher run its
 * MIOpen convolutions NHWC-native. */
int skd_abn_apply_nhwc(int64_t rows, int C, float *x, const float *residual, const float *mean,
                       const float *var, const float *weight, const float *bias, float eps, int activation,
                       float slope, skd_stream_t stream);
/* Channels-last (NHWC) TRAINING forms: x is (rows = N*H*W, C) row-major, C a power of two in [4, 1024].
 * Same maths and conventions as the NCHW entries above (statistics over the rows of each channel, running-stat
 * update with n = rows); the two dx entries take `accumulate`: nonzero = dweight / dbias are accumulated into like
 * bn.cu:217-229, zero = they are written, so the caller needs no zero-fill; `out` may equal `x` (the in-place InPlace-ABN) or be a
 * separate tensor (the BN -> [+ residual] -> ReLU fusion, which keeps x for backward).  They let MIOpen run its
 * NHWC-native fp32 kernels without NCHW<->NHWC transposes.  workspace: skd_abn_nhwc_workspace_floats(rows, C). */
int64_t skd_abn_nhwc_workspace_floats(int64_t rows, int C);
int skd_abn_stats_nhwc(int64_t rows, int C, const float *x, float *mean, float *var, float *workspace,
                       skd_stream_t stream);
int skd_abn_apply_nhwc_to(int64_t rows, int C, const float *x, const float *residual, float *out,
                          const float *mean, const float *var, const float *weight, const float *bias, float eps,
                          int activation, float slope, skd_stream_t stream);
int skd_abn_forward_train_nhwc(int64_t rows, int C, const float *x, const float *residual, float *out,
                               const float *weight, const float *bias, float *running_mean, float *running_var,
                               float *mean, float *var, float momentum, float eps, int activation, float slope,
                               float *workspace, skd_stream_t stream);
int skd_abn_backward_reduce_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *weight,
                                 const float *bias, float *edz, float *eydz, float eps, int activation,
                                 float slope, float *workspace, skd_stream_t stream);
int skd_abn_backward_dx_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *var,
                             const float *weight, const float *bias, const float *edz, const float *eydz,
                             float *dx, float *dweight, float *dbias, float eps, int activation, float slope,
                             int accumulate, skd_stream_t stream);
int skd_abn_relu_backward_reduce_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout,
                                      const float *mean, const float *var, float *edz, float *eydz, float eps,
                                      float *workspace, skd_stream_t stream);
int skd_abn_relu_backward_dx_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout,
                                  const float *mean, const float *var, const float *weight, const float *edz,
                                  const float *eydz, float *dx, float *dres, float *dweight, float *dbias,
                                  float eps, int accumulate, skd_stream_t stream);
/* The same two passes for a forward WITHOUT residual: the ReLU mask is then a function of x alone
 * (((x - mean) * invStd) * gamma + beta > 0, evaluated with the forward pass's own expression), so `out` is not read:
 * 8 instead of 12 bytes per element in the reduce pass, 12 instead of 16 in the dx pass. */
int skd_abn_relu_backward_reduce_nhwc_x(int64_t rows, int C, const float *x, const float *dout, const float *mean,
                                        const float *var, const float *weight, const float *bias, float *edz,
                                        float *eydz, float eps, float *workspace, skd_stream_t stream);
int skd_abn_relu_backward_dx_nhwc_x(int64_t rows, int C, const float *x, const float *dout, const float *mean,
                                    const float *var, const float *weight, const float *bias, const float *edz,
                                    const float *eydz, float *dx, float *dweight, float *dbias, float eps,
                                    int accumulate, skd_stream_t stream);
/* reduce + dx in ONE call (round 3).  When the tensor fits the chip's register file (rows * C up to ~9.4 M elements) both
 * passes run as a single launch that keeps (y, dz) in VGPRs between them -- 12 instead of 20 bytes per element; larger
 * tensors run the two entries above back to back.  edz / eydz are written as well (16-byte aligned for the one-launch
 * path).  skd_abn_forward_train_nhwc takes the same register-resident route for up to ~17.8 M elements (8 instead of 12
 * bytes per element).  SKD_ABN_FUSED=0 in the environment keeps the two-launch passes.
 * skd_abn_relu_backward_nhwc: out == NULL selects the mask-from-x form (then dres must be NULL). */
int skd_abn_backward_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *var, const float *weight,
                          const float *bias, float *edz, float *eydz, float *dx, float *dweight, float *dbias,
                          float eps, int activation, float slope, int accumulate, float *workspace,
                          skd_stream_t stream);
int skd_abn_relu_backward_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout,
                               const float *mean, const float *var, const float *weight, const float *bias,
                               float *edz, float *eydz, float *dx, float *dres, float *dweight, float *dbias,
                               float eps, int accumulate, float *workspace, skd_stream_t stream);
/* cross-replica combine in one launch (functions.py:196-197, 208-209): gathered is (G, 2, C) = every rank's
 * [mean, var]; writes the combined mean / var and, when the running buffers are given, updates them.
 * weights == NULL: the reference rule (equal per-rank sample counts), n = the POOLED count.
 * weights != NULL: G floats w_g = n_g / sum(n) -> exact pooled statistics for unequal shards (equal shards give the
 * reference rule back); n = THIS rank's count, rank = this rank's index into weights (pooled count = n / w[rank]). */
int skd_abn_combine_stats(int G, int C, const float *gathered, const float *weights, int rank, float *mean,
                          float *var, float *running_mean, float *running_var, float momentum, double n,
                          skd_stream_t stream);
/* running-stat update with an explicit sample count n (functions.py:209) */
int skd_abn_update_running(int C, float *running_mean, float *running_var, const float *mean,
                           const float *var, float momentum, double n, skd_stream_t stream);

/* backward: z = saved forward OUTPUT (post activation, read-only here -- the activation is undone
 * in registers instead of rewriting z/dz like functions.py:54-62).
 *   reduce : edz/eydz [C] (zero when training == 0, functions.py:146-147)
 *   dx     : dx (may be NULL), dweight/dbias (+=, may be NULL) from given edz/eydz
 *   skd_abn_backward = reduce + dx in one call (single rank). */
int skd_abn_backward_reduce(int N, int C, int S, const float *z, const float *dz, const float *weight,
                            const float *bias, float *edz, float *eydz, float eps, int activation,
                            float slope, float *workspace, skd_stream_t stream);
int skd_abn_backward_dx(int N, int C, int S, const float *z, const float *dz, const float *var,
                        const float *weight, const float *bias, const float *edz, const float *eydz,
                        float *dx, float *dweight, float *dbias, float eps, int activation,
                        float slope, skd_stream_t stream);
int skd_abn_backward(int N, int C, int S, const float *z, const float *dz, const float *var,
                     const float *weight, const float *bias, float *edz, float *eydz, float *dx,
                     float *dweight, float *dbias, float eps, int activation, float slope,
                     int training, float *workspace, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 3. Pixel-wise distillation loss, utils/criterion.py:219-226 (CriterionPixelWise.forward):
 *    loss = -sum_{n,h,w,c} softmax(T)_c * log_softmax(S)_c / (W*H)   (not divided by N).
 *    logits (N, C, H, W) fp32; one launch produces the loss AND dloss/dS
 *    (= (softmax(S) - softmax(T)) / (W*H)); grad_s may be NULL.
 *    workspace: skd_pixelwise_workspace_floats(N, HW) floats.
 * ---------------------------------------------------------------------------------- */
int64_t skd_pixelwise_workspace_floats(int N, int HW);
int skd_pixelwise_loss(int N, int C, int HW, const float *logits_s, const float *logits_t,
                       float *loss /* [1] */, float *grad_s /* (N,C,HW) or NULL */,
                       float *workspace, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 4. Pair-wise similarity loss, utils/criterion.py:236-245 + utils/utils.py:170-183.
 *    Stages (each its own entry so the autograd wrapper can keep what backward needs):
 *      pool   : MaxPool2d(kernel=stride=(kh,kw), pad 0, ceil_mode=True) with argmax
 *               (flat h*W+w index inside the plane, int32; first maximum in row-major scan
 *               order wins, NaN propagates -- PyTorch semantics)
 *      norm   : Fhat[b,c,m] = F[b,c,m] / (sqrt(sum_c F^2) + 1e-8)      (utils.py:170-176)
 *      gram   : G[b] = Fhat_T[b]^T Fhat_T[b] - Fhat_S[b]^T Fhat_S[b]  (M x M, fp32 MFMA);
 *               loss = sum G^2 / M^2 / B                               (utils.py:178-183)
 *      bwd    : dFhat_S = -4 * gscale / (M^2 B) * Fhat_S G ; dP = dFhat_S / norm   (autograd of the above with
 *               the norm a constant, utils.py:175)
 *      unpool : scatter dP through the argmax into a dense (B,C,H,W) gradient
 * ---------------------------------------------------------------------------------- */
int skd_maxpool_argmax(int planes, int H, int W, int kh, int kw, const float *x,
                       float *pooled /* (planes, OH*OW) */, int32_t *index /* same, or NULL */,
                       skd_stream_t stream);
/* leading dimension (floats) of the zero-padded node axis used by the GEMM stages: M rounded up
 * to the 128-wide MFMA tile */
int skd_pairwise_ldm(int M);
/* pooled (B, C, M) -> fhat (B, C, ldm) zero padded; optional node-major copy fhat_t (B, ldm, ldc)
 * (ldc >= C, multiple of 128, zero padded; NULL to skip); norm (B, M) (NULL to skip) */
int skd_channel_l2_normalise(int B, int C, int M, const float *pooled, float *fhat, int ldm,
                             float *fhat_t, int ldc, float *norm, skd_stream_t stream);
int64_t skd_pairwise_workspace_floats(int B, int M);
/* G (B, ldm, ldm) or NULL; loss [1] */
int skd_pairwise_gram_loss(int B, int Cs, int Ct, int M, int ldm, const float *fhat_s,
                           const float *fhat_t, float *G, float *loss, float *workspace,
                           skd_stream_t stream);
/* fhat_s (B, Cs, ldm): the normalised student panel the forward already holds (the entry makes its own node-major copy in the
 * workspace: round 5); grad_loss [1] on the device; dpooled (B, Cs, ldm), columns >= M zero.
 * The contraction over the nodes is split over workgroups for large M; the partials are combined in a fixed order
 * (deterministic).  workspace: skd_pairwise_backward_workspace_floats(B, Cs, M) floats, 16-byte aligned, REQUIRED. */
int64_t skd_pairwise_backward_workspace_floats(int B, int Cs, int M);
int skd_pairwise_backward(int B, int Cs, int M, int ldm, const float *fhat_s,
                          const float *G, const float *norm_s, const float *grad_loss,
                          float *dpooled, float *workspace, skd_stream_t stream);
/* Small graphs (M <= 64; the reference default --pool-scale 0.5 gives M = 9): norm + both Grams + loss (+ gradient)
 * of one image in one workgroup, no padding to MFMA tiles.  pooled_s (B, Cs, M), pooled_t (B, Ct, M);
 * loss[0] = sum (A_T - A_S)^2 / M^2 / B; dpooled (B, Cs, M) or NULL = d loss / d pooled_s for an upstream gradient
 * of 1 (the norm is a constant, utils.py:175).  workspace: B floats. */
int skd_pairwise_small(int B, int Cs, int Ct, int M, const float *pooled_s, const float *pooled_t, float *loss,
                       float *dpooled, float *workspace, skd_stream_t stream);
/* dpooled rows have stride ldp floats (ldp = ldm from above, or M for a dense tensor) */
int skd_maxunpool_scatter(int planes, int H, int W, int kh, int kw, const float *dpooled,
                          int64_t ldp, const int32_t *index, float *dx /* (planes, H, W) */,
                          skd_stream_t stream);

/* Channels-last forms of the pool / un-pool (round 5): x and dx are (B, H, W, C) with C % 4 == 0 and 16-byte aligned -- the
 * layout the PSP features have inside NetModel --, pooled / index keep the planar (B, C, OH*OW) order and the flat NCHW
 * index h*W+w of the entries above (bit-identical results), so the rest of the pair-wise chain is unchanged and no NCHW copy
 * of the features (17 + 69 MB per step at batch 8) is made for the criterion (utils/criterion.py:241-244). */
int skd_maxpool_argmax_nhwc(int B, int C, int H, int W, int kh, int kw, const float *x,
                            float *pooled /* (B, C, OH*OW) */, int32_t *index /* same, or NULL */, skd_stream_t stream);
int skd_maxunpool_scatter_nhwc(int B, int C, int H, int W, int kh, int kw, const float *dpooled, int64_t ldp,
                               const int32_t *index, float *dx /* (B, H, W, C) */, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 5. Spectral normalisation, networks/spectral.py:23-35 (one power iteration on .data):
 *      v <- l2normalize(W^T u); u <- l2normalize(W v); sigma = u . (W v); w = W / sigma
 *    W is (h, w) row-major (conv weight viewed as (out, in*kh*kw)); u, v updated in place.
 *    workspace: skd_spectral_workspace_floats(h, w) floats.
 *    backward (u, v constants): gW_bar = gW / sigma - (sum(gW * W_bar) / sigma^2) * u v^T
 * ---------------------------------------------------------------------------------- */
int64_t skd_spectral_workspace_floats(int h, int w);
int skd_spectral_norm_forward(int h, int w, const float *w_bar, float *u, float *v,
                              float *sigma /* [1] */, float *w_out /* (h,w) or NULL */,
                              float *workspace, skd_stream_t stream);
int skd_spectral_norm_backward(int h, int w, const float *w_bar, const float *u, const float *v,
                               const float *sigma, const float *grad_w, float *grad_w_bar,
                               float *workspace, skd_stream_t stream);
/* All spectrally normalised layers of a network in ONE call (round 4): the discriminator normalises four weights per
 * forward and runs four forwards per step -- 72 launches of 3-35 us per step with the single-layer entries.  The layers
 * are independent, so each phase runs for all of them in one launch (a workgroup looks its layer up in a by-value table):
 * 3 launches forward, 2 backward for any L <= 8; per layer the same arithmetic, bit-identical results.  h / w: HOST
 * arrays of L ints; the pointer arguments are HOST arrays of L DEVICE pointers (w_out may be NULL: u / v / sigma only).
 * workspace: sum over the layers of skd_spectral_workspace_floats(h_k, w_k) floats. */
int skd_spectral_norm_forward_multi(int L, const int *h, const int *w, const float *const *w_bar, float *const *u,
                                    float *const *v, float *const *sigma, float *const *w_out, float *workspace,
                                    skd_stream_t stream);
int skd_spectral_norm_backward_multi(int L, const int *h, const int *w, const float *const *w_bar, const float *const *u,
                                     const float *const *v, const float *const *sigma, const float *const *grad_w,
                                     float *const *grad_w_bar, float *workspace, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 7. CriterionDSN, utils/criterion.py:179-188, fused:
 *      loss = CE(up(main), target) + aux_weight * CE(up(dsn), target)
 *    up = bilinear upsample (h,w) -> (H,W), align_corners=True (F.upsample, criterion.py:182,185);
 *    CE = torch.nn.CrossEntropyLoss(ignore_index) with mean reduction over the non-ignored pixels
 *    (criterion.py:175).  logits (B, C, h, w) fp32, target (B, H, W) int64; a target outside [0, C) that
 *    is not ignore_index is skipped like ignore_index (PyTorch raises a device-side assert there).
 *    One call produces the loss and dloss/dlogits for both heads (grad pointers may be NULL;
 *    logits_dsn may be NULL -> single CE).  C <= 64.  No (B, C, H, W) tensor is ever materialised.
 *    workspace: skd_ce_dsn_workspace_floats(...) floats.
 * ---------------------------------------------------------------------------------- */
int64_t skd_ce_dsn_workspace_floats(int B, int C, int h, int w, int H, int W);
int skd_ce_dsn_forward(int B, int C, int h, int w, int H, int W, const float *logits_main,
                       const float *logits_dsn, const int64_t *target, int ignore_index,
                       float aux_weight, float *loss /* [1] */, float *grad_main /* (B,C,h,w) or NULL */,
                       float *grad_dsn /* (B,C,h,w) or NULL */, float *workspace, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 8. Pyramid pooling module data movement, PSPModule, networks/pspnet_combine.py:86-112:
 *      pool    : AdaptiveAvgPool2d(s) for every level s in `sizes` (host array, <= 4 levels, (1,2,3,6) in
 *                the reference) from ONE read of the (planes, H, W) feature map.  Output layout: level k
 *                occupies floats [planes * sum_{j<k} s_j^2, ...) as (planes, s_k * s_k); total
 *                skd_ppm_pooled_floats().  Bins: [floor(i*H/s), ceil((i+1)*H/s)) (adaptive_avg_pool2d).
 *      pool_backward : dx (planes, H, W) <- sum over levels / bins of gpooled[bin] / area(bin)  (overwrites)
 *      concat  : cat (B, L*Cout + Cfeat, H, W) <- [bilinear(prior_k, (H,W), align_corners=True) for k] + [feats]
 *                (F.upsample + torch.cat of pspnet_combine.py:110-111 without materialising the up-sampled
 *                priors).  priors: host array of L device pointers, prior_k is (B, Cout, s_k, s_k).
 *      concat_backward : gpriors[k] (B, Cout, s_k, s_k) <- pull-back of gcat[:, k*Cout:(k+1)*Cout]
 *                (the gradient of feats is the channel slice gcat[:, L*Cout:] itself).
 * ---------------------------------------------------------------------------------- */
int64_t skd_ppm_pooled_floats(int planes, int nsizes, const int *sizes);
int skd_ppm_pool(int planes, int H, int W, int nsizes, const int *sizes, const float *x, float *pooled,
                 skd_stream_t stream);
int skd_ppm_pool_backward(int planes, int H, int W, int nsizes, const int *sizes, const float *gpooled,
                          float *dx, skd_stream_t stream);
int skd_ppm_concat(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes,
                   const float *const *priors, const float *feats, float *cat, skd_stream_t stream);
int skd_ppm_concat_backward(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes,
                            const float *gcat, float *const *gpriors, skd_stream_t stream);
/* Channels-last (NHWC) forms of the four entries above, for networks whose activations are kept channels-last
 * (no NCHW <-> NHWC copies around the pyramid module).  Layouts: feats / dx / gfeats (B, H, W, C); pooled level k
 * (B, s_k, s_k, C) at floats [B * C * sum_{j<k} s_j^2, ...) of one buffer; prior_k / gprior_k (B, s_k, s_k, Cout);
 * cat / gcat (B, H, W, L*Cout + Cfeat).  C, Cout, Cfeat multiples of 4.  workspace: skd_ppm_nhwc_workspace_floats()
 * floats (pass C = 0 or Cout = 0 for the entry that is not used).  concat_backward writes the feature-map slice of
 * gcat to `gfeats` as a contiguous tensor (NULL: skip) and the prior gradients to gpriors (NULL: skip). */
int64_t skd_ppm_nhwc_workspace_floats(int B, int C, int Cout, int H, int W, int nsizes, const int *sizes);
int skd_ppm_pool_nhwc(int B, int C, int H, int W, int nsizes, const int *sizes, const float *x, float *pooled,
                      float *workspace, skd_stream_t stream);
int skd_ppm_pool_backward_nhwc(int B, int C, int H, int W, int nsizes, const int *sizes, const float *gpooled,
                               float *dx, skd_stream_t stream);
int skd_ppm_concat_nhwc(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes,
                        const float *const *priors, const float *feats, float *cat, skd_stream_t stream);
int skd_ppm_concat_backward_nhwc(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes,
                                 const float *gcat, float *const *gpriors, float *gfeats, float *workspace,
                                 skd_stream_t stream);

/* The pyramid priors folded THROUGH the 3x3 bottleneck convolution (networks/pspnet_combine.py:104-111: the
 * concatenation, the up-sampling of the priors and the convolution over their channels are all linear):
 *     bottleneck_conv(cat(up(prior_1..L), feats)) = conv3x3(feats; W[:, L*Cm:]) + fold(Z_1..L)
 *     Z_k (B, s_k, s_k, 9, Cout) = prior_k (B*s_k^2, Cm) x W[:, k*Cm:(k+1)*Cm] rearranged to (Cm, 9*Cout)   (caller's GEMM)
 *     fold[b][y][x][co] = sum_{k, tap (ty, tx) with (y + ty - 1, x + tx - 1) inside the map}
 *                         bilinear_{align_corners}(Z_k[b][.][.][tap][co]) at (y + ty - 1, x + tx - 1)
 * skd_ppm_fold_nhwc ADDS fold to `out` (B, H, W, Cout) in place (out holds the feature-map part of the convolution);
 * skd_ppm_fold_backward_nhwc writes gz_k = d loss / d Z_k from gout = d loss / d out (the gradient w.r.t. the
 * feature-map part is gout itself).  ldz = distance in floats between consecutive (b, jy, jx) rows of every Z_k / gz_k
 * (9 * Cout when dense; L * 9 * Cout when the caller computes all levels with ONE GEMM of the stacked priors against the
 * stacked weight blocks and passes pointers to the diagonal blocks).  3x3 kernel, padding 1, stride 1, dilation 1; Cout % 4 == 0;
 * 3 * sum_k s_k <= 64 (LDS).  workspace: skd_ppm_fold_nhwc_workspace_floats() floats (backward only). */
int64_t skd_ppm_fold_nhwc_workspace_floats(int B, int Cout, int H, int W, int nsizes, const int *sizes);
int skd_ppm_fold_nhwc(int B, int Cout, int H, int W, int nsizes, const int *sizes, const float *const *z, int64_t ldz,
                      float *out, skd_stream_t stream);
int skd_ppm_fold_backward_nhwc(int B, int Cout, int H, int W, int nsizes, const int *sizes, const float *gout,
                               float *const *gz, int64_t ldz, float *workspace, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 9. Whole-image evaluation tail, networks/evaluate.py:106-113, 186-206:
 *      up = bilinear upsample (h,w) -> (H,W), align_corners=True;  pred = argmax_c up (first maximum, uint8);
 *      confusion[gt * C + pred] += 1 for every pixel whose label is not ignore_index (int64 counts, ACCUMULATED
 *      into -- the caller zero-fills once per evaluation run, like evaluate.py:166).
 *    logits (B, C, h, w) fp32; target (B, H, W) int64 or NULL (prediction only); pred (B, H, W) uint8 or NULL.
 *    Integer outputs are bit-exact with the plain-C oracle (fp contraction is disabled in the kernel).  C <= 64.
 * ---------------------------------------------------------------------------------- */
int skd_seg_confusion(int B, int C, int h, int w, int H, int W, const float *logits, const int64_t *target,
                      int ignore_index, uint8_t *pred, int64_t *confusion, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 6. Deterministic two-stage sum (used by the loss kernels; exposed for tests).
 * ---------------------------------------------------------------------------------- */
int skd_sum_f32(int64_t n, const float *x, float *out /* [1] */, float scale, float *workspace,
                skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 11. 1x1 convolution + eval-mode InPlace-ABN (+ residual) + activation as one fp32-MFMA GEMM (frozen teacher,
 *     networks/pspnet_combine.py:65-84).  Channels-last operands:
 *       out[m][n] = act( ((sum_k x[m][k] * w[n][k] - mean[n]) * invstd[n]) * (|weight[n]| + eps) + bias[n] [+ residual[m][n]] )
 *     x (M, K) = (B*H*W, Cin); w (N, K) = the (Cout, Cin, 1, 1) convolution weight; residual / out (M, N); mean / var =
 *     the running statistics; weight / bias may be NULL (gamma 1, beta 0); activation: SKD_ACT_NONE / RELU / LEAKY_RELU.
 *     skd_conv1x1_abn_supported(): K % 16 == 0 and N % 128 == 0 (every stride-1 1x1 convolution of the ResNet101
 *     teacher except the two with 64 output channels); other problems stay on convolution + skd_abn_apply_nhwc.
 *     skd_conv1x1_abn_pro_nhwc (round 3): the same GEMM with the PRECEDING eval-mode BatchNorm + ReLU applied to x on its
 *     way into LDS -- x[m][k] <- relu(((x[m][k] - mean_k) * invstd_k) * gamma_k + beta_k) -- so that for a bottleneck tail
 *     conv2 -> bn2 -> relu -> conv3 -> bn3 -> + residual -> relu (pspnet_combine.py:71-82) x is the raw 3x3-convolution
 *     output and neither ABN pass exists.  ppack (4, K) = [mean | invstd | gamma | beta] of that BatchNorm, written by
 *     skd_abn_pack_eval_params (invstd = 1 / sqrt(var + eps), gamma = |weight| + eps or 1, beta = bias or 0; bn.cu:146-159):
 *     a frozen network packs once.  16-byte aligned; K <= 512 (the table rides in LDS next to the operand tiles).
 * ---------------------------------------------------------------------------------- */
int skd_conv1x1_abn_supported(int64_t M, int K, int N);
int skd_conv1x1_abn_nhwc(int64_t M, int K, int N, const float *x, const float *w, const float *residual, float *out,
                         const float *mean, const float *var, const float *weight, const float *bias, float eps,
                         int activation, float slope, skd_stream_t stream);
int skd_abn_pack_eval_params(int K, const float *mean, const float *var, const float *weight, const float *bias, float eps,
                             float *pack, skd_stream_t stream);
int skd_conv1x1_abn_pro_nhwc(int64_t M, int K, int N, const float *x, const float *w, const float *residual, float *out,
                             const float *mean, const float *var, const float *weight, const float *bias, float eps,
                             const float *ppack, int activation, float slope, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 12. One-hop exchange of the cross-replica InPlace-ABN statistics (round 3): replaces libs/functions.py:185-205
 *     (master / worker queues + comm.gather + comm.broadcast_coalesced of [mean, var]) and :263-280 ([edz, eydz]) --
 *     and the torch.distributed all_gather / all_reduce this package used first -- by ONE small kernel per exchange
 *     that stores this rank's vector straight into every peer's IPC-mapped mailbox (one xGMI hop), raises per-slot
 *     sequence flags, waits for the peers' flags in its own mailbox and applies the combine rule in the same launch.
 *       ctx = skd_sync_create(world, rank, handle_out)   allocates the rank's mailbox on the CURRENT device and writes
 *             its IPC handle (skd_sync_handle_bytes() bytes) to handle_out; NULL on failure;
 *       skd_sync_connect(ctx, all_handles)               all_handles = world x handle bytes in rank order (gathered by the
 *             caller with whatever host-side transport it has, e.g. dist.all_gather_object); opens the peers' mailboxes;
 *       the three exchanges are COLLECTIVES: every rank calls them in the same order, on any stream;
 *       skd_sync_all_gather    gathered (world, n) <- every rank's n floats, n <= skd_sync_max_floats();
 *       skd_abn_sync_stats     stat (2, C) = this replica's [mean | var] -> combined mean / var + running update: the
 *                              arguments and the arithmetic of skd_abn_combine_stats (bit-identical on the same data);
 *       skd_abn_sync_grad_stats stat (2, C) = [edz | eydz], in place <- sum_g w_g stat_g in rank order (w_g = weights[g],
 *                              or 1 / world when weights == NULL).
 *     A peer that does not arrive within the context's limit (skd_sync_set_timeout) poisons the outputs with NaN (no
 *     device hang) and raises status word 0 (section 13).  world <= 16.
 * ---------------------------------------------------------------------------------- */
int skd_sync_handle_bytes(void);
int skd_sync_max_floats(void);
void *skd_sync_create(int world, int rank, void *handle_out);
int skd_sync_connect(void *ctx, const void *all_handles);
int skd_sync_destroy(void *ctx);
int skd_sync_all_gather(void *ctx, int n, const float *src, float *gathered, skd_stream_t stream);
int skd_abn_sync_stats(void *ctx, int C, const float *stat, const float *weights, float *mean, float *var,
                       float *running_mean, float *running_var, float momentum, double n, skd_stream_t stream);
int skd_abn_sync_grad_stats(void *ctx, int C, float *stat, const float *weights, skd_stream_t stream);
/* How long an exchange waits for a peer before it gives up (outputs NaN + status word 0 raised, section 13).  A fresh
 * context waits 5 s -- its collective self-test must fail fast; the caller raises the limit to what torch.distributed
 * would have waited once the group is known to work (utils/parallel.py: SKD_SYNC_TIMEOUT_S, default 600). */
int skd_sync_set_timeout(void *ctx, double seconds);
/* InPlaceABNSync (libs/functions.py:165-294) for channels-last tensors in ONE call per pass: the arguments of
 * skd_abn_forward_train_nhwc / skd_abn_backward_nhwc / skd_abn_relu_backward_nhwc plus the mailbox context, the
 * per-replica sample weights (NULL = equal shards) and -- forward -- n, the pooled sample count without weights / this
 * replica's with (skd_abn_combine_stats).  When the tensor fits the register file the pass is ONE launch whose channel
 * blocks' last arrivers exchange their block's statistics through the mailboxes inside it (8 / 12 bytes per element, no
 * separate exchange launch); otherwise statistics -> skd_abn_sync_stats / skd_abn_sync_grad_stats -> normalise / dx.  Both
 * forms are ONE exchange of the context's sequence and interoperate (per-channel-block flag words): ranks with ragged
 * shards may take different forms.  SKD_ABN_SYNC_FUSED=0 keeps the three-launch form.  Backward: edz / eydz must be the
 * two halves of one (2, C) buffer.  workspace: skd_abn_nhwc_workspace_floats(). */
int skd_abn_forward_train_nhwc_sync(void *sync_ctx, int64_t rows, int C, const float *x, const float *residual, float *out,
                                    const float *weight, const float *bias, float *running_mean, float *running_var,
                                    float *mean, float *var, const float *replica_weights, float momentum, float eps,
                                    int activation, float slope, double n, float *workspace, skd_stream_t stream);
int skd_abn_backward_nhwc_sync(void *sync_ctx, int64_t rows, int C, const float *z, const float *dz, const float *var,
                               const float *weight, const float *bias, float *edz, float *eydz, float *dx, float *dweight,
                               float *dbias, const float *replica_weights, float eps, int activation, float slope,
                               int accumulate, float *workspace, skd_stream_t stream);
int skd_abn_relu_backward_nhwc_sync(void *sync_ctx, int64_t rows, int C, const float *x, const float *out, const float *dout,
                                    const float *mean, const float *var, const float *weight, const float *bias, float *edz,
                                    float *eydz, float *dx, float *dres, float *dweight, float *dbias,
                                    const float *replica_weights, float eps, int accumulate, float *workspace,
                                    skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 13. Device-raised error words and the co-residency cap of the one-launch InPlace-ABN passes (round 4).
 *     Two kinds of kernels wait inside a launch, both bounded: the mailbox exchanges (for a peer replica) and the
 *     register-resident one-launch ABN passes (a grid barrier per channel block).  A wait that runs out writes NaN into
 *     its outputs AND stores a code into a host-mapped buffer of skd_status_words() 32-bit words that the host reads
 *     without synchronising the device: word 0 = an exchange timed out (value: sequence number | 0x80000000), word 1 = a
 *     grid barrier timed out (the launch was not co-resident).  The Python side checks once per step and raises.
 *     skd_abn_set_fused_max_workgroups(n): upper bound of the workgroups of a one-launch pass on top of the device's own
 *     (its compute-unit count, queried per device -- a CPX partition or a CU-masked device gets a smaller grid or the
 *     two-launch path, never a barrier that cannot complete); ranks sharing one device must share its compute units.
 *     n > 0 sets the bound, n == 0 restores the default, n < 0 only queries; returns the effective cap on the current device.
 * ---------------------------------------------------------------------------------- */
int skd_status_words(void);
int skd_status_read(unsigned *out);
int skd_status_clear(void);
int skd_abn_set_fused_max_workgroups(int n);
/* The two operational switches of the one-launch passes as LIBRARY STATE (round 6): the one-launch (register-resident, grid
 * barrier) passes at all -- environment default SKD_ABN_FUSED, on -- and the cross-replica exchange inside them -- SKD_ABN_SYNC_FUSED,
 * on.  The environment is read ONCE (first query); set(1 / 0) overrides it for the process (utils/parallel.py: RCCL groups start
 * with sync_fused = 0), set(-1) returns to the environment at the next query; get() = the effective value.  No getenv on the call path. */
int skd_abn_set_fused(int on);
int skd_abn_get_fused(void);
int skd_abn_set_sync_fused(int on);
int skd_abn_get_sync_fused(void);
/* out[0] / out[1]: synchronised (*_sync) calls that ran as one launch with the exchange inside / as three launches (host counters) */
int skd_abn_sync_form_counts(int64_t *out);

/* ------------------------------------------------------------------------------------
 * 10. Training-sample transform of the Cityscapes loader on the device, dataset/datasets.py:173-210
 *     (CSDataSet.__getitem__ after the PNG decode), one launch per batch:
 *       label = lut[label] (id -> trainId, datasets.py:143-148,162-171); image / label resized by f (cv2.resize,
 *       INTER_LINEAR 8-bit fixed point / INTER_NEAREST -- bit-exact restatement, see csrc/input_pipeline.hip);
 *       image = float32(image) - mean; bottom / right padding to the crop size (image 0.0, label ignore_label);
 *       crop at (h_off, w_off); HWC -> CHW (or kept channels-last); horizontal mirror when flip < 0.
 *     images (B, H0, W0, 3) uint8 BGR and labels (B, H0, W0) uint8 raw ids are DEVICE buffers (labels / out_label may
 *     both be NULL: image only); scale / dst_h / dst_w / h_off / w_off / flip are per-sample DEVICE arrays holding the
 *     host's random draws (dst = cvRound(size * f), the size of the virtual scaled image); lut (256 bytes) is a DEVICE
 *     array; mean (3 floats, BGR order) is a HOST array.  out_image: (B, 3, crop_h, crop_w) fp32, or (B, crop_h, crop_w, 3)
 *     memory when channels_last != 0; out_label: (B, crop_h, crop_w) int64.
 * ---------------------------------------------------------------------------------- */
int skd_cs_transform(int B, int H0, int W0, const uint8_t *images, const uint8_t *labels, const uint8_t *lut,
                     const double *scale, const int *dst_h, const int *dst_w, const int *h_off, const int *w_off,
                     const int *flip, int crop_h, int crop_w, const float *mean, int ignore_label, float *out_image,
                     int channels_last, int64_t *out_label, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 12. The stem's MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=True) (networks/pspnet_combine.py:135,152 of
 *     both networks) for channels-last tensors: x (B, H, W, C) -> y (B, OH, OW, C), C % 4 == 0; OH / OW are the
 *     caller's (torch's pooling-shape rule; the entry checks that every window starts inside the input and that the
 *     windows cover it).  PyTorch's selection rule: windows scanned row-major, `val > max || isnan(val)` replaces -- the
 *     first maximum wins ties, NaN propagates.  arg (may be NULL for inference): the winner's position INSIDE its
 *     3x3 window, ky * 3 + kx in 0..8, one byte per output element (instead of the stock int64 flat index).
 *     backward: dx[b][y][x][c] = sum of dy over the <= 4 windows containing (y, x) whose arg points at it; every
 *     element of dx is written (no zero-fill, no atomics).
 * ---------------------------------------------------------------------------------- */
int skd_maxpool3x3s2_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, float *y, uint8_t *arg,
                          skd_stream_t stream);
int skd_maxpool3x3s2_backward_nhwc(int B, int C, int H, int W, int OH, int OW, const float *dy, const uint8_t *arg,
                                   float *dx, skd_stream_t stream);
/* Round 6: the TRAINING stem in two fused passes per direction (networks/pspnet_combine.py:176-180: conv3 -> bn3 -> relu3 ->
 * maxpool of the student): BatchNorm (given batch statistics: skd_abn_stats_nhwc / the cross-replica combine) -> ReLU -> the
 * max-pool above, x (B, H, W, C) the raw convolution output, C a power of two in [4, 1024].
 *   forward          pooled (B, OH, OW, C) = maxpool(relu(bn(x))) and arg, bit for bit what skd_abn_apply_nhwc_to(relu) followed by
 *                    skd_maxpool3x3s2_nhwc produce -- the (B, H, W, C) normalised tensor is never written;
 *   backward_reduce  edz / eydz [C] of the BatchNorm for the gradient un-pooled through arg and masked with relu'(bn(x))
 *                    (recomputed from x): skd_maxpool3x3s2_backward_nhwc + skd_abn_relu_backward_reduce_nhwc_x without the
 *                    (B, H, W, C) gradient; workspace: skd_abn_nhwc_workspace_floats(B * H * W, C);
 *   backward_dx      dx (B, H, W, C) (+ dweight / dbias, `accumulate` as in skd_abn_relu_backward_dx_nhwc_x) from the (exchanged,
 *                    for InPlaceABNSync) edz / eydz. */
int skd_abn_relu_maxpool3x3s2_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, const float *mean,
                                   const float *var, const float *weight, const float *bias, float eps, float *pooled,
                                   uint8_t *arg, skd_stream_t stream);
int skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x,
                                                   const float *dpooled, const uint8_t *arg, const float *mean,
                                                   const float *var, const float *weight, const float *bias, float *edz,
                                                   float *eydz, float eps, float *workspace, skd_stream_t stream);
int skd_abn_relu_maxpool3x3s2_backward_dx_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, const float *dpooled,
                                               const uint8_t *arg, const float *mean, const float *var, const float *weight,
                                               const float *bias, const float *edz, const float *eydz, float *dx,
                                               float *dweight, float *dbias, float eps, int accumulate, skd_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 14. The 19-class 1x1 classifier heads (networks/pspnet_combine.py:138-154: Conv2d(mid, num_classes, 1, bias=True)) on
 *     channels-last feature maps (round 6).  x (B*HW, K) = the feature map as it lies in memory, w (C, K) = the (C, K, 1, 1)
 *     convolution weight, bias (C) or NULL; out / gout (B, C, HW) = logits / their gradient in the reference's NCHW layout (what the
 *     criteria consume).  HBM-bound skinny GEMMs: the feature map is read once per direction.
 *       forward    out[b][c][p] = bias[c] + sum_k x[b*HW + p][k] * w[c][k]
 *       backward   gx[m][k] = sum_c gout[b][c][p] * w[c][k]   (channels-last, may be NULL)
 *                  gw[c][k] = sum_m gout[b][c][p] * x[m][k],  gb[c] = sum_m gout[b][c][p]   (WRITTEN, may be NULL; fixed summation
 *                  order: bit-reproducible);  workspace: skd_head1x1_backward_workspace_floats().
 *     skd_head1x1_supported(K, C, backward): C <= 20; forward K % 128 == 0 and K <= 1024; backward K == 128.  Other heads stay
 *     convolutions.
 * ---------------------------------------------------------------------------------- */
int skd_head1x1_supported(int K, int C, int backward);
int skd_head1x1_forward_nhwc(int B, int HW, int K, int C, const float *x, const float *w, const float *bias, float *out,
                             skd_stream_t stream);
int64_t skd_head1x1_backward_workspace_floats(int B, int HW, int K, int C);
int skd_head1x1_backward_nhwc(int B, int HW, int K, int C, const float *x, const float *w, const float *gout, float *gx,
                              float *gw, float *gb, float *workspace, skd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SKD_H_ */
