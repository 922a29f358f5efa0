/*
 * oracle/losses_ref.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C, host-memory restatement of the loss / spectral-norm sections of include/skd.h
 * (sections 3-6), following the reference's Python, so that (a) the HIP kernels can be compared
 * with an independent scalar implementation through the very same ctypes call sites and (b) the
 * host-side autograd wiring can be exercised on a box without a GPU (tests install this library as
 * the C-ABI double).  Pointers are HOST pointers; `stream` is ignored.
 *
 * Follows:
 *   pixel-wise loss        utils/criterion.py:219-226
 *   max-pool (+argmax)     nn.MaxPool2d(k=s, pad 0, ceil_mode=True), utils/criterion.py:243
 *                          (scan rule of PyTorch's max_pool2d: `val > max || isnan(val)` in
 *                          row-major window order -> first maximum, last NaN)
 *   L2 / similarity / sim_dis_compute   utils/utils.py:170-183
 *   spectral norm          networks/spectral.py:10-35
 * Sums are accumulated in double (the reference's float reduction order is unspecified).
 * Pinned against the reference's own Python by tests/test_oracle_c.py (through torch restatements
 * that tests/test_oracle_vs_reference.py pins to the reference) and tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef void *stream_t;

static int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

/* ---- deterministic sum ---------------------------------------------------------------------- */
int skd_sum_f32(int64_t n, const float *x, float *out, float scale, float *ws, stream_t st) {
  (void)ws; (void)st;
  if (n < 0 || !out) return 0;
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += x[i];
  out[0] = (float)(s * (double)scale);
  return 1;
}

/* ---- pixel-wise ----------------------------------------------------------------------------- */
int64_t skd_pixelwise_workspace_floats(int N, int HW) { (void)N; (void)HW; return 1; }

int skd_pixelwise_loss(int N, int C, int HW, const float *ls, const float *lt, float *loss,
                       float *grad, float *ws, stream_t st) {
  (void)ws; (void)st;
  if (N <= 0 || C <= 0 || HW <= 0 || !ls || !lt || !loss) return 0;
  double total = 0.0;
  const double inv_wh = 1.0 / (double)HW;
  for (int n = 0; n < N; ++n)
    for (int p = 0; p < HW; ++p) {
      const int64_t base = (int64_t)n * C * HW + p;
      double ms = -INFINITY, mt = -INFINITY;
      for (int c = 0; c < C; ++c) {
        if (ls[base + (int64_t)c * HW] > ms) ms = ls[base + (int64_t)c * HW];
        if (lt[base + (int64_t)c * HW] > mt) mt = lt[base + (int64_t)c * HW];
      }
      double zs = 0.0, zt = 0.0;
      for (int c = 0; c < C; ++c) {
        zs += exp((double)ls[base + (int64_t)c * HW] - ms);
        zt += exp((double)lt[base + (int64_t)c * HW] - mt);
      }
      const double lse = log(zs);
      for (int c = 0; c < C; ++c) {
        const double s = (double)ls[base + (int64_t)c * HW] - ms;
        const double pt = exp((double)lt[base + (int64_t)c * HW] - mt) / zt; /* softmax(T), criterion.py:223 */
        total -= pt * (s - lse);                                              /* -p_T * log_softmax(S), :225 */
        if (grad) grad[base + (int64_t)c * HW] = (float)((exp(s) / zs - pt) * inv_wh);
      }
    }
  loss[0] = (float)(total * inv_wh);                                          /* / W / H, not / N */
  return 1;
}

/* ---- max-pool with argmax ------------------------------------------------------------------- */
int skd_maxpool_argmax(int planes, int H, int W, int kh, int kw, const float *x, float *pooled,
                       int32_t *index, stream_t st) {
  (void)st;
  if (planes <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || !x || !pooled) return 0;
  const int OH = (int)cdiv64(H, kh), OW = (int)cdiv64(W, kw);
  for (int64_t p = 0; p < planes; ++p) {
    const float *px = x + p * (int64_t)H * W;
    for (int oh = 0; oh < OH; ++oh)
      for (int ow = 0; ow < OW; ++ow) {
        const int r0 = oh * kh, c0 = ow * kw;
        const int r1 = r0 + kh < H ? r0 + kh : H, c1 = c0 + kw < W ? c0 + kw : W;
        float bv = -INFINITY;
        int bi = r0 * W + c0;
        for (int r = r0; r < r1; ++r)
          for (int c = c0; c < c1; ++c) {
            const float v = px[(int64_t)r * W + c];
            if (v > bv || v != v) { bv = v; bi = r * W + c; }
          }
        const int64_t o = (p * OH + oh) * OW + ow;
        pooled[o] = bv;
        if (index) index[o] = bi;
      }
  }
  return 1;
}

int skd_maxunpool_scatter(int planes, int H, int W, int kh, int kw, const float *dpooled, int64_t ldp,
                          const int32_t *index, float *dx, stream_t st) {
  (void)st;
  if (planes <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || !dpooled || !index || !dx) return 0;
  const int OH = (int)cdiv64(H, kh), OW = (int)cdiv64(W, kw);
  memset(dx, 0, sizeof(float) * (size_t)planes * H * W);
  for (int64_t p = 0; p < planes; ++p)
    for (int m = 0; m < OH * OW; ++m)
      dx[p * (int64_t)H * W + index[p * (int64_t)OH * OW + m]] += dpooled[p * ldp + m];
  return 1;
}

/* channels-last forms (include/skd.h section 4): the same scan per channel on (B, H, W, C) data, planar outputs */
int skd_maxpool_argmax_nhwc(int B, int C, int H, int W, int kh, int kw, const float *x, float *pooled, int32_t *index, stream_t st) {
  (void)st;
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || !x || !pooled) return 0;
  const int OH = (int)cdiv64(H, kh), OW = (int)cdiv64(W, kw);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int oh = 0; oh < OH; ++oh)
        for (int ow = 0; ow < OW; ++ow) {
          const int r0 = oh * kh, c0 = ow * kw;
          const int r1 = r0 + kh < H ? r0 + kh : H, c1 = c0 + kw < W ? c0 + kw : W;
          float bv = -INFINITY;
          int bi = r0 * W + c0;
          for (int r = r0; r < r1; ++r)
            for (int q = c0; q < c1; ++q) {
              const float v = x[(((int64_t)b * H + r) * W + q) * C + c];
              if (v > bv || v != v) { bv = v; bi = r * W + q; }
            }
          const int64_t o = (((int64_t)b * C + c) * OH + oh) * OW + ow;
          pooled[o] = bv;
          if (index) index[o] = bi;
        }
  return 1;
}

int skd_maxunpool_scatter_nhwc(int B, int C, int H, int W, int kh, int kw, const float *dpooled, int64_t ldp, const int32_t *index,
                               float *dx, stream_t st) {
  (void)st;
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || !dpooled || !index || !dx) return 0;
  const int OH = (int)cdiv64(H, kh), OW = (int)cdiv64(W, kw);
  memset(dx, 0, sizeof(float) * (size_t)B * H * W * C);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int m = 0; m < OH * OW; ++m) {
        const int64_t plane = (int64_t)b * C + c;
        dx[((int64_t)b * H * W + index[plane * OH * OW + m]) * C + c] += dpooled[plane * ldp + m];
      }
  return 1;
}

/* ---- pair-wise similarity ------------------------------------------------------------------- */
int skd_pairwise_ldm(int M) { return M <= 0 ? 0 : (int)(cdiv64(M, 128) * 128); }
int64_t skd_pairwise_workspace_floats(int B, int M) { (void)B; (void)M; return 1; }

int skd_channel_l2_normalise(int B, int C, int M, const float *pooled, float *fhat, int ldm,
                             float *fhat_t, int ldc, float *norm, stream_t st) {
  (void)st;
  if (B <= 0 || C <= 0 || M <= 0 || !pooled || !fhat || ldm < M) return 0;
  if (fhat_t && ldc < C) return 0;
  for (int b = 0; b < B; ++b) {
    const float *src = pooled + (int64_t)b * C * M;
    float *dst = fhat + (int64_t)b * C * ldm;
    if (fhat_t) memset(fhat_t + (int64_t)b * ldm * ldc, 0, sizeof(float) * (size_t)ldm * ldc);
    for (int c = 0; c < C; ++c)
      for (int m = M; m < ldm; ++m) dst[(int64_t)c * ldm + m] = 0.f;
    for (int m = 0; m < M; ++m) {
      double ss = 0.0;
      for (int c = 0; c < C; ++c) ss += (double)src[(int64_t)c * M + m] * (double)src[(int64_t)c * M + m];
      const float nrm = (float)sqrt(ss) + 1e-8f;                     /* utils.py:170-171 */
      if (norm) norm[(int64_t)b * M + m] = nrm;
      for (int c = 0; c < C; ++c) {
        const float v = src[(int64_t)c * M + m] / nrm;               /* utils.py:176 */
        dst[(int64_t)c * ldm + m] = v;
        if (fhat_t) fhat_t[((int64_t)b * ldm + m) * ldc + c] = v;
      }
    }
  }
  return 1;
}

int skd_pairwise_gram_loss(int B, int Cs, int Ct, int M, int ldm, const float *fs, const float *ft,
                           float *G, float *loss, float *ws, stream_t st) {
  (void)ws; (void)st;
  if (B <= 0 || Cs <= 0 || Ct <= 0 || M <= 0 || !fs || !ft || !loss) return 0;
  if (ldm != skd_pairwise_ldm(M)) return 0;
  double total = 0.0;
  if (G) memset(G, 0, sizeof(float) * (size_t)B * ldm * ldm);
  /* node-major double copies so the channel contraction walks contiguous memory (M = 4225 is 11 GMAC per image);
     rows are independent -> OpenMP over i, per-row sums added in row order (deterministic). */
  double *ns = (double *)malloc(sizeof(double) * (size_t)M * Cs), *nt = (double *)malloc(sizeof(double) * (size_t)M * Ct);
  double *rows = (double *)malloc(sizeof(double) * (size_t)M);
  if (!ns || !nt || !rows) { free(ns); free(nt); free(rows); return 0; }
  for (int b = 0; b < B; ++b) {
    for (int c = 0; c < Cs; ++c)
      for (int m = 0; m < M; ++m) ns[(size_t)m * Cs + c] = (double)fs[((int64_t)b * Cs + c) * ldm + m];
    for (int c = 0; c < Ct; ++c)
      for (int m = 0; m < M; ++m) nt[(size_t)m * Ct + c] = (double)ft[((int64_t)b * Ct + c) * ldm + m];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M; ++i) {
      double row = 0.0;
      for (int j = 0; j < M; ++j) {
        double at = 0.0, as = 0.0;                                   /* einsum('icm,icn->imn'), utils.py:178 */
        for (int c = 0; c < Ct; ++c) at += nt[(size_t)i * Ct + c] * nt[(size_t)j * Ct + c];
        for (int c = 0; c < Cs; ++c) as += ns[(size_t)i * Cs + c] * ns[(size_t)j * Cs + c];
        const double g = at - as;
        row += g * g;
        if (G) G[((int64_t)b * ldm + i) * ldm + j] = (float)g;
      }
      rows[i] = row;
    }
    for (int i = 0; i < M; ++i) total += rows[i];
  }
  free(ns); free(nt); free(rows);
  loss[0] = (float)(total / ((double)M * (double)M) / (double)B);    /* utils.py:181 */
  return 1;
}

int64_t skd_pairwise_backward_workspace_floats(int B, int Cs, int M) { (void)B; (void)Cs; (void)M; return 1; }

int skd_pairwise_backward(int B, int Cs, int M, int ldm, const float *fs, const float *G,
                          const float *norm_s, const float *grad_loss, float *dpooled, float *ws, stream_t st) {
  (void)st; (void)ws;
  if (B <= 0 || Cs <= 0 || M <= 0 || !fs || !G || !norm_s || !grad_loss || !dpooled) return 0;
  /* L = sum G^2/(M^2 B), G = A_T - A_S, A_S = Fh^T Fh  =>  dL/dFh = -4/(M^2 B) Fh G ; dP = dFh / norm */
  const double coef = -4.0 / ((double)M * (double)M * (double)B) * (double)grad_loss[0];
  for (int b = 0; b < B; ++b) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < ldm; ++m) {
      double *acc = (double *)calloc((size_t)Cs, sizeof(double));
      if (m < M)
        for (int n = 0; n < M; ++n) {                                /* G is symmetric in exact arithmetic; index as written */
          const double g = (double)G[((int64_t)b * ldm + n) * ldm + m];
          const float *fcol = fs + (int64_t)b * Cs * ldm + n;          /* Fhat_S[b][c][n], channel-major */
          for (int c = 0; c < Cs; ++c) acc[c] += (double)fcol[(int64_t)c * ldm] * g;
        }
      for (int c = 0; c < Cs; ++c)
        dpooled[((int64_t)b * Cs + c) * ldm + m] = m < M ? (float)(acc[c] * coef / (double)norm_s[(int64_t)b * M + m]) : 0.f;
      free(acc);
    }
  }
  return 1;
}

/* small-graph entry: the same stages composed (normalise -> Gram / loss -> backward with an upstream gradient of 1) */
int skd_pairwise_small(int B, int Cs, int Ct, int M, const float *ps, const float *pt, float *loss, float *dpooled,
                       float *ws, stream_t st) {
  (void)ws;
  if (B <= 0 || Cs <= 0 || Ct <= 0 || M <= 0 || M > 64 || !ps || !pt || !loss) return 0;
  const int ldm = skd_pairwise_ldm(M), ldc = (int)(cdiv64(Cs, 128) * 128);
  float *fs = (float *)malloc(sizeof(float) * (size_t)B * Cs * ldm), *ft = (float *)malloc(sizeof(float) * (size_t)B * Ct * ldm);
  float *fst = (float *)malloc(sizeof(float) * (size_t)B * ldm * ldc), *nrm = (float *)malloc(sizeof(float) * (size_t)B * M);
  float *G = (float *)malloc(sizeof(float) * (size_t)B * ldm * ldm), *dp = (float *)malloc(sizeof(float) * (size_t)B * Cs * ldm);
  int r = fs && ft && fst && nrm && G && dp;
  const float one = 1.f;
  r = r && skd_channel_l2_normalise(B, Cs, M, ps, fs, ldm, fst, ldc, nrm, st) && skd_channel_l2_normalise(B, Ct, M, pt, ft, ldm, NULL, 0, NULL, st);
  r = r && skd_pairwise_gram_loss(B, Cs, Ct, M, ldm, fs, ft, G, loss, NULL, st);
  if (r && dpooled) {
    r = skd_pairwise_backward(B, Cs, M, ldm, fs, G, nrm, &one, dp, NULL, st);
    for (int64_t q = 0; r && q < (int64_t)B * Cs; ++q) memcpy(dpooled + q * M, dp + q * ldm, sizeof(float) * (size_t)M);
  }
  free(fs); free(ft); free(fst); free(nrm); free(G); free(dp);
  return r;
}

/* ---- spectral norm -------------------------------------------------------------------------- */
int64_t skd_spectral_workspace_floats(int h, int w) { (void)h; (void)w; return 1; }

int skd_spectral_norm_forward(int h, int w, const float *wb, float *u, float *v, float *sigma,
                              float *w_out, float *ws, stream_t st) {
  (void)ws; (void)st;
  if (h <= 0 || w <= 0 || !wb || !u || !v || !sigma) return 0;
  double *t = (double *)malloc(sizeof(double) * (size_t)(w > h ? w : h));
  if (!t) return 0;
  double nn = 0.0;
  for (int j = 0; j < w; ++j) {                                      /* v = l2normalize(W^T u), spectral.py:30 */
    double a = 0.0;
    for (int i = 0; i < h; ++i) a += (double)wb[(int64_t)i * w + j] * (double)u[i];
    t[j] = a;
    nn += a * a;
  }
  nn = sqrt(nn) + 1e-12;
  for (int j = 0; j < w; ++j) v[j] = (float)(t[j] / nn);
  double sn = 0.0;
  for (int i = 0; i < h; ++i) {                                      /* u = l2normalize(W v), spectral.py:31 */
    double a = 0.0;
    for (int j = 0; j < w; ++j) a += (double)wb[(int64_t)i * w + j] * (double)v[j];
    t[i] = a;
    sn += a * a;
  }
  sn = sqrt(sn) + 1e-12;
  double sg = 0.0;
  for (int i = 0; i < h; ++i) {
    u[i] = (float)(t[i] / sn);
    sg += (double)u[i] * t[i];                                       /* sigma = u . (W v), spectral.py:34 */
  }
  sigma[0] = (float)sg;
  if (w_out)
    for (int64_t q = 0; q < (int64_t)h * w; ++q) w_out[q] = wb[q] / sigma[0];   /* spectral.py:35 */
  free(t);
  return 1;
}

int skd_spectral_norm_backward(int h, int w, const float *wb, const float *u, const float *v,
                               const float *sigma, const float *gw, float *gwb, float *ws, stream_t st) {
  (void)ws; (void)st;
  if (h <= 0 || w <= 0 || !wb || !u || !v || !sigma || !gw || !gwb) return 0;
  /* w = wb / s, s = u^T wb v (u, v constants): gwb = gw/s - (sum(gw*wb)/s^2) u v^T */
  double dot = 0.0;
  for (int64_t q = 0; q < (int64_t)h * w; ++q) dot += (double)gw[q] * (double)wb[q];
  const double s = (double)sigma[0], coef = dot / (s * s);
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j)
      gwb[(int64_t)i * w + j] = (float)((double)gw[(int64_t)i * w + j] / s - coef * (double)u[i] * (double)v[j]);
  return 1;
}

/* all layers of a network in one call (the product runs 3 / 2 launches for all of them): layer by layer here */
int skd_spectral_norm_forward_multi(int L, const int *h, const int *w, const float *const *wb, float *const *u, float *const *v,
                                    float *const *sigma, float *const *w_out, float *ws, stream_t st) {
  if (L <= 0 || L > 8 || !h || !w || !wb || !u || !v || !sigma) return 0;
  for (int k = 0; k < L; ++k)
    if (!skd_spectral_norm_forward(h[k], w[k], wb[k], u[k], v[k], sigma[k], w_out ? w_out[k] : NULL, ws, st)) return 0;
  return 1;
}

int skd_spectral_norm_backward_multi(int L, const int *h, const int *w, const float *const *wb, const float *const *u,
                                     const float *const *v, const float *const *sigma, const float *const *gw, float *const *gwb,
                                     float *ws, stream_t st) {
  if (L <= 0 || L > 8 || !h || !w || !wb || !u || !v || !sigma || !gw || !gwb) return 0;
  for (int k = 0; k < L; ++k)
    if (!skd_spectral_norm_backward(h[k], w[k], wb[k], u[k], v[k], sigma[k], gw[k], gwb[k], ws, st)) return 0;
  return 1;
}

/* ---- CriterionDSN: bilinear upsample (align_corners) + cross-entropy(ignore_index), utils/criterion.py:179-188 ---- */
int64_t skd_ce_dsn_workspace_floats(int B, int C, int h, int w, int H, int W) {
  (void)B; (void)C; (void)h; (void)w; (void)H; (void)W;
  return 8;
}

/* PyTorch upsample_bilinear2d, align_corners=True: scale = (in-1)/(out-1) in float; src = scale*dst;
 * i0 = (int)src; i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1 */
static void tap_of(int dst, float scale, int in, int *i0, int *i1, float *l0, float *l1) {
  const float src = scale * (float)dst;
  *i0 = (int)src;
  if (*i0 > in - 1) *i0 = in - 1;
  *i1 = *i0 + (*i0 < in - 1 ? 1 : 0);
  *l1 = src - (float)*i0;
  *l0 = 1.f - *l1;
}

int skd_ce_dsn_forward(int B, int C, int h, int w, int H, int W, const float *lm, const float *ld,
                       const int64_t *target, int ignore_index, float aux_weight, float *loss,
                       float *gm, float *gd, float *ws, stream_t st) {
  (void)ws; (void)st;
  if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || !lm || !target || !loss) return 0;
  if (gd && !ld) return 0;
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int heads = ld ? 2 : 1, hw = h * w;
  double *acc = (double *)calloc((size_t)heads * B * C * hw, sizeof(double));
  double *v = (double *)malloc(sizeof(double) * (size_t)C);
  if (!acc || !v) return 0;
  double sum[2] = {0.0, 0.0}, cnt = 0.0;
  for (int b = 0; b < B; ++b)
    for (int Y = 0; Y < H; ++Y) {
      int y0, y1; float ly0, ly1;
      tap_of(Y, sy, h, &y0, &y1, &ly0, &ly1);
      for (int X = 0; X < W; ++X) {
        const int64_t t = target[((int64_t)b * H + Y) * W + X];
        if (t == (int64_t)ignore_index) continue;
        if (t < 0 || t >= C) { cnt = NAN; continue; }              /* F.cross_entropy would assert: poison the call */
        int x0, x1; float lx0, lx1;
        tap_of(X, sx, w, &x0, &x1, &lx0, &lx1);
        cnt += 1.0;
        const double w00 = (double)ly0 * lx0, w01 = (double)ly0 * lx1, w10 = (double)ly1 * lx0, w11 = (double)ly1 * lx1;
        for (int head = 0; head < heads; ++head) {
          const float *p = (head == 0 ? lm : ld) + (int64_t)b * C * hw;
          double mx = -INFINITY, z = 0.0;
          for (int c = 0; c < C; ++c) {
            const float *q = p + (int64_t)c * hw;
            v[c] = w00 * q[y0 * w + x0] + w01 * q[y0 * w + x1] + w10 * q[y1 * w + x0] + w11 * q[y1 * w + x1];
            if (v[c] > mx) mx = v[c];
          }
          for (int c = 0; c < C; ++c) z += exp(v[c] - mx);
          sum[head] += log(z) - (v[t] - mx);                      /* -log_softmax(up)[target] */
          double *a = acc + ((int64_t)head * B + b) * C * hw;
          for (int c = 0; c < C; ++c) {
            const double g = exp(v[c] - mx) / z - (c == (int)t ? 1.0 : 0.0);
            a[(int64_t)c * hw + y0 * w + x0] += w00 * g;
            a[(int64_t)c * hw + y0 * w + x1] += w01 * g;
            a[(int64_t)c * hw + y1 * w + x0] += w10 * g;
            a[(int64_t)c * hw + y1 * w + x1] += w11 * g;
          }
        }
      }
    }
  loss[0] = (float)(sum[0] / cnt + (ld ? (double)aux_weight * sum[1] / cnt : 0.0));   /* criterion.py:188 */
  for (int64_t i = 0; i < (int64_t)B * C * hw; ++i) {
    if (gm) gm[i] = (float)(acc[i] / cnt);
    if (gd) gd[i] = (float)((double)aux_weight * acc[(int64_t)B * C * hw + i] / cnt);
  }
  free(acc);
  free(v);
  return 1;
}

/* ---- pyramid pooling module, networks/pspnet_combine.py:86-112 ------------------------------------ */
static int bin_start(int i, int n, int s) { return (i * n) / s; }                 /* adaptive_avg_pool2d */
static int bin_end(int i, int n, int s) { return ((i + 1) * n + s - 1) / s; }

int64_t skd_ppm_pooled_floats(int planes, int nsizes, const int *sizes) {
  int64_t bins = 0;
  if (planes <= 0 || nsizes <= 0 || nsizes > 4 || !sizes) return 0;
  for (int k = 0; k < nsizes; ++k) bins += (int64_t)sizes[k] * sizes[k];
  return planes * bins;
}

int skd_ppm_pool(int planes, int H, int W, int nsizes, const int *sizes, const float *x, float *pooled, stream_t st) {
  (void)st;
  if (planes <= 0 || H <= 0 || W <= 0 || nsizes <= 0 || nsizes > 4 || !sizes || !x || !pooled) return 0;
  int64_t off = 0;
  for (int k = 0; k < nsizes; ++k) {
    const int s = sizes[k];
    for (int64_t p = 0; p < planes; ++p)
      for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j) {
          const int h0 = bin_start(i, H, s), h1 = bin_end(i, H, s), w0 = bin_start(j, W, s), w1 = bin_end(j, W, s);
          double a = 0.0;
          for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) a += x[p * (int64_t)H * W + (int64_t)h * W + w];
          pooled[off + p * (s * s) + i * s + j] = (float)(a / ((h1 - h0) * (w1 - w0)));
        }
    off += (int64_t)planes * s * s;
  }
  return 1;
}

int skd_ppm_pool_backward(int planes, int H, int W, int nsizes, const int *sizes, const float *g, float *dx, stream_t st) {
  (void)st;
  if (planes <= 0 || H <= 0 || W <= 0 || nsizes <= 0 || nsizes > 4 || !sizes || !g || !dx) return 0;
  double *acc = (double *)calloc((size_t)planes * H * W, sizeof(double));
  if (!acc) return 0;
  int64_t off = 0;
  for (int k = 0; k < nsizes; ++k) {
    const int s = sizes[k];
    for (int64_t p = 0; p < planes; ++p)
      for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j) {
          const int h0 = bin_start(i, H, s), h1 = bin_end(i, H, s), w0 = bin_start(j, W, s), w1 = bin_end(j, W, s);
          const double v = (double)g[off + p * (s * s) + i * s + j] / ((h1 - h0) * (w1 - w0));
          for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) acc[p * (int64_t)H * W + (int64_t)h * W + w] += v;
        }
    off += (int64_t)planes * s * s;
  }
  for (int64_t e = 0; e < (int64_t)planes * H * W; ++e) dx[e] = (float)acc[e];
  free(acc);
  return 1;
}

int skd_ppm_concat(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes,
                   const float *const *priors, const float *feats, float *cat, stream_t st) {
  (void)st;
  if (B <= 0 || Cout <= 0 || Cfeat < 0 || H <= 0 || W <= 0 || nsizes <= 0 || nsizes > 4 || !sizes || !priors || !cat) return 0;
  const int Ctot = nsizes * Cout + Cfeat, HW = H * W;
  for (int b = 0; b < B; ++b) {
    for (int k = 0; k < nsizes; ++k) {
      const int s = sizes[k];
      const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
      for (int c = 0; c < Cout; ++c) {
        const float *src = priors[k] + ((int64_t)b * Cout + c) * (s * s);
        float *out = cat + ((int64_t)b * Ctot + k * Cout + c) * HW;
        for (int Y = 0; Y < H; ++Y) {
          int y0, y1; float ly0, ly1;
          tap_of(Y, sy, s, &y0, &y1, &ly0, &ly1);
          for (int X = 0; X < W; ++X) {
            int x0, x1; float lx0, lx1;
            tap_of(X, sx, s, &x0, &x1, &lx0, &lx1);
            out[Y * W + X] = (float)((double)ly0 * ((double)lx0 * src[y0 * s + x0] + (double)lx1 * src[y0 * s + x1]) +
                                     (double)ly1 * ((double)lx0 * src[y1 * s + x0] + (double)lx1 * src[y1 * s + x1]));
          }
        }
      }
    }
    for (int c = 0; c < Cfeat; ++c)
      memcpy(cat + ((int64_t)b * Ctot + nsizes * Cout + c) * HW, feats + ((int64_t)b * Cfeat + c) * HW, sizeof(float) * (size_t)HW);
  }
  return 1;
}

int skd_ppm_concat_backward(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes,
                            const float *gcat, float *const *gpriors, stream_t st) {
  (void)st;
  if (B <= 0 || Cout <= 0 || Cfeat < 0 || H <= 0 || W <= 0 || nsizes <= 0 || nsizes > 4 || !sizes || !gcat || !gpriors) return 0;
  const int Ctot = nsizes * Cout + Cfeat, HW = H * W;
  for (int k = 0; k < nsizes; ++k) {
    const int s = sizes[k];
    const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
    double *acc = (double *)malloc(sizeof(double) * (size_t)s * s);
    if (!acc) return 0;
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < Cout; ++c) {
        const float *g = gcat + ((int64_t)b * Ctot + k * Cout + c) * HW;
        for (int q = 0; q < s * s; ++q) acc[q] = 0.0;
        for (int Y = 0; Y < H; ++Y) {
          int y0, y1; float ly0, ly1;
          tap_of(Y, sy, s, &y0, &y1, &ly0, &ly1);
          for (int X = 0; X < W; ++X) {
            int x0, x1; float lx0, lx1;
            tap_of(X, sx, s, &x0, &x1, &lx0, &lx1);
            const double v = g[Y * W + X];
            acc[y0 * s + x0] += (double)ly0 * lx0 * v;
            acc[y0 * s + x1] += (double)ly0 * lx1 * v;
            acc[y1 * s + x0] += (double)ly1 * lx0 * v;
            acc[y1 * s + x1] += (double)ly1 * lx1 * v;
          }
        }
        for (int q = 0; q < s * s; ++q) gpriors[k][((int64_t)b * Cout + c) * (s * s) + q] = (float)acc[q];
      }
    free(acc);
  }
  return 1;
}

/* ---- channels-last forms of the pyramid entries: layout conversion around the NCHW restatements above ---- */
static float *nhwc_to_nchw(int B, int C, int H, int W, const float *x) {      /* (B,H,W,C) -> new (B,C,H,W) */
  float *p = (float *)malloc(sizeof(float) * (size_t)B * C * H * W + 4);
  if (p && x)
    for (int b = 0; b < B; ++b)
      for (int64_t e = 0; e < (int64_t)H * W; ++e)
        for (int c = 0; c < C; ++c) p[((int64_t)b * C + c) * H * W + e] = x[(((int64_t)b * H * W) + e) * C + c];
  return p;
}
static void nchw_to_nhwc(int B, int C, int H, int W, const float *p, float *x) {
  for (int b = 0; b < B; ++b)
    for (int64_t e = 0; e < (int64_t)H * W; ++e)
      for (int c = 0; c < C; ++c) x[(((int64_t)b * H * W) + e) * C + c] = p[((int64_t)b * C + c) * H * W + e];
}

int64_t skd_ppm_nhwc_workspace_floats(int B, int C, int Cout, int H, int W, int nsizes, const int *sizes) {
  (void)C; (void)Cout; (void)H; (void)W;
  return (B > 0 && nsizes > 0 && nsizes <= 4 && sizes) ? 1 : 0;
}

int skd_ppm_pool_nhwc(int B, int C, int H, int W, int nsizes, const int *sizes, const float *x, float *pooled, float *ws,
                      stream_t st) {
  (void)ws;
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || !x || !pooled || nsizes <= 0 || nsizes > 4 || !sizes) return 0;
  float *px = nhwc_to_nchw(B, C, H, W, x);
  float *pp = (float *)malloc(sizeof(float) * (size_t)skd_ppm_pooled_floats(B * C, nsizes, sizes));
  int r = (px && pp) ? skd_ppm_pool(B * C, H, W, nsizes, sizes, px, pp, st) : 0;
  int64_t off = 0;
  for (int k = 0; r && k < nsizes; ++k) {
    const int s = sizes[k];
    nchw_to_nhwc(B, C, s, s, pp + off, pooled + off);
    off += (int64_t)B * C * s * s;
  }
  free(px); free(pp);
  return r;
}

int skd_ppm_pool_backward_nhwc(int B, int C, int H, int W, int nsizes, const int *sizes, const float *g, float *dx, stream_t st) {
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || !g || !dx || nsizes <= 0 || nsizes > 4 || !sizes) return 0;
  float *pg = (float *)malloc(sizeof(float) * (size_t)skd_ppm_pooled_floats(B * C, nsizes, sizes));
  float *pd = (float *)malloc(sizeof(float) * (size_t)B * C * H * W);
  if (!pg || !pd) { free(pg); free(pd); return 0; }
  int64_t off = 0;
  for (int k = 0; k < nsizes; ++k) {
    const int s = sizes[k];
    float *t = nhwc_to_nchw(B, C, s, s, g + off);
    if (!t) { free(pg); free(pd); return 0; }
    memcpy(pg + off, t, sizeof(float) * (size_t)B * C * s * s);
    free(t);
    off += (int64_t)B * C * s * s;
  }
  const int r = skd_ppm_pool_backward(B * C, H, W, nsizes, sizes, pg, pd, st);
  if (r) nchw_to_nhwc(B, C, H, W, pd, dx);
  free(pg); free(pd);
  return r;
}

int skd_ppm_concat_nhwc(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes, const float *const *priors,
                        const float *feats, float *cat, stream_t st) {
  if (B <= 0 || Cout <= 0 || (Cout & 3) || Cfeat < 0 || (Cfeat & 3) || H <= 0 || W <= 0 || !priors || !cat) return 0;
  if (nsizes <= 0 || nsizes > 4 || !sizes || (Cfeat > 0 && !feats)) return 0;
  const float *pp[4] = {0, 0, 0, 0};
  float *own[4] = {0, 0, 0, 0};
  for (int k = 0; k < nsizes; ++k) {
    if (!priors[k]) return 0;
    own[k] = nhwc_to_nchw(B, Cout, sizes[k], sizes[k], priors[k]);
    pp[k] = own[k];
  }
  const int Ctot = nsizes * Cout + Cfeat;
  float *pf = Cfeat > 0 ? nhwc_to_nchw(B, Cfeat, H, W, feats) : NULL;
  float *pc = (float *)malloc(sizeof(float) * (size_t)B * Ctot * H * W);
  const int r = pc ? skd_ppm_concat(B, Cout, Cfeat, H, W, nsizes, sizes, pp, pf, pc, st) : 0;
  if (r) nchw_to_nhwc(B, Ctot, H, W, pc, cat);
  for (int k = 0; k < 4; ++k) free(own[k]);
  free(pf); free(pc);
  return r;
}

int skd_ppm_concat_backward_nhwc(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes, const float *gcat,
                                 float *const *gpriors, float *gfeats, float *ws, stream_t st) {
  (void)ws;
  if (B <= 0 || Cout <= 0 || (Cout & 3) || Cfeat < 0 || (Cfeat & 3) || H <= 0 || W <= 0 || !gcat) return 0;
  if (nsizes <= 0 || nsizes > 4 || !sizes) return 0;
  const int Ctot = nsizes * Cout + Cfeat;
  if (gfeats && Cfeat > 0)
    for (int64_t e = 0; e < (int64_t)B * H * W; ++e)
      memcpy(gfeats + e * Cfeat, gcat + e * Ctot + nsizes * Cout, sizeof(float) * (size_t)Cfeat);
  if (!gpriors) return 1;
  float *pc = nhwc_to_nchw(B, Ctot, H, W, gcat);
  float *own[4] = {0, 0, 0, 0};
  for (int k = 0; k < nsizes; ++k) own[k] = (float *)malloc(sizeof(float) * (size_t)B * Cout * sizes[k] * sizes[k]);
  const int r = pc ? skd_ppm_concat_backward(B, Cout, Cfeat, H, W, nsizes, sizes, pc, own, st) : 0;
  for (int k = 0; k < nsizes; ++k) {
    if (r && gpriors[k]) nchw_to_nhwc(B, Cout, sizes[k], sizes[k], own[k], gpriors[k]);
    free(own[k]);
  }
  free(pc);
  return r;
}

/* ---- pyramid priors folded through the 3x3 bottleneck convolution (include/skd.h section 8; the identity
 * conv3x3(cat(up(priors), feats)) = conv3x3(feats) + fold(Z) of networks/pspnet_combine.py:104-111), evaluated here
 * by its definition: every tap reads the bilinear interpolation of its own Z map at the shifted position. ---- */
int64_t skd_ppm_fold_nhwc_workspace_floats(int B, int Cout, int H, int W, int nsizes, const int *sizes) {
  (void)H; (void)W; (void)sizes;
  if (B <= 0 || Cout <= 0 || nsizes <= 0) return 0;
  return 4;
}

int skd_ppm_fold_nhwc(int B, int Cout, int H, int W, int nsizes, const int *sizes, const float *const *z, int64_t ldz,
                      float *out, stream_t st) {
  (void)st;
  if (ldz < 9 * (int64_t)Cout) return 0;
  if (B <= 0 || Cout <= 0 || (Cout & 3) || H <= 0 || W <= 0 || !z || !out || nsizes <= 0 || nsizes > 4 || !sizes) return 0;
  for (int k = 0; k < nsizes; ++k)
    if (!z[k] || sizes[k] <= 0) return 0;
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float *o = out + (((int64_t)b * H + y) * W + x) * Cout;
        for (int c = 0; c < Cout; ++c) {
          double acc = (double)o[c];
          for (int k = 0; k < nsizes; ++k) {
            const int s = sizes[k];
            const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
            const float *zk = z[k] + (int64_t)b * s * s * ldz;
            for (int ty = 0; ty < 3; ++ty)
              for (int tx = 0; tx < 3; ++tx) {
                const int yp = y + ty - 1, xp = x + tx - 1;
                if (yp < 0 || yp >= H || xp < 0 || xp >= W) continue;
                int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
                tap_of(yp, sy, s, &y0, &y1, &ly0, &ly1);
                tap_of(xp, sx, s, &x0, &x1, &lx0, &lx1);
                const int tap = ty * 3 + tx;
#define ZAT(jy, jx) ((double)zk[((int64_t)(jy) * s + (jx)) * ldz + tap * Cout + c])
                acc += (double)ly0 * ((double)lx0 * ZAT(y0, x0) + (double)lx1 * ZAT(y0, x1)) +
                       (double)ly1 * ((double)lx0 * ZAT(y1, x0) + (double)lx1 * ZAT(y1, x1));
#undef ZAT
              }
          }
          o[c] = (float)acc;
        }
      }
  return 1;
}

int skd_ppm_fold_backward_nhwc(int B, int Cout, int H, int W, int nsizes, const int *sizes, const float *gout,
                               float *const *gz, int64_t ldz, float *ws, stream_t st) {
  (void)st; (void)ws;
  if (ldz < 9 * (int64_t)Cout) return 0;
  if (B <= 0 || Cout <= 0 || (Cout & 3) || H <= 0 || W <= 0 || !gout || !gz || nsizes <= 0 || nsizes > 4 || !sizes) return 0;
  for (int k = 0; k < nsizes; ++k) {
    if (!gz[k] || sizes[k] <= 0) return 0;
    const int s = sizes[k];
    const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
    const int64_t n = (int64_t)B * s * s * 9 * Cout;
    double *acc = (double *)calloc((size_t)n, sizeof(double));
    if (!acc) return 0;
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          const float *g = gout + (((int64_t)b * H + y) * W + x) * Cout;
          for (int ty = 0; ty < 3; ++ty)
            for (int tx = 0; tx < 3; ++tx) {
              const int yp = y + ty - 1, xp = x + tx - 1;
              if (yp < 0 || yp >= H || xp < 0 || xp >= W) continue;
              int yy[2], xx[2]; float ly[2], lx[2];
              tap_of(yp, sy, s, &yy[0], &yy[1], &ly[0], &ly[1]);
              tap_of(xp, sx, s, &xx[0], &xx[1], &lx[0], &lx[1]);
              for (int a = 0; a < 2; ++a)
                for (int d = 0; d < 2; ++d) {
                  double *dst = acc + ((((int64_t)b * s + yy[a]) * s + xx[d]) * 9 + ty * 3 + tx) * Cout;
                  const double wt = (double)ly[a] * (double)lx[d];
                  for (int c = 0; c < Cout; ++c) dst[c] += wt * (double)g[c];
                }
            }
        }
    for (int64_t r = 0; r < (int64_t)B * s * s; ++r)
      for (int i = 0; i < 9 * Cout; ++i) gz[k][r * ldz + i] = (float)acc[r * 9 * Cout + i];
    free(acc);
  }
  return 1;
}

/* ---- stem max-pool 3x3 / stride 2 / padding 1 (ceil_mode shapes given by the caller), channels-last; PyTorch's
 * max_pool2d_with_indices rule (val > max || isnan(val), row-major scan from the first valid position),
 * networks/pspnet_combine.py:135,152 ---- */
static int pool_geom_ok(int B, int C, int H, int W, int OH, int OW) {
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return 0;
  return 2 * (OH - 1) - 1 < H && 2 * (OW - 1) - 1 < W && 2 * (OH - 1) + 1 >= H - 1 && 2 * (OW - 1) + 1 >= W - 1;
}

int skd_maxpool3x3s2_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, float *y, uint8_t *arg, stream_t st) {
  (void)st;
  if (!pool_geom_ok(B, C, H, W, OH, OW) || !x || !y) return 0;
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < OH; ++oy)
      for (int ox = 0; ox < OW; ++ox)
        for (int c = 0; c < C; ++c) {
          const int ys = 2 * oy - 1, xs = 2 * ox - 1;
          const int y0 = ys < 0 ? 0 : ys, x0 = xs < 0 ? 0 : xs;
          const int y1 = ys + 3 < H ? ys + 3 : H, x1 = xs + 3 < W ? xs + 3 : W;
          float best = -INFINITY;
          int a = (y0 - ys) * 3 + (x0 - xs);
          for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) {
              const float v = x[(((int64_t)b * H + yy) * W + xx) * C + c];
              if (v > best || v != v) { best = v; a = (yy - ys) * 3 + (xx - xs); }
            }
          const int64_t o = (((int64_t)b * OH + oy) * OW + ox) * C + c;
          y[o] = best;
          if (arg) arg[o] = (uint8_t)a;
        }
  return 1;
}

int skd_maxpool3x3s2_backward_nhwc(int B, int C, int H, int W, int OH, int OW, const float *dy, const uint8_t *arg, float *dx,
                                   stream_t st) {
  (void)st;
  if (!pool_geom_ok(B, C, H, W, OH, OW) || !dy || !arg || !dx) return 0;
  double *acc = (double *)calloc((size_t)B * H * W * C, sizeof(double));
  if (!acc) return 0;
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < OH; ++oy)
      for (int ox = 0; ox < OW; ++ox)
        for (int c = 0; c < C; ++c) {
          const int64_t o = (((int64_t)b * OH + oy) * OW + ox) * C + c;
          const int yy = 2 * oy - 1 + arg[o] / 3, xx = 2 * ox - 1 + arg[o] % 3;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) { free(acc); return 0; }
          acc[(((int64_t)b * H + yy) * W + xx) * C + c] += (double)dy[o];
        }
  for (int64_t i = 0; i < (int64_t)B * H * W * C; ++i) dx[i] = (float)acc[i];
  free(acc);
  return 1;
}

/* ---- evaluation tail: upsample + argmax + confusion matrix, networks/evaluate.py:106-113, 136-154, 186-198 ---- */
int skd_seg_confusion(int B, int C, int h, int w, int H, int W, const float *logits, const int64_t *target,
                      int ignore_index, uint8_t *pred, int64_t *confusion, stream_t st) {
  (void)st;
  if (B <= 0 || C <= 0 || C > 64 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || !logits) return 0;
  if (target && !confusion) return 0;
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int hw = h * w;
  for (int b = 0; b < B; ++b)
    for (int Y = 0; Y < H; ++Y) {
      int y0, y1; float ly0, ly1;
      tap_of(Y, sy, h, &y0, &y1, &ly0, &ly1);
      for (int X = 0; X < W; ++X) {
        int x0, x1; float lx0, lx1;
        tap_of(X, sx, w, &x0, &x1, &lx0, &lx1);
        const float *p = logits + (int64_t)b * C * hw;
        float best = 0.f;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
          const float *q = p + (int64_t)c * hw;
          /* upsample_bilinear2d: h0l*(w0l*v00 + w1l*v01) + h1l*(w0l*v10 + w1l*v11), every op rounded to float */
          const float v = ly0 * (lx0 * q[y0 * w + x0] + lx1 * q[y0 * w + x1]) + ly1 * (lx0 * q[y1 * w + x0] + lx1 * q[y1 * w + x1]);
          if (c == 0 || v > best) { best = v; arg = c; }               /* np.argmax: first maximum */
        }
        const int64_t pix = ((int64_t)b * H + Y) * W + X;
        if (pred) pred[pix] = (uint8_t)arg;
        if (target) {
          const int64_t t = target[pix];
          if (t != (int64_t)ignore_index && t >= 0 && t < C) confusion[t * C + arg] += 1;   /* evaluate.py:144-152 */
        }
      }
    }
  return 1;
}

/* ---- the 19-class 1x1 classifier heads on channels-last feature maps (include/skd.h section 14; networks/pspnet_combine.py:138-154):
 *      plain loops, double accumulation ---- */
int skd_head1x1_supported(int K, int C, int backward) {
  if (C <= 0 || C > 20 || K <= 0) return 0;
  return backward ? K == 128 : (K % 128 == 0 && K <= 1024);
}

int skd_head1x1_forward_nhwc(int B, int HW, int K, int C, const float *x, const float *w, const float *bias, float *out, stream_t st) {
  (void)st;
  if (B <= 0 || HW <= 0 || !skd_head1x1_supported(K, C, 0) || !x || !w || !out) return 0;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int p = 0; p < HW; ++p) {
        const float *xr = x + ((int64_t)b * HW + p) * K;
        double a = 0.0;
        for (int k = 0; k < K; ++k) a += (double)xr[k] * (double)w[(int64_t)c * K + k];
        out[((int64_t)b * C + c) * HW + p] = (float)(a + (bias ? (double)bias[c] : 0.0));
      }
  return 1;
}

int64_t skd_head1x1_backward_workspace_floats(int B, int HW, int K, int C) {
  (void)B; (void)HW; (void)K; (void)C;
  return 1;
}

int skd_head1x1_backward_nhwc(int B, int HW, int K, int C, const float *x, const float *w, const float *gout, float *gx, float *gw,
                              float *gb, float *workspace, stream_t st) {
  (void)st;
  if (B <= 0 || HW <= 0 || !skd_head1x1_supported(K, C, 1) || !w || !gout || !workspace) return 0;
  if (gw && !x) return 0;
  const int64_t M = (int64_t)B * HW;
  if (gx)
    for (int64_t m = 0; m < M; ++m) {
      const int64_t b = m / HW, p = m - b * HW;
      for (int k = 0; k < K; ++k) {
        double a = 0.0;
        for (int c = 0; c < C; ++c) a += (double)gout[(b * C + c) * HW + p] * (double)w[(int64_t)c * K + k];
        gx[m * K + k] = (float)a;
      }
    }
  for (int c = 0; c < C; ++c) {
    double sb = 0.0;
    for (int64_t m = 0; m < M; ++m) sb += (double)gout[((m / HW) * C + c) * HW + (m % HW)];
    if (gb) gb[c] = (float)sb;
    if (gw)
      for (int k = 0; k < K; ++k) {
        double a = 0.0;
        for (int64_t m = 0; m < M; ++m) a += (double)gout[((m / HW) * C + c) * HW + (m % HW)] * (double)x[m * K + k];
        gw[(int64_t)c * K + k] = (float)a;
      }
  }
  return 1;
}
