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
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < M; ++j) {
        double at = 0.0, as = 0.0;                                   /* einsum('icm,icn->imn'), utils.py:178 */
        for (int c = 0; c < Ct; ++c) at += (double)ft[((int64_t)b * Ct + c) * ldm + i] * ft[((int64_t)b * Ct + c) * ldm + j];
        for (int c = 0; c < Cs; ++c) as += (double)fs[((int64_t)b * Cs + c) * ldm + i] * fs[((int64_t)b * Cs + c) * ldm + j];
        const double g = at - as;
        total += g * g;
        if (G) G[((int64_t)b * ldm + i) * ldm + j] = (float)g;
      }
  loss[0] = (float)(total / ((double)M * (double)M) / (double)B);    /* utils.py:181 */
  return 1;
}

int skd_pairwise_backward(int B, int Cs, int M, int ldm, int ldc, const float *fst, const float *G,
                          const float *norm_s, const float *grad_loss, float *dpooled, stream_t st) {
  (void)st;
  if (B <= 0 || Cs <= 0 || M <= 0 || !fst || !G || !norm_s || !grad_loss || !dpooled) return 0;
  /* L = sum G^2/(M^2 B), G = A_T - A_S, A_S = Fh^T Fh  =>  dL/dFh = -4/(M^2 B) Fh G ; dP = dFh / norm */
  const double coef = -4.0 / ((double)M * (double)M * (double)B) * (double)grad_loss[0];
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < Cs; ++c)
      for (int m = 0; m < ldm; ++m) {
        double acc = 0.0;
        if (m < M) {
          for (int n = 0; n < M; ++n)
            acc += (double)fst[((int64_t)b * ldm + n) * ldc + c] * (double)G[((int64_t)b * ldm + n) * ldm + m];
          acc = acc * coef / (double)norm_s[(int64_t)b * M + m];
        }
        dpooled[((int64_t)b * Cs + c) * ldm + m] = (float)acc;
      }
  return 1;
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
