/*
 * oracle/abn_ref.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C, host-memory restatement of the reference's native InPlace-ABN kernels and of the
 * op sequence its autograd functions run around them.  Same C ABI as the InPlace-ABN part of
 * include/skd.h, but the pointers are HOST pointers and `stream` is ignored, so the very same
 * ctypes call sites can be checked against it (and tests can swap it in as a CPU double).
 *
 * Follows:
 *   K1 mean_var_kernel      libs/src/bn.cu:125-138  (two passes: mean, then biased variance)
 *   K2 forward_kernel       libs/src/bn.cu:140-165
 *   K3 edz_eydz_kernel      libs/src/bn.cu:167-184
 *   K4 backward_kernel      libs/src/bn.cu:186-232
 *   K5-K9 activations       libs/src/bn.cu:302-377
 *   op order / running stats  libs/functions.py:70-162
 * Reductions accumulate in double (the reference's float tree order is not reproducible and
 * not part of its contract); element-wise arithmetic is float, in the reference's order.
 * Built by oracle/Makefile into oracle/libabn_ref.so.  Parity status: unpinned by any
 * reference-run output (bn.cu cannot be built here); pinned against closed-form autograd.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ACT_NONE 0
#define ACT_LEAKY 1
#define ACT_ELU 2
#define ACT_RELU 3 /* nn.ReLU applied to the BatchNorm2d output, networks/pspnet_combine.py:36,68,72 */

typedef void *stream_t;

static float gamma_of(const float *w, int c, float eps) { return w ? fabsf(w[c]) + eps : 1.f; }
static float beta_of(const float *b, int c) { return b ? b[c] : 0.f; }
static float inv_std_of(float var, float eps) {
  float r = 0.f;
  if (var != 0.f || eps != 0.f) r = 1.f / sqrtf(var + eps);
  return r;
}

int skd_abi_version(void) { return 1; }
int skd_target_arch(void) { return 0; /* host */ }

/* ---- K1 ---- */
int skd_bn_mean_var(int N, int C, int S, const float *x, float *mean, float *var, stream_t st) {
  (void)st;
  if (N <= 0 || C <= 0 || S <= 0) return 0;
  const double norm = 1.0 / ((double)N * (double)S);
  for (int c = 0; c < C; ++c) {
    double s = 0.0;
    for (int n = 0; n < N; ++n) {
      const float *row = x + ((int64_t)n * C + c) * S;
      for (int i = 0; i < S; ++i) s += row[i];
    }
    const float m = (float)(s * norm);
    double v = 0.0;
    for (int n = 0; n < N; ++n) {
      const float *row = x + ((int64_t)n * C + c) * S;
      for (int i = 0; i < S; ++i) {
        const double d = (double)row[i] - (double)m;
        v += d * d;
      }
    }
    mean[c] = m;
    var[c] = (float)(v * norm);
  }
  return 1;
}

/* ---- K2 ---- */
int skd_bn_forward(int N, int C, int S, const float *x, const float *mean, const float *var,
                   const float *weight, const float *bias, float *y, float *z, float eps, stream_t st) {
  (void)st;
  if (N <= 0 || C <= 0 || S <= 0) return 0;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      const float m = mean[c], is = inv_std_of(var[c], eps);
      const float g = gamma_of(weight, c, eps), b = beta_of(bias, c);
      const int64_t off = ((int64_t)n * C + c) * S;
      for (int i = 0; i < S; ++i) {
        const float yy = (x[off + i] - m) * is;
        const float zz = yy * g + b;
        y[off + i] = yy;
        z[off + i] = zz; /* y and z may alias: z wins, as in the reference */
      }
    }
  return 1;
}

/* ---- K3 ---- */
int skd_bn_edz_eydz(int N, int C, int S, const float *z, const float *dz, const float *weight,
                    const float *bias, float *edz, float *eydz, float eps, stream_t st) {
  (void)st;
  if (N <= 0 || C <= 0 || S <= 0) return 0;
  const double norm = 1.0 / ((double)N * (double)S);
  for (int c = 0; c < C; ++c) {
    const float g = gamma_of(weight, c, eps), b = beta_of(bias, c);
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n) {
      const int64_t off = ((int64_t)n * C + c) * S;
      for (int i = 0; i < S; ++i) {
        const float yy = (z[off + i] - b) / g;
        s1 += dz[off + i];
        s2 += (double)(yy * dz[off + i]);
      }
    }
    edz[c] = (float)(s1 * norm);
    eydz[c] = (float)(s2 * norm);
  }
  return 1;
}

/* ---- K4 ---- */
int skd_bn_backward(int N, int C, int S, const float *dz, const float *z, const float *var,
                    const float *weight, const float *bias, const float *edz, const float *eydz,
                    float *dx, float *dweight, float *dbias, float eps, stream_t st) {
  (void)st;
  if (N <= 0 || C <= 0 || S <= 0) return 0;
  for (int c = 0; c < C; ++c) {
    const float g = gamma_of(weight, c, eps), b = beta_of(bias, c);
    if (dx) {
      const float mul = g * inv_std_of(var[c], eps);
      for (int n = 0; n < N; ++n) {
        const int64_t off = ((int64_t)n * C + c) * S;
        for (int i = 0; i < S; ++i) {
          const float yy = (z[off + i] - b) / g;
          dx[off + i] = (dz[off + i] - edz[c] - yy * eydz[c]) * mul;
        }
      }
    }
    const float norm = (float)N * (float)S;
    if (dweight) {
      if (weight[c] > 0.f)
        dweight[c] += eydz[c] * norm;
      else if (weight[c] < 0.f)
        dweight[c] -= eydz[c] * norm;
    }
    if (dbias) dbias[c] += edz[c] * norm;
  }
  return 1;
}

/* ---- K5-K9 ---- */
int skd_leaky_relu(int64_t n, float *x, float slope, stream_t st) {
  (void)st;
  for (int64_t i = 0; i < n; ++i)
    if (x[i] < 0.f) x[i] = x[i] * slope;
  return 1;
}
int skd_leaky_relu_backward(int64_t n, const float *x, float *dx, float slope, stream_t st) {
  (void)st;
  for (int64_t i = 0; i < n; ++i)
    if (x[i] < 0.f) dx[i] = dx[i] * slope;
  return 1;
}
int skd_elu(int64_t n, float *x, stream_t st) {
  (void)st;
  for (int64_t i = 0; i < n; ++i)
    if (x[i] < 0.f) x[i] = expf(x[i]) - 1.f;
  return 1;
}
int skd_elu_backward(int64_t n, const float *x, float *dx, stream_t st) {
  (void)st;
  for (int64_t i = 0; i < n; ++i)
    if (x[i] < 0.f) dx[i] = dx[i] * (x[i] + 1.f);
  return 1;
}
int skd_elu_inv(int64_t n, float *x, stream_t st) {
  (void)st;
  for (int64_t i = 0; i < n; ++i)
    if (x[i] < 0.f) x[i] = log1pf(x[i]);
  return 1;
}

/* ---- the fused entries, composed exactly like libs/functions.py does ---- */
int64_t skd_abn_workspace_floats(int N, int C, int S) {
  (void)N; (void)S;
  return C > 0 ? 2 * (int64_t)C : 0;
}

int skd_abn_stats(int N, int C, int S, const float *x, float *mean, float *var, float *ws, stream_t st) {
  (void)ws;
  return skd_bn_mean_var(N, C, S, x, mean, var, st);
}

int skd_abn_update_running(int C, float *rm, float *rv, const float *mean, const float *var,
                           float momentum, double n, stream_t st) {
  (void)st;
  const float nf = (float)n;
  for (int c = 0; c < C; ++c) { /* functions.py:90-91 */
    rm[c] = rm[c] * (1.f - momentum) + momentum * mean[c];
    rv[c] = rv[c] * (1.f - momentum) + momentum * var[c] * nf / (nf - 1.f);
  }
  return 1;
}

static void act_forward(int act, int64_t n, float *x, float slope) { /* functions.py:45-51 */
  if (act == ACT_LEAKY) skd_leaky_relu(n, x, slope, 0);
  else if (act == ACT_ELU) skd_elu(n, x, 0);
  else if (act == ACT_RELU)
    for (int64_t i = 0; i < n; ++i)
      if (x[i] < 0.f) x[i] = 0.f;
}

int skd_abn_apply(int N, int C, int S, float *x, const float *mean, const float *var,
                  const float *weight, const float *bias, float eps, int act, float slope, stream_t st) {
  if (!skd_bn_forward(N, C, S, x, mean, var, weight, bias, x, x, eps, st)) return 0;
  act_forward(act, (int64_t)N * C * S, x, slope);
  return 1;
}

/* out = bn(conv); out = out + residual; relu(out)   (networks/pspnet_combine.py:37-43, 78-82) */
int skd_abn_apply_residual(int N, int C, int S, float *x, const float *residual, const float *mean,
                           const float *var, const float *weight, const float *bias, float eps, int act,
                           float slope, stream_t st) {
  if (!residual) return 0;
  if (!skd_bn_forward(N, C, S, x, mean, var, weight, bias, x, x, eps, st)) return 0;
  for (int64_t i = 0; i < (int64_t)N * C * S; ++i) x[i] = x[i] + residual[i];
  act_forward(act, (int64_t)N * C * S, x, slope);
  return 1;
}

int skd_abn_forward_train(int N, int C, int S, float *x, const float *weight, const float *bias,
                          float *rm, float *rv, float *mean, float *var, float momentum, float eps,
                          int act, float slope, float *ws, stream_t st) {
  (void)ws;
  if (!skd_bn_mean_var(N, C, S, x, mean, var, st)) return 0;
  if (rm && rv) skd_abn_update_running(C, rm, rv, mean, var, momentum, (double)N * (double)S, st);
  return skd_abn_apply(N, C, S, x, mean, var, weight, bias, eps, act, slope, st);
}

/* undo the activation on private copies (functions.py:54-62 does it in place on z and dz) */
static int undo_act(int act, int64_t n, const float *z, const float *dz, float slope, float **zc,
                    float **dzc) {
  *zc = (float *)malloc(sizeof(float) * (size_t)n);
  *dzc = (float *)malloc(sizeof(float) * (size_t)n);
  if (!*zc || !*dzc) return 0;
  memcpy(*zc, z, sizeof(float) * (size_t)n);
  memcpy(*dzc, dz, sizeof(float) * (size_t)n);
  if (act == ACT_LEAKY) {
    skd_leaky_relu_backward(n, *zc, *dzc, slope, 0);
    skd_leaky_relu(n, *zc, 1.f / slope, 0);
  } else if (act == ACT_ELU) {
    skd_elu_backward(n, *zc, *dzc, 0);
    skd_elu_inv(n, *zc, 0);
  }
  return 1;
}

int skd_abn_backward_reduce(int N, int C, int S, const float *z, const float *dz, const float *weight,
                            const float *bias, float *edz, float *eydz, float eps, int act,
                            float slope, float *ws, stream_t st) {
  (void)ws;
  float *zc, *dzc;
  if (N <= 0 || C <= 0 || S <= 0 || act == ACT_RELU) return 0;
  if (!undo_act(act, (int64_t)N * C * S, z, dz, slope, &zc, &dzc)) return 0;
  const int r = skd_bn_edz_eydz(N, C, S, zc, dzc, weight, bias, edz, eydz, eps, st);
  free(zc);
  free(dzc);
  return r;
}

int skd_abn_backward_dx(int N, int C, int S, const float *z, const float *dz, const float *var,
                        const float *weight, const float *bias, const float *edz, const float *eydz,
                        float *dx, float *dweight, float *dbias, float eps, int act, float slope,
                        stream_t st) {
  float *zc, *dzc;
  if (N <= 0 || C <= 0 || S <= 0 || act == ACT_RELU) return 0;
  if (!undo_act(act, (int64_t)N * C * S, z, dz, slope, &zc, &dzc)) return 0;
  const int r = skd_bn_backward(N, C, S, dzc, zc, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, st);
  free(zc);
  free(dzc);
  return r;
}

int skd_abn_backward(int N, int C, int S, const float *z, const float *dz, const float *var,
                     const float *weight, const float *bias, float *edz, float *eydz, float *dx,
                     float *dweight, float *dbias, float eps, int act, float slope, int training,
                     float *ws, stream_t st) {
  if (training) {
    if (!skd_abn_backward_reduce(N, C, S, z, dz, weight, bias, edz, eydz, eps, act, slope, ws, st)) return 0;
  } else { /* functions.py:146-147 */
    memset(edz, 0, sizeof(float) * (size_t)C);
    memset(eydz, 0, sizeof(float) * (size_t)C);
  }
  return skd_abn_backward_dx(N, C, S, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, act, slope, st);
}

/* ---- training-time BN -> (+residual) -> ReLU, out of place (networks/pspnet_combine.py:36-43, 68-82) ---- */
int skd_abn_apply_to(int N, int C, int S, const float *x, const float *residual, float *out, const float *mean,
                     const float *var, const float *weight, const float *bias, float eps, int act, float slope,
                     stream_t st) {
  if (N <= 0 || C <= 0 || S <= 0 || !x || !out) return 0;
  memcpy(out, x, sizeof(float) * (size_t)N * C * S);
  if (residual) return skd_abn_apply_residual(N, C, S, out, residual, mean, var, weight, bias, eps, act, slope, st);
  return skd_abn_apply(N, C, S, out, mean, var, weight, bias, eps, act, slope, st);
}

int skd_abn_forward_train_to(int N, int C, int S, const float *x, const float *residual, float *out,
                             const float *weight, const float *bias, float *rm, float *rv, float *mean,
                             float *var, float momentum, float eps, int act, float slope, float *ws, stream_t st) {
  (void)ws;
  if (!skd_bn_mean_var(N, C, S, x, mean, var, st)) return 0;
  if (rm && rv) skd_abn_update_running(C, rm, rv, mean, var, momentum, (double)N * (double)S, st);
  return skd_abn_apply_to(N, C, S, x, residual, out, mean, var, weight, bias, eps, act, slope, st);
}

int skd_abn_relu_backward_reduce(int N, int C, int S, const float *x, const float *out, const float *dout,
                                 const float *mean, const float *var, float *edz, float *eydz, float eps,
                                 float *ws, stream_t st) {
  (void)ws; (void)st;
  if (N <= 0 || C <= 0 || S <= 0) return 0;
  const double norm = 1.0 / ((double)N * (double)S);
  for (int c = 0; c < C; ++c) {
    const float is = inv_std_of(var[c], eps);
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n) {
      const int64_t off = ((int64_t)n * C + c) * S;
      for (int i = 0; i < S; ++i) {
        const float dz = out[off + i] > 0.f ? dout[off + i] : 0.f;      /* ReLU backward */
        const float y = (x[off + i] - mean[c]) * is;                     /* bn.cu:158 */
        s1 += dz;
        s2 += (double)y * dz;
      }
    }
    edz[c] = (float)(s1 * norm);
    eydz[c] = (float)(s2 * norm);
  }
  return 1;
}

int skd_abn_relu_backward_dx(int N, int C, int S, const float *x, const float *out, const float *dout,
                             const float *mean, const float *var, const float *weight, const float *edz,
                             const float *eydz, float *dx, float *dres, float *dweight, float *dbias, float eps,
                             stream_t st) {
  (void)st;
  if (N <= 0 || C <= 0 || S <= 0 || !dx) return 0;
  for (int c = 0; c < C; ++c) {
    const float is = inv_std_of(var[c], eps);
    const float mul = gamma_of(weight, c, eps) * is;                     /* bn.cu:203 */
    for (int n = 0; n < N; ++n) {
      const int64_t off = ((int64_t)n * C + c) * S;
      for (int i = 0; i < S; ++i) {
        const float dz = out[off + i] > 0.f ? dout[off + i] : 0.f;
        const float y = (x[off + i] - mean[c]) * is;
        dx[off + i] = (dz - edz[c] - y * eydz[c]) * mul;                 /* bn.cu:209 */
        if (dres) dres[off + i] = dz;
      }
    }
    const float norm = (float)N * (float)S;
    if (dweight) {
      if (weight[c] > 0.f) dweight[c] += eydz[c] * norm;                 /* bn.cu:217-223 */
      else if (weight[c] < 0.f) dweight[c] -= eydz[c] * norm;
    }
    if (dbias) dbias[c] += edz[c] * norm;                                /* bn.cu:226-229 */
  }
  return 1;
}

/* ---- inference BN -> (+residual) -> activation on a channels-last (rows, C) tensor ---- */
int skd_abn_apply_nhwc(int64_t rows, int C, float *x, const float *residual, const float *mean, const float *var,
                       const float *weight, const float *bias, float eps, int act, float slope, stream_t st) {
  (void)st;
  if (rows <= 0 || C <= 0 || (C & 3) || !x) return 0;
  for (int64_t r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      const float y = (x[r * C + c] - mean[c]) * inv_std_of(var[c], eps);            /* bn.cu:158 */
      float z = y * gamma_of(weight, c, eps) + beta_of(bias, c);                     /* bn.cu:159 */
      if (residual) z = z + residual[r * C + c];
      x[r * C + c] = z;
    }
  act_forward(act, rows * C, x, slope);
  return 1;
}

/* ---- channels-last (NHWC) training forms: x is (rows, C) row-major.  Restated by transposing to the (1, C, rows)
 * NCHW problem the entries above solve, calling them, and transposing back -- the maths is layout independent. ---- */
static float *to_planes(int64_t rows, int C, const float *x) {
  float *p = (float *)malloc(sizeof(float) * (size_t)rows * C);
  if (p && x)
    for (int64_t r = 0; r < rows; ++r)
      for (int c = 0; c < C; ++c) p[(int64_t)c * rows + r] = x[r * C + c];
  return p;
}
static void from_planes(int64_t rows, int C, const float *p, float *x) {
  for (int64_t r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) x[r * C + c] = p[(int64_t)c * rows + r];
}
static int nhwc_ok(int64_t rows, int C) { return rows > 0 && rows < 2147483647 && C >= 4 && C <= 1024 && (C & (C - 1)) == 0; }

int64_t skd_abn_nhwc_workspace_floats(int64_t rows, int C) { return nhwc_ok(rows, C) ? 2 * (int64_t)C : 0; }

int skd_abn_stats_nhwc(int64_t rows, int C, const float *x, float *mean, float *var, float *ws, stream_t st) {
  (void)ws;
  if (!nhwc_ok(rows, C)) return 0;
  float *p = to_planes(rows, C, x);
  const int r = p ? skd_bn_mean_var(1, C, (int)rows, p, mean, var, st) : 0;
  free(p);
  return r;
}

int skd_abn_apply_nhwc_to(int64_t rows, int C, const float *x, const float *residual, float *out, const float *mean,
                          const float *var, const float *weight, const float *bias, float eps, int act, float slope,
                          stream_t st) {
  if (!nhwc_ok(rows, C) || !x || !out) return 0;
  if (out != x) memcpy(out, x, sizeof(float) * (size_t)rows * C);
  return skd_abn_apply_nhwc(rows, C, out, residual, mean, var, weight, bias, eps, act, slope, st);
}

int skd_abn_forward_train_nhwc(int64_t rows, int C, const float *x, const float *residual, float *out,
                               const float *weight, const float *bias, float *rm, float *rv, float *mean, float *var,
                               float momentum, float eps, int act, float slope, float *ws, stream_t st) {
  if (!skd_abn_stats_nhwc(rows, C, x, mean, var, ws, st)) return 0;
  if (rm && rv) skd_abn_update_running(C, rm, rv, mean, var, momentum, (double)rows, st);
  return skd_abn_apply_nhwc_to(rows, C, x, residual, out, mean, var, weight, bias, eps, act, slope, st);
}

int skd_abn_backward_reduce_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *weight,
                                 const float *bias, float *edz, float *eydz, float eps, int act, float slope,
                                 float *ws, stream_t st) {
  if (!nhwc_ok(rows, C)) return 0;
  float *pz = to_planes(rows, C, z), *pd = to_planes(rows, C, dz);
  const int r = (pz && pd) ? skd_abn_backward_reduce(1, C, (int)rows, pz, pd, weight, bias, edz, eydz, eps, act, slope, ws, st) : 0;
  free(pz); free(pd);
  return r;
}

int skd_abn_backward_dx_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *var,
                             const float *weight, const float *bias, const float *edz, const float *eydz, float *dx,
                             float *dweight, float *dbias, float eps, int act, float slope, int accumulate,
                             stream_t st) {
  if (!nhwc_ok(rows, C) || !dx) return 0;
  float *pz = to_planes(rows, C, z), *pd = to_planes(rows, C, dz), *px = to_planes(rows, C, NULL);
  int r = 0;
  if (!accumulate) {                       /* accumulate == 0: dweight / dbias are written, not added to */
    if (dweight) memset(dweight, 0, sizeof(float) * (size_t)C);
    if (dbias) memset(dbias, 0, sizeof(float) * (size_t)C);
  }
  if (pz && pd && px) {
    r = skd_abn_backward_dx(1, C, (int)rows, pz, pd, var, weight, bias, edz, eydz, px, dweight, dbias, eps, act, slope, st);
    if (r) from_planes(rows, C, px, dx);
  }
  free(pz); free(pd); free(px);
  return r;
}

int skd_abn_relu_backward_reduce_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout,
                                      const float *mean, const float *var, float *edz, float *eydz, float eps,
                                      float *ws, stream_t st) {
  if (!nhwc_ok(rows, C)) return 0;
  float *px = to_planes(rows, C, x), *po = to_planes(rows, C, out), *pd = to_planes(rows, C, dout);
  const int r = (px && po && pd) ? skd_abn_relu_backward_reduce(1, C, (int)rows, px, po, pd, mean, var, edz, eydz, eps, ws, st) : 0;
  free(px); free(po); free(pd);
  return r;
}

int skd_abn_relu_backward_dx_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout,
                                  const float *mean, const float *var, const float *weight, const float *edz,
                                  const float *eydz, float *dx, float *dres, float *dweight, float *dbias, float eps,
                                  int accumulate, stream_t st) {
  if (!nhwc_ok(rows, C) || !dx) return 0;
  if (!accumulate) {
    if (dweight) memset(dweight, 0, sizeof(float) * (size_t)C);
    if (dbias) memset(dbias, 0, sizeof(float) * (size_t)C);
  }
  float *px = to_planes(rows, C, x), *po = to_planes(rows, C, out), *pd = to_planes(rows, C, dout);
  float *pdx = to_planes(rows, C, NULL), *pdr = dres ? to_planes(rows, C, NULL) : NULL;
  int r = 0;
  if (px && po && pd && pdx && (!dres || pdr)) {
    r = skd_abn_relu_backward_dx(1, C, (int)rows, px, po, pd, mean, var, weight, edz, eydz, pdx, pdr, dweight, dbias, eps, st);
    if (r) {
      from_planes(rows, C, pdx, dx);
      if (dres) from_planes(rows, C, pdr, dres);
    }
  }
  free(px); free(po); free(pd); free(pdx); free(pdr);
  return r;
}

/* forward WITHOUT residual: the ReLU mask is a function of x alone, so `out` need not be read -- here it is simply
 * re-created with the oracle's own forward expression and handed to the entries above */
static float *relu_out_from_x(int64_t rows, int C, const float *x, const float *mean, const float *var, const float *weight,
                              const float *bias, float eps) {
  float *out = (float *)malloc(sizeof(float) * (size_t)rows * C);
  if (out && !skd_abn_apply_nhwc_to(rows, C, x, NULL, out, mean, var, weight, bias, eps, ACT_RELU, 0.f, NULL)) { free(out); out = NULL; }
  return out;
}

int skd_abn_relu_backward_reduce_nhwc_x(int64_t rows, int C, const float *x, const float *dout, const float *mean,
                                        const float *var, const float *weight, const float *bias, float *edz, float *eydz,
                                        float eps, float *ws, stream_t st) {
  if (!nhwc_ok(rows, C) || !x || !dout) return 0;
  float *out = relu_out_from_x(rows, C, x, mean, var, weight, bias, eps);
  const int r = out ? skd_abn_relu_backward_reduce_nhwc(rows, C, x, out, dout, mean, var, edz, eydz, eps, ws, st) : 0;
  free(out);
  return r;
}

int skd_abn_relu_backward_dx_nhwc_x(int64_t rows, int C, const float *x, const float *dout, const float *mean,
                                    const float *var, const float *weight, const float *bias, const float *edz,
                                    const float *eydz, float *dx, float *dweight, float *dbias, float eps, int accumulate,
                                    stream_t st) {
  if (!nhwc_ok(rows, C) || !x || !dout || !dx) return 0;
  float *out = relu_out_from_x(rows, C, x, mean, var, weight, bias, eps);
  const int r = out ? skd_abn_relu_backward_dx_nhwc(rows, C, x, out, dout, mean, var, weight, edz, eydz, dx, NULL, dweight,
                                                    dbias, eps, accumulate, st) : 0;
  free(out);
  return r;
}

/* reduce + dx in one call: the product fuses them into one launch, the maths is the two entries above */
int skd_abn_backward_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *var, const float *weight,
                          const float *bias, float *edz, float *eydz, float *dx, float *dweight, float *dbias, float eps,
                          int act, float slope, int accumulate, float *ws, stream_t st) {
  if (!skd_abn_backward_reduce_nhwc(rows, C, z, dz, weight, bias, edz, eydz, eps, act, slope, ws, st)) return 0;
  return skd_abn_backward_dx_nhwc(rows, C, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, act, slope, accumulate, st);
}

int skd_abn_relu_backward_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout, const float *mean,
                               const float *var, const float *weight, const float *bias, float *edz, float *eydz, float *dx,
                               float *dres, float *dweight, float *dbias, float eps, int accumulate, float *ws, stream_t st) {
  if (out == NULL) {
    if (dres != NULL) return 0;
    if (!skd_abn_relu_backward_reduce_nhwc_x(rows, C, x, dout, mean, var, weight, bias, edz, eydz, eps, ws, st)) return 0;
    return skd_abn_relu_backward_dx_nhwc_x(rows, C, x, dout, mean, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, accumulate, st);
  }
  if (!skd_abn_relu_backward_reduce_nhwc(rows, C, x, out, dout, mean, var, edz, eydz, eps, ws, st)) return 0;
  return skd_abn_relu_backward_dx_nhwc(rows, C, x, out, dout, mean, var, weight, edz, eydz, dx, dres, dweight, dbias, eps, accumulate, st);
}

/* ---- cross-replica combine, libs/functions.py:196-197 + 208-209 ----
 * weights == NULL is the reference rule; weights (w_g = n_g / sum n) the pooled statistics of unequal shards. */
int skd_abn_combine_stats(int G, int C, const float *gathered, const float *weights, int rank, float *mean, float *var,
                          float *rm, float *rv, float momentum, double n, stream_t st) {
  if (G <= 0 || C <= 0 || !gathered || !mean || !var || (weights && (rank < 0 || rank >= G))) return 0;
  for (int c = 0; c < C; ++c) {
    double m = 0.0, v = 0.0;
    for (int g = 0; g < G; ++g) m += (weights ? (double)weights[g] : 1.0 / G) * gathered[((int64_t)g * 2) * C + c];   /* means.mean(0) */
    for (int g = 0; g < G; ++g) {
      const double d = m - gathered[((int64_t)g * 2) * C + c];
      v += (weights ? (double)weights[g] : 1.0 / G) * (gathered[((int64_t)g * 2 + 1) * C + c] + d * d);   /* (vars + (mean - means)**2).mean(0) */
    }
    mean[c] = (float)m;
    var[c] = (float)v;
  }
  if (weights) n = n / (double)weights[rank];
  if (rm && rv) skd_abn_update_running(C, rm, rv, mean, var, momentum, n, st);
  return 1;
}

/* ---- 1x1 convolution + eval-mode ABN (+ residual) + activation, channels-last (include/skd.h section 11):
 *      the convolution as a plain dot product in double, then the forward formula of bn.cu:146-159 ---- */
int skd_conv1x1_abn_supported(int64_t M, int K, int N) { return M > 0 && K > 0 && N > 0 && K % 16 == 0 && N % 128 == 0; }

int skd_conv1x1_abn_nhwc(int64_t M, int K, int N, const float *x, const float *w, const float *residual, float *out,
                         const float *mean, const float *var, const float *weight, const float *bias, float eps, int act,
                         float slope, stream_t st) {
  (void)st;
  if (!skd_conv1x1_abn_supported(M, K, N) || !x || !w || !out || !mean || !var) return 0;
  if (act != ACT_NONE && act != ACT_RELU && act != ACT_LEAKY) return 0;
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double a = 0.0;
      for (int k = 0; k < K; ++k) a += (double)x[m * K + k] * (double)w[(int64_t)n * K + k];
      const float conv = (float)a;
      const float is = (var[n] != 0.f || eps != 0.f) ? 1.f / sqrtf(var[n] + eps) : 0.f;
      const float ga = weight ? fabsf(weight[n]) + eps : 1.f, be = bias ? bias[n] : 0.f;
      float z = ((conv - mean[n]) * is) * ga + be;
      if (residual) z += residual[m * N + n];
      if (act == ACT_RELU) z = z < 0.f ? 0.f : z;
      if (act == ACT_LEAKY) z = z < 0.f ? z * slope : z;
      out[m * N + n] = z;
    }
  return 1;
}

/* [mean | invstd | gamma | beta] of an eval-mode InPlace-ABN, bn.cu:146-159 */
int skd_abn_pack_eval_params(int K, const float *mean, const float *var, const float *weight, const float *bias, float eps,
                             float *pack, stream_t st) {
  (void)st;
  if (K <= 0 || !mean || !var || !pack) return 0;
  for (int k = 0; k < K; ++k) {
    pack[k] = mean[k];
    pack[K + k] = (var[k] != 0.f || eps != 0.f) ? 1.f / sqrtf(var[k] + eps) : 0.f;
    pack[2 * (int64_t)K + k] = weight ? fabsf(weight[k]) + eps : 1.f;
    pack[3 * (int64_t)K + k] = bias ? bias[k] : 0.f;
  }
  return 1;
}

/* the same GEMM with the preceding eval-mode BN + ReLU applied to x first (bn.cu:146-159 + ReLU), materialised here */
int skd_conv1x1_abn_pro_nhwc(int64_t M, int K, int N, const float *x, const float *w, const float *residual, float *out,
                             const float *mean, const float *var, const float *weight, const float *bias, float eps,
                             const float *ppack, int act, float slope, stream_t st) {
  if (!skd_conv1x1_abn_supported(M, K, N) || !x || !ppack) return 0;
  float *a = (float *)malloc(sizeof(float) * (size_t)M * K);
  if (!a) return 0;
  for (int64_t m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) {
      const float z = ((x[m * K + k] - ppack[k]) * ppack[K + k]) * ppack[2 * (int64_t)K + k] + ppack[3 * (int64_t)K + k];
      a[m * K + k] = z < 0.f ? 0.f : z;
    }
  const int r = skd_conv1x1_abn_nhwc(M, K, N, a, w, residual, out, mean, var, weight, bias, eps, act, slope, st);
  free(a);
  return r;
}
