// ce_dsn.hip -- CriterionDSN fused: bilinear upsample (align_corners) + cross-entropy(ignore_index)
// for the main and the deep-supervision logits, forward + gradient, for gfx950.
//
// Reference: CriterionDSN.forward, utils/criterion.py:179-188
//     up    = F.upsample(preds[k], size=(H, W), mode='bilinear', align_corners=True)   k = 0, 1
//     loss  = CE(up0, target, ignore_index=255) + 0.4 * CE(up1, target, ignore_index=255)
// The reference materialises both (B, C, H, W) upsampled tensors (159 MB each at B=8, 19 classes,
// 512x512), their log-softmax and, in backward, the same again -- roughly 2 GB of HBM traffic per step
// for 2 x 2.6 MB of logits.  Here nothing of size H x W x C ever exists in memory:
//   stage A  one lane per (image, output row Y, source column x).  It walks the <= ~18 output pixels X
//            of row Y whose horizontal footprint touches source column x, rebuilds their C upsampled
//            logits from the 4 neighbouring source pixels (the 65x65 logit maps are L2 resident),
//            evaluates log-softmax, adds -log p[target] to the loss for the pixels it "owns"
//            (x0(X) == x, each pixel counted once) and accumulates the horizontally pulled-back gradient
//                 rowgrad[b, head, c, Y, x] = sum_X wx(X, x) * (softmax_c - [c == target])
//            (40 MB at B=8; written and read once, coalesced along x).
//   stage B  one lane per source logit: the vertical pull-back  sum_Y wy(Y, y) * rowgrad[.., Y, x],
//            scaled by head_weight / n_valid (the CE mean) -> dloss/dlogits, (B, C, h, w).
// Separable gather formulation: no atomics, fixed summation order, bit-reproducible.
// Index/weight arithmetic follows PyTorch's upsample_bilinear2d (align_corners=True):
//     scale = (in - 1) / (out - 1) (fp32); src = scale * dst; i0 = (int)src; i1 = i0 + (i0 < in - 1);
//     l1 = src - i0; l0 = 1 - l1.
// HBM-bound on the target read (8 B / pixel, int64) + rowgrad; the exp/log work is ~3 GFLOP.
#include "skd_common.hpp"

namespace skd {
namespace {

struct Tap {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ Tap tap_of(int dst, float scale, int in) {
  Tap t;
  const float src = scale * (float)dst;
  t.i0 = (int)src;
  if (t.i0 > in - 1) t.i0 = in - 1;
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

static inline float scale_of(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// conservative range of destination indices whose taps can touch source index s
__device__ __forceinline__ void dst_range(int s, float scale, int out, int &lo, int &hi) {
  if (scale <= 0.f) {
    lo = 0;
    hi = out - 1;
    return;
  }
  lo = (int)floorf((float)(s - 1) / scale) - 1;
  hi = (int)ceilf((float)(s + 1) / scale) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

// part[(wg*3 + 0..2)] = sum of -log p (main), sum of -log p (dsn), number of valid pixels
template <int CMAX, bool TWO>
__global__ __launch_bounds__(kThreads) void ce_rows_kernel(const float *__restrict__ lm,
                                                          const float *__restrict__ ld,
                                                          const int64_t *__restrict__ target,
                                                          float *__restrict__ rowgrad,  // (B, heads, C, H, w) or NULL
                                                          float *__restrict__ part, int B, int C, int h,
                                                          int w, int H, int W, int ignore_index,
                                                          float sy, float sx) {
  __shared__ float red[2 * kWavesPerWG];
  __shared__ float red2[2 * kWavesPerWG];
  const int64_t total = (int64_t)B * H * w;
  const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  float loss_m = 0.f, loss_d = 0.f, cnt = 0.f, bad = 0.f;
  if (tid < total) {
    const int x = (int)(tid % w);
    const int Y = (int)((tid / w) % H);
    const int b = (int)(tid / ((int64_t)w * H));
    const Tap ty = tap_of(Y, sy, h);
    const int hw = h * w;
    const float *pm = lm + (int64_t)b * C * hw;
    const float *pd = TWO ? ld + (int64_t)b * C * hw : nullptr;
    const int64_t *trow = target + ((int64_t)b * H + Y) * W;
    int Xlo, Xhi;
    dst_range(x, sx, W, Xlo, Xhi);
    const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
#pragma unroll
    for (int head = 0; head < (TWO ? 2 : 1); ++head) {
      const float *p = head == 0 ? pm : pd;
      // vertical interpolation once per thread: the three source columns x-1, x, x+1 of this row pair
      float a0[CMAX], a1[CMAX], a2[CMAX], g[CMAX];
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) {
          const float *q = p + (int64_t)c * hw;
          a0[c] = ty.l0 * q[ty.i0 * w + xm] + ty.l1 * q[ty.i1 * w + xm];
          a1[c] = ty.l0 * q[ty.i0 * w + x] + ty.l1 * q[ty.i1 * w + x];
          a2[c] = ty.l0 * q[ty.i0 * w + xp] + ty.l1 * q[ty.i1 * w + xp];
          g[c] = 0.f;
        }
      float loss = 0.f;
      for (int X = Xlo; X <= Xhi; ++X) {
        const Tap tx = tap_of(X, sx, w);
        float wt = 0.f;
        if (tx.i0 == x) wt += tx.l0;
        if (tx.i1 == x) wt += tx.l1;
        const bool own = tx.i0 == x;
        if (wt == 0.f && !own) continue;
        if (tx.i0 != x && tx.i0 != x - 1) continue;      // only taps {x-1, x} -> x or {x, x+1} reach column x
        const int64_t t = trow[X];
        if (t == (int64_t)ignore_index) continue;
        if (t < 0 || t >= C) {               // F.cross_entropy asserts on such a label; here it poisons the loss (NaN)
          if (own && head == 0) bad += 1.f;
          continue;
        }
        if (own && head == 0) cnt += 1.f;
        const bool left = tx.i0 == x - 1 && x > 0;       // taps (x-1, x); otherwise (x, x+1) [or (x, x) at the border]
        const bool same = tx.i1 == tx.i0;
        float v[CMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) {
            const float lo = left ? a0[c] : a1[c];
            const float hi = same ? lo : (left ? a1[c] : a2[c]);
            v[c] = tx.l0 * lo + tx.l1 * hi;
            mx = fmaxf(mx, v[c]);
          }
        float z = 0.f, vt = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) {
            if (c == (int)t) vt = v[c] - mx;
            v[c] = expf(v[c] - mx);
            z += v[c];
          }
        if (own) loss += logf(z) - vt;
        if (rowgrad != nullptr && wt != 0.f) {
          const float sc = wt / z;
#pragma unroll
          for (int c = 0; c < CMAX; ++c)
            if (c < C) g[c] += v[c] * sc - (c == (int)t ? wt : 0.f);
        }
      }
      if (head == 0) loss_m = loss; else loss_d = loss;
      if (rowgrad != nullptr) {
        const int heads = TWO ? 2 : 1;
        float *o = rowgrad + ((((int64_t)b * heads + head) * C) * H + Y) * w + x;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) o[(int64_t)c * H * w] = g[c];
      }
    }
  }
  block_sum2(loss_m, loss_d, red);
  block_sum2(cnt, bad, red2);
  if (threadIdx.x == 0) {
    part[(int64_t)blockIdx.x * 3 + 0] = loss_m;
    part[(int64_t)blockIdx.x * 3 + 1] = loss_d;
    // a label outside [0, C) that is not ignore_index (raw Cityscapes ids, a mis-mapped label file) must not shrink the
    // valid set silently: the valid count becomes NaN, and with it the loss and every gradient of this call
    part[(int64_t)blockIdx.x * 3 + 2] = bad > 0.f ? __builtin_nanf("") : cnt;
  }
}

// stat[0] = loss, stat[1] = n_valid, stat[2] = mean CE main, stat[3] = mean CE dsn
__global__ __launch_bounds__(kThreads) void ce_finalize_kernel(const float *__restrict__ part, int64_t nwg,
                                                              float aux_weight, float *__restrict__ loss,
                                                              float *__restrict__ stat) {
  __shared__ double red[3][kWavesPerWG];
  double a = 0.0, b = 0.0, c = 0.0;
  for (int64_t i = threadIdx.x; i < nwg; i += kThreads) {
    a += (double)part[i * 3];
    b += (double)part[i * 3 + 1];
    c += (double)part[i * 3 + 2];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) {
    red[0][wid] = a;
    red[1][wid] = b;
    red[2][wid] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0, sc = 0.0;
    for (int k = 0; k < kWavesPerWG; ++k) {
      sa += red[0][k];
      sb += red[1][k];
      sc += red[2][k];
    }
    // CrossEntropyLoss(reduction='mean', ignore_index): sum over valid / number of valid (NaN when none)
    const double lmain = sa / sc, ldsn = sb / sc;
    loss[0] = (float)(lmain + (double)aux_weight * ldsn);   // criterion.py:188
    stat[0] = loss[0];
    stat[1] = (float)sc;
    stat[2] = (float)lmain;
    stat[3] = (float)ldsn;
  }
}

// grad[b, c, y, x] = head_weight / n_valid * sum_Y wy(Y, y) * rowgrad[b, head, c, Y, x]
__global__ __launch_bounds__(kThreads) void ce_cols_kernel(const float *__restrict__ rowgrad,
                                                          const float *__restrict__ stat,
                                                          float *__restrict__ gm, float *__restrict__ gd,
                                                          int B, int C, int h, int w, int H, int heads,
                                                          float aux_weight, float sy) {
  const int64_t total = (int64_t)B * heads * C * h * w;
  const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (tid >= total) return;
  const int x = (int)(tid % w);
  const int y = (int)((tid / w) % h);
  const int c = (int)((tid / ((int64_t)w * h)) % C);
  const int head = (int)((tid / ((int64_t)w * h * C)) % heads);
  const int b = (int)(tid / ((int64_t)w * h * C * heads));
  const float *src = rowgrad + (((int64_t)b * heads + head) * C + c) * (int64_t)H * w + x;
  int Ylo, Yhi;
  dst_range(y, sy, H, Ylo, Yhi);
  float acc = 0.f;
  for (int Y = Ylo; Y <= Yhi; ++Y) {
    const Tap t = tap_of(Y, sy, h);
    float wt = 0.f;
    if (t.i0 == y) wt += t.l0;
    if (t.i1 == y) wt += t.l1;
    if (wt != 0.f) acc += wt * src[(int64_t)Y * w];
  }
  const float scale = (head == 0 ? 1.f : aux_weight) / stat[1];
  float *dst = head == 0 ? gm : gd;
  if (dst != nullptr) dst[(((int64_t)b * C + c) * h + y) * w + x] = acc * scale;
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

int64_t skd_ce_dsn_workspace_floats(int B, int C, int h, int w, int H, int W) {
  (void)W;
  if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0) return 8;
  const int64_t wgs = cdiv((int64_t)B * H * w, kThreads);
  return 8 + wgs * 3 + (int64_t)B * 2 * C * H * w;
}

int skd_ce_dsn_forward(int B, int C, int h, int w, int H, int W, const float *logits_main,
                       const float *logits_dsn, const int64_t *target, int ignore_index, float aux_weight,
                       float *loss, float *grad_main, float *grad_dsn, float *workspace, skd_stream_t stream) {
  if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return 0;
  if (!logits_main || !target || !loss || !workspace) return 0;
  if (grad_dsn && !logits_dsn) return 0;
  if (C > 64) return 0;  // class count of this path: 19 (Cityscapes), 11 (CamVid), 21 (VOC)
  hipStream_t st = as_stream(stream);
  const bool two = logits_dsn != nullptr;
  const int heads = two ? 2 : 1;
  const int64_t wgs = cdiv((int64_t)B * H * w, kThreads);
  float *stat = workspace;
  float *part = workspace + 8;
  float *rowgrad = (grad_main || grad_dsn) ? part + wgs * 3 : nullptr;
  const float sy = scale_of(h, H), sx = scale_of(w, W);
  const dim3 grid((unsigned)wgs), block(kThreads);
#define SKD_CE(CM)                                                                                         \
  do {                                                                                                     \
    if (two)                                                                                               \
      ce_rows_kernel<CM, true><<<grid, block, 0, st>>>(logits_main, logits_dsn, target, rowgrad, part, B, \
                                                       C, h, w, H, W, ignore_index, sy, sx);               \
    else                                                                                                   \
      ce_rows_kernel<CM, false><<<grid, block, 0, st>>>(logits_main, logits_dsn, target, rowgrad, part, B, \
                                                        C, h, w, H, W, ignore_index, sy, sx);              \
  } while (0)
  if (C <= 24) SKD_CE(24);
  else SKD_CE(64);
#undef SKD_CE
  ce_finalize_kernel<<<dim3(1), block, 0, st>>>(part, wgs, two ? aux_weight : 0.f, loss, stat);
  if (rowgrad != nullptr) {
    const int64_t n = (int64_t)B * heads * C * h * w;
    ce_cols_kernel<<<dim3((unsigned)cdiv(n, kThreads)), block, 0, st>>>(rowgrad, stat, grad_main, grad_dsn, B, C,
                                                                       h, w, H, heads, aux_weight, sy);
  }
  return ok();
}

}  // extern "C"
