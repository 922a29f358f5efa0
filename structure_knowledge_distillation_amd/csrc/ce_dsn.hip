// ce_dsn.hip -- CriterionDSN fused: bilinear upsample (align_corners) + cross-entropy(ignore_index)
// for the main and the deep-supervision logits, forward + gradient, for gfx950.
//
// Reference: CriterionDSN.forward, utils/criterion.py:179-188
//     up    = F.upsample(preds[k], size=(H, W), mode='bilinear', align_corners=True)   k = 0, 1
//     loss  = CE(up0, target, ignore_index=255) + 0.4 * CE(up1, target, ignore_index=255)
// The reference materialises both (B, C, H, W) upsampled tensors (159 MB each at B=8, 19 classes,
// 512x512), their log-softmax and, in backward, the same again -- roughly 2 GB of HBM traffic per step
// for 2 x 2.6 MB of logits.  Here nothing of size H x W x C ever exists in memory, and (round 5) nothing of size
// H x w x C either:
//   cells    one lane per SOURCE CELL (image, j, i) and head: the output pixels whose top-left tap is source pixel (j, i)
//            -- ~8 x 8 of them at 65 -> 512 -- are all interpolated from the cell's four corner logits, so the lane keeps
//            those in registers, evaluates every pixel's softmax exactly ONCE, adds -log p[target] to the loss and pulls
//            (softmax - onehot) back onto the four corners with the bilinear weights (row sums in registers, the four
//            corner accumulators in lane-private LDS).  A workgroup owns a tile of 8 x 16 cells of both heads; its target
//            rectangle (~66 x 130 int64) is read once, coalesced, into LDS as bytes.  The tile's (8+1) x (16+1) NODES are
//            then summed from the <= 4 cells around each (fixed order) and written as per-tile partials.
//   nodes    one lane per source logit: the <= 4 tiles that share the node, summed in a fixed order and scaled by
//            head_weight / n_valid (the CE mean is only known when every workgroup has finished) -> dloss/dlogits.
// Round 1-4 used a separable formulation (rows kernel -> (B, heads, C, H, w) row gradients -> columns kernel): every pixel's
// softmax was evaluated twice and the 40 MB intermediate was written and read back: 147 MB of HBM traffic for 27 MB of
// algorithmic bytes (profiles/r04g_pmc.json), 340 us.  Now: target 16.8 MB + logits + 2 x 6.1 MB of node partials.
// Gather formulation: no float atomics, fixed summation order, bit-reproducible.
// Index/weight arithmetic follows PyTorch's upsample_bilinear2d (align_corners=True):
//     scale = (in - 1) / (out - 1) (fp32); src = scale * dst; i0 = (int)src; i1 = i0 + (i0 < in - 1);
//     l1 = src - i0; l0 = 1 - l1.
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

// the destination indices d whose first tap is source index s (tap_of(d).i0 == s): contiguous, possibly empty (lo > hi)
__device__ __forceinline__ void cell_range(int s, float scale, int in, int out, int &lo, int &hi) {
  int a, b;
  dst_range(s, scale, out, a, b);
  lo = 1;
  hi = 0;
  for (int d = a; d <= b; ++d)
    if (tap_of(d, scale, in).i0 == s) {
      if (lo > hi) lo = d;
      hi = d;
    }
}

constexpr int kCeTgtMax = 16384;      // bytes of LDS for a tile's target rectangle (8 x 16 cells at 65 -> 512: ~66 x 130)
constexpr unsigned char kCeIgnore = 255, kCeBad = 254;

// Tile geometry per class-count bucket: cells per tile (TJ x TI) x heads = threads per workgroup.
//   C <= 24: 8 x 16 cells -> 256 threads with two heads; lane-private corner accumulators 4 x CMAX x 256 floats of LDS
//   C <= 64: 8 x  8 cells -> 128 threads
template <int CMAX>
struct CeTile {
  static constexpr int TJ = 8, TI = CMAX <= 24 ? 16 : 8;
};

// part[(wg*3 + 0..2)] = sum of -log p (main), sum of -log p (dsn), number of valid pixels (NaN when a label is out of range)
// pnodes: (B, heads, C, NTy, TJ + 1, NTx, TI + 1) per-tile node sums of the UNSCALED gradient, or NULL (loss only)
template <int CMAX, bool TWO>
__global__ __launch_bounds__(CeTile<CMAX>::TJ * CeTile<CMAX>::TI * (TWO ? 2 : 1)) void ce_cells_kernel(
    const float *__restrict__ lm, const float *__restrict__ ld, const int64_t *__restrict__ target,
    float *__restrict__ pnodes, float *__restrict__ part, int B, int C, int h, int w, int H, int W, int ignore_index,
    float sy, float sx, int NTy, int NTx) {
  constexpr int TJ = CeTile<CMAX>::TJ, TI = CeTile<CMAX>::TI, CELLS = TJ * TI, NT = CELLS * (TWO ? 2 : 1), NW = NT / kWave;
  extern __shared__ float acc[];                         // [4 corners][CMAX][NT]: column `tid` is private to the lane
  __shared__ unsigned char tgt[kCeTgtMax];
  __shared__ int rect[4];
  __shared__ float red[4][NW > 0 ? NW : 1];
  const int tid = threadIdx.x;
  const int head = tid / CELLS, cell = tid % CELLS, lj = cell / TI, li = cell % TI;
  const int tx = (int)(blockIdx.x % NTx), ty = (int)((blockIdx.x / NTx) % NTy), b = (int)(blockIdx.x / ((unsigned)NTx * NTy));
  const int j = ty * TJ + lj, i = tx * TI + li;
  const bool valid = j < h && i < w;
  const bool grad = pnodes != nullptr;
  if (tid == 0) {
    rect[0] = 0x7fffffff;
    rect[1] = -1;
    rect[2] = 0x7fffffff;
    rect[3] = -1;
  }
  if (grad)
    for (int k = tid; k < 4 * CMAX * NT; k += NT) acc[k] = 0.f;
  int Ylo = 1, Yhi = 0, Xlo = 1, Xhi = 0;
  if (valid) {
    cell_range(j, sy, h, H, Ylo, Yhi);
    cell_range(i, sx, w, W, Xlo, Xhi);
  }
  const bool work = valid && Ylo <= Yhi && Xlo <= Xhi;
  __syncthreads();
  if (work && head == 0) {                               // integer min / max: order-independent
    atomicMin(&rect[0], Ylo);
    atomicMax(&rect[1], Yhi);
    atomicMin(&rect[2], Xlo);
    atomicMax(&rect[3], Xhi);
  }
  __syncthreads();
  const int RY0 = rect[0], RX0 = rect[2], RH = rect[1] - rect[0] + 1, RW = rect[3] - rect[2] + 1;
  const bool staged = RH > 0 && RW > 0 && (int64_t)RH * RW <= kCeTgtMax;
  if (staged) {
    // the tile's target rectangle, once, coalesced along X: int64 -> one byte (class, 255 = ignored, 254 = out of range)
    for (int k = tid; k < RH * RW; k += NT) {
      const int ry = k / RW, rx = k - ry * RW;
      const int64_t t = target[((int64_t)b * H + RY0 + ry) * W + RX0 + rx];
      tgt[k] = t == (int64_t)ignore_index ? kCeIgnore : ((t < 0 || t >= C) ? kCeBad : (unsigned char)t);
    }
  }
  __syncthreads();
  float loss = 0.f, cnt = 0.f, bad = 0.f;
  if (work) {
    const int hw = h * w;
    // the head is uniform per wave (CELLS is a multiple of 64): say so, so that the channel planes are addressed as a SCALAR base
    // + a 32-bit lane offset (otherwise the compiler keeps 4 x CMAX loop-invariant 64-bit lane addresses alive: 150+ registers)
    const int head_u = __builtin_amdgcn_readfirstlane(head);
    const float *p = (head_u == 0 ? lm : ld) + (int64_t)b * C * hw;
    const int j1 = j + (j < h - 1 ? 1 : 0), i1 = i + (i < w - 1 ? 1 : 0);
    const unsigned o00 = j * w + i, o01 = j * w + i1, o10 = j1 * w + i, o11 = j1 * w + i1;     // the cell's four corner logits
    for (int Y = Ylo; Y <= Yhi; ++Y) {
      const Tap tY = tap_of(Y, sy, h);
      float t0[CMAX], t1[CMAX], rL[CMAX], rR[CMAX];
      // vertical interpolation of the cell's two columns, once per row (the 4 x C corner values come from L1 / L2 again for
      // every row: keeping them in 4 x CMAX more registers spills).  Channels C <= c < CMAX are padding: a large negative
      // logit whose softmax term is exactly 0 -- no `if (c < C)` in the loops below (with a run-time C the compiler turned each
      // of them into a branch: 158 branches and 250 registers for 12 channels)
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        const float *q = p + (int64_t)(c < C ? c : 0) * hw;
        const float q00 = q[o00], q01 = q[o01], q10 = q[o10], q11 = q[o11];
        t0[c] = c < C ? tY.l0 * q00 + tY.l1 * q10 : -1e30f;
        t1[c] = c < C ? tY.l0 * q01 + tY.l1 * q11 : -1e30f;
        rL[c] = 0.f;
        rR[c] = 0.f;
      }
      const int64_t *trow = target + ((int64_t)b * H + Y) * W;
      const unsigned char *srow = tgt + (Y - RY0) * RW - RX0;
      for (int X = Xlo; X <= Xhi; ++X) {
        int t;
        if (staged) {
          t = srow[X];
        } else {
          const int64_t tt = trow[X];
          t = tt == (int64_t)ignore_index ? kCeIgnore : ((tt < 0 || tt >= C) ? kCeBad : (int)tt);
        }
        if (t == kCeIgnore) continue;
        if (t == kCeBad) {                   // F.cross_entropy asserts on such a label; here it poisons the loss (NaN)
          bad += 1.f;
          continue;
        }
        cnt += 1.f;
        const Tap tX = tap_of(X, sx, w);
        float v[CMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
          v[c] = tX.l0 * t0[c] + tX.l1 * t1[c];
          mx = fmaxf(mx, v[c]);
        }
        float z = 0.f, vt = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
          vt = c == t ? v[c] - mx : vt;
          v[c] = __expf(v[c] - mx);
          z += v[c];
        }
        loss += logf(z) - vt;
        if (grad) {
          const float wl = tX.l0 / z, wr = tX.l1 / z;
#pragma unroll
          for (int c = 0; c < CMAX; ++c) {
            rL[c] += v[c] * wl - (c == t ? tX.l0 : 0.f);
            rR[c] += v[c] * wr - (c == t ? tX.l1 : 0.f);
          }
        }
      }
      if (grad) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
          acc[(0 * CMAX + c) * NT + tid] += tY.l0 * rL[c];
          acc[(1 * CMAX + c) * NT + tid] += tY.l0 * rR[c];
          acc[(2 * CMAX + c) * NT + tid] += tY.l1 * rL[c];
          acc[(3 * CMAX + c) * NT + tid] += tY.l1 * rR[c];
        }
      }
    }
    if (grad) {
      // border cells: both taps of an axis are the same source pixel (i1 == i0) -> that axis' second corner IS the first
      if (j1 == j) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
          acc[(0 * CMAX + c) * NT + tid] += acc[(2 * CMAX + c) * NT + tid];
          acc[(1 * CMAX + c) * NT + tid] += acc[(3 * CMAX + c) * NT + tid];
          acc[(2 * CMAX + c) * NT + tid] = 0.f;
          acc[(3 * CMAX + c) * NT + tid] = 0.f;
        }
      }
      if (i1 == i) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
          acc[(0 * CMAX + c) * NT + tid] += acc[(1 * CMAX + c) * NT + tid];
          acc[(2 * CMAX + c) * NT + tid] += acc[(3 * CMAX + c) * NT + tid];
          acc[(1 * CMAX + c) * NT + tid] = 0.f;
          acc[(3 * CMAX + c) * NT + tid] = 0.f;
        }
      }
    }
  }
  // loss partials of the workgroup: head 0 lanes carry the main loss and the counts, head 1 lanes the deep-supervision loss
  float lm_ = head == 0 ? loss : 0.f, ld_ = head == 0 ? 0.f : loss;
  float cn_ = head == 0 ? cnt : 0.f, bd_ = head == 0 ? bad : 0.f;
  lm_ = wave_sum(lm_);
  ld_ = wave_sum(ld_);
  cn_ = wave_sum(cn_);
  bd_ = wave_sum(bd_);
  if ((tid & (kWave - 1)) == 0) {
    red[0][tid / kWave] = lm_;
    red[1][tid / kWave] = ld_;
    red[2][tid / kWave] = cn_;
    red[3][tid / kWave] = bd_;
  }
  __syncthreads();
  if (tid == 0) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int k = 0; k < NW; ++k) {
      s0 += red[0][k];
      s1 += red[1][k];
      s2 += red[2][k];
      s3 += red[3][k];
    }
    part[(int64_t)blockIdx.x * 3 + 0] = s0;
    part[(int64_t)blockIdx.x * 3 + 1] = s1;
    // a label outside [0, C) that is not ignore_index (raw Cityscapes ids, a mis-mapped label file) must not shrink the
    // valid set silently: the valid count becomes NaN, and with it the loss and every gradient of this call
    part[(int64_t)blockIdx.x * 3 + 2] = s3 > 0.f ? __builtin_nanf("") : s2;
  }
  if (!grad) return;
  // the tile's nodes: node (ly, lx) = cell (ly, lx) corner 00 + cell (ly, lx - 1) corner 01 + cell (ly - 1, lx) corner 10 +
  // cell (ly - 1, lx - 1) corner 11, in this order (cells outside the tile / the map hold zeros or are skipped)
  constexpr int heads = TWO ? 2 : 1;
  const int rowlen = NTx * (TI + 1);
  for (int k = tid; k < heads * C * (TJ + 1) * (TI + 1); k += NT) {
    const int lx = k % (TI + 1), ly = (k / (TI + 1)) % (TJ + 1), c = (k / ((TI + 1) * (TJ + 1))) % C, hd = k / ((TI + 1) * (TJ + 1) * C);
    const int y = ty * TJ + ly, x = tx * TI + lx;
    if (y >= h || x >= w) continue;
    float s = 0.f;
    const int base = hd * CELLS;
    if (ly < TJ && lx < TI) s += acc[(0 * CMAX + c) * NT + base + ly * TI + lx];
    if (ly < TJ && lx > 0) s += acc[(1 * CMAX + c) * NT + base + ly * TI + lx - 1];
    if (ly > 0 && lx < TI) s += acc[(2 * CMAX + c) * NT + base + (ly - 1) * TI + lx];
    if (ly > 0 && lx > 0) s += acc[(3 * CMAX + c) * NT + base + (ly - 1) * TI + lx - 1];
    pnodes[(((((int64_t)b * heads + hd) * C + c) * NTy + ty) * (TJ + 1) + ly) * rowlen + tx * (TI + 1) + lx] = s;
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

// grad[b, c, y, x] = head_weight / n_valid * (the node's partial sums of the <= 4 tiles that share it, fixed order)
template <int TJ, int TI>
__global__ __launch_bounds__(kThreads) void ce_nodes_kernel(const float *__restrict__ pnodes, const float *__restrict__ stat,
                                                           float *__restrict__ gm, float *__restrict__ gd, int B, int C,
                                                           int h, int w, int heads, float aux_weight, int NTy, int NTx) {
  const int64_t total = (int64_t)B * heads * C * h * w;
  const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (tid >= total) return;
  const int x = (int)(tid % w);
  const int y = (int)((tid / w) % h);
  const int c = (int)((tid / ((int64_t)w * h)) % C);
  const int head = (int)((tid / ((int64_t)w * h * C)) % heads);
  const int b = (int)(tid / ((int64_t)w * h * C * heads));
  const int ty = y / TJ, ly = y - ty * TJ, tx = x / TI, lx = x - tx * TI;
  const int rowlen = NTx * (TI + 1);
  const float *base = pnodes + (((int64_t)b * heads + head) * C + c) * NTy * (TJ + 1) * rowlen;
  auto at = [&](int ty_, int ly_, int tx_, int lx_) { return base[((int64_t)ty_ * (TJ + 1) + ly_) * rowlen + tx_ * (TI + 1) + lx_]; };
  const bool up = ly == 0 && ty > 0, left = lx == 0 && tx > 0;
  float s = at(ty, ly, tx, lx);
  if (left) s += at(ty, ly, tx - 1, TI);
  if (up) s += at(ty - 1, TJ, tx, lx);
  if (up && left) s += at(ty - 1, TJ, tx - 1, TI);
  const float scale = (head == 0 ? 1.f : aux_weight) / stat[1];
  float *dst = head == 0 ? gm : gd;
  if (dst != nullptr) dst[(((int64_t)b * C + c) * h + y) * w + x] = s * scale;
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

static int ce_cmax(int C) { return C <= 12 ? 12 : (C <= 19 ? 19 : (C <= 24 ? 24 : 64)); }
static void ce_tiles(int C, int h, int w, int &TJ, int &TI, int &NTy, int &NTx) {
  TJ = 8;
  TI = ce_cmax(C) <= 24 ? 16 : 8;
  NTy = (int)cdiv(h, TJ);
  NTx = (int)cdiv(w, TI);
}

int64_t skd_ce_dsn_workspace_floats(int B, int C, int h, int w, int H, int W) {
  (void)H;
  (void)W;
  if (B <= 0 || C <= 0 || h <= 0 || w <= 0) return 8;
  int TJ, TI, NTy, NTx;
  ce_tiles(C, h, w, TJ, TI, NTy, NTx);
  const int64_t wgs = (int64_t)B * NTy * NTx;
  return 8 + wgs * 3 + (int64_t)B * 2 * C * NTy * (TJ + 1) * NTx * (TI + 1);
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
  int TJ, TI, NTy, NTx;
  ce_tiles(C, h, w, TJ, TI, NTy, NTx);
  const int64_t wgs = (int64_t)B * NTy * NTx;
  if (wgs > 2147483647) return 0;
  float *stat = workspace;
  float *part = workspace + 8;
  float *pnodes = (grad_main || grad_dsn) ? part + wgs * 3 : nullptr;
  const float sy = scale_of(h, H), sx = scale_of(w, W);
#define SKD_CE_LAUNCH(CM, TWO_)                                                                                              \
  do {                                                                                                                       \
    constexpr int NT_ = CeTile<CM>::TJ * CeTile<CM>::TI * (TWO_ ? 2 : 1);                                                    \
    const size_t lds_ = pnodes ? sizeof(float) * 4 * CM * NT_ : 0;                                                           \
    static PerDeviceFlag attr_;                                                                                              \
    bool *done_ = attr_.get();                                                                                               \
    if (done_ && !*done_) {                                                                                                  \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(ce_cells_kernel<CM, TWO_>),                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * 4 * CM * NT_)) != hipSuccess) \
        return 0;                                                                                                            \
      *done_ = true;                                                                                                         \
    }                                                                                                                        \
    ce_cells_kernel<CM, TWO_><<<dim3((unsigned)wgs), dim3(NT_), lds_, st>>>(logits_main, logits_dsn, target, pnodes, part, B, \
                                                                           C, h, w, H, W, ignore_index, sy, sx, NTy, NTx);   \
  } while (0)
#define SKD_CE(CM)              \
  do {                          \
    if (two) SKD_CE_LAUNCH(CM, true); \
    else SKD_CE_LAUNCH(CM, false);    \
  } while (0)
  switch (ce_cmax(C)) {
    case 12: SKD_CE(12); break;
    case 19: SKD_CE(19); break;
    case 24: SKD_CE(24); break;
    default: SKD_CE(64); break;
  }
#undef SKD_CE
#undef SKD_CE_LAUNCH
  ce_finalize_kernel<<<dim3(1), dim3(kThreads), 0, st>>>(part, wgs, two ? aux_weight : 0.f, loss, stat);
  if (pnodes != nullptr) {
    const int64_t n = (int64_t)B * heads * C * h * w;
    const dim3 grid((unsigned)cdiv(n, kThreads)), block(kThreads);
    if (TI == 16)
      ce_nodes_kernel<8, 16><<<grid, block, 0, st>>>(pnodes, stat, grad_main, grad_dsn, B, C, h, w, heads, aux_weight, NTy, NTx);
    else
      ce_nodes_kernel<8, 8><<<grid, block, 0, st>>>(pnodes, stat, grad_main, grad_dsn, B, C, h, w, heads, aux_weight, NTy, NTx);
  }
  return ok();
}

}  // extern "C"
