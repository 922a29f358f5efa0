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
//   cells    EIGHT lanes per SOURCE CELL (image, j, i) and head (one per output row of the cell, folded with DPP): the output
//            pixels whose top-left tap is source pixel (j, i) -- ~8 x 8 of them at 65 -> 512 -- are all interpolated from the
//            cell's four corner logits, so the lanes keep those in registers, evaluate every pixel's softmax exactly ONCE, add
//            -log p[target] to the loss and pull (softmax - onehot) back onto the four corners with the bilinear weights (row
//            sums in registers, the four corner accumulators in LDS).  A workgroup (512 threads) owns a tile of 8 x 8 cells
//            (kCeTJ x kCeTI) of both heads; its target rectangle (~66 x 66 int64 at 65 -> 512, kCeTgtMax bytes) is read once,
//            coalesced, into LDS as bytes.  The tile's (8+1) x (8+1) NODES are then summed from the <= 4 cells around each
//            (fixed order) and written as per-tile partials.  Tiles are walked in an XCD-aware order (neighbouring tiles share
//            target and logit lines: one L2 fetches them, not eight).
//   nodes    one lane per source logit: the <= 4 tiles that share the node, summed in a fixed order and scaled by
//            head_weight / n_valid (the CE mean is only known when every workgroup has finished) -> dloss/dlogits.
// Round 1-4 used a separable formulation (rows kernel -> (B, heads, C, H, w) row gradients -> columns kernel): every pixel's
// softmax was evaluated twice and the 40 MB intermediate was written and read back: 147 MB of HBM traffic for 27 MB of
// algorithmic bytes (profiles/r04g_pmc.json), 340 us.  Now (profiles/r05e_pmc.json): target 16.8 MB + logits 5.3 MB read once,
// 11.8 MB of node partials written (row-contiguous per tile) and 7.9 MB of them read back + 5.3 MB of gradients: 42.7 MB, 131 us.
// Class counts: C <= 24 runs the CMAX = 24 instantiation (this network: 19).  24 < C <= 64 runs CMAX = 64: five 64-float
// per-lane arrays under __launch_bounds__(512, 2) SPILL to scratch and the corner sums need 64 KB of dynamic + 8 KB of static LDS
// (fits gfx950's 160 KB, nothing smaller) -- a correct but slow path that exists so that other label sets work at all.
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

// Sum over the 8 lanes 8g .. 8g+7 of a wave, result in all of them, as three DPP adds (no LDS round trip, unlike ds_bpermute):
// lane ^ 1 and lane ^ 2 inside each quad, then the mirrored lane of the other quad (every lane of a quad holds the quad's sum by
// then).  A fixed order of additions: bit-reproducible.
__device__ __forceinline__ float group8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}

constexpr int kCeTJ = 8, kCeTI = 8;   // source cells per tile
constexpr int kCeRows = 8;            // lanes per cell: lane r of a cell takes the cell's output rows Ylo + r, Ylo + r + 8, ...
constexpr int kCeCells = kCeTJ * kCeTI, kCeThreads = kCeCells * kCeRows;   // 512 threads = 8 waves
constexpr int kCeTgtMax = 8192;       // bytes of LDS for a tile's target rectangle (8 x 8 cells at 65 -> 512: ~66 x 66)
constexpr unsigned char kCeIgnore = 255, kCeBad = 254;

// part[(wg*3 + 0..2)] = sum of -log p (main), sum of -log p (dsn), number of valid pixels (NaN when a label is out of range)
// pnodes: (B, heads, C, NTy, TJ + 1, NTx, TI + 1) per-tile node sums of the UNSCALED gradient, or NULL (loss only).  (A tile-major
// layout was measured too -- 7.0 instead of 11.8 MB written, but 15.4 instead of 7.9 MB read back by ce_nodes: gpurun r05c / r05d.)
//
// Thread layout (round 5, second version).  The first cell formulation gave one lane a whole cell (64 pixels x ~420 instructions,
// 256 registers): 1056 waves for 1024 SIMDs -- one wave per SIMD, nothing to hide a dependent-issue stall behind: 254 us
// (gpurun r05b), still VALU-latency-bound.  Now EIGHT lanes share a cell, one output row each (8448 waves, 2-3 per SIMD): a lane
// interpolates its row's two column logits, walks the row's ~8 pixels (softmax once per pixel, row sums rL / rR of
// (softmax - onehot) x horizontal weight in registers), and the eight lanes' rows are folded onto the cell's four corners with a
// 3-step butterfly inside the lane group (fixed order: bit-reproducible); lane 0 of the group adds the result into the
// workgroup's corner table in LDS.  The two heads run one after the other over the same staged target rectangle.
#ifndef SKD_CE_WAVES_PER_SIMD
#define SKD_CE_WAVES_PER_SIMD 2       // tools/ce_lab.py measures 2 / 3 / 4 (register budget 256 / 168 / 128 per lane)
#endif
template <int CMAX, bool TWO>
__global__ __launch_bounds__(kCeThreads, SKD_CE_WAVES_PER_SIMD) void ce_cells_kernel(
    const float *__restrict__ lm, const float *__restrict__ ld, const int64_t *__restrict__ target,
    float *__restrict__ pnodes, float *__restrict__ part, int B, int C, int h, int w, int H, int W, int ignore_index,
    float sy, float sx, int NTy, int NTx) {
  constexpr int TJ = kCeTJ, TI = kCeTI, CELLS = kCeCells, NT = kCeThreads, NW = NT / kWave;
  extern __shared__ float csum[];                        // [4 corners][CMAX][CELLS]: the tile's corner sums of the current head
  __shared__ unsigned char tgt[kCeTgtMax];
  __shared__ int rect[4];
  __shared__ float red[4][NW];
  const int tid = threadIdx.x;
  const int rslot = tid & (kCeRows - 1), cell = tid >> 3, lj = cell / TI, li = cell % TI;
  // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so XCD k takes the k-th CONTIGUOUS
  // eighth of the tile list -- whole images at batch 8.  Neighbouring tiles share 128-byte lines of the 65-float logit rows and
  // of the target rows; spread over eight L2s every one of them fetched those lines again (43.8 MB read from the fabric for
  // 22 MB of data, profiles/r05d_pmc.json).
  const unsigned ntiles = (unsigned)B * NTy * NTx, chunk = (ntiles + 7u) / 8u;
  const unsigned tile = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int tx = (int)(tile % NTx), ty = (int)((tile / NTx) % NTy), b = (int)(tile / ((unsigned)NTx * NTy));
  const int j = ty * TJ + lj, i = tx * TI + li;
  const bool valid = j < h && i < w;
  const bool grad = pnodes != nullptr;
  if (tid == 0) {
    rect[0] = 0x7fffffff;
    rect[1] = -1;
    rect[2] = 0x7fffffff;
    rect[3] = -1;
  }
  int Ylo = 1, Yhi = 0, Xlo = 1, Xhi = 0;
  if (valid) {
    cell_range(j, sy, h, H, Ylo, Yhi);
    cell_range(i, sx, w, W, Xlo, Xhi);
  }
  const bool work = valid && Ylo <= Yhi && Xlo <= Xhi;     // the same for the 8 lanes of a cell
  __syncthreads();
  if (work && rslot == 0) {                              // integer min / max: order-independent
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
  const int hw = h * w;
  const int j1 = j + (j < h - 1 ? 1 : 0), i1 = i + (i < w - 1 ? 1 : 0);
  const unsigned o00 = j * w + i, o01 = j * w + i1, o10 = j1 * w + i, o11 = j1 * w + i1;     // the cell's four corner logits
  constexpr float kLog2e = 1.4426950408889634f;
  constexpr int heads = TWO ? 2 : 1;
  const int rowlen = NTx * (TI + 1);
  float loss_m = 0.f, loss_d = 0.f, cnt = 0.f, bad = 0.f;
  for (int head = 0; head < heads; ++head) {
    if (grad)
      for (int k = tid; k < 4 * CMAX * CELLS; k += NT) csum[k] = 0.f;
    __syncthreads();                                     // (also: the staged targets are visible)
    const float *p = (head == 0 ? lm : ld) + (int64_t)b * C * hw;
    float loss = 0.f;
    if (work) {
      for (int Y0 = Ylo; Y0 <= Yhi; Y0 += kCeRows) {    // one pass for cells of <= 8 rows (every up-sampling factor <= 8)
        const int Y = Y0 + rslot;
        const bool on = Y <= Yhi;
        const Tap tY = tap_of(on ? Y : Ylo, sy, h);
        float t0[CMAX], t1[CMAX], rL[CMAX], rR[CMAX];
        // vertical interpolation of the cell's two columns for THIS lane's row.  Channels C <= c < CMAX are padding: a large negative
        // logit whose softmax term is exactly 0 -- no `if (c < C)` in the pixel loop (with a run-time C the compiler turned each
        // of them into a branch: 158 branches and 250 registers for 12 channels)
        // (the offsets are laundered through an empty asm so that the 4 x C corner loads are NOT hoisted out of this loop: the loop
        // runs once for every up-sampling factor <= 8, but hoisted the corners would stay live across the pixel loop: +76 registers)
        unsigned p00 = o00, p01 = o01, p10 = o10, p11 = o11;
        asm volatile("" : "+v"(p00), "+v"(p01), "+v"(p10), "+v"(p11));
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
          const float *q = p + (int64_t)(c < C ? c : 0) * hw;
          const float q00 = q[p00], q01 = q[p01], q10 = q[p10], q11 = q[p11];
          t0[c] = c < C ? tY.l0 * q00 + tY.l1 * q10 : -1e30f;
          t1[c] = c < C ? tY.l0 * q01 + tY.l1 * q11 : -1e30f;
          rL[c] = 0.f;
          rR[c] = 0.f;
        }
        if (on) {
          const int64_t *trow = target + ((int64_t)b * H + Y) * W;
          const unsigned char *srow = tgt + (Y - RY0) * RW - RX0;
          auto label = [&](int X) -> int {
            if (staged) return srow[X];
            const int64_t tt = trow[X];
            return tt == (int64_t)ignore_index ? kCeIgnore : ((tt < 0 || tt >= C) ? kCeBad : (int)tt);
          };
          int t_next = label(Xlo);
          for (int X = Xlo; X <= Xhi; ++X) {
            const int t = t_next;
            t_next = label(X < Xhi ? X + 1 : X);       // the next pixel's label is fetched behind this pixel's arithmetic
            // branch-free: an ignored / out-of-range pixel runs the same arithmetic with zero weights (divergent `continue`s made
            // the compiler copy the rL / rR arrays around the branch: ~40 moves per pixel)
            const bool okp = t < kCeBad;
            const float first = head == 0 ? 1.f : 0.f;
            bad += t == kCeBad ? first : 0.f;   // F.cross_entropy asserts on such a label; here it poisons the loss (NaN)
            cnt += okp ? first : 0.f;
            const Tap tX = tap_of(X, sx, w);
            // (four interleaved max / sum chains instead of 19-long dependent ones.  Also measured and NOT adopted, tools/ce_lab.py: two
            // pixels per iteration as independent instruction streams -- 141 vs 136 us --, 3 / 4 waves per SIMD -- 169 / 201 vs 157 us)
            float v[CMAX];
            float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
              v[c] = tX.l0 * t0[c] + tX.l1 * t1[c];
              m4[c & 3] = fmaxf(m4[c & 3], v[c]);
            }
            const float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
            const float mxs = mx * kLog2e;
            float z4[4] = {0.f, 0.f, 0.f, 0.f}, vt = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
              vt = c == t ? v[c] : vt;
              v[c] = __builtin_amdgcn_exp2f(fmaf(v[c], kLog2e, -mxs));     // exp(v - max)
              z4[c & 3] += v[c];
            }
            const float z = (z4[0] + z4[1]) + (z4[2] + z4[3]);
            loss += okp ? logf(z) - (vt - mx) : 0.f;
            if (grad) {
              const float iz = 1.f / z;
              const float wl = okp ? tX.l0 : 0.f, wr = okp ? tX.l1 : 0.f;
#pragma unroll
              for (int c = 0; c < CMAX; ++c) {
                const float d = fmaf(v[c], iz, c == t ? -1.f : 0.f);       // softmax - onehot
                rL[c] = fmaf(d, wl, rL[c]);
                rR[c] = fmaf(d, wr, rR[c]);
              }
            }
          }
        }
        if (grad) {
          // fold the 8 rows of the cell onto its four corners: corner(k) += sum_rows wy(k) * r{L,R}; butterfly over the lane group
          // (lanes 8g .. 8g+7), the same order for every launch
          const float w0 = on ? tY.l0 : 0.f, w1 = on ? tY.l1 : 0.f;
#pragma unroll
          for (int c = 0; c < CMAX; ++c) {
            float a0 = w0 * rL[c], a1 = w0 * rR[c], a2 = w1 * rL[c], a3 = w1 * rR[c];
            a0 = group8_sum(a0);
            a1 = group8_sum(a1);
            a2 = group8_sum(a2);
            a3 = group8_sum(a3);
            if (rslot == 0) {
              csum[(0 * CMAX + c) * CELLS + cell] += a0;
              csum[(1 * CMAX + c) * CELLS + cell] += a1;
              csum[(2 * CMAX + c) * CELLS + cell] += a2;
              csum[(3 * CMAX + c) * CELLS + cell] += a3;
            }
          }
        }
      }
      if (grad && rslot == 0) {
        // border cells: both taps of an axis are the same source pixel (i1 == i0) -> that axis' second corner IS the first
        if (j1 == j) {
#pragma unroll
          for (int c = 0; c < CMAX; ++c) {
            csum[(0 * CMAX + c) * CELLS + cell] += csum[(2 * CMAX + c) * CELLS + cell];
            csum[(1 * CMAX + c) * CELLS + cell] += csum[(3 * CMAX + c) * CELLS + cell];
            csum[(2 * CMAX + c) * CELLS + cell] = 0.f;
            csum[(3 * CMAX + c) * CELLS + cell] = 0.f;
          }
        }
        if (i1 == i) {
#pragma unroll
          for (int c = 0; c < CMAX; ++c) {
            csum[(0 * CMAX + c) * CELLS + cell] += csum[(1 * CMAX + c) * CELLS + cell];
            csum[(2 * CMAX + c) * CELLS + cell] += csum[(3 * CMAX + c) * CELLS + cell];
            csum[(1 * CMAX + c) * CELLS + cell] = 0.f;
            csum[(3 * CMAX + c) * CELLS + cell] = 0.f;
          }
        }
      }
    }
    if (head == 0) loss_m = loss; else loss_d = loss;
    if (grad) {
      __syncthreads();
      // the tile's nodes: node (ly, lx) = cell (ly, lx) corner 00 + cell (ly, lx - 1) corner 01 + cell (ly - 1, lx) corner 10 +
      // cell (ly - 1, lx - 1) corner 11, in this order (cells outside the tile / the map hold zeros or are skipped)
      for (int k = tid; k < C * (TJ + 1) * (TI + 1); k += NT) {
        const int lx = k % (TI + 1), ly = (k / (TI + 1)) % (TJ + 1), c = k / ((TI + 1) * (TJ + 1));
        const int y = ty * TJ + ly, x = tx * TI + lx;
        if (y >= h || x >= w) continue;
        float s = 0.f;
        if (ly < TJ && lx < TI) s += csum[(0 * CMAX + c) * CELLS + ly * TI + lx];
        if (ly < TJ && lx > 0) s += csum[(1 * CMAX + c) * CELLS + ly * TI + lx - 1];
        if (ly > 0 && lx < TI) s += csum[(2 * CMAX + c) * CELLS + (ly - 1) * TI + lx];
        if (ly > 0 && lx > 0) s += csum[(3 * CMAX + c) * CELLS + (ly - 1) * TI + lx - 1];
        pnodes[(((((int64_t)b * heads + head) * C + c) * NTy + ty) * (TJ + 1) + ly) * rowlen + tx * (TI + 1) + lx] = s;
      }
      __syncthreads();                                   // before the next head clears the table
    }
  }
  // loss partials of the workgroup
  float lm_ = wave_sum(loss_m), ld_ = wave_sum(loss_d), cn_ = wave_sum(cnt), bd_ = wave_sum(bad);
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
    part[(int64_t)tile * 3 + 0] = s0;
    part[(int64_t)tile * 3 + 1] = s1;
    // a label outside [0, C) that is not ignore_index (raw Cityscapes ids, a mis-mapped label file) must not shrink the
    // valid set silently: the valid count becomes NaN, and with it the loss and every gradient of this call
    part[(int64_t)tile * 3 + 2] = s3 > 0.f ? __builtin_nanf("") : s2;
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
static void ce_tiles(int h, int w, int &NTy, int &NTx) {
  NTy = (int)cdiv(h, kCeTJ);
  NTx = (int)cdiv(w, kCeTI);
}

int64_t skd_ce_dsn_workspace_floats(int B, int C, int h, int w, int H, int W) {
  (void)H;
  (void)W;
  if (B <= 0 || C <= 0 || h <= 0 || w <= 0) return 8;
  int NTy, NTx;
  ce_tiles(h, w, NTy, NTx);
  const int64_t wgs = (int64_t)B * NTy * NTx;
  return 8 + wgs * 3 + (int64_t)B * 2 * C * NTy * (kCeTJ + 1) * NTx * (kCeTI + 1);
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
  int NTy, NTx;
  ce_tiles(h, w, NTy, NTx);
  const int64_t wgs = (int64_t)B * NTy * NTx;
  if (wgs > 2147483647) return 0;
  float *stat = workspace;
  float *part = workspace + 8;
  float *pnodes = (grad_main || grad_dsn) ? part + wgs * 3 : nullptr;
  const float sy = scale_of(h, H), sx = scale_of(w, W);
#define SKD_CE_LAUNCH(CM, TWO_)                                                                                              \
  do {                                                                                                                       \
    const size_t lds_ = sizeof(float) * 4 * CM * kCeCells;                                                                   \
    static PerDeviceFlag attr_;                                                                                              \
    bool *done_ = attr_.get();                                                                                               \
    if (done_ && !*done_) {                                                                                                  \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(ce_cells_kernel<CM, TWO_>),                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_) != hipSuccess)                          \
        return 0;                                                                                                            \
      *done_ = true;                                                                                                         \
    }                                                                                                                        \
    ce_cells_kernel<CM, TWO_><<<dim3((unsigned)(8 * cdiv(wgs, 8))), dim3(kCeThreads), lds_, st>>>(logits_main, logits_dsn, target, pnodes,  \
                                                                                 part, B, C, h, w, H, W, ignore_index, sy, \
                                                                                 sx, NTy, NTx);                              \
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
    ce_nodes_kernel<kCeTJ, kCeTI><<<grid, block, 0, st>>>(pnodes, stat, grad_main, grad_dsn, B, C, h, w, heads, aux_weight, NTy, NTx);
  }
  return ok();
}

}  // extern "C"
