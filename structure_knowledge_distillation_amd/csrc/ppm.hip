// ppm.hip -- pyramid pooling module data movement for gfx950.
//
// Reference: PSPModule, networks/pspnet_combine.py:86-112
//     priors = [upsample(stage(feats), (h, w), bilinear, align_corners=True) for stage in stages] + [feats]
//     stage  = AdaptiveAvgPool2d(size) -> conv1x1 -> InPlaceABNSync            size in (1, 2, 3, 6)
//     bottle = bottleneck(torch.cat(priors, 1))
// In the reference this is 4 adaptive-average-pool launches (each re-reading the whole feature map),
// 4 bilinear up-samplings to full resolution, a torch.cat that copies everything again, and in backward
// atomically-scattered adaptive-pool / up-sampling gradients onto 1x1 ... 6x6 targets.  On MI355X those
// stock kernels cost ~18 ms per step (rocprofv3, profiles/): tiny outputs + atomics = serialised.
// Here (all gather formulations, no atomics, deterministic):
//   ppm_pool      all pyramid levels from ONE read of the feature map; per plane, column sums per row-bin
//                 in registers / LDS, then one lane per output bin.
//   ppm_pool_bwd  dfeat[h, w] = sum over levels and bins containing (h, w) of g[bin] / area(bin).
//   ppm_concat    writes the up-sampled priors and the feature map straight into the concatenated
//                 (B, L*Cout + Cfeat, H, W) tensor -- the up-sampled priors never exist on their own.
//   ppm_concat_bwd  gradient of the priors = separable pull-back of the matching channel slice, one
//                 workgroup per plane staged in LDS (the feature-map slice's gradient is a view).
// Bin edges follow adaptive_avg_pool2d: [floor(i*H/s), ceil((i+1)*H/s)); bilinear taps follow
// upsample_bilinear2d with align_corners=True (see ce_dsn.hip).  HBM-bound: one read of the features,
// one write of the concatenated tensor.
#include "skd_common.hpp"

namespace skd {
namespace {

constexpr int kMaxLevels = 4;

struct Levels {
  int n;
  int size[kMaxLevels];
  int bin_off[kMaxLevels];  // offset of the level's first bin inside one plane's bin list (sum s^2)
  int row_off[kMaxLevels];  // offset of the level's first row-bin (sum s)
  int bins, rows;
};

static bool make_levels(int nsizes, const int *sizes, Levels &lv) {
  if (nsizes <= 0 || nsizes > kMaxLevels || !sizes) return false;
  lv.n = nsizes;
  lv.bins = lv.rows = 0;
  for (int k = 0; k < nsizes; ++k) {
    if (sizes[k] <= 0 || sizes[k] > 64) return false;
    lv.size[k] = sizes[k];
    lv.bin_off[k] = lv.bins;
    lv.row_off[k] = lv.rows;
    lv.bins += sizes[k] * sizes[k];
    lv.rows += sizes[k];
  }
  for (int k = nsizes; k < kMaxLevels; ++k) lv.size[k] = lv.bin_off[k] = lv.row_off[k] = 0;
  return true;
}

__device__ __forceinline__ int bin_start(int i, int n, int s) { return (i * n) / s; }
__device__ __forceinline__ int bin_end(int i, int n, int s) { return ((i + 1) * n + s - 1) / s; }

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

// pooled layout: level k block at planes * bin_off[k]; inside it [plane][s*s]
__global__ __launch_bounds__(kThreads) void ppm_pool_kernel(const float *__restrict__ x,
                                                           float *__restrict__ pooled, int64_t planes,
                                                           int H, int W, int ppw, Levels lv) {
  extern __shared__ __attribute__((aligned(16))) float rowacc[];  // [ppw][lv.rows][W]
  const int t = threadIdx.x;
  const int pl = t / W, w = t - pl * W;
  const int64_t plane = (int64_t)blockIdx.x * ppw + pl;
  const bool live = pl < ppw && plane < planes;
  if (live) {
    const float *px = x + plane * (int64_t)H * W + w;
    float *acc = rowacc + (int64_t)pl * lv.rows * W + w;
    for (int k = 0; k < lv.n; ++k) {
      const int s = lv.size[k];
      for (int i = 0; i < s; ++i) {
        const int h0 = bin_start(i, H, s), h1 = bin_end(i, H, s);
        float a = 0.f;
        for (int h = h0; h < h1; ++h) a += px[(int64_t)h * W];
        acc[(int64_t)(lv.row_off[k] + i) * W] = a;
      }
    }
  }
  __syncthreads();
  // one lane per (plane, bin)
  for (int o = t; o < ppw * lv.bins; o += kThreads) {
    const int p2 = o / lv.bins, bin = o - p2 * lv.bins;
    const int64_t plane2 = (int64_t)blockIdx.x * ppw + p2;
    if (plane2 >= planes) continue;
    int k = 0;
    while (k + 1 < lv.n && bin >= lv.bin_off[k + 1]) ++k;
    const int s = lv.size[k];
    const int local = bin - lv.bin_off[k];
    const int i = local / s, j = local - i * s;
    const int w0 = bin_start(j, W, s), w1 = bin_end(j, W, s);
    const int h0 = bin_start(i, H, s), h1 = bin_end(i, H, s);
    const float *acc = rowacc + ((int64_t)p2 * lv.rows + lv.row_off[k] + i) * W;
    float a = 0.f;
    for (int ww = w0; ww < w1; ++ww) a += acc[ww];
    pooled[planes * lv.bin_off[k] + plane2 * (s * s) + local] = a / (float)((h1 - h0) * (w1 - w0));
  }
}

// One workgroup per plane.  Per level and coordinate, the (<= 2, since H, W >= s) overlapping bins that contain it
// and their 1/length are tabulated in LDS once; every element then needs 4 levels x (<= 2 x 2) table look-ups
// instead of re-deriving bin edges with integer divisions.
__global__ __launch_bounds__(kThreads) void ppm_pool_bwd_kernel(const float *__restrict__ g,
                                                               float *__restrict__ dx, int64_t planes,
                                                               int H, int W, Levels lv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // layout: gl[lv.bins] | first[L*(H+W)] (int) | count[L*(H+W)] (int) | inv[L*(H+W)*2]
  const int HWsum = H + W;
  float *gl = sm;
  int *first = reinterpret_cast<int *>(sm + lv.bins);
  int *count = first + lv.n * HWsum;
  float *inv = reinterpret_cast<float *>(count + lv.n * HWsum);
  const int64_t plane = blockIdx.x;
  for (int o = threadIdx.x; o < lv.bins; o += kThreads) {
    int k = 0;
    while (k + 1 < lv.n && o >= lv.bin_off[k + 1]) ++k;
    gl[o] = g[planes * lv.bin_off[k] + plane * (lv.size[k] * lv.size[k]) + (o - lv.bin_off[k])];
  }
  for (int o = threadIdx.x; o < lv.n * HWsum; o += kThreads) {
    const int k = o / HWsum, pos = o - k * HWsum;
    const int s = lv.size[k];
    const int n = pos < H ? H : W;            // rows first, then columns
    const int c = pos < H ? pos : pos - H;
    const int ia = (c * s) / n;
    int f = ia;
    if (ia > 0 && c < bin_end(ia - 1, n, s)) f = ia - 1;
    int cnt = 1;
    if (f + 1 < s && c >= bin_start(f + 1, n, s)) cnt = 2;
    first[o] = f;
    count[o] = cnt;
    inv[2 * o] = 1.f / (float)(bin_end(f, n, s) - bin_start(f, n, s));
    inv[2 * o + 1] = cnt == 2 ? 1.f / (float)(bin_end(f + 1, n, s) - bin_start(f + 1, n, s)) : 0.f;
  }
  __syncthreads();
  float *out = dx + plane * (int64_t)H * W;
  for (int e = threadIdx.x; e < H * W; e += kThreads) {
    const int h = e / W, w = e - h * W;
    float acc = 0.f;
    for (int k = 0; k < lv.n; ++k) {
      const int s = lv.size[k];
      const int ro = k * HWsum + h, co = k * HWsum + H + w;
      const float *gk = gl + lv.bin_off[k];
      const int i0 = first[ro], j0 = first[co];
      float t = gk[i0 * s + j0] * inv[2 * co];
      if (count[co] == 2) t += gk[i0 * s + j0 + 1] * inv[2 * co + 1];
      acc += t * inv[2 * ro];
      if (count[ro] == 2) {
        float u = gk[(i0 + 1) * s + j0] * inv[2 * co];
        if (count[co] == 2) u += gk[(i0 + 1) * s + j0 + 1] * inv[2 * co + 1];
        acc += u * inv[2 * ro + 1];
      }
    }
    out[e] = acc;
  }
}

struct PriorPtrs {
  const float *p[kMaxLevels];
};
struct GradPtrs {
  float *p[kMaxLevels];
};

// cat[b][k*Cout + c] = bilinear(prior_k[b][c]);  cat[b][L*Cout + c'] = feats[b][c']
__global__ __launch_bounds__(kThreads) void ppm_concat_kernel(PriorPtrs pr, const float *__restrict__ feats,
                                                             float *__restrict__ cat, int B, int Cout,
                                                             int Cfeat, int H, int W, Levels lv) {
  const int HW = H * W;
  const int Ctot = lv.n * Cout + Cfeat;
  const int64_t plane = blockIdx.y;  // b * Ctot + channel
  const int b = (int)(plane / Ctot), ch = (int)(plane % Ctot);
  float *out = cat + plane * (int64_t)HW;
  if (ch >= lv.n * Cout) {
    const float *src = feats + ((int64_t)b * Cfeat + (ch - lv.n * Cout)) * HW;
    for (int e = blockIdx.x * kThreads + threadIdx.x; e < HW; e += gridDim.x * kThreads) out[e] = src[e];
    return;
  }
  const int k = ch / Cout, c = ch - k * Cout;
  const int s = lv.size[k];
  const float *src = pr.p[k] + ((int64_t)b * Cout + c) * (s * s);
  const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f;
  const float sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
  for (int e = blockIdx.x * kThreads + threadIdx.x; e < HW; e += gridDim.x * kThreads) {
    const int Y = e / W, X = e - Y * W;
    const Tap ty = tap_of(Y, sy, s), tx = tap_of(X, sx, s);
    out[e] = ty.l0 * (tx.l0 * src[ty.i0 * s + tx.i0] + tx.l1 * src[ty.i0 * s + tx.i1]) +
             ty.l1 * (tx.l0 * src[ty.i1 * s + tx.i0] + tx.l1 * src[ty.i1 * s + tx.i1]);
  }
}

// one workgroup per (b, level, c) plane of gcat: gprior[y][x] = sum_{Y,X} wy(Y,y) wx(X,x) gcat[Y][X]
__global__ __launch_bounds__(kThreads) void ppm_concat_bwd_kernel(const float *__restrict__ gcat, GradPtrs gp,
                                                                 int B, int Cout, int Cfeat, int H, int W,
                                                                 Levels lv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // plane[H*W] + rowacc[H * smax]
  const int HW = H * W;
  const int Ctot = lv.n * Cout + Cfeat;
  const int idx = blockIdx.x;  // b * (L*Cout) + k*Cout + c
  const int b = idx / (lv.n * Cout), ch = idx - b * (lv.n * Cout);
  const int k = ch / Cout, c = ch - k * Cout;
  const int s = lv.size[k];
  float *plane = sm;
  float *rowacc = sm + HW;
  const float *src = gcat + ((int64_t)b * Ctot + ch) * HW;
  for (int e = threadIdx.x; e < HW; e += kThreads) plane[e] = src[e];
  __syncthreads();
  const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f;
  const float sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
  for (int o = threadIdx.x; o < H * s; o += kThreads) {
    const int Y = o / s, x = o - Y * s;
    float a = 0.f;
    for (int X = 0; X < W; ++X) {
      const Tap t = tap_of(X, sx, s);
      float wt = 0.f;
      if (t.i0 == x) wt += t.l0;
      if (t.i1 == x) wt += t.l1;
      if (wt != 0.f) a += wt * plane[Y * W + X];
    }
    rowacc[o] = a;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < s * s; o += kThreads) {
    const int y = o / s, x = o - y * s;
    float a = 0.f;
    for (int Y = 0; Y < H; ++Y) {
      const Tap t = tap_of(Y, sy, s);
      float wt = 0.f;
      if (t.i0 == y) wt += t.l0;
      if (t.i1 == y) wt += t.l1;
      if (wt != 0.f) a += wt * rowacc[Y * s + x];
    }
    gp.p[k][((int64_t)b * Cout + c) * (s * s) + o] = a;
  }
}


// =============================================================================================
// Channels-last (NHWC) forms.  Both networks keep their activations channels-last (MIOpen's fp32 kernels on gfx950
// are NHWC-native); with only the NCHW kernels above the pyramid module cost four layout copies per network and step
// (feature map -> NCHW, concatenated tensor -> NHWC, and the same for the gradients: 277 + 554 MB for the teacher
// alone, ~1.5 ms per step).  Layouts: feats (B, H, W, C); pooled level k (B, s, s, C); priors (B, s, s, Cout);
// cat (B, H, W, L*Cout + Cfeat); every channel count a multiple of 4 (a thread moves channel quads).
//   pool      the bin edges of all levels cut each axis into <= kMaxCuts segments inside which the set of covering
//             bins is constant; ONE read of the feature map produces the (segment x segment) cell sums
//             (ppm_cells_nhwc), a tiny second kernel adds the cells of every bin (ppm_bins_nhwc).
//   pool_bwd  per pixel: <= 2 x 2 covering bins per level, looked up from per-coordinate tables in LDS.
//   concat    per pixel and channel quad: four-tap bilinear from the (L2-resident) priors, or a copy of feats.
//   concat_bwd  separable pull-back: row pass (reduce over X, fused with the contiguous copy of the feature-map
//             slice of the gradient) -> column pass (reduce over Y).  No atomics anywhere.
// =============================================================================================
constexpr int kMaxCuts = 40;   // 2 * sum(sizes) + 1 for sizes (1, 2, 3, 6) is 25

struct Cuts {
  int ny, nx;                  // number of segments per axis
  int y[kMaxCuts], x[kMaxCuts];
};

static bool make_cuts(int H, int W, const Levels &lv, Cuts &c) {
  auto build = [&](int n, int *out, int &cnt) {
    bool mark[4096];
    if (n + 1 > 4096) return false;
    for (int i = 0; i <= n; ++i) mark[i] = false;
    mark[0] = mark[n] = true;
    for (int k = 0; k < lv.n; ++k)
      for (int i = 0; i < lv.size[k]; ++i) {
        mark[(i * n) / lv.size[k]] = true;
        mark[((i + 1) * n + lv.size[k] - 1) / lv.size[k]] = true;
      }
    cnt = 0;
    for (int i = 0; i <= n; ++i)
      if (mark[i]) {
        if (cnt >= kMaxCuts) return false;
        out[cnt++] = i;
      }
    cnt -= 1;  // segments
    return cnt >= 1;
  };
  return build(H, c.y, c.ny) && build(W, c.x, c.nx);
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 b) { return make_float4(s * a.x + b.x, s * a.y + b.y, s * a.z + b.z, s * a.w + b.w); }

// cells[b][vy][vx][C] = sum of feats over the pixels of segment cell (vy, vx).   grid (ny*nx, B)
__global__ __launch_bounds__(kThreads) void ppm_cells_nhwc_kernel(const float *__restrict__ x, float *__restrict__ cells,
                                                                 int H, int W, int C4, Cuts cu) {
  __shared__ float4 red[kThreads];
  const int cell = blockIdx.x, b = blockIdx.y;
  const int vy = cell / cu.nx, vx = cell - vy * cu.nx;
  const int h0 = cu.y[vy], h1 = cu.y[vy + 1], w0 = cu.x[vx], w1 = cu.x[vx + 1];
  const int cw = w1 - w0, npix = (h1 - h0) * cw;
  const int Q = C4 < kThreads ? C4 : kThreads;   // quads handled per pass
  const int P = kThreads / Q;                    // pixel lanes (1 when C4 >= 256)
  const int t = threadIdx.x;
  const int ql = t % Q, pl = t / Q;
  const float *base = x + (int64_t)b * H * W * C4 * 4;
  float *out = cells + ((int64_t)b * cu.ny * cu.nx + cell) * C4 * 4;
  for (int q0 = 0; q0 < C4; q0 += Q) {
    const int q = q0 + ql;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < C4 && pl < P) {
      for (int p = pl; p < npix; p += 4 * P) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pp = p + u * P;
          if (pp < npix) {
            const int hh = h0 + pp / cw, ww = w0 + pp % cw;
            v[u] = *reinterpret_cast<const float4 *>(base + ((int64_t)hh * W + ww) * C4 * 4 + q * 4);
          } else {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
        acc = f4add(acc, f4add(f4add(v[0], v[1]), f4add(v[2], v[3])));
      }
    }
    if (P > 1) {
      red[t] = acc;
      __syncthreads();
      if (pl == 0 && q < C4) {
        for (int r = 1; r < P; ++r) acc = f4add(acc, red[r * Q + ql]);
        *reinterpret_cast<float4 *>(out + q * 4) = acc;
      }
      __syncthreads();
    } else if (q < C4) {
      *reinterpret_cast<float4 *>(out + q * 4) = acc;
    }
  }
}

// pooled level k, (B, s, s, C): mean over the bin = sum of its cells / area.   grid (bins, B, ceil(C4 / 64)): a wave of
// channel quads x four groups that split the bin's cell list; the groups' sums meet in LDS in group order
__global__ __launch_bounds__(kThreads) void ppm_bins_nhwc_kernel(const float *__restrict__ cells, float *__restrict__ pooled,
                                                                int B, int H, int W, int C4, Cuts cu, Levels lv) {
  __shared__ float4 part[kThreads];
  const int bin = blockIdx.x, b = blockIdx.y;
  int k = 0;
  while (k + 1 < lv.n && bin >= lv.bin_off[k + 1]) ++k;
  const int s = lv.size[k], local = bin - lv.bin_off[k];
  const int i = local / s, j = local - i * s;
  const int h0 = bin_start(i, H, s), h1 = bin_end(i, H, s), w0 = bin_start(j, W, s), w1 = bin_end(j, W, s);
  int va = 0, vb = 0, ua = 0, ub = 0;
  while (cu.y[va] < h0) ++va;
  vb = va;
  while (cu.y[vb] < h1) ++vb;
  while (cu.x[ua] < w0) ++ua;
  ub = ua;
  while (cu.x[ub] < w1) ++ub;
  const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
  const float *src = cells + (int64_t)b * cu.ny * cu.nx * C4 * 4;
  float *dst = pooled + ((int64_t)B * lv.bin_off[k] + ((int64_t)b * s * s + local)) * C4 * 4;
  constexpr int kGroups = kThreads / 64;
  const int ql = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int q = blockIdx.z * 64 + ql;
  const int nu = ub - ua, n = (vb - va) * nu;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < C4)
    for (int c = g; c < n; c += kGroups) {
      const int v = va + c / nu, u = ua + c % nu;
      acc = f4add(acc, *reinterpret_cast<const float4 *>(src + ((int64_t)v * cu.nx + u) * C4 * 4 + q * 4));
    }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (g == 0 && q < C4) {
    for (int r = 1; r < kGroups; ++r) acc = f4add(acc, part[r * 64 + ql]);
    *reinterpret_cast<float4 *>(dst + q * 4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

// dx[b][h][w][c] = sum over levels and covering bins of g[bin][c] / area(bin).   grid (H, B)
__global__ __launch_bounds__(kThreads) void ppm_pool_bwd_nhwc_kernel(const float *__restrict__ g, float *__restrict__ dx,
                                                                    int B, int H, int W, int C4, Levels lv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // first[L*W] (int) | count[L*W] (int) | inv[L*W*2]
  int *first = reinterpret_cast<int *>(sm);
  int *count = first + lv.n * W;
  float *inv = reinterpret_cast<float *>(count + lv.n * W);
  const int h = blockIdx.x, b = blockIdx.y;
  for (int o = threadIdx.x; o < lv.n * W; o += kThreads) {
    const int k = o / W, c = o - k * W, s = lv.size[k];
    const int ia = (c * s) / W;
    int f = ia;
    if (ia > 0 && c < bin_end(ia - 1, W, s)) f = ia - 1;
    int cnt = 1;
    if (f + 1 < s && c >= bin_start(f + 1, W, s)) cnt = 2;
    first[o] = f;
    count[o] = cnt;
    inv[2 * o] = 1.f / (float)(bin_end(f, W, s) - bin_start(f, W, s));
    inv[2 * o + 1] = cnt == 2 ? 1.f / (float)(bin_end(f + 1, W, s) - bin_start(f + 1, W, s)) : 0.f;
  }
  __syncthreads();
  // the row's own covering bins per level (uniform over the workgroup)
  int rf[kMaxLevels], rc[kMaxLevels];
  float ri[kMaxLevels][2];
  for (int k = 0; k < lv.n; ++k) {
    const int s = lv.size[k];
    const int ia = (h * s) / H;
    int f = ia;
    if (ia > 0 && h < bin_end(ia - 1, H, s)) f = ia - 1;
    rc[k] = (f + 1 < s && h >= bin_start(f + 1, H, s)) ? 2 : 1;
    rf[k] = f;
    ri[k][0] = 1.f / (float)(bin_end(f, H, s) - bin_start(f, H, s));
    ri[k][1] = rc[k] == 2 ? 1.f / (float)(bin_end(f + 1, H, s) - bin_start(f + 1, H, s)) : 0.f;
  }
  float *out = dx + ((int64_t)b * H + h) * W * C4 * 4;
  for (int item = threadIdx.x; item < W * C4; item += kThreads) {
    const int w = item / C4, q = item - w * C4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < lv.n; ++k) {
      const int s = lv.size[k], o = k * W + w;
      const float *gk = g + ((int64_t)B * lv.bin_off[k] + (int64_t)b * s * s) * C4 * 4 + q * 4;
      for (int a = 0; a < rc[k]; ++a)
        for (int c = 0; c < count[o]; ++c) {
          const float wt = ri[k][a] * inv[2 * o + c];
          acc = f4fma(wt, *reinterpret_cast<const float4 *>(gk + (int64_t)((rf[k] + a) * s + first[o] + c) * C4 * 4), acc);
        }
    }
    *reinterpret_cast<float4 *>(out + (int64_t)item * 4) = acc;
  }
}

// cat[b][h][w][k*Cout + c] = bilinear(prior_k[b][:, :, c]);  cat[b][h][w][L*Cout + c'] = feats[b][h][w][c'].  grid (chunks, H, B)
__global__ __launch_bounds__(kThreads) void ppm_concat_nhwc_kernel(PriorPtrs pr, const float *__restrict__ feats,
                                                                  float *__restrict__ cat, int Cout4, int Cf4, int H,
                                                                  int W, Levels lv) {
  const int h = blockIdx.y, b = blockIdx.z;
  const int Ct4 = lv.n * Cout4 + Cf4;
  const int64_t row = ((int64_t)b * H + h) * W;
  for (int item = blockIdx.x * kThreads + threadIdx.x; item < W * Ct4; item += gridDim.x * kThreads) {
    const int w = item / Ct4, q = item - w * Ct4;
    float4 v;
    if (q >= lv.n * Cout4) {
      v = *reinterpret_cast<const float4 *>(feats + ((row + w) * Cf4 + (q - lv.n * Cout4)) * 4);
    } else {
      const int k = q / Cout4, cq = q - k * Cout4, s = lv.size[k];
      const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f;
      const float sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
      const Tap ty = tap_of(h, sy, s), tx = tap_of(w, sx, s);
      const float *src = pr.p[k] + (int64_t)b * s * s * Cout4 * 4 + cq * 4;
      const float4 a00 = *reinterpret_cast<const float4 *>(src + (int64_t)(ty.i0 * s + tx.i0) * Cout4 * 4);
      const float4 a01 = *reinterpret_cast<const float4 *>(src + (int64_t)(ty.i0 * s + tx.i1) * Cout4 * 4);
      const float4 a10 = *reinterpret_cast<const float4 *>(src + (int64_t)(ty.i1 * s + tx.i0) * Cout4 * 4);
      const float4 a11 = *reinterpret_cast<const float4 *>(src + (int64_t)(ty.i1 * s + tx.i1) * Cout4 * 4);
      // same association as the NCHW kernel: ty.l0 * (tx.l0 * a00 + tx.l1 * a01) + ty.l1 * (tx.l0 * a10 + tx.l1 * a11)
      v.x = ty.l0 * (tx.l0 * a00.x + tx.l1 * a01.x) + ty.l1 * (tx.l0 * a10.x + tx.l1 * a11.x);
      v.y = ty.l0 * (tx.l0 * a00.y + tx.l1 * a01.y) + ty.l1 * (tx.l0 * a10.y + tx.l1 * a11.y);
      v.z = ty.l0 * (tx.l0 * a00.z + tx.l1 * a01.z) + ty.l1 * (tx.l0 * a10.z + tx.l1 * a11.z);
      v.w = ty.l0 * (tx.l0 * a00.w + tx.l1 * a01.w) + ty.l1 * (tx.l0 * a10.w + tx.l1 * a11.w);
    }
    *reinterpret_cast<float4 *>(cat + ((row + w) * Ct4 + q) * 4) = v;
  }
}

// row pass of the pull-back: rowacc[b][Y][rowbin(k, x)][Cout] = sum_X wx(X, x) * gcat[b][Y][X][k*Cout + c], and the
// feature-map slice of gcat copied out contiguously.   grid (H, B)
__global__ __launch_bounds__(kThreads) void ppm_concat_bwd_rows_nhwc_kernel(const float *__restrict__ gcat,
                                                                           float *__restrict__ rowacc,
                                                                           float *__restrict__ gfeats, int Cout4,
                                                                           int Cf4, int H, int W, Levels lv) {
  const int Y = blockIdx.x, b = blockIdx.y;
  const int Ct4 = lv.n * Cout4 + Cf4;
  const float *src = gcat + ((int64_t)b * H + Y) * W * Ct4 * 4;
  if (gfeats != nullptr) {
    float *dst = gfeats + ((int64_t)b * H + Y) * W * Cf4 * 4;
    for (int item = threadIdx.x; item < W * Cf4; item += kThreads) {
      const int w = item / Cf4, q = item - w * Cf4;
      *reinterpret_cast<float4 *>(dst + (int64_t)item * 4) =
          *reinterpret_cast<const float4 *>(src + ((int64_t)w * Ct4 + lv.n * Cout4 + q) * 4);
    }
  }
  if (rowacc == nullptr) return;
  float *out = rowacc + ((int64_t)b * H + Y) * lv.rows * Cout4 * 4;
  for (int item = threadIdx.x; item < lv.rows * Cout4; item += kThreads) {
    const int rb = item / Cout4, cq = item - rb * Cout4;
    int k = 0;
    while (k + 1 < lv.n && rb >= lv.row_off[k + 1]) ++k;
    const int s = lv.size[k], x = rb - lv.row_off[k];
    const float sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int X = 0; X < W; ++X) {
      const Tap t = tap_of(X, sx, s);
      float wt = 0.f;
      if (t.i0 == x) wt += t.l0;
      if (t.i1 == x) wt += t.l1;
      if (wt != 0.f) acc = f4fma(wt, *reinterpret_cast<const float4 *>(src + ((int64_t)X * Ct4 + k * Cout4 + cq) * 4), acc);
    }
    *reinterpret_cast<float4 *>(out + (int64_t)item * 4) = acc;
  }
}

// column pass: gprior_k[b][y][x][c] = sum_Y wy(Y, y) * rowacc[b][Y][rowbin(k, x)][c].   grid (bins, B)
__global__ __launch_bounds__(kThreads) void ppm_concat_bwd_cols_nhwc_kernel(const float *__restrict__ rowacc, GradPtrs gp,
                                                                           int Cout4, int H, Levels lv) {
  const int bin = blockIdx.x, b = blockIdx.y;
  int k = 0;
  while (k + 1 < lv.n && bin >= lv.bin_off[k + 1]) ++k;
  const int s = lv.size[k], local = bin - lv.bin_off[k];
  const int y = local / s, x = local - y * s;
  const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f;
  float *dst = gp.p[k] + ((int64_t)b * s * s + local) * Cout4 * 4;
  for (int cq = threadIdx.x; cq < Cout4; cq += kThreads) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int Y = 0; Y < H; ++Y) {
      const Tap t = tap_of(Y, sy, s);
      float wt = 0.f;
      if (t.i0 == y) wt += t.l0;
      if (t.i1 == y) wt += t.l1;
      if (wt != 0.f)
        acc = f4fma(wt, *reinterpret_cast<const float4 *>(rowacc + (((int64_t)b * H + Y) * lv.rows + lv.row_off[k] + x) * Cout4 * 4 + cq * 4), acc);
    }
    *reinterpret_cast<float4 *>(dst + cq * 4) = acc;
  }
}


// ---- pyramid priors folded through the 3x3 bottleneck convolution (channels-last) ---------------------------------
// bottleneck(cat(up(prior_1) .. up(prior_L), feats)) is linear in every input, and up(prior_k) is a bilinear function
// of s_k x s_k values, so  conv3x3(cat) = conv3x3(feats; W_feats) + sum_k fold_k  with
//     Z_k[b][jy][jx][tap][co] = sum_ci prior_k[b][jy][jx][ci] * W[co][k*Cm + ci][tap]          (a (B s^2) x Cm x 9 Cout GEMM)
//     fold_k[b][y][x][co]     = sum_{tap = (ty, tx), (y', x') = (y + ty - 1, x + tx - 1) inside the map}
//                               sum_{jy in taps(y'), jx in taps(x')} ly(y', jy) * lx(x', jx) * Z_k[b][jy][jx][tap][co]
// (zero padding = taps outside the map dropped).  Half of the bottleneck's input channels (2048 of 4096 in the teacher,
// 512 of 1024 in the student) never meet the convolution: 638 GFLOP per teacher forward at batch 8 become a 1.9 GFLOP
// GEMM plus this kernel, and the concatenated tensor (554 MB) is never written.
// Separable evaluation: a thread owns one (b, x, channel quad) column; it first contracts the x direction into
// T[ty][jy] (3 * sum_k s_k float4 entries, kept in LDS as private indexed storage -- no sharing, no barriers), then walks
// down y adding <= 6 entries per level to the convolution output in place.
constexpr int kFoldThreads = 64;

struct FoldPtrs {
  const float *z[kMaxLevels];
};
struct FoldGradPtrs {
  float *z[kMaxLevels];
};

__device__ __forceinline__ float4 lds_get(const float4 *t, int e) { return t[e * kFoldThreads + threadIdx.x]; }

__global__ __launch_bounds__(kFoldThreads) void ppm_fold_nhwc_kernel(FoldPtrs zp, float *__restrict__ out, int B, int H,
                                                                     int W, int C4, int ychunk, Levels lv, int k0, int k1,
                                                                     int64_t ldz) {
  extern __shared__ __attribute__((aligned(16))) float4 tcol[];  // [3 * rows of levels k0..k1-1][kFoldThreads]
  const int64_t col = (int64_t)blockIdx.x * kFoldThreads + threadIdx.x;
  if (col >= (int64_t)B * W * C4) return;
  const int q = (int)(col % C4), x = (int)((col / C4) % W), b = (int)(col / ((int64_t)C4 * W));
  const int e0 = 3 * lv.row_off[k0];
  for (int k = k0; k < k1; ++k) {
    const int s = lv.size[k];
    const float sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
    const float *zk = zp.z[k] + (int64_t)b * s * s * ldz + q * 4;
    Tap tx3[3];
    bool vx[3];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int xp = x + tx - 1;
      vx[tx] = xp >= 0 && xp < W;
      tx3[tx] = tap_of(vx[tx] ? xp : 0, sx, s);
    }
    for (int ty = 0; ty < 3; ++ty)
      for (int jy = 0; jy < s; ++jy) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
          if (vx[tx]) {
            const float *r = zk + (int64_t)(jy * s) * ldz + (ty * 3 + tx) * C4 * 4;
            acc = f4fma(tx3[tx].l0, *reinterpret_cast<const float4 *>(r + (int64_t)tx3[tx].i0 * ldz), acc);
            acc = f4fma(tx3[tx].l1, *reinterpret_cast<const float4 *>(r + (int64_t)tx3[tx].i1 * ldz), acc);
          }
        tcol[(3 * lv.row_off[k] - e0 + ty * s + jy) * kFoldThreads + threadIdx.x] = acc;
      }
  }
  const int y0 = blockIdx.y * ychunk, y1 = y0 + ychunk < H ? y0 + ychunk : H;
  const int64_t ystride = (int64_t)W * C4 * 4;
  float *o = out + (((int64_t)b * H + y0) * W + x) * C4 * 4 + q * 4;
  constexpr int U = 4;
  for (int y = y0; y < y1; y += U, o += U * ystride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (y + u < y1) v[u] = *reinterpret_cast<const float4 *>(o + u * ystride);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (y + u >= y1) break;
      for (int k = k0; k < k1; ++k) {
        const int s = lv.size[k];
        const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int yp = y + u + ty - 1;
          if (yp < 0 || yp >= H) continue;
          const Tap t = tap_of(yp, sy, s);
          const int e = 3 * lv.row_off[k] - e0 + ty * s;
          v[u] = f4fma(t.l0, lds_get(tcol, e + t.i0), v[u]);
          v[u] = f4fma(t.l1, lds_get(tcol, e + t.i1), v[u]);
        }
      }
      *reinterpret_cast<float4 *>(o + u * ystride) = v[u];
    }
  }
}

// transpose, column pass: TT[b][chunk][e = (k, ty, jy)][x][c] = sum_{y in chunk} ly(y + ty - 1, jy) * g[b][y][x][c]
__global__ __launch_bounds__(kFoldThreads) void ppm_fold_bwd_cols_nhwc_kernel(const float *__restrict__ g,
                                                                              float *__restrict__ ws, int B, int H, int W,
                                                                              int C4, int ychunk, Levels lv) {
  extern __shared__ __attribute__((aligned(16))) float4 tcol[];
  const int64_t col = (int64_t)blockIdx.x * kFoldThreads + threadIdx.x;
  if (col >= (int64_t)B * W * C4) return;
  const int q = (int)(col % C4), x = (int)((col / C4) % W), b = (int)(col / ((int64_t)C4 * W));
  const int E = 3 * lv.rows;
  for (int e = 0; e < E; ++e) tcol[e * kFoldThreads + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int y0 = blockIdx.y * ychunk, y1 = y0 + ychunk < H ? y0 + ychunk : H;
  const int64_t ystride = (int64_t)W * C4 * 4;
  const float *src = g + (((int64_t)b * H + y0) * W + x) * C4 * 4 + q * 4;
  constexpr int U = 4;
  for (int y = y0; y < y1; y += U, src += U * ystride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (y + u < y1) v[u] = *reinterpret_cast<const float4 *>(src + u * ystride);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (y + u >= y1) break;
      for (int k = 0; k < lv.n; ++k) {
        const int s = lv.size[k];
        const float sy = H > 1 ? (float)(s - 1) / (float)(H - 1) : 0.f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int yp = y + u + ty - 1;
          if (yp < 0 || yp >= H) continue;
          const Tap t = tap_of(yp, sy, s);
          const int e = 3 * lv.row_off[k] + ty * s;
          float4 *a0 = tcol + (e + t.i0) * kFoldThreads + threadIdx.x;
          *a0 = f4fma(t.l0, v[u], *a0);
          float4 *a1 = tcol + (e + t.i1) * kFoldThreads + threadIdx.x;
          *a1 = f4fma(t.l1, v[u], *a1);
        }
      }
    }
  }
  float *dst = ws + ((((int64_t)blockIdx.y * B + b) * E) * W + x) * C4 * 4 + q * 4;
  for (int e = 0; e < E; ++e) *reinterpret_cast<float4 *>(dst + (int64_t)e * W * C4 * 4) = tcol[e * kFoldThreads + threadIdx.x];
}

// transpose, bin pass: dZ_k[b][jy][jx][tap][c] = sum_chunks sum_x lx(x + tx - 1, jx) * TT[..][(k, ty, jy)][x][c], fixed order.
// grid (lv.bins * 9, B): one workgroup per (bin, tap, image); threads = channel quads x column groups, the groups' partial
// sums are combined through LDS in group order (deterministic)
__global__ __launch_bounds__(kThreads) void ppm_fold_bwd_bins_nhwc_kernel(const float *__restrict__ ws, FoldGradPtrs gz,
                                                                         int B, int W, int C4, int chunks, Levels lv,
                                                                         int64_t ldz) {
  __shared__ float4 part[kThreads];
  const int bin = blockIdx.x / 9, tap = blockIdx.x - bin * 9, b = blockIdx.y;
  int k = 0;
  while (k + 1 < lv.n && bin >= lv.bin_off[k + 1]) ++k;
  const int s = lv.size[k], j = bin - lv.bin_off[k], jy = j / s, jx = j - jy * s;
  const float sx = W > 1 ? (float)(s - 1) / (float)(W - 1) : 0.f;
  const int E = 3 * lv.rows, ty = tap / 3, tx = tap - ty * 3;
  const int e = 3 * lv.row_off[k] + ty * s + jy;
  // columns x' = x + tx - 1 whose bilinear taps touch bin column jx: (jx - 1) / sx < x' < (jx + 1) / sx (one column of
  // margin either side; weights outside the support are exactly zero, so the margin only costs a load)
  int lo = 0, hi = W - 1;
  if (s > 1) {
    lo = (int)floorf((float)(jx - 1) / sx) - 1 - (tx - 1);
    hi = (int)ceilf((float)(jx + 1) / sx) + 1 - (tx - 1);
    if (lo < 0) lo = 0;
    if (hi > W - 1) hi = W - 1;
  }
  const int qn = C4 < kThreads ? C4 : kThreads, G = kThreads / qn;
  const int ql = threadIdx.x % qn, xg = threadIdx.x / qn;
  for (int q0 = 0; q0 < C4; q0 += qn) {
    const int q = q0 + ql;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xg < G && q < C4)
      for (int c = 0; c < chunks; ++c) {
        const float *src = ws + ((((int64_t)c * B + b) * E + e) * W) * C4 * 4 + q * 4;
        for (int x = lo + xg; x <= hi; x += G) {
          const int xp = x + tx - 1;
          if (xp < 0 || xp >= W) continue;
          const Tap t = tap_of(xp, sx, s);
          float wt = 0.f;
          if (t.i0 == jx) wt += t.l0;
          if (t.i1 == jx) wt += t.l1;
          acc = f4fma(wt, *reinterpret_cast<const float4 *>(src + (int64_t)x * C4 * 4), acc);
        }
      }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (xg == 0 && q < C4) {
      float4 sum = part[ql];
      for (int g = 1; g < G; ++g) {
        const float4 v = part[g * qn + ql];
        sum.x += v.x;
        sum.y += v.y;
        sum.z += v.z;
        sum.w += v.w;
      }
      *reinterpret_cast<float4 *>(gz.z[k] + (((int64_t)b * s + jy) * s + jx) * ldz + tap * C4 * 4 + q * 4) = sum;
    }
    __syncthreads();
  }
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

int64_t skd_ppm_pooled_floats(int planes, int nsizes, const int *sizes) {
  Levels lv;
  if (planes <= 0 || !make_levels(nsizes, sizes, lv)) return 0;
  return (int64_t)planes * lv.bins;
}

int skd_ppm_pool(int planes, int H, int W, int nsizes, const int *sizes, const float *x, float *pooled,
                 skd_stream_t stream) {
  Levels lv;
  if (planes <= 0 || H <= 0 || W <= 0 || !x || !pooled || !make_levels(nsizes, sizes, lv)) return 0;
  if (W > kThreads) return 0;  // feature maps of this path are 33..129 wide
  const int ppw = kThreads / W;
  const size_t smem = sizeof(float) * (size_t)ppw * lv.rows * W;
  if (smem > 64 * 1024) return 0;
  ppm_pool_kernel<<<dim3((unsigned)cdiv(planes, ppw)), dim3(kThreads), smem, as_stream(stream)>>>(
      x, pooled, planes, H, W, ppw, lv);
  return ok();
}

int skd_ppm_pool_backward(int planes, int H, int W, int nsizes, const int *sizes, const float *gpooled,
                          float *dx, skd_stream_t stream) {
  Levels lv;
  if (planes <= 0 || H <= 0 || W <= 0 || !gpooled || !dx || !make_levels(nsizes, sizes, lv)) return 0;
  for (int k = 0; k < nsizes; ++k)
    if (sizes[k] > H || sizes[k] > W) return 0;  // a pixel would sit in more than two overlapping bins
  const size_t smem = sizeof(float) * ((size_t)lv.bins + (size_t)lv.n * (H + W) * 4);
  if (smem > 64 * 1024) return 0;
  ppm_pool_bwd_kernel<<<dim3((unsigned)planes), dim3(kThreads), smem, as_stream(stream)>>>(gpooled, dx, planes, H, W, lv);
  return ok();
}

int skd_ppm_concat(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes,
                   const float *const *priors, const float *feats, float *cat, skd_stream_t stream) {
  Levels lv;
  if (B <= 0 || Cout <= 0 || Cfeat < 0 || H <= 0 || W <= 0 || !priors || !cat || !make_levels(nsizes, sizes, lv)) return 0;
  if (Cfeat > 0 && !feats) return 0;
  PriorPtrs pr;
  for (int k = 0; k < kMaxLevels; ++k) pr.p[k] = k < nsizes ? priors[k] : nullptr;
  for (int k = 0; k < nsizes; ++k)
    if (!pr.p[k]) return 0;
  const int64_t planes = (int64_t)B * (nsizes * Cout + Cfeat);
  if (planes > 65535 * 64) return 0;
  int gx = (int)cdiv((int64_t)H * W, kThreads * 4);
  if (gx < 1) gx = 1;
  hipStream_t st = as_stream(stream);
  // gridDim.y <= 65535: fold planes in slabs (B * Ctot is 8192 .. 32768 on this path)
  if (planes > 65535) return 0;
  ppm_concat_kernel<<<dim3(gx, (unsigned)planes), dim3(kThreads), 0, st>>>(pr, feats, cat, B, Cout, Cfeat, H, W, lv);
  return ok();
}

int skd_ppm_concat_backward(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes,
                            const float *gcat, float *const *gpriors, skd_stream_t stream) {
  Levels lv;
  if (B <= 0 || Cout <= 0 || Cfeat < 0 || H <= 0 || W <= 0 || !gcat || !gpriors || !make_levels(nsizes, sizes, lv)) return 0;
  GradPtrs gp;
  int smax = 0;
  for (int k = 0; k < kMaxLevels; ++k) gp.p[k] = k < nsizes ? gpriors[k] : nullptr;
  for (int k = 0; k < nsizes; ++k) {
    if (!gp.p[k]) return 0;
    if (sizes[k] > smax) smax = sizes[k];
  }
  const size_t smem = sizeof(float) * ((size_t)H * W + (size_t)H * smax);
  if (smem > 150 * 1024) return 0;
  static PerDeviceFlag attr_flag;
  bool *attr_set = attr_flag.get();
  if (attr_set == nullptr) return 0;
  if (!*attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ppm_concat_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *attr_set = true;
  }
  ppm_concat_bwd_kernel<<<dim3((unsigned)(B * nsizes * Cout)), dim3(kThreads), smem, as_stream(stream)>>>(
      gcat, gp, B, Cout, Cfeat, H, W, lv);
  return ok();
}


// ---- channels-last entries (layouts in the section header above) ----------------------------------------------

int64_t skd_ppm_nhwc_workspace_floats(int B, int C, int Cout, int H, int W, int nsizes, const int *sizes) {
  Levels lv;
  Cuts cu;
  if (B <= 0 || H <= 0 || W <= 0 || !make_levels(nsizes, sizes, lv) || !make_cuts(H, W, lv, cu)) return 0;
  const int64_t cells = (int64_t)B * cu.ny * cu.nx * (C > 0 ? C : 0);
  const int64_t rows = (int64_t)B * H * lv.rows * (Cout > 0 ? Cout : 0);
  return cells > rows ? cells : rows;
}

int skd_ppm_pool_nhwc(int B, int C, int H, int W, int nsizes, const int *sizes, const float *x, float *pooled,
                      float *workspace, skd_stream_t stream) {
  Levels lv;
  Cuts cu;
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || !x || !pooled || !workspace) return 0;
  if (!make_levels(nsizes, sizes, lv) || !make_cuts(H, W, lv, cu)) return 0;
  hipStream_t st = as_stream(stream);
  ppm_cells_nhwc_kernel<<<dim3((unsigned)(cu.ny * cu.nx), (unsigned)B), dim3(kThreads), 0, st>>>(x, workspace, H, W, C / 4, cu);
  ppm_bins_nhwc_kernel<<<dim3((unsigned)lv.bins, (unsigned)B, (unsigned)cdiv(C / 4, 64)), dim3(kThreads), 0, st>>>(workspace, pooled, B, H, W, C / 4, cu, lv);
  return ok();
}

int skd_ppm_pool_backward_nhwc(int B, int C, int H, int W, int nsizes, const int *sizes, const float *gpooled, float *dx,
                               skd_stream_t stream) {
  Levels lv;
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || !gpooled || !dx || !make_levels(nsizes, sizes, lv)) return 0;
  for (int k = 0; k < nsizes; ++k)
    if (sizes[k] > H || sizes[k] > W) return 0;
  const size_t smem = sizeof(float) * (size_t)lv.n * W * 4;
  if (smem > 64 * 1024) return 0;
  ppm_pool_bwd_nhwc_kernel<<<dim3((unsigned)H, (unsigned)B), dim3(kThreads), smem, as_stream(stream)>>>(gpooled, dx, B, H, W, C / 4, lv);
  return ok();
}

int skd_ppm_concat_nhwc(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes, const float *const *priors,
                        const float *feats, float *cat, skd_stream_t stream) {
  Levels lv;
  if (B <= 0 || Cout <= 0 || (Cout & 3) || Cfeat < 0 || (Cfeat & 3) || H <= 0 || W <= 0 || !priors || !cat) return 0;
  if (!make_levels(nsizes, sizes, lv) || (Cfeat > 0 && !feats) || H > 65535 || B > 65535) return 0;
  PriorPtrs pr;
  for (int k = 0; k < kMaxLevels; ++k) pr.p[k] = k < nsizes ? priors[k] : nullptr;
  for (int k = 0; k < nsizes; ++k)
    if (!pr.p[k]) return 0;
  const int Ct4 = (nsizes * Cout + Cfeat) / 4;
  int gx = (int)cdiv((int64_t)W * Ct4, kThreads * 16);
  if (gx < 1) gx = 1;
  ppm_concat_nhwc_kernel<<<dim3((unsigned)gx, (unsigned)H, (unsigned)B), dim3(kThreads), 0, as_stream(stream)>>>(
      pr, feats, cat, Cout / 4, Cfeat / 4, H, W, lv);
  return ok();
}

int skd_ppm_concat_backward_nhwc(int B, int Cout, int Cfeat, int H, int W, int nsizes, const int *sizes, const float *gcat,
                                 float *const *gpriors, float *gfeats, float *workspace, skd_stream_t stream) {
  Levels lv;
  if (B <= 0 || Cout <= 0 || (Cout & 3) || Cfeat < 0 || (Cfeat & 3) || H <= 0 || W <= 0 || !gcat) return 0;
  if (!make_levels(nsizes, sizes, lv) || H > 65535 || B > 65535) return 0;
  if (gpriors && !workspace) return 0;
  GradPtrs gp;
  for (int k = 0; k < kMaxLevels; ++k) gp.p[k] = (gpriors && k < nsizes) ? gpriors[k] : nullptr;
  if (gpriors)
    for (int k = 0; k < nsizes; ++k)
      if (!gp.p[k]) return 0;
  hipStream_t st = as_stream(stream);
  ppm_concat_bwd_rows_nhwc_kernel<<<dim3((unsigned)H, (unsigned)B), dim3(kThreads), 0, st>>>(
      gcat, gpriors ? workspace : nullptr, Cfeat > 0 ? gfeats : nullptr, Cout / 4, Cfeat / 4, H, W, lv);
  if (gpriors)
    ppm_concat_bwd_cols_nhwc_kernel<<<dim3((unsigned)lv.bins, (unsigned)B), dim3(kThreads), 0, st>>>(workspace, gp, Cout / 4, H, lv);
  return ok();
}

static int fold_chunks(int B, int H, int W, int C4) {
  // ~1024 single-wave workgroups keep every CU's LDS full once; more y-chunks when the map has few columns
  const int64_t wgs = cdiv((int64_t)B * W * C4, kFoldThreads);
  int chunks = (int)(1024 / (wgs > 0 ? wgs : 1));
  if (chunks < 1) chunks = 1;
  if (chunks > 8) chunks = 8;
  if (chunks > H) chunks = H;
  return chunks;
}

int64_t skd_ppm_fold_nhwc_workspace_floats(int B, int Cout, int H, int W, int nsizes, const int *sizes) {
  Levels lv;
  if (B <= 0 || Cout <= 0 || (Cout & 3) || H <= 0 || W <= 0 || !make_levels(nsizes, sizes, lv)) return 0;
  return (int64_t)fold_chunks(B, H, W, Cout / 4) * B * 3 * lv.rows * W * Cout;
}

int skd_ppm_fold_nhwc(int B, int Cout, int H, int W, int nsizes, const int *sizes, const float *const *z, int64_t ldz,
                      float *out, skd_stream_t stream) {
  Levels lv;
  if (B <= 0 || Cout <= 0 || (Cout & 3) || H <= 0 || W <= 0 || !z || !out || !make_levels(nsizes, sizes, lv)) return 0;
  if (ldz < 9 * (int64_t)Cout || (ldz & 3)) return 0;
  const size_t smem = sizeof(float4) * (size_t)3 * lv.rows * kFoldThreads;
  if (smem > 64 * 1024) return 0;
  FoldPtrs zp;
  for (int k = 0; k < kMaxLevels; ++k) zp.z[k] = k < nsizes ? z[k] : nullptr;
  for (int k = 0; k < nsizes; ++k)
    if (!zp.z[k]) return 0;
  const int C4 = Cout / 4, chunks = fold_chunks(B, H, W, C4), ychunk = (int)cdiv(H, chunks);
  const dim3 grid((unsigned)cdiv((int64_t)B * W * C4, kFoldThreads), (unsigned)cdiv(H, ychunk));
  // The per-column table lives in LDS (48 bytes per level row and thread): all levels of (1, 2, 3, 6) at once would leave
  // one wave per SIMD and, at the teacher's size, 1040 workgroups for 1024 slots.  Two launches over ~half the level rows
  // each (here {1, 2, 3} and {6}) run two waves per SIMD in one round; the second pass over `out` costs less than that.
  int k0 = 0;
  while (k0 < nsizes) {
    int k1 = k0 + 1, rows = lv.size[k0];
    while (k1 < nsizes && (rows + lv.size[k1]) * 2 <= lv.rows + 1) rows += lv.size[k1++];
    const size_t part = sizeof(float4) * (size_t)3 * rows * kFoldThreads;
    ppm_fold_nhwc_kernel<<<grid, dim3(kFoldThreads), part, as_stream(stream)>>>(zp, out, B, H, W, C4, ychunk, lv, k0, k1, ldz);
    k0 = k1;
  }
  return ok();
}

int skd_ppm_fold_backward_nhwc(int B, int Cout, int H, int W, int nsizes, const int *sizes, const float *gout,
                               float *const *gz, int64_t ldz, float *workspace, skd_stream_t stream) {
  Levels lv;
  if (B <= 0 || Cout <= 0 || (Cout & 3) || H <= 0 || W <= 0 || !gout || !gz || !workspace) return 0;
  if (ldz < 9 * (int64_t)Cout || (ldz & 3)) return 0;
  if (!make_levels(nsizes, sizes, lv) || B > 65535) return 0;
  const size_t smem = sizeof(float4) * (size_t)3 * lv.rows * kFoldThreads;
  if (smem > 64 * 1024) return 0;
  FoldGradPtrs gp;
  for (int k = 0; k < kMaxLevels; ++k) gp.z[k] = k < nsizes ? gz[k] : nullptr;
  for (int k = 0; k < nsizes; ++k)
    if (!gp.z[k]) return 0;
  const int C4 = Cout / 4, chunks0 = fold_chunks(B, H, W, C4), ychunk = (int)cdiv(H, chunks0);
  const int chunks = (int)cdiv(H, ychunk);
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)cdiv((int64_t)B * W * C4, kFoldThreads), (unsigned)chunks);
  ppm_fold_bwd_cols_nhwc_kernel<<<grid, dim3(kFoldThreads), smem, st>>>(gout, workspace, B, H, W, C4, ychunk, lv);
  ppm_fold_bwd_bins_nhwc_kernel<<<dim3((unsigned)lv.bins * 9, (unsigned)B), dim3(kThreads), 0, st>>>(workspace, gp, B, W, C4, chunks, lv, ldz);
  return ok();
}

}  // extern "C"
