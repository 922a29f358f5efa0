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
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ppm_concat_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  ppm_concat_bwd_kernel<<<dim3((unsigned)(B * nsizes * Cout)), dim3(kThreads), smem, as_stream(stream)>>>(
      gcat, gp, B, Cout, Cfeat, H, W, lv);
  return ok();
}

}  // extern "C"
