// maxpool.hip -- the stem's MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=True) of both networks
// (networks/pspnet_combine.py:135, 152: 128 channels, 256 x 256 -> 129 x 129 at a 512 x 512 crop) for channels-last tensors.
//
// The stock channels-last pooling kernels write an int64 flat index per output element (134 MB next to a 67 MB result
// at batch 8) and scatter the gradient back with one thread per OUTPUT element; measured 235 us forward / 210 us
// backward per call (profiles/r02c_bench_kernel_stats.csv).  Here a thread owns one (pixel, channel quad):
//   forward   nine 16-byte loads, the running maximum under PyTorch's rule (`val > max || isnan(val)`, windows scanned
//             row-major from the first valid position, so ties keep the FIRST maximum and NaN propagates), and the
//             winner's position inside its window as one byte (0..8) -- none at all for the frozen teacher;
//   backward  gather form over 2 x 2 blocks of INPUT positions: the four windows that can contain them are loaded once and every
//             position adds the gradients of the ones it won -- no atomics, every element of dx written exactly once, deterministic.
// HBM-bound: forward reads x once (neighbouring windows hit L2) and writes 4 + 1 bytes per output element; backward
// reads 4 + 1 bytes per output element and writes dx once.
#include "skd_common.hpp"

namespace skd {
namespace {

__device__ __forceinline__ void take(float v, int code, float &best, int &arg) {
  if (v > best || v != v) {
    best = v;
    arg = code;
  }
}

template <bool ARG>
__global__ __launch_bounds__(kThreads) void maxpool3x3s2_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                    uint8_t *__restrict__ arg, int64_t items, int H,
                                                                    int W, int OH, int OW, int C4) {
  const int64_t item = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (item >= items) return;
  const int q = (int)(item % C4);
  const int ox = (int)((item / C4) % OW);
  const int oy = (int)((item / ((int64_t)C4 * OW)) % OH);
  const int b = (int)(item / ((int64_t)C4 * OW * OH));
  const int ys = 2 * oy - 1, xs = 2 * ox - 1;
  const int y0 = ys < 0 ? 0 : ys, x0 = xs < 0 ? 0 : xs;
  const int y1 = ys + 3 < H ? ys + 3 : H, x1 = xs + 3 < W ? xs + 3 : W;
  const float ninf = -__builtin_huge_valf();
  float4 best = make_float4(ninf, ninf, ninf, ninf);
  const int first = (y0 - ys) * 3 + (x0 - xs);
  int a0 = first, a1 = first, a2 = first, a3 = first;
  const float *src = x + ((int64_t)b * H * W) * C4 * 4 + q * 4;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      const float4 v = *reinterpret_cast<const float4 *>(src + ((int64_t)yy * W + xx) * C4 * 4);
      const int code = (yy - ys) * 3 + (xx - xs);
      take(v.x, code, best.x, a0);
      take(v.y, code, best.y, a1);
      take(v.z, code, best.z, a2);
      take(v.w, code, best.w, a3);
    }
  *reinterpret_cast<float4 *>(y + item * 4) = best;
  if (ARG) *reinterpret_cast<uchar4 *>(arg + item * 4) = make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
}

// Backward, gather form over 2 x 2 INPUT blocks (round 5): the four positions (2i, 2j) .. (2i+1, 2j+1) only ever belong to the four
// windows (i, j), (i, j+1), (i+1, j), (i+1, j+1) -- with the codes below -- so a lane loads those four (gradient quad + argmax
// bytes) once and writes four 16-byte results, instead of one lane per position re-reading 1 / 2 / 2 / 4 windows (9 loads of
// 20 bytes for 64 bytes written; measured 2.3 TB/s).  Same terms in the same order per position
// (windows in (oy, ox) order): bit-identical to the one-position form.
//   position      window (i, j)   (i, j+1)   (i+1, j)   (i+1, j+1)
//   (2i,   2j)        4
//   (2i,   2j+1)      5              3
//   (2i+1, 2j)        7                         1
//   (2i+1, 2j+1)      8              6          2           0
__global__ __launch_bounds__(kThreads) void maxpool3x3s2_bwd_nhwc_kernel(const float *__restrict__ dy,
                                                                        const uint8_t *__restrict__ arg,
                                                                        float *__restrict__ dx, int64_t items, int H, int W,
                                                                        int OH, int OW, int C4, int H2, int W2) {
  const int64_t item = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (item >= items) return;
  const int q = (int)(item % C4);
  const int j = (int)((item / C4) % W2);
  const int i = (int)((item / ((int64_t)C4 * W2)) % H2);
  const int b = (int)(item / ((int64_t)C4 * W2 * H2));
  float4 g[4];
  uchar4 a[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int oy = i + (k >> 1), ox = j + (k & 1);
    if (oy < OH && ox < OW) {
      const int64_t o = ((((int64_t)b * OH + oy) * OW + ox) * C4 + q) * 4;
      a[k] = *reinterpret_cast<const uchar4 *>(arg + o);
      g[k] = *reinterpret_cast<const float4 *>(dy + o);
    } else {
      a[k] = make_uchar4(255, 255, 255, 255);          // no such window: matches no code
      g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  auto pick = [](const uchar4 &aa, const float4 &gg, int code, float4 &acc) {
    if (aa.x == code) acc.x += gg.x;
    if (aa.y == code) acc.y += gg.y;
    if (aa.z == code) acc.z += gg.z;
    if (aa.w == code) acc.w += gg.w;
  };
  const int yy = 2 * i, xx = 2 * j;
  float *out = dx + ((((int64_t)b * H + yy) * W + xx) * C4 + q) * 4;
  const int64_t dxs = (int64_t)C4 * 4, dys = (int64_t)W * C4 * 4;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  pick(a[0], g[0], 4, r);
  *reinterpret_cast<float4 *>(out) = r;
  if (xx + 1 < W) {
    r = make_float4(0.f, 0.f, 0.f, 0.f);
    pick(a[0], g[0], 5, r);
    pick(a[1], g[1], 3, r);
    *reinterpret_cast<float4 *>(out + dxs) = r;
  }
  if (yy + 1 < H) {
    r = make_float4(0.f, 0.f, 0.f, 0.f);
    pick(a[0], g[0], 7, r);
    pick(a[2], g[2], 1, r);
    *reinterpret_cast<float4 *>(out + dys) = r;
    if (xx + 1 < W) {
      r = make_float4(0.f, 0.f, 0.f, 0.f);
      pick(a[0], g[0], 8, r);
      pick(a[1], g[1], 6, r);
      pick(a[2], g[2], 2, r);
      pick(a[3], g[3], 0, r);
      *reinterpret_cast<float4 *>(out + dys + dxs) = r;
    }
  }
}

bool pool_geom_ok(int B, int C, int H, int W, int OH, int OW) {
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return false;
  // every window starts inside the input (ceil_mode rule) and the windows cover it
  return 2 * (OH - 1) - 1 < H && 2 * (OW - 1) - 1 < W && 2 * (OH - 1) + 1 >= H - 1 && 2 * (OW - 1) + 1 >= W - 1;
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

int skd_maxpool3x3s2_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, float *y, uint8_t *arg,
                          skd_stream_t stream) {
  if (!pool_geom_ok(B, C, H, W, OH, OW) || !x || !y) return 0;
  const int64_t items = (int64_t)B * OH * OW * (C / 4);
  const dim3 grid((unsigned)cdiv(items, kThreads));
  if (arg)
    maxpool3x3s2_nhwc_kernel<true><<<grid, dim3(kThreads), 0, as_stream(stream)>>>(x, y, arg, items, H, W, OH, OW, C / 4);
  else
    maxpool3x3s2_nhwc_kernel<false><<<grid, dim3(kThreads), 0, as_stream(stream)>>>(x, y, nullptr, items, H, W, OH, OW, C / 4);
  return ok();
}

int skd_maxpool3x3s2_backward_nhwc(int B, int C, int H, int W, int OH, int OW, const float *dy, const uint8_t *arg, float *dx,
                                   skd_stream_t stream) {
  if (!pool_geom_ok(B, C, H, W, OH, OW) || !dy || !arg || !dx) return 0;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;          // 2 x 2 input blocks
  const int64_t items = (int64_t)B * H2 * W2 * (C / 4);
  maxpool3x3s2_bwd_nhwc_kernel<<<dim3((unsigned)cdiv(items, kThreads)), dim3(kThreads), 0, as_stream(stream)>>>(
      dy, arg, dx, items, H, W, OH, OW, C / 4, H2, W2);
  return ok();
}

}  // extern "C"
