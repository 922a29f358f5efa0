// evaluate.hip -- whole-image evaluation tail for gfx950: bilinear upsample (align_corners) of the class logits to
// the label size, argmax over classes, confusion-matrix accumulation -- fused, bit-exact integer outputs.
//
// Reference: networks/evaluate.py:106-113 (predict_whole: net(image)[0] -> nn.Upsample(size, bilinear,
// align_corners=True)), :186 (np.argmax over the class axis -> uint8), :193-198 (pixels with label != 255 only),
// :136-154 (get_confusion_matrix: bincount of gt * class_num + pred), :200-206 (IoU = tp / max(1, pos + res - tp)).
// The reference materialises the (1, C, 1024, 2048) up-sampled logits on the GPU (159 MB at 19 classes), copies them
// to the host and does argmax / bincount in numpy.  Here one lane per output pixel rebuilds its C up-sampled logits
// from the 4 neighbouring source pixels (the logit map is L2 resident), takes the first maximum (numpy's argmax
// rule), writes the uint8 prediction and bumps a per-workgroup LDS histogram that is flushed with integer atomics:
// only the label (8 B/pixel) and the prediction (1 B/pixel) touch HBM.
// Floating-point contraction is OFF in this file: the four-term interpolation is evaluated with individually rounded
// mul/add exactly like the plain-C oracle, so predictions and counts match it bit for bit.
#include "skd_common.hpp"

#pragma clang fp contract(off)

namespace skd {
namespace {

constexpr int kMaxEvalClasses = 64;

__global__ __launch_bounds__(kThreads) void seg_confusion_kernel(const float *__restrict__ logits,
                                                                const int64_t *__restrict__ target,
                                                                unsigned char *__restrict__ pred,
                                                                unsigned long long *__restrict__ confusion, int B,
                                                                int C, int h, int w, int H, int W, int ignore_index,
                                                                float sy, float sx) {
  extern __shared__ unsigned int hist[];  // C * C
  for (int i = threadIdx.x; i < C * C; i += kThreads) hist[i] = 0u;
  __syncthreads();
  const int64_t total = (int64_t)B * H * W;
  const int hw = h * w;
  for (int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x; pix < total; pix += (int64_t)gridDim.x * kThreads) {
    const int X = (int)(pix % W);
    const int Y = (int)((pix / W) % H);
    const int b = (int)(pix / ((int64_t)W * H));
    // upsample_bilinear2d, align_corners=True: src = scale * dst; i0 = (int)src; i1 = i0 + (i0 < in - 1)
    const float fy = sy * (float)Y, fx = sx * (float)X;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > h - 1) y0 = h - 1;
    if (x0 > w - 1) x0 = w - 1;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float *p = logits + (int64_t)b * C * hw;
    float best = 0.f;
    int arg = 0;
    for (int c = 0; c < C; ++c) {
      const float *q = p + (int64_t)c * hw;
      const float v = ly0 * (lx0 * q[y0 * w + x0] + lx1 * q[y0 * w + x1]) + ly1 * (lx0 * q[y1 * w + x0] + lx1 * q[y1 * w + x1]);
      if (c == 0 || v > best) {  // first maximum wins (numpy argmax); a NaN never replaces the running maximum
        best = v;
        arg = c;
      }
    }
    if (pred != nullptr) pred[pix] = (unsigned char)arg;
    if (target != nullptr) {
      const int64_t t = target[pix];
      if (t != (int64_t)ignore_index && t >= 0 && t < C) atomicAdd(&hist[(int)t * C + arg], 1u);
    }
  }
  __syncthreads();
  if (confusion != nullptr)
    for (int i = threadIdx.x; i < C * C; i += kThreads)
      if (hist[i] != 0u) atomicAdd(&confusion[i], (unsigned long long)hist[i]);
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

int skd_seg_confusion(int B, int C, int h, int w, int H, int W, const float *logits, const int64_t *target,
                      int ignore_index, uint8_t *pred, int64_t *confusion, skd_stream_t stream) {
  if (B <= 0 || C <= 0 || C > kMaxEvalClasses || h <= 0 || w <= 0 || H <= 0 || W <= 0 || !logits) return 0;
  if (target != nullptr && confusion == nullptr) return 0;
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int64_t total = (int64_t)B * H * W;
  int64_t wgs = cdiv(total, (int64_t)kThreads * 8);  // ~8 pixels per lane: one histogram flush per 2048 pixels
  if (wgs < 1) wgs = 1;
  if (wgs > 8192) wgs = 8192;
  seg_confusion_kernel<<<dim3((unsigned)wgs), dim3(kThreads), sizeof(unsigned int) * C * C, as_stream(stream)>>>(
      logits, target, pred, reinterpret_cast<unsigned long long *>(confusion), B, C, h, w, H, W, ignore_index, sy, sx);
  return ok();
}

}  // extern "C"
