// pixelwise.hip -- pixel-wise distillation loss (fwd + grad in one launch) for gfx950.
//
// Reference: CriterionPixelWise.forward, utils/criterion.py:219-226
//   loss = sum( -softmax(T, dim=class) * log_softmax(S, dim=class) ) / W / H      (NOT / N)
// The reference permutes both logit tensors to NHWC, makes them contiguous, and runs softmax,
// log-softmax, multiply, sum as separate ops (~8 launches, 6 intermediate tensors).  Here one
// lane owns one pixel: it reads the C class logits of S and T straight from NCHW (lanes are
// consecutive pixels -> every class plane is read with fully coalesced 256-B wave accesses),
// keeps them in registers, and emits both the pixel's loss term and dLoss/dS
// (= (softmax(S) - softmax(T)) / (W*H)).  HBM-bound: 3 * 4 * N*C*HW bytes.
#include "skd_common.hpp"

namespace skd {
namespace {

constexpr int kMaxRegClasses = 32;

template <int CMAX>
__global__ __launch_bounds__(kThreads) void pixelwise_kernel(const float *__restrict__ ls,
                                                            const float *__restrict__ lt,
                                                            float *__restrict__ grad,
                                                            float *__restrict__ part, int N, int C,
                                                            int HW, float inv_wh) {
  __shared__ float red[2 * kWavesPerWG];
  const int64_t total = (int64_t)N * HW;
  const int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  float term = 0.f, unused = 0.f;
  if (pix < total) {
    const int n = (int)(pix / HW);
    const int p = (int)(pix % HW);
    const int64_t base = (int64_t)n * C * HW + p;
    float s[CMAX], t[CMAX];
    float ms = -INFINITY, mt = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < C) {
        s[c] = ls[base + (int64_t)c * HW];
        t[c] = lt[base + (int64_t)c * HW];
        ms = fmaxf(ms, s[c]);
        mt = fmaxf(mt, t[c]);
      }
    }
    float zs = 0.f, zt = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < C) {
        s[c] -= ms;
        t[c] = expf(t[c] - mt);
        zs += expf(s[c]);
        zt += t[c];
      }
    }
    const float lse = logf(zs);
    const float inv_zt = 1.f / zt, inv_zs = 1.f / zs;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < C) {
        const float pt = t[c] * inv_zt;          // softmax(T)_c
        const float logps = s[c] - lse;           // log_softmax(S)_c
        term -= pt * logps;
        if (grad != nullptr) grad[base + (int64_t)c * HW] = (expf(s[c]) * inv_zs - pt) * inv_wh;
      }
    }
  }
  block_sum2(term, unused, red);
  if (threadIdx.x == 0) part[blockIdx.x] = term;
}

// any class count: three passes over the class planes (served by L1/L2 after the first).
__global__ __launch_bounds__(kThreads) void pixelwise_generic_kernel(const float *__restrict__ ls,
                                                                    const float *__restrict__ lt,
                                                                    float *__restrict__ grad,
                                                                    float *__restrict__ part, int N,
                                                                    int C, int HW, float inv_wh) {
  __shared__ float red[2 * kWavesPerWG];
  const int64_t total = (int64_t)N * HW;
  const int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  float term = 0.f, unused = 0.f;
  if (pix < total) {
    const int n = (int)(pix / HW);
    const int p = (int)(pix % HW);
    const int64_t base = (int64_t)n * C * HW + p;
    float ms = -INFINITY, mt = -INFINITY;
    for (int c = 0; c < C; ++c) {
      ms = fmaxf(ms, ls[base + (int64_t)c * HW]);
      mt = fmaxf(mt, lt[base + (int64_t)c * HW]);
    }
    float zs = 0.f, zt = 0.f;
    for (int c = 0; c < C; ++c) {
      zs += expf(ls[base + (int64_t)c * HW] - ms);
      zt += expf(lt[base + (int64_t)c * HW] - mt);
    }
    const float lse = logf(zs);
    for (int c = 0; c < C; ++c) {
      const float sc = ls[base + (int64_t)c * HW] - ms;
      const float pt = expf(lt[base + (int64_t)c * HW] - mt) / zt;
      term -= pt * (sc - lse);
      if (grad != nullptr) grad[base + (int64_t)c * HW] = (expf(sc) / zs - pt) * inv_wh;
    }
  }
  block_sum2(term, unused, red);
  if (threadIdx.x == 0) part[blockIdx.x] = term;
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

int64_t skd_pixelwise_workspace_floats(int N, int HW) {
  if (N <= 0 || HW <= 0) return 1;
  return cdiv((int64_t)N * HW, kThreads);
}

int skd_pixelwise_loss(int N, int C, int HW, const float *logits_s, const float *logits_t, float *loss,
                       float *grad_s, float *workspace, skd_stream_t stream) {
  if (N <= 0 || C <= 0 || HW <= 0 || !logits_s || !logits_t || !loss || !workspace) return 0;
  hipStream_t st = as_stream(stream);
  const int64_t wgs = cdiv((int64_t)N * HW, kThreads);
  // criterion.py:225 divides the sum by W and by H; the two factors only ever appear as W*H
  const float inv_wh = 1.f / (float)HW;
  if (C <= kMaxRegClasses)
    pixelwise_kernel<kMaxRegClasses><<<dim3((unsigned)wgs), dim3(kThreads), 0, st>>>(
        logits_s, logits_t, grad_s, workspace, N, C, HW, inv_wh);
  else
    pixelwise_generic_kernel<<<dim3((unsigned)wgs), dim3(kThreads), 0, st>>>(
        logits_s, logits_t, grad_s, workspace, N, C, HW, inv_wh);
  if (!ok()) return 0;
  return launch_final_sum(workspace, wgs, loss, 1.0 / (double)HW, st);
}

}  // extern "C"
