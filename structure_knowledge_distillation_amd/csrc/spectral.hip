// spectral.hip -- spectral-norm power iteration + weight rescale for gfx950.
//
// Reference: SpectralNorm._update_u_v, networks/spectral.py:23-35 (power_iterations = 1):
//     v <- l2normalize(W^T u)        l2normalize(x) = x / (||x|| + 1e-12)   (spectral.py:10-11)
//     u <- l2normalize(W v)
//     sigma = u . (W v) ;  weight = W / sigma
// The reference issues ~12 small launches per layer per forward (t(), mv, norm, div, dot,
// expand, div) and touches W four times.  Here three launches per layer:
//   1. t = W^T u        column-block x row-slice workgroups, coalesced 256-B row segments,
//                       partial t per row slice (deterministic, no atomics)
//   2. s = W (t/|t|)    every workgroup re-derives |t| from the (L2-resident) partials, one
//                       wave per row; workgroup 0 also publishes v
//   3. u, sigma, W/sigma  every workgroup re-derives |s| (<= a few KiB), grid-stride float4
//                       rescale; workgroup 0 publishes u and sigma
// A kernel boundary (~1.5 us) is cheaper on this chip than an in-kernel grid barrier (4-7 us),
// so the three phases stay separate launches.  HBM/L2-bound: W is read three times and
// written once (the last two reads hit L2 / Infinity Cache: largest W is 8 MiB).
// Backward (u, v are constants, sigma = u^T W v):
//     gW_bar = gW / sigma - (sum(gW * W_bar) / sigma^2) * u v^T
#include "skd_common.hpp"

namespace skd {
namespace {

constexpr float kNormEps = 1e-12f;  // spectral.py:10
constexpr int kRowSlices = 8;       // row slices in phase 1

// phase 1: tpart[rs][j] = sum_{i in slice rs} W[i][j] * u[i]
// grid (ceil(w/256), kRowSlices); one thread per column, rows of the slice streamed.
__global__ __launch_bounds__(kThreads) void sn_wt_u_kernel(const float *__restrict__ W,
                                                          const float *__restrict__ u,
                                                          float *__restrict__ tpart, int h, int w) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const int rs = blockIdx.y;
  const int rows = (h + kRowSlices - 1) / kRowSlices;
  const int i0 = rs * rows, i1 = min(h, i0 + rows);
  float acc = 0.f;
  if (j < w) {
#pragma unroll 8
    for (int i = i0; i < i1; ++i) acc += W[(int64_t)i * w + j] * u[i];
    tpart[(int64_t)rs * w + j] = acc;
  }
}

// ||sum_rs tpart||^2 over all columns, computed redundantly by each workgroup; also leaves the
// combined, un-normalised t in LDS-free form by recomputation at use.
__device__ __forceinline__ float combined_t(const float *__restrict__ tpart, int w, int j) {
  float t = 0.f;
#pragma unroll
  for (int rs = 0; rs < kRowSlices; ++rs) t += tpart[(int64_t)rs * w + j];
  return t;
}

__device__ __forceinline__ float block_sumsq_bcast(float part, float *scratch) {
  float other = 0.f;
  block_sum2(part, other, scratch);
  __shared__ float bc;
  if (threadIdx.x == 0) bc = part;
  __syncthreads();
  return bc;
}

// phase 2: s[i] = sum_j W[i][j] * v[j], v = t / (|t| + eps); one wave per row.
__global__ __launch_bounds__(kThreads) void sn_w_v_kernel(const float *__restrict__ W,
                                                         const float *__restrict__ tpart,
                                                         float *__restrict__ v_out,
                                                         float *__restrict__ s_out, int h, int w) {
  __shared__ float red[2 * kWavesPerWG];
  float ss = 0.f;
  for (int j = threadIdx.x; j < w; j += kThreads) {
    const float t = combined_t(tpart, w, j);
    ss += t * t;
  }
  const float tn = sqrtf(block_sumsq_bcast(ss, red));
  const float inv = 1.f / (tn + kNormEps);
  if (blockIdx.x == 0)
    for (int j = threadIdx.x; j < w; j += kThreads) v_out[j] = combined_t(tpart, w, j) * inv;
  const int lane = threadIdx.x & (kWave - 1);
  const int i = blockIdx.x * kWavesPerWG + threadIdx.x / kWave;
  if (i < h) {
    float acc = 0.f;
    for (int j = lane; j < w; j += kWave) acc += W[(int64_t)i * w + j] * (combined_t(tpart, w, j) * inv);
    acc = wave_sum(acc);
    if (lane == 0) s_out[i] = acc;
  }
}

// phase 3: u = s/(|s|+eps); sigma = u.s; w_out = W / sigma
__global__ __launch_bounds__(kThreads) void sn_finish_kernel(const float *__restrict__ W,
                                                            const float *__restrict__ s,
                                                            float *__restrict__ u_out,
                                                            float *__restrict__ sigma_out,
                                                            float *__restrict__ w_out, int h, int w) {
  __shared__ float red[2 * kWavesPerWG];
  float ss = 0.f;
  for (int i = threadIdx.x; i < h; i += kThreads) ss += s[i] * s[i];
  const float s2 = block_sumsq_bcast(ss, red);
  const float inv = 1.f / (sqrtf(s2) + kNormEps);
  // sigma = sum_i u_i s_i with u_i = s_i * inv  (spectral.py:34)
  float dot = 0.f;
  for (int i = threadIdx.x; i < h; i += kThreads) dot += (s[i] * inv) * s[i];
  const float sigma = block_sumsq_bcast(dot, red);
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < h; i += kThreads) u_out[i] = s[i] * inv;
    if (threadIdx.x == 0) sigma_out[0] = sigma;
  }
  if (w_out != nullptr) {
    const int64_t n = (int64_t)h * w;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(w_out)) & 15) == 0) {
      const int64_t n4 = n >> 2;
      for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < n4; q += stride) {
        float4 x = reinterpret_cast<const float4 *>(W)[q];
        x.x /= sigma;  // spectral.py:35  w / sigma
        x.y /= sigma;
        x.z /= sigma;
        x.w /= sigma;
        reinterpret_cast<float4 *>(w_out)[q] = x;
      }
    } else {
      for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < n; q += stride) w_out[q] = W[q] / sigma;
    }
  }
}

// backward stage 1: partial sums of gW * W_bar
__global__ __launch_bounds__(kThreads) void sn_bwd_dot_kernel(const float *__restrict__ gw,
                                                             const float *__restrict__ wbar,
                                                             float *__restrict__ part, int64_t n,
                                                             int64_t per_wg) {
  __shared__ float red[2 * kWavesPerWG];
  const int64_t lo = (int64_t)blockIdx.x * per_wg, hi = min(n, lo + per_wg);
  float acc = 0.f, unused = 0.f;
  for (int64_t q = lo + threadIdx.x; q < hi; q += kThreads) acc += gw[q] * wbar[q];
  block_sum2(acc, unused, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// backward stage 2: gW_bar[i][j] = gW[i][j]/sigma - (dot/sigma^2) u[i] v[j]
__global__ __launch_bounds__(kThreads) void sn_bwd_apply_kernel(const float *__restrict__ gw,
                                                               const float *__restrict__ u,
                                                               const float *__restrict__ v,
                                                               const float *__restrict__ sigma,
                                                               const float *__restrict__ part,
                                                               int nparts, float *__restrict__ gwbar,
                                                               int h, int w) {
  __shared__ double dred[kWavesPerWG];
  __shared__ float coef_s;
  double d = 0.0;
  for (int q = threadIdx.x; q < nparts; q += kThreads) d += (double)part[q];
  d = wave_sum(d);
  if ((threadIdx.x & (kWave - 1)) == 0) dred[threadIdx.x / kWave] = d;
  __syncthreads();
  const float sg = sigma[0];
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kWavesPerWG; ++k) t += dred[k];
    coef_s = (float)(t / ((double)sg * (double)sg));
  }
  __syncthreads();
  const float coef = coef_s;
  const int64_t n = (int64_t)h * w;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < n; q += stride) {
    const int i = (int)(q / w), j = (int)(q - (int64_t)i * w);
    gwbar[q] = gw[q] / sg - coef * u[i] * v[j];
  }
}

constexpr int kBwdParts = 256;

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

int64_t skd_spectral_workspace_floats(int h, int w) {
  if (h <= 0 || w <= 0) return 1;
  const int64_t fwd = (int64_t)kRowSlices * w + h;
  return fwd > kBwdParts ? fwd : kBwdParts;
}

int skd_spectral_norm_forward(int h, int w, const float *w_bar, float *u, float *v, float *sigma,
                              float *w_out, float *workspace, skd_stream_t stream) {
  if (h <= 0 || w <= 0 || !w_bar || !u || !v || !sigma || !workspace) return 0;
  hipStream_t st = as_stream(stream);
  float *tpart = workspace;
  float *s = workspace + (int64_t)kRowSlices * w;
  sn_wt_u_kernel<<<dim3((unsigned)cdiv(w, kThreads), kRowSlices), dim3(kThreads), 0, st>>>(w_bar, u, tpart, h, w);
  sn_w_v_kernel<<<dim3((unsigned)cdiv(h, kWavesPerWG)), dim3(kThreads), 0, st>>>(w_bar, tpart, v, s, h, w);
  int64_t g = cdiv((int64_t)h * w, (int64_t)kThreads * 4 * 4);
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  sn_finish_kernel<<<dim3((unsigned)g), dim3(kThreads), 0, st>>>(w_bar, s, u, sigma, w_out, h, w);
  return ok();
}

int skd_spectral_norm_backward(int h, int w, const float *w_bar, const float *u, const float *v,
                               const float *sigma, const float *grad_w, float *grad_w_bar,
                               float *workspace, skd_stream_t stream) {
  if (h <= 0 || w <= 0 || !w_bar || !u || !v || !sigma || !grad_w || !grad_w_bar || !workspace) return 0;
  hipStream_t st = as_stream(stream);
  const int64_t n = (int64_t)h * w;
  int64_t parts = cdiv(n, 4096);
  if (parts > kBwdParts) parts = kBwdParts;
  const int64_t per = cdiv(n, parts);
  parts = cdiv(n, per);
  sn_bwd_dot_kernel<<<dim3((unsigned)parts), dim3(kThreads), 0, st>>>(grad_w, w_bar, workspace, n, per);
  int64_t g = cdiv(n, (int64_t)kThreads * 8);
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  sn_bwd_apply_kernel<<<dim3((unsigned)g), dim3(kThreads), 0, st>>>(grad_w, u, v, sigma, workspace, (int)parts,
                                                                   grad_w_bar, h, w);
  return ok();
}

}  // extern "C"
