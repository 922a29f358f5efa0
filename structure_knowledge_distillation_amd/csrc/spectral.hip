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
__device__ __forceinline__ void sn_wt_u_body(const float *__restrict__ W, const float *__restrict__ u,
                                             float *__restrict__ tpart, int h, int w, int bx, int by) {
  const int j = bx * kThreads + threadIdx.x;
  const int rs = by;
  const int rows = (h + kRowSlices - 1) / kRowSlices;
  const int i0 = rs * rows, i1 = min(h, i0 + rows);
  float acc = 0.f;
  if (j < w) {
#pragma unroll 8
    for (int i = i0; i < i1; ++i) acc += W[(int64_t)i * w + j] * u[i];
    tpart[(int64_t)rs * w + j] = acc;
  }
}

__global__ __launch_bounds__(kThreads) void sn_wt_u_kernel(const float *__restrict__ W,
                                                          const float *__restrict__ u,
                                                          float *__restrict__ tpart, int h, int w) {
  sn_wt_u_body(W, u, tpart, h, w, blockIdx.x, blockIdx.y);
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
__device__ __forceinline__ void sn_w_v_body(const float *__restrict__ W, const float *__restrict__ tpart,
                                            float *__restrict__ v_out, float *__restrict__ s_out, int h, int w, int bx) {
  __shared__ float red[2 * kWavesPerWG];
  float ss = 0.f;
  for (int j = threadIdx.x; j < w; j += kThreads) {
    const float t = combined_t(tpart, w, j);
    ss += t * t;
  }
  const float tn = sqrtf(block_sumsq_bcast(ss, red));
  const float inv = 1.f / (tn + kNormEps);
  if (bx == 0)
    for (int j = threadIdx.x; j < w; j += kThreads) v_out[j] = combined_t(tpart, w, j) * inv;
  const int lane = threadIdx.x & (kWave - 1);
  const int i = bx * kWavesPerWG + threadIdx.x / kWave;
  if (i < h) {
    float acc = 0.f;
    for (int j = lane; j < w; j += kWave) acc += W[(int64_t)i * w + j] * (combined_t(tpart, w, j) * inv);
    acc = wave_sum(acc);
    if (lane == 0) s_out[i] = acc;
  }
}

__global__ __launch_bounds__(kThreads) void sn_w_v_kernel(const float *__restrict__ W,
                                                         const float *__restrict__ tpart,
                                                         float *__restrict__ v_out,
                                                         float *__restrict__ s_out, int h, int w) {
  sn_w_v_body(W, tpart, v_out, s_out, h, w, blockIdx.x);
}

// phase 3: u = s/(|s|+eps); sigma = u.s; w_out = W / sigma
__device__ __forceinline__ void sn_finish_body(const float *__restrict__ W, const float *__restrict__ s, float *__restrict__ u_out,
                                               float *__restrict__ sigma_out, float *__restrict__ w_out, int h, int w, int bx,
                                               int nblocks) {
  __shared__ float red[2 * kWavesPerWG];
  float ss = 0.f;
  for (int i = threadIdx.x; i < h; i += kThreads) ss += s[i] * s[i];
  const float s2 = block_sumsq_bcast(ss, red);
  const float inv = 1.f / (sqrtf(s2) + kNormEps);
  // sigma = sum_i u_i s_i with u_i = s_i * inv  (spectral.py:34)
  float dot = 0.f;
  for (int i = threadIdx.x; i < h; i += kThreads) dot += (s[i] * inv) * s[i];
  const float sigma = block_sumsq_bcast(dot, red);
  if (bx == 0) {
    for (int i = threadIdx.x; i < h; i += kThreads) u_out[i] = s[i] * inv;
    if (threadIdx.x == 0) sigma_out[0] = sigma;
  }
  if (w_out != nullptr) {
    const int64_t n = (int64_t)h * w;
    const int64_t stride = (int64_t)nblocks * kThreads;
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(w_out)) & 15) == 0) {
      const int64_t n4 = n >> 2;
      for (int64_t q = (int64_t)bx * kThreads + threadIdx.x; q < n4; q += stride) {
        float4 x = reinterpret_cast<const float4 *>(W)[q];
        x.x /= sigma;  // spectral.py:35  w / sigma
        x.y /= sigma;
        x.z /= sigma;
        x.w /= sigma;
        reinterpret_cast<float4 *>(w_out)[q] = x;
      }
    } else {
      for (int64_t q = (int64_t)bx * kThreads + threadIdx.x; q < n; q += stride) w_out[q] = W[q] / sigma;
    }
  }
}

__global__ __launch_bounds__(kThreads) void sn_finish_kernel(const float *__restrict__ W,
                                                            const float *__restrict__ s,
                                                            float *__restrict__ u_out,
                                                            float *__restrict__ sigma_out,
                                                            float *__restrict__ w_out, int h, int w) {
  sn_finish_body(W, s, u_out, sigma_out, w_out, h, w, blockIdx.x, gridDim.x);
}

// backward stage 1: partial sums of gW * W_bar
__device__ __forceinline__ void sn_bwd_dot_body(const float *__restrict__ gw, const float *__restrict__ wbar,
                                                float *__restrict__ part, int64_t n, int64_t per_wg, int bx) {
  __shared__ float red[2 * kWavesPerWG];
  const int64_t lo = (int64_t)bx * per_wg, hi = min(n, lo + per_wg);
  float acc = 0.f, unused = 0.f;
  for (int64_t q = lo + threadIdx.x; q < hi; q += kThreads) acc += gw[q] * wbar[q];
  block_sum2(acc, unused, red);
  if (threadIdx.x == 0) part[bx] = acc;
}

__global__ __launch_bounds__(kThreads) void sn_bwd_dot_kernel(const float *__restrict__ gw,
                                                             const float *__restrict__ wbar,
                                                             float *__restrict__ part, int64_t n,
                                                             int64_t per_wg) {
  sn_bwd_dot_body(gw, wbar, part, n, per_wg, blockIdx.x);
}

// backward stage 2: gW_bar[i][j] = gW[i][j]/sigma - (dot/sigma^2) u[i] v[j]
__device__ __forceinline__ void sn_bwd_apply_body(const float *__restrict__ gw, const float *__restrict__ u,
                                                  const float *__restrict__ v, const float *__restrict__ sigma,
                                                  const float *__restrict__ part, int nparts, float *__restrict__ gwbar, int h,
                                                  int w, int bx, int nblocks) {
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
  const int64_t stride = (int64_t)nblocks * kThreads;
  for (int64_t q = (int64_t)bx * kThreads + threadIdx.x; q < n; q += stride) {
    const int i = (int)(q / w), j = (int)(q - (int64_t)i * w);
    gwbar[q] = gw[q] / sg - coef * u[i] * v[j];
  }
}

__global__ __launch_bounds__(kThreads) void sn_bwd_apply_kernel(const float *__restrict__ gw,
                                                               const float *__restrict__ u,
                                                               const float *__restrict__ v,
                                                               const float *__restrict__ sigma,
                                                               const float *__restrict__ part,
                                                               int nparts, float *__restrict__ gwbar,
                                                               int h, int w) {
  sn_bwd_apply_body(gw, u, v, sigma, part, nparts, gwbar, h, w, blockIdx.x, gridDim.x);
}

constexpr int kBwdParts = 256;

// ---- several layers per launch (round 4) -------------------------------------------------------------------------
// The discriminator normalises four weights per forward (sagan_models.py:117-136) and runs four forwards per step: 48
// forward + 24 backward launches of 3-35 us on 78 KB ... 8 MB matrices, each waiting for the previous one's tail -- 1.16 ms
// per step at 2-3 % of the HBM roofline.  The layers are independent of each other, so the SAME three phases
// (two in backward) run for ALL layers in one launch each: a workgroup finds its layer in a small table passed by value.
// 72 launches per step become 18; every phase's grid is the sum of the layers' grids, so the small layers ride in the
// shadow of the 8 MB one instead of each paying its own launch + drain.  No grid barrier (an in-kernel barrier would have
// to be co-resident with the student's one-launch ABN passes on the other stream: DESIGN.md section 3).
constexpr int kSnMaxLayers = 8;
struct SnLayer {
  const float *W;       // w_bar (forward) / also w_bar in backward
  float *u, *v, *sigma, *w_out;
  float *tpart, *s;     // forward workspace of this layer
  const float *gw;      // backward: gradient w.r.t. the normalised weight
  float *gwbar, *part;  // backward: output, partial sums
  int64_t per_wg;       // backward stage 1
  int h, w, nparts, nb; // nb: this layer's workgroups in the current launch
};
struct SnBatch {
  SnLayer l[kSnMaxLayers];
  int first[kSnMaxLayers + 1];   // first[k] = first workgroup (blockIdx.x) of layer k in the current launch
  int L;
};
__device__ __forceinline__ int sn_layer_of(const SnBatch &b, int bx) {
  int k = 0;
  while (k + 1 < b.L && bx >= b.first[k + 1]) ++k;
  return k;
}
__global__ __launch_bounds__(kThreads) void sn_wt_u_multi_kernel(SnBatch b) {
  const int k = sn_layer_of(b, blockIdx.x);
  const SnLayer &l = b.l[k];
  sn_wt_u_body(l.W, l.u, l.tpart, l.h, l.w, blockIdx.x - b.first[k], blockIdx.y);
}
__global__ __launch_bounds__(kThreads) void sn_w_v_multi_kernel(SnBatch b) {
  const int k = sn_layer_of(b, blockIdx.x);
  const SnLayer &l = b.l[k];
  sn_w_v_body(l.W, l.tpart, l.v, l.s, l.h, l.w, blockIdx.x - b.first[k]);
}
__global__ __launch_bounds__(kThreads) void sn_finish_multi_kernel(SnBatch b) {
  const int k = sn_layer_of(b, blockIdx.x);
  const SnLayer &l = b.l[k];
  sn_finish_body(l.W, l.s, l.u, l.sigma, l.w_out, l.h, l.w, blockIdx.x - b.first[k], l.nb);
}
__global__ __launch_bounds__(kThreads) void sn_bwd_dot_multi_kernel(SnBatch b) {
  const int k = sn_layer_of(b, blockIdx.x);
  const SnLayer &l = b.l[k];
  sn_bwd_dot_body(l.gw, l.W, l.part, (int64_t)l.h * l.w, l.per_wg, blockIdx.x - b.first[k]);
}
__global__ __launch_bounds__(kThreads) void sn_bwd_apply_multi_kernel(SnBatch b) {
  const int k = sn_layer_of(b, blockIdx.x);
  const SnLayer &l = b.l[k];
  sn_bwd_apply_body(l.gw, l.u, l.v, l.sigma, l.part, l.nparts, l.gwbar, l.h, l.w, blockIdx.x - b.first[k], l.nb);
}

static int64_t sn_ws_floats(int h, int w) {
  const int64_t fwd = (int64_t)kRowSlices * w + h;
  return ((fwd > kBwdParts ? fwd : kBwdParts) + 3) / 4 * 4;      // keeps every layer's slice 16-byte aligned
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

int64_t skd_spectral_workspace_floats(int h, int w) {
  if (h <= 0 || w <= 0) return 1;
  return sn_ws_floats(h, w);
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


// ---- all layers of a network in one call: 3 launches forward, 2 backward, whatever L <= 8 is ----
// h / w: HOST arrays of L ints; the pointer arrays are HOST arrays of L DEVICE pointers.  workspace: sum over the layers of
// skd_spectral_workspace_floats(h_k, w_k) floats.  Same arithmetic per layer as the single-layer entries (bit-identical).
int skd_spectral_norm_forward_multi(int L, const int *h, const int *w, const float *const *w_bar, float *const *u, float *const *v,
                                    float *const *sigma, float *const *w_out, float *workspace, skd_stream_t stream) {
  if (L <= 0 || L > kSnMaxLayers || !h || !w || !w_bar || !u || !v || !sigma || !workspace) return 0;
  hipStream_t st = as_stream(stream);
  SnBatch b = {};
  b.L = L;
  float *ws = workspace;
  for (int k = 0; k < L; ++k) {
    if (h[k] <= 0 || w[k] <= 0 || !w_bar[k] || !u[k] || !v[k] || !sigma[k]) return 0;
    SnLayer &l = b.l[k];
    l.W = w_bar[k]; l.u = u[k]; l.v = v[k]; l.sigma = sigma[k]; l.w_out = w_out ? w_out[k] : nullptr;
    l.h = h[k]; l.w = w[k];
    l.tpart = ws;
    l.s = ws + (int64_t)kRowSlices * w[k];
    ws += sn_ws_floats(h[k], w[k]);
  }
  int tot = 0;
  for (int k = 0; k < L; ++k) { b.first[k] = tot; b.l[k].nb = (int)cdiv(b.l[k].w, kThreads); tot += b.l[k].nb; }
  b.first[L] = tot;
  sn_wt_u_multi_kernel<<<dim3((unsigned)tot, kRowSlices), dim3(kThreads), 0, st>>>(b);
  tot = 0;
  for (int k = 0; k < L; ++k) { b.first[k] = tot; b.l[k].nb = (int)cdiv(b.l[k].h, kWavesPerWG); tot += b.l[k].nb; }
  b.first[L] = tot;
  sn_w_v_multi_kernel<<<dim3((unsigned)tot), dim3(kThreads), 0, st>>>(b);
  tot = 0;
  for (int k = 0; k < L; ++k) {
    int64_t g = cdiv((int64_t)b.l[k].h * b.l[k].w, (int64_t)kThreads * 4 * 4);
    if (g < 1) g = 1;
    if (g > 1024) g = 1024;
    b.first[k] = tot; b.l[k].nb = (int)g; tot += (int)g;
  }
  b.first[L] = tot;
  sn_finish_multi_kernel<<<dim3((unsigned)tot), dim3(kThreads), 0, st>>>(b);
  return ok();
}

int skd_spectral_norm_backward_multi(int L, const int *h, const int *w, const float *const *w_bar, const float *const *u,
                                     const float *const *v, const float *const *sigma, const float *const *grad_w,
                                     float *const *grad_w_bar, float *workspace, skd_stream_t stream) {
  if (L <= 0 || L > kSnMaxLayers || !h || !w || !w_bar || !u || !v || !sigma || !grad_w || !grad_w_bar || !workspace) return 0;
  hipStream_t st = as_stream(stream);
  SnBatch b = {};
  b.L = L;
  float *ws = workspace;
  int tot = 0;
  for (int k = 0; k < L; ++k) {
    if (h[k] <= 0 || w[k] <= 0 || !w_bar[k] || !u[k] || !v[k] || !sigma[k] || !grad_w[k] || !grad_w_bar[k]) return 0;
    SnLayer &l = b.l[k];
    l.W = w_bar[k]; l.u = const_cast<float *>(u[k]); l.v = const_cast<float *>(v[k]); l.sigma = const_cast<float *>(sigma[k]);
    l.gw = grad_w[k]; l.gwbar = grad_w_bar[k]; l.h = h[k]; l.w = w[k];
    l.part = ws;
    ws += sn_ws_floats(h[k], w[k]);
    const int64_t n = (int64_t)h[k] * w[k];
    int64_t parts = cdiv(n, 4096);
    if (parts > kBwdParts) parts = kBwdParts;
    l.per_wg = cdiv(n, parts);
    parts = cdiv(n, l.per_wg);
    l.nparts = (int)parts;
    b.first[k] = tot; l.nb = (int)parts; tot += (int)parts;
  }
  b.first[L] = tot;
  sn_bwd_dot_multi_kernel<<<dim3((unsigned)tot), dim3(kThreads), 0, st>>>(b);
  tot = 0;
  for (int k = 0; k < L; ++k) {
    int64_t g = cdiv((int64_t)b.l[k].h * b.l[k].w, (int64_t)kThreads * 8);
    if (g < 1) g = 1;
    if (g > 1024) g = 1024;
    b.first[k] = tot; b.l[k].nb = (int)g; tot += (int)g;
  }
  b.first[L] = tot;
  sn_bwd_apply_multi_kernel<<<dim3((unsigned)tot), dim3(kThreads), 0, st>>>(b);
  return ok();
}

}  // extern "C"
