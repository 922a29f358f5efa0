// abn.hip -- InPlace-ABN (fused batch-norm + activation) for gfx950.
//
// Replaces the reference's only native component, libs/src/bn.cu (K1 mean_var 125-138,
// K2 forward 140-165, K3 edz_eydz 167-184, K4 backward 186-232, K5-K9 activations 302-377)
// and the op ordering of libs/functions.py:70-162.  Not a translation: the reference launches
// C workgroups (one per channel, each looping over all N*S elements, two passes for the
// variance, separate thrust passes for the activation).  Here:
//   * work is cut into contiguous runs of <= kChunk floats of ONE (n, c) row, so a launch has
//     N*C*pieces workgroups (4096 for an (8,512,65,65) tensor) that each stream 16-32 KiB with
//     16-byte loads -- the kernels are HBM-bound and sized to cover all 256 CUs several times;
//   * statistics are ONE pass: sums of (x-K) and (x-K)^2 around a per-channel pivot K = median of three samples of
//     the channel (shifted-data variance; no catastrophic cancellation, no second read of x);
//   * per-workgroup partials are combined by a tiny finalize kernel in double precision in a
//     fixed order (deterministic, no atomics), which also performs the running-stat update;
//   * normalise + affine(|w|+eps) + activation is one in-place pass; backward undoes the
//     activation in registers (z is never rewritten) and fuses it into both backward passes.
// Algorithmic bytes/element (fp32): train fwd 12 (R,R,W), train bwd 20 (R z,dz; R z,dz, W dx),
// eval fwd 8 -- versus 16-24 / 20-40 for the reference launch sequence.
#include <stdlib.h>

#include <atomic>

#include "skd_common.hpp"
#include "sync_dev.hpp"

namespace skd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16-byte global access with an optional non-temporal (streaming) cache policy.  Measured HBM-cold on MI355X
// (tools/abn_microbench.py cold): nt loads AND nt stores together lift the in-place apply pass from 4.6-5.4 to
// 4.9-6.0 TB/s when every tensor comes from HBM (PyTorch's own copy_/relu_ reach 5.2-5.9); either alone does
// nothing.  INSIDE the training step, however, the convolution output is still partly in the Infinity Cache
// when the apply pass reads it, and the same policy LOWERS the pass from 4.70 to 4.22 TB/s (bench.py A/B) --
// so the policy is "normal" (apply_nt_mode below).
__device__ __forceinline__ float4 ldg4(const float *p, bool nt) {
  if (nt) {
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
  }
  return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ void stg4(float *p, float4 v, bool nt) {
  if (nt) {
    f32x4 t;
    t[0] = v.x;
    t[1] = v.y;
    t[2] = v.z;
    t[3] = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(p));
  } else {
    *reinterpret_cast<float4 *>(p) = v;
  }
}

constexpr int kChunk = 8192;  // max floats per workgroup run (32 KiB)

// How the (N, C, S) tensor is cut into workgroup items.
//   S large : every (n, c) row is split into `pieces` runs of `chunk` floats, rows_per_item = 1
//   S small : one item covers `rows_per_item` consecutive n of the same channel
// P = items per channel = number of partial slots per channel.
struct Plan {
  int pieces, chunk, rows_per_item, groups, P;
  int64_t items;
};

static Plan make_plan(int N, int C, int S) {
  Plan p;
  if (2 * S >= kChunk) {
    p.pieces = (int)cdiv(S, kChunk);
    p.chunk = (int)((cdiv(S, p.pieces) + 3) & ~(int64_t)3);
    p.pieces = (int)cdiv(S, p.chunk);
    p.rows_per_item = 1;
    p.groups = N;
  } else {
    p.pieces = 1;
    p.chunk = S;
    int r = kChunk / (S > 0 ? S : 1);
    if (r < 1) r = 1;
    if (r > N) r = N;
    p.rows_per_item = r;
    p.groups = (int)cdiv(N, r);
  }
  p.P = p.groups * p.pieces;
  p.items = (int64_t)p.P * C;
  return p;
}

struct Item {
  int c, p, n0, n1, start, len;
};

// item index -> (channel, partial slot, row range, run inside the row).  Items are numbered in
// memory order for rows_per_item == 1 so that neighbouring workgroups touch neighbouring DRAM.
__device__ __forceinline__ Item decode(int64_t w, int N, int C, int S, const Plan &pl) {
  Item it;
  if (pl.rows_per_item == 1) {
    const int piece = (int)(w % pl.pieces);
    const int64_t row = w / pl.pieces;
    const int n = (int)(row / C);
    it.c = (int)(row % C);
    it.n0 = n;
    it.n1 = n + 1;
    it.start = piece * pl.chunk;
    it.len = min(pl.chunk, S - it.start);
    it.p = n * pl.pieces + piece;
  } else {
    it.c = (int)(w % C);
    const int g = (int)(w / C);
    it.n0 = g * pl.rows_per_item;
    it.n1 = min(N, it.n0 + pl.rows_per_item);
    it.start = 0;
    it.len = S;
    it.p = g;
  }
  return it;
}

__device__ __forceinline__ float gamma_of(const float *weight, int c, float eps) {
  return weight != nullptr ? fabsf(weight[c]) + eps : 1.f;  // bn.cu:153
}
__device__ __forceinline__ float beta_of(const float *bias, int c) {
  return bias != nullptr ? bias[c] : 0.f;  // bn.cu:154
}
__device__ __forceinline__ float inv_std_of(float var, float eps) {
  return (var != 0.f || eps != 0.f) ? 1.f / sqrtf(var + eps) : 0.f;  // bn.cu:148-151
}

// functions.py:91,209: running_var takes var * n / (n - 1).  With ONE sample per channel (the PSP 1x1 stage at
// batch 1 on a single replica) the reference divides by zero and poisons the buffer with
// NaN / inf; here n == 1 keeps the (zero) biased variance instead -- the one deliberate deviation, see DESIGN.md.
__device__ __forceinline__ float unbiased_of(float var, float n) { return n > 1.f ? var * n / (n - 1.f) : var; }

// Pivot of the one-pass (shifted) statistics: the MEDIAN of three samples of the channel -- first, middle and last element
// of the tensor's channel.  The shifted variance loses ~k^2 * 2^-24 of relative accuracy when the pivot sits k sigma from
// the mean (bn.cu:125-138 is two-pass and has no such term); a single sample as pivot makes that k the tail of the data
// (one outlier element 100 sigma off: 6e-4), the median of three needs TWO outliers among the three probes.  NaN-free
// ordering: fminf / fmaxf return the non-NaN operand, and a NaN anywhere in the channel poisons the sums regardless.
__device__ __forceinline__ float median3(float a, float b, float c) {
  return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}
__device__ __forceinline__ float pivot_nchw(const float *x, int c, int N, int C, int S) {
  const float a = x[(int64_t)c * S];
  const float b = x[((int64_t)(N / 2) * C + c) * S + S / 2];
  const float d = x[((int64_t)(N - 1) * C + c) * S + (S - 1)];
  return median3(a, b, d);
}

template <int ACT>
__device__ __forceinline__ float act_fwd(float z, float slope) {
  if (ACT == SKD_ACT_LEAKY_RELU) return z < 0.f ? z * slope : z;        // bn.cu:302-315
  if (ACT == SKD_ACT_ELU) return z < 0.f ? expf(z) - 1.f : z;           // bn.cu:333-346
  if (ACT == SKD_ACT_RELU) return z < 0.f ? 0.f : z;                    // nn.ReLU after BatchNorm2d, pspnet_combine.py:36,68,72
  return z;
}
// undo the activation on (z, dz) in registers: functions.py:54-62 / bn.cu:317-331,348-377
template <int ACT>
__device__ __forceinline__ void act_undo(float &z, float &dz, float slope, float inv_slope) {
  if (ACT == SKD_ACT_LEAKY_RELU) {
    if (z < 0.f) {
      dz *= slope;
      z *= inv_slope;
    }
  } else if (ACT == SKD_ACT_ELU) {
    if (z < 0.f) {
      dz *= (z + 1.f);
      z = log1pf(z);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1': shifted one-pass statistics.  part[(c*P + p)*2 + {0,1}] = sum(x-K), sum((x-K)^2)
// ---------------------------------------------------------------------------------------------
struct StatsOp {
  const float *row;
  float K, s1, s2;
  __device__ __forceinline__ float ld1(int i) const { return row[i]; }
  __device__ __forceinline__ void use1(int, float v) {
    const float d = v - K;
    s1 += d;
    s2 += d * d;
  }
  __device__ __forceinline__ float4 ld4(int i) const {
    return *reinterpret_cast<const float4 *>(row + i);
  }
  __device__ __forceinline__ void use4(int, float4 v) {
    const float d0 = v.x - K, d1 = v.y - K, d2 = v.z - K, d3 = v.w - K;
    s1 += (d0 + d1) + (d2 + d3);
    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
};

__global__ __launch_bounds__(kThreads) void abn_stats_partial_kernel(const float *__restrict__ x,
                                                                    float *__restrict__ part, int N,
                                                                    int C, int S, Plan pl) {
  __shared__ float red[2 * kWavesPerWG];
  const Item it = decode(blockIdx.x, N, C, S, pl);
  StatsOp op;
  op.K = pivot_nchw(x, it.c, N, C, S);  // per-channel pivot (three uniform loads)
  op.s1 = 0.f;
  op.s2 = 0.f;
  for (int n = it.n0; n < it.n1; ++n) {
    op.row = x + ((int64_t)n * C + it.c) * S + it.start;
    stream_run(reinterpret_cast<uintptr_t>(op.row), it.len, op);
  }
  float a = op.s1, b = op.s2;
  block_sum2(a, b, red);
  if (threadIdx.x == 0) {
    float *dst = part + ((int64_t)it.c * pl.P + it.p) * 2;
    dst[0] = a;
    dst[1] = b;
  }
}

// One wave per channel: combine P partials in double, emit mean / biased var, optionally update
// the running statistics (functions.py:90-91: running_var uses var*n/(n-1), n = count*replicas).
__global__ __launch_bounds__(kThreads) void abn_stats_finalize_kernel(
    const float *__restrict__ x, const float *__restrict__ part, float *__restrict__ mean,
    float *__restrict__ var, float *running_mean, float *running_var, int N, int C, int S, int P,
    float momentum, double n_total) {
  const int lane = threadIdx.x & (kWave - 1);
  const int c = blockIdx.x * kWavesPerWG + threadIdx.x / kWave;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int p = lane; p < P; p += kWave) {
    s1 += (double)part[((int64_t)c * P + p) * 2];
    s2 += (double)part[((int64_t)c * P + p) * 2 + 1];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    const double cnt = (double)N * (double)S;
    const double K = (double)pivot_nchw(x, c, N, C, S);
    const double d = s1 / cnt;
    double v = s2 / cnt - d * d;
    if (v < 0.0) v = 0.0;
    const float m_f = (float)(K + d), v_f = (float)v;
    mean[c] = m_f;
    var[c] = v_f;
    if (running_mean != nullptr) running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * m_f;
    if (running_var != nullptr) {
      const float nf = (float)n_total;
      running_var[c] = running_var[c] * (1.f - momentum) + momentum * unbiased_of(v_f, nf);
    }
  }
}

// Cross-replica combine (libs/functions.py:196-197, 208-209) in one launch: gathered is (G, 2, C) = per-rank
// [mean, var]; mean = means.mean(0); var = (vars + (mean - means)^2).mean(0); running stats with n = count * G.
// The reference rule assumes every replica saw the same number of samples.  `weights` (G floats summing to one,
// w_g = n_g / sum n, may be NULL) generalises it to unequal shards -- the exact pooled statistics; with equal shards
// it is the reference rule.  With weights, `nf` is THIS rank's count and the pooled count is nf / weights[rank].
__global__ void abn_combine_stats_kernel(int G, int C, const float *__restrict__ gathered,
                                         const float *__restrict__ weights, int rank, float *__restrict__ mean,
                                         float *__restrict__ var, float *running_mean, float *running_var,
                                         float momentum, float nf) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float m, v;
  combine_channel(G, C, c, [&](int g, int j) { return gathered[(int64_t)g * 2 * C + j]; }, weights, rank, nf, momentum, m, v,
                  running_mean, running_var);
  mean[c] = m;
  var[c] = v;
}

__global__ void abn_update_running_kernel(int C, float *running_mean, float *running_var,
                                          const float *mean, const float *var, float momentum,
                                          float nf) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * mean[c];
  running_var[c] = running_var[c] * (1.f - momentum) + momentum * unbiased_of(var[c], nf);
}

struct F4x2 {
  float4 a, b;
};
struct F1x2 {
  float a, b;
};

// ---------------------------------------------------------------------------------------------
// K2': normalise + affine + activation.  Writes z (and y when y != z, the legacy two-output form).
// ---------------------------------------------------------------------------------------------
template <int ACT, bool WRITE_Y>
struct ApplyOp {
  const float *xin;
  float *yout, *zout;
  float mean, inv_std, gamma, beta, slope;
  int nt;  // bit 0: non-temporal loads, bit 1: non-temporal stores (streaming data is touched exactly once)
  __device__ __forceinline__ float one(float v, float &y) const {
    y = (v - mean) * inv_std;  // bn.cu:158
    return act_fwd<ACT>(y * gamma + beta, slope);  // bn.cu:159 (+ fused K5/K7)
  }
  __device__ __forceinline__ float ld1(int i) const { return xin[i]; }
  __device__ __forceinline__ void use1(int i, float v) const {
    float y;
    const float z = one(v, y);
    if (WRITE_Y) yout[i] = y;
    zout[i] = z;
  }
  __device__ __forceinline__ float4 ld4(int i) const {
    return ldg4(xin + i, nt & 1);
  }
  __device__ __forceinline__ void use4(int i, float4 v) const {
    float4 y, z;
    z.x = one(v.x, y.x);
    z.y = one(v.y, y.y);
    z.z = one(v.z, y.z);
    z.w = one(v.w, y.w);
    if (WRITE_Y) *reinterpret_cast<float4 *>(yout + i) = y;
    stg4(zout + i, z, nt & 2);
  }
};

// z = act(bn(x) + r): the tail of a residual block (pspnet_combine.py:41-43 / 80-82) in one pass.
template <int ACT>
struct ApplyResOp {
  const float *xin, *rin;
  float *zout;
  float mean, inv_std, gamma, beta, slope;
  int nt;
  __device__ __forceinline__ float one(float v, float r) const {
    return act_fwd<ACT>(((v - mean) * inv_std) * gamma + beta + r, slope);
  }
  __device__ __forceinline__ F1x2 ld1(int i) const { return F1x2{xin[i], rin[i]}; }
  __device__ __forceinline__ void use1(int i, F1x2 v) const { zout[i] = one(v.a, v.b); }
  __device__ __forceinline__ F4x2 ld4(int i) const {
    return F4x2{ldg4(xin + i, nt & 1), ldg4(rin + i, nt & 1)};
  }
  __device__ __forceinline__ void use4(int i, F4x2 v) const {
    float4 z;
    z.x = one(v.a.x, v.b.x);
    z.y = one(v.a.y, v.b.y);
    z.z = one(v.a.z, v.b.z);
    z.w = one(v.a.w, v.b.w);
    stg4(zout + i, z, nt & 2);
  }
};

template <int ACT, bool WRITE_Y>
__global__ __launch_bounds__(kThreads) void abn_apply_kernel(
    const float *x, const float *__restrict__ mean, const float *__restrict__ var,
    const float *__restrict__ weight, const float *__restrict__ bias, float *y, float *z, float eps,
    float slope, int N, int C, int S, Plan pl, int reverse, int nt) {
  // `reverse`: walk the items backwards so that a pass that follows the statistics pass starts
  // on the lines that pass touched last (still resident in L2 / Infinity Cache).
  const int64_t w = reverse ? (pl.items - 1 - (int64_t)blockIdx.x) : (int64_t)blockIdx.x;
  const Item it = decode(w, N, C, S, pl);
  ApplyOp<ACT, WRITE_Y> op;
  op.nt = nt;
  op.mean = mean[it.c];
  op.inv_std = inv_std_of(var[it.c], eps);
  op.gamma = gamma_of(weight, it.c, eps);
  op.beta = beta_of(bias, it.c);
  op.slope = slope;
  for (int n = it.n0; n < it.n1; ++n) {
    const int64_t off = ((int64_t)n * C + it.c) * S + it.start;
    op.xin = x + off;
    op.yout = y + off;
    op.zout = z + off;
    stream_run(reinterpret_cast<uintptr_t>(op.xin), it.len, op);
  }
}

template <int ACT>
__global__ __launch_bounds__(kThreads) void abn_apply_residual_kernel(
    const float *x, const float *res, const float *__restrict__ mean, const float *__restrict__ var,
    const float *__restrict__ weight, const float *__restrict__ bias, float *z, float eps, float slope,
    int N, int C, int S, Plan pl, int nt) {
  const Item it = decode(blockIdx.x, N, C, S, pl);
  ApplyResOp<ACT> op;
  op.nt = nt;
  op.mean = mean[it.c];
  op.inv_std = inv_std_of(var[it.c], eps);
  op.gamma = gamma_of(weight, it.c, eps);
  op.beta = beta_of(bias, it.c);
  op.slope = slope;
  for (int n = it.n0; n < it.n1; ++n) {
    const int64_t off = ((int64_t)n * C + it.c) * S + it.start;
    op.xin = x + off;
    op.rin = res + off;
    op.zout = z + off;
    stream_run(reinterpret_cast<uintptr_t>(op.xin), it.len, op);
  }
}


// ---------------------------------------------------------------------------------------------
// Inference BN -> (+ residual) -> activation for channels-last (NHWC) tensors: x is (rows = N*H*W, C) row-major,
// the channel is the fastest dimension, so every thread keeps ONE channel quad's parameters in registers and
// streams rows with 16-byte accesses.  Used by the frozen teacher, whose convolutions run NHWC-native in MIOpen
// (no NCHW<->NHWC transposes around the igemm kernels).  C must be a multiple of 4.
// ---------------------------------------------------------------------------------------------
// Fast form for power-of-two channel counts (every layer of this path): a workgroup owns a contiguous run of
// 256 * 8 quads (32 KiB); quad q of thread t sits at base + u*256 + t, so its channel quad is (t + u*256) mod C4 --
// constant for C4 <= 256, cycling through NSETS = C4/256 register sets above that.  Eight 16-byte loads per lane.
template <int ACT, bool HAS_RES, int NSETS>
__global__ __launch_bounds__(kThreads) void abn_apply_nhwc_kernel(float *x, const float *res,
                                                                 const float *__restrict__ mean,
                                                                 const float *__restrict__ var,
                                                                 const float *__restrict__ weight,
                                                                 const float *__restrict__ bias, float eps,
                                                                 float slope, int64_t quads, int C4) {
  constexpr int U = 8;
  const int64_t base = (int64_t)blockIdx.x * (kThreads * U);
  float m[NSETS][4], is[NSETS][4], g[NSETS][4], b[NSETS][4];
#pragma unroll
  for (int s = 0; s < NSETS; ++s) {
    const int c = ((threadIdx.x + s * kThreads) & (C4 - 1)) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      m[s][k] = mean[c + k];
      is[s][k] = inv_std_of(var[c + k], eps);
      g[s][k] = gamma_of(weight, c + k, eps);
      b[s][k] = beta_of(bias, c + k);
    }
  }
  float4 v[U], r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t q = base + u * kThreads + threadIdx.x;
    if (q < quads) {
      v[u] = *reinterpret_cast<const float4 *>(x + 4 * q);
      if (HAS_RES) r[u] = *reinterpret_cast<const float4 *>(res + 4 * q);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t q = base + u * kThreads + threadIdx.x;
    if (q < quads) {
      constexpr int dummy = 0;
      const int s = NSETS == 1 ? dummy : (u % NSETS);
      float4 z;
      z.x = act_fwd<ACT>(((v[u].x - m[s][0]) * is[s][0]) * g[s][0] + b[s][0] + (HAS_RES ? r[u].x : 0.f), slope);
      z.y = act_fwd<ACT>(((v[u].y - m[s][1]) * is[s][1]) * g[s][1] + b[s][1] + (HAS_RES ? r[u].y : 0.f), slope);
      z.z = act_fwd<ACT>(((v[u].z - m[s][2]) * is[s][2]) * g[s][2] + b[s][2] + (HAS_RES ? r[u].z : 0.f), slope);
      z.w = act_fwd<ACT>(((v[u].w - m[s][3]) * is[s][3]) * g[s][3] + b[s][3] + (HAS_RES ? r[u].w : 0.f), slope);
      *reinterpret_cast<float4 *>(x + 4 * q) = z;
    }
  }
}

// generic channel counts (any C % 4 == 0): grid-stride, total threads a multiple of C4
template <int ACT, bool HAS_RES>
__global__ __launch_bounds__(kThreads) void abn_apply_nhwc_generic_kernel(float *x, const float *res,
                                                                 const float *__restrict__ mean,
                                                                 const float *__restrict__ var,
                                                                 const float *__restrict__ weight,
                                                                 const float *__restrict__ bias, float eps,
                                                                 float slope, int64_t quads, int C4) {
  // total threads is a multiple of C4, so a thread's channel quad never changes while it strides
  const int64_t T = (int64_t)gridDim.x * kThreads;
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int c = (int)(t % C4) * 4;
  float m[4], is[4], g[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = mean[c + k];
    is[k] = inv_std_of(var[c + k], eps);
    g[k] = gamma_of(weight, c + k, eps);
    b[k] = beta_of(bias, c + k);
  }
  for (int64_t q = t; q < quads; q += 4 * T) {
    float4 v[4], r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t qq = q + u * T;
      if (qq < quads) {
        v[u] = *reinterpret_cast<const float4 *>(x + 4 * qq);
        if (HAS_RES) r[u] = *reinterpret_cast<const float4 *>(res + 4 * qq);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t qq = q + u * T;
      if (qq < quads) {
        float4 z;
        z.x = act_fwd<ACT>(((v[u].x - m[0]) * is[0]) * g[0] + b[0] + (HAS_RES ? r[u].x : 0.f), slope);
        z.y = act_fwd<ACT>(((v[u].y - m[1]) * is[1]) * g[1] + b[1] + (HAS_RES ? r[u].y : 0.f), slope);
        z.z = act_fwd<ACT>(((v[u].z - m[2]) * is[2]) * g[2] + b[2] + (HAS_RES ? r[u].z : 0.f), slope);
        z.w = act_fwd<ACT>(((v[u].w - m[3]) * is[3]) * g[3] + b[3] + (HAS_RES ? r[u].w : 0.f), slope);
        *reinterpret_cast<float4 *>(x + 4 * qq) = z;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K3': edz / eydz partial sums with the activation undone in registers.
// ---------------------------------------------------------------------------------------------
template <int ACT>
struct GradReduceOp {
  const float *z, *dz;
  float beta, inv_gamma_unused, gamma, slope, inv_slope, s1, s2;
  __device__ __forceinline__ void acc(float zv, float dzv) {
    act_undo<ACT>(zv, dzv, slope, inv_slope);
    const float y = (zv - beta) / gamma;  // bn.cu:52
    s1 += dzv;
    s2 += y * dzv;
  }
  __device__ __forceinline__ F1x2 ld1(int i) const { return F1x2{z[i], dz[i]}; }
  __device__ __forceinline__ void use1(int, F1x2 v) { acc(v.a, v.b); }
  __device__ __forceinline__ F4x2 ld4(int i) const {
    return F4x2{*reinterpret_cast<const float4 *>(z + i), *reinterpret_cast<const float4 *>(dz + i)};
  }
  __device__ __forceinline__ void use4(int, F4x2 v) {
    acc(v.a.x, v.b.x);
    acc(v.a.y, v.b.y);
    acc(v.a.z, v.b.z);
    acc(v.a.w, v.b.w);
  }
};

template <int ACT>
__global__ __launch_bounds__(kThreads) void abn_grad_partial_kernel(
    const float *__restrict__ z, const float *__restrict__ dz, const float *__restrict__ weight,
    const float *__restrict__ bias, float *__restrict__ part, float eps, float slope, int N, int C,
    int S, Plan pl) {
  __shared__ float red[2 * kWavesPerWG];
  const Item it = decode(blockIdx.x, N, C, S, pl);
  GradReduceOp<ACT> op;
  op.gamma = gamma_of(weight, it.c, eps);
  op.beta = beta_of(bias, it.c);
  op.slope = slope;
  op.inv_slope = 1.f / slope;
  op.s1 = 0.f;
  op.s2 = 0.f;
  for (int n = it.n0; n < it.n1; ++n) {
    const int64_t off = ((int64_t)n * C + it.c) * S + it.start;
    op.z = z + off;
    op.dz = dz + off;
    stream_run(reinterpret_cast<uintptr_t>(op.z), it.len, op);
  }
  float a = op.s1, b = op.s2;
  block_sum2(a, b, red);
  if (threadIdx.x == 0) {
    float *dst = part + ((int64_t)it.c * pl.P + it.p) * 2;
    dst[0] = a;
    dst[1] = b;
  }
}

__global__ __launch_bounds__(kThreads) void abn_grad_finalize_kernel(const float *__restrict__ part,
                                                                    float *__restrict__ edz,
                                                                    float *__restrict__ eydz, int N,
                                                                    int C, int S, int P) {
  const int lane = threadIdx.x & (kWave - 1);
  const int c = blockIdx.x * kWavesPerWG + threadIdx.x / kWave;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int p = lane; p < P; p += kWave) {
    s1 += (double)part[((int64_t)c * P + p) * 2];
    s2 += (double)part[((int64_t)c * P + p) * 2 + 1];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    const double cnt = (double)N * (double)S;
    edz[c] = (float)(s1 / cnt);   // bn.cu:176
    eydz[c] = (float)(s2 / cnt);  // bn.cu:177
  }
}

// ---------------------------------------------------------------------------------------------
// K4': dx = (dz - edz - y*eydz) * gamma * invStd, with the activation undone in registers.
//      dweight += sign(w)*eydz*N*S, dbias += edz*N*S by the first item of each channel.
// ---------------------------------------------------------------------------------------------
template <int ACT>
struct GradDxOp {
  const float *z, *dz;
  float *dx;
  float beta, gamma, slope, inv_slope, edz, eydz, mul;
  int nt;
  __device__ __forceinline__ float one(float zv, float dzv) const {
    act_undo<ACT>(zv, dzv, slope, inv_slope);
    const float y = (zv - beta) / gamma;   // bn.cu:208
    return (dzv - edz - y * eydz) * mul;   // bn.cu:209
  }
  __device__ __forceinline__ F1x2 ld1(int i) const { return F1x2{z[i], dz[i]}; }
  __device__ __forceinline__ void use1(int i, F1x2 v) const { dx[i] = one(v.a, v.b); }
  __device__ __forceinline__ F4x2 ld4(int i) const {
    // z is the saved forward output (the next layer's backward may still want it): normal load; dz is dead after this
    return F4x2{*reinterpret_cast<const float4 *>(z + i), ldg4(dz + i, nt & 1)};
  }
  __device__ __forceinline__ void use4(int i, F4x2 v) const {
    float4 r;
    r.x = one(v.a.x, v.b.x);
    r.y = one(v.a.y, v.b.y);
    r.z = one(v.a.z, v.b.z);
    r.w = one(v.a.w, v.b.w);
    stg4(dx + i, r, nt & 2);
  }
};

template <int ACT>
__global__ __launch_bounds__(kThreads) void abn_grad_dx_kernel(
    const float *z, const float *dz, const float *__restrict__ var,
    const float *__restrict__ weight, const float *__restrict__ bias,
    const float *__restrict__ edz, const float *__restrict__ eydz, float *dx, float *dweight,
    float *dbias, float eps, float slope, int N, int C, int S, Plan pl, int reverse, int apply_nt) {
  const int64_t w = reverse ? (pl.items - 1 - (int64_t)blockIdx.x) : (int64_t)blockIdx.x;
  const Item it = decode(w, N, C, S, pl);
  GradDxOp<ACT> op;
  op.nt = apply_nt;
  op.gamma = gamma_of(weight, it.c, eps);
  op.beta = beta_of(bias, it.c);
  op.slope = slope;
  op.inv_slope = 1.f / slope;
  op.edz = edz[it.c];
  op.eydz = eydz[it.c];
  op.mul = op.gamma * inv_std_of(var[it.c], eps);  // bn.cu:203
  if (dx != nullptr) {
    for (int n = it.n0; n < it.n1; ++n) {
      const int64_t off = ((int64_t)n * C + it.c) * S + it.start;
      op.z = z + off;
      op.dz = dz + off;
      op.dx = dx + off;
      stream_run(reinterpret_cast<uintptr_t>(op.z), it.len, op);
    }
  }
  if (it.p == 0 && threadIdx.x == 0) {
    const float norm = (float)N * (float)S;  // bn.cu:215
    if (dweight != nullptr) {
      const float wv = weight[it.c];
      if (wv > 0.f)
        dweight[it.c] += op.eydz * norm;  // bn.cu:219-222
      else if (wv < 0.f)
        dweight[it.c] -= op.eydz * norm;
    }
    if (dbias != nullptr) dbias[it.c] += op.edz * norm;  // bn.cu:228
  }
}


// ---------------------------------------------------------------------------------------------
// Training-time BN -> (+ residual) -> ReLU in one op, OUT OF PLACE: the convolution output x is kept
// (backward recomputes y = (x - mean) * invStd from it), `out` is what the next layer reads; the ReLU
// mask is `out > 0`.  Replaces InPlace-ABN(activation='none') + nn.ReLU (+ `out + residual`) of
// networks/pspnet_combine.py:36-43, 68-82 -- same tensors kept alive (reference: z and relu(z)), but
// 12 B/element forward instead of 20 (32 with the residual add) and no separate ReLU backward pass.
// ---------------------------------------------------------------------------------------------
struct F4x3 {
  float4 a, b, c;
};
struct F1x3 {
  float a, b, c;
};

struct ReluGradReduceOp {
  const float *x, *out, *dout;
  float mean, inv_std, s1, s2;
  __device__ __forceinline__ void acc(float xv, float ov, float dv) {
    const float dz = ov > 0.f ? dv : 0.f;
    const float y = (xv - mean) * inv_std;
    s1 += dz;
    s2 += y * dz;
  }
  __device__ __forceinline__ F1x3 ld1(int i) const { return F1x3{x[i], out[i], dout[i]}; }
  __device__ __forceinline__ void use1(int, F1x3 v) { acc(v.a, v.b, v.c); }
  __device__ __forceinline__ F4x3 ld4(int i) const {
    return F4x3{*reinterpret_cast<const float4 *>(x + i), *reinterpret_cast<const float4 *>(out + i),
                *reinterpret_cast<const float4 *>(dout + i)};
  }
  __device__ __forceinline__ void use4(int, F4x3 v) {
    acc(v.a.x, v.b.x, v.c.x);
    acc(v.a.y, v.b.y, v.c.y);
    acc(v.a.z, v.b.z, v.c.z);
    acc(v.a.w, v.b.w, v.c.w);
  }
};

__global__ __launch_bounds__(kThreads) void abn_relu_grad_partial_kernel(
    const float *__restrict__ x, const float *__restrict__ out, const float *__restrict__ dout,
    const float *__restrict__ mean, const float *__restrict__ var, float *__restrict__ part, float eps, int N,
    int C, int S, Plan pl) {
  __shared__ float red[2 * kWavesPerWG];
  const Item it = decode(blockIdx.x, N, C, S, pl);
  ReluGradReduceOp op;
  op.mean = mean[it.c];
  op.inv_std = inv_std_of(var[it.c], eps);
  op.s1 = 0.f;
  op.s2 = 0.f;
  for (int n = it.n0; n < it.n1; ++n) {
    const int64_t off = ((int64_t)n * C + it.c) * S + it.start;
    op.x = x + off;
    op.out = out + off;
    op.dout = dout + off;
    stream_run(reinterpret_cast<uintptr_t>(op.x), it.len, op);
  }
  float a = op.s1, b = op.s2;
  block_sum2(a, b, red);
  if (threadIdx.x == 0) {
    float *dst = part + ((int64_t)it.c * pl.P + it.p) * 2;
    dst[0] = a;
    dst[1] = b;
  }
}

template <bool WRITE_RES>
struct ReluGradDxOp {
  const float *x, *out, *dout;
  float *dx, *dres;
  float mean, inv_std, edz, eydz, mul;
  int nt;
  __device__ __forceinline__ float one(float xv, float ov, float dv, float &dz) const {
    dz = ov > 0.f ? dv : 0.f;
    const float y = (xv - mean) * inv_std;
    return (dz - edz - y * eydz) * mul;  // bn.cu:209
  }
  __device__ __forceinline__ F1x3 ld1(int i) const { return F1x3{x[i], out[i], dout[i]}; }
  __device__ __forceinline__ void use1(int i, F1x3 v) const {
    float dz;
    dx[i] = one(v.a, v.b, v.c, dz);
    if (WRITE_RES) dres[i] = dz;
  }
  __device__ __forceinline__ F4x3 ld4(int i) const {
    // x, out and dout are all read for the last time here
    return F4x3{ldg4(x + i, nt & 1), ldg4(out + i, nt & 1), ldg4(dout + i, nt & 1)};
  }
  __device__ __forceinline__ void use4(int i, F4x3 v) const {
    float4 r, d;
    r.x = one(v.a.x, v.b.x, v.c.x, d.x);
    r.y = one(v.a.y, v.b.y, v.c.y, d.y);
    r.z = one(v.a.z, v.b.z, v.c.z, d.z);
    r.w = one(v.a.w, v.b.w, v.c.w, d.w);
    stg4(dx + i, r, nt & 2);
    if (WRITE_RES) stg4(dres + i, d, nt & 2);
  }
};

template <bool WRITE_RES>
__global__ __launch_bounds__(kThreads) void abn_relu_grad_dx_kernel(
    const float *x, const float *out, const float *dout, const float *__restrict__ mean,
    const float *__restrict__ var, const float *__restrict__ weight, const float *__restrict__ edz,
    const float *__restrict__ eydz, float *dx, float *dres, float *dweight, float *dbias, float eps, int N, int C,
    int S, Plan pl, int apply_nt) {
  const int64_t w = pl.items - 1 - (int64_t)blockIdx.x;  // start on the lines the reduce pass touched last
  const Item it = decode(w, N, C, S, pl);
  ReluGradDxOp<WRITE_RES> op;
  op.nt = apply_nt;
  op.mean = mean[it.c];
  op.inv_std = inv_std_of(var[it.c], eps);
  op.edz = edz[it.c];
  op.eydz = eydz[it.c];
  op.mul = gamma_of(weight, it.c, eps) * op.inv_std;
  for (int n = it.n0; n < it.n1; ++n) {
    const int64_t off = ((int64_t)n * C + it.c) * S + it.start;
    op.x = x + off;
    op.out = out + off;
    op.dout = dout + off;
    op.dx = dx + off;
    op.dres = WRITE_RES ? dres + off : nullptr;
    stream_run(reinterpret_cast<uintptr_t>(op.x), it.len, op);
  }
  if (it.p == 0 && threadIdx.x == 0) {
    const float norm = (float)N * (float)S;
    if (dweight != nullptr) {
      const float wv = weight[it.c];
      if (wv > 0.f)
        dweight[it.c] += op.eydz * norm;
      else if (wv < 0.f)
        dweight[it.c] -= op.eydz * norm;
    }
    if (dbias != nullptr) dbias[it.c] += op.edz * norm;
  }
}

// ---------------------------------------------------------------------------------------------
// K5-K9: stand-alone activations (legacy ABI only; the fused path never launches them).
// ---------------------------------------------------------------------------------------------
template <int KIND>  // 0 leaky fwd, 1 leaky bwd, 2 elu fwd, 3 elu bwd, 4 elu inv
__global__ __launch_bounds__(kThreads) void act_kernel(int64_t n, const float *x, float *out,
                                                       float slope) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float xv = x[i];
    if (xv < 0.f) {
      if (KIND == 0) out[i] = xv * slope;
      if (KIND == 1) out[i] = out[i] * slope;
      if (KIND == 2) out[i] = expf(xv) - 1.f;
      if (KIND == 3) out[i] = out[i] * (xv + 1.f);
      if (KIND == 4) out[i] = log1pf(xv);
    }
  }
}

static int act_grid(int64_t n) {
  int64_t g = cdiv(n, kThreads);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// One grow-only scratch buffer per device for the legacy entries, whose reference signatures have
// no workspace argument.  (The fused entries take the workspace from the caller.)
struct Scratch {
  float *ptr = nullptr;
  int64_t floats = 0;
};
static Scratch g_scratch[64];

static float *legacy_scratch(int64_t floats) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  Scratch &s = g_scratch[dev];
  if (s.floats < floats) {
    if (s.ptr != nullptr) {
      (void)hipDeviceSynchronize();
      (void)hipFree(s.ptr);
    }
    s.ptr = nullptr;
    s.floats = 0;
    int64_t want = floats < (1 << 16) ? (1 << 16) : floats;
    if (hipMalloc(reinterpret_cast<void **>(&s.ptr), want * sizeof(float)) != hipSuccess) return nullptr;
    s.floats = want;
  }
  return s.ptr;
}

static bool same_phase(const void *a, const void *b) {
  return ((reinterpret_cast<uintptr_t>(a) ^ reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

// Cache policy of the planar apply / dx passes (bit 0 non-temporal loads, bit 1 non-temporal stores): normal.  The streaming
// policy was an experiment switch until round 5; its verdict (+11 % on HBM-cold tensors, -10 % inside the step where the
// convolution output is still in the Infinity Cache: profiles/r02*) is recorded and the switch is gone.
static constexpr int apply_nt_mode() { return 0; }

template <bool WRITE_Y>
static void launch_apply(int act, const Plan &pl, hipStream_t st, const float *x, const float *mean,
                         const float *var, const float *weight, const float *bias, float *y, float *z,
                         float eps, float slope, int N, int C, int S, int reverse) {
  const dim3 grid((unsigned)pl.items), block(kThreads);
  switch (act) {
    case SKD_ACT_LEAKY_RELU:
      abn_apply_kernel<SKD_ACT_LEAKY_RELU, WRITE_Y><<<grid, block, 0, st>>>(x, mean, var, weight, bias, y, z, eps, slope, N, C, S, pl, reverse, apply_nt_mode());
      break;
    case SKD_ACT_ELU:
      abn_apply_kernel<SKD_ACT_ELU, WRITE_Y><<<grid, block, 0, st>>>(x, mean, var, weight, bias, y, z, eps, slope, N, C, S, pl, reverse, apply_nt_mode());
      break;
    case SKD_ACT_RELU:
      abn_apply_kernel<SKD_ACT_RELU, WRITE_Y><<<grid, block, 0, st>>>(x, mean, var, weight, bias, y, z, eps, slope, N, C, S, pl, reverse, apply_nt_mode());
      break;
    default:
      abn_apply_kernel<SKD_ACT_NONE, WRITE_Y><<<grid, block, 0, st>>>(x, mean, var, weight, bias, y, z, eps, slope, N, C, S, pl, reverse, apply_nt_mode());
  }
}

template <int ACT, bool HAS_RES>
static void launch_apply_nhwc_act(int64_t quads, int C4, float *x, const float *res, const float *mean,
                                  const float *var, const float *weight, const float *bias, float eps, float slope,
                                  hipStream_t st) {
  const dim3 block(kThreads);
  const bool pow2 = (C4 & (C4 - 1)) == 0;
  if (pow2 && C4 <= 4 * kThreads) {
    const dim3 grid((unsigned)cdiv(quads, (int64_t)kThreads * 8));
    if (C4 <= kThreads)
      abn_apply_nhwc_kernel<ACT, HAS_RES, 1><<<grid, block, 0, st>>>(x, res, mean, var, weight, bias, eps, slope, quads, C4);
    else if (C4 == 2 * kThreads)
      abn_apply_nhwc_kernel<ACT, HAS_RES, 2><<<grid, block, 0, st>>>(x, res, mean, var, weight, bias, eps, slope, quads, C4);
    else
      abn_apply_nhwc_kernel<ACT, HAS_RES, 4><<<grid, block, 0, st>>>(x, res, mean, var, weight, bias, eps, slope, quads, C4);
    return;
  }
  // threads = a multiple of lcm(C4, 256), about quads / 4 (four 16-byte accesses in flight per thread)
  int64_t unit = C4;
  while (unit % kThreads != 0) unit *= 2;
  int64_t threads = cdiv(cdiv(quads, 4), unit) * unit;
  const int64_t cap = cdiv((int64_t)256 * 8 * kThreads, unit) * unit;   // ~8 workgroups per CU, then stride
  if (threads > cap) threads = cap;
  if (threads < unit) threads = unit;
  abn_apply_nhwc_generic_kernel<ACT, HAS_RES><<<dim3((unsigned)(threads / kThreads)), block, 0, st>>>(
      x, res, mean, var, weight, bias, eps, slope, quads, C4);
}

template <bool HAS_RES>
static int launch_apply_nhwc(int64_t rows, int C, float *x, const float *res, const float *mean, const float *var,
                             const float *weight, const float *bias, float eps, int act, float slope, hipStream_t st) {
  const int C4 = C / 4;
  const int64_t quads = rows * C4;
  switch (act) {
    case SKD_ACT_NONE:
      launch_apply_nhwc_act<SKD_ACT_NONE, HAS_RES>(quads, C4, x, res, mean, var, weight, bias, eps, slope, st);
      break;
    case SKD_ACT_LEAKY_RELU:
      launch_apply_nhwc_act<SKD_ACT_LEAKY_RELU, HAS_RES>(quads, C4, x, res, mean, var, weight, bias, eps, slope, st);
      break;
    case SKD_ACT_RELU:
      launch_apply_nhwc_act<SKD_ACT_RELU, HAS_RES>(quads, C4, x, res, mean, var, weight, bias, eps, slope, st);
      break;
    default:
      return 0;
  }
  return ok();
}


// =============================================================================================
// Channels-last (NHWC) TRAINING kernels: x is (rows = N*H*W, C) row-major.
// MIOpen's fastest fp32 kernels on gfx950 are NHWC implicit-GEMM kernels; handing it NCHW tensors costs a
// transpose before and after each of them (4.5 ms per step for the student's forward/backward, profiles/).
// With the channel as the fastest dimension a thread owns ONE channel quad (its statistics accumulate in
// registers, its parameters live in registers) and a workgroup walks a contiguous slab of rows:
//   256 threads = (256 / C4) rows x C4 channel quads per pass, 8 passes in flight per loop trip.
// The two apply-type passes (normalise, dx) below give every 32 KiB slab its own 256-thread workgroup; the reductions
// (statistics, edz / eydz) are the one-launch kernels of the "second design" section further down.
// Power-of-two C with 4 <= C <= 1024 (every training layer of this path: 64 ... 512).
// =============================================================================================
constexpr int kNhwcRowsPerThread = 8;

struct NhwcGeom {
  int C4, log2C4, rpp;  // channel quads, log2, rows per pass (256 / C4)
  int rows_per_wg;      // rpp * kNhwcRowsPerThread
  int P;                // workgroups = partial slots per channel
};

static bool make_nhwc_geom(int64_t rows, int C, NhwcGeom &g) {
  if (rows <= 0 || C < 4 || C > 4 * kThreads || (C & (C - 1))) return false;
  g.C4 = C / 4;
  g.log2C4 = 0;
  while ((1 << g.log2C4) < g.C4) ++g.log2C4;
  g.rpp = kThreads / g.C4;
  g.rows_per_wg = g.rpp * kNhwcRowsPerThread;
  const int64_t P = cdiv(rows, g.rows_per_wg);
  if (P > (1 << 24)) return false;
  g.P = (int)P;
  return true;
}

// The pre-activation of the fused BN (+ residual) + ReLU, as ONE expression shared by the forward pass and by the backward
// passes that recompute the ReLU mask from x instead of reading `out` (MODE 2): same instructions, same bits, same sign.
__device__ __forceinline__ float bn_pre(float x, float m, float is, float gm, float b) {
  return __builtin_fmaf((x - m) * is, gm, b);
}

// K2 (NHWC), out of place or in place: out = act(bn(x) [+ residual]) with given mean / var
template <int ACT, bool HAS_RES>
__global__ __launch_bounds__(kThreads) void abn_apply_nhwc_train_kernel(const float *x, const float *res, float *out,
                                                                       const float *__restrict__ mean,
                                                                       const float *__restrict__ var,
                                                                       const float *__restrict__ weight,
                                                                       const float *__restrict__ bias, float eps,
                                                                       float slope, int64_t rows, NhwcGeom g) {
  const int t = threadIdx.x;
  const int cq = t & (g.C4 - 1), rsub = t >> g.log2C4;
  float m[4], is[4], gm[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = mean[cq * 4 + k];
    is[k] = inv_std_of(var[cq * 4 + k], eps);
    gm[k] = gamma_of(weight, cq * 4 + k, eps);
    b[k] = beta_of(bias, cq * 4 + k);
  }
  const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_wg + rsub;
  float4 v[kNhwcRowsPerThread], r4[kNhwcRowsPerThread];
#pragma unroll
  for (int u = 0; u < kNhwcRowsPerThread; ++u) {
    const int64_t r = r0 + (int64_t)u * g.rpp;
    if (r < rows) {
      const int64_t o = (r << (g.log2C4 + 2)) + cq * 4;
      v[u] = *reinterpret_cast<const float4 *>(x + o);
      if (HAS_RES) r4[u] = *reinterpret_cast<const float4 *>(res + o);
    }
  }
#pragma unroll
  for (int u = 0; u < kNhwcRowsPerThread; ++u) {
    const int64_t r = r0 + (int64_t)u * g.rpp;
    if (r < rows) {
      float4 z;
      z.x = act_fwd<ACT>(HAS_RES ? bn_pre(v[u].x, m[0], is[0], gm[0], b[0]) + r4[u].x : bn_pre(v[u].x, m[0], is[0], gm[0], b[0]), slope);
      z.y = act_fwd<ACT>(HAS_RES ? bn_pre(v[u].y, m[1], is[1], gm[1], b[1]) + r4[u].y : bn_pre(v[u].y, m[1], is[1], gm[1], b[1]), slope);
      z.z = act_fwd<ACT>(HAS_RES ? bn_pre(v[u].z, m[2], is[2], gm[2], b[2]) + r4[u].z : bn_pre(v[u].z, m[2], is[2], gm[2], b[2]), slope);
      z.w = act_fwd<ACT>(HAS_RES ? bn_pre(v[u].w, m[3], is[3], gm[3], b[3]) + r4[u].w : bn_pre(v[u].w, m[3], is[3], gm[3], b[3]), slope);
      *reinterpret_cast<float4 *>(out + (r << (g.log2C4 + 2)) + cq * 4) = z;
    }
  }
}

// K4 (NHWC): dx (and dres for MODE 1), dweight / dbias by workgroup 0.  MODE 2 = MODE 1 for a forward WITHOUT residual:
// inputs (x, dout), the ReLU mask is recomputed from x (bn_pre > 0) -- 4 bytes per element less than reading `out`.
template <int ACT, int MODE, bool WRITE_RES>
__global__ __launch_bounds__(kThreads) void abn_grad_dx_nhwc_kernel(
    const float *a_, const float *b_, const float *c_, const float *__restrict__ mean,
    const float *__restrict__ var, const float *__restrict__ weight, const float *__restrict__ bias,
    const float *__restrict__ edz, const float *__restrict__ eydz, float *dx, float *dres, float *dweight,
    float *dbias, float eps, float slope, int64_t rows, NhwcGeom g, int accumulate) {
  const int t = threadIdx.x;
  const int cq = t & (g.C4 - 1), rsub = t >> g.log2C4;
  float p0[4], p1[4], e[4], ey[4], mul[4], gm[4], bt[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cq * 4 + k;
    const float gam = gamma_of(weight, c, eps), is = inv_std_of(var[c], eps);
    if (MODE == 0) {
      p0[k] = beta_of(bias, c);
      p1[k] = gam;
    } else {
      p0[k] = mean[c];
      p1[k] = is;
    }
    gm[k] = gam;
    bt[k] = MODE == 2 ? beta_of(bias, c) : 0.f;
    e[k] = edz[c];
    ey[k] = eydz[c];
    mul[k] = gam * is;
  }
  const float inv_slope = 1.f / slope;
  const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_wg + rsub;
  float4 va[kNhwcRowsPerThread], vb[kNhwcRowsPerThread], vc[kNhwcRowsPerThread];
#pragma unroll
  for (int u = 0; u < kNhwcRowsPerThread; ++u) {
    const int64_t r = r0 + (int64_t)u * g.rpp;
    if (r < rows) {
      const int64_t o = (r << (g.log2C4 + 2)) + cq * 4;
      va[u] = *reinterpret_cast<const float4 *>(a_ + o);
      vb[u] = *reinterpret_cast<const float4 *>(b_ + o);
      if (MODE == 1) vc[u] = *reinterpret_cast<const float4 *>(c_ + o);
    }
  }
#pragma unroll
  for (int u = 0; u < kNhwcRowsPerThread; ++u) {
    const int64_t r = r0 + (int64_t)u * g.rpp;
    if (r < rows) {
      const float A[4] = {va[u].x, va[u].y, va[u].z, va[u].w};
      const float B[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
      const float Cc[4] = {MODE == 1 ? vc[u].x : 0.f, MODE == 1 ? vc[u].y : 0.f, MODE == 1 ? vc[u].z : 0.f,
                           MODE == 1 ? vc[u].w : 0.f};
      float D[4], R[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float y, dz;
        if (MODE == 0) {
          float zv = A[k];
          dz = B[k];
          act_undo<ACT>(zv, dz, slope, inv_slope);
          y = (zv - p0[k]) / p1[k];
        } else if (MODE == 1) {
          dz = B[k] > 0.f ? Cc[k] : 0.f;
          y = (A[k] - p0[k]) * p1[k];
        } else {
          dz = bn_pre(A[k], p0[k], p1[k], gm[k], bt[k]) > 0.f ? B[k] : 0.f;   // (x, dout)
          y = (A[k] - p0[k]) * p1[k];
        }
        D[k] = (dz - e[k] - y * ey[k]) * mul[k];
        R[k] = dz;
      }
      const int64_t o = (r << (g.log2C4 + 2)) + cq * 4;
      *reinterpret_cast<float4 *>(dx + o) = make_float4(D[0], D[1], D[2], D[3]);
      if (WRITE_RES) *reinterpret_cast<float4 *>(dres + o) = make_float4(R[0], R[1], R[2], R[3]);
    }
  }
  if (blockIdx.x == 0 && rsub == 0) {
    const float norm = (float)rows;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = cq * 4 + k;
      if (dweight != nullptr) {   // bn.cu:217-229 accumulates; accumulate == 0 writes (no zero-fill needed before the call)
        const float wv = weight[c];
        const float gwt = wv > 0.f ? ey[k] * norm : (wv < 0.f ? -ey[k] * norm : 0.f);
        dweight[c] = accumulate ? dweight[c] + gwt : gwt;
      }
      if (dbias != nullptr) dbias[c] = accumulate ? dbias[c] + e[k] * norm : e[k] * norm;
    }
  }
}


// =============================================================================================
// Channels-last reductions, second design (round 2): ONE launch per reduction, finalize included.
//
// Round 1 gave every 32 KiB slab its own workgroup (2113 workgroups for an (8,512,65,65) tensor), each scattering
// 2*C floats as 8-byte stores with a stride of 8*P bytes, followed by a one-wave-per-channel finalize launch over the
// strided partials: the partial traffic rivalled the tensor itself and the finalize launch plus its two kernel
// boundaries cost ~14 us of every ~50 us call (0.24-0.43 of HBM peak).  Now:
//   * at most 256 workgroups of 1024 threads (16 waves, one workgroup per CU), each looping over row slabs with a
//     grid stride; a thread owns one channel quad and keeps its two sums in registers for the whole launch;
//   * wide tensors are cut into CB <= 4 channel blocks of >= 64 channels (256-byte row segments), so a workgroup's
//     partial is 2*CW floats and a channel block's partials total <= 128 KiB;
//   * partials are workgroup-major and contiguous: part[cb][rg][CW4][8] (per quad: four first sums, four second sums);
//   * the LAST workgroup of a channel block to arrive (write-through partial stores -> drained -> agent-scope ticket;
//     MI355X_MICROARCH.md "inter-workgroup visibility") sums the RG partial rows in double precision in a fixed
//     order -- bit-identical whichever workgroup happens to be last -- and finishes the statistics in place
//     (mean / var + running update, or edz / eydz).  No finalize launch, no atomically accumulated floats.
// Tickets live in a library-owned, zero-initialised counter pool (one slot per launch, round robin; the last arriver
// re-arms its counter), so the caller's workspace needs no initialisation.
// =============================================================================================
constexpr int kRedThreads = 1024;
constexpr int kRedMaxWG = 256;
constexpr int kRedMaxCB = 4;
constexpr int kRedSlots = 4096;

struct RedGeom {
  int C4, log2C4;     // channel quads per row
  int CB;             // channel blocks
  int CW4, log2CW4;   // quads per channel block
  int rpp;            // rows per pass of one workgroup (1024 / CW4)
  int RG;             // row groups = workgroups per channel block = partial rows per channel block
  int L;              // floats per partial row (CW4 * 8)
};

static bool make_red_geom(int64_t rows, int C, int U, RedGeom &g) {
  if (rows <= 0 || rows > 2147483647 || C < 4 || C > 4 * kThreads || (C & (C - 1))) return false;
  g.C4 = C / 4;
  g.log2C4 = 0;
  while ((1 << g.log2C4) < g.C4) ++g.log2C4;
  g.CB = C >= 256 ? 4 : (C >= 128 ? 2 : 1);
  g.CW4 = g.C4 / g.CB;
  g.log2CW4 = 0;
  while ((1 << g.log2CW4) < g.CW4) ++g.log2CW4;
  g.rpp = kRedThreads / g.CW4;
  const int64_t want = cdiv(rows, (int64_t)g.rpp * U);
  const int64_t cap = kRedMaxWG / g.CB;
  g.RG = (int)(want < cap ? want : cap);
  g.L = g.CW4 * 8;
  return true;
}

struct RedPool {
  unsigned *ptr = nullptr;
};
static RedPool g_red_pool[64];
static unsigned g_red_next = 0;

// one ticket counter per channel block for this launch (zero on entry, zero again when the launch retires)
static unsigned *red_counters() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  RedPool &p = g_red_pool[dev];
  if (p.ptr == nullptr) {
    unsigned *q = nullptr;
    const size_t bytes = sizeof(unsigned) * kRedSlots * kRedMaxCB * 2;   // ticket counters, then generation words
    if (hipMalloc(reinterpret_cast<void **>(&q), bytes) != hipSuccess) return nullptr;
    if (hipMemset(q, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipFree(q);
      return nullptr;
    }
    p.ptr = q;
  }
  const unsigned slot = __atomic_fetch_add(&g_red_next, 1u, __ATOMIC_RELAXED) % kRedSlots;
  return p.ptr + (size_t)slot * kRedMaxCB;
}
// the generation words (fused one-launch passes) that belong to a counter slot
static unsigned *red_gens(unsigned *counters) { return counters + (size_t)kRedSlots * kRedMaxCB; }

// 16-byte write-through store / L1-bypassing load (sc0 sc1): the hand-off traffic of the reductions below.  A plain
// store would stay dirty in the producer XCD's L2 until an agent-scope release (buffer_wbl2) flushes that WHOLE L2 --
// right after a convolution that is megabytes of unrelated dirty lines on the reduction's critical path.  With
// write-through partials the producer only drains its own stores (s_waitcnt vmcnt(0)) before taking its ticket, and
// the last arriver reads them with sc1 loads: no release / acquire fence at all (MI355X_MICROARCH.md, "valid forms":
// sc0 sc1 stores and loads on both sides).
__device__ __forceinline__ void store_wt16(float *p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void load_wt16x4(const float *p0, const float *p1, const float *p2, const float *p3, f32x4 &v0,
                                            f32x4 &v1, f32x4 &v2, f32x4 &v3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
      "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}

// Workgroup epilogue of a channels-last reduction.  In: every thread's eight running sums.  Out: `true` in all
// threads of the channel block's last-arriving workgroup, with the block's totals in fin[cq * 8 + k] (double).
// lds: kRedThreads * 4 doubles; fin: L doubles.
__device__ __forceinline__ bool red_finish(float (&s1)[4], float (&s2)[4], float *__restrict__ part,
                                           unsigned *counter, const RedGeom &g, int cb, int rg, double *lds,
                                           double *fin, unsigned *ticket_s) {
  const int t = threadIdx.x, lane = t & (kWave - 1);
  float a[8] = {s1[0], s1[1], s1[2], s1[3], s2[0], s2[1], s2[2], s2[3]};
  // lanes of a wave that share a channel quad differ only in the row bits of the lane index: butterfly over those
  for (int m = kWave / 2; m >= g.CW4; m >>= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += __shfl_xor(a[k], m, kWave);
  }
  float *ldsf = reinterpret_cast<float *>(lds);
  const int cq = t & (g.CW4 - 1);
  int slot, nslots;
  bool writer;
  if (g.CW4 < kWave) {
    slot = t / kWave;
    nslots = kRedThreads / kWave;
    writer = lane < g.CW4;
  } else {
    slot = t >> g.log2CW4;
    nslots = kRedThreads >> g.log2CW4;
    writer = true;
  }
  if (writer) {
    float *o = ldsf + ((int64_t)slot * g.CW4 + cq) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = a[k];
  }
  __syncthreads();
  const int L4 = g.L >> 2;                 // float4 per partial row (a power of two, 2 ... 128)
  float *mine = part + ((int64_t)cb * g.RG + rg) * g.L;
  if (t < L4) {                            // fixed-order sum over the slots, one 16-byte write-through store per lane
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < nslots; ++sl) {
      const float4 v = *reinterpret_cast<const float4 *>(ldsf + sl * g.L + 4 * t);
      s[0] += v.x;
      s[1] += v.y;
      s[2] += v.z;
      s[3] += v.w;
    }
    store_wt16(mine + 4 * t, s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's partial has left for memory
  }
  __syncthreads();
  if (t == 0) *ticket_s = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*ticket_s != (unsigned)g.RG - 1u) return false;
  if (t == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for a later launch
  // ---- last arriver: fixed-order double-precision sum of the RG partial rows of this channel block ----
  const int NP = kRedThreads / L4;         // row phases
  const int j4 = t & (L4 - 1), ph = t / L4;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const float *col = part + (int64_t)cb * g.RG * g.L + j4 * 4;
  for (int r = ph; r < g.RG; r += 4 * NP) {
    f32x4 v[4];
    const float *q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + u * NP;
      q[u] = col + (int64_t)(rr < g.RG ? rr : r) * g.L;     // out-of-range phases re-read row r and are not added
    }
    load_wt16x4(q[0], q[1], q[2], q[3], v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + u * NP < g.RG) {
        acc[0] += (double)v[u][0];
        acc[1] += (double)v[u][1];
        acc[2] += (double)v[u][2];
        acc[3] += (double)v[u][3];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) lds[((int64_t)ph * L4 + j4) * 4 + q] = acc[q];
  __syncthreads();
  for (int i = t; i < g.L; i += kRedThreads) {
    double s = 0.0;
    for (int p = 0; p < NP; ++p) s += lds[(int64_t)p * g.L + i];
    fin[i] = s;
  }
  __syncthreads();
  return true;
}

// K1 (NHWC): shifted sums of (x - K), (x - K)^2 with K = median3 of the channel; mean / biased var (+ running update) by the last arriver
template <int U>
__global__ __launch_bounds__(kRedThreads) void abn_stats_nhwc2_kernel(
    const float *__restrict__ x, float *__restrict__ part, unsigned *counters, float *__restrict__ mean,
    float *__restrict__ var, float *running_mean, float *running_var, int64_t rows, RedGeom g, float momentum,
    float n_total) {
  __shared__ double lds[kRedThreads * 4];
  __shared__ double fin[kRedThreads * 2];
  __shared__ unsigned ticket_s;
  const int t = threadIdx.x;
  const int cb = blockIdx.x % g.CB, rg = blockIdx.x / g.CB;
  const int cq = t & (g.CW4 - 1), rsub = t >> g.log2CW4;
  const int col = (cb * g.CW4 + cq) * 4;
  float4 K;   // per-channel pivot: median of the channel's first / middle / last row (median3 above)
  {
    const float4 ka = *reinterpret_cast<const float4 *>(x + col);
    const float4 kb = *reinterpret_cast<const float4 *>(x + ((rows / 2) << (g.log2C4 + 2)) + col);
    const float4 kc = *reinterpret_cast<const float4 *>(x + ((rows - 1) << (g.log2C4 + 2)) + col);
    K = make_float4(median3(ka.x, kb.x, kc.x), median3(ka.y, kb.y, kc.y), median3(ka.z, kb.z, kc.z), median3(ka.w, kb.w, kc.w));
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t slab = (int64_t)g.rpp * U;
  for (int64_t base = (int64_t)rg * slab + rsub; base < rows; base += (int64_t)g.RG * slab) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + (int64_t)u * g.rpp;
      if (r < rows) v[u] = *reinterpret_cast<const float4 *>(x + (r << (g.log2C4 + 2)) + col);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + (int64_t)u * g.rpp;
      if (r < rows) {
        const float d0 = v[u].x - K.x, d1 = v[u].y - K.y, d2 = v[u].z - K.z, d3 = v[u].w - K.w;
        s1[0] += d0; s1[1] += d1; s1[2] += d2; s1[3] += d3;
        s2[0] += d0 * d0; s2[1] += d1 * d1; s2[2] += d2 * d2; s2[3] += d3 * d3;
      }
    }
  }
  if (!red_finish(s1, s2, part, counters + cb, g, cb, rg, lds, fin, &ticket_s)) return;
  const int CW = g.CW4 * 4;
  if (t < CW) {
    const int c = cb * CW + t;
    const double cnt = (double)rows;
    const double d = fin[(t >> 2) * 8 + (t & 3)] / cnt;
    double v = fin[(t >> 2) * 8 + 4 + (t & 3)] / cnt - d * d;
    if (v < 0.0) v = 0.0;
    const int64_t ldr = (int64_t)g.C4 * 4;
    const double Kc = (double)median3(x[c], x[(rows / 2) * ldr + c], x[(rows - 1) * ldr + c]);   // the same pivot as above
    const float m_f = (float)(Kc + d), v_f = (float)v;
    mean[c] = m_f;
    var[c] = v_f;
    if (running_mean != nullptr) running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * m_f;
    if (running_var != nullptr) running_var[c] = running_var[c] * (1.f - momentum) + momentum * unbiased_of(v_f, n_total);
  }
}

// K3 (NHWC): edz / eydz.  MODE 0: y from the saved OUTPUT z (activation ACT undone in registers, the in-place ABN);
// MODE 1: fused BN+ReLU: inputs (x, out, dout), y from x, mask = out > 0.
// MODE 2: the same for a forward without residual: inputs (x, dout), mask = bn_pre(x) > 0 recomputed.
template <int ACT, int MODE, int U>
__global__ __launch_bounds__(kRedThreads) void abn_grad_nhwc2_kernel(
    const float *__restrict__ a_, const float *__restrict__ b_, const float *__restrict__ c_,
    const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ weight,
    const float *__restrict__ bias, float *__restrict__ part, unsigned *counters, float *__restrict__ edz,
    float *__restrict__ eydz, float eps, float slope, int64_t rows, RedGeom g) {
  __shared__ double lds[kRedThreads * 4];
  __shared__ double fin[kRedThreads * 2];
  __shared__ unsigned ticket_s;
  const int t = threadIdx.x;
  const int cb = blockIdx.x % g.CB, rg = blockIdx.x / g.CB;
  const int cq = t & (g.CW4 - 1), rsub = t >> g.log2CW4;
  const int col = (cb * g.CW4 + cq) * 4;
  float p0[4], p1[4], gm[4], bt[4];  // MODE 0: beta, gamma   MODE 1 / 2: mean, inv_std (+ gamma, beta for the mask in MODE 2)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (MODE == 0) {
      p0[k] = beta_of(bias, col + k);
      p1[k] = gamma_of(weight, col + k, eps);
    } else {
      p0[k] = mean[col + k];
      p1[k] = inv_std_of(var[col + k], eps);
    }
    gm[k] = MODE == 2 ? gamma_of(weight, col + k, eps) : 0.f;
    bt[k] = MODE == 2 ? beta_of(bias, col + k) : 0.f;
  }
  const float inv_slope = 1.f / slope;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t slab = (int64_t)g.rpp * U;
  for (int64_t base = (int64_t)rg * slab + rsub; base < rows; base += (int64_t)g.RG * slab) {
    float4 va[U], vb[U], vc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + (int64_t)u * g.rpp;
      if (r < rows) {
        const int64_t o = (r << (g.log2C4 + 2)) + col;
        va[u] = *reinterpret_cast<const float4 *>(a_ + o);
        vb[u] = *reinterpret_cast<const float4 *>(b_ + o);
        if (MODE == 1) vc[u] = *reinterpret_cast<const float4 *>(c_ + o);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + (int64_t)u * g.rpp;
      if (r < rows) {
        const float A[4] = {va[u].x, va[u].y, va[u].z, va[u].w};
        const float B[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
        const float Cc[4] = {MODE == 1 ? vc[u].x : 0.f, MODE == 1 ? vc[u].y : 0.f, MODE == 1 ? vc[u].z : 0.f,
                             MODE == 1 ? vc[u].w : 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float y, dz;
          if (MODE == 0) {
            float zv = A[k];
            dz = B[k];
            act_undo<ACT>(zv, dz, slope, inv_slope);
            y = (zv - p0[k]) / p1[k];
          } else if (MODE == 1) {
            dz = B[k] > 0.f ? Cc[k] : 0.f;          // (x, out, dout)
            y = (A[k] - p0[k]) * p1[k];
          } else {
            dz = bn_pre(A[k], p0[k], p1[k], gm[k], bt[k]) > 0.f ? B[k] : 0.f;   // (x, dout)
            y = (A[k] - p0[k]) * p1[k];
          }
          s1[k] += dz;
          s2[k] += y * dz;
        }
      }
    }
  }
  if (!red_finish(s1, s2, part, counters + cb, g, cb, rg, lds, fin, &ticket_s)) return;
  const int CW = g.CW4 * 4;
  if (t < CW) {
    const int c = cb * CW + t;
    const double cnt = (double)rows;
    edz[c] = (float)(fin[(t >> 2) * 8 + (t & 3)] / cnt);       // bn.cu:176
    eydz[c] = (float)(fin[(t >> 2) * 8 + 4 + (t & 3)] / cnt);  // bn.cu:177
  }
}

// =============================================================================================
// Channels-last TRAINING passes as ONE launch with the tensor held in registers (round 3).
//
// The two-launch form above reads x for the statistics, ends the launch, and reads x again for the normalisation:
// 12 bytes per element, and for the student's 4-70 MB tensors a second launch whose whole life is 8-20 us.  The chip's
// register file is larger than those tensors (256 CUs x 512 KiB = 128 MiB): a launch of <= 256 workgroups x 1024
// threads in which a thread keeps its NR <= 17 float4 of x in VGPRs can hold up to ~71 MB.  So:
//   phase 1  every thread loads its rows ONCE, accumulates the shifted sums, red_finish() as before;
//   hand-off the channel block's last arriver finishes mean / var (+ running statistics), stores them write-through
//            and bumps the block's generation word; the other workgroups of the block spin (bounded) on that word --
//            a grid barrier per channel block, safe because the launch never exceeds one workgroup per CU
//            (MI355X_MICROARCH.md: "size the grid ... bound every spin");
//   phase 2  normalise + affine + activation (+ residual) from the registers, one store.
// HBM traffic: 8 bytes per element instead of 12 (forward), 12 instead of 20 (backward: z and dz are held, y and the
// masked dz are kept instead of the raw inputs).  Tensors that do not fit (the 256 x 256 stem layers, C = 512 in the
// backward) keep the two-launch path.  A spin that times out (2 s of wall clock: the grid was not co-resident) poisons
// the statistics with NaN instead of hanging the device: loud, not fatal.
// The generation word is read by every workgroup BEFORE it takes its ticket and bumped by the last arriver AFTER all
// tickets: no host-side sequence number, so a captured launch can be replayed.
// =============================================================================================
constexpr uint64_t kFuseSpinTicks = 200000000ull;   // wall_clock64(): 100 MHz

__device__ __forceinline__ unsigned gen_load(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// thread 0 spins until the generation word moves on from gen0; false = timed out (status word kStatusFusedTimeout raised)
__device__ __forceinline__ bool gen_wait(const unsigned *p, unsigned gen0, uint64_t ticks, unsigned *status) {
  const uint64_t t0 = wall_clock64();
  while (gen_load(p) == gen0) {
    __builtin_amdgcn_s_sleep(4);
    if (wall_clock64() - t0 > ticks) {
      raise_status(status, kStatusFusedTimeout, 1u);
      return false;
    }
  }
  return true;
}
__device__ __forceinline__ void store_wt4(float *p, float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load_wt16(const float *p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct FuseGeom {
  RedGeom r;
  int nr;        // rows per thread actually used (<= NR of the instantiation)
};

// NRmax rows per thread: geometry with the most workgroups (<= 256 / CB per channel block) and the fewest rows each
static bool make_fuse_geom(int64_t rows, int C, int nr_max, int max_wg, FuseGeom &f) {
  if (max_wg < kRedMaxCB || !make_red_geom(rows, C, 1, f.r)) return false;
  RedGeom &g = f.r;
  const int64_t cap = max_wg / g.CB;
  const int64_t want = cdiv(rows, g.rpp);
  g.RG = (int)(want < cap ? want : cap);
  const int64_t nr = cdiv(rows, (int64_t)g.RG * g.rpp);
  if (nr > nr_max) return false;
  f.nr = (int)nr;
  g.RG = (int)cdiv(rows, (int64_t)g.rpp * f.nr);     // drop workgroups that would own no row
  return true;
}

// SYNC: the cross-replica exchange of InPlaceABNSync (libs/functions.py:185-209) happens INSIDE the launch: the channel
// block's last arriver stores the block's local [mean | var] into every replica's mailbox, raises / awaits the block's flag
// word (sync_dev.hpp) and applies the combine rule before it releases the block's workgroups -- the register-resident
// one-launch form survives N > 1 (round 3 fell back to stats + exchange + apply: three launches, 12 B/element).
template <int ACT, bool HAS_RES, int NR, bool SYNC>
__global__ __launch_bounds__(kRedThreads) void abn_fwd_fused_nhwc_kernel(
    const float *x, const float *res, float *out, float *__restrict__ part, unsigned *counters, unsigned *gens,
    float *mean, float *var, float *running_mean, float *running_var, const float *__restrict__ weight,
    const float *__restrict__ bias, int64_t rows, RedGeom g, int nr, float momentum, float eps, float slope,
    SyncArgs sy, const float *__restrict__ rweights, float n_pooled) {
  __shared__ double lds[kRedThreads * 4];
  __shared__ double fin[kRedThreads * 2];
  __shared__ unsigned ticket_s;
  __shared__ unsigned gen_s;
  const int t = threadIdx.x;
  const int cb = blockIdx.x % g.CB, rg = blockIdx.x / g.CB;
  const int cq = t & (g.CW4 - 1), rsub = t >> g.log2CW4;
  const int col = (cb * g.CW4 + cq) * 4;
  if (t == 0) gen_s = gen_load(gens + cb);              // before this workgroup's ticket, hence before the bump
  const int r0 = rg * g.rpp * nr + rsub;      // 32-bit row / element indices: a fused launch holds < 2^27 elements
  const int nrows = (int)rows;
  float4 v[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = r0 + u * g.rpp;
    if (u < nr && r < nrows) v[u] = *reinterpret_cast<const float4 *>(x + ((r << (g.log2C4 + 2)) + col));
  }
  float4 K;
  {
    const float4 ka = *reinterpret_cast<const float4 *>(x + col);
    const float4 kb = *reinterpret_cast<const float4 *>(x + ((rows / 2) << (g.log2C4 + 2)) + col);
    const float4 kc = *reinterpret_cast<const float4 *>(x + ((rows - 1) << (g.log2C4 + 2)) + col);
    K = make_float4(median3(ka.x, kb.x, kc.x), median3(ka.y, kb.y, kc.y), median3(ka.z, kb.z, kc.z), median3(ka.w, kb.w, kc.w));
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = r0 + u * g.rpp;
    if (u < nr && r < nrows) {
      const float d0 = v[u].x - K.x, d1 = v[u].y - K.y, d2 = v[u].z - K.z, d3 = v[u].w - K.w;
      s1[0] += d0; s1[1] += d1; s1[2] += d2; s1[3] += d3;
      s2[0] += d0 * d0; s2[1] += d1 * d1; s2[2] += d2 * d2; s2[3] += d3 * d3;
    }
  }
  const int CW = g.CW4 * 4;
  bool good = true;
  float loc_m = 0.f, loc_v = 0.f;            // SYNC: this replica's statistics of channel cb * CW + t
  if (red_finish(s1, s2, part, counters + cb, g, cb, rg, lds, fin, &ticket_s)) {
    if (t < CW) {
      const int c = cb * CW + t;
      const double cnt = (double)rows;
      const double d = fin[(t >> 2) * 8 + (t & 3)] / cnt;
      double vv = fin[(t >> 2) * 8 + 4 + (t & 3)] / cnt - d * d;
      if (vv < 0.0) vv = 0.0;
      const int64_t ldr = (int64_t)g.C4 * 4;
      const double Kc = (double)median3(x[c], x[(rows / 2) * ldr + c], x[(rows - 1) * ldr + c]);
      const float m_f = (float)(Kc + d), v_f = (float)vv;
      if (SYNC) {
        loc_m = m_f;
        loc_v = v_f;
      } else {
        store_wt4(mean + c, m_f);
        store_wt4(var + c, v_f);
        if (running_mean != nullptr) running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * m_f;
        if (running_var != nullptr) running_var[c] = running_var[c] * (1.f - momentum) + momentum * unbiased_of(v_f, (float)rows);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if (SYNC) {
      const int Ct = g.C4 * 4, c = cb * CW + t;
      const bool arrived = sync_exchange(sy, cb, 1, [&](float *dst) {
        if (t < CW) {
          store_sys(dst + c, loc_m);
          store_sys(dst + Ct + c, loc_v);
        }
      }, &ticket_s);
      if (t < CW) {
        float m, v;
        combine_channel(sy.d.world, Ct, c, [&](int gq, int j) { return sync_payload(sy.d, sy.seq, gq, j); }, rweights, sy.d.rank,
                        n_pooled, momentum, m, v, arrived ? running_mean : nullptr, arrived ? running_var : nullptr);
        store_wt4(mean + c, arrived ? m : __builtin_nanf(""));
        store_wt4(var + c, arrived ? v : __builtin_nanf(""));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (t == 0) __hip_atomic_store(gens + cb, gen_s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    // (SYNC: the block's last arriver may itself be waiting for a peer replica: its limit on top of the grid barrier's own)
    if (t == 0) ticket_s = gen_wait(gens + cb, gen_s, kFuseSpinTicks + (SYNC ? sy.spin_ticks : 0ull), sy.status) ? 1u : 0u;
    __syncthreads();
    good = ticket_s != 0u;
  }
  const f32x4 m4 = load_wt16(mean + col), v4 = load_wt16(var + col);
  float m[4], is[4], gm[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = good ? m4[k] : __builtin_nanf("");
    is[k] = inv_std_of(v4[k], eps);
    gm[k] = gamma_of(weight, col + k, eps);
    b[k] = beta_of(bias, col + k);
  }
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = r0 + u * g.rpp;
    if (u < nr && r < nrows) {
      const int o = (r << (g.log2C4 + 2)) + col;
      float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS_RES) r4 = *reinterpret_cast<const float4 *>(res + o);
      float4 z;
      z.x = act_fwd<ACT>(HAS_RES ? bn_pre(v[u].x, m[0], is[0], gm[0], b[0]) + r4.x : bn_pre(v[u].x, m[0], is[0], gm[0], b[0]), slope);
      z.y = act_fwd<ACT>(HAS_RES ? bn_pre(v[u].y, m[1], is[1], gm[1], b[1]) + r4.y : bn_pre(v[u].y, m[1], is[1], gm[1], b[1]), slope);
      z.z = act_fwd<ACT>(HAS_RES ? bn_pre(v[u].z, m[2], is[2], gm[2], b[2]) + r4.z : bn_pre(v[u].z, m[2], is[2], gm[2], b[2]), slope);
      z.w = act_fwd<ACT>(HAS_RES ? bn_pre(v[u].w, m[3], is[3], gm[3], b[3]) + r4.w : bn_pre(v[u].w, m[3], is[3], gm[3], b[3]), slope);
      *reinterpret_cast<float4 *>(out + o) = z;
    }
  }
}

// Backward, both passes in one launch.  MODE as in abn_grad_nhwc2_kernel / abn_grad_dx_nhwc_kernel; a thread keeps (y, dz) of
// its NR rows -- the masked / activation-undone gradient and the normalised input -- between the phases.
// SYNC: [edz | eydz] of the channel block are exchanged by the block's last arriver inside the launch (functions.py:263-280).
template <int ACT, int MODE, bool WRITE_RES, int NR, bool SYNC>
__global__ __launch_bounds__(kRedThreads) void abn_bwd_fused_nhwc_kernel(
    const float *a_, const float *b_, const float *c_, const float *__restrict__ mean, const float *__restrict__ var,
    const float *__restrict__ weight, const float *__restrict__ bias, float *__restrict__ part, unsigned *counters,
    unsigned *gens, float *edz, float *eydz, float *dx, float *dres, float *dweight, float *dbias, float eps, float slope,
    int64_t rows, RedGeom g, int nr, int accumulate, SyncArgs sy, const float *__restrict__ rweights) {
  __shared__ double lds[kRedThreads * 4];
  __shared__ double fin[kRedThreads * 2];
  __shared__ unsigned ticket_s;
  __shared__ unsigned gen_s;
  const int t = threadIdx.x;
  const int cb = blockIdx.x % g.CB, rg = blockIdx.x / g.CB;
  const int cq = t & (g.CW4 - 1), rsub = t >> g.log2CW4;
  const int col = (cb * g.CW4 + cq) * 4;
  if (t == 0) gen_s = gen_load(gens + cb);
  float p0[4], p1[4], gm[4], bt[4], mul[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gam = gamma_of(weight, col + k, eps), is = inv_std_of(var[col + k], eps);
    if (MODE == 0) {
      p0[k] = beta_of(bias, col + k);
      p1[k] = gam;
    } else {
      p0[k] = mean[col + k];
      p1[k] = is;
    }
    gm[k] = gam;
    bt[k] = MODE == 2 ? beta_of(bias, col + k) : 0.f;
    mul[k] = gam * is;
  }
  const float inv_slope = 1.f / slope;
  const int r0 = rg * g.rpp * nr + rsub;      // 32-bit row / element indices: a fused launch holds < 2^27 elements
  const int nrows = (int)rows;
  float4 Y[NR], DZ[NR];
  constexpr int CH = 2;                    // rows loaded per trip: bounds the registers holding raw inputs
#pragma unroll
  for (int u0 = 0; u0 < NR; u0 += CH) {
    float4 va[CH], vb[CH], vc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int u = u0 + j;
      const int r = r0 + u * g.rpp;
      if (u < NR && u < nr && r < nrows) {
        const int o = (r << (g.log2C4 + 2)) + col;
        va[j] = *reinterpret_cast<const float4 *>(a_ + o);
        vb[j] = *reinterpret_cast<const float4 *>(b_ + o);
        if (MODE == 1) vc[j] = *reinterpret_cast<const float4 *>(c_ + o);
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int u = u0 + j;
      const int r = r0 + u * g.rpp;
      if (u < NR && u < nr && r < nrows) {
        const float A[4] = {va[j].x, va[j].y, va[j].z, va[j].w};
        const float B[4] = {vb[j].x, vb[j].y, vb[j].z, vb[j].w};
        const float Cc[4] = {MODE == 1 ? vc[j].x : 0.f, MODE == 1 ? vc[j].y : 0.f, MODE == 1 ? vc[j].z : 0.f,
                             MODE == 1 ? vc[j].w : 0.f};
        float y[4], dz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (MODE == 0) {
            float zv = A[k];
            dz[k] = B[k];
            act_undo<ACT>(zv, dz[k], slope, inv_slope);
            y[k] = (zv - p0[k]) / p1[k];
          } else if (MODE == 1) {
            dz[k] = B[k] > 0.f ? Cc[k] : 0.f;
            y[k] = (A[k] - p0[k]) * p1[k];
          } else {
            dz[k] = bn_pre(A[k], p0[k], p1[k], gm[k], bt[k]) > 0.f ? B[k] : 0.f;
            y[k] = (A[k] - p0[k]) * p1[k];
          }
        }
        Y[u] = make_float4(y[0], y[1], y[2], y[3]);
        DZ[u] = make_float4(dz[0], dz[1], dz[2], dz[3]);
      }
    }
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = r0 + u * g.rpp;
    if (u < nr && r < nrows) {
      s1[0] += DZ[u].x; s1[1] += DZ[u].y; s1[2] += DZ[u].z; s1[3] += DZ[u].w;
      s2[0] += Y[u].x * DZ[u].x; s2[1] += Y[u].y * DZ[u].y; s2[2] += Y[u].z * DZ[u].z; s2[3] += Y[u].w * DZ[u].w;
    }
  }
  const int CW = g.CW4 * 4;
  bool good = true;
  if (red_finish(s1, s2, part, counters + cb, g, cb, rg, lds, fin, &ticket_s)) {
    float loc_e = 0.f, loc_ey = 0.f;
    if (t < CW) {
      const double cnt = (double)rows;
      loc_e = (float)(fin[(t >> 2) * 8 + (t & 3)] / cnt);          // bn.cu:176
      loc_ey = (float)(fin[(t >> 2) * 8 + 4 + (t & 3)] / cnt);     // bn.cu:177
    }
    bool arrived = true;
    if (SYNC) {
      const int Ct = g.C4 * 4, c = cb * CW + t;
      arrived = sync_exchange(sy, cb, 1, [&](float *dst) {
        if (t < CW) {
          store_sys(dst + c, loc_e);
          store_sys(dst + Ct + c, loc_ey);
        }
      }, &ticket_s);
      if (t < CW) {
        loc_e = arrived ? sync_weighted_sum(sy.d, sy.seq, c, rweights) : __builtin_nanf("");
        loc_ey = arrived ? sync_weighted_sum(sy.d, sy.seq, Ct + c, rweights) : __builtin_nanf("");
      }
    }
    if (t < CW) {
      const int c = cb * CW + t;
      const float e_f = loc_e, ey_f = loc_ey;
      store_wt4(edz + c, e_f);
      store_wt4(eydz + c, ey_f);
      const float norm = (float)rows;
      if (dweight != nullptr) {   // bn.cu:217-229 accumulates; accumulate == 0 writes
        const float wv = weight[c];
        const float gwt = wv > 0.f ? ey_f * norm : (wv < 0.f ? -ey_f * norm : 0.f);
        dweight[c] = accumulate ? dweight[c] + gwt : gwt;
      }
      if (dbias != nullptr) dbias[c] = accumulate ? dbias[c] + e_f * norm : e_f * norm;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (t == 0) __hip_atomic_store(gens + cb, gen_s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (t == 0) ticket_s = gen_wait(gens + cb, gen_s, kFuseSpinTicks + (SYNC ? sy.spin_ticks : 0ull), sy.status) ? 1u : 0u;
    __syncthreads();
    good = ticket_s != 0u;
  }
  const f32x4 e4 = load_wt16(edz + col), ey4 = load_wt16(eydz + col);
  float e[4], ey[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    e[k] = good ? e4[k] : __builtin_nanf("");
    ey[k] = ey4[k];
  }
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = r0 + u * g.rpp;
    if (u < nr && r < nrows) {
      const int o = (r << (g.log2C4 + 2)) + col;
      float4 D;
      D.x = (DZ[u].x - e[0] - Y[u].x * ey[0]) * mul[0];
      D.y = (DZ[u].y - e[1] - Y[u].y * ey[1]) * mul[1];
      D.z = (DZ[u].z - e[2] - Y[u].z * ey[2]) * mul[2];
      D.w = (DZ[u].w - e[3] - Y[u].w * ey[3]) * mul[3];
      *reinterpret_cast<float4 *>(dx + o) = D;
      if (WRITE_RES) *reinterpret_cast<float4 *>(dres + o) = DZ[u];
    }
  }
}

// =============================================================================================
// Student stem (round 6): training BatchNorm -> ReLU -> MaxPool2d(3, 2, 1, ceil_mode) on the (8, 128, 256, 256)
// conv3 output (networks/pspnet_combine.py:176-180) WITHOUT the 268 MB normalised tensor.  The reference (and rounds 1-5) ran
// statistics, normalise + ReLU (read 268 MB, write 268 MB), pool (read 268 MB) and in backward un-pool (write 268 MB), reduce
// (read 2 x 268), dx (read 2 x 268, write 268).  Here:
//   forward   statistics (the ordinary pass), then ONE kernel that reads x, evaluates y = relu(bn(x)) per element with the
//             forward's own expression (bn_pre) and pools y under PyTorch's rule, writing only the 68 MB pooled map + one byte of
//             argmax per element (the winner's position inside its window, csrc/maxpool.hip's code) -- bit for bit what
//             normalise-then-pool produces, indices included (ties at zero after the ReLU go to the first window position in both);
//   backward  two passes over 2 x 2 INPUT blocks (the four positions of a block only ever belong to the same four windows, see
//             maxpool.hip): the pooled gradient is gathered through the argmax bytes, masked with bn_pre(x) > 0 recomputed from x,
//             and enters the usual edz / eydz reduction and dx formula.  The un-pooled 268 MB gradient never exists.
// =============================================================================================
__device__ __forceinline__ void pool_take(float v, int code, float &best, int &arg) {
  if (v > best || v != v) {        // PyTorch's max-pool rule (first maximum wins, NaN propagates); maxpool.hip: take()
    best = v;
    arg = code;
  }
}

__global__ __launch_bounds__(kThreads) void abn_relu_maxpool_nhwc_kernel(
    const float *__restrict__ x, float *__restrict__ pooled, uint8_t *__restrict__ arg, const float *__restrict__ mean,
    const float *__restrict__ var, const float *__restrict__ weight, const float *__restrict__ bias, float eps, int64_t items,
    int H, int W, int OH, int OW, int C4) {
  const int64_t item = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (item >= items) return;
  const int q = (int)(item % C4);
  const int ox = (int)((item / C4) % OW);
  const int oy = (int)((item / ((int64_t)C4 * OW)) % OH);
  const int b = (int)(item / ((int64_t)C4 * OW * OH));
  float m[4], is[4], gm[4], bt[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = mean[q * 4 + k];
    is[k] = inv_std_of(var[q * 4 + k], eps);
    gm[k] = gamma_of(weight, q * 4 + k, eps);
    bt[k] = beta_of(bias, q * 4 + k);
  }
  const int ys = 2 * oy - 1, xs = 2 * ox - 1;
  const int y0 = ys < 0 ? 0 : ys, x0 = xs < 0 ? 0 : xs;
  const int y1 = ys + 3 < H ? ys + 3 : H, x1 = xs + 3 < W ? xs + 3 : W;
  const float ninf = -__builtin_huge_valf();
  float best[4] = {ninf, ninf, ninf, ninf};
  const int first = (y0 - ys) * 3 + (x0 - xs);
  int a[4] = {first, first, first, first};
  const float *src = x + ((int64_t)b * H * W) * C4 * 4 + q * 4;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      const float4 v = *reinterpret_cast<const float4 *>(src + ((int64_t)yy * W + xx) * C4 * 4);
      const int code = (yy - ys) * 3 + (xx - xs);
      const float V[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) pool_take(act_fwd<SKD_ACT_RELU>(bn_pre(V[k], m[k], is[k], gm[k], bt[k]), 0.f), code, best[k], a[k]);
    }
  *reinterpret_cast<float4 *>(pooled + item * 4) = make_float4(best[0], best[1], best[2], best[3]);
  *reinterpret_cast<uchar4 *>(arg + item * 4) =
      make_uchar4((unsigned char)a[0], (unsigned char)a[1], (unsigned char)a[2], (unsigned char)a[3]);
}

// One 2 x 2 input block (i, j) of image b, channel quad at column `col`: the four windows (i, j), (i, j+1), (i+1, j), (i+1, j+1)
// that can contain its positions, and per position p = 2 * dy + dx the gathered gradient (maxpool.hip's code table, terms in
// window order).  valid[p]: the position exists.
struct PoolBlock {
  float g[4][4];      // [position][channel]
  bool valid[4];
};
__device__ __forceinline__ void pool_block_gather(const float *__restrict__ dy, const uint8_t *__restrict__ arg, int b, int i, int j,
                                                  int H, int W, int OH, int OW, int C, int col, PoolBlock &o) {
  float4 g[4];
  uchar4 a[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int oy = i + (k >> 1), ox = j + (k & 1);
    if (oy < OH && ox < OW) {
      const int64_t off = (((int64_t)b * OH + oy) * OW + ox) * C + col;
      a[k] = *reinterpret_cast<const uchar4 *>(arg + off);
      g[k] = *reinterpret_cast<const float4 *>(dy + off);
    } else {
      a[k] = make_uchar4(255, 255, 255, 255);          // no such window: matches no code
      g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // position -> (window, code) pairs, in window order:  (2i, 2j): (0, 4)   (2i, 2j+1): (0, 5) (1, 3)   (2i+1, 2j): (0, 7) (2, 1)
  //                                                     (2i+1, 2j+1): (0, 8) (1, 6) (2, 2) (3, 0)
  const unsigned char A[4][4] = {{a[0].x, a[0].y, a[0].z, a[0].w}, {a[1].x, a[1].y, a[1].z, a[1].w},
                                 {a[2].x, a[2].y, a[2].z, a[2].w}, {a[3].x, a[3].y, a[3].z, a[3].w}};
  const float G[4][4] = {{g[0].x, g[0].y, g[0].z, g[0].w}, {g[1].x, g[1].y, g[1].z, g[1].w},
                         {g[2].x, g[2].y, g[2].z, g[2].w}, {g[3].x, g[3].y, g[3].z, g[3].w}};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    if (A[0][c] == 4) r0 += G[0][c];
    if (A[0][c] == 5) r1 += G[0][c];
    if (A[1][c] == 3) r1 += G[1][c];
    if (A[0][c] == 7) r2 += G[0][c];
    if (A[2][c] == 1) r2 += G[2][c];
    if (A[0][c] == 8) r3 += G[0][c];
    if (A[1][c] == 6) r3 += G[1][c];
    if (A[2][c] == 2) r3 += G[2][c];
    if (A[3][c] == 0) r3 += G[3][c];
    o.g[0][c] = r0;
    o.g[1][c] = r1;
    o.g[2][c] = r2;
    o.g[3][c] = r3;
  }
  o.valid[0] = true;
  o.valid[1] = 2 * j + 1 < W;
  o.valid[2] = 2 * i + 1 < H;
  o.valid[3] = o.valid[1] && o.valid[2];
}

// edz / eydz of the fused stem: rows of the reduction geometry = 2 x 2 blocks (B * H2 * W2), the expectation over B * H * W positions
template <int U>
__global__ __launch_bounds__(kRedThreads) void abn_pool_grad_nhwc2_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const uint8_t *__restrict__ arg, const float *__restrict__ mean,
    const float *__restrict__ var, const float *__restrict__ weight, const float *__restrict__ bias, float *__restrict__ part,
    unsigned *counters, float *__restrict__ edz, float *__restrict__ eydz, float eps, int64_t blocks, int H, int W, int OH,
    int OW, int H2, int W2, RedGeom g) {
  __shared__ double lds[kRedThreads * 4];
  __shared__ double fin[kRedThreads * 2];
  __shared__ unsigned ticket_s;
  const int t = threadIdx.x;
  const int cb = blockIdx.x % g.CB, rg = blockIdx.x / g.CB;
  const int cq = t & (g.CW4 - 1), rsub = t >> g.log2CW4;
  const int col = (cb * g.CW4 + cq) * 4, C = g.C4 * 4;
  float m[4], is[4], gm[4], bt[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = mean[col + k];
    is[k] = inv_std_of(var[col + k], eps);
    gm[k] = gamma_of(weight, col + k, eps);
    bt[k] = beta_of(bias, col + k);
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t slab = (int64_t)g.rpp * U;
  for (int64_t base = (int64_t)rg * slab + rsub; base < blocks; base += (int64_t)g.RG * slab) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + (int64_t)u * g.rpp;
      if (r < blocks) {
        const int j = (int)(r % W2), i = (int)((r / W2) % H2), b = (int)(r / ((int64_t)W2 * H2));
        PoolBlock pb;
        pool_block_gather(dy, arg, b, i, j, H, W, OH, OW, C, col, pb);
        const float *px = x + (((int64_t)b * H + 2 * i) * W + 2 * j) * C + col;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (!pb.valid[p]) continue;
          const float4 v = *reinterpret_cast<const float4 *>(px + ((int64_t)(p >> 1) * W + (p & 1)) * C);
          const float X[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float dz = bn_pre(X[k], m[k], is[k], gm[k], bt[k]) > 0.f ? pb.g[p][k] : 0.f;
            const float y = (X[k] - m[k]) * is[k];
            s1[k] += dz;
            s2[k] += y * dz;
          }
        }
      }
    }
  }
  if (!red_finish(s1, s2, part, counters + cb, g, cb, rg, lds, fin, &ticket_s)) return;
  const int CW = g.CW4 * 4;
  if (t < CW) {
    const int c = cb * CW + t;
    const double cnt = (double)(blocks / ((int64_t)H2 * W2)) * H * W;      // B * H * W positions
    edz[c] = (float)(fin[(t >> 2) * 8 + (t & 3)] / cnt);
    eydz[c] = (float)(fin[(t >> 2) * 8 + 4 + (t & 3)] / cnt);
  }
}

// dx of the fused stem: a thread owns one (2 x 2 block, channel quad); dweight / dbias by the first C4 threads of workgroup 0
__global__ __launch_bounds__(kThreads) void abn_pool_grad_dx_nhwc_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const uint8_t *__restrict__ arg, const float *__restrict__ mean,
    const float *__restrict__ var, const float *__restrict__ weight, const float *__restrict__ bias, const float *__restrict__ edz,
    const float *__restrict__ eydz, float *__restrict__ dx, float *dweight, float *dbias, float eps, int64_t items, int H, int W,
    int OH, int OW, int H2, int W2, int C4, int accumulate, float norm) {
  const int64_t item = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (item >= items) return;
  const int q = (int)(item % C4);
  const int j = (int)((item / C4) % W2);
  const int i = (int)((item / ((int64_t)C4 * W2)) % H2);
  const int b = (int)(item / ((int64_t)C4 * W2 * H2));
  const int col = q * 4, C = C4 * 4;
  float m[4], is[4], gm[4], bt[4], e[4], ey[4], mul[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = mean[col + k];
    is[k] = inv_std_of(var[col + k], eps);
    gm[k] = gamma_of(weight, col + k, eps);
    bt[k] = beta_of(bias, col + k);
    e[k] = edz[col + k];
    ey[k] = eydz[col + k];
    mul[k] = gm[k] * is[k];
  }
  PoolBlock pb;
  pool_block_gather(dy, arg, b, i, j, H, W, OH, OW, C, col, pb);
  const int64_t o0 = (((int64_t)b * H + 2 * i) * W + 2 * j) * C + col;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (!pb.valid[p]) continue;
    const int64_t o = o0 + ((int64_t)(p >> 1) * W + (p & 1)) * C;
    const float4 v = *reinterpret_cast<const float4 *>(x + o);
    const float X[4] = {v.x, v.y, v.z, v.w};
    float D[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dz = bn_pre(X[k], m[k], is[k], gm[k], bt[k]) > 0.f ? pb.g[p][k] : 0.f;
      const float y = (X[k] - m[k]) * is[k];
      D[k] = (dz - e[k] - y * ey[k]) * mul[k];
    }
    *reinterpret_cast<float4 *>(dx + o) = make_float4(D[0], D[1], D[2], D[3]);
  }
  if (item < C4) {                                  // (item == q here: block 0, one thread per channel quad)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = col + k;
      if (dweight != nullptr) {                     // bn.cu:217-229 accumulates; accumulate == 0 writes
        const float wv = weight[c];
        const float gwt = wv > 0.f ? ey[k] * norm : (wv < 0.f ? -ey[k] * norm : 0.f);
        dweight[c] = accumulate ? dweight[c] + gwt : gwt;
      }
      if (dbias != nullptr) dbias[c] = accumulate ? dbias[c] + e[k] * norm : e[k] * norm;
    }
  }
}

constexpr int kStatsU = 8, kGrad0U = 8, kGrad1U = 4;

// Rows in flight per thread and trip (U): the tuned maximum for large tensors; halved while the launch would leave
// workgroup slots unused -- (8, 128, 65, 65) at U = 8 gives 134 workgroups for 256 CUs, at U = 4 it gives 256.
static int pick_u(int64_t rows, int C, int umax, RedGeom &g) {
  int u = umax;
  if (!make_red_geom(rows, C, u, g)) return 0;
  while (u > 2 && (int64_t)g.RG * g.CB < kRedMaxWG) {
    RedGeom h;
    if (!make_red_geom(rows, C, u / 2, h) || h.RG == g.RG) break;
    u /= 2;
    g = h;
  }
  return u;
}
#define SKD_RED_DISPATCH(U_, KERNEL_EXPR)                   \
  switch (U_) {                                             \
    case 8: { constexpr int UU = 8; KERNEL_EXPR; } break;   \
    case 4: { constexpr int UU = 4; KERNEL_EXPR; } break;   \
    default: { constexpr int UU = 2; KERNEL_EXPR; } break;  \
  }

static int launch_stats_nhwc2(int64_t rows, int C, const float *x, float *mean, float *var, float *running_mean,
                              float *running_var, float momentum, float *workspace, hipStream_t st) {
  RedGeom g;
  const int u = pick_u(rows, C, kStatsU, g);
  if (u == 0) return 0;
  unsigned *cnt = red_counters();
  if (cnt == nullptr) return 0;
  SKD_RED_DISPATCH(u, (abn_stats_nhwc2_kernel<UU><<<dim3((unsigned)(g.RG * g.CB)), dim3(kRedThreads), 0, st>>>(
      x, workspace, cnt, mean, var, running_mean, running_var, rows, g, momentum, (float)rows)));
  return ok();
}

template <bool HAS_RES>
static int launch_apply_nhwc_train(int act, const float *x, const float *res, float *out, const float *mean,
                                   const float *var, const float *weight, const float *bias, float eps, float slope,
                                   int64_t rows, const NhwcGeom &g, hipStream_t st) {
  const dim3 grid((unsigned)g.P), block(kThreads);
  switch (act) {
    case SKD_ACT_NONE:
      abn_apply_nhwc_train_kernel<SKD_ACT_NONE, HAS_RES><<<grid, block, 0, st>>>(x, res, out, mean, var, weight, bias, eps, slope, rows, g);
      break;
    case SKD_ACT_LEAKY_RELU:
      abn_apply_nhwc_train_kernel<SKD_ACT_LEAKY_RELU, HAS_RES><<<grid, block, 0, st>>>(x, res, out, mean, var, weight, bias, eps, slope, rows, g);
      break;
    case SKD_ACT_ELU:
      abn_apply_nhwc_train_kernel<SKD_ACT_ELU, HAS_RES><<<grid, block, 0, st>>>(x, res, out, mean, var, weight, bias, eps, slope, rows, g);
      break;
    case SKD_ACT_RELU:
      abn_apply_nhwc_train_kernel<SKD_ACT_RELU, HAS_RES><<<grid, block, 0, st>>>(x, res, out, mean, var, weight, bias, eps, slope, rows, g);
      break;
    default:
      return 0;
  }
  return ok();
}

// SKD_ABN_FUSED=0 keeps the two-launch passes -- the operational switch for a device that is shared with another grid-barrier
// launch (multi-tenant GPU); default: the register-resident one-launch passes when they fit.
// SKD_ABN_SYNC_FUSED=0: the synchronised entries keep the three-launch form (statistics, exchange kernel, normalise).
// Both are LIBRARY STATE (round 6): the environment is read ONCE, at the first query (getenv from autograd threads
// while the host language calls setenv is not safe under glibc, and a process-wide variable is the wrong place for a default that
// depends on the process group); skd_abn_set_fused / skd_abn_set_sync_fused change it afterwards (tests, utils/parallel.py's RCCL
// default, bench.py's fallback chain), -1 = read the environment again at the next query.
constexpr int kFuseFwdMaxNR = 17, kFuseBwdMaxNR = 9;
static std::atomic<int> g_fused_state{-1}, g_sync_fused_state{-1};
static bool switch_state(std::atomic<int> &state, const char *name) {
  int v = state.load(std::memory_order_relaxed);
  if (v < 0) {
    const char *e = getenv(name);
    v = !(e != nullptr && e[0] == '0');
    state.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}
static bool fused_enabled() { return switch_state(g_fused_state, "SKD_ABN_FUSED"); }
static bool sync_fused_enabled() { return switch_state(g_sync_fused_state, "SKD_ABN_SYNC_FUSED"); }

// The grid barrier of the one-launch passes needs every workgroup of the launch co-resident: the grid is
// capped by what THIS device can hold at one 1024-thread workgroup per compute unit -- 256 on a whole MI355X, 32 on a
// CPX partition -- queried once per device, never assumed (a CU MASK is not visible in that attribute: see below).  skd_abn_set_fused_max_workgroups()
// lowers it further (ranks that share one device must share its compute units: utils/parallel.py does that).  A cap too
// small for a tensor (rows per thread > NR) simply sends that call to the two-launch path.
struct FuseCap {
  int cap[64];
  bool known[64];
};
static FuseCap g_fuse_cap = {};
static int g_fuse_user_cap = kRedMaxWG;

static int fuse_wg_cap() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (!g_fuse_cap.known[dev]) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      (void)hipGetLastError();
      cus = 0;
    }
    const char *e = getenv("SKD_ABN_FUSED_MAXWG");
    int env_cap = kRedMaxWG;
    if (e != nullptr && e[0] != 0) env_cap = atoi(e);
    int cap = cus < kRedMaxWG ? cus : kRedMaxWG;
    if (env_cap < cap) cap = env_cap;
    // A CU mask is NOT reflected in hipDeviceAttributeMultiprocessorCount: a masked process that trusted the
    // attribute would launch a grid barrier that cannot be co-resident and sit in it until the time limit.  Parsing the mask
    // formats is not this library's business: with a mask in the environment the one-launch passes are OFF (two-launch path)
    // unless the user states the usable compute units with SKD_ABN_FUSED_MAXWG.
    const char *m1 = getenv("HSA_CU_MASK"), *m2 = getenv("ROC_GLOBAL_CU_MASK");
    if (((m1 != nullptr && m1[0] != 0) || (m2 != nullptr && m2[0] != 0)) && !(e != nullptr && e[0] != 0)) cap = 0;
    g_fuse_cap.cap[dev] = cap > 0 ? cap : 0;
    g_fuse_cap.known[dev] = true;
  }
  const int cap = g_fuse_cap.cap[dev];
  return cap < g_fuse_user_cap ? cap : g_fuse_user_cap;
}

// one workgroup per compute unit must actually be launchable for THIS instantiation (registers, LDS): checked once per
// instantiation and device with the occupancy query, not assumed
template <class K>
static bool fuse_kernel_fits(K kernel, PerDeviceFlag &checked, PerDeviceFlag &fits) {
  bool *c = checked.get(), *f = fits.get();
  if (c == nullptr || f == nullptr) return false;
  if (!*c) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void *>(kernel), kRedThreads, 0) != hipSuccess) {
      (void)hipGetLastError();
      blocks = 0;
    }
    *f = blocks >= 1;
    *c = true;
  }
  return *f;
}
#define SKD_FUSE_LAUNCH(KERNEL, ...)                                                        \
  do {                                                                                      \
    static PerDeviceFlag checked_, fits_;                                                   \
    if (!fuse_kernel_fits(KERNEL, checked_, fits_)) return -1;                              \
    KERNEL<<<grid, block, 0, st>>>(__VA_ARGS__);                                            \
  } while (0)

// ---- can THIS device launch every synchronised one-launch instantiation?   -------------------------------------
// SKD_FUSE_LAUNCH checks its instantiation when it is about to launch -- for the synchronised entries that is AFTER sync_next()
// has drawn the exchange's sequence number, when "take the other form" is no longer possible: the rank would have to fail while
// its peers spin for it.  So the synchronised entries ask this BEFORE they draw: one occupancy query per instantiation, once per
// device; if any of them cannot hold a 1024-thread workgroup on a compute unit (fewer registers / LDS than gfx950), the device
// takes the three-launch form for every synchronised call (the two forms interoperate rank by rank).
template <class K>
static bool one_block_fits(K kernel) {
  int blocks = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void *>(kernel), kRedThreads, 0) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return blocks >= 1;
}
template <int ACT, bool RES>
static bool fwd_sync_fit() {
  return one_block_fits(abn_fwd_fused_nhwc_kernel<ACT, RES, 5, true>) && one_block_fits(abn_fwd_fused_nhwc_kernel<ACT, RES, 9, true>) &&
         one_block_fits(abn_fwd_fused_nhwc_kernel<ACT, RES, 17, true>);
}
template <int ACT, int MODE, bool WRITE_RES>
static bool bwd_sync_fit() {
  return one_block_fits(abn_bwd_fused_nhwc_kernel<ACT, MODE, WRITE_RES, 5, true>) &&
         one_block_fits(abn_bwd_fused_nhwc_kernel<ACT, MODE, WRITE_RES, 9, true>);
}
static bool sync_fused_fits_device() {
  static PerDeviceFlag checked, fits;
  bool *c = checked.get(), *f = fits.get();
  if (c == nullptr || f == nullptr) return false;
  if (!*c) {
    *f = fwd_sync_fit<SKD_ACT_RELU, true>() && fwd_sync_fit<SKD_ACT_RELU, false>() && fwd_sync_fit<SKD_ACT_LEAKY_RELU, false>() &&
         fwd_sync_fit<SKD_ACT_NONE, false>() && bwd_sync_fit<SKD_ACT_LEAKY_RELU, 0, false>() && bwd_sync_fit<SKD_ACT_NONE, 0, false>() &&
         bwd_sync_fit<SKD_ACT_NONE, 2, false>() && bwd_sync_fit<SKD_ACT_NONE, 1, true>() && bwd_sync_fit<SKD_ACT_NONE, 1, false>();
    *c = true;
  }
  return *f;
}

// how often the synchronised entries took each form (tests and bench.py report it: "did the exchange really run inside the kernel?")
static int64_t g_sync_form_calls[2] = {0, 0};     // [one launch with the exchange inside, statistics + exchange kernel + normalise]

static SyncArgs no_sync() {
  SyncArgs a = {};
  a.status = status_words();
  return a;
}

// returns -1 when the call does not take the fused path (not enabled, activation / size not covered), else ok().
// sy != nullptr: the cross-replica exchange happens inside the launch (the caller has drawn sy with sync_next()).
template <bool HAS_RES, bool SYNC>
static int launch_fwd_fused(int act, int64_t rows, int C, const float *x, const float *res, float *out, float *mean,
                            float *var, float *running_mean, float *running_var, const float *weight, const float *bias,
                            float momentum, float eps, float slope, float *workspace, hipStream_t st, const FuseGeom &f,
                            const SyncArgs &sy, const float *rweights, float n_pooled) {
  unsigned *cnt = red_counters();
  if (cnt == nullptr) return 0;
  const dim3 grid((unsigned)(f.r.RG * f.r.CB)), block(kRedThreads);
#define SKD_FWD_FUSED(ACT_, NR_)                                                                                              \
  SKD_FUSE_LAUNCH((abn_fwd_fused_nhwc_kernel<ACT_, HAS_RES, NR_, SYNC>), x, res, out, workspace, cnt, red_gens(cnt), mean, var, \
                  running_mean, running_var, weight, bias, rows, f.r, f.nr, momentum, eps, slope, sy, rweights, n_pooled)
#define SKD_FWD_FUSED_NR(ACT_)                 \
  if (f.nr <= 5) SKD_FWD_FUSED(ACT_, 5);       \
  else if (f.nr <= 9) SKD_FWD_FUSED(ACT_, 9);  \
  else SKD_FWD_FUSED(ACT_, 17)
  if (act == SKD_ACT_RELU) {
    SKD_FWD_FUSED_NR(SKD_ACT_RELU);
  } else if (!HAS_RES && act == SKD_ACT_LEAKY_RELU) {
    SKD_FWD_FUSED_NR(SKD_ACT_LEAKY_RELU);
  } else if (!HAS_RES && act == SKD_ACT_NONE) {
    SKD_FWD_FUSED_NR(SKD_ACT_NONE);
  } else {
    return -1;
  }
#undef SKD_FWD_FUSED_NR
#undef SKD_FWD_FUSED
  return ok();
}

// does this forward take the one-launch form?  (decided BEFORE a sequence number is drawn for the exchange)
static bool fwd_fused_geom(int act, bool has_res, int64_t rows, int C, FuseGeom &f) {
  if (!fused_enabled() || act == SKD_ACT_ELU || (has_res && act != SKD_ACT_RELU)) return false;
  if (act != SKD_ACT_RELU && act != SKD_ACT_LEAKY_RELU && act != SKD_ACT_NONE) return false;
  return make_fuse_geom(rows, C, kFuseFwdMaxNR, fuse_wg_cap(), f);
}
static bool bwd_fused_geom(int64_t rows, int C, FuseGeom &f) {
  return fused_enabled() && make_fuse_geom(rows, C, kFuseBwdMaxNR, fuse_wg_cap(), f);
}

// MODE 0: (z, dz) of the in-place ABN; MODE 1: (x, out, dout); MODE 2: (x, dout).  -1 = not taken.
template <int ACT, int MODE, bool WRITE_RES, bool SYNC>
static int launch_bwd_fused(int64_t rows, int C, const float *a, const float *b, const float *c, const float *mean,
                            const float *var, const float *weight, const float *bias, float *edz, float *eydz, float *dx,
                            float *dres, float *dweight, float *dbias, float eps, float slope, int accumulate,
                            float *workspace, hipStream_t st, const FuseGeom &f, const SyncArgs &sy, const float *rweights) {
  unsigned *cnt = red_counters();
  if (cnt == nullptr) return 0;
  const dim3 grid((unsigned)(f.r.RG * f.r.CB)), block(kRedThreads);
  // instantiations sized to the step's layers (5 rows per thread at C = 128, 9 at C = 64 / 256): one row more and the 18 registers
  // per row (y, dz) of the NR = 10 form spilled 11-13 of them (PMC: 14.3 instead of 12 bytes per element)
  if (f.nr <= 5)
    SKD_FUSE_LAUNCH((abn_bwd_fused_nhwc_kernel<ACT, MODE, WRITE_RES, 5, SYNC>), a, b, c, mean, var, weight, bias, workspace, cnt,
                    red_gens(cnt), edz, eydz, dx, dres, dweight, dbias, eps, slope, rows, f.r, f.nr, accumulate, sy, rweights);
  else
    SKD_FUSE_LAUNCH((abn_bwd_fused_nhwc_kernel<ACT, MODE, WRITE_RES, 9, SYNC>), a, b, c, mean, var, weight, bias, workspace, cnt,
                    red_gens(cnt), edz, eydz, dx, dres, dweight, dbias, eps, slope, rows, f.r, f.nr, accumulate, sy, rweights);
  return ok();
}

static int valid_dims(int N, int C, int S) { return N > 0 && C > 0 && S > 0; }

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

int64_t skd_abn_workspace_floats(int N, int C, int S) {
  if (!valid_dims(N, C, S)) return 0;
  const Plan pl = make_plan(N, C, S);
  return (int64_t)pl.P * C * 2;
}

int skd_abn_stats(int N, int C, int S, const float *x, float *mean, float *var, float *workspace,
                  skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !x || !mean || !var || !workspace) return 0;
  const Plan pl = make_plan(N, C, S);
  hipStream_t st = as_stream(stream);
  abn_stats_partial_kernel<<<dim3((unsigned)pl.items), dim3(kThreads), 0, st>>>(x, workspace, N, C, S, pl);
  abn_stats_finalize_kernel<<<dim3((unsigned)cdiv(C, kWavesPerWG)), dim3(kThreads), 0, st>>>(
      x, workspace, mean, var, nullptr, nullptr, N, C, S, pl.P, 0.f, 0.0);
  return ok();
}

int skd_abn_combine_stats(int G, int C, const float *gathered, const float *weights, int rank, float *mean, float *var,
                          float *running_mean, float *running_var, float momentum, double n, skd_stream_t stream) {
  if (G <= 0 || C <= 0 || !gathered || !mean || !var || (weights && (rank < 0 || rank >= G))) return 0;
  abn_combine_stats_kernel<<<dim3((unsigned)cdiv(C, 256)), dim3(256), 0, as_stream(stream)>>>(
      G, C, gathered, weights, rank, mean, var, running_mean, running_var, momentum, (float)n);
  return ok();
}

int skd_abn_update_running(int C, float *running_mean, float *running_var, const float *mean,
                           const float *var, float momentum, double n, skd_stream_t stream) {
  if (C <= 0 || !running_mean || !running_var || !mean || !var) return 0;
  abn_update_running_kernel<<<dim3((unsigned)cdiv(C, 256)), dim3(256), 0, as_stream(stream)>>>(
      C, running_mean, running_var, mean, var, momentum, (float)n);
  return ok();
}

int skd_abn_apply(int N, int C, int S, float *x, const float *mean, const float *var,
                  const float *weight, const float *bias, float eps, int activation, float slope,
                  skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !x || !mean || !var) return 0;
  const Plan pl = make_plan(N, C, S);
  launch_apply<false>(activation, pl, as_stream(stream), x, mean, var, weight, bias, x, x, eps, slope, N, C, S, 0);
  return ok();
}

int skd_abn_apply_residual(int N, int C, int S, float *x, const float *residual, const float *mean,
                           const float *var, const float *weight, const float *bias, float eps,
                           int activation, float slope, skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !x || !residual || !mean || !var) return 0;
  if (!same_phase(x, residual)) return 0;
  const Plan pl = make_plan(N, C, S);
  const dim3 grid((unsigned)pl.items), block(kThreads);
  hipStream_t st = as_stream(stream);
  switch (activation) {
    case SKD_ACT_LEAKY_RELU:
      abn_apply_residual_kernel<SKD_ACT_LEAKY_RELU><<<grid, block, 0, st>>>(x, residual, mean, var, weight, bias, x, eps, slope, N, C, S, pl, apply_nt_mode());
      break;
    case SKD_ACT_RELU:
      abn_apply_residual_kernel<SKD_ACT_RELU><<<grid, block, 0, st>>>(x, residual, mean, var, weight, bias, x, eps, slope, N, C, S, pl, apply_nt_mode());
      break;
    case SKD_ACT_NONE:
      abn_apply_residual_kernel<SKD_ACT_NONE><<<grid, block, 0, st>>>(x, residual, mean, var, weight, bias, x, eps, slope, N, C, S, pl, apply_nt_mode());
      break;
    default:
      return 0;
  }
  return ok();
}

int skd_abn_forward_train(int N, int C, int S, float *x, const float *weight, const float *bias,
                          float *running_mean, float *running_var, float *mean, float *var,
                          float momentum, float eps, int activation, float slope, float *workspace,
                          skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !x || !mean || !var || !workspace) return 0;
  const Plan pl = make_plan(N, C, S);
  hipStream_t st = as_stream(stream);
  abn_stats_partial_kernel<<<dim3((unsigned)pl.items), dim3(kThreads), 0, st>>>(x, workspace, N, C, S, pl);
  abn_stats_finalize_kernel<<<dim3((unsigned)cdiv(C, kWavesPerWG)), dim3(kThreads), 0, st>>>(
      x, workspace, mean, var, running_mean, running_var, N, C, S, pl.P, momentum,
      (double)N * (double)S);
  launch_apply<false>(activation, pl, st, x, mean, var, weight, bias, x, x, eps, slope, N, C, S, 1);
  return ok();
}

int skd_abn_backward_reduce(int N, int C, int S, const float *z, const float *dz, const float *weight,
                            const float *bias, float *edz, float *eydz, float eps, int activation,
                            float slope, float *workspace, skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !z || !dz || !edz || !eydz || !workspace) return 0;
  if (activation == SKD_ACT_RELU) return 0;  // not invertible from the output: forward-only activation
  if (!same_phase(z, dz)) return 0;
  const Plan pl = make_plan(N, C, S);
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)pl.items), block(kThreads);
  switch (activation) {
    case SKD_ACT_LEAKY_RELU:
      abn_grad_partial_kernel<SKD_ACT_LEAKY_RELU><<<grid, block, 0, st>>>(z, dz, weight, bias, workspace, eps, slope, N, C, S, pl);
      break;
    case SKD_ACT_ELU:
      abn_grad_partial_kernel<SKD_ACT_ELU><<<grid, block, 0, st>>>(z, dz, weight, bias, workspace, eps, slope, N, C, S, pl);
      break;
    default:
      abn_grad_partial_kernel<SKD_ACT_NONE><<<grid, block, 0, st>>>(z, dz, weight, bias, workspace, eps, slope, N, C, S, pl);
  }
  abn_grad_finalize_kernel<<<dim3((unsigned)cdiv(C, kWavesPerWG)), dim3(kThreads), 0, st>>>(
      workspace, edz, eydz, N, C, S, pl.P);
  return ok();
}

int skd_abn_backward_dx(int N, int C, int S, const float *z, const float *dz, const float *var,
                        const float *weight, const float *bias, const float *edz, const float *eydz,
                        float *dx, float *dweight, float *dbias, float eps, int activation,
                        float slope, skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !z || !dz || !var || !edz || !eydz) return 0;
  if (activation == SKD_ACT_RELU) return 0;
  if (dweight && !weight) return 0;
  if (!same_phase(z, dz) || (dx && !same_phase(z, dx))) return 0;
  const Plan pl = make_plan(N, C, S);
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)pl.items), block(kThreads);
  switch (activation) {
    case SKD_ACT_LEAKY_RELU:
      abn_grad_dx_kernel<SKD_ACT_LEAKY_RELU><<<grid, block, 0, st>>>(z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, slope, N, C, S, pl, 1, apply_nt_mode());
      break;
    case SKD_ACT_ELU:
      abn_grad_dx_kernel<SKD_ACT_ELU><<<grid, block, 0, st>>>(z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, slope, N, C, S, pl, 1, apply_nt_mode());
      break;
    default:
      abn_grad_dx_kernel<SKD_ACT_NONE><<<grid, block, 0, st>>>(z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, slope, N, C, S, pl, 1, apply_nt_mode());
  }
  return ok();
}

int skd_abn_backward(int N, int C, int S, const float *z, const float *dz, const float *var,
                     const float *weight, const float *bias, float *edz, float *eydz, float *dx,
                     float *dweight, float *dbias, float eps, int activation, float slope,
                     int training, float *workspace, skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !edz || !eydz) return 0;
  if (training) {
    if (!skd_abn_backward_reduce(N, C, S, z, dz, weight, bias, edz, eydz, eps, activation, slope,
                                 workspace, stream))
      return 0;
  } else {
    // functions.py:146-147: inference-mode backward uses edz = eydz = 0
    if (hipMemsetAsync(edz, 0, sizeof(float) * C, as_stream(stream)) != hipSuccess) return 0;
    if (hipMemsetAsync(eydz, 0, sizeof(float) * C, as_stream(stream)) != hipSuccess) return 0;
  }
  return skd_abn_backward_dx(N, C, S, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps,
                             activation, slope, stream);
}



int skd_abn_apply_nhwc(int64_t rows, int C, float *x, const float *residual, const float *mean, const float *var,
                       const float *weight, const float *bias, float eps, int activation, float slope,
                       skd_stream_t stream) {
  if (rows <= 0 || C <= 0 || (C & 3) || !x || !mean || !var) return 0;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (residual && (reinterpret_cast<uintptr_t>(residual) & 15))) return 0;
  hipStream_t st = as_stream(stream);
  return residual ? launch_apply_nhwc<true>(rows, C, x, residual, mean, var, weight, bias, eps, activation, slope, st)
                  : launch_apply_nhwc<false>(rows, C, x, residual, mean, var, weight, bias, eps, activation, slope, st);
}

// ---- out-of-place BN -> (+residual) -> activation (the training-time ReLU fusion) ------------------------

static int launch_apply_to(int N, int C, int S, const float *x, const float *residual, float *out,
                           const float *mean, const float *var, const float *weight, const float *bias, float eps,
                           int activation, float slope, hipStream_t st, int reverse) {
  const Plan pl = make_plan(N, C, S);
  if (residual == nullptr) {
    launch_apply<false>(activation, pl, st, x, mean, var, weight, bias, out, out, eps, slope, N, C, S, reverse);
    return ok();
  }
  const dim3 grid((unsigned)pl.items), block(kThreads);
  switch (activation) {
    case SKD_ACT_LEAKY_RELU:
      abn_apply_residual_kernel<SKD_ACT_LEAKY_RELU><<<grid, block, 0, st>>>(x, residual, mean, var, weight, bias, out, eps, slope, N, C, S, pl, apply_nt_mode());
      break;
    case SKD_ACT_RELU:
      abn_apply_residual_kernel<SKD_ACT_RELU><<<grid, block, 0, st>>>(x, residual, mean, var, weight, bias, out, eps, slope, N, C, S, pl, apply_nt_mode());
      break;
    case SKD_ACT_NONE:
      abn_apply_residual_kernel<SKD_ACT_NONE><<<grid, block, 0, st>>>(x, residual, mean, var, weight, bias, out, eps, slope, N, C, S, pl, apply_nt_mode());
      break;
    default:
      return 0;
  }
  return ok();
}

int skd_abn_apply_to(int N, int C, int S, const float *x, const float *residual, float *out, const float *mean,
                     const float *var, const float *weight, const float *bias, float eps, int activation,
                     float slope, skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !x || !out || !mean || !var) return 0;
  if (!same_phase(x, out) || (residual && !same_phase(x, residual))) return 0;
  return launch_apply_to(N, C, S, x, residual, out, mean, var, weight, bias, eps, activation, slope,
                         as_stream(stream), 0);
}

int skd_abn_forward_train_to(int N, int C, int S, const float *x, const float *residual, float *out,
                             const float *weight, const float *bias, float *running_mean, float *running_var,
                             float *mean, float *var, float momentum, float eps, int activation, float slope,
                             float *workspace, skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !x || !out || !mean || !var || !workspace) return 0;
  if (!same_phase(x, out) || (residual && !same_phase(x, residual))) return 0;
  const Plan pl = make_plan(N, C, S);
  hipStream_t st = as_stream(stream);
  abn_stats_partial_kernel<<<dim3((unsigned)pl.items), dim3(kThreads), 0, st>>>(x, workspace, N, C, S, pl);
  abn_stats_finalize_kernel<<<dim3((unsigned)cdiv(C, kWavesPerWG)), dim3(kThreads), 0, st>>>(
      x, workspace, mean, var, running_mean, running_var, N, C, S, pl.P, momentum, (double)N * (double)S);
  return launch_apply_to(N, C, S, x, residual, out, mean, var, weight, bias, eps, activation, slope, st,
                         residual == nullptr ? 1 : 0);
}

int skd_abn_relu_backward_reduce(int N, int C, int S, const float *x, const float *out, const float *dout,
                                 const float *mean, const float *var, float *edz, float *eydz, float eps,
                                 float *workspace, skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !x || !out || !dout || !mean || !var || !edz || !eydz || !workspace) return 0;
  if (!same_phase(x, out) || !same_phase(x, dout)) return 0;
  const Plan pl = make_plan(N, C, S);
  hipStream_t st = as_stream(stream);
  abn_relu_grad_partial_kernel<<<dim3((unsigned)pl.items), dim3(kThreads), 0, st>>>(x, out, dout, mean, var,
                                                                                   workspace, eps, N, C, S, pl);
  abn_grad_finalize_kernel<<<dim3((unsigned)cdiv(C, kWavesPerWG)), dim3(kThreads), 0, st>>>(workspace, edz, eydz,
                                                                                           N, C, S, pl.P);
  return ok();
}

int skd_abn_relu_backward_dx(int N, int C, int S, const float *x, const float *out, const float *dout,
                             const float *mean, const float *var, const float *weight, const float *edz,
                             const float *eydz, float *dx, float *dres, float *dweight, float *dbias, float eps,
                             skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !x || !out || !dout || !mean || !var || !edz || !eydz || !dx) return 0;
  if (dweight && !weight) return 0;
  if (!same_phase(x, out) || !same_phase(x, dout) || !same_phase(x, dx) || (dres && !same_phase(x, dres))) return 0;
  const Plan pl = make_plan(N, C, S);
  const dim3 grid((unsigned)pl.items), block(kThreads);
  hipStream_t st = as_stream(stream);
  if (dres != nullptr)
    abn_relu_grad_dx_kernel<true><<<grid, block, 0, st>>>(x, out, dout, mean, var, weight, edz, eydz, dx, dres, dweight, dbias, eps, N, C, S, pl, apply_nt_mode());
  else
    abn_relu_grad_dx_kernel<false><<<grid, block, 0, st>>>(x, out, dout, mean, var, weight, edz, eydz, dx, dres, dweight, dbias, eps, N, C, S, pl, apply_nt_mode());
  return ok();
}


// ---- channels-last (NHWC) training entries ---------------------------------------------------------------

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int64_t skd_abn_nhwc_workspace_floats(int64_t rows, int C) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g)) return 0;
  // workgroup-major partial rows of the one-launch reductions: CB * RG <= kRedMaxWG rows of 2 * C / CB floats, then 2 * C floats
  // for this replica's own [mean | var] in the synchronised entries' three-launch form
  return (int64_t)kRedMaxWG * 2 * C + 2 * (int64_t)C;
}

int skd_abn_stats_nhwc(int64_t rows, int C, const float *x, float *mean, float *var, float *workspace,
                       skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || !x || !mean || !var || !workspace || !aligned16(x)) return 0;
  return launch_stats_nhwc2(rows, C, x, mean, var, nullptr, nullptr, 0.f, workspace, as_stream(stream));
}

int skd_abn_apply_nhwc_to(int64_t rows, int C, const float *x, const float *residual, float *out,
                          const float *mean, const float *var, const float *weight, const float *bias, float eps,
                          int activation, float slope, skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || !x || !out || !mean || !var) return 0;
  if (!aligned16(x) || !aligned16(out) || (residual && !aligned16(residual))) return 0;
  hipStream_t st = as_stream(stream);
  return residual ? launch_apply_nhwc_train<true>(activation, x, residual, out, mean, var, weight, bias, eps, slope, rows, g, st)
                  : launch_apply_nhwc_train<false>(activation, x, residual, out, mean, var, weight, bias, eps, slope, rows, g, st);
}

int skd_abn_forward_train_nhwc(int64_t rows, int C, const float *x, const float *residual, float *out,
                               const float *weight, const float *bias, float *running_mean, float *running_var,
                               float *mean, float *var, float momentum, float eps, int activation, float slope,
                               float *workspace, skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || rows > 2147483647 || !x || !out || !mean || !var || !workspace) return 0;
  if (!aligned16(x) || !aligned16(out) || (residual && !aligned16(residual))) return 0;
  hipStream_t st = as_stream(stream);
  FuseGeom f;
  if (aligned16(mean) && aligned16(var) && fwd_fused_geom(activation, residual != nullptr, rows, C, f)) {
    const SyncArgs sy = no_sync();
    const int r = residual ? launch_fwd_fused<true, false>(activation, rows, C, x, residual, out, mean, var, running_mean, running_var,
                                                           weight, bias, momentum, eps, slope, workspace, st, f, sy, nullptr, 0.f)
                           : launch_fwd_fused<false, false>(activation, rows, C, x, residual, out, mean, var, running_mean, running_var,
                                                            weight, bias, momentum, eps, slope, workspace, st, f, sy, nullptr, 0.f);
    if (r >= 0) return r;
  }
  if (!launch_stats_nhwc2(rows, C, x, mean, var, running_mean, running_var, momentum, workspace, st)) return 0;
  return residual ? launch_apply_nhwc_train<true>(activation, x, residual, out, mean, var, weight, bias, eps, slope, rows, g, st)
                  : launch_apply_nhwc_train<false>(activation, x, residual, out, mean, var, weight, bias, eps, slope, rows, g, st);
}

// InPlaceABNSync forward (libs/functions.py:165-218) for a channels-last tensor in ONE call: the arguments of
// skd_abn_forward_train_nhwc plus the mailbox context (skd_sync_create), the per-replica sample weights (NULL = equal
// shards) and n (the pooled sample count without weights, this replica's with: skd_abn_combine_stats).  One
// register-resident launch with the exchange inside it when the tensor fits; else statistics -> skd_abn_sync_stats ->
// normalise.  Either way ONE exchange (one sequence number) in the same place of every rank's call order.
int skd_abn_forward_train_nhwc_sync(void *sync_ctx, int64_t rows, int C, const float *x, const float *residual, float *out,
                                    const float *weight, const float *bias, float *running_mean, float *running_var,
                                    float *mean, float *var, const float *replica_weights, float momentum, float eps,
                                    int activation, float slope, double n, float *workspace, skd_stream_t stream) {
  NhwcGeom g;
  if (!sync_ctx || !make_nhwc_geom(rows, C, g) || rows > 2147483647 || !x || !out || !mean || !var || !workspace) return 0;
  if (!aligned16(x) || !aligned16(out) || (residual && !aligned16(residual)) || 2 * C > kSyncMaxFloats) return 0;
  hipStream_t st = as_stream(stream);
  FuseGeom f;
  if (sync_fused_enabled() && sync_fused_fits_device() && aligned16(mean) && aligned16(var) &&
      fwd_fused_geom(activation, residual != nullptr, rows, C, f)) {
    SyncArgs sy;
    if (!sync_next(sync_ctx, sy)) return 0;
    ++g_sync_form_calls[0];
    const int r = residual ? launch_fwd_fused<true, true>(activation, rows, C, x, residual, out, mean, var, running_mean, running_var,
                                                          weight, bias, momentum, eps, slope, workspace, st, f, sy, replica_weights, (float)n)
                           : launch_fwd_fused<false, true>(activation, rows, C, x, residual, out, mean, var, running_mean, running_var,
                                                           weight, bias, momentum, eps, slope, workspace, st, f, sy, replica_weights, (float)n);
    return r > 0 ? 1 : 0;          // (a sequence number has been drawn: "not taken" is no longer an option)
  }
  ++g_sync_form_calls[1];
  float *local = workspace + (int64_t)kRedMaxWG * 2 * C;       // this replica's [mean | var]
  if (!launch_stats_nhwc2(rows, C, x, local, local + C, nullptr, nullptr, 0.f, workspace, st)) return 0;
  if (!skd_abn_sync_stats(sync_ctx, C, local, replica_weights, mean, var, running_mean, running_var, momentum, n, stream)) return 0;
  return residual ? launch_apply_nhwc_train<true>(activation, x, residual, out, mean, var, weight, bias, eps, slope, rows, g, st)
                  : launch_apply_nhwc_train<false>(activation, x, residual, out, mean, var, weight, bias, eps, slope, rows, g, st);
}

int skd_abn_backward_reduce_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *weight,
                                 const float *bias, float *edz, float *eydz, float eps, int activation, float slope,
                                 float *workspace, skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || rows > 2147483647 || !z || !dz || !edz || !eydz || !workspace) return 0;
  if (!aligned16(z) || !aligned16(dz) || activation == SKD_ACT_RELU) return 0;
  hipStream_t st = as_stream(stream);
  RedGeom rg;
  const int u = pick_u(rows, C, kGrad0U, rg);
  if (u == 0) return 0;
  unsigned *cnt = red_counters();
  if (cnt == nullptr) return 0;
  const dim3 grid((unsigned)(rg.RG * rg.CB)), block(kRedThreads);
  switch (activation) {
    case SKD_ACT_LEAKY_RELU:
      SKD_RED_DISPATCH(u, (abn_grad_nhwc2_kernel<SKD_ACT_LEAKY_RELU, 0, UU><<<grid, block, 0, st>>>(z, dz, nullptr, nullptr, nullptr, weight, bias, workspace, cnt, edz, eydz, eps, slope, rows, rg)));
      break;
    case SKD_ACT_ELU:
      SKD_RED_DISPATCH(u, (abn_grad_nhwc2_kernel<SKD_ACT_ELU, 0, UU><<<grid, block, 0, st>>>(z, dz, nullptr, nullptr, nullptr, weight, bias, workspace, cnt, edz, eydz, eps, slope, rows, rg)));
      break;
    default:
      SKD_RED_DISPATCH(u, (abn_grad_nhwc2_kernel<SKD_ACT_NONE, 0, UU><<<grid, block, 0, st>>>(z, dz, nullptr, nullptr, nullptr, weight, bias, workspace, cnt, edz, eydz, eps, slope, rows, rg)));
  }
  return ok();
}

int skd_abn_backward_dx_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *var,
                             const float *weight, const float *bias, const float *edz, const float *eydz, float *dx,
                             float *dweight, float *dbias, float eps, int activation, float slope, int accumulate,
                             skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || !z || !dz || !var || !edz || !eydz || !dx) return 0;
  if (!aligned16(z) || !aligned16(dz) || !aligned16(dx) || activation == SKD_ACT_RELU || (dweight && !weight)) return 0;
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)g.P), block(kThreads);
  switch (activation) {
    case SKD_ACT_LEAKY_RELU:
      abn_grad_dx_nhwc_kernel<SKD_ACT_LEAKY_RELU, 0, false><<<grid, block, 0, st>>>(z, dz, nullptr, nullptr, var, weight, bias, edz, eydz, dx, nullptr, dweight, dbias, eps, slope, rows, g, accumulate);
      break;
    case SKD_ACT_ELU:
      abn_grad_dx_nhwc_kernel<SKD_ACT_ELU, 0, false><<<grid, block, 0, st>>>(z, dz, nullptr, nullptr, var, weight, bias, edz, eydz, dx, nullptr, dweight, dbias, eps, slope, rows, g, accumulate);
      break;
    default:
      abn_grad_dx_nhwc_kernel<SKD_ACT_NONE, 0, false><<<grid, block, 0, st>>>(z, dz, nullptr, nullptr, var, weight, bias, edz, eydz, dx, nullptr, dweight, dbias, eps, slope, rows, g, accumulate);
  }
  return ok();
}

int skd_abn_relu_backward_reduce_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout,
                                      const float *mean, const float *var, float *edz, float *eydz, float eps,
                                      float *workspace, skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || rows > 2147483647 || !x || !out || !dout || !mean || !var || !edz || !eydz || !workspace) return 0;
  if (!aligned16(x) || !aligned16(out) || !aligned16(dout)) return 0;
  hipStream_t st = as_stream(stream);
  RedGeom rg;
  const int u = pick_u(rows, C, kGrad1U, rg);
  if (u == 0) return 0;
  unsigned *cnt = red_counters();
  if (cnt == nullptr) return 0;
  if (u == 4)
    abn_grad_nhwc2_kernel<SKD_ACT_NONE, 1, 4><<<dim3((unsigned)(rg.RG * rg.CB)), dim3(kRedThreads), 0, st>>>(
        x, out, dout, mean, var, nullptr, nullptr, workspace, cnt, edz, eydz, eps, 0.f, rows, rg);
  else
    abn_grad_nhwc2_kernel<SKD_ACT_NONE, 1, 2><<<dim3((unsigned)(rg.RG * rg.CB)), dim3(kRedThreads), 0, st>>>(
        x, out, dout, mean, var, nullptr, nullptr, workspace, cnt, edz, eydz, eps, 0.f, rows, rg);
  return ok();
}

int skd_abn_relu_backward_reduce_nhwc_x(int64_t rows, int C, const float *x, const float *dout, const float *mean,
                                        const float *var, const float *weight, const float *bias, float *edz, float *eydz,
                                        float eps, float *workspace, skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || rows > 2147483647 || !x || !dout || !mean || !var || !edz || !eydz || !workspace) return 0;
  if (!aligned16(x) || !aligned16(dout)) return 0;
  hipStream_t st = as_stream(stream);
  RedGeom rg;
  const int u = pick_u(rows, C, kGrad0U, rg);
  if (u == 0) return 0;
  unsigned *cnt = red_counters();
  if (cnt == nullptr) return 0;
  SKD_RED_DISPATCH(u, (abn_grad_nhwc2_kernel<SKD_ACT_NONE, 2, UU><<<dim3((unsigned)(rg.RG * rg.CB)), dim3(kRedThreads), 0, st>>>(
      x, dout, nullptr, mean, var, weight, bias, workspace, cnt, edz, eydz, eps, 0.f, rows, rg)));
  return ok();
}

int skd_abn_relu_backward_dx_nhwc_x(int64_t rows, int C, const float *x, const float *dout, const float *mean,
                                    const float *var, const float *weight, const float *bias, const float *edz,
                                    const float *eydz, float *dx, float *dweight, float *dbias, float eps, int accumulate,
                                    skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || !x || !dout || !mean || !var || !edz || !eydz || !dx) return 0;
  if (!aligned16(x) || !aligned16(dout) || !aligned16(dx) || (dweight && !weight)) return 0;
  abn_grad_dx_nhwc_kernel<SKD_ACT_NONE, 2, false><<<dim3((unsigned)g.P), dim3(kThreads), 0, as_stream(stream)>>>(
      x, dout, nullptr, mean, var, weight, bias, edz, eydz, dx, nullptr, dweight, dbias, eps, 0.f, rows, g, accumulate);
  return ok();
}

int skd_abn_relu_backward_dx_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout,
                                  const float *mean, const float *var, const float *weight, const float *edz,
                                  const float *eydz, float *dx, float *dres, float *dweight, float *dbias, float eps,
                                  int accumulate, skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || !x || !out || !dout || !mean || !var || !edz || !eydz || !dx) return 0;
  if (!aligned16(x) || !aligned16(out) || !aligned16(dout) || !aligned16(dx) || (dres && !aligned16(dres)) || (dweight && !weight)) return 0;
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)g.P), block(kThreads);
  if (dres != nullptr)
    abn_grad_dx_nhwc_kernel<SKD_ACT_NONE, 1, true><<<grid, block, 0, st>>>(x, out, dout, mean, var, weight, nullptr, edz, eydz, dx, dres, dweight, dbias, eps, 0.f, rows, g, accumulate);
  else
    abn_grad_dx_nhwc_kernel<SKD_ACT_NONE, 1, false><<<grid, block, 0, st>>>(x, out, dout, mean, var, weight, nullptr, edz, eydz, dx, dres, dweight, dbias, eps, 0.f, rows, g, accumulate);
  return ok();
}

// ---- student stem: BN -> ReLU -> MaxPool2d(3, 2, 1, ceil_mode) fused (round 6; include/skd.h section 1b) ----------------------
static bool stem_geom_ok(int B, int C, int H, int W, int OH, int OW) {
  if (B <= 0 || C < 4 || C > 4 * kThreads || (C & (C - 1)) || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return false;
  // every window starts inside the input (ceil_mode rule) and the windows cover it (csrc/maxpool.hip: pool_geom_ok)
  return 2 * (OH - 1) - 1 < H && 2 * (OW - 1) - 1 < W && 2 * (OH - 1) + 1 >= H - 1 && 2 * (OW - 1) + 1 >= W - 1;
}

int skd_abn_relu_maxpool3x3s2_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, const float *mean, const float *var,
                                   const float *weight, const float *bias, float eps, float *pooled, uint8_t *arg,
                                   skd_stream_t stream) {
  if (!stem_geom_ok(B, C, H, W, OH, OW) || !x || !mean || !var || !pooled || !arg) return 0;
  if (!aligned16(x) || !aligned16(pooled) || (reinterpret_cast<uintptr_t>(arg) & 3)) return 0;
  const int64_t items = (int64_t)B * OH * OW * (C / 4);
  if (cdiv(items, kThreads) > 2147483647) return 0;
  abn_relu_maxpool_nhwc_kernel<<<dim3((unsigned)cdiv(items, kThreads)), dim3(kThreads), 0, as_stream(stream)>>>(
      x, pooled, arg, mean, var, weight, bias, eps, items, H, W, OH, OW, C / 4);
  return ok();
}

int skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, const float *dpooled,
                                                   const uint8_t *arg, const float *mean, const float *var, const float *weight,
                                                   const float *bias, float *edz, float *eydz, float eps, float *workspace,
                                                   skd_stream_t stream) {
  if (!stem_geom_ok(B, C, H, W, OH, OW) || !x || !dpooled || !arg || !mean || !var || !edz || !eydz || !workspace) return 0;
  if (!aligned16(x) || !aligned16(dpooled) || (reinterpret_cast<uintptr_t>(arg) & 3)) return 0;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const int64_t blocks = (int64_t)B * H2 * W2;
  RedGeom rg;
  const int u = pick_u(blocks, C, 2, rg);                // (two blocks = eight positions in flight per thread)
  if (u == 0) return 0;
  unsigned *cnt = red_counters();
  if (cnt == nullptr) return 0;
  abn_pool_grad_nhwc2_kernel<2><<<dim3((unsigned)(rg.RG * rg.CB)), dim3(kRedThreads), 0, as_stream(stream)>>>(
      x, dpooled, arg, mean, var, weight, bias, workspace, cnt, edz, eydz, eps, blocks, H, W, OH, OW, H2, W2, rg);
  return ok();
}

int skd_abn_relu_maxpool3x3s2_backward_dx_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, const float *dpooled,
                                               const uint8_t *arg, const float *mean, const float *var, const float *weight,
                                               const float *bias, const float *edz, const float *eydz, float *dx, float *dweight,
                                               float *dbias, float eps, int accumulate, skd_stream_t stream) {
  if (!stem_geom_ok(B, C, H, W, OH, OW) || !x || !dpooled || !arg || !mean || !var || !edz || !eydz || !dx) return 0;
  if (!aligned16(x) || !aligned16(dpooled) || !aligned16(dx) || (reinterpret_cast<uintptr_t>(arg) & 3) || (dweight && !weight)) return 0;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const int64_t items = (int64_t)B * H2 * W2 * (C / 4);
  if (cdiv(items, kThreads) > 2147483647) return 0;
  abn_pool_grad_dx_nhwc_kernel<<<dim3((unsigned)cdiv(items, kThreads)), dim3(kThreads), 0, as_stream(stream)>>>(
      x, dpooled, arg, mean, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, items, H, W, OH, OW, H2, W2, C / 4, accumulate,
      (float)((int64_t)B * H * W));
  return ok();
}

// One-call channels-last backward (reduce + dx): ONE register-resident launch when the tensor fits (see "one launch with the
// tensor held in registers" above), the two launches of the entries above otherwise.  edz / eydz are outputs as well.
// sync_ctx != NULL (the *_sync entries below): InPlaceABNSync's backward (libs/functions.py:257-294) -- [edz | eydz] are
// exchanged and averaged over the replicas between the reduction and the dx pass: inside the one launch, or by
// skd_abn_sync_grad_stats between the two (edz, eydz must then be the two halves of ONE (2, C) buffer).
static int abn_backward_nhwc_any(void *sync_ctx, const float *rweights, int64_t rows, int C, const float *z, const float *dz,
                                 const float *var, const float *weight, const float *bias, float *edz, float *eydz, float *dx,
                                 float *dweight, float *dbias, float eps, int activation, float slope, int accumulate,
                                 float *workspace, skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || rows > 2147483647 || !z || !dz || !var || !edz || !eydz || !dx || !workspace) return 0;
  if (!aligned16(z) || !aligned16(dz) || !aligned16(dx) || activation == SKD_ACT_RELU || (dweight && !weight)) return 0;
  if (sync_ctx && (eydz != edz + C || 2 * C > kSyncMaxFloats)) return 0;
  hipStream_t st = as_stream(stream);
  FuseGeom f;
  if (aligned16(edz) && aligned16(eydz) && activation != SKD_ACT_ELU && (!sync_ctx || (sync_fused_enabled() && sync_fused_fits_device())) &&
      bwd_fused_geom(rows, C, f)) {
    int r;
    if (sync_ctx) {
      SyncArgs sy;
      if (!sync_next(sync_ctx, sy)) return 0;
      ++g_sync_form_calls[0];
      r = activation == SKD_ACT_LEAKY_RELU
              ? launch_bwd_fused<SKD_ACT_LEAKY_RELU, 0, false, true>(rows, C, z, dz, nullptr, nullptr, var, weight, bias, edz, eydz, dx, nullptr,
                                                                     dweight, dbias, eps, slope, accumulate, workspace, st, f, sy, rweights)
              : launch_bwd_fused<SKD_ACT_NONE, 0, false, true>(rows, C, z, dz, nullptr, nullptr, var, weight, bias, edz, eydz, dx, nullptr,
                                                               dweight, dbias, eps, slope, accumulate, workspace, st, f, sy, rweights);
      return r > 0 ? 1 : 0;
    }
    const SyncArgs sy = no_sync();
    r = activation == SKD_ACT_LEAKY_RELU
            ? launch_bwd_fused<SKD_ACT_LEAKY_RELU, 0, false, false>(rows, C, z, dz, nullptr, nullptr, var, weight, bias, edz, eydz, dx, nullptr,
                                                                    dweight, dbias, eps, slope, accumulate, workspace, st, f, sy, nullptr)
            : launch_bwd_fused<SKD_ACT_NONE, 0, false, false>(rows, C, z, dz, nullptr, nullptr, var, weight, bias, edz, eydz, dx, nullptr,
                                                              dweight, dbias, eps, slope, accumulate, workspace, st, f, sy, nullptr);
    if (r >= 0) return r;
  }
  if (!skd_abn_backward_reduce_nhwc(rows, C, z, dz, weight, bias, edz, eydz, eps, activation, slope, workspace, stream)) return 0;
  if (sync_ctx) ++g_sync_form_calls[1];
  if (sync_ctx && !skd_abn_sync_grad_stats(sync_ctx, C, edz, rweights, stream)) return 0;
  return skd_abn_backward_dx_nhwc(rows, C, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, activation, slope,
                                  accumulate, stream);
}

int skd_abn_backward_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *var, const float *weight,
                          const float *bias, float *edz, float *eydz, float *dx, float *dweight, float *dbias, float eps,
                          int activation, float slope, int accumulate, float *workspace, skd_stream_t stream) {
  return abn_backward_nhwc_any(nullptr, nullptr, rows, C, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, activation,
                               slope, accumulate, workspace, stream);
}

int skd_abn_backward_nhwc_sync(void *sync_ctx, int64_t rows, int C, const float *z, const float *dz, const float *var,
                               const float *weight, const float *bias, float *edz, float *eydz, float *dx, float *dweight,
                               float *dbias, const float *replica_weights, float eps, int activation, float slope, int accumulate,
                               float *workspace, skd_stream_t stream) {
  if (!sync_ctx) return 0;
  return abn_backward_nhwc_any(sync_ctx, replica_weights, rows, C, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps,
                               activation, slope, accumulate, workspace, stream);
}

// The same for the fused BN + ReLU (+ residual) op: out == NULL -> the mask is recomputed from x (forward without residual).
static int abn_relu_backward_nhwc_any(void *sync_ctx, const float *rweights, int64_t rows, int C, const float *x, const float *out,
                                      const float *dout, const float *mean, const float *var, const float *weight, const float *bias,
                                      float *edz, float *eydz, float *dx, float *dres, float *dweight, float *dbias, float eps,
                                      int accumulate, float *workspace, skd_stream_t stream) {
  NhwcGeom g;
  if (!make_nhwc_geom(rows, C, g) || rows > 2147483647 || !x || !dout || !mean || !var || !edz || !eydz || !dx || !workspace) return 0;
  if (!aligned16(x) || !aligned16(dout) || !aligned16(dx) || (out && !aligned16(out)) || (dres && !aligned16(dres)) || (dweight && !weight)) return 0;
  if (out == nullptr && dres != nullptr) return 0;
  if (sync_ctx && (eydz != edz + C || 2 * C > kSyncMaxFloats)) return 0;
  hipStream_t st = as_stream(stream);
  FuseGeom f;
  if (aligned16(edz) && aligned16(eydz) && (!sync_ctx || (sync_fused_enabled() && sync_fused_fits_device())) && bwd_fused_geom(rows, C, f)) {
    int r;
    if (sync_ctx) {
      SyncArgs sy;
      if (!sync_next(sync_ctx, sy)) return 0;
      ++g_sync_form_calls[0];
      if (out == nullptr)
        r = launch_bwd_fused<SKD_ACT_NONE, 2, false, true>(rows, C, x, dout, nullptr, mean, var, weight, bias, edz, eydz, dx, nullptr, dweight,
                                                           dbias, eps, 0.f, accumulate, workspace, st, f, sy, rweights);
      else if (dres != nullptr)
        r = launch_bwd_fused<SKD_ACT_NONE, 1, true, true>(rows, C, x, out, dout, mean, var, weight, bias, edz, eydz, dx, dres, dweight, dbias,
                                                          eps, 0.f, accumulate, workspace, st, f, sy, rweights);
      else
        r = launch_bwd_fused<SKD_ACT_NONE, 1, false, true>(rows, C, x, out, dout, mean, var, weight, bias, edz, eydz, dx, nullptr, dweight,
                                                           dbias, eps, 0.f, accumulate, workspace, st, f, sy, rweights);
      return r > 0 ? 1 : 0;
    }
    const SyncArgs sy = no_sync();
    if (out == nullptr)
      r = launch_bwd_fused<SKD_ACT_NONE, 2, false, false>(rows, C, x, dout, nullptr, mean, var, weight, bias, edz, eydz, dx, nullptr, dweight,
                                                          dbias, eps, 0.f, accumulate, workspace, st, f, sy, nullptr);
    else if (dres != nullptr)
      r = launch_bwd_fused<SKD_ACT_NONE, 1, true, false>(rows, C, x, out, dout, mean, var, weight, bias, edz, eydz, dx, dres, dweight, dbias,
                                                         eps, 0.f, accumulate, workspace, st, f, sy, nullptr);
    else
      r = launch_bwd_fused<SKD_ACT_NONE, 1, false, false>(rows, C, x, out, dout, mean, var, weight, bias, edz, eydz, dx, nullptr, dweight,
                                                          dbias, eps, 0.f, accumulate, workspace, st, f, sy, nullptr);
    if (r >= 0) return r;
  }
  if (sync_ctx) ++g_sync_form_calls[1];
  if (out == nullptr) {
    if (!skd_abn_relu_backward_reduce_nhwc_x(rows, C, x, dout, mean, var, weight, bias, edz, eydz, eps, workspace, stream)) return 0;
    if (sync_ctx && !skd_abn_sync_grad_stats(sync_ctx, C, edz, rweights, stream)) return 0;
    return skd_abn_relu_backward_dx_nhwc_x(rows, C, x, dout, mean, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, accumulate,
                                           stream);
  }
  if (!skd_abn_relu_backward_reduce_nhwc(rows, C, x, out, dout, mean, var, edz, eydz, eps, workspace, stream)) return 0;
  if (sync_ctx && !skd_abn_sync_grad_stats(sync_ctx, C, edz, rweights, stream)) return 0;
  return skd_abn_relu_backward_dx_nhwc(rows, C, x, out, dout, mean, var, weight, edz, eydz, dx, dres, dweight, dbias, eps, accumulate,
                                       stream);
}

int skd_abn_relu_backward_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout, const float *mean,
                               const float *var, const float *weight, const float *bias, float *edz, float *eydz, float *dx,
                               float *dres, float *dweight, float *dbias, float eps, int accumulate, float *workspace,
                               skd_stream_t stream) {
  return abn_relu_backward_nhwc_any(nullptr, nullptr, rows, C, x, out, dout, mean, var, weight, bias, edz, eydz, dx, dres, dweight, dbias,
                                    eps, accumulate, workspace, stream);
}

int skd_abn_relu_backward_nhwc_sync(void *sync_ctx, int64_t rows, int C, const float *x, const float *out, const float *dout,
                                    const float *mean, const float *var, const float *weight, const float *bias, float *edz,
                                    float *eydz, float *dx, float *dres, float *dweight, float *dbias, const float *replica_weights,
                                    float eps, int accumulate, float *workspace, skd_stream_t stream) {
  if (!sync_ctx) return 0;
  return abn_relu_backward_nhwc_any(sync_ctx, replica_weights, rows, C, x, out, dout, mean, var, weight, bias, edz, eydz, dx, dres, dweight,
                                    dbias, eps, accumulate, workspace, stream);
}

// out[0] = synchronised calls that ran as ONE launch with the exchange inside, out[1] = as statistics + exchange kernel +
// normalise (three launches), since the library was loaded.  Host counters, process-wide.
int skd_abn_sync_form_counts(int64_t *out) {
  if (!out) return 0;
  out[0] = g_sync_form_calls[0];
  out[1] = g_sync_form_calls[1];
  return 1;
}

// Upper bound of the workgroups of a one-launch (grid-barrier) pass, on top of the device's own limit (compute units):
// ranks that share ONE device must share its compute units or their barrier kernels cannot all be resident.  n > 0 sets the
// bound, n == 0 restores the default, n < 0 only queries.  Process-wide; returns the effective cap on the current device.
// 1 / 0: the one-launch passes on / off; -1: back to the environment (SKD_ABN_FUSED / SKD_ABN_SYNC_FUSED, default on), read at
// the next query.  The getters return the EFFECTIVE state (and resolve an unset one).
int skd_abn_set_fused(int on) {
  g_fused_state.store(on < 0 ? -1 : (on != 0), std::memory_order_relaxed);
  return 1;
}
int skd_abn_get_fused(void) { return fused_enabled() ? 1 : 0; }
int skd_abn_set_sync_fused(int on) {
  g_sync_fused_state.store(on < 0 ? -1 : (on != 0), std::memory_order_relaxed);
  return 1;
}
int skd_abn_get_sync_fused(void) { return sync_fused_enabled() ? 1 : 0; }

int skd_abn_set_fused_max_workgroups(int n) {
  if (n >= 0) g_fuse_user_cap = (n > 0 && n < kRedMaxWG) ? n : kRedMaxWG;
  return fuse_wg_cap();
}

// ---- legacy drop-in entries ---------------------------------------------------------------------

int skd_bn_mean_var(int N, int C, int S, const float *x, float *mean, float *var, skd_stream_t stream) {
  if (!valid_dims(N, C, S)) return 0;
  float *ws = legacy_scratch(skd_abn_workspace_floats(N, C, S));
  if (ws == nullptr) return 0;
  return skd_abn_stats(N, C, S, x, mean, var, ws, stream);
}

int skd_bn_forward(int N, int C, int S, const float *x, const float *mean, const float *var,
                   const float *weight, const float *bias, float *y, float *z, float eps,
                   skd_stream_t stream) {
  if (!valid_dims(N, C, S) || !x || !mean || !var || !y || !z) return 0;
  if (!same_phase(x, y) || !same_phase(x, z)) return 0;
  const Plan pl = make_plan(N, C, S);
  if (y == z)
    launch_apply<false>(SKD_ACT_NONE, pl, as_stream(stream), x, mean, var, weight, bias, y, z, eps, 0.f, N, C, S, 0);
  else
    launch_apply<true>(SKD_ACT_NONE, pl, as_stream(stream), x, mean, var, weight, bias, y, z, eps, 0.f, N, C, S, 0);
  return ok();
}

int skd_bn_edz_eydz(int N, int C, int S, const float *z, const float *dz, const float *weight,
                    const float *bias, float *edz, float *eydz, float eps, skd_stream_t stream) {
  if (!valid_dims(N, C, S)) return 0;
  float *ws = legacy_scratch(skd_abn_workspace_floats(N, C, S));
  if (ws == nullptr) return 0;
  return skd_abn_backward_reduce(N, C, S, z, dz, weight, bias, edz, eydz, eps, SKD_ACT_NONE, 0.f, ws, stream);
}

int skd_bn_backward(int N, int C, int S, const float *dz, const float *z, const float *var,
                    const float *weight, const float *bias, const float *edz, const float *eydz,
                    float *dx, float *dweight, float *dbias, float eps, skd_stream_t stream) {
  return skd_abn_backward_dx(N, C, S, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps,
                             SKD_ACT_NONE, 0.f, stream);
}

int skd_leaky_relu(int64_t N, float *x, float slope, skd_stream_t stream) {
  if (N < 0 || (N > 0 && !x)) return 0;
  if (N == 0) return 1;
  act_kernel<0><<<dim3(act_grid(N)), dim3(kThreads), 0, as_stream(stream)>>>(N, x, x, slope);
  return ok();
}
int skd_leaky_relu_backward(int64_t N, const float *x, float *dx, float slope, skd_stream_t stream) {
  if (N < 0 || (N > 0 && (!x || !dx))) return 0;
  if (N == 0) return 1;
  act_kernel<1><<<dim3(act_grid(N)), dim3(kThreads), 0, as_stream(stream)>>>(N, x, dx, slope);
  return ok();
}
int skd_elu(int64_t N, float *x, skd_stream_t stream) {
  if (N < 0 || (N > 0 && !x)) return 0;
  if (N == 0) return 1;
  act_kernel<2><<<dim3(act_grid(N)), dim3(kThreads), 0, as_stream(stream)>>>(N, x, x, 0.f);
  return ok();
}
int skd_elu_backward(int64_t N, const float *x, float *dx, skd_stream_t stream) {
  if (N < 0 || (N > 0 && (!x || !dx))) return 0;
  if (N == 0) return 1;
  act_kernel<3><<<dim3(act_grid(N)), dim3(kThreads), 0, as_stream(stream)>>>(N, x, dx, 0.f);
  return ok();
}
int skd_elu_inv(int64_t N, float *x, skd_stream_t stream) {
  if (N < 0 || (N > 0 && !x)) return 0;
  if (N == 0) return 1;
  act_kernel<4><<<dim3(act_grid(N)), dim3(kThreads), 0, as_stream(stream)>>>(N, x, x, 0.f);
  return ok();
}

}  // extern "C"
