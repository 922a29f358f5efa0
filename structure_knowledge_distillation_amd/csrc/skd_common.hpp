// skd_common.hpp -- shared device helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skd.h"

namespace skd {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kThreads = 256;      // 4 waves per workgroup: one per SIMD
constexpr int kWavesPerWG = kThreads / kWave;

static inline hipStream_t as_stream(skd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ok() { return hipGetLastError() == hipSuccess ? 1 : 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device-raised error words (status.hip; include/skd.h section 13) -------------------------------------------
constexpr int kStatusWords = 4;
constexpr int kStatusSyncTimeout = 0;    // a cross-replica mailbox exchange gave up waiting for a peer (sync.hip, abn.hip)
constexpr int kStatusFusedTimeout = 1;   // the grid barrier of a one-launch InPlace-ABN pass timed out: grid not co-resident
// host-mapped, system-coherent buffer of kStatusWords words as a DEVICE pointer (nullptr: allocation failed -> not reported)
unsigned *status_words();
__device__ __forceinline__ void raise_status(unsigned *status, int which, unsigned code) {
  if (status != nullptr) __hip_atomic_store(status + which, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the FIRST failure's code survives until the host clears the word (once an exchange has timed out every later
// exchange of the step gives up early and would otherwise overwrite the word with its own sequence number)
__device__ __forceinline__ void raise_status_first(unsigned *status, int which, unsigned code) {
  if (status == nullptr) return;
  unsigned expected = 0u;
  __hip_atomic_compare_exchange_strong(status + which, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One-time initialisation that is PER DEVICE (hipFuncSetAttribute, attribute queries): a process may drive several
// devices (the reference's own threading model), so a plain `static bool` is wrong there.
struct PerDeviceFlag {
  bool done[64] = {};
  // the current device's flag, or nullptr when the device cannot be determined
  bool *get() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    return &done[dev];
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = kWave / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = kWave / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = kWave / 2; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
  return v;
}

// Workgroup-wide sum of up to two values; result valid in thread 0 (and wave 0).
// `scratch` must hold 2*kWavesPerWG floats of LDS.
__device__ __forceinline__ void block_sum2(float &a, float &b, float *scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) {
    scratch[wid] = a;
    scratch[kWavesPerWG + wid] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < kWavesPerWG; ++w) {
      sa += scratch[w];
      sb += scratch[kWavesPerWG + w];
    }
    a = sa;
    b = sb;
  }
}

// Cross-replica combine of the batch statistics for ONE channel (libs/functions.py:196-197 + the running update of
// :208-209): mean = sum_g w_g mean_g, var = sum_g w_g (var_g + (mean - mean_g)^2), w_g = 1 / G without `weights`.
// `ld(g, j)` returns element j (0 <= j < 2 C) of rank g's [mean | var] vector.  One definition for the gathered-buffer kernel
// (abn.hip) and the one-hop exchange kernel (sync.hip), so that both produce the same bits.
template <class L>
__device__ __forceinline__ void combine_channel(int G, int C, int c, L ld, const float *__restrict__ weights, int rank, float nf,
                                                float momentum, float &m_out, float &v_out, float *running_mean,
                                                float *running_var) {
  float m = 0.f;
  for (int g = 0; g < G; ++g) m += (weights ? weights[g] : 1.f) * ld(g, c);
  if (!weights) m /= (float)G;
  float v = 0.f;
  for (int g = 0; g < G; ++g) {
    const float d = m - ld(g, c);
    v += (weights ? weights[g] : 1.f) * (ld(g, C + c) + d * d);
  }
  if (!weights) v /= (float)G;
  if (weights) nf = nf / weights[rank];
  m_out = m;
  v_out = v;
  if (running_mean != nullptr) running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * m;
  if (running_var != nullptr) running_var[c] = running_var[c] * (1.f - momentum) + momentum * (nf > 1.f ? v * nf / (nf - 1.f) : v);
}

// Stream a contiguous, arbitrarily aligned run of `len` floats with the whole workgroup:
// scalar head up to the first 16-byte boundary, float4 body, scalar tail.
// The functor supplies split load / use halves so that the four 16-B loads of a lane are
// all issued before the first dependent use (stores to a possibly aliasing pointer would
// otherwise serialise them):
//   auto  op.ld1(int i)            -> scalar payload of element i
//   void  op.use1(int i, payload)
//   auto  op.ld4(int i)            -> vector payload of elements i..i+3
//   void  op.use4(int i, payload)
// `addr_phase_bytes` is the byte address of element 0 of the run for ONE of the tensors; all
// tensors touched by the functor must share that 16-B phase (checked on the host side).
template <class Op>
__device__ __forceinline__ void stream_run(uintptr_t addr_phase_bytes, int len, Op &op) {
  const int mis = (int)((addr_phase_bytes >> 2) & 3);
  int head = (4 - mis) & 3;
  if (head > len) head = len;
  const int t = threadIdx.x;
  if (t < head) op.use1(t, op.ld1(t));
  const int nvec = (len - head) >> 2;
  for (int i = t; i < nvec; i += kThreads * 4) {
    const int i1 = i + kThreads, i2 = i + 2 * kThreads, i3 = i + 3 * kThreads;
    const int e0 = head + 4 * i;
    const int e1 = i1 < nvec ? head + 4 * i1 : e0;
    const int e2 = i2 < nvec ? head + 4 * i2 : e0;
    const int e3 = i3 < nvec ? head + 4 * i3 : e0;
    auto a0 = op.ld4(e0);
    auto a1 = op.ld4(e1);
    auto a2 = op.ld4(e2);
    auto a3 = op.ld4(e3);
    op.use4(e0, a0);
    if (i1 < nvec) op.use4(e1, a1);
    if (i2 < nvec) op.use4(e2, a2);
    if (i3 < nvec) op.use4(e3, a3);
  }
  const int done = head + (nvec << 2);
  if (t < len - done) op.use1(done + t, op.ld1(done + t));
}

}  // namespace skd

// Defined in reduce.hip: out[0] = scale * sum(part[0..n)) with a single workgroup, double
// accumulation, fixed order (deterministic).  `n` is a partial count (small).
namespace skd {
int launch_final_sum(const float *part, int64_t n, float *out, double scale, hipStream_t st);
}
