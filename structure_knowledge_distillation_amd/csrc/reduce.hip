// reduce.hip -- deterministic two-stage sum + library identification.
#include "skd_common.hpp"

namespace skd {
namespace {

// stage 1: each workgroup sums a contiguous slab, float4 loads, writes one partial.
__global__ __launch_bounds__(kThreads) void sum_stage1_kernel(const float *__restrict__ x, int64_t n,
                                                             int64_t per_wg, float *__restrict__ part) {
  __shared__ float red[2 * kWavesPerWG];
  const int64_t lo = (int64_t)blockIdx.x * per_wg;
  const int64_t hi = lo + per_wg < n ? lo + per_wg : n;
  float s = 0.f, c = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) s += x[i];
  block_sum2(s, c, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(kThreads) void sum_final_kernel(const float *__restrict__ part, int64_t n,
                                                            float *__restrict__ out, double scale) {
  __shared__ double red[kWavesPerWG];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) s += (double)part[i];
  s = wave_sum(s);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kWavesPerWG; ++w) t += red[w];
    out[0] = (float)(t * scale);
  }
}

constexpr int kMaxStage1 = 1024;

}  // namespace

int launch_final_sum(const float *part, int64_t n, float *out, double scale, hipStream_t st) {
  sum_final_kernel<<<dim3(1), dim3(kThreads), 0, st>>>(part, n, out, scale);
  return ok();
}

}  // namespace skd

using namespace skd;

extern "C" {

int skd_abi_version(void) { return 1; }
int skd_target_arch(void) { return 950; }

int skd_sum_f32(int64_t n, const float *x, float *out, float scale, float *workspace,
                skd_stream_t stream) {
  if (n < 0 || !out || (n > 0 && (!x || !workspace))) return 0;
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return 0;
    return 1;
  }
  int64_t wgs = cdiv(n, 4096);
  if (wgs > kMaxStage1) wgs = kMaxStage1;
  const int64_t per_wg = cdiv(n, wgs);
  wgs = cdiv(n, per_wg);
  sum_stage1_kernel<<<dim3((unsigned)wgs), dim3(kThreads), 0, st>>>(x, n, per_wg, workspace);
  return launch_final_sum(workspace, wgs, out, (double)scale, st);
}

}  // extern "C"
