// conv1x1.hip -- 1x1 convolution + eval-mode InPlace-ABN (+ residual) + activation as ONE fp32-MFMA GEMM for gfx950.
//
// The frozen teacher (ResNet101, eval, no grad) spends 3.9 ms per step in its BatchNorm passes although every one of
// them follows a convolution whose output it merely rescales (networks/pspnet_combine.py:65-84: conv1 -> bn1 -> relu,
// conv3 -> bn3 -> + residual -> relu).  For the 1x1 convolutions -- in channels-last memory a plain GEMM
//     Y[M][N] = act( ((X[M][K] . W[N][K]^T - mean[n]) * invstd[n]) * gamma[n] + beta[n]  [+ R[M][N]] ),   M = B*H*W
// -- the normalisation, the residual add and the ReLU move into the epilogue here: the convolution output is written
// once, already activated, and the separate 8 / 12 byte-per-element ABN pass disappears.  The
// epilogue evaluates exactly the formula of abn_apply (bn.cu:146-159 + ReLU) on the accumulator, so nothing is folded
// into the weights and checkpoints / state-dict semantics are untouched.
//
// Kernel: 128 x 128 output tile per 256-thread workgroup (4 waves, each 64 x 64 = 2 x 2 v_mfma_f32_32x32x2_f32 blocks:
// exact fp32, 64 accumulator registers), K streamed in tiles of 32 through a double-buffered LDS ring.  Both operands are
// K-contiguous in memory (activations channels-last, weights (N, K, 1, 1)), and they stay K-contiguous in LDS: rows of
// 32 + 4 floats, so a lane fetches FOUR consecutive k of its row with one conflict-free ds_read_b128 (row stride 36
// floats spreads 16 lanes x 16 bytes over all 64 banks) and feeds four MFMAs from it -- the MFMA's two k-slots
// (lanes 0-31 / 32-63) are simply assigned k = 0..3 and k = 4..7 of each group of eight, identically for A and B.
// Per 8 k: 4 ds_read_b128 -> 16 MFMAs (1024 matrix-pipe cycles); per K-tile and thread 8 global float4 loads and 8
// ds_write_b128.  73.7 KB of LDS per workgroup -> 2 workgroups (2 waves per SIMD) per CU.
// Bound: fp32 MFMA (157.3 TFLOP/s); algorithmic flops 2*M*N*K; epilogue traffic 4*M*N (+ 4*M*N residual) bytes.
#include "skd_common.hpp"

namespace skd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTM = 128, kTN = 128, kBK = 32, kLDK = kBK + 4;
constexpr int kStageFloats = (kTM + kTN) * kLDK;           // 9216 floats = 36,864 bytes
constexpr size_t kConvLds = sizeof(float) * 2 * kStageFloats;

__device__ __forceinline__ float inv_std_of(float var, float eps) { return (var != 0.f || eps != 0.f) ? 1.f / sqrtf(var + eps) : 0.f; }

// Staging registers are plain float4 variables handled by macros: two named sets (a*, b* suffix 0 / 1) that ping-pong.
// (Held in a struct passed by reference they ended up in scratch memory once two sets were live across the loop.)
// thread t -> row t/8 (+32h), float4 t%8 of the K-tile.  Rows beyond the matrix are clamped (never stored).
#define SKD_GLOAD(S, k0)                                                                          \
  do {                                                                                            \
    const float *xa = X + (k0) + gkq, *wb = Wt + (k0) + gkq;                                       \
    xa0##S = *reinterpret_cast<const float4 *>(xa + grow0);                                        \
    xa1##S = *reinterpret_cast<const float4 *>(xa + grow1);                                        \
    xa2##S = *reinterpret_cast<const float4 *>(xa + grow2);                                        \
    xa3##S = *reinterpret_cast<const float4 *>(xa + grow3);                                        \
    wb0##S = *reinterpret_cast<const float4 *>(wb + wrow0);                                        \
    wb1##S = *reinterpret_cast<const float4 *>(wb + wrow0 + 32 * (int64_t)K);                      \
    wb2##S = *reinterpret_cast<const float4 *>(wb + wrow0 + 64 * (int64_t)K);                      \
    wb3##S = *reinterpret_cast<const float4 *>(wb + wrow0 + 96 * (int64_t)K);                      \
  } while (0)
#define SKD_SSTORE(S, stage)                                                                      \
  do {                                                                                            \
    float *sa = (stage) + srow, *sb = (stage) + kTM * kLDK + srow;                                 \
    *reinterpret_cast<float4 *>(sa) = xa0##S;                                                      \
    *reinterpret_cast<float4 *>(sa + 32 * kLDK) = xa1##S;                                          \
    *reinterpret_cast<float4 *>(sa + 64 * kLDK) = xa2##S;                                          \
    *reinterpret_cast<float4 *>(sa + 96 * kLDK) = xa3##S;                                          \
    *reinterpret_cast<float4 *>(sb) = wb0##S;                                                      \
    *reinterpret_cast<float4 *>(sb + 32 * kLDK) = wb1##S;                                          \
    *reinterpret_cast<float4 *>(sb + 64 * kLDK) = wb2##S;                                          \
    *reinterpret_cast<float4 *>(sb + 96 * kLDK) = wb3##S;                                          \
  } while (0)

__device__ __forceinline__ void tile_mma(const float *stage, f32x16 (&acc)[2][2]) {
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int wi = (wid >> 1) * 64, wj = (wid & 1) * 64;
  const int half = lane >> 5, r = lane & 31;
  const float *pa = stage + (wi + r) * kLDK + half * 4;
  const float *pb = stage + (kTM + wj + r) * kLDK + half * 4;
#pragma unroll
  for (int g = 0; g < kBK / 8; ++g) {
    const float4 a0 = *reinterpret_cast<const float4 *>(pa + g * 8);
    const float4 a1 = *reinterpret_cast<const float4 *>(pa + 32 * kLDK + g * 8);
    const float4 b0 = *reinterpret_cast<const float4 *>(pb + g * 8);
    const float4 b1 = *reinterpret_cast<const float4 *>(pb + 32 * kLDK + g * 8);
    const float A0[4] = {a0.x, a0.y, a0.z, a0.w}, A1[4] = {a1.x, a1.y, a1.z, a1.w};
    const float B0[4] = {b0.x, b0.y, b0.z, b0.w}, B1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[t], B0[t], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[t], B1[t], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[t], B0[t], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[t], B1[t], acc[1][1], 0, 0, 0);
    }
  }
}

template <int ACT, bool HAS_RES>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv1x1_abn_kernel(const float *__restrict__ X, const float *__restrict__ Wt,
                                                                  const float *__restrict__ R, float *__restrict__ Y,
                                                                  const float *__restrict__ mean, const float *__restrict__ var,
                                                                  const float *__restrict__ weight, const float *__restrict__ bias,
                                                                  float eps, float slope, int64_t M, int K, int N, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // consecutive workgroups walk the N tiles of one row panel: the 128 x K activation panel is read from HBM once and
  // re-used out of L2 by its tiles_n neighbours; the (N, K) weights stay L2-resident throughout
  const int tn = blockIdx.x % tiles_n;
  const int64_t tm = blockIdx.x / tiles_n;
  const int64_t m0 = tm * kTM;
  const int n0 = tn * kTN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  const int nk = K / kBK;
  // Software pipeline, two K-tiles deep, unrolled by two so that the two staging register sets ping-pong WITHOUT copies:
  // while tile kt is multiplied out of one LDS stage, tile kt + 1 moves from its staging set into the other stage
  // (its loads were issued a whole MFMA phase earlier) and the loads of tile kt + 2 are issued into the set that has
  // just been drained.  (A single loop-carried staging set makes the compiler load into temporaries and copy them at
  // the bottom of the trip -- an HBM-latency wait per K-tile; loads placed at the top of a trip are sunk to their use.)
  // K is a multiple of 64; indices past the end are clamped to the last tile (loaded, never used).
  const int gt = threadIdx.x, grow = gt >> 3, gkq = (gt & 7) * 4;
  const int srow = grow * kLDK + gkq;
  auto clampm = [&](int64_t m) { return (m > M - 1 ? M - 1 : m) * K; };
  const int64_t grow0 = clampm(m0 + grow), grow1 = clampm(m0 + grow + 32), grow2 = clampm(m0 + grow + 64), grow3 = clampm(m0 + grow + 96);
  const int64_t wrow0 = (int64_t)(n0 + grow) * K;
  float4 xa00, xa10, xa20, xa30, wb00, wb10, wb20, wb30;     // staging set 0
  float4 xa01, xa11, xa21, xa31, wb01, wb11, wb21, wb31;     // staging set 1
  SKD_GLOAD(0, 0);
  SKD_SSTORE(0, lds);
  SKD_GLOAD(1, kBK);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    tile_mma(lds, acc);
    SKD_SSTORE(1, lds + kStageFloats);                                          // tile kt + 1
    SKD_GLOAD(0, (kt + 2 < nk ? kt + 2 : nk - 1) * kBK);
    __syncthreads();
    tile_mma(lds + kStageFloats, acc);
    SKD_SSTORE(0, lds);                                                         // tile kt + 2
    SKD_GLOAD(1, (kt + 3 < nk ? kt + 3 : nk - 1) * kBK);
    __syncthreads();
  }
  // ---- epilogue: the eval-mode InPlace-ABN formula on the accumulator (+ residual) + activation ----
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int wi = (wid >> 1) * 64, wj = (wid & 1) * 64;
#pragma unroll
  for (int bj = 0; bj < 2; ++bj) {
    const int col = n0 + wj + bj * 32 + (lane & 31);
    const float mu = mean[col], is = inv_std_of(var[col], eps);
    const float ga = weight != nullptr ? fabsf(weight[col]) + eps : 1.f;     // bn.cu:153
    const float be = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = m0 + wi + bi * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < M) {
          float z = ((acc[bi][bj][q] - mu) * is) * ga + be;                 // bn.cu:158-159
          if (HAS_RES) z += R[row * N + col];
          if (ACT == SKD_ACT_RELU) z = z < 0.f ? 0.f : z;
          if (ACT == SKD_ACT_LEAKY_RELU) z = z < 0.f ? z * slope : z;
          Y[row * N + col] = z;
        }
      }
    }
  }
}

template <int ACT, bool HAS_RES>
static int launch(const float *X, const float *Wt, const float *R, float *Y, const float *mean, const float *var,
                  const float *weight, const float *bias, float eps, float slope, int64_t M, int K, int N, hipStream_t st) {
  static bool ready = false;
  if (!ready) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_abn_kernel<ACT, HAS_RES>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kConvLds) != hipSuccess) return 0;
    ready = true;
  }
  const int tiles_n = N / kTN;
  const int64_t tiles_m = cdiv(M, kTM);
  if (tiles_m * tiles_n > 2147483647) return 0;
  conv1x1_abn_kernel<ACT, HAS_RES><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(kThreads), kConvLds, st>>>(
      X, Wt, R, Y, mean, var, weight, bias, eps, slope, M, K, N, tiles_n);
  return ok();
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

// 1 when the fused kernel takes the problem (K a multiple of 64, N a multiple of 128), 0 when the caller must run
// the convolution and the ABN pass separately.
int skd_conv1x1_abn_supported(int64_t M, int K, int N) { return M > 0 && K > 0 && N > 0 && K % (2 * kBK) == 0 && N % kTN == 0; }

int skd_conv1x1_abn_nhwc(int64_t M, int K, int N, const float *x, const float *w, const float *residual, float *out,
                         const float *mean, const float *var, const float *weight, const float *bias, float eps,
                         int activation, float slope, skd_stream_t stream) {
  if (!skd_conv1x1_abn_supported(M, K, N) || !x || !w || !out || !mean || !var) return 0;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) return 0;
  hipStream_t st = as_stream(stream);
#define SKD_C11(A)                                                                                                       \
  return residual ? launch<A, true>(x, w, residual, out, mean, var, weight, bias, eps, slope, M, K, N, st)               \
                  : launch<A, false>(x, w, residual, out, mean, var, weight, bias, eps, slope, M, K, N, st)
  switch (activation) {
    case SKD_ACT_NONE: SKD_C11(SKD_ACT_NONE);
    case SKD_ACT_RELU: SKD_C11(SKD_ACT_RELU);
    case SKD_ACT_LEAKY_RELU: SKD_C11(SKD_ACT_LEAKY_RELU);
    default: return 0;
  }
#undef SKD_C11
}

}  // extern "C"
