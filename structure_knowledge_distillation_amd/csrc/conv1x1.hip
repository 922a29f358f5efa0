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
// exact fp32, 64 accumulator registers), K streamed in tiles of 16 through a double-buffered LDS ring.  Both operands are
// K-contiguous in memory (activations channels-last, weights (N, K, 1, 1)), and they stay K-contiguous in LDS: rows of
// 16 + 4 floats, so a lane fetches FOUR consecutive k of its row with one conflict-free ds_read_b128 (row stride 20
// floats puts the 16 lanes of each of the instruction's four lane groups on 16 distinct 4-bank slots) and feeds four MFMAs
// from it -- the MFMA's two k-slots (lanes 0-31 / 32-63) are simply assigned k = 0..3 and k = 4..7 of each group of eight,
// identically for A and B.  Per 8 k: 4 ds_read_b128 -> 16 MFMAs (1024 matrix-pipe cycles), the reads of the next group issued
// after the first four MFMAs of the current one (sched_group_barrier); per K-tile and thread 4 global float4 loads and 4
// ds_write_b128.  41 KB of LDS per workgroup, <= 168 VGPRs -> 3 workgroups (3 waves per SIMD) per CU.
// Measured decomposition of the main loop (tools/gemm_lab variants, profiles/r03f/g_*): LDS reads cost nothing, LDS writes +
// barrier 4 %, the global loads 13 % (two tiles of look-ahead do not help: not latency); an MFMA-only loop with this tile
// shape and epilogue reaches 0.60-0.86 of the nominal 157.3 TFLOP/s depending on how well M x N / (128 x 128) divides the
// 256 CUs and on the sustained clock.
// Bound: fp32 MFMA (157.3 TFLOP/s); algorithmic flops 2*M*N*K; epilogue traffic 4*M*N (+ 4*M*N residual) bytes.
#include <stdio.h>
#include <stdlib.h>

#include "skd_common.hpp"

namespace skd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// K-tile 16 (round 3, tools/gemm_lab variants 0 / 5, profiles/r03f_gemm_lab_variants.jsonl): the same speed as 32 on the long-K
// problems and +7 % at K = 256 -- half the LDS per workgroup, so THREE workgroups (12 waves) share a CU and the prologue /
// epilogue of one tile hides behind two neighbours' MFMAs instead of one.
constexpr int kTM = 128, kTN = 128, kBK = 16, kLDK = kBK + 4, kMinWG = 3;
constexpr int kQK = kBK / 4;                               // float4 per panel row
constexpr int kRPP = kThreads / kQK;                       // panel rows covered by one pass of the workgroup (64)
constexpr int kRB = kTN / kRPP;                            // float4 of the B panel per thread and K-tile (2); A: TM / kRPP (2 or 1)
constexpr int kStageFloats = (kTM + kTN) * kLDK;           // 5120 floats = 20,480 bytes
constexpr size_t kConvLds = sizeof(float) * 2 * kStageFloats;
constexpr int kProMaxK = 512;                             // prologue form: the (4, K) parameter table rides in LDS (<= 8 KB)

__device__ __forceinline__ float inv_std_of(float var, float eps) { return (var != 0.f || eps != 0.f) ? 1.f / sqrtf(var + eps) : 0.f; }

// Per K-tile a thread moves four float4 of the A panel (rows t/8 + 32h, k quad t%8) and four of the B panel from global memory
// to LDS.  The loads are issued at the top of a trip, the LDS stores after the trip's MFMAs (a whole K-tile of matrix-pipe
// time for them to land), the staging registers are not loop-carried (nothing for the compiler to copy).
template <int TM>          // TM = rows of the output tile: 128, or 64 for the tiles of a launch's last, partial round (see the kernel)
struct Staging {
  float4 a[TM / kRPP], b[kRB];
};
struct ProParams {         // PRO: the k quad's mean / invstd / gamma / beta of the BatchNorm applied to A on the way in
  float4 pm, pi, pg, pb;   // (kept apart from Staging: a struct with members that one instantiation never writes stayed in scratch)
};

template <bool PRO, int TM>
__device__ __forceinline__ void stage_load(Staging<TM> &s, const float *__restrict__ X, const float *__restrict__ Wt, int k0,
                                           int gkq, const int64_t (&arow)[TM / kRPP], int64_t wrow0, int K) {
  const float *xa = X + k0 + gkq, *wb = Wt + k0 + gkq;
#pragma unroll
  for (int h = 0; h < TM / kRPP; ++h) s.a[h] = *reinterpret_cast<const float4 *>(xa + arow[h]);
#pragma unroll
  for (int h = 0; h < kRB; ++h) s.b[h] = *reinterpret_cast<const float4 *>(wb + wrow0 + (int64_t)(kRPP * h) * K);
}

// relu(bn(x)) with the expression of abn_apply (bn.cu:146-159 + ReLU): ((x - mean) * invstd) * gamma + beta
__device__ __forceinline__ float pro_one(float x, float m, float is, float g, float b) {
  const float z = __builtin_fmaf((x - m) * is, g, b);
  return z < 0.f ? 0.f : z;
}

template <bool PRO, int TM>
__device__ __forceinline__ void stage_store(const Staging<TM> s, float *stage, int srow, const float *ptab, int K, int kq) {
  constexpr int kRA = TM / kRPP;
  float *sa = stage + srow, *sb = stage + TM * kLDK + srow;
  if (PRO) {   // ptab (LDS copy of ppack): mean | invstd | gamma | beta, each K floats; kq = first of this thread's four k
    ProParams pp;
    pp.pm = *reinterpret_cast<const float4 *>(ptab + kq);
    pp.pi = *reinterpret_cast<const float4 *>(ptab + K + kq);
    pp.pg = *reinterpret_cast<const float4 *>(ptab + 2 * K + kq);
    pp.pb = *reinterpret_cast<const float4 *>(ptab + 3 * K + kq);
#pragma unroll
    for (int h = 0; h < kRA; ++h) {
      float4 v = s.a[h];
      v.x = pro_one(v.x, pp.pm.x, pp.pi.x, pp.pg.x, pp.pb.x);
      v.y = pro_one(v.y, pp.pm.y, pp.pi.y, pp.pg.y, pp.pb.y);
      v.z = pro_one(v.z, pp.pm.z, pp.pi.z, pp.pg.z, pp.pb.z);
      v.w = pro_one(v.w, pp.pm.w, pp.pi.w, pp.pg.w, pp.pb.w);
      *reinterpret_cast<float4 *>(sa + kRPP * h * kLDK) = v;
    }
  } else {
#pragma unroll
    for (int h = 0; h < kRA; ++h) *reinterpret_cast<float4 *>(sa + kRPP * h * kLDK) = s.a[h];
  }
#pragma unroll
  for (int h = 0; h < kRB; ++h) *reinterpret_cast<float4 *>(sb + kRPP * h * kLDK) = s.b[h];
}

// One K-tile out of LDS: per group of 8 k conflict-free ds_read_b128 (a0 [, a1], b0, b1: FOUR consecutive k of the lane's row) feed
// 8 * WM MFMAs; the reads of group g + 1 are issued before the MFMAs of group g (software pipelining in registers).
// Waves 2 x 2, wave tile (TM / 2) x 64 = WM x 2 blocks of 32 x 32 (WM = TM / 64).
template <int TM>
__device__ __forceinline__ void tile_mma(const float *stage, f32x16 (&acc)[TM / 64][2]) {
  constexpr int WM = TM / 64;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int wi = (wid >> 1) * (TM / 2), wj = (wid & 1) * 64;
  const int half = lane >> 5, r = lane & 31;
  const float *pa = stage + (wi + r) * kLDK + half * 4;
  const float *pb = stage + (TM + wj + r) * kLDK + half * 4;
  float4 a[WM], b0 = *reinterpret_cast<const float4 *>(pb), b1 = *reinterpret_cast<const float4 *>(pb + 32 * kLDK);
#pragma unroll
  for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const float4 *>(pa + 32 * i * kLDK);
  __builtin_amdgcn_sched_group_barrier(0x100, WM + 2, 0);  // the reads of group 0 (the pipeline below is matched in order)
#pragma unroll
  for (int g = 0; g < kBK / 8; ++g) {
    float4 na[WM], nb0 = b0, nb1 = b1;
#pragma unroll
    for (int i = 0; i < WM; ++i) na[i] = a[i];
    if (g + 1 < kBK / 8) {
#pragma unroll
      for (int i = 0; i < WM; ++i) na[i] = *reinterpret_cast<const float4 *>(pa + 32 * i * kLDK + (g + 1) * 8);
      nb0 = *reinterpret_cast<const float4 *>(pb + (g + 1) * 8);
      nb1 = *reinterpret_cast<const float4 *>(pb + 32 * kLDK + (g + 1) * 8);
    }
    const float B0[4] = {b0.x, b0.y, b0.z, b0.w}, B1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const float av = t == 0 ? a[i].x : t == 1 ? a[i].y : t == 2 ? a[i].z : a[i].w;
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, B0[t], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, B1[t], acc[i][1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = na[i];
    b0 = nb0; b1 = nb1;
    // pin the issue order (left alone the scheduler sinks the reads to just before their first use and every group starts
    // with an exposed LDS round trip): the first k's MFMAs, the next group's reads, the other MFMAs (their matrix-pipe cycles cover them)
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * WM, 0);
    if (g + 1 < kBK / 8) __builtin_amdgcn_sched_group_barrier(0x100, WM + 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6 * WM, 0);
  }
}

// Epilogue of one 32 x 32 accumulator block: rows row0 + frag_row(q), column col.  FULL: every row of the tile exists (all
// but the last row tile) -- no per-element branches, and the sixteen residual loads are issued together before the first
// use (round 2 interleaved load -> wait -> store per element behind a branch: 64 serialised HBM round trips per lane).
// NT: the residual is read and the output written with the non-temporal hint (round 6).  For WIDE outputs (the super-tile order,
// K = 512 / N = 2048) the 554 MB residual + output stream of a launch shares each XCD's 4 MB L2 with the operands it is trying to
// keep (2 MB of activations per panel group, 1 MB of weights per chunk) and evicts them: counters showed the activations fetched
// ~4 x (617 MB read for 350 MB algorithmic, profiles/r05j_pmc.json case 135).  Neither stream is read again by this kernel.
template <int ACT, bool HAS_RES, bool FULL, bool NT>
__device__ __forceinline__ void store_block(const f32x16 &acc, const float *__restrict__ R, float *__restrict__ Y, int64_t row0,
                                            int col, int64_t M, int N, float mu, float is, float ga, float be, float slope) {
  const int lane = threadIdx.x & (kWave - 1);
  float rv[16];
  int64_t off[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int64_t row = row0 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
    off[q] = row * N + col;
    rv[q] = 0.f;
    if (HAS_RES && (FULL || row < M)) rv[q] = NT ? __builtin_nontemporal_load(R + off[q]) : R[off[q]];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int64_t row = row0 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
    float z = ((acc[q] - mu) * is) * ga + be;                 // bn.cu:158-159
    if (HAS_RES) z += rv[q];
    if (ACT == SKD_ACT_RELU) z = z < 0.f ? 0.f : z;
    if (ACT == SKD_ACT_LEAKY_RELU) z = z < 0.f ? z * slope : z;
    if (FULL || row < M) {
      if (NT) __builtin_nontemporal_store(z, Y + off[q]);
      else Y[off[q]] = z;
    }
  }
}

// One output tile of TM x 128: K loop through the double-buffered LDS ring, then the epilogue.
template <int TM, int ACT, bool HAS_RES, bool PRO, bool NT>
__device__ __forceinline__ void conv1x1_tile(const float *__restrict__ X, const float *__restrict__ Wt, const float *__restrict__ R,
                                             float *__restrict__ Y, const float *__restrict__ mean, const float *__restrict__ var,
                                             const float *__restrict__ weight, const float *__restrict__ bias,
                                             const float *__restrict__ ppack, float eps, float slope, int64_t M, int K, int N,
                                             int64_t m0, int n0, float *lds) {
  constexpr int WM = TM / 64, kRA = TM / kRPP;
  f32x16 acc[WM][2];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  const int nk = K / kBK;
  const int gt = threadIdx.x, grow = gt / kQK, gkq = (gt % kQK) * 4;
  const int srow = grow * kLDK + gkq;
  int64_t arow[kRA];   // rows beyond the matrix are clamped (loaded, multiplied, never stored)
#pragma unroll
  for (int h = 0; h < kRA; ++h) {
    const int64_t m = m0 + grow + kRPP * h;
    arow[h] = (m > M - 1 ? M - 1 : m) * K;
  }
  const int64_t wrow0 = (int64_t)(n0 + grow) * K;
  float *ptab = lds + 2 * kStageFloats;             // PRO: the (4, K) parameter table, K <= kProMaxK
  if (PRO) {
    for (int i = gt * 4; i < 4 * K; i += kThreads * 4) *reinterpret_cast<float4 *>(ptab + i) = *reinterpret_cast<const float4 *>(ppack + i);
    __syncthreads();
  }
  Staging<TM> st;
  stage_load<PRO, TM>(st, X, Wt, 0, gkq, arow, wrow0, K);
  stage_store<PRO, TM>(st, lds, srow, ptab, K, gkq);
  __syncthreads();
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) stage_load<PRO, TM>(st, X, Wt, (kt + 1) * kBK, gkq, arow, wrow0, K);
    tile_mma<TM>(lds + stage * kStageFloats, acc);
    if (more) stage_store<PRO, TM>(st, lds + (stage ^ 1) * kStageFloats, srow, ptab, K, (kt + 1) * kBK + gkq);
    __syncthreads();
    stage ^= 1;
  }
  // ---- epilogue: the eval-mode InPlace-ABN formula on the accumulator (+ residual) + activation ----
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int wi = (wid >> 1) * (TM / 2), wj = (wid & 1) * 64;
  const bool full = m0 + TM <= M;
#pragma unroll
  for (int bj = 0; bj < 2; ++bj) {
    const int col = n0 + wj + bj * 32 + (lane & 31);
    const float mu = mean[col], is = inv_std_of(var[col], eps);
    const float ga = weight != nullptr ? fabsf(weight[col]) + eps : 1.f;     // bn.cu:153
    const float be = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
    for (int bi = 0; bi < WM; ++bi) {
      const int64_t row0 = m0 + wi + bi * 32;
      if (full)
        store_block<ACT, HAS_RES, true, NT>(acc[bi][bj], R, Y, row0, col, M, N, mu, is, ga, be, slope);
      else
        store_block<ACT, HAS_RES, false, NT>(acc[bi][bj], R, Y, row0, col, M, N, mu, is, ga, be, slope);
    }
  }
}

template <int ACT, bool HAS_RES, bool PRO, bool NT>
__global__ __launch_bounds__(kThreads, kMinWG) void conv1x1_abn_kernel(
    const float *__restrict__ X, const float *__restrict__ Wt, const float *__restrict__ R, float *__restrict__ Y,
    const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ weight,
    const float *__restrict__ bias, const float *__restrict__ ppack, float eps, float slope, int64_t M, int K, int N,
    int tiles_n, int pm, int ct, int p_full) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // XCD-aware tile order.  Workgroup b runs on XCD b % 8 (observed placement, MI355X_MICROARCH.md; used for traffic only, any
  // placement is correct) and every XCD has its own L2: the tiles_n workgroups that share one 128 x K activation panel are
  // therefore given to ONE XCD, back to back in its dispatch order -- the panel is fetched into one L2 instead of eight.
  // Counters before (profiles/r03d_gemm_lab_pmc.json, K = 256, N = 1024): 421 MB fetched for 174 MB of algorithmic reads
  // (8 x the 35 MB of activations); time-neutral in isolation (the Infinity Cache absorbs the fills), less traffic beside the
  // D stream.  Row panel p lives on XCD p % 8; the grid is padded to a multiple of 8 panels, the padding exits here.
  //
  // Round 4: for WIDE outputs that order thrashes the weights instead.  At K = 512, N = 2048 the 16 column
  // tiles of a panel stream all of W (4 MB = one XCD's whole L2) past every panel: counters showed 850 MB fetched for 350 MB of
  // algorithmic reads.  So an XCD walks SUPER-TILES: groups of `pm` row panels (<= 2 MB of activations) x chunks of `ct` column
  // tiles (<= 1 MB of weights): for group: for chunk: for panel in group: for column tile in chunk.  A weight chunk is reused
  // by pm panels back to back, a panel group stays L2-resident across the chunks; W traffic falls from ~one sweep per panel to one
  // per group.  ct == tiles_n (narrow outputs, e.g. K = 256 / N = 1024: 23 of the 33 launches of a teacher forward) degenerates
  // to the panel-major order above.
  //
  // Round 6: HALF-HEIGHT tiles for the launch's last, partial round.  The layer-3 problem is 2120 tiles of 128 x 128 on 768 slots
  // (3 workgroups per CU): 2.76 rounds -- and the lab shows what the ragged last round costs: the same core reaches 0.72 of the
  // peak on a tile count that divides the chip and 0.63-0.69 on this one (profiles/r06t_gemm_lab_quantisation.jsonl).  The row
  // panels p < p_full are 128 rows high, the panels behind them (the rows a whole number of rounds does not cover) 64: twice the
  // workgroups of half the duration fill the last round's slots (p_full is a multiple of 8: whole XCD rows).
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int per_group = pm * tiles_n;
  const int group = j / per_group, r = j - group * per_group;
  const int per_chunk = pm * ct;
  const int chunk = r / per_chunk, r2 = r - chunk * per_chunk;
  const int pl = r2 / ct;
  const int tn = chunk * ct + (r2 - pl * ct);
  const int64_t tm = ((int64_t)group * pm + pl) * 8 + xcd;
  if (tn >= tiles_n) return;
  const int n0 = tn * kTN;
  if (tm < p_full) {
    const int64_t m0 = tm * kTM;
    if (m0 >= M) return;
    conv1x1_tile<kTM, ACT, HAS_RES, PRO, NT>(X, Wt, R, Y, mean, var, weight, bias, ppack, eps, slope, M, K, N, m0, n0, lds);
  } else {
    const int64_t m0 = (int64_t)p_full * kTM + (tm - p_full) * (kTM / 2);
    if (m0 >= M) return;
    conv1x1_tile<kTM / 2, ACT, HAS_RES, PRO, NT>(X, Wt, R, Y, mean, var, weight, bias, ppack, eps, slope, M, K, N, m0, n0, lds);
  }
}

__global__ void pack_eval_params_kernel(int K, const float *__restrict__ mean, const float *__restrict__ var,
                                        const float *__restrict__ weight, const float *__restrict__ bias, float eps,
                                        float *__restrict__ pack) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  pack[k] = mean[k];
  pack[K + k] = inv_std_of(var[k], eps);
  pack[2 * (int64_t)K + k] = weight != nullptr ? fabsf(weight[k]) + eps : 1.f;   // bn.cu:153
  pack[3 * (int64_t)K + k] = bias != nullptr ? bias[k] : 0.f;
}

template <int ACT, bool HAS_RES, bool PRO, bool NT = false>
static int launch(const float *X, const float *Wt, const float *R, float *Y, const float *mean, const float *var,
                  const float *weight, const float *bias, const float *ppack, float eps, float slope, int64_t M, int K, int N,
                  hipStream_t st) {
  static PerDeviceFlag ready;          // per instantiation AND per device
  bool *rdy = ready.get();
  if (rdy == nullptr) return 0;
  if (!*rdy) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_abn_kernel<ACT, HAS_RES, PRO, NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kConvLds + sizeof(float) * 4 * kProMaxK)) != hipSuccess) return 0;
    *rdy = true;
  }
  const size_t lds_bytes = kConvLds + (PRO ? sizeof(float) * 4 * (size_t)K : 0);
  const int tiles_n = N / kTN;
  const int64_t tiles_m = cdiv(M, kTM);
  // super-tile geometry (see the kernel): column-tile chunks of <= 1 MB of weights, panel groups of <= 2 MB of activations
  const int64_t tile_bytes = (int64_t)kTN * K * sizeof(float);
  // (ten geometries were swept with the counters: profiles/r04g_pmc.json, profiles/r04h_two_ranks_one_gpu_bisect.txt; this is the
  // one that minimised L2 -> fabric reads at K = 512 / N = 2048: 851 -> 617 MB per launch)
  int ct = (int)((1 << 20) / tile_bytes), pm = (int)((2 << 20) / tile_bytes);
  if (ct < 1) ct = 1;
  if (pm < 1) pm = 1;
  if (ct >= tiles_n) { ct = tiles_n; pm = 1; }              // narrow output: plain panel-major order
  else if (!NT) return launch<ACT, HAS_RES, PRO, true>(X, Wt, R, Y, mean, var, weight, bias, ppack, eps, slope, M, K, N, st);   // wide: NT epilogue
  while (tiles_n % ct) --ct;                                // chunks of equal width (tiles_n is a power of two in this network)
  // half-height panels for the last, partial round; slots = 3 workgroups per compute unit
  int64_t p_full = tiles_m, panels = tiles_m;
  {
    static PerDeviceFlag cu_known;
    static int cus[64] = {};
    bool *known = cu_known.get();
    int dev = 0;
    if (known != nullptr && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      if (!*known) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus[dev] = n;
        *known = true;
      }
      const int64_t slots = (int64_t)kMinWG * cus[dev];
      const int64_t tiles = tiles_m * tiles_n;
      if (slots > 0 && tiles > slots && tiles % slots != 0) {
        p_full = (tiles / slots) * slots / tiles_n / 8 * 8;            // whole rounds, whole XCD rows of panels
        const int64_t rest_rows = M - p_full * kTM;
        panels = p_full + cdiv(rest_rows, kTM / 2);
      }
    }
  }
  const int64_t panels_per_xcd = cdiv(cdiv(panels, 8), pm) * pm;   // row panels padded to whole groups on each of the 8 XCDs
  if (panels_per_xcd * 8 * tiles_n > 2147483647) return 0;
  const int64_t grid = panels_per_xcd * 8 * tiles_n;
  conv1x1_abn_kernel<ACT, HAS_RES, PRO, NT><<<dim3((unsigned)grid), dim3(kThreads), lds_bytes, st>>>(
      X, Wt, R, Y, mean, var, weight, bias, ppack, eps, slope, M, K, N, tiles_n, pm, ct, (int)p_full);
  return ok();
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

// 1 when the fused kernel takes the problem (K a multiple of 16, N a multiple of 128), 0 when the caller must run
// the convolution and the ABN pass separately.
int skd_conv1x1_abn_supported(int64_t M, int K, int N) { return M > 0 && K > 0 && N > 0 && K % kBK == 0 && N % kTN == 0; }

static int conv1x1_dispatch(int64_t M, int K, int N, const float *x, const float *w, const float *residual, float *out,
                            const float *mean, const float *var, const float *weight, const float *bias, const float *ppack,
                            float eps, int activation, float slope, skd_stream_t stream) {
  if (!skd_conv1x1_abn_supported(M, K, N) || !x || !w || !out || !mean || !var) return 0;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(ppack)) & 15) return 0;
  hipStream_t st = as_stream(stream);
#define SKD_C11(A)                                                                                                        \
  if (ppack != nullptr)                                                                                                   \
    return residual ? launch<A, true, true>(x, w, residual, out, mean, var, weight, bias, ppack, eps, slope, M, K, N, st)   \
                    : launch<A, false, true>(x, w, residual, out, mean, var, weight, bias, ppack, eps, slope, M, K, N, st); \
  return residual ? launch<A, true, false>(x, w, residual, out, mean, var, weight, bias, nullptr, eps, slope, M, K, N, st) \
                  : launch<A, false, false>(x, w, residual, out, mean, var, weight, bias, nullptr, eps, slope, M, K, N, st)
  switch (activation) {
    case SKD_ACT_NONE: SKD_C11(SKD_ACT_NONE);
    case SKD_ACT_RELU: SKD_C11(SKD_ACT_RELU);
    case SKD_ACT_LEAKY_RELU: SKD_C11(SKD_ACT_LEAKY_RELU);
    default: return 0;
  }
#undef SKD_C11
}

int skd_conv1x1_abn_nhwc(int64_t M, int K, int N, const float *x, const float *w, const float *residual, float *out,
                         const float *mean, const float *var, const float *weight, const float *bias, float eps,
                         int activation, float slope, skd_stream_t stream) {
  return conv1x1_dispatch(M, K, N, x, w, residual, out, mean, var, weight, bias, nullptr, eps, activation, slope, stream);
}

// pack (4, K) = [mean | invstd(var, eps) | |weight| + eps (1 when NULL) | bias (0 when NULL)]: the per-channel constants of an
// eval-mode InPlace-ABN (bn.cu:146-159) in the form the prologue below consumes.  A frozen network computes it once.
int skd_abn_pack_eval_params(int K, const float *mean, const float *var, const float *weight, const float *bias, float eps,
                             float *pack, skd_stream_t stream) {
  if (K <= 0 || !mean || !var || !pack) return 0;
  pack_eval_params_kernel<<<dim3((unsigned)cdiv(K, 256)), dim3(256), 0, as_stream(stream)>>>(K, mean, var, weight, bias, eps, pack);
  return ok();
}

// The same GEMM with the PRECEDING eval-mode BatchNorm + ReLU applied to x on the way into LDS:
//   a[m][k] = relu(((x[m][k] - mean_k) * invstd_k) * gamma_k + beta_k),   (mean | invstd | gamma | beta) = ppack (4, K)
// (networks/pspnet_combine.py:71-75: conv2 -> bn2 -> relu -> conv3 -> bn3 -> + residual -> relu): x is the raw output of the
// 3x3 convolution, neither ABN pass of the block tail exists any more.  ppack: skd_abn_pack_eval_params, 16-byte aligned.
int skd_conv1x1_abn_pro_nhwc(int64_t M, int K, int N, const float *x, const float *w, const float *residual, float *out,
                             const float *mean, const float *var, const float *weight, const float *bias, float eps,
                             const float *ppack, int activation, float slope, skd_stream_t stream) {
  if (!ppack || K > kProMaxK) return 0;
  return conv1x1_dispatch(M, K, N, x, w, residual, out, mean, var, weight, bias, ppack, eps, activation, slope, stream);
}

}  // extern "C"
