// gemm_lab_kernels.hip -- EXPERIMENT variants of the fp32-MFMA TN GEMM core (both operands K-contiguous: C[M][N] = X[M][K] . W[N][K]^T),
// timed by tools/gemm_lab.  Not shipped: what wins here moves into csrc/conv1x1.hip / csrc/pairwise.hip.
//
// Knobs (template parameters): BK (K-tile), WM x WN MFMA blocks per wave (wave tile 32*WM x 32*WN), 2 x 2 waves per workgroup,
// minimum workgroups per CU (register budget), MODE:
//   0 full kernel                         1 no global loads in the loop (the first tile's registers are re-stored every trip)
//   2 no LDS writes / barrier either      3 MFMA only (operands from registers)
// The epilogue is a plain coalesced dword store of C (no BN, no residual): the variants isolate the main loop.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lab {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kWave = 64;

template <int RA, int RB>
struct Stg {
  float4 a[RA], b[RB];
};

template <int RA, int RB, int RPP, int SEL = 3>
__device__ __forceinline__ void lab_gload(Stg<RA, RB> &o, const float *__restrict__ X, const float *__restrict__ Wt, int64_t m0, int grow, int gkq,
                                          int64_t wrow0, int64_t M, int K, int k0, int KW = 0) {
  if (KW == 0) KW = K;            // row stride of the B operand (the padded-weights experiment)
  if (SEL & 1) {
#pragma unroll
    for (int h = 0; h < RA; ++h) {
      const int64_t m = m0 + grow + RPP * h;
      o.a[h] = *reinterpret_cast<const float4 *>(X + (m > M - 1 ? M - 1 : m) * K + k0 + gkq);
    }
  }
  if (SEL & 2) {
#pragma unroll
    for (int h = 0; h < RB; ++h) o.b[h] = *reinterpret_cast<const float4 *>(Wt + wrow0 + (int64_t)(RPP * h) * KW + k0 + gkq);
  }
}
template <int RA, int RB, int RPP, int LDK, int TM>
__device__ __forceinline__ void lab_sstore(const Stg<RA, RB> v, float *stage, int srow) {
#pragma unroll
  for (int h = 0; h < RA; ++h) *reinterpret_cast<float4 *>(stage + srow + RPP * h * LDK) = v.a[h];
#pragma unroll
  for (int h = 0; h < RB; ++h) *reinterpret_cast<float4 *>(stage + TM * LDK + srow + RPP * h * LDK) = v.b[h];
}

template <int BK, int WM, int WN, int MINWG, int MODE, int WVM = 2, int WVN = 2>
__global__ __launch_bounds__(64 * WVM * WVN, MINWG) void tn_gemm_kernel(const float *__restrict__ X, const float *__restrict__ Wt,
                                                             float *__restrict__ Y, int64_t M, int K, int N, int tiles_n, int ld, int ldw) {
  constexpr int NT = 64 * WVM * WVN;             // threads
  constexpr int TM = 32 * WM * WVM, TN = 32 * WN * WVN;
  constexpr int LDK = BK + 4;
  constexpr int STAGE = (TM + TN) * LDK;
  constexpr int RA = TM * BK / 4 / NT, RB = TN * BK / 4 / NT;     // float4 per thread and K-tile
  constexpr int QK = BK / 4;                                      // float4 per panel row
  constexpr int ROWS_PER_PASS = NT / QK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tn = blockIdx.x % tiles_n;
  const int64_t tm = blockIdx.x / tiles_n;
  const int64_t m0 = tm * TM;
  const int n0 = tn * TN;
  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  const int nk = K / BK;
  const int t = threadIdx.x, grow = t / QK, gkq = (t % QK) * 4;
  const int srow = grow * LDK + gkq;
  const int64_t wrow0 = (int64_t)(n0 + grow) * ldw;
  Stg<RA, RB> st;
  // MODE 7: every workgroup starts its K loop at a different K-tile (and wraps): at any instant the workgroups of the chip read
  // DIFFERENT 64-byte columns of their power-of-two-strided rows (L2 channel spreading)
  const int rot = MODE == 7 ? (int)((blockIdx.x * 5u) % (unsigned)nk) : 0;
  auto ktile = [&](int kt) { int q = kt + rot; return (q >= nk ? q - nk : q) * BK; };
  const int lane = t & (kWave - 1), wid = t / kWave;
  const int wi = (wid / WVN) * 32 * WM, wj = (wid % WVN) * 32 * WN;
  const int half = lane >> 5, r = lane & 31;
  auto mma = [&](const float *stage) {
    const float *pa = stage + (wi + r) * LDK + half * 4;
    const float *pb = stage + (TM + wj + r) * LDK + half * 4;
    float4 a[WM], b[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const float4 *>(pa + 32 * i * LDK);
#pragma unroll
    for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const float4 *>(pb + 32 * j * LDK);
    __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float4 na[WM], nb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) na[i] = a[i];
#pragma unroll
      for (int j = 0; j < WN; ++j) nb[j] = b[j];
      if (g + 1 < BK / 8) {
#pragma unroll
        for (int i = 0; i < WM; ++i) na[i] = *reinterpret_cast<const float4 *>(pa + 32 * i * LDK + (g + 1) * 8);
#pragma unroll
        for (int j = 0; j < WN; ++j) nb[j] = *reinterpret_cast<const float4 *>(pb + 32 * j * LDK + (g + 1) * 8);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const float av = q == 0 ? a[i].x : q == 1 ? a[i].y : q == 2 ? a[i].z : a[i].w;
            const float bv = q == 0 ? b[j].x : q == 1 ? b[j].y : q == 2 ? b[j].z : b[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = na[i];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = nb[j];
      __builtin_amdgcn_sched_group_barrier(0x008, WM * WN, 0);
      if (g + 1 < BK / 8) __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * WM * WN, 0);
    }
  };
  lab_gload<RA, RB, ROWS_PER_PASS>(st, X, Wt, m0, grow, gkq, wrow0, M, ld, ktile(0), ldw);
  lab_sstore<RA, RB, ROWS_PER_PASS, LDK, TM>(st, lds, srow);
  __syncthreads();
  int stage = 0;
  if (MODE == 3) {
    float av = X[t], bv = Wt[t];
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
      for (int q = 0; q < BK / 2; ++q)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
    }
  } else {
    if (MODE == 6) {
      // two K-tiles ahead: st holds tile kt + 1 (already loaded), st2 receives tile kt + 2 while tile kt is multiplied
      Stg<RA, RB> st2;
      if (nk > 1) lab_gload<RA, RB, ROWS_PER_PASS>(st, X, Wt, m0, grow, gkq, wrow0, M, ld, BK);
      for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 2 < nk) lab_gload<RA, RB, ROWS_PER_PASS>(st2, X, Wt, m0, grow, gkq, wrow0, M, ld, (kt + 2) * BK);
        mma(lds + stage * STAGE);
        if (kt + 1 < nk) lab_sstore<RA, RB, ROWS_PER_PASS, LDK, TM>(st, lds + (stage ^ 1) * STAGE, srow);
        __syncthreads();
        stage ^= 1;
        if (kt + 1 >= nk) break;
        if (kt + 3 < nk) lab_gload<RA, RB, ROWS_PER_PASS>(st, X, Wt, m0, grow, gkq, wrow0, M, ld, (kt + 3) * BK);
        mma(lds + stage * STAGE);
        if (kt + 2 < nk) lab_sstore<RA, RB, ROWS_PER_PASS, LDK, TM>(st2, lds + (stage ^ 1) * STAGE, srow);
        __syncthreads();
        stage ^= 1;
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if ((MODE == 0 || MODE == 7) && more) lab_gload<RA, RB, ROWS_PER_PASS>(st, X, Wt, m0, grow, gkq, wrow0, M, ld, ktile(kt + 1), ldw);
        if (MODE == 4 && more) lab_gload<RA, RB, ROWS_PER_PASS, 1>(st, X, Wt, m0, grow, gkq, wrow0, M, ld, (kt + 1) * BK);
        if (MODE == 5 && more) lab_gload<RA, RB, ROWS_PER_PASS, 2>(st, X, Wt, m0, grow, gkq, wrow0, M, ld, (kt + 1) * BK);
        mma(lds + stage * STAGE);
        if (MODE != 2) {
          if (more) lab_sstore<RA, RB, ROWS_PER_PASS, LDK, TM>(st, lds + (stage ^ 1) * STAGE, srow);
          __syncthreads();
          stage ^= 1;
        }
      }
    }
  }
  // plain epilogue
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col = n0 + wj + j * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = m0 + wi + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < M) Y[row * N + col] = acc[i][j][q];
      }
  }
}

template <int BK, int WM, int WN, int MINWG, int MODE, int WVM = 2, int WVN = 2>
int launch_tn(const float *X, const float *Wt, float *Y, int64_t M, int K, int N, int pad = 0, int padw = -1) {
  if (padw < 0) padw = pad;
  constexpr int TM = 32 * WM * WVM, TN = 32 * WN * WVN;
  constexpr size_t lds = sizeof(float) * 2 * (TM + TN) * (BK + 4);
  if (K % BK || N % TN) return 0;
  static bool ready = false;
  if (!ready) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(tn_gemm_kernel<BK, WM, WN, MINWG, MODE, WVM, WVN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return 0;
    ready = true;
  }
  const int tiles_n = N / TN;
  const int64_t tiles_m = (M + TM - 1) / TM;
  tn_gemm_kernel<BK, WM, WN, MINWG, MODE, WVM, WVN><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(64 * WVM * WVN), lds, 0>>>(X, Wt, Y, M, K, N, tiles_n, K + pad, K + padw);
  return hipGetLastError() == hipSuccess;
}

// ---- variant: operands DMA'd straight into LDS (global_load_lds_dwordx4: no VGPR round trip, no ds_write) ------------------
// A wave-instruction writes 64 x 16 B = 1 KiB of CONTIGUOUS LDS (lane i -> base + 16 i), so the panels are unpadded
// [128 rows][BK = 16 floats] and the bank spreading that the +4-float row padding gave is done by an XOR swizzle instead:
// the 16-byte slot (row, q) holds k-quad q ^ ((row >> 2) & 3) -- chosen on the GLOBAL side (each lane picks which 16 bytes of
// its row it fetches), undone on the read side.  ds_read_b128's 16-lane groups then hit 16 distinct 4-bank slots.
template <int MINWG>
__global__ __launch_bounds__(256, MINWG) void tn_gemm_glds_kernel(const float *__restrict__ X, const float *__restrict__ Wt,
                                                                  float *__restrict__ Y, int64_t M, int K, int N, int tiles_n) {
  constexpr int BK = 16, TM = 128, TN = 128;
  constexpr int PANEL = TM * BK;                 // floats per operand panel (8 KiB)
  constexpr int STAGE = 2 * PANEL;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tn = blockIdx.x % tiles_n;
  const int64_t tm = blockIdx.x / tiles_n;
  const int64_t m0 = tm * TM;
  const int n0 = tn * TN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  const int nk = K / BK;
  const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
  // DMA geometry: wave w, instruction h in {0, 1} fills slots [(2 w + h) * 64, + 64) of a panel; slot s = (row s / 4, position s % 4)
  int64_t ga[2], gb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int s = (2 * wid + h) * 64 + lane, row = s >> 2, kq = (s & 3) ^ ((row >> 2) & 3);
    const int64_t m = m0 + row;
    ga[h] = (m > M - 1 ? M - 1 : m) * K + 4 * kq;
    gb[h] = (int64_t)(n0 + row) * K + 4 * kq;
  }
  auto dma = [&](float *stage, int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_global_load_lds(X + ga[h] + k0, stage + (2 * wid + h) * 256, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(Wt + gb[h] + k0, stage + PANEL + (2 * wid + h) * 256, 16, 0, 0);
    }
  };
  const int wi = (wid >> 1) * 64, wj = (wid & 1) * 64;
  const int half = lane >> 5, r = lane & 31;
  const int fr = (r >> 2) & 3;                    // (row >> 2) & 3 of rows wi + r and wi + r + 32 alike
  auto mma = [&](const float *stage) {
    const float *pa = stage + (wi + r) * BK, *pb = stage + PANEL + (wj + r) * BK;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      const int q = ((2 * g + half) ^ fr) * 4;
      const float4 a0 = *reinterpret_cast<const float4 *>(pa + q), a1 = *reinterpret_cast<const float4 *>(pa + 32 * BK + q);
      const float4 b0 = *reinterpret_cast<const float4 *>(pb + q), b1 = *reinterpret_cast<const float4 *>(pb + 32 * BK + q);
      const float A0[4] = {a0.x, a0.y, a0.z, a0.w}, A1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float B0[4] = {b0.x, b0.y, b0.z, b0.w}, B1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[u], B0[u], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[u], B1[u], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[u], B0[u], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[u], B1[u], acc[1][1], 0, 0, 0);
      }
    }
  };
  dma(lds, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) dma(lds + (stage ^ 1) * STAGE, (kt + 1) * BK);
    mma(lds + stage * STAGE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage ^= 1;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wj + j * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = m0 + wi + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < M) Y[row * N + col] = acc[i][j][q];
      }
  }
}

template <int MINWG>
int launch_glds(const float *X, const float *Wt, float *Y, int64_t M, int K, int N) {
  constexpr size_t lds = sizeof(float) * 2 * 2 * 128 * 16;      // 32 KiB
  if (K % 16 || N % 128) return 0;
  const int tiles_n = N / 128;
  const int64_t tiles_m = (M + 127) / 128;
  tn_gemm_glds_kernel<MINWG><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(256), lds, 0>>>(X, Wt, Y, M, K, N, tiles_n);
  return hipGetLastError() == hipSuccess;
}

// ---- variant: persistent workgroups with a dynamic tile counter -------------------------------------------------------------
// grid = 256 CUs x MINWG workgroups; each workgroup draws 128 x 128 tiles from an atomic counter until none is left (no
// half-empty last round: M x N / (128 x 128) need not divide the workgroup slots) and issues the global loads of its NEXT
// tile's first K-panel before the epilogue of the current one (the prologue latency hides under the stores).
template <int MINWG>
__global__ __launch_bounds__(256, MINWG) void tn_gemm_persistent_kernel(const float *__restrict__ X, const float *__restrict__ Wt,
                                                                        float *__restrict__ Y, int64_t M, int K, int N, int tiles_n,
                                                                        int ntiles, unsigned *counter) {
  constexpr int BK = 16, TM = 128, TN = 128, LDK = BK + 4, STAGE = (TM + TN) * LDK;
  constexpr int QK = BK / 4, RPP = 256 / QK, RA = TM / RPP, RB = TN / RPP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int next_s;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int grow = t / QK, gkq = (t % QK) * 4, srow = grow * LDK + gkq;
  const int wi = (wid >> 1) * 64, wj = (wid & 1) * 64;
  const int half = lane >> 5, r = lane & 31;
  const int nk = K / BK;
  auto mma = [&](const float *stage, f32x16 (&acc)[2][2]) {
    const float *pa = stage + (wi + r) * LDK + half * 4;
    const float *pb = stage + (TM + wj + r) * LDK + half * 4;
    float4 a0 = *reinterpret_cast<const float4 *>(pa), a1 = *reinterpret_cast<const float4 *>(pa + 32 * LDK);
    float4 b0 = *reinterpret_cast<const float4 *>(pb), b1 = *reinterpret_cast<const float4 *>(pb + 32 * LDK);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (g + 1 < BK / 8) {
        na0 = *reinterpret_cast<const float4 *>(pa + (g + 1) * 8);
        na1 = *reinterpret_cast<const float4 *>(pa + 32 * LDK + (g + 1) * 8);
        nb0 = *reinterpret_cast<const float4 *>(pb + (g + 1) * 8);
        nb1 = *reinterpret_cast<const float4 *>(pb + 32 * LDK + (g + 1) * 8);
      }
      const float A0[4] = {a0.x, a0.y, a0.z, a0.w}, A1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float B0[4] = {b0.x, b0.y, b0.z, b0.w}, B1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[u], B0[u], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[u], B1[u], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[u], B0[u], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[u], B1[u], acc[1][1], 0, 0, 0);
      }
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      if (g + 1 < BK / 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    }
  };
  if (t == 0) next_s = (int)atomicAdd(counter, 1u);
  __syncthreads();
  int tile = next_s;
  Stg<RA, RB> st;
  if (tile < ntiles) {
    const int64_t m0 = (int64_t)(tile / tiles_n) * TM;
    lab_gload<RA, RB, RPP>(st, X, Wt, m0, grow, gkq, (int64_t)((tile % tiles_n) * TN + grow) * K, M, K, 0);
  }
  while (tile < ntiles) {
    const int64_t m0 = (int64_t)(tile / tiles_n) * TM;
    const int n0 = (tile % tiles_n) * TN;
    const int64_t wrow0 = (int64_t)(n0 + grow) * K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    __syncthreads();                                   // every wave is done with next_s and with both LDS stages of the last tile
    if (t == 0) next_s = (int)atomicAdd(counter, 1u);  // the tile after this one
    lab_sstore<RA, RB, RPP, LDK, TM>(st, lds, srow);
    __syncthreads();
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) lab_gload<RA, RB, RPP>(st, X, Wt, m0, grow, gkq, wrow0, M, K, (kt + 1) * BK);
      mma(lds + stage * STAGE, acc);
      if (more) lab_sstore<RA, RB, RPP, LDK, TM>(st, lds + (stage ^ 1) * STAGE, srow);
      __syncthreads();
      stage ^= 1;
    }
    const int nxt = next_s;
    if (nxt < ntiles) {                                // the next tile's first panel travels while this tile is stored
      const int64_t nm0 = (int64_t)(nxt / tiles_n) * TM;
      lab_gload<RA, RB, RPP>(st, X, Wt, nm0, grow, gkq, (int64_t)((nxt % tiles_n) * TN + grow) * K, M, K, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wj + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int64_t row = m0 + wi + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          if (row < M) Y[row * N + col] = acc[i][j][q];
        }
    }
    tile = nxt;
  }
}

static unsigned *g_tile_counter = nullptr;
template <int MINWG>
int launch_persistent(const float *X, const float *Wt, float *Y, int64_t M, int K, int N) {
  constexpr size_t lds = sizeof(float) * 2 * 256 * 20;
  if (K % 16 || N % 128) return 0;
  if (g_tile_counter == nullptr && hipMalloc(reinterpret_cast<void **>(&g_tile_counter), 4) != hipSuccess) return 0;
  if (hipMemsetAsync(g_tile_counter, 0, 4, 0) != hipSuccess) return 0;
  const int tiles_n = N / 128;
  const int64_t tiles_m = (M + 127) / 128;
  tn_gemm_persistent_kernel<MINWG><<<dim3(256 * MINWG), dim3(256), lds, 0>>>(X, Wt, Y, M, K, N, tiles_n, (int)(tiles_m * tiles_n), g_tile_counter);
  return hipGetLastError() == hipSuccess;
}

}  // namespace lab

// C-callable table for gemm_lab.cpp
extern "C" int lab_tn_gemm(int variant, const float *X, const float *Wt, float *Y, int64_t M, int K, int N) {
  using namespace lab;
  switch (variant) {
    case 0: return launch_tn<32, 2, 2, 2, 0>(X, Wt, Y, M, K, N);    // the shipped structure: 128 x 128 x 32, 2 workgroups / CU
    case 1: return launch_tn<32, 2, 2, 2, 1>(X, Wt, Y, M, K, N);    //   ... without global loads in the loop
    case 2: return launch_tn<32, 2, 2, 2, 2>(X, Wt, Y, M, K, N);    //   ... without LDS writes / barriers either
    case 3: return launch_tn<32, 2, 2, 2, 3>(X, Wt, Y, M, K, N);    //   ... MFMA only
    case 4: return launch_tn<16, 2, 2, 4, 0>(X, Wt, Y, M, K, N);    // BK 16: 36 KB LDS, up to 4 workgroups / CU
    case 5: return launch_tn<16, 2, 2, 3, 0>(X, Wt, Y, M, K, N);    // BK 16, 3 workgroups / CU (168 VGPRs)
    case 6: return launch_tn<32, 4, 2, 1, 0>(X, Wt, Y, M, K, N);    // 256 x 128 tile, wave tile 128 x 64, 1 workgroup / CU
    case 7: return launch_tn<32, 2, 4, 1, 0>(X, Wt, Y, M, K, N);    // 128 x 256 tile, wave tile 64 x 128
    case 8: return launch_tn<16, 4, 4, 1, 0>(X, Wt, Y, M, K, N);    // 256 x 256 x 16, wave tile 128 x 128 (256 accumulator registers)
    case 9: return launch_tn<32, 2, 2, 1, 0>(X, Wt, Y, M, K, N);    // shipped tile, 1 workgroup / CU allowed to use 512 registers
    case 10: return launch_tn<32, 2, 2, 2, 4>(X, Wt, Y, M, K, N);   // only the A operand is loaded in the loop
    case 11: return launch_tn<32, 2, 2, 2, 5>(X, Wt, Y, M, K, N);   // only the B operand
    case 12: return launch_tn<32, 2, 2, 2, 6>(X, Wt, Y, M, K, N);   // loads issued two K-tiles ahead (two staging register sets)
    case 13: return launch_glds<3>(X, Wt, Y, M, K, N);              // global -> LDS DMA, XOR-swizzled unpadded panels, 3 workgroups / CU
    case 14: return launch_glds<4>(X, Wt, Y, M, K, N);              //   ... 4 workgroups / CU
    case 15: return launch_tn<16, 2, 2, 2, 0, 4, 2>(X, Wt, Y, M, K, N);   // 256 x 128 x 16, 8 waves of 64 x 64, 2 workgroups / CU
    case 16: return launch_tn<16, 2, 2, 2, 0, 2, 4>(X, Wt, Y, M, K, N);   // 128 x 256 x 16, 8 waves
    case 17: return launch_tn<16, 2, 2, 1, 0, 4, 4>(X, Wt, Y, M, K, N);   // 256 x 256 x 16, 16 waves of 64 x 64, 1 workgroup / CU
    case 18: return launch_persistent<3>(X, Wt, Y, M, K, N);              // persistent 128 x 128 x 16, dynamic tile counter, 3 workgroups / CU
    case 19: return launch_persistent<2>(X, Wt, Y, M, K, N);              //   ... 2 workgroups / CU
    case 20: return launch_tn<16, 4, 2, 2, 0>(X, Wt, Y, M, K, N);         // 256 x 128 x 16, 4 waves of 128 x 64 (128 accumulators), 2 workgroups / CU
    case 21: return launch_tn<16, 2, 4, 2, 0>(X, Wt, Y, M, K, N);         // 128 x 256 x 16, 4 waves of 64 x 128
    case 22: return launch_tn<16, 4, 4, 1, 0>(X, Wt, Y, M, K, N);         // (= 8) 256 x 256 x 16, 4 waves of 128 x 128, 1 workgroup / CU
    // round 6: SMALL tiles at HIGH occupancy -- what MIOpen's igemm kernels (bt128x64x16/32, 0.78-0.88 of the peak) do
    case 23: return launch_tn<16, 2, 1, 5, 0>(X, Wt, Y, M, K, N);         // 128 x 64 x 16, 4 waves of 64 x 32 (32 accumulators), 5 workgroups / CU
    case 24: return launch_tn<16, 1, 2, 5, 0>(X, Wt, Y, M, K, N);         // 64 x 128 x 16, 4 waves of 32 x 64
    case 25: return launch_tn<16, 1, 1, 8, 0>(X, Wt, Y, M, K, N);         // 64 x 64 x 16, 4 waves of 32 x 32, 8 workgroups / CU
    case 26: return launch_tn<16, 2, 1, 4, 0>(X, Wt, Y, M, K, N);         // 128 x 64 x 16, 4 workgroups / CU (128 registers)
    case 27: return launch_tn<32, 2, 1, 2, 0>(X, Wt, Y, M, K, N);         // 128 x 64 x 32, 2 workgroups / CU (55 KB LDS)
    case 28: return launch_tn<16, 2, 1, 5, 1>(X, Wt, Y, M, K, N);         // 128 x 64 x 16 5wg, no global loads in the loop
    case 29: return launch_tn<16, 2, 1, 5, 3>(X, Wt, Y, M, K, N);         // 128 x 64 x 16 5wg, MFMA only
    // round 6: is it the L2 CHANNELS?  rows of both operands are a power of two apart (K * 4 bytes): (a) row stride K + 32 floats
    // (data differs: no check), (b) K loop rotated per workgroup
    case 30: return launch_tn<16, 2, 2, 3, 0>(X, Wt, Y, M, K, N, 32);     // 128 x 128 x 16 3wg, row stride K + 32
    case 31: return launch_tn<16, 2, 2, 3, 7>(X, Wt, Y, M, K, N);         // 128 x 128 x 16 3wg, rotated K loop
    case 32: return launch_tn<16, 2, 1, 5, 0>(X, Wt, Y, M, K, N, 32);     // 128 x 64 x 16 5wg, row stride K + 32
    case 33: return launch_tn<16, 2, 1, 5, 7>(X, Wt, Y, M, K, N);         // 128 x 64 x 16 5wg, rotated K loop
    case 34: return launch_tn<32, 2, 2, 2, 0>(X, Wt, Y, M, K, N, 32);     // 128 x 128 x 32 2wg, row stride K + 32
    case 35: return launch_tn<32, 2, 2, 2, 7>(X, Wt, Y, M, K, N);         // 128 x 128 x 32 2wg, rotated K loop
    case 36: return launch_tn<16, 2, 2, 3, 0>(X, Wt, Y, M, K, N, 32, 0);  // 128 x 128 x 16 3wg, only the ACTIVATION rows padded
    case 37: return launch_tn<16, 2, 2, 3, 0>(X, Wt, Y, M, K, N, 0, 32);  // 128 x 128 x 16 3wg, only the WEIGHT rows padded
    case 38: return launch_tn<16, 2, 2, 3, 0>(X, Wt, Y, M, K, N, 0, 4);   //   ... weight rows padded by 4 floats (16 bytes)
    case 39: return launch_tn<16, 2, 2, 3, 0>(X, Wt, Y, M, K, N, 0, 16);  //   ... by 16 floats (64 bytes)
    default: return -1;
  }
}
extern "C" const char *lab_tn_name(int variant) {
  static const char *names[] = {"128x128x32 2wg/cu",          "128x128x32 no-gload",       "128x128x32 no-gload no-lds-write",
                                "128x128x32 mfma-only",       "128x128x16 4wg/cu",         "128x128x16 3wg/cu",
                                "256x128x32 1wg/cu",          "128x256x32 1wg/cu",         "256x256x16 1wg/cu",
                                "128x128x32 1wg/cu",          "128x128x32 no-gload-B",     "128x128x32 no-gload-A",
                                "128x128x32 2wg/cu 2-ahead",  "128x128x16 glds 3wg/cu",    "128x128x16 glds 4wg/cu",
                                "256x128x16 8 waves 2wg/cu",  "128x256x16 8 waves 2wg/cu", "256x256x16 16 waves 1wg/cu",
                                "128x128x16 persistent 3wg/cu", "128x128x16 persistent 2wg/cu",
                                "256x128x16 4 waves 2wg/cu",  "128x256x16 4 waves 2wg/cu", "256x256x16 4 waves 1wg/cu",
                                "128x64x16 5wg/cu",           "64x128x16 5wg/cu",          "64x64x16 8wg/cu",
                                "128x64x16 4wg/cu",           "128x64x32 2wg/cu",          "128x64x16 5wg/cu no-gload",
                                "128x64x16 5wg/cu mfma-only",
                                "128x128x16 3wg/cu stride K+32 no-check", "128x128x16 3wg/cu rotated-K",
                                "128x64x16 5wg/cu stride K+32 no-check",  "128x64x16 5wg/cu rotated-K",
                                "128x128x32 2wg/cu stride K+32 no-check", "128x128x32 2wg/cu rotated-K",
                                "128x128x16 3wg/cu X stride K+32 no-check", "128x128x16 3wg/cu W stride K+32 no-check",
                                "128x128x16 3wg/cu W stride K+4 no-check",  "128x128x16 3wg/cu W stride K+16 no-check"};
  return variant >= 0 && variant < 40 ? names[variant] : nullptr;
}
