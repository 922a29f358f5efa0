// pairwise.hip -- pair-wise feature-similarity distillation loss for gfx950.
//
// Reference: CriterionPairWiseforWholeFeatAfterPool.forward, utils/criterion.py:236-245, and
// L2 / similarity / sim_dis_compute, utils/utils.py:170-183:
//     P   = MaxPool2d(k=s=(kh,kw), pad 0, ceil_mode=True)(F)            (B, C, OH, OW), M = OH*OW
//     Fh  = P / (sqrt(sum_c P^2) + 1e-8)      (norm detached)
//     A   = einsum('icm,icn->imn', Fh, Fh)                               (B, M, M)
//     L   = sum((A_T - A_S)^2) / M^2 / B
// The reference runs ~15 stock launches and materialises A_S and A_T.  Here:
//   pool      one wave per (plane, output-row band); lanes own columns (coalesced row reads),
//             running (max, argmax) per column, then a short segmented reduce in LDS.
//   normalise one workgroup per (image, 64 nodes): channel L2 norm + scaled copy into a
//             zero-padded (B, C, ldm) buffer, ldm = M rounded up to 128 -> the GEMM tiles below
//             need no edge masks and all their loads are 16-byte aligned.  (An optional node-major copy
//             (B, ldm, ldc) is still offered by the entry; the backward GEMM no longer needs it.)
//   gram      G = Fh_T^T Fh_T - Fh_S^T Fh_S accumulated in ONE set of fp32 MFMA accumulators
//             (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate -- there is no
//             TF32/xf32 on gfx950 and bf16 would break the 1e-4 loss tolerance).  128x128 output
//             tile per 256-thread workgroup, 4 waves x (2x2) 32x32 MFMA tiles, K streamed through
//             a double-buffered LDS ring of k-major panels; both operands are rows of the same
//             channel-major matrix, so LDS reads are conflict-free ds_read_b32 with no transposes.
//             The epilogue squares and reduces the tile (A_S/A_T never exist) and stores G.
//   backward  dFh_S = -4 g/(M^2 B) * Fh_S G with the same tile routine (operands Fh_S, transposed on the way into
//             LDS, and G; node range split over workgroups with a fixed-order combine), scaled by the detached norm;
//             unpool scatters through the argmax while writing the dense gradient once (no zero-fill pass, no atomics).
// Bounds: pool/unpool/normalise are HBM-bound; gram is fp32-MFMA-bound for M >~ 100
// (2*B*M^2*(C_S+C_T) flop), launch/latency-bound at the reference default M = 9.
#include "skd_common.hpp"

namespace skd {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// =============================================================================================
// max-pool with argmax
// =============================================================================================
struct Cand {
  float v;
  int idx;
};
// PyTorch scan semantics: `if (val > maxval || isnan(val)) take` in row-major order
//  -> first maximum wins among ordinary values, the LAST NaN wins when any NaN is present.
__device__ __forceinline__ bool better(const Cand &a, const Cand &b) {
  const bool an = a.v != a.v, bn = b.v != b.v;
  if (an) return bn ? a.idx > b.idx : true;
  if (bn) return false;
  return a.v > b.v || (a.v == b.v && a.idx < b.idx);
}

constexpr int kPoolMaxW = 1024;  // columns per plane row handled by the wave-per-band kernel

template <int NJ>  // NJ = ceil(W / 64) column slots per lane
__global__ __launch_bounds__(kThreads) void maxpool_band_kernel(const float *__restrict__ x,
                                                               float *__restrict__ pooled,
                                                               int32_t *__restrict__ index,
                                                               int64_t planes, int H, int W, int kh,
                                                               int kw, int OH, int OW,
                                                               int bands_per_wave) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  float *colv = reinterpret_cast<float *>(smem_raw) + (size_t)wid * W;
  int *coli = reinterpret_cast<int *>(smem_raw + sizeof(float) * (size_t)kWavesPerWG * W) + (size_t)wid * W;

  const int groups = (OH + bands_per_wave - 1) / bands_per_wave;  // band groups per plane
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerWG + wid;
  const bool live = unit < planes * groups;
  const int64_t plane = live ? unit / groups : 0;
  const int g = live ? (int)(unit % groups) : 0;
  const float *px = x + plane * (int64_t)H * W;

  for (int bi = 0; bi < bands_per_wave; ++bi) {
    const int oh = g * bands_per_wave + bi;
    const bool on = live && oh < OH;
    if (on) {
      const int r0 = oh * kh;
      const int r1 = min(H, r0 + kh);
      Cand best[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        best[j].v = -INFINITY;
        best[j].idx = r0 * W + j * kWave + lane;
      }
#pragma unroll 4
      for (int r = r0; r < r1; ++r) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = j * kWave + lane;
          if (col < W) {
            const float v = px[(int64_t)r * W + col];
            if (v > best[j].v || v != v) {
              best[j].v = v;
              best[j].idx = r * W + col;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = j * kWave + lane;
        if (col < W) {
          colv[col] = best[j].v;
          coli[col] = best[j].idx;
        }
      }
    }
    __syncthreads();
    if (on) {
      for (int ow = lane; ow < OW; ow += kWave) {
        const int c0 = ow * kw, c1 = min(W, c0 + kw);
        Cand acc{colv[c0], coli[c0]};
        for (int c = c0 + 1; c < c1; ++c) {
          const Cand cand{colv[c], coli[c]};
          if (better(cand, acc)) acc = cand;
        }
        const int64_t o = plane * (int64_t)OH * OW + (int64_t)oh * OW + ow;
        pooled[o] = acc.v;
        if (index != nullptr) index[o] = acc.idx;
      }
    }
    __syncthreads();
  }
}

// fallback for very wide planes: one thread per output cell
__global__ __launch_bounds__(kThreads) void maxpool_cell_kernel(const float *__restrict__ x,
                                                               float *__restrict__ pooled,
                                                               int32_t *__restrict__ index,
                                                               int64_t planes, int H, int W, int kh,
                                                               int kw, int OH, int OW) {
  const int64_t total = planes * OH * OW;
  const int64_t o = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (o >= total) return;
  const int ow = (int)(o % OW);
  const int oh = (int)((o / OW) % OH);
  const int64_t plane = o / ((int64_t)OW * OH);
  const float *px = x + plane * (int64_t)H * W;
  const int r0 = oh * kh, r1 = min(H, r0 + kh), c0 = ow * kw, c1 = min(W, c0 + kw);
  float bv = -INFINITY;
  int bi = r0 * W + c0;
  for (int r = r0; r < r1; ++r)
    for (int c = c0; c < c1; ++c) {
      const float v = px[(int64_t)r * W + c];
      if (v > bv || v != v) {
        bv = v;
        bi = r * W + c;
      }
    }
  pooled[o] = bv;
  if (index != nullptr) index[o] = bi;
}

// dense un-pool: every input position is written exactly once
__global__ __launch_bounds__(kThreads) void maxunpool_kernel(const float *__restrict__ dpooled,
                                                            const int32_t *__restrict__ index,
                                                            float *__restrict__ dx, int H, int W,
                                                            int kh, int kw, int OH, int OW,
                                                            int64_t ldp) {
  const int64_t plane = blockIdx.y;
  const int HW = H * W;
  const float *dp = dpooled + plane * ldp;
  const int32_t *ix = index + plane * (int64_t)OH * OW;
  float *out = dx + plane * (int64_t)HW;
  for (int e = blockIdx.x * kThreads + threadIdx.x; e < HW; e += gridDim.x * kThreads) {
    const int h = e / W, w = e - h * W;
    const int m = (h / kh) * OW + (w / kw);
    out[e] = (ix[m] == e) ? dp[m] : 0.f;
  }
}

// =============================================================================================
// the same pool / un-pool for CHANNELS-LAST features (B, H, W, C), C a multiple of 4 (round 5)
// =============================================================================================
// The PSP features reach the criterion channels-last (both networks run NHWC inside NetModel).  Rounds 1-4 copied them to NCHW
// first (17 + 69 MB read + written per step) only so that maxpool_band_kernel could read planes.  Here a lane owns one channel
// QUAD: every load is an aligned 16-byte piece of a row that is contiguous over (w, c), the scan order per channel is PyTorch's
// (row-major, first maximum wins, a NaN wins over everything and the last NaN wins), and pooled values / argmax indices come
// out in the reference's planar (B, C, OH * OW) order with the flat NCHW index h * W + w -- bit-identical to the planar kernels.
struct Cand4 {
  float v[4];
  int idx[4];
};
__device__ __forceinline__ void scan4(Cand4 &b, const float4 x, int flat) {
  const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (xv[k] > b.v[k] || xv[k] != xv[k]) {
      b.v[k] = xv[k];
      b.idx[k] = flat;
    }
}

// LARGE windows (the reference default pools 65 x 65 to 3 x 3): one workgroup per (image, output cell, block of QB channel
// quads); its 256 lanes = QB quads x S column slots scan the window's rows, then the S candidates of a quad are merged in slot
// order with the planar kernel's tie rule.  grid: B * OH * OW * ceil(C4 / QB)
__global__ __launch_bounds__(kThreads) void maxpool_window_nhwc_kernel(const float *__restrict__ x, float *__restrict__ pooled,
                                                                      int32_t *__restrict__ index, int C4, int H, int W,
                                                                      int kh, int kw, int OH, int OW, int QB, int S) {
  __shared__ Cand4 cand[kThreads];
  const int nblk = (C4 + QB - 1) / QB;
  const int cb = (int)(blockIdx.x % nblk);
  const int m = (int)((blockIdx.x / nblk) % (OH * OW));
  const int b = (int)(blockIdx.x / ((unsigned)nblk * OH * OW));
  const int q = threadIdx.x % QB, s = threadIdx.x / QB;
  const int quad = cb * QB + q;
  const int oh = m / OW, ow = m - oh * OW;
  const int r0 = oh * kh, r1 = min(H, r0 + kh), c0 = ow * kw, c1 = min(W, c0 + kw);
  const bool on = s < S && quad < C4;
  Cand4 best;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    best.v[k] = -INFINITY;
    best.idx[k] = r0 * W + c0;
  }
  if (on) {
    // One row per trip, four 16-byte loads in flight per lane.  Three attempts to go deeper were MEASURED and lost (gpurun r05e / r05g /
    // r05k, profiles/r05k_pooling_variants.txt): fewer quads per workgroup for more workgroups (26 / 36 us instead of 19 / 27), two
    // rows per trip through register arrays with an order-independent compare (70 / 89 us), two rows per trip in plain registers
    // (36 / 63 us).  2.6 TB/s on the 69 MB teacher map is where this formulation stands.
    const float *px = x + (int64_t)b * H * W * C4 * 4 + (int64_t)quad * 4;
    for (int r = r0; r < r1; ++r) {
      const float *pr = px + (int64_t)r * W * C4 * 4;
#pragma unroll 4
      for (int c = c0 + s; c < c1; c += S) scan4(best, *reinterpret_cast<const float4 *>(pr + (int64_t)c * C4 * 4), r * W + c);
    }
  }
  cand[threadIdx.x] = best;
  __syncthreads();
  if (s == 0 && quad < C4) {
    for (int t = 1; t < S; ++t) {
      const Cand4 &o = cand[t * QB + q];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const Cand a{o.v[k], o.idx[k]}, cur{best.v[k], best.idx[k]};
        if (better(a, cur)) {
          best.v[k] = a.v;
          best.idx[k] = a.idx;
        }
      }
    }
    const int M = OH * OW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t o = ((int64_t)b * C4 * 4 + quad * 4 + k) * M + m;
      pooled[o] = best.v[k];
      if (index != nullptr) index[o] = best.idx[k];
    }
  }
}

// SMALL windows (many output cells): a workgroup owns 32 consecutive output cells x 8 channel quads of one image; a lane scans
// one (cell, quad) window with 16-byte loads, the (32 channels x 32 cells) results are turned through LDS and leave as 128-byte
// runs along the cell axis of the planar (B, C, OH * OW) outputs.  (The first version let every lane store its four channels
// straight away: 4-byte writes 4 * M * 4 bytes apart -- 1010 MB of write traffic for 139 MB at M = 4225, 393 us; gpurun r05c.)
// grid: B * ceil(M / 32) * ceil(C4 / 8)
__global__ __launch_bounds__(kThreads) void maxpool_cell_nhwc_kernel(const float *__restrict__ x, float *__restrict__ pooled,
                                                                    int32_t *__restrict__ index, int C4, int H, int W, int kh, int kw,
                                                                    int OH, int OW) {
  __shared__ float sv[32][33];
  __shared__ int si[32][33];
  const int M = OH * OW;
  const int nq = (C4 + 7) / 8, nm = (M + 31) / 32;
  const int qt = (int)(blockIdx.x % nq), mt = (int)((blockIdx.x / nq) % nm), b = (int)(blockIdx.x / ((unsigned)nq * nm));
  const int ql = threadIdx.x & 7, cl = threadIdx.x >> 3;
  const int quad = qt * 8 + ql, m = mt * 32 + cl;
  if (quad < C4 && m < M) {
    const int oh = m / OW, ow = m - oh * OW;
    const int r0 = oh * kh, r1 = min(H, r0 + kh), c0 = ow * kw, c1 = min(W, c0 + kw);
    Cand4 best;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      best.v[k] = -INFINITY;
      best.idx[k] = r0 * W + c0;
    }
    const float *px = x + (int64_t)b * H * W * C4 * 4 + (int64_t)quad * 4;
    for (int r = r0; r < r1; ++r)
      for (int c = c0; c < c1; ++c) scan4(best, *reinterpret_cast<const float4 *>(px + ((int64_t)r * W + c) * C4 * 4), r * W + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sv[ql * 4 + k][cl] = best.v[k];
      si[ql * 4 + k][cl] = best.idx[k];
    }
  }
  __syncthreads();
  const int ch_l = threadIdx.x >> 3, ch = qt * 32 + ch_l;
  if (ch >= C4 * 4) return;
  const int64_t row = ((int64_t)b * C4 * 4 + ch) * M;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cell = (threadIdx.x & 7) * 4 + k, mm = mt * 32 + cell;
    if (mm < M) {
      pooled[row + mm] = sv[ch_l][cell];
      if (index != nullptr) index[row + mm] = si[ch_l][cell];
    }
  }
}

// dense un-pool into a channels-last gradient: every (b, h, w, c) written exactly once, 16 bytes per lane
__global__ __launch_bounds__(kThreads) void maxunpool_nhwc_kernel(const float *__restrict__ dpooled, const int32_t *__restrict__ index,
                                                                 float *__restrict__ dx, int64_t total, int C4, int H, int W, int kh,
                                                                 int kw, int OH, int OW, int64_t ldp) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (t >= total) return;
  const int quad = (int)(t % C4);
  const int e = (int)((t / C4) % ((int64_t)H * W));
  const int b = (int)(t / ((int64_t)C4 * H * W));
  const int h = e / W, w = e - h * W;
  const int M = OH * OW, m = (h / kh) * OW + (w / kw);
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t plane = (int64_t)b * C4 * 4 + quad * 4 + k;
    o[k] = index[plane * M + m] == e ? dpooled[plane * ldp + m] : 0.f;
  }
  *reinterpret_cast<float4 *>(dx + t * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// =============================================================================================
// channel L2 norm + normalise into padded buffers
// =============================================================================================
// grid (ceil(ldm/64), B); block 256 = 64 nodes x 4 channel slices
__global__ __launch_bounds__(kThreads) void l2_normalise_kernel(const float *__restrict__ pooled,
                                                               float *__restrict__ fhat,
                                                               float *__restrict__ fhat_t,
                                                               float *__restrict__ norm, int C, int M,
                                                               int ldm, int ldc) {
  __shared__ float ssq[kWavesPerWG][kWave];
  const int lane = threadIdx.x & (kWave - 1), slice = threadIdx.x / kWave;
  const int b = blockIdx.y;
  const int m = blockIdx.x * kWave + lane;
  const bool valid = m < M;
  const float *src = pooled + (int64_t)b * C * M;
  float acc = 0.f;
  if (valid)
    for (int c = slice; c < C; c += kWavesPerWG) {
      const float v = src[(int64_t)c * M + m];
      acc += v * v;
    }
  ssq[slice][lane] = acc;
  __syncthreads();
  const float tot = (ssq[0][lane] + ssq[1][lane]) + (ssq[2][lane] + ssq[3][lane]);
  const float nrm = sqrtf(tot) + 1e-8f;  // utils.py:170-171
  if (valid && slice == 0 && norm != nullptr) norm[(int64_t)b * M + m] = nrm;
  if (m < ldm) {
    float *dst = fhat + (int64_t)b * C * ldm;
    for (int c = slice; c < C; c += kWavesPerWG)
      dst[(int64_t)c * ldm + m] = valid ? src[(int64_t)c * M + m] / nrm : 0.f;  // utils.py:176
  }
  if (fhat_t != nullptr && m < ldm) {
    // node-major copy (B, ldm, ldc), zero padded in both directions
    float *dst = fhat_t + ((int64_t)b * ldm + m) * ldc;
    for (int c = slice; c < ldc; c += kWavesPerWG)
      dst[c] = (valid && c < C) ? src[(int64_t)c * M + m] / nrm : 0.f;
  }
}

// =============================================================================================
// 128x128 fp32 MFMA tile:  acc[i][j] += sign * sum_k A(k, i) * B(k, j)
//
// Round-3 rewrite (profiles/r02h_pmc.json: matrix pipe 0.52 busy in the Gram kernel, 0.31 in the
// backward).  The disassembly of the round-2 routine showed where the other half went:
//   * the K segments lived in a dynamically indexed array -> SCRATCH memory; every K-tile re-read its descriptor with
//     scratch loads and waited (s_waitcnt vmcnt(0)) before each of its four global loads, serialising
//     4 x (scratch + HBM latency) in front of every 4096 MFMA cycles;
//   * every k-step was "ds_read; s_waitcnt lgkmcnt(0); 4 MFMA": the wave parked for the LDS latency sixteen times per K-tile;
//   * the teacher / student sign was a v_cndmask + s_nop in front of every MFMA pair.
// Now: segments are plain scalars selected with uniform branches (no arrays, global_load with SGPR bases), the sign is
// applied once per element when a panel is stored to LDS, the LDS operands of k-step s+1 are fetched before the MFMAs
// of k-step s are issued (software pipelining in registers), and full K-tiles take a guard-free path.
// The backward GEMM needs Fhat_S node-major.  Round 2 had channel_l2_normalise write that copy with uncoalesced 4-byte stores
// (1.9x its algorithmic traffic); rounds 3-4 transposed the channel-major panel on the way into LDS instead (scalar ds_write_b32
// at a 129-float stride) and the backward sat at 0.53 of the fp32 MFMA peak while the Gram kernel, same tiles, reached 0.63;
// round 5 makes the copy with a coalesced 32 x 32 LDS transpose inside skd_pairwise_backward (18 MB beside 606 MB of G at
// M = 4225) so that BOTH GEMMs run the same k-major main loop.  Its K = ldm contraction is split over gridDim.z workgroups with
// a fixed-order combine.
// =============================================================================================
constexpr int kTile = 128;
constexpr int kBK = 32;             // K-tile: 16 k-steps of v_mfma_f32_32x32x2_f32
// TI = rows of the output tile (columns of the k-major A panel): 128 (wave tile 64 x 64 = 2 x 2 MFMA blocks) or -- round 6, for
// graphs whose 128 x 128 tiles do not fill the chip (M = 1089: 360 workgroups on 512 slots) -- 64 (wave tile 32 x 64 = 1 x 2).
template <int TI>
struct Panels {
  static constexpr int offB = kBK * TI;                   // B panel follows the A panel
  static constexpr int stage = kBK * (TI + kTile);        // floats per pipeline stage
  static constexpr size_t lds_bytes = sizeof(float) * 2 * stage;   // double buffered: 64 KiB (2 workgroups / CU) or 48 KiB (3)
};

// One K segment, both operands k-major: A(k, i) at A[k * lda + i], B(k, j) at B[k * ldb + j]; rows k >= K read as zero.
// All tile-local: i in [0, TI), j in [0, 128).  (Rounds 3-4 also staged an i-major A through a transposing store for the backward
// GEMM; since round 5 the backward reads a node-major copy instead -- transpose_pad_kernel -- and that path is gone.)
struct Seg {
  const float *A, *B;
  int lda, ldb, K;
  float sign;
};

template <int TI>
struct TileRegs {
  float4 a[TI / 32], b[4];
};

// (the segment arrives as scalars BY VALUE: a `const Seg &` picked with `first ? s0 : s1` made the compiler keep both structs
// in scratch memory and re-load the fields -- with a wait -- in front of every K-tile)
template <int TI>
__device__ __forceinline__ void panel_load(const float *__restrict__ sA, const float *__restrict__ sB, int lda, int ldb, int K,
                                           int k0, TileRegs<TI> &r) {
  const int t = threadIdx.x;
  const int row = t >> 5, c4 = (t & 31) * 4;            // B: 8 panel rows x 128 columns per pass
  constexpr int AQ = TI / 4;                             // float4 per A panel row (32 or 16)
  constexpr int AR = kThreads / AQ;                      // A panel rows per pass (8 or 16)
  constexpr int AH = kBK / AR;                           // passes (4 or 2)
  const int arow = t / AQ, ac4 = (t % AQ) * 4;
  if (k0 + kBK <= K) {                                   // uniform: full K-tile, no guards
#pragma unroll
    for (int h = 0; h < 4; ++h) r.b[h] = *reinterpret_cast<const float4 *>(sB + (int64_t)(k0 + row + 8 * h) * ldb + c4);
#pragma unroll
    for (int h = 0; h < AH; ++h) r.a[h] = *reinterpret_cast<const float4 *>(sA + (int64_t)(k0 + arow + AR * h) * lda + ac4);
  } else {                                               // last, partial K-tile of a segment: rows k >= K are zero
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int k = k0 + row + 8 * h;
      r.b[h] = zero;
      if (k < K) r.b[h] = *reinterpret_cast<const float4 *>(sB + (int64_t)k * ldb + c4);
    }
#pragma unroll
    for (int h = 0; h < AH; ++h) {
      const int k = k0 + arow + AR * h;
      r.a[h] = zero;
      if (k < K) r.a[h] = *reinterpret_cast<const float4 *>(sA + (int64_t)k * lda + ac4);
    }
  }
}

template <int TI>
__device__ __forceinline__ void panel_store(float *st, const TileRegs<TI> &r, float sign) {
  const int t = threadIdx.x;
  const int row = t >> 5, c4 = (t & 31) * 4;
  constexpr int AQ = TI / 4, AR = kThreads / AQ, AH = kBK / AR;
  const int arow = t / AQ, ac4 = (t % AQ) * 4;
#pragma unroll
  for (int h = 0; h < 4; ++h) *reinterpret_cast<float4 *>(st + Panels<TI>::offB + (row + 8 * h) * kTile + c4) = r.b[h];
#pragma unroll
  for (int h = 0; h < AH; ++h) {
    float4 v = r.a[h];
    v.x *= sign; v.y *= sign; v.z *= sign; v.w *= sign;
    *reinterpret_cast<float4 *>(st + (arow + AR * h) * TI + ac4) = v;
  }
}

// 16 k-steps on one LDS stage.  Lane (l31 = lane & 31, kk = lane >> 5) feeds column / row l31 of the 32-wide operand
// slices with k = 2 * ks + kk; the operands of step ks + 1 are read before the MFMAs of step ks are issued.
// Waves: 2 x 2; wave tile (TI / 2) x 64 = WM x 2 MFMA blocks, WM = TI / 64.
template <int TI>
__device__ __forceinline__ void tile_compute(const float *st, f32x16 (&acc)[TI / 64][2]) {
  constexpr int WM = TI / 64;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int wi = (wid >> 1) * (TI / 2), wj = (wid & 1) * 64;
  const int kk = lane >> 5, l31 = lane & 31;
  const float *pa = st + kk * TI + wi + l31;
  const float *pb = st + Panels<TI>::offB + kk * kTile + wj + l31;
  float a[WM], b0 = pb[0], b1 = pb[32];
#pragma unroll
  for (int i = 0; i < WM; ++i) a[i] = pa[32 * i];
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // the LDS reads (ds_read2_b32) of step 0
#pragma unroll
  for (int ks = 0; ks < kBK / 2; ++ks) {
    float na[WM], nb0 = 0.f, nb1 = 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i) na[i] = 0.f;
    if (ks + 1 < kBK / 2) {
#pragma unroll
      for (int i = 0; i < WM; ++i) na[i] = pa[(ks + 1) * 2 * TI + 32 * i];
      nb0 = pb[(ks + 1) * 2 * kTile];
      nb1 = pb[(ks + 1) * 2 * kTile + 32];
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b0, acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b1, acc[i][1], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = na[i];
    b0 = nb0; b1 = nb1;
    // pin the issue order: the first MFMAs of step ks, the LDS reads (ds_read2_b32) of step ks + 1, the other MFMAs.
    // The compiler waits with a full lgkmcnt(0) before step ks + 1 whatever is in flight, so the reads must be OLD by
    // then: issued in the middle of the MFMA group they have >= 128 cycles of matrix-pipe time to land (left to itself
    // the scheduler put them right in front of the wait: one LDS round trip parked per MFMA group, 16 per K-tile).
    __builtin_amdgcn_sched_group_barrier(0x008, WM, 0);                         // MFMA
    if (ks + 1 < kBK / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
    __builtin_amdgcn_sched_group_barrier(0x008, WM, 0);                         // MFMA
  }
}

// Runs segment s0 then (NSEG == 2) segment s1 through the double-buffered LDS ring into one accumulator set.
template <int NSEG, int TI = kTile>
__device__ __forceinline__ void tile_gemm(const Seg s0, const Seg s1, float *lds, f32x16 (&acc)[TI / 64][2]) {
#pragma unroll
  for (int i = 0; i < TI / 64; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int n0 = (s0.K + kBK - 1) / kBK;
  const int n1 = NSEG == 2 ? (s1.K + kBK - 1) / kBK : 0;
  const int total = n0 + n1;
  if (total == 0) return;
  constexpr int kStage = Panels<TI>::stage;
  TileRegs<TI> regs;
  {
    const bool f = n0 > 0;
    panel_load<TI>(f ? s0.A : s1.A, f ? s0.B : s1.B, f ? s0.lda : s1.lda, f ? s0.ldb : s1.ldb, f ? s0.K : s1.K, 0, regs);
    panel_store<TI>(lds, regs, f ? s0.sign : s1.sign);
  }
  __syncthreads();
  int stage = 0;
  for (int it = 0; it < total; ++it) {
    const int nx = it + 1;
    const bool more = nx < total;
    const bool first = nx < n0;                         // uniform: which segment the NEXT K-tile belongs to
    if (more)                                           // scalar selects of by-value fields: everything stays in SGPRs
      panel_load<TI>(first ? s0.A : s1.A, first ? s0.B : s1.B, first ? s0.lda : s1.lda, first ? s0.ldb : s1.ldb,
                     first ? s0.K : s1.K, first ? nx * kBK : (nx - n0) * kBK, regs);
    tile_compute<TI>(lds + stage * kStage, acc);
    if (more) panel_store<TI>(lds + (stage ^ 1) * kStage, regs, first ? s0.sign : s1.sign);
    __syncthreads();
    stage ^= 1;
  }
}

// C/D fragment coordinates of v_mfma_f32_32x32x2_f32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// gram + loss.  G is symmetric, so only the nt*(nt+1)/2 tiles with ti <= tj are computed: an off-diagonal
// tile counts twice in the loss and is stored twice (as is, and transposed into the mirror position, 16 bytes
// per lane per store: the C/D fragment holds 4 consecutive rows per register quad).
// grid = (nt*(nt+1)/2 * (128 / TI), B): blockIdx.x -> (ti, tj) by walking the rows of the upper triangle; TI = 64: two workgroups
// per 128 x 128 tile (its upper and its lower 64 rows), twice the workgroups of half the size for graphs that do not fill the chip.
template <int TI>
__global__ __launch_bounds__(kThreads, TI == kTile ? 2 : 3) void gram_loss_kernel(const float *__restrict__ fs,
                                                                const float *__restrict__ ft,
                                                                float *__restrict__ G,
                                                                float *__restrict__ part, int Cs,
                                                                int Ct, int ldm, int nt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // Panels<TI>::lds_bytes
  __shared__ float red[2 * kWavesPerWG];
  constexpr int HALVES = kTile / TI, WM = TI / 64;
  const int b = blockIdx.y;
  const int half = HALVES == 1 ? 0 : (int)(blockIdx.x % HALVES);
  int ti = 0, rem = blockIdx.x / HALVES;
  while (rem >= nt - ti) {
    rem -= nt - ti;
    ++ti;
  }
  const int tj = ti + rem;
  const int i0 = ti * kTile + half * TI, j0 = tj * kTile;
  Seg st_, ss_;                                          // teacher (+), student (-): one accumulator set
  st_.A = ft + (int64_t)b * Ct * ldm + i0;
  st_.B = ft + (int64_t)b * Ct * ldm + j0;
  st_.lda = st_.ldb = ldm;
  st_.K = Ct;
  st_.sign = 1.f;
  ss_.A = fs + (int64_t)b * Cs * ldm + i0;
  ss_.B = fs + (int64_t)b * Cs * ldm + j0;
  ss_.lda = ss_.ldb = ldm;
  ss_.K = Cs;
  ss_.sign = -1.f;
  f32x16 acc[WM][2];
  tile_gemm<2, TI>(st_, ss_, lds, acc);

  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int wi = (wid >> 1) * (TI / 2), wj = (wid & 1) * 64;
  const bool mirror = ti != tj;
  float *Gb = G != nullptr ? G + (int64_t)b * ldm * ldm : nullptr;
  float sq = 0.f, unused = 0.f;
#pragma unroll
  for (int bi = 0; bi < WM; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      const int col = j0 + wj + bj * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row0 = i0 + wi + bi * 32 + 8 * q + 4 * (lane >> 5);  // rows row0 .. row0+3 = regs 4q .. 4q+3
        float4 v;
        v.x = acc[bi][bj][4 * q + 0];
        v.y = acc[bi][bj][4 * q + 1];
        v.z = acc[bi][bj][4 * q + 2];
        v.w = acc[bi][bj][4 * q + 3];
        sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);  // padded rows/cols are exactly zero
        if (Gb != nullptr) {
          Gb[(int64_t)(row0 + 0) * ldm + col] = v.x;
          Gb[(int64_t)(row0 + 1) * ldm + col] = v.y;
          Gb[(int64_t)(row0 + 2) * ldm + col] = v.z;
          Gb[(int64_t)(row0 + 3) * ldm + col] = v.w;
          if (mirror) *reinterpret_cast<float4 *>(Gb + (int64_t)col * ldm + row0) = v;  // G[col][row0..row0+3]
        }
      }
    }
  if (mirror) sq *= 2.f;
  block_sum2(sq, unused, red);
  if (threadIdx.x == 0) part[(int64_t)b * gridDim.x + blockIdx.x] = sq;
}

// node-major, zero-padded copy of the student panel for the backward GEMM: ft[b][n][c] = fs[b][c][n] (c >= Cs: 0), ldc a multiple
// of 128.  32 x 32 tiles through LDS; grid (ldm / 32, ldc / 32, B), block 256 = 32 x 8.
__global__ __launch_bounds__(kThreads) void transpose_pad_kernel(const float *__restrict__ fs, float *__restrict__ ft, int Cs, int ldm,
                                                                int ldc) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int b = blockIdx.z, n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ty + 8 * r;
    tile[ty + 8 * r][tx] = c < Cs ? fs[((int64_t)b * Cs + c) * ldm + n0 + tx] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) ft[((int64_t)b * ldm + n0 + ty + 8 * r) * ldc + c0 + tx] = tile[tx][ty + 8 * r];
}

// backward: D[c][m] = sum_n Fhat_S[c][n] * G[n][m];  dP[c][m] = coef * gscale * D[c][m] / norm[m]
// Round 5: the A operand comes from a NODE-MAJOR copy of the student panel (transpose_pad_kernel, 18 MB at M = 4225 next to the
// 606 MB of G), so both operands are k-major and the contraction runs through the very main loop of the Gram kernel.
//
// Round 6: STREAM-K over (output tile, K-tile) units instead of a fixed split of the node range.  The problem has few output
// tiles (Cs = 128 -> ONE row of 34 tiles per image at M = 4225: 272 in a batch of 8) and a long contraction (136 K-tiles); rounds
// 3-5 cut every tile's node range into 4 fixed pieces: 1088 workgroups on the chip's 512 slots (two 64 KB-LDS workgroups per CU)
// = 2.1 rounds, the last one a sixth full, and 71 MB of partials written and read back (0.50-0.56 of the fp32 MFMA peak against
// the Gram kernel's 0.73 with the same main loop).  Now the 272 x 136 = 36 992 units are dealt
// out as ONE contiguous range per workgroup, `nwg` = one round of the chip (2 per CU) -- every slot gets 72 or 73 K-tiles, nobody
// waits for a straggler round.  A workgroup's range touches at most a head tile and a tail tile it does not own alone (their raw
// 128 x 128 accumulators go to slot [w][0] / [w][1]) and whole tiles in between (scaled and written straight to dpooled).
// pairwise_bwd_fixup_kernel then adds, per output tile, the slots of the workgroups that covered it IN WORKGROUP ORDER
// (deterministic, bit-reproducible) and applies the scale.
struct StreamK {
  int64_t total;       // units = tiles * nkt
  int nwg, nkt;        // workgroups of the launch, K-tiles per output tile
};
__host__ __device__ __forceinline__ int64_t sk_first_unit(const StreamK &k, int w) { return (int64_t)w * k.total / k.nwg; }
// the workgroup whose range contains unit u: the largest w with sk_first_unit(w) <= u
__host__ __device__ __forceinline__ int sk_owner(const StreamK &k, int64_t u) { return (int)(((u + 1) * k.nwg - 1) / k.total); }

template <int TI>      // rows (channels) of an output tile: 128, or 64 for problems that do not fill the chip with 128-row tiles
__global__ __launch_bounds__(kThreads, TI == kTile ? 2 : 3) void pairwise_bwd_kernel(const float *__restrict__ ft,
                                                                   const float *__restrict__ G,
                                                                   const float *__restrict__ norm,
                                                                   const float *__restrict__ gscale,
                                                                   float *__restrict__ dpooled,
                                                                   float *__restrict__ part, int Cs,
                                                                   int M, int ldm, int ldc, int ntm, int ntc, StreamK sk,
                                                                   float coef) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // Panels<kTile>::lds_bytes
  const int w = blockIdx.x;
  const int64_t u0 = sk_first_unit(sk, w), u1 = sk_first_unit(sk, w + 1);
  constexpr int WM = TI / 64;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int wi = (wid >> 1) * (TI / 2), wj = (wid & 1) * 64;
  const float scale = coef * gscale[0];
  const int tiles_per_image = ntm * ntc;
  bool first = true;
  for (int64_t u = u0; u < u1;) {
    const int64_t t = u / sk.nkt;
    const int kt0 = (int)(u - t * sk.nkt);
    const int64_t tile_end = (t + 1) * sk.nkt;
    const int kt1 = (int)((u1 < tile_end ? u1 : tile_end) - t * sk.nkt);
    const int b = (int)(t / tiles_per_image), rem = (int)(t - (int64_t)b * tiles_per_image);
    const int tc = rem / ntm, tm = rem - tc * ntm;
    const int c0 = tc * TI, m0 = tm * kTile;
    Seg s;
    s.A = ft + ((int64_t)b * ldm + (int64_t)kt0 * kBK) * ldc + c0;     // A(k = n, i = c) = Fhat_S[c][n], node-major copy: k-major
    s.lda = ldc;
    s.B = G + ((int64_t)b * ldm + (int64_t)kt0 * kBK) * ldm + m0;      // B(k = n, j = m) = G[n][m]
    s.ldb = ldm;
    s.K = (kt1 - kt0) * kBK;
    s.sign = 1.f;
    f32x16 acc[WM][2];
    tile_gemm<1, TI>(s, s, lds, acc);
    __syncthreads();                                   // the LDS ring is reused by the next segment's first panel
    const bool whole = kt0 == 0 && kt1 == sk.nkt;
    if (whole) {
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const int m = m0 + wj + bj * 32 + (lane & 31);
        const float inv = m < M ? scale / norm[(int64_t)b * M + m] : 0.f;
#pragma unroll
        for (int bi = 0; bi < WM; ++bi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = c0 + wi + bi * 32 + frag_row(r, lane);
            if (c < Cs) dpooled[((int64_t)b * Cs + c) * ldm + m] = acc[bi][bj][r] * inv;
          }
      }
    } else {
      // raw partial of a shared tile: slot 0 = this workgroup's first segment, slot 1 = a later (its last) one; (TI, 128) row-major
      float *dst = part + ((int64_t)w * 2 + (first ? 0 : 1)) * (TI * kTile);
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const int ml = wj + bj * 32 + (lane & 31);
#pragma unroll
        for (int bi = 0; bi < WM; ++bi)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[(wi + bi * 32 + frag_row(r, lane)) * kTile + ml] = acc[bi][bj][r];
      }
    }
    first = false;
    u = t * sk.nkt + kt1;
  }
}

// Four workgroups per output tile (32 rows each): dpooled[b][c][m] = coef * gscale / norm[b][m] * sum over the covering
// workgroups' slots, in workgroup order (m >= M: 0).  A tile one workgroup computed alone was written by pairwise_bwd_kernel.
__global__ __launch_bounds__(kThreads) void pairwise_bwd_fixup_kernel(const float *__restrict__ part,
                                                                     const float *__restrict__ norm,
                                                                     const float *__restrict__ gscale,
                                                                     float *__restrict__ dpooled, int Cs, int M, int ldm,
                                                                     int ntm, int ntc, StreamK sk, float coef, int TI) {
  const int64_t t = blockIdx.x;
  const int64_t ub = t * sk.nkt, ue = ub + sk.nkt;
  const int wf = sk_owner(sk, ub), wl = sk_owner(sk, ue - 1);
  if (wf == wl) return;                                  // one workgroup owned the whole tile
  const int tiles_per_image = ntm * ntc;
  const int b = (int)(t / tiles_per_image), rem = (int)(t - (int64_t)b * tiles_per_image);
  const int tc = rem / ntm, tm = rem - tc * ntm;
  const int c0 = tc * TI, m0 = tm * kTile;
  const float scale = coef * gscale[0];
  const float *nr = norm + (int64_t)b * M;
  const int per = TI * kTile / 16;                       // float4 per quarter of the tile
  for (int q = blockIdx.y * per + threadIdx.x; q < (blockIdx.y + 1) * per; q += kThreads) {
    const int cl = q >> 5, ml = (q & 31) * 4;            // 32 float4 per tile row
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = wf; w <= wl; ++w) {
      // tile t is workgroup w's FIRST segment iff its range starts inside t
      const int slot = sk_first_unit(sk, w) >= ub ? 0 : 1;
      const float4 v = *reinterpret_cast<const float4 *>(part + ((int64_t)w * 2 + slot) * ((int64_t)TI * kTile) + cl * kTile + ml);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int c = c0 + cl, m = m0 + ml;
    if (c >= Cs) continue;
    float4 o;
    o.x = m + 0 < M ? s.x * (scale / nr[m + 0]) : 0.f;
    o.y = m + 1 < M ? s.y * (scale / nr[m + 1]) : 0.f;
    o.z = m + 2 < M ? s.z * (scale / nr[m + 2]) : 0.f;
    o.w = m + 3 < M ? s.w * (scale / nr[m + 3]) : 0.f;
    *reinterpret_cast<float4 *>(dpooled + ((int64_t)b * Cs + c) * ldm + m) = o;
  }
}


// =============================================================================================
// Small graphs (M <= 64 nodes): the whole similarity loss of one image in ONE workgroup.
// The reference's default --pool-scale 0.5 pools the 65 x 65 feature map to 3 x 3 = 9 nodes (criterion.py:241-244);
// through the MFMA path above those 9 nodes were padded to a 128-wide tile and cost five launches (two normalise,
// Gram, partial sum, backward GEMM: ~0.3 ms per step for an 81-entry matrix).  Here one 256-thread workgroup per
// image keeps everything on chip: channel norms -> normalised features in LDS, chunk by chunk -> both Gram matrices
// in registers (M^2 <= 4096 pairs, 16 per thread) -> G = A_T - A_S in LDS -> loss partial and, when asked,
// dL/dP = -4 / (M^2 B) * (Fhat_S G) / norm_S.  fp32 FMAs in a fixed order (no MFMA: K <= 512 and M <= 64 is 2.6 MFLOP
// per image), deterministic.
// =============================================================================================
constexpr int kSmallM = 64;
constexpr int kSmallChunk = 32;   // channels per LDS chunk

__global__ __launch_bounds__(kThreads) void pairwise_small_kernel(const float *__restrict__ ps, const float *__restrict__ pt,
                                                                 float *__restrict__ part, float *__restrict__ dpooled,
                                                                 int Cs, int Ct, int M, float coef) {
  __shared__ float nrm[2][kSmallM];                  // [0] student, [1] teacher: sqrt(sum_c f^2) + 1e-8
  __shared__ float tile[kSmallChunk][kSmallM + 1];
  __shared__ float Gs[kSmallM][kSmallM + 1];
  __shared__ float red[2 * kWavesPerWG];
  __shared__ float ssq[kThreads];
  const int b = blockIdx.x, t = threadIdx.x;
  const float *S = ps + (int64_t)b * Cs * M, *T = pt + (int64_t)b * Ct * M;
  // ---- channel norms (utils.py:170-171) ----
  const int lanes = kThreads / kSmallM;              // 4 channel slices per node
  for (int which = 0; which < 2; ++which) {
    const float *F = which == 0 ? S : T;
    const int C = which == 0 ? Cs : Ct;
    const int m = t % kSmallM, sl = t / kSmallM;
    float a = 0.f;
    if (m < M)
      for (int c = sl; c < C; c += lanes) {
        const float v = F[(int64_t)c * M + m];
        a += v * v;
      }
    ssq[t] = a;
    __syncthreads();
    if (t < M) nrm[which][t] = sqrtf((ssq[t] + ssq[t + kSmallM]) + (ssq[t + 2 * kSmallM] + ssq[t + 3 * kSmallM])) + 1e-8f;
    __syncthreads();
  }
  // ---- Gram matrices: thread owns pairs p = t, t + 256, ... of the M x M matrix ----
  float acc[kSmallM * kSmallM / kThreads];
#pragma unroll
  for (int k = 0; k < kSmallM * kSmallM / kThreads; ++k) acc[k] = 0.f;
  const int npairs = M * M;
  for (int which = 1; which >= 0; --which) {         // teacher (+), then student (-)
    const float *F = which == 0 ? S : T;
    const int C = which == 0 ? Cs : Ct;
    const float sign = which == 0 ? -1.f : 1.f;
    for (int c0 = 0; c0 < C; c0 += kSmallChunk) {
      for (int e = t; e < kSmallChunk * M; e += kThreads) {
        const int c = e / M, m = e - c * M;
        tile[c][m] = c0 + c < C ? F[(int64_t)(c0 + c) * M + m] / nrm[which][m] : 0.f;   // utils.py:176
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kSmallM * kSmallM / kThreads; ++k) {
        const int p = t + k * kThreads;
        if (p < npairs) {
          const int i = p / M, j = p - i * M;
          float a = 0.f;
#pragma unroll 8
          for (int c = 0; c < kSmallChunk; ++c) a += tile[c][i] * tile[c][j];
          acc[k] += sign * a;
        }
      }
      __syncthreads();
    }
  }
  // ---- G = A_T - A_S, loss partial (utils.py:178-183) ----
  float sq = 0.f, unused = 0.f;
#pragma unroll
  for (int k = 0; k < kSmallM * kSmallM / kThreads; ++k) {
    const int p = t + k * kThreads;
    if (p < npairs) {
      Gs[p / M][p % M] = acc[k];
      sq += acc[k] * acc[k];
    }
  }
  block_sum2(sq, unused, red);       // contains a __syncthreads(): Gs is complete afterwards
  if (t == 0) part[b] = sq;
  if (dpooled == nullptr) return;
  __syncthreads();
  // ---- dP[c][m] = coef * sum_n Fhat_S[c][n] G[n][m] / norm_S[m] ----
  float *out = dpooled + (int64_t)b * Cs * M;
  for (int e = t; e < Cs * M; e += kThreads) {
    const int c = e / M, m = e - c * M;
    float a = 0.f;
    for (int n = 0; n < M; ++n) a += (S[(int64_t)c * M + n] / nrm[0][n]) * Gs[n][m];
    out[e] = coef * a / nrm[0][m];
  }
}

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

int skd_maxpool_argmax(int planes, int H, int W, int kh, int kw, const float *x, float *pooled,
                       int32_t *index, skd_stream_t stream) {
  if (planes <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || !x || !pooled) return 0;
  if ((int64_t)H * W >= (int64_t)1 << 31) return 0;
  hipStream_t st = as_stream(stream);
  const int OH = (int)cdiv(H, kh), OW = (int)cdiv(W, kw);
  if (W <= kPoolMaxW) {
    int bpw = kh >= 16 ? 1 : (int)cdiv(16, kh);
    if (bpw > OH) bpw = OH;
    const int groups = (int)cdiv(OH, bpw);
    const int64_t units = (int64_t)planes * groups;
    const dim3 grid((unsigned)cdiv(units, kWavesPerWG)), block(kThreads);
    const size_t smem = (size_t)kWavesPerWG * W * (sizeof(float) + sizeof(int));
    const int nj = (int)cdiv(W, kWave);
#define SKD_POOL(NJ) \
  maxpool_band_kernel<NJ><<<grid, block, smem, st>>>(x, pooled, index, planes, H, W, kh, kw, OH, OW, bpw)
    if (nj <= 1) SKD_POOL(1);
    else if (nj <= 2) SKD_POOL(2);
    else if (nj <= 4) SKD_POOL(4);
    else if (nj <= 8) SKD_POOL(8);
    else SKD_POOL(16);
#undef SKD_POOL
  } else {
    const int64_t total = (int64_t)planes * OH * OW;
    maxpool_cell_kernel<<<dim3((unsigned)cdiv(total, kThreads)), dim3(kThreads), 0, st>>>(
        x, pooled, index, planes, H, W, kh, kw, OH, OW);
  }
  return ok();
}

int skd_maxunpool_scatter(int planes, int H, int W, int kh, int kw, const float *dpooled, int64_t ldp,
                          const int32_t *index, float *dx, skd_stream_t stream) {
  if (planes <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || !dpooled || !index || !dx) return 0;
  if (planes > 65535 * 1024) return 0;
  const int OH = (int)cdiv(H, kh), OW = (int)cdiv(W, kw);
  int gx = (int)cdiv((int64_t)H * W, kThreads);
  if (gx > 64) gx = 64;
  // gridDim.y is limited to 65535: fold planes in chunks
  hipStream_t st = as_stream(stream);
  for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
    const int np = (int)((planes - p0) < 65535 ? (planes - p0) : 65535);
    maxunpool_kernel<<<dim3(gx, np), dim3(kThreads), 0, st>>>(
        dpooled + p0 * ldp, index + p0 * (int64_t)OH * OW, dx + p0 * (int64_t)H * W, H, W, kh, kw, OH, OW, ldp);
  }
  return ok();
}

// Channels-last variants (include/skd.h section 4): x (B, H, W, C), C % 4 == 0, 16-byte aligned; pooled / index (B, C, OH * OW) in the
// reference's planar order with the flat NCHW index; dx (B, H, W, C).
int skd_maxpool_argmax_nhwc(int B, int C, int H, int W, int kh, int kw, const float *x, float *pooled, int32_t *index,
                            skd_stream_t stream) {
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || !x || !pooled) return 0;
  if ((int64_t)H * W >= (int64_t)1 << 31 || (reinterpret_cast<uintptr_t>(x) & 15)) return 0;
  hipStream_t st = as_stream(stream);
  const int OH = (int)cdiv(H, kh), OW = (int)cdiv(W, kw), C4 = C / 4;
  const int wh = kh < H ? kh : H, ww = kw < W ? kw : W;
  if ((int64_t)wh * ww >= 64) {
    // (fewer quads per workgroup for more workgroups on the student's small map was measured and is WORSE -- 26 instead of 19 us:
    // 128-byte row pieces and a longer merge; gpurun r05e)
    const int QB = C4 < 32 ? C4 : 32;
    int S = kThreads / QB;
    if (S > ww) S = ww;                                   // no more column slots than window columns
    const int64_t grid = (int64_t)B * OH * OW * cdiv(C4, QB);
    if (grid > 2147483647) return 0;
    maxpool_window_nhwc_kernel<<<dim3((unsigned)grid), dim3(kThreads), 0, st>>>(x, pooled, index, C4, H, W, kh, kw, OH, OW, QB, S);
  } else {
    const int64_t grid = (int64_t)B * cdiv((int64_t)OH * OW, 32) * cdiv(C4, 8);
    if (grid > 2147483647) return 0;
    maxpool_cell_nhwc_kernel<<<dim3((unsigned)grid), dim3(kThreads), 0, st>>>(x, pooled, index, C4, H, W, kh, kw, OH, OW);
  }
  return ok();
}

int skd_maxunpool_scatter_nhwc(int B, int C, int H, int W, int kh, int kw, const float *dpooled, int64_t ldp, const int32_t *index,
                               float *dx, skd_stream_t stream) {
  if (B <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || !dpooled || !index || !dx) return 0;
  if (reinterpret_cast<uintptr_t>(dx) & 15) return 0;
  const int OH = (int)cdiv(H, kh), OW = (int)cdiv(W, kw), C4 = C / 4;
  const int64_t total = (int64_t)B * H * W * C4;
  if (cdiv(total, kThreads) > 2147483647) return 0;
  maxunpool_nhwc_kernel<<<dim3((unsigned)cdiv(total, kThreads)), dim3(kThreads), 0, as_stream(stream)>>>(dpooled, index, dx, total, C4,
                                                                                                        H, W, kh, kw, OH, OW, ldp);
  return ok();
}

int skd_pairwise_ldm(int M) { return M <= 0 ? 0 : (int)(cdiv(M, kTile) * kTile); }

int skd_channel_l2_normalise(int B, int C, int M, const float *pooled, float *fhat, int ldm,
                             float *fhat_t, int ldc, float *norm, skd_stream_t stream) {
  if (B <= 0 || C <= 0 || M <= 0 || !pooled || !fhat || ldm < M) return 0;
  if (fhat_t != nullptr && ldc < C) return 0;
  if (B > 65535) return 0;
  l2_normalise_kernel<<<dim3((unsigned)cdiv(ldm, kWave), B), dim3(kThreads), 0, as_stream(stream)>>>(
      pooled, fhat, fhat_t, norm, C, M, ldm, ldc);
  return ok();
}

int64_t skd_pairwise_workspace_floats(int B, int M) {
  if (B <= 0 || M <= 0) return 1;
  const int64_t nt = cdiv(M, kTile);
  return nt * (nt + 1) / 2 * B * 2;          // (x 2: the half-height tiles of small graphs write one partial each)
}

static bool gemm_lds_ready() {
  static PerDeviceFlag flag;
  bool *donep = flag.get();
  if (donep == nullptr) return false;
  bool &done = *donep;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(gram_loss_kernel<kTile>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)Panels<kTile>::lds_bytes) != hipSuccess) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(gram_loss_kernel<64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)Panels<64>::lds_bytes) != hipSuccess) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(pairwise_bwd_kernel<kTile>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)Panels<kTile>::lds_bytes) != hipSuccess) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(pairwise_bwd_kernel<64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)Panels<64>::lds_bytes) != hipSuccess) return false;
    done = true;
  }
  return true;
}

int skd_pairwise_gram_loss(int B, int Cs, int Ct, int M, int ldm, const float *fhat_s,
                           const float *fhat_t, float *G, float *loss, float *workspace,
                           skd_stream_t stream) {
  if (B <= 0 || Cs <= 0 || Ct <= 0 || M <= 0 || !fhat_s || !fhat_t || !loss || !workspace) return 0;
  if (ldm != skd_pairwise_ldm(M) || B > 65535) return 0;
  if ((reinterpret_cast<uintptr_t>(fhat_s) | reinterpret_cast<uintptr_t>(fhat_t)) & 15) return 0;
  hipStream_t st = as_stream(stream);
  const int nt = ldm / kTile;
  const int ntri = nt * (nt + 1) / 2;
  if (!gemm_lds_ready()) return 0;
  // fewer 128 x 128 tiles than two rounds of the chip's 512 slots: half-height tiles, twice the workgroups (M = 1089: 720 instead of 360)
  const int halves = (int64_t)ntri * B < 1024 ? 2 : 1;
  if (halves == 2)
    gram_loss_kernel<64><<<dim3((unsigned)ntri * 2, B), dim3(kThreads), Panels<64>::lds_bytes, st>>>(fhat_s, fhat_t, G, workspace, Cs,
                                                                                                     Ct, ldm, nt);
  else
    gram_loss_kernel<kTile><<<dim3((unsigned)ntri, B), dim3(kThreads), Panels<kTile>::lds_bytes, st>>>(fhat_s, fhat_t, G, workspace,
                                                                                                         Cs, Ct, ldm, nt);
  if (!ok()) return 0;
  // utils.py:181: / (M*M) / B
  return launch_final_sum(workspace, (int64_t)ntri * halves * B, loss, 1.0 / ((double)M * (double)M) / (double)B, st);
}

// Stream-K geometry of the backward contraction (see pairwise_bwd_kernel): one round of the chip -- 2 workgroups (64 KB of LDS
// each) per compute unit, 512 on a whole MI355X; sized WITHOUT asking the device (the workspace query must work on a host without
// one) -- but never fewer than 8 K-tiles per workgroup (pipeline fill / drain) and never more workgroups than units.
constexpr int kStreamWG = 512;
// rows of an output tile: 128, or 64 when 128-row tiles would give fewer than 512 workgroups of >= 8 K-tiles (M = 1089: 324)
static int bwd_tile_rows(int B, int Cs, int ldm) {
  const int64_t units = (int64_t)(ldm / kTile) * cdiv(Cs, kTile) * B * (ldm / kBK);
  return units / 8 < kStreamWG ? 64 : kTile;
}
static StreamK bwd_stream_k(int B, int Cs, int ldm) {
  StreamK k;
  k.nkt = ldm / kBK;
  k.total = (int64_t)(ldm / kTile) * cdiv(Cs, bwd_tile_rows(B, Cs, ldm)) * B * k.nkt;
  int64_t n = k.total / 8;
  if (n > kStreamWG) n = kStreamWG;
  if (n < 1) n = 1;
  k.nwg = (int)n;
  return k;
}

// [node-major copy of the student panel: B * ldm * ldc][2 slots of 128 x 128 floats per workgroup]
int64_t skd_pairwise_backward_workspace_floats(int B, int Cs, int M) {
  if (B <= 0 || Cs <= 0 || M <= 0) return 1;
  const int ldm = skd_pairwise_ldm(M);
  const StreamK k = bwd_stream_k(B, Cs, ldm);
  const int64_t ldc = cdiv(Cs, kTile) * kTile;
  return (int64_t)B * ldm * ldc + (int64_t)k.nwg * 2 * kTile * kTile;
}

int skd_pairwise_backward(int B, int Cs, int M, int ldm, const float *fhat_s, const float *G,
                          const float *norm_s, const float *grad_loss, float *dpooled, float *workspace,
                          skd_stream_t stream) {
  if (B <= 0 || Cs <= 0 || M <= 0 || !fhat_s || !G || !norm_s || !grad_loss || !dpooled) return 0;
  if (ldm != skd_pairwise_ldm(M) || B > 65535) return 0;
  if ((reinterpret_cast<uintptr_t>(fhat_s) | reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(dpooled)) & 15) return 0;
  const int TI = bwd_tile_rows(B, Cs, ldm);
  const int ntm = ldm / kTile, ntc = (int)cdiv(Cs, TI);
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15)) return 0;
  const int ldc = (int)cdiv(Cs, kTile) * kTile;          // node-major copy: channel rows padded to 128 (covers 64-row tiles too)
  const StreamK sk = bwd_stream_k(B, Cs, ldm);
  const int64_t tiles = (int64_t)ntm * ntc * B;
  if (tiles > 2147483647) return 0;
  float *ft = workspace, *part = workspace + (int64_t)B * ldm * ldc;
  // dL/dA_S = -2 G /(M^2 B); dFhat = Fhat (dA + dA^T) = -4/(M^2 B) Fhat G   (G symmetric)
  const float coef = (float)(-4.0 / ((double)M * (double)M * (double)B));
  if (!gemm_lds_ready()) return 0;
  hipStream_t st = as_stream(stream);
  transpose_pad_kernel<<<dim3((unsigned)(ldm / 32), (unsigned)(ldc / 32), B), dim3(kThreads), 0, st>>>(fhat_s, ft, Cs, ldm, ldc);
  if (TI == 64)
    pairwise_bwd_kernel<64><<<dim3((unsigned)sk.nwg), dim3(kThreads), Panels<64>::lds_bytes, st>>>(
        ft, G, norm_s, grad_loss, dpooled, part, Cs, M, ldm, ldc, ntm, ntc, sk, coef);
  else
    pairwise_bwd_kernel<kTile><<<dim3((unsigned)sk.nwg), dim3(kThreads), Panels<kTile>::lds_bytes, st>>>(
        ft, G, norm_s, grad_loss, dpooled, part, Cs, M, ldm, ldc, ntm, ntc, sk, coef);
  if (!ok()) return 0;
  pairwise_bwd_fixup_kernel<<<dim3((unsigned)tiles, 4), dim3(kThreads), 0, st>>>(part, norm_s, grad_loss, dpooled, Cs, M, ldm, ntm, ntc,
                                                                            sk, coef, TI);
  return ok();
}


// Whole similarity loss for small graphs (M <= 64) in one launch per call (+ the deterministic final sum):
// pooled_s (B, Cs, M), pooled_t (B, Ct, M) -> loss[0] = sum (A_T - A_S)^2 / M^2 / B and, when dpooled != NULL,
// dpooled (B, Cs, M) = dloss / dpooled_s (for an upstream gradient of 1; utils.py:170-183 with the norm detached).
// workspace: B floats.
int skd_pairwise_small(int B, int Cs, int Ct, int M, const float *pooled_s, const float *pooled_t, float *loss,
                       float *dpooled, float *workspace, skd_stream_t stream) {
  if (B <= 0 || Cs <= 0 || Ct <= 0 || M <= 0 || M > kSmallM || !pooled_s || !pooled_t || !loss || !workspace) return 0;
  hipStream_t st = as_stream(stream);
  const float coef = (float)(-4.0 / ((double)M * (double)M * (double)B));
  pairwise_small_kernel<<<dim3((unsigned)B), dim3(kThreads), 0, st>>>(pooled_s, pooled_t, workspace, dpooled, Cs, Ct, M, coef);
  if (!ok()) return 0;
  return launch_final_sum(workspace, B, loss, 1.0 / ((double)M * (double)M) / (double)B, st);
}

}  // extern "C"
