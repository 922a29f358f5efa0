// head.hip -- the 19-class 1x1 classifier heads of the PSPNets (networks/pspnet_combine.py:138-154: `head` and the last layer of
// `dsn`, Conv2d(mid, num_classes, 1, bias=True)) for channels-last feature maps, forward and backward, for gfx950.
//
// As convolutions these are (B*H*W, K) x (K, 19) GEMMs with 0.16 GFLOP each -- nothing for a matrix pipe, everything for HBM: the
// forward has to read the 17 MB feature map once and write 2.6 MB of logits, the backward to read it again with the logit gradient
// and write the 17 MB feature gradient.  The convolution library spends 41 us on the forward and 123 us on the backward of ONE student
// head (profiles/r06w_conv_shapes.md: 0.03 / 0.017 of the fp32 MFMA peak, i.e. ~0.5 TB/s) -- 0.33 ms per step for the two student
// heads, 0.07 ms for the teacher's -- and hands the logits over channels-last, so that the criteria (which want the reference's NCHW
// layout) pay a layout copy on top, and the feature gradient NCHW, so that the InPlace-ABN backward pays another.  Here:
//   forward   one workgroup per 64 rows: the rows' K-chunk (128 floats each) goes to LDS with coalesced 16-byte loads; wave w owns the
//             classes w, w + 4, ... -- its weight rows are wave-uniform and come through scalar loads --, lane r row r; logits are
//             written straight in NCHW (64 consecutive positions per class: 256-byte stores).
//   backward  (K = 128) half a wave per row, lane q = channel quad q: g[m][0..18] from an LDS tile (broadcast), the weight quad of
//             every class in registers: dx[m][4q..] = sum_c g[m][c] W[c][4q..] (one coalesced 512-byte store per row) and, from the
//             same loads, dW[c][4q..] += g[m][c] x[m][4q..], db[c] += g[m][c]; per-workgroup partials in a fixed order, a small second
//             kernel adds them in workgroup order: no atomics, bit-reproducible.
// Algorithmic bytes: forward 4 * M * (K + C), backward 4 * M * (2 K + C).  Bound: HBM.
#include "skd_common.hpp"

namespace skd {
namespace {

constexpr int kHeadRows = 64;            // rows per workgroup tile
constexpr int kHeadKC = 128;             // K chunk
constexpr int kHeadLd = kHeadKC + 4;     // LDS row stride of the x tile (conflict-free ds_read_b128 at a 132-float stride)
constexpr int kHeadMaxC = 20;            // classes: 4 waves x 5
constexpr int kHeadCPW = kHeadMaxC / 4;  // classes per wave

// out[b][c][p] = bias[c] + sum_k x[(b * HW + p)][k] * w[c][k]
// The weight rows a wave needs are the same for all its lanes: read through UNIFORM pointers (scalar loads, scalar cache) -- the first
// version kept the weight in LDS and spent five of six LDS reads on those broadcasts (84 us at K = 512, LDS-bandwidth-bound).
__global__ __launch_bounds__(kThreads) void head_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ bias, float *__restrict__ out, int64_t M,
                                                           int HW, int K, int C) {
  __shared__ __attribute__((aligned(16))) float xl[kHeadRows * kHeadLd];     // [64][132]
  const int t = threadIdx.x, r = t & (kHeadRows - 1);
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);                     // wave index, provably uniform
  const int64_t m0 = (int64_t)blockIdx.x * kHeadRows;
  float acc[kHeadCPW];
#pragma unroll
  for (int j = 0; j < kHeadCPW; ++j) acc[j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += kHeadKC) {
    if (k0) __syncthreads();                         // the previous chunk's readers
    for (int i = t; i < kHeadRows * (kHeadKC / 4); i += kThreads) {
      const int row = i >> 5, q = i & 31;
      const int64_t m = m0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) v = *reinterpret_cast<const float4 *>(x + m * K + k0 + q * 4);
      *reinterpret_cast<float4 *>(xl + row * kHeadLd + q * 4) = v;
    }
    __syncthreads();
    const float *xr = xl + r * kHeadLd;
#pragma unroll
    for (int j = 0; j < kHeadCPW; ++j) {
      const int c = wv + 4 * j;
      if (c < C) {                                   // (uniform)
        const float *wr = w + (int64_t)c * K + k0;   // uniform pointer
        float a = acc[j];
#pragma unroll 8
        for (int q = 0; q < kHeadKC / 4; ++q) {
          const float4 xv = *reinterpret_cast<const float4 *>(xr + q * 4);
          a = __builtin_fmaf(xv.x, wr[q * 4 + 0], a);
          a = __builtin_fmaf(xv.y, wr[q * 4 + 1], a);
          a = __builtin_fmaf(xv.z, wr[q * 4 + 2], a);
          a = __builtin_fmaf(xv.w, wr[q * 4 + 3], a);
        }
        acc[j] = a;
      }
    }
  }
  const int64_t m = m0 + r;
  if (m < M) {
    const int64_t b = m / HW, p = m - b * HW;
#pragma unroll
    for (int j = 0; j < kHeadCPW; ++j) {
      const int c = wv + 4 * j;
      if (c < C) out[(b * C + c) * HW + p] = acc[j] + (bias != nullptr ? bias[c] : 0.f);
    }
  }
}

// K = 128.  gx[m][4q..4q+3] = sum_c g[b][c][p] w[c][4q..]; per-workgroup partials of dW (C x 128) and db (C) into part[wg][C][132]
// (column 128 of a row = the bias partial).
template <int CMAX>
__global__ __launch_bounds__(kThreads, 2) void head_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ g, float *__restrict__ gx,
                                                              float *__restrict__ part, int64_t M, int HW, int C, int blocks) {
  __shared__ __attribute__((aligned(16))) float gl[kHeadRows][CMAX];        // the block's logit gradients, row-major
  __shared__ __attribute__((aligned(16))) float redbuf[8 * 5 * kHeadKC];     // cross-half-wave reduction, five classes per round (20 KB)
  const int t = threadIdx.x, q = t & 31, hw8 = t >> 5;                       // lane's channel quad, half-wave index 0..7
  float4 wq[CMAX], aw[CMAX];
  float ab[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    wq[c] = c < C ? *reinterpret_cast<const float4 *>(w + c * kHeadKC + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    aw[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[c] = 0.f;
  }
  for (int blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
    const int64_t m0 = (int64_t)blk * kHeadRows;
    __syncthreads();
    for (int i = t; i < kHeadRows * CMAX; i += kThreads) {                   // class-major reads: 64 consecutive positions per class
      const int c = i >> 6, row = i & 63;
      const int64_t m = m0 + row;
      float v = 0.f;
      if (c < C && m < M) {
        const int64_t b = m / HW, p = m - b * HW;
        v = g[(b * C + c) * HW + p];
      }
      gl[row][c] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < kHeadRows / 8; ++rr) {
      const int row = hw8 + 8 * rr;
      const int64_t m = m0 + row;
      if (m >= M) continue;                                                   // (uniform per half-wave)
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x != nullptr) xv = *reinterpret_cast<const float4 *>(x + m * kHeadKC + q * 4);
      float gv[CMAX];
#pragma unroll
      for (int c4 = 0; c4 < CMAX / 4; ++c4) {
        const float4 v = *reinterpret_cast<const float4 *>(&gl[row][c4 * 4]);   // broadcast within the half-wave
        gv[c4 * 4 + 0] = v.x; gv[c4 * 4 + 1] = v.y; gv[c4 * 4 + 2] = v.z; gv[c4 * 4 + 3] = v.w;
      }
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        d.x = __builtin_fmaf(gv[c], wq[c].x, d.x);
        d.y = __builtin_fmaf(gv[c], wq[c].y, d.y);
        d.z = __builtin_fmaf(gv[c], wq[c].z, d.z);
        d.w = __builtin_fmaf(gv[c], wq[c].w, d.w);
        aw[c].x = __builtin_fmaf(gv[c], xv.x, aw[c].x);
        aw[c].y = __builtin_fmaf(gv[c], xv.y, aw[c].y);
        aw[c].z = __builtin_fmaf(gv[c], xv.z, aw[c].z);
        aw[c].w = __builtin_fmaf(gv[c], xv.w, aw[c].w);
        ab[c] += gv[c];
      }
      if (gx != nullptr) *reinterpret_cast<float4 *>(gx + m * kHeadKC + q * 4) = d;
    }
  }
  // workgroup partial: the eight half-waves' sums in half-wave order, five classes at a time through LDS (8 x 5 x 128 floats)
  float *mine = part + (size_t)blockIdx.x * CMAX * kHeadLd;
  float *red = &redbuf[0];
#pragma unroll
  for (int c0 = 0; c0 < CMAX; c0 += 5) {
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) *reinterpret_cast<float4 *>(red + ((hw8 * 5 + cc) * kHeadKC) + q * 4) = aw[c0 + cc];
    __syncthreads();
    for (int i = t; i < 5 * kHeadKC; i += kThreads) {
      const int cc = i >> 7, k = i & (kHeadKC - 1);
      float sum = 0.f;
#pragma unroll
      for (int h = 0; h < 8; ++h) sum += red[(h * 5 + cc) * kHeadKC + k];
      mine[(c0 + cc) * kHeadLd + k] = sum;
    }
  }
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c) red[hw8 * CMAX + c] = ab[c];
  }
  __syncthreads();
  if (t < CMAX) {
    float sum = 0.f;
    for (int h = 0; h < 8; ++h) sum += red[h * CMAX + t];
    mine[t * kHeadLd + kHeadKC] = sum;
  }
}

// gw[c][k] = sum over the workgroups' partials in a FIXED order: eight segments of the workgroup range per column (one thread each,
// four interleaved chains), then the segments in order; gb[c] likewise.  grid (C), block 1024 = 128 columns x 8 segments; the bias
// column (128) is summed by the first eight threads afterwards.
__global__ __launch_bounds__(1024) void head_bwd_finish_kernel(const float *__restrict__ part, float *__restrict__ gw,
                                                              float *__restrict__ gb, int C, int CMAX, int nwg) {
  __shared__ float seg[8][kHeadKC + 1];
  const int c = blockIdx.x, k = threadIdx.x & (kHeadKC - 1), j = threadIdx.x >> 7;
  const size_t stride = (size_t)CMAX * kHeadLd;
  const int per = (nwg + 7) / 8, w0 = j * per, w1 = (w0 + per < nwg ? w0 + per : nwg);
  auto column = [&](int col) {
    const float *p = part + (size_t)c * kHeadLd + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int wgi = w0;
    for (; wgi + 4 <= w1; wgi += 4) {
      s0 += p[(size_t)wgi * stride];
      s1 += p[(size_t)(wgi + 1) * stride];
      s2 += p[(size_t)(wgi + 2) * stride];
      s3 += p[(size_t)(wgi + 3) * stride];
    }
    for (; wgi < w1; ++wgi) s0 += p[(size_t)wgi * stride];
    return (s0 + s1) + (s2 + s3);
  };
  seg[j][k] = column(k);
  if (k == 0) seg[j][kHeadKC] = column(kHeadKC);
  __syncthreads();
  if (threadIdx.x <= kHeadKC) {
    const int col = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) s += seg[jj][col];
    if (col < kHeadKC) {
      if (gw != nullptr) gw[c * kHeadKC + col] = s;
    } else if (gb != nullptr) {
      gb[c] = s;
    }
  }
}

constexpr int kHeadBwdWG = 512;          // two workgroups per compute unit of a whole MI355X

}  // namespace
}  // namespace skd

using namespace skd;

extern "C" {

// 1 when the entries below take the head: forward K % 128 == 0, K <= 1024, C <= 20; backward K == 128.
int skd_head1x1_supported(int K, int C, int backward) {
  if (C <= 0 || C > kHeadMaxC || K <= 0) return 0;
  return backward ? K == kHeadKC : (K % kHeadKC == 0 && K <= 1024);
}

int skd_head1x1_forward_nhwc(int B, int HW, int K, int C, const float *x, const float *w, const float *bias, float *out,
                             skd_stream_t stream) {
  if (B <= 0 || HW <= 0 || !skd_head1x1_supported(K, C, 0) || !x || !w || !out) return 0;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) return 0;
  const int64_t M = (int64_t)B * HW;
  head_fwd_kernel<<<dim3((unsigned)cdiv(M, kHeadRows)), dim3(kThreads), 0, as_stream(stream)>>>(x, w, bias, out, M, HW, K, C);
  return ok();
}

int64_t skd_head1x1_backward_workspace_floats(int B, int HW, int K, int C) {
  if (B <= 0 || HW <= 0 || !skd_head1x1_supported(K, C, 1)) return 1;
  return (int64_t)kHeadBwdWG * kHeadMaxC * kHeadLd;
}

// gout (B, C, HW) NCHW; gx (B*HW, K) channels-last (may be NULL), gw (C, K) and gb (C) WRITTEN (may be NULL); x may be NULL when
// gw is.  workspace: skd_head1x1_backward_workspace_floats().
int skd_head1x1_backward_nhwc(int B, int HW, int K, int C, const float *x, const float *w, const float *gout, float *gx, float *gw,
                              float *gb, float *workspace, skd_stream_t stream) {
  if (B <= 0 || HW <= 0 || !skd_head1x1_supported(K, C, 1) || !w || !gout || !workspace) return 0;
  if (gw != nullptr && x == nullptr) return 0;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(gx)) & 15) return 0;
  const int64_t M = (int64_t)B * HW;
  const int blocks = (int)cdiv(M, kHeadRows);
  const int nwg = blocks < kHeadBwdWG ? blocks : kHeadBwdWG;
  hipStream_t st = as_stream(stream);
  head_bwd_kernel<kHeadMaxC><<<dim3((unsigned)nwg), dim3(kThreads), 0, st>>>(x, w, gout, gx, workspace, M, HW, C, blocks);
  if (!ok()) return 0;
  if (gw != nullptr || gb != nullptr) {
    head_bwd_finish_kernel<<<dim3((unsigned)C), dim3(1024), 0, st>>>(workspace, gw, gb, C, kHeadMaxC, nwg);
    return ok();
  }
  return 1;
}

}  // extern "C"
