// sync.hip -- one-hop exchange of the cross-replica InPlace-ABN statistics for gfx950 over IPC-mapped device memory.
//
// Replaces the reference's master / worker queues + comm.gather + broadcast_coalesced (libs/functions.py:185-205, 263-280,
// libs/bn.py) and this package's first implementation of them (torch.distributed all_gather / all_reduce, 29 + 29 blocking
// collectives per step on the compute stream, each a library launch with tens of microseconds of latency for a 1-4 KB
// message).  Per-channel statistics are tiny; what matters is latency.  On an 8 x MI355X node every GPU reaches every other
// GPU in ONE xGMI hop and can store into its memory directly, so the exchange is a single small kernel:
//
// (Device side: sync_dev.hpp, shared with the fused forms in abn.hip.)
//   mailbox   every rank owns one device buffer (uncached / fine-grained: hipDeviceMallocUncached, the allocation type RCCL
//             uses for its own flags) and exports it with hipIpcGetMemHandle; every rank opens all the others'.
//             Layout: [parity 2][writer G][ 4 flag words (one per channel block) | pad to 64 B | payload kSyncMaxFloats floats ].
//   exchange  ONE 256-thread workgroup on the caller's stream: (1) stores this rank's payload into slot [parity][rank] of
//             EVERY rank's mailbox (system-scope write-through stores; own mailbox included), (2) drains them, fences at
//             system scope and publishes the launch's sequence number in each of those slots' flag words, (3) spins
//             (bounded) until all G flag words of ITS OWN mailbox carry the sequence number, (4) reads the G payloads from its
//             own (local) mailbox and finishes the job in the same launch: the combine rule of functions.py:196-197 with
//             the running-statistics update of :208-209 (bit-identical to skd_abn_combine_stats on gathered data), or the
//             rank-ordered weighted sum of [edz, eydz] of :271-272.
//   parity    consecutive exchanges alternate between two slot sets; a rank can only be one exchange ahead of a peer
//             (it needs the peer's flag of exchange s to finish s), so a slot written for exchange s + 2 has been read
//             by its owner for exchange s.  Sequence numbers are a host-side counter per context: every rank calls the
//             exchanges in the same order (they are collectives, like the calls they replace).
// A spin that times out (a peer never arrived within skd_sync_set_timeout's limit: 5 s while a context is being set up, then
// as long as torch.distributed would have waited) poisons the outputs with NaN instead of hanging the device AND raises the
// status word kStatusSyncTimeout (status.hip), which the host checks once per step and turns into an exception.
// Single-device use (two ranks sharing one GPU, tests/test_distributed_gpu.py) runs the very same code: the "remote"
// mailbox is then another process's allocation on the same device.
#include <stdlib.h>
#include <string.h>

#include "sync_dev.hpp"

namespace skd {
namespace {

constexpr double kSetupTimeoutSeconds = 5.0;         // until skd_sync_set_timeout: the self-test of a fresh context must fail fast

struct SyncCtx {
  SyncDev dev;
  float *own = nullptr;
  void *opened[kSyncMaxWorld] = {};
  unsigned seq = 0;
  int device = 0;
  uint64_t spin_ticks = (uint64_t)(kSetupTimeoutSeconds * (double)kSyncTicksPerSecond);
};

// all-gather: gathered (G, n)
__global__ __launch_bounds__(kThreads) void sync_gather_kernel(SyncArgs a, int n, const float *__restrict__ src,
                                                               float *__restrict__ gathered) {
  __shared__ unsigned ok_s;
  const bool good = sync_exchange(a, 0, 1, [&](float *dst) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) store_sys(dst + i, src[i]);
  }, &ok_s);
  for (int r = 0; r < a.d.world; ++r)
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      gathered[(int64_t)r * n + i] = good ? sync_payload(a.d, a.seq, r, i) : __builtin_nanf("");
}

// forward statistics: stat = [mean (C), var (C)] of this replica -> the combined mean / var (+ running update), the arithmetic
// of abn_combine_stats_kernel (abn.hip) term for term
__global__ __launch_bounds__(kThreads) void sync_stats_kernel(SyncArgs a, int C, const float *__restrict__ stat,
                                                              const float *__restrict__ weights, float *__restrict__ mean,
                                                              float *__restrict__ var, float *running_mean, float *running_var,
                                                              float momentum, float nf) {
  __shared__ unsigned ok_s;
  const bool good = sync_exchange(a, 0, sync_channel_blocks(C), [&](float *dst) {
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) store_sys(dst + i, stat[i]);
  }, &ok_s);
  const int G = a.d.world;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float m, v;
    combine_channel(G, C, c, [&](int g, int j) { return sync_payload(a.d, a.seq, g, j); }, weights, a.d.rank, nf, momentum, m, v,
                    good ? running_mean : nullptr, good ? running_var : nullptr);
    mean[c] = good ? m : __builtin_nanf("");
    var[c] = good ? v : __builtin_nanf("");
  }
}

// backward statistics: stat = [edz (C), eydz (C)] -> sum_g w_g stat_g (w_g = 1 / G without weights), in rank order, in place
__global__ __launch_bounds__(kThreads) void sync_grad_stats_kernel(SyncArgs a, int C, float *__restrict__ stat,
                                                                   const float *__restrict__ weights) {
  __shared__ unsigned ok_s;
  const bool good = sync_exchange(a, 0, sync_channel_blocks(C), [&](float *dst) {
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) store_sys(dst + i, stat[i]);
  }, &ok_s);
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    const float s = sync_weighted_sum(a.d, a.seq, i, weights);
    stat[i] = good ? s : __builtin_nanf("");
  }
}

static bool connected(const SyncCtx *ctx) {
  if (!ctx) return false;
  for (int r = 0; r < ctx->dev.world; ++r)
    if (ctx->dev.mail[r] == nullptr) return false;
  return true;
}

}  // namespace

bool sync_next(void *ctx_, SyncArgs &out) {
  SyncCtx *ctx = static_cast<SyncCtx *>(ctx_);
  if (!connected(ctx)) return false;
  out.d = ctx->dev;
  out.seq = (++ctx->seq) & 0x7fffffffu;
  if (out.seq == 0u) out.seq = ctx->seq = 2u;          // the mailbox starts zero-filled: 0 never names an exchange (keeps parity)
  out.spin_ticks = ctx->spin_ticks;
  out.status = status_words();
  return true;
}

}  // namespace skd

using namespace skd;

extern "C" {

int skd_sync_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }
int skd_sync_max_floats(void) { return kSyncMaxFloats; }

// Allocates this rank's mailbox on the current device and writes its IPC handle (skd_sync_handle_bytes() bytes) to handle_out.
void *skd_sync_create(int world, int rank, void *handle_out) {
  if (world < 1 || world > kSyncMaxWorld || rank < 0 || rank >= world || !handle_out) return nullptr;
  SyncCtx *ctx = new SyncCtx();
  ctx->dev.world = world;
  ctx->dev.rank = rank;
  for (int r = 0; r < kSyncMaxWorld; ++r) ctx->dev.mail[r] = nullptr;
  const size_t bytes = sizeof(float) * 2 * (size_t)world * kSlotFloats;
  void *p = nullptr;
  if (hipGetDevice(&ctx->device) != hipSuccess ||
      hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    if (p) (void)hipFree(p);
    delete ctx;
    (void)hipGetLastError();
    return nullptr;
  }
  ctx->own = static_cast<float *>(p);
  ctx->dev.mail[rank] = ctx->own;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipFree(p);
    delete ctx;
    (void)hipGetLastError();
    return nullptr;
  }
  memcpy(handle_out, &h, sizeof h);
  return ctx;
}

// all_handles: world x skd_sync_handle_bytes() bytes in rank order (this rank's own entry is ignored).  Every rank must have
// created its mailbox (and zero-filled it) before any rank's first exchange: call after a barrier / the handle all-gather.
int skd_sync_connect(void *ctx_, const void *all_handles) {
  SyncCtx *ctx = static_cast<SyncCtx *>(ctx_);
  if (!ctx || !all_handles) return 0;
  for (int r = 0; r < ctx->dev.world; ++r) {
    if (r == ctx->dev.rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const char *>(all_handles) + (size_t)r * sizeof h, sizeof h);
    void *p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    ctx->opened[r] = p;
    ctx->dev.mail[r] = static_cast<float *>(p);
  }
  return 1;
}

int skd_sync_destroy(void *ctx_) {
  SyncCtx *ctx = static_cast<SyncCtx *>(ctx_);
  if (!ctx) return 0;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < kSyncMaxWorld; ++r)
    if (ctx->opened[r]) (void)hipIpcCloseMemHandle(ctx->opened[r]);
  if (ctx->own) (void)hipFree(ctx->own);
  delete ctx;
  return 1;
}

// How long an exchange waits for a peer before it gives up (NaN outputs + status word).  A fresh context waits 5 s (its
// self-test must fail fast on a broken setup); the caller raises it once the group is known to work.
int skd_sync_set_timeout(void *ctx_, double seconds) {
  SyncCtx *ctx = static_cast<SyncCtx *>(ctx_);
  if (!ctx || !(seconds > 0.0) || seconds > 1e6) return 0;
  ctx->spin_ticks = (uint64_t)(seconds * (double)kSyncTicksPerSecond);
  return 1;
}

// all-gather of n floats per rank: gathered (world, n).  Collective: every rank, same order.
int skd_sync_all_gather(void *ctx_, int n, const float *src, float *gathered, skd_stream_t stream) {
  SyncArgs a;
  if (n <= 0 || n > kSyncMaxFloats || !src || !gathered || !sync_next(ctx_, a)) return 0;
  sync_gather_kernel<<<dim3(1), dim3(kThreads), 0, as_stream(stream)>>>(a, n, src, gathered);
  return ok();
}

// stat (2, C) = this replica's [mean, var]  ->  combined mean / var (C each) + running update: exchange + the rule of
// skd_abn_combine_stats (same arguments: weights NULL or world floats, n = pooled count without weights, this rank's with) in ONE launch.
int skd_abn_sync_stats(void *ctx_, int C, const float *stat, const float *weights, float *mean, float *var, float *running_mean,
                       float *running_var, float momentum, double n, skd_stream_t stream) {
  SyncArgs a;
  if (C <= 0 || 2 * C > kSyncMaxFloats || !stat || !mean || !var || !sync_next(ctx_, a)) return 0;
  sync_stats_kernel<<<dim3(1), dim3(kThreads), 0, as_stream(stream)>>>(a, C, stat, weights, mean, var, running_mean, running_var,
                                                                        momentum, (float)n);
  return ok();
}

// stat (2, C) = this replica's [edz, eydz]  ->  in place: sum_g w_g stat_g (plain mean without weights), rank order
int skd_abn_sync_grad_stats(void *ctx_, int C, float *stat, const float *weights, skd_stream_t stream) {
  SyncArgs a;
  if (C <= 0 || 2 * C > kSyncMaxFloats || !stat || !sync_next(ctx_, a)) return 0;
  sync_grad_stats_kernel<<<dim3(1), dim3(kThreads), 0, as_stream(stream)>>>(a, C, stat, weights);
  return ok();
}

}  // extern "C"
