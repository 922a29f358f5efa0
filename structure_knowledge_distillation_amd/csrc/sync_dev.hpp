// sync_dev.hpp -- device side of the one-hop mailbox exchange (sync.hip), shared with the one-launch InPlace-ABN passes of
// abn.hip that perform the exchange INSIDE the statistics kernel (round 4).
//
//   mailbox   [parity 2][writer G][ 16-word header | payload kSyncMaxFloats floats ]; header word k (k < 4) is the flag
//             of CHANNEL BLOCK k of that writer's payload (a plain all-gather uses word 0 only).
//   exchange  one workgroup: payload stores into slot [parity][rank] of EVERY rank's mailbox (system scope, write-through),
//             drain, system-scope release, the launch's sequence number into the flag words it owns, bounded wait for the
//             same words of all G writers in its OWN mailbox, system-scope acquire.
// Flags are per channel block so that the two forms of an exchange interoperate: the stand-alone kernel (one workgroup,
// all channel blocks: sync.hip) and the fused form in which the last-arriving workgroup of EACH channel block of a
// register-resident ABN launch exchanges its own block (abn.hip).  Which form a rank takes depends on whether ITS tensor
// fits the register file -- with ragged shards that differs between ranks -- and must not matter to the protocol: one
// exchange = one sequence number on every rank, one flag word per (writer, channel block), the same payload layout.
#pragma once
#include "skd_common.hpp"

namespace skd {

constexpr int kSyncMaxWorld = 16;
constexpr int kSyncMaxFloats = 4096;                 // payload floats per slot (2 * C for C <= 2048)
constexpr int kSlotHeaderFloats = 16;                // flag words + padding: the payload starts 64 bytes into the slot
constexpr int kSyncFlagWords = 4;                    // one per channel block (kRedMaxCB of abn.hip)
constexpr int kSlotFloats = kSlotHeaderFloats + kSyncMaxFloats;
constexpr uint64_t kSyncTicksPerSecond = 100000000ull;   // wall_clock64(): 100 MHz

struct SyncDev {           // passed to the kernels by value
  float *mail[kSyncMaxWorld];   // mail[r] = rank r's mailbox as mapped into THIS process (mail[rank] = the local one)
  int world, rank;
};

// what a launch needs to take part in exchange number `seq`
struct SyncArgs {
  SyncDev d;
  unsigned seq;
  uint64_t spin_ticks;     // how long to wait for a peer (skd_sync_set_timeout)
  unsigned *status;        // device-raised error words (status.hip) or nullptr
};

// channel blocks of a C-channel statistics vector: the split of abn.hip's channels-last reductions (make_red_geom)
__host__ __device__ __forceinline__ int sync_channel_blocks(int C) {
  if (C < 4 || (C & (C - 1)) || C > 1024) return 1;       // shapes the channels-last kernels do not take: one block
  return C >= 256 ? 4 : (C >= 128 ? 2 : 1);
}

__device__ __forceinline__ float *slot_of(float *mailbox, int world, int parity, int writer) {
  return mailbox + ((int64_t)parity * world + writer) * kSlotFloats;
}
__device__ __forceinline__ void store_sys(float *p, float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float load_sys(const float *p) {
  float v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// One workgroup's part of exchange a.seq: `store(payload_base_of_the_slot)` is called once per rank r by ALL threads and
// stores this workgroup's share of the payload into rank r's mailbox (store_sys); the workgroup owns the flag words
// [word_lo, word_lo + nwords).  Returns, in every thread, false when a peer did not arrive within a.spin_ticks (the
// status word kStatusSyncTimeout is raised with the sequence number).  ok_s: one word of LDS.
template <class Store>
__device__ __forceinline__ bool sync_exchange(const SyncArgs &a, int word_lo, int nwords, Store store, unsigned *ok_s) {
  const SyncDev &d = a.d;
  const int t = threadIdx.x, parity = (int)(a.seq & 1u);
  for (int r = 0; r < d.world; ++r) store(slot_of(d.mail[r], d.world, parity, d.rank) + kSlotHeaderFloats);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope
  __syncthreads();
  const int nflags = d.world * nwords;
  if (t < nflags) {
    const int peer = t / nwords, w = word_lo + t % nwords;
    unsigned *flag = reinterpret_cast<unsigned *>(slot_of(d.mail[peer], d.world, parity, d.rank)) + w;
    __hip_atomic_store(flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (t == 0) *ok_s = 1u;
  __syncthreads();
  if (t < nflags) {
    const int writer = t / nwords, w = word_lo + t % nwords;
    const unsigned *flag = reinterpret_cast<const unsigned *>(slot_of(d.mail[d.rank], d.world, parity, writer)) + w;
    // A wait gives up after a.spin_ticks -- or, once it has lasted 0.1 s, as soon as an EARLIER exchange of this process has
    // already timed out (its status word is still raised: the host has not consumed it yet).  The step is poisoned by then
    // anyway; without this every one of the 58 exchanges of a step would sit out the full limit for a peer that is gone, and
    // a benchmark / trainer that wants to fall back to another transport (bench.py --gpus N) would wait an hour to learn it.
    // The word lives in host memory: it is only looked at from waits that are already long (never on the fast path).
    const uint64_t t0 = wall_clock64();
    uint64_t look = kSyncTicksPerSecond / 10;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
      __builtin_amdgcn_s_sleep(8);
      const uint64_t dt = wall_clock64() - t0;
      bool give_up = dt > a.spin_ticks;
      if (!give_up && dt > look && a.status != nullptr) {
        give_up = __hip_atomic_load(a.status + kStatusSyncTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
        look += kSyncTicksPerSecond / 100;
      }
      if (give_up) {
        *ok_s = 0u;
        raise_status_first(a.status, kStatusSyncTimeout, a.seq | 0x80000000u);      // the word names the exchange that timed out FIRST
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return *ok_s != 0u;
}

// element j of writer g's payload in THIS rank's mailbox
__device__ __forceinline__ float sync_payload(const SyncDev &d, unsigned seq, int g, int j) {
  return load_sys(slot_of(d.mail[d.rank], d.world, (int)(seq & 1u), g) + kSlotHeaderFloats + j);
}

// rank-ordered weighted sum of element j over the writers: the bits of mul_(w[rank]) + all_reduce(SUM) / of mean over ranks
__device__ __forceinline__ float sync_weighted_sum(const SyncDev &d, unsigned seq, int j, const float *__restrict__ weights) {
  float s = 0.f;
  for (int g = 0; g < d.world; ++g) {
    const float v = sync_payload(d, seq, g, j);
    s = __fadd_rn(s, weights ? __fmul_rn(weights[g], v) : v);     // product rounded, then added
  }
  if (!weights) s /= (float)d.world;
  return s;
}

}  // namespace skd

// sync.hip: the launch arguments of the NEXT exchange of a context (bumps its sequence number); false when ctx is not a
// connected context.  Host side of the fused forms in abn.hip.
namespace skd {
bool sync_next(void *ctx, SyncArgs &out);
}
