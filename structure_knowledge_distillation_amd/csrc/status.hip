// status.hip -- device-raised error words the host reads WITHOUT synchronising the device (include/skd.h section 13).
//
// Two kernels of this library wait inside a launch: the cross-replica mailbox exchange (sync.hip: for the peers' flags)
// and the register-resident one-launch InPlace-ABN passes (abn.hip: a grid barrier per channel block).  Both waits are
// bounded, and a wait that runs out used to poison its outputs with NaN and nothing else: the fast rank got
// NaN statistics, the late rank finished normally, and the gradient all-reduce then spread the NaN to every replica with
// no host-visible cause.  Now the kernel that gives up ALSO stores a code into a small host-mapped, system-coherent buffer
// (hipHostMalloc: the device writes straight into host memory; one allocation per process, portable across its devices)
// and the host checks it with a plain load -- once per step (NetModel.optimize_parameters) and whenever it reads a
// logged scalar (after the .item() that synchronises anyway) -- and raises.  The reference's NCCL path would have waited:
// the exchange's own time limit is therefore as long as torch.distributed's default (skd_sync_set_timeout, 600 s).
#include <mutex>

#include "skd_common.hpp"

namespace skd {
namespace {
std::once_flag g_status_once;
unsigned *g_status_host = nullptr;
unsigned *g_status_dev = nullptr;
}  // namespace

unsigned *status_words() {
  std::call_once(g_status_once, [] {
    void *p = nullptr;
    if (hipHostMalloc(&p, sizeof(unsigned) * kStatusWords, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) !=
        hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    for (int i = 0; i < kStatusWords; ++i) static_cast<volatile unsigned *>(p)[i] = 0u;
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipHostFree(p);
      return;
    }
    g_status_host = static_cast<unsigned *>(p);
    g_status_dev = static_cast<unsigned *>(d);
  });
  return g_status_dev;
}

}  // namespace skd

using namespace skd;

extern "C" {

int skd_status_words(void) { return kStatusWords; }

// Copies the status words (skd_status_words() of them) to `out` (HOST memory).  No device synchronisation: a word is
// nonzero once the kernel that raised it has stored it.  1 = ok, 0 = the buffer could not be allocated.
int skd_status_read(unsigned *out) {
  if (!out) return 0;
  (void)status_words();
  if (!g_status_host) return 0;
  for (int i = 0; i < kStatusWords; ++i) out[i] = __atomic_load_n(g_status_host + i, __ATOMIC_ACQUIRE);
  return 1;
}

int skd_status_clear(void) {
  (void)status_words();
  if (!g_status_host) return 0;
  for (int i = 0; i < kStatusWords; ++i) __atomic_store_n(g_status_host + i, 0u, __ATOMIC_RELEASE);
  return 1;
}

}  // extern "C"
