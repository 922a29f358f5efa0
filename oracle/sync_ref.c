/* sync_ref.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): host restatement of csrc/sync.hip, the one-hop exchange of
 * the cross-replica InPlace-ABN statistics (libs/functions.py:185-209 forward, :263-280 backward), behind the same C ABI
 * (include/skd.h section 12).  The product maps every rank's mailbox into every process with HIP IPC and lets ONE kernel
 * store / flag / spin / combine; here the mailboxes are POSIX shared-memory segments and the same four steps run on the host
 * (C11 atomics for the flag words), so the world-2 gloo tests on a box without a GPU execute the protocol itself: slot layout,
 * parity alternation, sequence numbers, the combine rule and the rank-ordered weighted sum. */
#define _GNU_SOURCE
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

typedef void *stream_t;
#define MAX_WORLD 16
#define MAX_FLOATS 4096
#define HEADER_FLOATS 16
#define SLOT_FLOATS (HEADER_FLOATS + MAX_FLOATS)
#define HANDLE_BYTES 64

typedef struct {
  float *mail[MAX_WORLD];
  size_t bytes;
  int world, rank;
  unsigned seq;
  double timeout_s;        /* skd_sync_set_timeout: 5 s while a context is being set up */
  char name[HANDLE_BYTES];
} sync_ctx;

/* device-raised error words of csrc/status.hip (include/skd.h section 13), here plain host words */
#define STATUS_WORDS 4
static unsigned g_status[STATUS_WORDS];
int skd_status_words(void) { return STATUS_WORDS; }
int skd_status_read(unsigned *out) {
  if (!out) return 0;
  for (int i = 0; i < STATUS_WORDS; ++i) out[i] = __atomic_load_n(&g_status[i], __ATOMIC_ACQUIRE);
  return 1;
}
int skd_status_clear(void) {
  for (int i = 0; i < STATUS_WORDS; ++i) __atomic_store_n(&g_status[i], 0u, __ATOMIC_RELEASE);
  return 1;
}
/* channel blocks of a C-channel statistics vector = flag words of its exchange (csrc/sync_dev.hpp) */
static int channel_blocks(int C) {
  if (C < 4 || (C & (C - 1)) || C > 1024) return 1;
  return C >= 256 ? 4 : (C >= 128 ? 2 : 1);
}

int skd_sync_handle_bytes(void) { return HANDLE_BYTES; }
int skd_sync_max_floats(void) { return MAX_FLOATS; }

static float *slot_of(float *mailbox, int world, int parity, int writer) {
  return mailbox + ((size_t)parity * world + writer) * SLOT_FLOATS;
}

void *skd_sync_create(int world, int rank, void *handle_out) {
  static unsigned counter = 0;
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world || !handle_out) return NULL;
  sync_ctx *c = (sync_ctx *)calloc(1, sizeof *c);
  if (!c) return NULL;
  c->world = world;
  c->rank = rank;
  c->timeout_s = 5.0;
  c->bytes = sizeof(float) * 2 * (size_t)world * SLOT_FLOATS;
  snprintf(c->name, sizeof c->name, "/skdsync_%ld_%u", (long)getpid(), counter++);
  const int fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
    if (fd >= 0) { close(fd); shm_unlink(c->name); }
    free(c);
    return NULL;
  }
  void *p = mmap(NULL, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { shm_unlink(c->name); free(c); return NULL; }
  memset(p, 0, c->bytes);
  c->mail[rank] = (float *)p;
  memset(handle_out, 0, HANDLE_BYTES);
  memcpy(handle_out, c->name, strlen(c->name) + 1);
  return c;
}

int skd_sync_connect(void *ctx, const void *all_handles) {
  sync_ctx *c = (sync_ctx *)ctx;
  if (!c || !all_handles) return 0;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    char name[HANDLE_BYTES];
    memcpy(name, (const char *)all_handles + (size_t)r * HANDLE_BYTES, HANDLE_BYTES);
    name[HANDLE_BYTES - 1] = 0;
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return 0;
    void *p = mmap(NULL, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 0;
    c->mail[r] = (float *)p;
  }
  return 1;
}

int skd_sync_destroy(void *ctx) {
  sync_ctx *c = (sync_ctx *)ctx;
  if (!c) return 0;
  for (int r = 0; r < c->world; ++r)
    if (c->mail[r]) munmap(c->mail[r], c->bytes);
  shm_unlink(c->name);
  free(c);
  return 1;
}

static int connected(const sync_ctx *c) {
  if (!c) return 0;
  for (int r = 0; r < c->world; ++r)
    if (!c->mail[r]) return 0;
  return 1;
}

int skd_sync_set_timeout(void *ctx, double seconds) {
  sync_ctx *c = (sync_ctx *)ctx;
  if (!c || !(seconds > 0.0) || seconds > 1e6) return 0;
  c->timeout_s = seconds;
  return 1;
}

/* steps (1)-(3) of csrc/sync_dev.hpp: payload into every mailbox, the sequence number into the `nwords` flag words (one per
 * channel block) of this writer's slot, wait for the same words of all writers in the own mailbox (bounded: NaN + status word) */
static int exchange(sync_ctx *c, unsigned seq, int n, const float *src, int nwords) {
  const int parity = (int)(seq & 1u);
  for (int r = 0; r < c->world; ++r) memcpy(slot_of(c->mail[r], c->world, parity, c->rank) + HEADER_FLOATS, src, sizeof(float) * (size_t)n);
  for (int r = 0; r < c->world; ++r)
    for (int w = 0; w < nwords; ++w)
      __atomic_store_n((unsigned *)slot_of(c->mail[r], c->world, parity, c->rank) + w, seq, __ATOMIC_RELEASE);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int r = 0; r < c->world; ++r) {
    for (int w = 0; w < nwords; ++w) {
      const unsigned *flag = (const unsigned *)slot_of(c->mail[c->rank], c->world, parity, r) + w;
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        /* csrc/sync_dev.hpp: a wait that has lasted 0.1 s gives up at once when an earlier exchange already timed out */
        if (dt > c->timeout_s || (dt > 0.1 && __atomic_load_n(&g_status[0], __ATOMIC_ACQUIRE) != 0u)) {
          { unsigned expected = 0u;       /* the first timed-out exchange's sequence number survives (csrc/sync_dev.hpp) */
            __atomic_compare_exchange_n(&g_status[0], &expected, seq | 0x80000000u, 0, __ATOMIC_RELEASE, __ATOMIC_RELAXED); }
          return 0;
        }
        usleep(20);
      }
    }
  }
  return 1;
}

static int gather_words(void *ctx, int n, const float *src, float *gathered, int nwords) {
  sync_ctx *c = (sync_ctx *)ctx;
  if (!connected(c) || n <= 0 || n > MAX_FLOATS || !src || !gathered) return 0;
  const unsigned seq = ++c->seq;
  const int good = exchange(c, seq, n, src, nwords);
  for (int r = 0; r < c->world; ++r)
    for (int i = 0; i < n; ++i) gathered[(size_t)r * n + i] = good ? slot_of(c->mail[c->rank], c->world, (int)(seq & 1u), r)[HEADER_FLOATS + i] : NAN;
  return 1;
}

int skd_sync_all_gather(void *ctx, int n, const float *src, float *gathered, stream_t st) {
  (void)st;
  return gather_words(ctx, n, src, gathered, 1);
}

int skd_abn_combine_stats(int G, int C, const float *gathered, const float *weights, int rank, float *mean, float *var, float *rm,
                          float *rv, float momentum, double n, stream_t st);

int skd_abn_sync_stats(void *ctx, int C, const float *stat, const float *weights, float *mean, float *var, float *running_mean,
                       float *running_var, float momentum, double n, stream_t st) {
  sync_ctx *c = (sync_ctx *)ctx;
  if (!connected(c) || C <= 0 || 2 * C > MAX_FLOATS || !stat || !mean || !var) return 0;
  float *g = (float *)malloc(sizeof(float) * (size_t)c->world * 2 * C);
  if (!g) return 0;
  int r = gather_words(ctx, 2 * C, stat, g, channel_blocks(C));
  r = r && skd_abn_combine_stats(c->world, C, g, weights, c->rank, mean, var, running_mean, running_var, momentum, n, st);
  free(g);
  return r;
}

int skd_abn_sync_grad_stats(void *ctx, int C, float *stat, const float *weights, stream_t st) {
  (void)st;
  sync_ctx *c = (sync_ctx *)ctx;
  if (!connected(c) || C <= 0 || 2 * C > MAX_FLOATS || !stat) return 0;
  float *g = (float *)malloc(sizeof(float) * (size_t)c->world * 2 * C);
  if (!g) return 0;
  const int r = gather_words(ctx, 2 * C, stat, g, channel_blocks(C));
  for (int i = 0; r && i < 2 * C; ++i) {
    float s = 0.f;
    for (int q = 0; q < c->world; ++q) {
      const float v = g[(size_t)q * 2 * C + i];
      const float term = weights ? weights[q] * v : v;     /* product rounded to float, then added */
      s = s + term;
    }
    if (!weights) s /= (float)c->world;
    stat[i] = s;
  }
  free(g);
  return r;
}

static int64_t g_three_step_calls = 0;

/* ---- InPlaceABNSync for channels-last tensors in one call (include/skd.h section 2, "*_sync" entries): the product runs one
 *      register-resident launch with the exchange inside it when the tensor fits; the arithmetic is statistics -> exchange +
 *      combine -> normalise (forward) and reduce -> exchange + weighted sum -> dx (backward), restated here from the pieces ---- */
int64_t skd_abn_nhwc_workspace_floats(int64_t rows, int C);
int skd_abn_stats_nhwc(int64_t rows, int C, const float *x, float *mean, float *var, float *ws, stream_t st);
int skd_abn_apply_nhwc_to(int64_t rows, int C, const float *x, const float *residual, float *out, const float *mean, const float *var,
                          const float *weight, const float *bias, float eps, int act, float slope, stream_t st);
int skd_abn_backward_reduce_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *weight, const float *bias,
                                 float *edz, float *eydz, float eps, int act, float slope, float *ws, stream_t st);
int skd_abn_backward_dx_nhwc(int64_t rows, int C, const float *z, const float *dz, const float *var, const float *weight,
                             const float *bias, const float *edz, const float *eydz, float *dx, float *dweight, float *dbias, float eps,
                             int act, float slope, int accumulate, stream_t st);
int skd_abn_relu_backward_reduce_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout, const float *mean,
                                      const float *var, float *edz, float *eydz, float eps, float *ws, stream_t st);
int skd_abn_relu_backward_dx_nhwc(int64_t rows, int C, const float *x, const float *out, const float *dout, const float *mean,
                                  const float *var, const float *weight, const float *edz, const float *eydz, float *dx, float *dres,
                                  float *dweight, float *dbias, float eps, int accumulate, stream_t st);
int skd_abn_relu_backward_reduce_nhwc_x(int64_t rows, int C, const float *x, const float *dout, const float *mean, const float *var,
                                        const float *weight, const float *bias, float *edz, float *eydz, float eps, float *ws, stream_t st);
int skd_abn_relu_backward_dx_nhwc_x(int64_t rows, int C, const float *x, const float *dout, const float *mean, const float *var,
                                    const float *weight, const float *bias, const float *edz, const float *eydz, float *dx,
                                    float *dweight, float *dbias, float eps, int accumulate, stream_t st);

int skd_abn_forward_train_nhwc_sync(void *ctx, int64_t rows, int C, const float *x, const float *residual, float *out,
                                    const float *weight, const float *bias, float *rm, float *rv, float *mean, float *var,
                                    const float *replica_weights, float momentum, float eps, int act, float slope, double n, float *ws,
                                    stream_t st) {
  if (!ctx || !x || !out || !mean || !var) return 0;
  ++g_three_step_calls;
  float *local = (float *)malloc(sizeof(float) * 2 * (size_t)C);
  if (!local) return 0;
  int r = skd_abn_stats_nhwc(rows, C, x, local, local + C, ws, st);
  r = r && skd_abn_sync_stats(ctx, C, local, replica_weights, mean, var, rm, rv, momentum, n, st);
  free(local);
  return r && skd_abn_apply_nhwc_to(rows, C, x, residual, out, mean, var, weight, bias, eps, act, slope, st);
}

int skd_abn_backward_nhwc_sync(void *ctx, int64_t rows, int C, const float *z, const float *dz, const float *var, const float *weight,
                               const float *bias, float *edz, float *eydz, float *dx, float *dweight, float *dbias,
                               const float *replica_weights, float eps, int act, float slope, int accumulate, float *ws, stream_t st) {
  if (!ctx || eydz != edz + C) return 0;
  ++g_three_step_calls;
  if (!skd_abn_backward_reduce_nhwc(rows, C, z, dz, weight, bias, edz, eydz, eps, act, slope, ws, st)) return 0;
  if (!skd_abn_sync_grad_stats(ctx, C, edz, replica_weights, st)) return 0;
  return skd_abn_backward_dx_nhwc(rows, C, z, dz, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, act, slope, accumulate, st);
}

int skd_abn_relu_backward_nhwc_sync(void *ctx, int64_t rows, int C, const float *x, const float *out, const float *dout,
                                    const float *mean, const float *var, const float *weight, const float *bias, float *edz, float *eydz,
                                    float *dx, float *dres, float *dweight, float *dbias, const float *replica_weights, float eps,
                                    int accumulate, float *ws, stream_t st) {
  if (!ctx || eydz != edz + C || (out == NULL && dres != NULL)) return 0;
  ++g_three_step_calls;
  if (out == NULL) {
    if (!skd_abn_relu_backward_reduce_nhwc_x(rows, C, x, dout, mean, var, weight, bias, edz, eydz, eps, ws, st)) return 0;
    if (!skd_abn_sync_grad_stats(ctx, C, edz, replica_weights, st)) return 0;
    return skd_abn_relu_backward_dx_nhwc_x(rows, C, x, dout, mean, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, accumulate, st);
  }
  if (!skd_abn_relu_backward_reduce_nhwc(rows, C, x, out, dout, mean, var, edz, eydz, eps, ws, st)) return 0;
  if (!skd_abn_sync_grad_stats(ctx, C, edz, replica_weights, st)) return 0;
  return skd_abn_relu_backward_dx_nhwc(rows, C, x, out, dout, mean, var, weight, edz, eydz, dx, dres, dweight, dbias, eps, accumulate, st);
}

/* the grid-barrier cap of the product's one-launch passes has no host counterpart: accepted, reports "whole device" */
int skd_abn_set_fused_max_workgroups(int n) { (void)n; return 256; }

/* ---- the training stem fused (include/skd.h section 12, round 6): restated as the op SEQUENCE it replaces -- normalise + ReLU,
 *      then the max-pool (networks/pspnet_combine.py:176-180); backward: un-pool, then the BatchNorm + ReLU backward passes ---- */
int skd_maxpool3x3s2_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, float *y, uint8_t *arg, stream_t st);
int skd_maxpool3x3s2_backward_nhwc(int B, int C, int H, int W, int OH, int OW, const float *dy, const uint8_t *arg, float *dx,
                                   stream_t st);

int skd_abn_relu_maxpool3x3s2_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, const float *mean, const float *var,
                                   const float *weight, const float *bias, float eps, float *pooled, uint8_t *arg, stream_t st) {
  if (B <= 0 || C < 4 || (C & (C - 1)) || C > 1024 || !x || !mean || !var || !pooled || !arg) return 0;
  const int64_t rows = (int64_t)B * H * W;
  float *y = (float *)malloc(sizeof(float) * (size_t)rows * C);
  if (!y) return 0;
  int ok_ = skd_abn_apply_nhwc_to(rows, C, x, NULL, y, mean, var, weight, bias, eps, 3 /* ReLU */, 0.f, st);
  if (ok_) ok_ = skd_maxpool3x3s2_nhwc(B, C, H, W, OH, OW, y, pooled, arg, st);
  free(y);
  return ok_;
}

int skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, const float *dpooled,
                                                   const uint8_t *arg, const float *mean, const float *var, const float *weight,
                                                   const float *bias, float *edz, float *eydz, float eps, float *ws, stream_t st) {
  if (B <= 0 || C < 4 || (C & (C - 1)) || C > 1024 || !x || !dpooled || !arg || !mean || !var || !edz || !eydz || !ws) return 0;
  const int64_t rows = (int64_t)B * H * W;
  float *dy = (float *)malloc(sizeof(float) * (size_t)rows * C);
  if (!dy) return 0;
  int ok_ = skd_maxpool3x3s2_backward_nhwc(B, C, H, W, OH, OW, dpooled, arg, dy, st);
  if (ok_) ok_ = skd_abn_relu_backward_reduce_nhwc_x(rows, C, x, dy, mean, var, weight, bias, edz, eydz, eps, ws, st);
  free(dy);
  return ok_;
}

int skd_abn_relu_maxpool3x3s2_backward_dx_nhwc(int B, int C, int H, int W, int OH, int OW, const float *x, const float *dpooled,
                                               const uint8_t *arg, const float *mean, const float *var, const float *weight,
                                               const float *bias, const float *edz, const float *eydz, float *dx, float *dweight,
                                               float *dbias, float eps, int accumulate, stream_t st) {
  if (B <= 0 || C < 4 || (C & (C - 1)) || C > 1024 || !x || !dpooled || !arg || !mean || !var || !edz || !eydz || !dx) return 0;
  const int64_t rows = (int64_t)B * H * W;
  float *dy = (float *)malloc(sizeof(float) * (size_t)rows * C);
  if (!dy) return 0;
  int ok_ = skd_maxpool3x3s2_backward_nhwc(B, C, H, W, OH, OW, dpooled, arg, dy, st);
  if (ok_) ok_ = skd_abn_relu_backward_dx_nhwc_x(rows, C, x, dy, mean, var, weight, bias, edz, eydz, dx, dweight, dbias, eps, accumulate, st);
  free(dy);
  return ok_;
}

/* the two switches of the one-launch passes (include/skd.h section 13): the host double keeps the STATE (environment read once,
 * setters override) so that the host logic above it can be exercised; it always runs the three-step form itself */
static int g_fused_state = -1, g_sync_fused_state = -1;
static int switch_state(int *state, const char *name) {
  if (*state < 0) {
    const char *e = getenv(name);
    *state = !(e != NULL && e[0] == '0');
  }
  return *state;
}
int skd_abn_set_fused(int on) { g_fused_state = on < 0 ? -1 : (on != 0); return 1; }
int skd_abn_get_fused(void) { return switch_state(&g_fused_state, "SKD_ABN_FUSED"); }
int skd_abn_set_sync_fused(int on) { g_sync_fused_state = on < 0 ? -1 : (on != 0); return 1; }
int skd_abn_get_sync_fused(void) { return switch_state(&g_sync_fused_state, "SKD_ABN_SYNC_FUSED"); }

/* the host double always runs the three-step form */
int skd_abn_sync_form_counts(int64_t *out) {
  if (!out) return 0;
  out[0] = 0;
  out[1] = g_three_step_calls;
  return 1;
}
