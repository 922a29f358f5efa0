// gemm_lab.cpp -- standalone timing / checking harness for the fp32-MFMA GEMM kernels of libskd_hip.so (no Python, no torch:
// starts in milliseconds on a gpurun box, so one call can time many variants and be wrapped in rocprofv3 --pmc cheaply).
//
//   build (here, cross-compiled):  tools/build_gemm_lab.sh     -> tools/gemm_lab (git-ignored, travels with gpurun)
//   run   (GPU box):               tools/gemm_lab [time|pmc] [reps] [filter] [variants-only]
//                                  (filter: substring of the case name; variants-only: just the experiment kernels of
//                                  tools/gemm_lab_kernels.hip)
//
// Cases: the pair-wise Gram / backward GEMMs at M = 1089 and 4225 (B = 8, Cs = 128, Ct = 512), the 1x1-convolution +
// InPlace-ABN GEMM (skd_conv1x1_abn_nhwc) at the frozen teacher's shapes, and -- as the library baseline the fused kernel
// has to beat -- hipBLASLt's matmul with the same work folded into its epilogue (D = relu(W x + bias [+ residual])).
// conv1x1 results are spot-checked against a double-precision host evaluation of the formula in include/skd.h.
// One JSON line per case.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "skd.h"

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
      exit(2);                                                                                 \
    }                                                                                          \
  } while (0)
#define CKB(x)                                                                    \
  do {                                                                            \
    hipblasStatus_t s_ = (x);                                                     \
    if (s_ != HIPBLAS_STATUS_SUCCESS) {                                           \
      fprintf(stderr, "%s:%d %s -> hipblas status %d\n", __FILE__, __LINE__, #x, (int)s_); \
      exit(3);                                                                    \
    }                                                                             \
  } while (0)

static uint32_t g_seed = 12345u;
static float frand() {  // uniform (-1, 1), xorshift
  g_seed ^= g_seed << 13;
  g_seed ^= g_seed >> 17;
  g_seed ^= g_seed << 5;
  return (float)((g_seed >> 8) * (1.0 / 8388608.0) - 1.0);
}
static float *dev_random(size_t n, float scale, std::vector<float> *keep = nullptr) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = frand() * scale;
  float *d = nullptr;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  if (keep) keep->swap(h);
  return d;
}
static float *dev_empty(size_t n) {
  float *d = nullptr;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemset(d, 0xff, n * sizeof(float)));
  return d;
}

struct Timer {
  hipEvent_t a, b;
  Timer() {
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
  }
  template <class F>
  double us(F &&f, int reps) {
    // round 6: warm up for >= 40 ms of GPU time, not for two launches.  Cases that follow a host-side check (a 100+ MB read-back and
    // a second of host arithmetic: the GPU idles and drops its clocks) were timed 5-15 % slower than cases that follow another
    // timing loop -- and the variants WITHOUT a check ("no-gload", "mfma-only", "no-check") looked that much faster than they are.
    float ms = 0.f;
    for (int round = 0; round < 50; ++round) {
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < 4; ++i) f();
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float part = 0.f;
      CK(hipEventElapsedTime(&part, a, b));
      ms += part;
      if (ms >= 40.f) break;
    }
    // median of five groups of `reps` launches
    std::vector<float> t;
    for (int g = 0; g < 5; ++g) {
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < reps; ++i) f();
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
      t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[2] * 1e3 / reps;
  }
};

static bool g_pmc = false;
static int g_reps = 10;
static const char *g_filter = "";
static Timer *g_timer;

template <class F>
static void run_case(const std::string &name, double flops, double bytes, F &&f, const std::string &extra = "") {
  if (g_filter[0] && name.find(g_filter) == std::string::npos) return;
  if (g_pmc) {
    f();
    f();
    CK(hipDeviceSynchronize());
    printf("{\"case\": \"%s\", \"pmc\": true}\n", name.c_str());
    return;
  }
  const double us = g_timer->us(f, g_reps);
  printf("{\"case\": \"%s\", \"us\": %.2f", name.c_str(), us);
  if (flops > 0) printf(", \"tflops\": %.2f, \"frac_fp32_mfma\": %.4f", flops / us * 1e-6, flops / us * 1e-6 / 157.3);
  if (bytes > 0) printf(", \"GBs\": %.1f", bytes / us * 1e-3);
  printf("%s}\n", extra.c_str());
  fflush(stdout);
}

// ------------------------------------------------------------------------------------------------------------------
static void pairwise_cases() {
  const int B = 8, Cs = 128, Ct = 512;
  for (int M : {1089, 4225}) {
    const int ldm = skd_pairwise_ldm(M);
    float *ps = dev_random((size_t)B * Cs * M, 1.f), *pt = dev_random((size_t)B * Ct * M, 1.f);
    float *fs = dev_empty((size_t)B * Cs * ldm), *ft = dev_empty((size_t)B * Ct * ldm), *nrm = dev_empty((size_t)B * M);
    float *G = dev_empty((size_t)B * ldm * ldm), *loss = dev_empty(4), *dp = dev_empty((size_t)B * Cs * ldm);
    float *ws = dev_empty((size_t)skd_pairwise_workspace_floats(B, M) + 4);
    float *bws = dev_empty((size_t)skd_pairwise_backward_workspace_floats(B, Cs, M) + 4);
    std::vector<float> one(1, 1.f);
    float *gl = nullptr;
    CK(hipMalloc(&gl, 4));
    CK(hipMemcpy(gl, one.data(), 4, hipMemcpyHostToDevice));
    if (!skd_channel_l2_normalise(B, Cs, M, ps, fs, ldm, nullptr, 0, nrm, nullptr) ||
        !skd_channel_l2_normalise(B, Ct, M, pt, ft, ldm, nullptr, 0, nullptr, nullptr)) {
      fprintf(stderr, "l2_normalise failed\n");
      exit(4);
    }
    const double nt = ldm / 128, tri = nt * (nt + 1) / 2 / (nt * nt);
    char tag[64], extra[160];
    snprintf(tag, sizeof tag, "M=%d", M);
    const double full = 2.0 * B * (double)M * M * (Cs + Ct);
    snprintf(extra, sizeof extra, ", \"executed_fraction_of_full_matrix\": %.4f, \"convention\": \"flops of the padded upper triangle actually issued\"", tri);
    const double issued = 2.0 * B * (double)ldm * ldm * (Cs + Ct) * tri;
    (void)full;
    run_case(std::string("pairwise_gram_loss ") + tag, issued, 0, [&] { skd_pairwise_gram_loss(B, Cs, Ct, M, ldm, fs, ft, G, loss, ws, nullptr); }, extra);
    run_case(std::string("pairwise_gram_loss(no G) ") + tag, issued, 0, [&] { skd_pairwise_gram_loss(B, Cs, Ct, M, ldm, fs, ft, nullptr, loss, ws, nullptr); }, extra);
    const double bw = 2.0 * B * (double)ldm * ldm * 128;   // Cs padded to the 128-row tile
    run_case(std::string("pairwise_backward ") + tag, bw, 0, [&] { skd_pairwise_backward(B, Cs, M, ldm, fs, G, nrm, gl, dp, bws, nullptr); });
    CK(hipDeviceSynchronize());
    for (float *p : {ps, pt, fs, ft, nrm, G, loss, dp, ws, bws, gl}) CK(hipFree(p));
  }
}

// ------------------------------------------------------------------------------------------------------------------
struct ConvShape {
  int64_t M;
  int K, N;
  bool res;
  int count;  // calls per step of the batch-8 512x512 teacher forward
};

static hipblasLtHandle_t g_lt;

// D (M x N row-major) = relu(X (M x K) W^T (K x N) + bias[n] [+ R]).  Column-major view: D^T (N x M) = W (N x K, stored K-major =
// op T on a K x N column-major matrix with ld K) * X^T (K x M column-major, ld K) -- bias along D^T's rows = n.
struct LtGemm {
  hipblasLtMatmulDesc_t desc;
  hipblasLtMatrixLayout_t a, b, c, d;
  hipblasLtMatmulHeuristicResult_t heur;
  void *wsp = nullptr;
  size_t wsz = 64u << 20;
  bool ok = false;
  LtGemm(int64_t M, int K, int N, const float *bias, bool relu) {
    CKB(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    CKB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta));
    CKB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb));
    hipblasLtEpilogue_t ep = relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS;
    CKB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof ep));
    CKB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias));
    CKB(hipblasLtMatrixLayoutCreate(&a, HIP_R_32F, K, N, K));   // W as K x N column-major (ld K), transposed by op
    CKB(hipblasLtMatrixLayoutCreate(&b, HIP_R_32F, K, M, K));   // X^T
    CKB(hipblasLtMatrixLayoutCreate(&c, HIP_R_32F, N, M, N));
    CKB(hipblasLtMatrixLayoutCreate(&d, HIP_R_32F, N, M, N));
    hipblasLtMatmulPreference_t pref;
    CKB(hipblasLtMatmulPreferenceCreate(&pref));
    CKB(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof wsz));
    int found = 0;
    hipblasStatus_t s = hipblasLtMatmulAlgoGetHeuristic(g_lt, desc, a, b, c, d, pref, 1, &heur, &found);
    ok = (s == HIPBLAS_STATUS_SUCCESS && found > 0);
    CK(hipMalloc(&wsp, wsz));
  }
  void run(const float *W, const float *X, const float *R, float *D) {
    const float alpha = 1.f, beta = R ? 1.f : 0.f;
    CKB(hipblasLtMatmul(g_lt, desc, &alpha, W, a, X, b, &beta, R ? R : D, c, D, d, &heur.algo, wsp, wsz, 0));
  }
};

static void conv_cases() {
  const ConvShape shapes[] = {{33800, 256, 1024, true, 23},  {33800, 1024, 256, false, 22}, {33800, 512, 2048, true, 3},
                              {33800, 2048, 512, false, 2},  {33800, 128, 512, true, 4},   {33800, 512, 128, false, 3},
                              {133128, 64, 256, true, 3},    {133128, 256, 128, false, 1}, {33800, 1024, 2048, false, 1},
                              {33800, 512, 2048, false, 3}, {32768, 256, 1024, true, 0}, {36864, 256, 1024, true, 0}};
  const float eps = 1e-5f;
  for (const ConvShape &s : shapes) {
    std::vector<float> hx, hw, hr, hm, hv, hg, hb;
    float *x = dev_random((size_t)s.M * (s.K + 32), 1.f, &hx), *w = dev_random((size_t)s.N * (s.K + 32), 0.05f, &hw);   // (+32: the padded-stride variants)
    float *r = dev_random((size_t)s.M * s.N, 1.f, &hr), *out = dev_empty((size_t)s.M * s.N);
    float *mean = dev_random(s.N, 0.2f, &hm), *var = dev_random(s.N, 0.4f, &hv), *gam = dev_random(s.N, 1.f, &hg), *bet = dev_random(s.N, 0.5f, &hb);
    {  // var must be positive
      for (float &v : hv) v = fabsf(v) + 0.3f;
      CK(hipMemcpy(var, hv.data(), s.N * sizeof(float), hipMemcpyHostToDevice));
    }
    char tag[96];
    snprintf(tag, sizeof tag, "M=%lld K=%d N=%d%s x%d", (long long)s.M, s.K, s.N, s.res ? " +res" : "", s.count);
    const double flops = 2.0 * s.M * s.K * s.N;
    const double bytes = 4.0 * ((double)s.M * s.K + (double)s.N * s.K + (double)s.M * s.N * (s.res ? 2 : 1));
    auto check = [&](const char *what, bool folded) {
      std::vector<float> ho((size_t)s.M * s.N);
      CK(hipMemcpy(ho.data(), out, ho.size() * sizeof(float), hipMemcpyDeviceToHost));
      double worst = 0;
      uint32_t sd = 777;
      for (int t = 0; t < 1024; ++t) {
        sd = sd * 1664525u + 1013904223u;
        const int64_t m = t < 64 ? (t < 32 ? t * 4 % s.M : s.M - 1 - t % 32) : (int64_t)(sd >> 4) % s.M;
        sd = sd * 1664525u + 1013904223u;
        const int n = (int)((sd >> 4) % (uint32_t)s.N);
        double acc = 0;
        for (int k = 0; k < s.K; ++k) acc += (double)hx[(size_t)m * s.K + k] * (double)hw[(size_t)n * s.K + k];
        double z = ((acc - hm[n]) / sqrt((double)hv[n] + eps)) * ((double)fabsf(hg[n]) + eps) + hb[n];
        if (s.res) z += hr[(size_t)m * s.N + n];
        if (z < 0) z = 0;
        const double err = fabs(z - (double)ho[(size_t)m * s.N + n]) / (fabs(z) + 1.0);
        if (err > worst) worst = err;
      }
      char buf[96];
      snprintf(buf, sizeof buf, ", \"max_err_%s\": %.2e", what, worst);
      (void)folded;
      return std::string(buf);
    };
    if (skd_conv1x1_abn_supported(s.M, s.K, s.N)) {
      CK(hipMemset(out, 0xff, (size_t)s.M * s.N * sizeof(float)));
      int rc = skd_conv1x1_abn_nhwc(s.M, s.K, s.N, x, w, s.res ? r : nullptr, out, mean, var, gam, bet, eps, SKD_ACT_RELU, 0.f, nullptr);
      CK(hipDeviceSynchronize());
      std::string ex = check("vs_fp64", false) + (rc ? "" : ", \"rc\": 0");
      run_case(std::string("conv1x1_abn ") + tag, flops, bytes,
               [&] { skd_conv1x1_abn_nhwc(s.M, s.K, s.N, x, w, s.res ? r : nullptr, out, mean, var, gam, bet, eps, SKD_ACT_RELU, 0.f, nullptr); }, ex);
    }
    if (s.res && skd_conv1x1_abn_supported(s.M, s.K, s.N)) {   // with the preceding BN + ReLU applied to x on the way in (timing only:
      std::vector<float> hpv;                                    // parity is tests/test_kernels_gpu.py's job)
      float *pm = dev_random(s.K, 0.2f), *pv = dev_random(s.K, 0.4f, &hpv), *pw = dev_random(s.K, 1.f), *pb = dev_random(s.K, 0.5f);
      for (float &v : hpv) v = fabsf(v) + 0.3f;
      CK(hipMemcpy(pv, hpv.data(), s.K * sizeof(float), hipMemcpyHostToDevice));
      float *pk = dev_empty((size_t)4 * s.K);
      skd_abn_pack_eval_params(s.K, pm, pv, pw, pb, eps, pk, nullptr);
      run_case(std::string("conv1x1_abn_pro ") + tag, flops, bytes, [&] {
        skd_conv1x1_abn_pro_nhwc(s.M, s.K, s.N, x, w, r, out, mean, var, gam, bet, eps, pk, SKD_ACT_RELU, 0.f, nullptr);
      });
      CK(hipDeviceSynchronize());
      for (float *p : {pm, pv, pw, pb, pk}) CK(hipFree(p));
    }
    {  // hipBLASLt, BN folded into the weights and a bias: w' = w * gamma / sigma, b' = beta - mean * gamma / sigma
      std::vector<float> wf((size_t)s.N * s.K), bf(s.N);
      for (int n = 0; n < s.N; ++n) {
        const double sc = ((double)fabsf(hg[n]) + eps) / sqrt((double)hv[n] + eps);
        for (int k = 0; k < s.K; ++k) wf[(size_t)n * s.K + k] = (float)(hw[(size_t)n * s.K + k] * sc);
        bf[n] = (float)(hb[n] - hm[n] * sc);
      }
      float *w2 = nullptr, *b2 = nullptr;
      CK(hipMalloc(&w2, wf.size() * 4));
      CK(hipMalloc(&b2, bf.size() * 4));
      CK(hipMemcpy(w2, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(b2, bf.data(), bf.size() * 4, hipMemcpyHostToDevice));
      LtGemm lt(s.M, s.K, s.N, b2, true);
      if (lt.ok) {
        CK(hipMemset(out, 0xff, (size_t)s.M * s.N * sizeof(float)));
        lt.run(w2, x, s.res ? r : nullptr, out);
        CK(hipDeviceSynchronize());
        std::string ex = check("vs_fp64", true);
        run_case(std::string("hipblaslt_bias_relu ") + tag, flops, bytes, [&] { lt.run(w2, x, s.res ? r : nullptr, out); }, ex);
      } else {
        printf("{\"case\": \"hipblaslt_bias_relu %s\", \"error\": \"no heuristic\"}\n", tag);
      }
      CK(hipDeviceSynchronize());
      CK(hipFree(w2));
      CK(hipFree(b2));
      CK(hipFree(lt.wsp));
    }
    CK(hipDeviceSynchronize());
    for (float *p : {x, w, r, out, mean, var, gam, bet}) CK(hipFree(p));
  }
}

extern "C" int lab_tn_gemm(int variant, const float *X, const float *Wt, float *Y, int64_t M, int K, int N);
extern "C" const char *lab_tn_name(int variant);

// experiment variants of the TN GEMM core (tools/gemm_lab_kernels.hip): plain C = X W^T, main-loop decomposition and tile shapes
static void variant_cases() {
  const ConvShape shapes[] = {{33800, 1024, 256, false, 22}, {33800, 256, 1024, false, 23}, {33800, 1024, 2048, false, 1},
                              {33800, 512, 2048, false, 3},
                              // round 6: the SAME problem with a tile count that divides the chip (256 x 8 tiles of 128 x 128 = 8 per CU,
                              // against 2120 = 8.28 per CU at M = 33800): what does tile quantisation cost?
                              {32768, 256, 1024, false, 0}, {36864, 256, 1024, false, 0}};
  for (const ConvShape &s : shapes) {
    std::vector<float> hx, hw;
    float *x = dev_random((size_t)s.M * (s.K + 32), 1.f, &hx), *w = dev_random((size_t)s.N * (s.K + 32), 0.05f, &hw);   // (+32: the padded-stride variants)
    float *out = dev_empty((size_t)s.M * s.N);
    const double flops = 2.0 * s.M * s.K * s.N;
    for (int v = 0; lab_tn_name(v) != nullptr; ++v) {
      char tag[128];
      snprintf(tag, sizeof tag, "variant %d [%s] M=%lld K=%d N=%d", v, lab_tn_name(v), (long long)s.M, s.K, s.N);
      if (g_filter[0] && std::string(tag).find(g_filter) == std::string::npos) continue;
      CK(hipMemset(out, 0xff, (size_t)s.M * s.N * sizeof(float)));
      const int rc = lab_tn_gemm(v, x, w, out, s.M, s.K, s.N);
      if (rc != 1) {
        printf("{\"case\": \"%s\", \"skipped\": %d}\n", tag, rc);
        continue;
      }
      CK(hipDeviceSynchronize());
      std::string ex;
      const bool exact = strstr(lab_tn_name(v), "no-") == nullptr && strstr(lab_tn_name(v), "only") == nullptr;
      if (exact) {
        std::vector<float> ho((size_t)s.M * s.N);
        CK(hipMemcpy(ho.data(), out, ho.size() * sizeof(float), hipMemcpyDeviceToHost));
        double worst = 0;
        uint32_t sd = 4242;
        for (int t = 0; t < 512; ++t) {
          sd = sd * 1664525u + 1013904223u;
          const int64_t m = t < 32 ? s.M - 1 - t : (int64_t)(sd >> 4) % s.M;
          sd = sd * 1664525u + 1013904223u;
          const int n = (int)((sd >> 4) % (uint32_t)s.N);
          double acc = 0;
          for (int k = 0; k < s.K; ++k) acc += (double)hx[(size_t)m * s.K + k] * (double)hw[(size_t)n * s.K + k];
          const double err = fabs(acc - (double)ho[(size_t)m * s.N + n]) / (fabs(acc) + 1.0);
          if (err > worst) worst = err;
        }
        char buf[64];
        snprintf(buf, sizeof buf, ", \"max_err_vs_fp64\": %.2e", worst);
        ex = buf;
      }
      run_case(tag, flops, 0, [&] { lab_tn_gemm(v, x, w, out, s.M, s.K, s.N); }, ex);
    }
    CK(hipDeviceSynchronize());
    for (float *p : {x, w, out}) CK(hipFree(p));
  }
}

int main(int argc, char **argv) {
  if (argc > 1) g_pmc = strcmp(argv[1], "pmc") == 0;
  if (argc > 2) g_reps = atoi(argv[2]);
  if (argc > 3) g_filter = argv[3];
  if (skd_target_arch() != 950) {
    fprintf(stderr, "libskd_hip.so was not built for gfx950\n");
    return 1;
  }
  Timer t;
  g_timer = &t;
  CKB(hipblasLtCreate(&g_lt));
  if (!(argc > 4 && strcmp(argv[4], "variants-only") == 0)) {
    pairwise_cases();
    conv_cases();
  }
  variant_cases();
  return 0;
}
