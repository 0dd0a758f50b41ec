// Diagnostic: cycle stamps inside the fused logit-layer kernels (csrc/logit.hip built with
// SAGE_LOGIT_PROF), Reddit-like shape: 13,312 rows of 256, 512 seeds x 25 sources, 41 outputs.
// Build: hipcc -O3 --offload-arch=gfx950 -Iinclude -Igraphsage_amd/csrc scratch/logit_prof.hip -o scratch/logit_prof
#define SAGE_LOGIT_PROF 1
#include "../graphsage_amd/csrc/logit.hip"
#include "../graphsage_amd/csrc/api.hip"
// (logit.hip's fallback for weights beyond the LDS calls into linear.hip, which this harness does not link)
extern "C" int sage_linear_dweight2(const float*, sage_i64, const float*, sage_i64, const sage_i64*, int, float*, sage_i64,
                                    const float*, sage_i64, const sage_i64*, int, float*, sage_i64, sage_i64,
                                    const sage_i64*, int, void*) { return SAGE_EINVAL; }
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define SK(x) do { int r_ = (x); if (r_ != 0) { printf("sage error %d at %d: %s\n", r_, __LINE__, sage_last_error()); return 1; } } while (0)
int main() {
  const int T = 13312, H = 256, M = 512, K = 25, N = 41, LDW = 256;
  std::vector<float> h((size_t)T * H), w((size_t)2 * N * LDW);
  std::vector<sage_i64> seg(M + 1), src((size_t)M * K), lab(M);
  srand(3);
  for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : w) v = (rand() % 2001 - 1000) / 5000.f;
  for (int i = 0; i <= M; ++i) seg[i] = (sage_i64)i * K;
  for (auto& v : src) v = rand() % T;
  for (auto& v : lab) v = rand() % N;
  float *dh_, *dw, *agg, *y, *inv, *dz, *rl, *loss, *dgrad;
  sage_i64 *dseg, *dsrc, *dlab;
  CK(hipMalloc(&dh_, h.size() * 4)); CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&agg, (size_t)M * H * 4));
  CK(hipMalloc(&y, M * 64 * 4)); CK(hipMalloc(&inv, M * 4)); CK(hipMalloc(&dz, M * 64 * 4)); CK(hipMalloc(&rl, M * 4));
  CK(hipMalloc(&loss, 4)); CK(hipMalloc(&dgrad, h.size() * 4)); CK(hipMalloc(&dseg, seg.size() * 8));
  CK(hipMalloc(&dsrc, src.size() * 8)); CK(hipMalloc(&dlab, lab.size() * 8));
  CK(hipMemcpy(dh_, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dseg, seg.data(), seg.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dsrc, src.data(), src.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlab, lab.data(), lab.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(dgrad, 0, h.size() * 4));
  float* hroot; CK(hipMalloc(&hroot, (size_t)M * H * 4));
  const bool pre = getenv("PRE") != nullptr;      // PRE=1: the rows are pre-activation rows, normalised on their way in
  auto fwd = [&]() { return sage_logit_layer_fwd(dh_, H, dseg, dsrc, M, H, SAGE_REDUCE_MEAN, dw, LDW, dw + (size_t)N * LDW, LDW, N, dlab, nullptr, 1.f, agg, H, y, 64, inv, dz, 64, rl, pre ? hroot : nullptr, H, nullptr); };
  auto bwd = [&]() { return sage_logit_layer_bwd(dz, 64, dseg, dsrc, M, H, SAGE_REDUCE_MEAN, dw, LDW, dw + (size_t)N * LDW, LDW, N, dgrad, H, rl, loss, nullptr); };
  int* wr; CK(hipMalloc(&wr, (size_t)T * 4));
  SK(sage_count_row_writers(dseg, dsrc, M, (sage_i64)M * K, wr, T, nullptr));
  const bool solo = getenv("SOLO") != nullptr;    // SOLO=1: rows with one writer are stored (dgrad is not re-zeroed: timing only)
  auto fb = [&]() { return sage_logit_layer_fwd_bwd(dh_, H, dseg, dsrc, M, H, SAGE_REDUCE_MEAN, dw, LDW, dw + (size_t)N * LDW, LDW, N, dlab, nullptr, 1.f, agg, H, y, 64, inv, dz, 64, rl, pre ? hroot : nullptr, H, dgrad, H, solo ? wr : nullptr, nullptr); };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < 5; ++i) { SK(fwd()); SK(bwd()); }
    CK(hipEventRecord(e0)); for (int i = 0; i < 50; ++i) SK(fwd()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("fwd %.1f us\n", ms * 20);
    CK(hipEventRecord(e0)); for (int i = 0; i < 50; ++i) SK(bwd()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("bwd %.1f us\n", ms * 20);
  }
  for (int i = 0; i < 5; ++i) SK(fb());
  CK(hipEventRecord(e0)); for (int i = 0; i < 50; ++i) SK(fb()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1)); printf("fwd+bwd in one launch %.1f us\n", ms * 20);
  {
    unsigned long long s2[2][16];
    CK(hipMemcpyFromSymbol(s2, HIP_SYMBOL(sage::g_logit_prof), sizeof(s2)));
    printf("fwd+bwd stamps (x10 ns): "); for (int i = 1; i <= 9; ++i) printf("%lld ", (long long)(s2[0][i] - s2[0][0])); printf("\n");
    for (int i = 0; i < 3; ++i) { SK(fwd()); SK(bwd()); }
  }
  unsigned long long st[2][16];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(sage::g_logit_prof), sizeof(st)));
  // s_memtime counts at 100 MHz on gfx950
  printf("fwd stamps (x10 ns): "); for (int i = 1; i <= 6; ++i) printf("%lld ", (long long)(st[0][i] - st[0][0])); printf("\n");
  printf("bwd stamps (x10 ns): "); for (int i = 1; i <= 4; ++i) printf("%lld ", (long long)(st[1][i] - st[1][0])); printf("\n");
  return 0;
}
