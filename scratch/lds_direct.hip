// Experiment: fp32 MFMA GEMM C[M][N] = A[M][K] * B[N][K]^T with global -> LDS loads that bypass the
// registers (global_load_lds_dwordx4), XOR-swizzled 128-byte rows, three LDS stages.
// Build: hipcc -O3 --offload-arch=gfx950 scratch/lds_direct.hip -o scratch/lds_direct
// Run:   scratch/lds_direct          (prints max |err| on sampled entries and TF/s)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 64, BN = 128, BK = 32, NT = 256;
constexpr int STAGE_FLOATS = (BM + BN) * BK;   // 6144 floats = 24 KB
constexpr int NSTAGE = 3;

__global__ void __launch_bounds__(NT, 2)
gemm_lds_direct(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                int N, int K) {
  extern __shared__ __attribute__((aligned(1024))) float lds[];   // NSTAGE * STAGE_FLOATS floats (72 KB)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;
  const long m0 = (long)blockIdx.x * BM, n0 = (long)blockIdx.y * BN;

  // this wave's 6 chunks of 8 rows x 128 B per stage: chunk c < 8 -> A rows 8c.., else B rows 8(c-8)..
  const float* src[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int c = wave * 6 + j;
    const int r = (c < 8 ? c * 8 : (c - 8) * 8) + (lane >> 3);
    const int slot = (lane & 7) ^ (r & 7);            // logical k slot stored at physical slot lane&7
    const float* base = c < 8 ? A + (m0 + r) * (long)K : B + (n0 + r) * (long)K;
    src[j] = base + slot * 4;
  }
  auto issue = [&](int step, int stage) {
    float* st = lds + stage * STAGE_FLOATS;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int c = wave * 6 + j;
#ifndef NO_LOADS
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + step * BK), (lptr_t)(st + c * 256), 16, 0, 0);
#else
      (void)st; (void)c;
#endif
    }
  };

  floatx16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  // LDS reads as inline asm with hand-placed waits: the compiler's own model drains vmcnt to 0 in
  // front of every LDS read that may alias an LDS-DMA write, and __syncthreads() carries a fence
  // that does the same
  auto rd = [](unsigned addr) -> f4v {
    f4v v;
#ifdef NO_LDS
    v = f4v{1.f, 2.f, 3.f, (float)addr};      // no LDS read at all: what the MFMA stream alone takes
#else
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
#endif
    return v;
  };
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;
  const int ra = wm0 + l31, rb0 = wn0 + l31, rb1 = wn0 + 32 + l31;
  unsigned offa[4], offb0[4], offb1[4];   // byte offsets inside a stage, per group
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    offa[g] = 4u * (ra * BK + 4 * ((2 * g + half) ^ (ra & 7)));
    offb0[g] = 4u * (BM * BK + rb0 * BK + 4 * ((2 * g + half) ^ (rb0 & 7)));
    offb1[g] = 4u * (BM * BK + rb1 * BK + 4 * ((2 * g + half) ^ (rb1 & 7)));
  }
  const int nsteps = K / BK;
  issue(0, 0);
  issue(1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nsteps; ++t) {
    if (t + 2 < nsteps) issue(t + 2, (t + 2) % NSTAGE);
    const unsigned sb = lds0 + 4u * (t % NSTAGE) * STAGE_FLOATS;
    f4v a[4], b0[4], b1[4];
    a[0] = rd(sb + offa[0]); b0[0] = rd(sb + offb0[0]); b1[0] = rd(sb + offb1[0]);
    a[1] = rd(sb + offa[1]); b0[1] = rd(sb + offb0[1]); b1[1] = rd(sb + offb1[1]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 2 < 4) { a[g + 2] = rd(sb + offa[g + 2]); b0[g + 2] = rd(sb + offb0[g + 2]); b1[g + 2] = rd(sb + offb1[g + 2]); }
      __builtin_amdgcn_sched_barrier(0);
      // reads complete in order: all but the youngest 6 / 3 / 0 are done
      if (g < 2) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[g]), "+v"(b0[g]), "+v"(b1[g]));
      else if (g == 2) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[g]), "+v"(b0[g]), "+v"(b1[g]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[g]), "+v"(b0[g]), "+v"(b1[g]));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][q], b0[g][q], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][q], b1[g][q], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef NO_VMWAIT
    if (t + 2 < nsteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long col = n0 + wn0 + j * 32 + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long row = m0 + wm0 + (e & 3) + 8 * (e >> 2) + 4 * half;
      if (row < M && col < N) C[row * N + col] = acc[j][e];
    }
  }
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main() {
  const int M = 13312, N = 256, K = 1216;
  std::vector<float> a((size_t)M * K), b((size_t)N * K);
  srand(1);
  for (auto& v : a) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : b) v = (rand() % 2001 - 1000) / 1000.f;
  float *da, *db, *dc;
  CK(hipMalloc(&da, a.size() * 4));
  CK(hipMalloc(&db, b.size() * 4));
  CK(hipMalloc(&dc, (size_t)M * N * 4));
  CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  dim3 grid(M / BM, N / BN);
  const size_t lds_bytes = (size_t)NSTAGE * STAGE_FLOATS * 4;
  CK(hipFuncSetAttribute((const void*)gemm_lds_direct, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  gemm_lds_direct<<<grid, NT, lds_bytes>>>(da, db, dc, M, N, K);
  CK(hipDeviceSynchronize());
  std::vector<float> c((size_t)M * N);
  CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int s = 0; s < 4000; ++s) {
    const int r = rand() % M, col = rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)a[(size_t)r * K + k] * b[(size_t)col * K + k];
    maxerr = fmax(maxerr, fabs(ref - c[(size_t)r * N + col]));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) gemm_lds_direct<<<grid, NT, lds_bytes>>>(da, db, dc, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("max|err| %.3g   %.1f us   %.1f TF/s   (%d tiles)\n", maxerr, us, 2.0 * M * N * K / us * 1e-6,
         grid.x * grid.y);
  return maxerr < 1e-2 ? 0 : 2;
}
