// Experiment: fp32-accurate GEMM C[M][N] = A[M][K] * B[N][K]^T on the bf16 MFMA.
// Every fp32 operand is split (round-to-nearest) into three bf16 terms hi + mid + lo = x exactly;
// six of the nine cross products are accumulated in fp32 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo,
// lo*hi): the three dropped ones are below 2^-26 of the product, under fp32's own rounding.
// v_mfma_f32_32x32x16_bf16 runs at 16x the flops of v_mfma_f32_32x32x2_f32, so six of them per
// k16 cost 3/8 of the fp32 MFMA time; the split costs 5.5 VALU ops per loaded element.
// Build: hipcc -O3 --offload-arch=gfx950 scratch/bf16x3_gemm.hip -o scratch/bf16x3_gemm
// Run:   scratch/bf16x3_gemm      (max |err| against fp64 for this kernel and for a plain fp32
//                                  dot product, and the time per launch)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 64, BN = 128, BK = 16, NT = 256;
constexpr int ROWB = 48;                         // bytes per LDS row: 16 bf16 + 16 bytes of padding
constexpr int PLANE = (BM + BN) * ROWB;          // one plane (hi / mid / lo) of A and B rows
constexpr int STAGE = 3 * PLANE;                 // 27,648 bytes

__device__ inline unsigned cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// two floats -> packed bf16 pairs of their hi / mid / lo terms
__device__ inline void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk(r0, r1);
  const float q0 = r0 - __uint_as_float(mid << 16), q1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk(q0, q1);
}

__global__ void __launch_bounds__(NT, 2)
gemm_bf16x3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N,
            int K) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;
  const long m0 = (long)blockIdx.x * BM, n0 = (long)blockIdx.y * BN;

  // loader: unit u = 4 consecutive k of one row; rows 0..63 = A, 64..191 = B; 4 units per row
  const float* src[3];
  int dsto[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int u = threadIdx.x + p * NT;          // 0 .. 767
    const int r = u >> 2, c = (u & 3) * 4;
    src[p] = (r < BM ? A + (m0 + r) * (long)K : B + (n0 + r - BM) * (long)K) + c;
    dsto[p] = r * ROWB + c * 2;
  }
  float4 v[3];
  auto issue = [&](int step) {
#pragma unroll
    for (int p = 0; p < 3; ++p) v[p] = *reinterpret_cast<const float4*>(src[p] + step * BK);
  };
  auto store = [&](int stage) {
    unsigned char* st = lds + stage * STAGE;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      unsigned h0, m0_, l0, h1, m1, l1;
      split2(v[p].x, v[p].y, h0, m0_, l0);
      split2(v[p].z, v[p].w, h1, m1, l1);
      *reinterpret_cast<u32x2*>(st + dsto[p]) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + PLANE + dsto[p]) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(st + 2 * PLANE + dsto[p]) = u32x2{l0, l1};
    }
  };

  floatx16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const int offa = (wm0 + l31) * ROWB + half * 16;
  const int offb0 = (BM + wn0 + l31) * ROWB + half * 16, offb1 = offb0 + 32 * ROWB;
  auto frag = [&](const unsigned char* p) -> bf16x8 {
    const u32x4 w = *reinterpret_cast<const u32x4*>(p);
    return __builtin_bit_cast(bf16x8, w);
  };

  const int nsteps = K / BK;
  issue(0);
  store(0);
  __syncthreads();
  for (int t = 0; t < nsteps; ++t) {
    if (t + 1 < nsteps) issue(t + 1);
    const unsigned char* st = lds + (t & 1) * STAGE;
    bf16x8 a[3], b0[3], b1[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a[p] = frag(st + p * PLANE + offa);
      b0[p] = frag(st + p * PLANE + offb0);
      b1[p] = frag(st + p * PLANE + offb1);
    }
    // smallest terms first
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], b0[PB[q]], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], b1[PB[q]], acc[1], 0, 0, 0);
    }
    if (t + 1 < nsteps) store((t + 1) & 1);
    __syncthreads();
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
  const int M = 13312, N = 256, K = 1216;      // multiples of the tile: no bounds handling in the loader
  std::vector<float> a((size_t)M * K), b((size_t)N * K);
  srand(1);
  for (auto& v : a) v = (rand() % 200001 - 100000) / 100000.f * ((rand() & 7) == 0 ? 37.f : 1.f);
  for (auto& v : b) v = (rand() % 200001 - 100000) / 1000000.f;
  float *da, *db, *dc;
  CK(hipMalloc(&da, a.size() * 4));
  CK(hipMalloc(&db, b.size() * 4));
  CK(hipMalloc(&dc, (size_t)M * N * 4));
  CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  dim3 grid(M / BM, N / BN);
  gemm_bf16x3<<<grid, NT>>>(da, db, dc, M, N, K);
  CK(hipDeviceSynchronize());
  std::vector<float> c((size_t)M * N);
  CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
  double err_k = 0, err_f = 0, scale = 0;
  for (int s = 0; s < 4000; ++s) {
    const int r = rand() % M, col = rand() % N;
    double ref = 0, mag = 0;
    float f32 = 0.f;
    for (int k = 0; k < K; ++k) {
      const double p = (double)a[(size_t)r * K + k] * b[(size_t)col * K + k];
      ref += p;
      mag += fabs(p);
      f32 = fmaf(a[(size_t)r * K + k], b[(size_t)col * K + k], f32);
    }
    err_k = fmax(err_k, fabs(ref - c[(size_t)r * N + col]) / mag);
    err_f = fmax(err_f, fabs(ref - f32) / mag);
    scale = fmax(scale, mag);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) gemm_bf16x3<<<grid, NT>>>(da, db, dc, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("max err / sum|a*b|: bf16x3 kernel %.3g, sequential fp32 fma %.3g (2^-24 = %.3g)\n", err_k, err_f,
         ldexp(1.0, -24));
  printf("%.1f us   %.1f fp32-equivalent TF/s   (%d tiles)\n", us, 2.0 * M * N * K / us * 1e-6, grid.x * grid.y);
  return err_k < 1e-6 ? 0 : 2;
}
