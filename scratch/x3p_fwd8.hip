// Experiment: forward contraction on PRE-SPLIT bf16x3 rows (csrc/common.h: 192-byte blocks of
// 32 k = [hi | mid | lo]), whole 64 x 256 tiles, 8 waves (2 x 4, wave tile 32 x 64), the K-step's
// tile (320 rows x 192 B = 60 KB) brought in by LDS-DMA (global_load_lds_dwordx4) into two LDS
// stages, fragments by inline-asm ds_read_b128 with counted waits, one barrier per K-step.
// C[M][N] = A[M][K] * B[N][K]^T, six of the nine cross products.
// Build: hipcc -O3 --offload-arch=gfx950 scratch/x3p_fwd8.hip -o scratch/x3p_fwd8
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifndef SRCMODE
#define SRCMODE 0
#endif
constexpr int BM = 64, BN = 256, NT = 512;
constexpr int ROWS = BM + BN;          // 320 rows per stage
constexpr int PLANE = ROWS * 64;       // one term of all rows: 20 chunks of 16 rows x 64 B
constexpr int STAGE = 3 * PLANE;       // 61,440 B
constexpr int NCHUNK = STAGE / 1024;   // 60
constexpr int CPW = (NCHUNK + 7) / 8;  // chunks per wave (the last one only on waves 0..3)

__device__ inline unsigned cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk(r0, r1);
  const float q0 = r0 - __uint_as_float(mid << 16), q1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk(q0, q1);
}
// fp32 [rows][K] (K a multiple of 32) -> x3 rows
__global__ void to_x3(const float* __restrict__ x, long rows, int K, unsigned char* __restrict__ out) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int quads = K / 4;
  const long r = q / quads;
  if (r >= rows) return;
  const int f = (int)(q - r * quads) * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + r * K + f);
  unsigned h0, m0, l0, h1, m1, l1;
  split2(v.x, v.y, h0, m0, l0);
  split2(v.z, v.w, h1, m1, l1);
  unsigned char* p = out + r * (long)(K / 32 * 192) + (f >> 5) * 192 + (f & 31) * 2;
  *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
  *reinterpret_cast<u32x2*>(p + 64) = u32x2{m0, m1};
  *reinterpret_cast<u32x2*>(p + 128) = u32x2{l0, l1};
}

template <int OFF>
__device__ inline bf16x8 rd(unsigned addr) {
  u32x4 v;
#ifdef KO_READS
  v = u32x4{addr, addr + OFF, 0x3f803f80u, 0x3f803f80u};
  asm volatile("" : "+v"(v));
  return __builtin_bit_cast(bf16x8, v);
#endif
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return __builtin_bit_cast(bf16x8, v);
}

__global__ void __launch_bounds__(NT)
x3p_fwd8(const unsigned char* __restrict__ A3, const unsigned char* __restrict__ B3, float* __restrict__ C,
         int M, int N, int K) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, kg = lane >> 4;
  const long m0 = (long)blockIdx.x * BM;
  const long ld3 = (long)(K / 32) * 192;
  const int S = K / 32;

  // LDS-DMA chunks of this wave: chunk c = wave + 8 i covers 16 rows x 64 B of one plane; lane l
  // writes physical slot l & 3 of row l >> 2 and therefore fetches logical slot (l & 3) ^ f(row)
  const unsigned char* src[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = wave + 8 * i;
    const int p = c / 20, rb = c % 20;
    const int row = rb * 16 + (lane >> 2);
    const int slot = ((lane & 3) ^ (-((lane >> 2) >> 2))) & 3;
    long grow = row < BM ? m0 + row : row - BM;
    if (row < BM && grow >= M) grow = M - 1;
    src[i] = (row < BM ? A3 : B3) + grow * ld3 + p * 64 + slot * 16;
#if SRCMODE == 1   // timing only: 8 rows x 128 contiguous bytes per instruction
    {
      const int row8 = rb * 16 + (p & 1) * 8 + (lane >> 3);
      long g8 = row8 < BM ? m0 + row8 : row8 - BM;
      if (row8 < BM && g8 >= M) g8 = M - 1;
      src[i] = (row8 < BM ? A3 : B3) + g8 * ld3 + (lane & 7) * 16;
    }
#endif
  }
#ifdef SKEW
  const int skew = (blockIdx.x * SKEW) % S;
#else
  const int skew = 0;
#endif
  auto issue = [&](int step_, int stage) {
    int step = step_ + skew;
    if (step >= S) step -= S;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = wave + 8 * i;
#if SRCMODE == 2
      if (c < NCHUNK && c % 20 < 4)
#elif SRCMODE == 3
      if (c < NCHUNK && c % 20 >= 4)
#else
      if (c < NCHUNK)   // wave-uniform
#endif
        __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (long)step * 192), (lptr_t)(lds + stage * STAGE + c * 1024),
                                         16, 0, 0);
    }
  };

  floatx4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;
  const unsigned lane_off = r16 * 64 + (((kg ^ (-(r16 >> 2))) & 3) << 4);
  const unsigned a_off = lds0 + lane_off + wr * 32 * 64;
  const unsigned b_off = lds0 + lane_off + (BM + wc * 64) * 64;

#ifdef KO_MFMA
#define MF(ai, pa, bj, pb) asm volatile("" : "+v"(acc[ai][bj]) : "v"(a[ai][pa]), "v"(b[bj][pb]))
#else
#define MF(ai, pa, bj, pb) acc[ai][bj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ai][pa], b[bj][pb], acc[ai][bj], 0, 0, 0)
#endif
#define PROD(pa, pb)                                              \
  do {                                                            \
    MF(0, pa, 0, pb); MF(1, pa, 0, pb); MF(0, pa, 1, pb); MF(1, pa, 1, pb); \
    MF(0, pa, 2, pb); MF(1, pa, 2, pb); MF(0, pa, 3, pb); MF(1, pa, 3, pb); \
  } while (0)

  auto step = [&](auto stage_, int t) {
    constexpr int ST = decltype(stage_)::value;
    const unsigned ab = a_off + ST * STAGE, bb = b_off + ST * STAGE;
    bf16x8 a[2][3], b[4][3];
    // G1: lo of A, hi of B
    a[0][2] = rd<2 * PLANE>(ab); a[1][2] = rd<2 * PLANE + 1024>(ab);
    b[0][0] = rd<0>(bb); b[1][0] = rd<1024>(bb); b[2][0] = rd<2048>(bb); b[3][0] = rd<3072>(bb);
    // G2: hi of A, lo of B
    a[0][0] = rd<0>(ab); a[1][0] = rd<1024>(ab);
    b[0][2] = rd<2 * PLANE>(bb); b[1][2] = rd<2 * PLANE + 1024>(bb); b[2][2] = rd<2 * PLANE + 2048>(bb); b[3][2] = rd<2 * PLANE + 3072>(bb);
    __builtin_amdgcn_sched_barrier(0);
#ifndef KO_GLDS
    if (t + 1 < S) issue(t + 1, ST ^ 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(6)"
                 : "+v"(a[0][2]), "+v"(a[1][2]), "+v"(b[0][0]), "+v"(b[1][0]), "+v"(b[2][0]), "+v"(b[3][0]));
    __builtin_amdgcn_sched_barrier(0);
    PROD(2, 0);
    __builtin_amdgcn_sched_barrier(0);
    // G3: mid of both
    a[0][1] = rd<PLANE>(ab); a[1][1] = rd<PLANE + 1024>(ab);
    b[0][1] = rd<PLANE>(bb); b[1][1] = rd<PLANE + 1024>(bb); b[2][1] = rd<PLANE + 2048>(bb); b[3][1] = rd<PLANE + 3072>(bb);
    asm volatile("s_waitcnt lgkmcnt(6)"
                 : "+v"(a[0][0]), "+v"(a[1][0]), "+v"(b[0][2]), "+v"(b[1][2]), "+v"(b[2][2]), "+v"(b[3][2]));
    __builtin_amdgcn_sched_barrier(0);
    PROD(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][1]), "+v"(a[1][1]), "+v"(b[0][1]), "+v"(b[1][1]), "+v"(b[2][1]), "+v"(b[3][1]));
    __builtin_amdgcn_sched_barrier(0);
    PROD(1, 1);
    PROD(1, 0);
    PROD(0, 1);
    PROD(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int t = 0;
  for (; t + 1 < S; t += 2) {
    step(S0{}, t);
    step(S1{}, t + 1);
  }
  if (t < S) step(S0{}, t);

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long col = wc * 64 + j * 16 + r16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long row = m0 + wr * 32 + i * 16 + kg * 4 + e;
        if (row < M) C[row * N + col] = acc[i][j][e];
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
  const int M = getenv("MROWS") ? atoi(getenv("MROWS")) : 13100, N = 256, K = 1216;
  std::vector<float> a((size_t)M * K), b((size_t)N * K);
  srand(1);
  for (auto& v : a) v = (rand() % 200001 - 100000) / 100000.f * ((rand() & 7) == 0 ? 37.f : 1.f);
  for (auto& v : b) v = (rand() % 200001 - 100000) / 1000000.f;
  float *da, *db, *dc;
  unsigned char *da3, *db3;
  const size_t ld3 = (size_t)K / 32 * 192;
  CK(hipMalloc(&da, a.size() * 4));
  CK(hipMalloc(&db, b.size() * 4));
  CK(hipMalloc(&da3, (size_t)M * ld3));
  CK(hipMalloc(&db3, (size_t)N * ld3));
  CK(hipMalloc(&dc, (size_t)M * N * 4));
  CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  to_x3<<<(unsigned)(((size_t)M * K / 4 + 255) / 256), 256>>>(da, M, K, da3);
  to_x3<<<(unsigned)(((size_t)N * K / 4 + 255) / 256), 256>>>(db, N, K, db3);
  CK(hipDeviceSynchronize());
  const int lds_bytes = 2 * STAGE;
  CK(hipFuncSetAttribute((const void*)x3p_fwd8, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  dim3 grid((M + BM - 1) / BM);
  CK(hipMemset(dc, 0, (size_t)M * N * 4));
  x3p_fwd8<<<grid, NT, lds_bytes>>>(da3, db3, dc, M, N, K);
  CK(hipDeviceSynchronize());
  std::vector<float> c((size_t)M * N);
  CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
  double err_k = 0, err_f = 0;
  for (int s = 0; s < 4000; ++s) {
    const int r = s < 64 ? M - 1 - s : rand() % M, col = rand() % N;
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
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 20;
  for (int i = 0; i < 3; ++i) x3p_fwd8<<<grid, NT, lds_bytes>>>(da3, db3, dc, M, N, K);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) x3p_fwd8<<<grid, NT, lds_bytes>>>(da3, db3, dc, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("max err / sum|a*b|: x3p_fwd8 %.3g, sequential fp32 fma %.3g (2^-24 = %.3g)\n", err_k, err_f, ldexp(1.0, -24));
  printf("M %d: %d tiles of %d x %d: %.1f us   %.1f fp32-equivalent TF/s\n", M, grid.x, BM, BN, us,
         2.0 * M * N * K / us * 1e-6);
  return err_k < 1e-5 ? 0 : 2;
}
