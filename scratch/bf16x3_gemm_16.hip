// Variant of scratch/bf16x3_gemm.hip on v_mfma_f32_16x16x32_bf16 (BK = 32, one LDS stage, unpadded
// 64-byte LDS rows with the 16-byte slots XOR-swizzled so that the 16x16x32 fragment reads are
// conflict-free).
// Experiment: fp32-accurate GEMM C[M][N] = A[M][K] * B[N][K]^T on the bf16 MFMA.
// Every fp32 operand is split (round-to-nearest) into three bf16 terms hi + mid + lo = x exactly;
// six of the nine cross products are accumulated in fp32 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo,
// lo*hi): the three dropped ones are at most 2^-24 of the product (2^-28 on average): fp32's own rounding.
// v_mfma_f32_32x32x16_bf16 runs at 16x the flops of v_mfma_f32_32x32x2_f32, so six of them per
// k16 cost 3/8 of the fp32 MFMA time; the split costs 5.5 VALU ops per loaded element.
// Build: hipcc -O3 --offload-arch=gfx950 scratch/bf16x3_gemm.hip -o scratch/bf16x3_gemm
// Run:   scratch/bf16x3_gemm      (max |err| against fp64 for this kernel and for a plain fp32
//                                  dot product, and the time per launch)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef BKV
#define BKV 16
#endif
#ifndef NSTAGE
#define NSTAGE 2
#endif
#ifndef WGS
#define WGS 2
#endif
#ifndef BMV
#define BMV 64
#endif
constexpr int BM = BMV, BN = 128, BK = BKV, NT = 256;
constexpr int TM = BM / 64;
constexpr int TI = BM / 32;                        // 16-row blocks per wave tile (wave tile = BM/2 x 64)
#ifndef NPROD
#define NPROD 6      // knock-out: fewer of the six products (wrong numbers, timing only)
#endif
constexpr int ROWB = BK * 2;                     // 64-byte rows, no padding: slot' = slot ^ ((-(row >> 2)) & 3)
constexpr int PLANE = (BM + BN) * ROWB;          // one plane (hi / mid / lo) of A and B rows
constexpr int STAGE = 3 * PLANE;
constexpr int UPR = BK / 4;                      // float4 units per row
constexpr int PASSES = (BM + BN) * UPR / NT;

__device__ inline unsigned cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// two floats -> packed bf16 pairs of their hi / mid / lo terms
__device__ inline void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
#ifdef KO_SPLIT   // knock-out: truncated hi only, no residuals (wrong numbers, timing only)
  hi = (__float_as_uint(x0) >> 16) | (__float_as_uint(x1) & 0xffff0000u);
  mid = hi;
  lo = hi;
  return;
#endif
  hi = cvt_pk(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk(r0, r1);
  const float q0 = r0 - __uint_as_float(mid << 16), q1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk(q0, q1);
}

// stream-K: the grid is the number of resident workgroups; each takes an equal range of
// (tile, K-step) pairs and adds its partial tiles into the zeroed C
__global__ void __launch_bounds__(NT, WGS)
gemm_bf16x3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N,
            int K) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * 64;
  const int S = K / BK, tiles_n = N / BN;
  const long total = (long)(M / BM) * tiles_n * S;
  unsigned lb = blockIdx.x;
  if (gridDim.x % 8 == 0) lb = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
  long it = total * lb / gridDim.x;
  const long end = total * (lb + 1) / gridDim.x;

  const int r16 = lane & 15, kg = lane >> 4;
  const int sw = ((kg ^ (-(r16 >> 2))) & 3) << 4;            // (row & 15) >> 2 == r16 >> 2 for every 16-row block
  const int offa = (wm0 + r16) * ROWB + sw;                  // + 16 i rows
  const int offb = (BM + wn0 + r16) * ROWB + sw;             // + 16 j rows
  auto frag = [&](const unsigned char* p) -> bf16x8 {
    const u32x4 w = *reinterpret_cast<const u32x4*>(p);
    return __builtin_bit_cast(bf16x8, w);
  };

  while (it < end) {
    const long tile = it / S;
    const int u0 = (int)(it - tile * S);
    const int u1 = end - it < S - u0 ? u0 + (int)(end - it) : S;
    const long m0 = tile / tiles_n * BM, n0 = tile % tiles_n * BN;
    it += u1 - u0;

    const float* src[PASSES];
    int dsto[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int u = threadIdx.x + p * NT;
      const int r = u / UPR, c = (u % UPR) * 4;
      src[p] = (r < BM ? A + (m0 + r) * (long)K : B + (n0 + r - BM) * (long)K) + c;
      dsto[p] = r * ROWB + ((((c >> 3) ^ (-((r & 15) >> 2))) & 3) << 4) + ((c >> 2) & 1) * 8;
    }
    // loads run TWO steps ahead in two register sets (set = parity of the step they hold);
    // requests past the last step are clamped to it and their LDS image is never read
    float4 v[2][PASSES];
    auto issue = [&](auto set, int step) {
      const int sc = step < u1 ? step : u1 - 1;
#ifdef KO_LOADS   // knock-out: no global loads in the loop (wrong numbers, timing only)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) v[decltype(set)::value][p] = make_float4(1.f + sc, 2.f, 3.f, (float)threadIdx.x);
      return;
#endif
#pragma unroll
      for (int p = 0; p < PASSES; ++p) v[decltype(set)::value][p] = *reinterpret_cast<const float4*>(src[p] + sc * BK);
    };
    auto store = [&](auto set, int stage) {
      unsigned char* st = lds + stage * STAGE;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const float4 x = v[decltype(set)::value][p];
        unsigned h0, m0_, l0, h1, m1, l1;
        split2(x.x, x.y, h0, m0_, l0);
        split2(x.z, x.w, h1, m1, l1);
        *reinterpret_cast<u32x2*>(st + dsto[p]) = u32x2{h0, h1};
#ifndef KO_LDSW   // knock-out: one plane written instead of three
        *reinterpret_cast<u32x2*>(st + PLANE + dsto[p]) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(st + 2 * PLANE + dsto[p]) = u32x2{l0, l1};
#endif
      }
    };

    floatx4 acc[TI][4];                                      // (BM/2) x 64 wave tile as TI x 4 blocks of 16 x 16
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    auto mfmas = [&](const unsigned char* st) {
      bf16x8 a[TI][3], b[4][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TI; ++i) a[i][p] = frag(st + p * PLANE + offa + i * 16 * ROWB);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j][p] = frag(st + p * PLANE + offb + j * 16 * ROWB);
      }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
      for (int q = 0; q < NPROD; ++q)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // one step: LDS stage `cur` holds step t, set HOLD holds step t+1, set FREE takes step t+2
    auto step = [&](auto hold, auto free_, int t, int cur) {
      issue(free_, t + 2);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(lds + (NSTAGE == 2 ? cur : 0) * STAGE);
      __builtin_amdgcn_sched_barrier(0);
#ifndef KO_BARRIER   // knock-out: no barriers in the loop (races; timing only)
      if (NSTAGE == 1) __syncthreads();
#endif
      store(hold, NSTAGE == 2 ? cur ^ 1 : 0);
#ifndef KO_BARRIER
      __syncthreads();
#endif
    };
    issue(S0{}, u0);
    issue(S1{}, u0 + 1);
    __builtin_amdgcn_sched_barrier(0);
    store(S0{}, 0);
    __syncthreads();
    int t = u0;
    for (; t + 1 < u1; t += 2) {
      step(S1{}, S0{}, t, 0);
      step(S0{}, S1{}, t + 1, 1);
    }
    if (t < u1) step(S1{}, S0{}, t, 0);
    // C/D of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long col = n0 + wn0 + j * 16 + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const long row = m0 + wm0 + i * 16 + kg * 4 + e;
          atomicAdd(C + row * N + col, acc[i][j][e]);
        }
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
#ifndef MV
#define MV 13312
#define NV 256
#define KV 1216
#endif
  const int M = MV, N = NV, K = KV;      // multiples of the tile: no bounds handling in the loader
                                         // (-DMV=81408 -DNV=640 -DKV=608: the pool Linear's shape)
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
  dim3 grid(256 * WGS);
  CK(hipMemset(dc, 0, (size_t)M * N * 4));
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
  const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 20;   // REPS=30000: seconds of load, to watch clocks / power beside it
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) gemm_bf16x3<<<grid, NT>>>(da, db, dc, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("max err / sum|a*b|: bf16x3 kernel %.3g, sequential fp32 fma %.3g (2^-24 = %.3g)\n", err_k, err_f,
         ldexp(1.0, -24));
  printf("BM %d BK %d stages %d wgs/cu %d: %.1f us   %.1f fp32-equivalent TF/s\n", BM, BK, NSTAGE, WGS, us,
         2.0 * M * N * K / us * 1e-6);
  return 0;
}
