// LDS-fed fp32 MFMA loops (no global traffic in the loop): which fragment-read pattern keeps the
// matrix pipe busy at 1-2 waves per SIMD?  Tile 64x128 per 4-wave block, BK = 32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int BK = 32, BM = 64, BN = 128;

// V=1: k-major As[k][m], ds_read_b32 per operand (current kernel)
// V=2: m-major As[m][k] (row stride 36), ds_read_b128: lane half h takes k = 8g + 4h .. +3
// V=3: k-major, A ds_read_b32, B one ds_read_b64 per k: the wave's two column tiles interleaved (n = 2c + j)
template <int V, bool BARRIER>
__global__ void __launch_bounds__(256, 2) k(const float* in, float* out, int ksteps) {
  constexpr int LDA = V == 2 ? BK + 4 : BM + (V == 1 ? 2 : 4), LDB = V == 2 ? BK + 4 : BN + (V == 1 ? 2 : 4);
  constexpr int ASZ = V == 2 ? BM * LDA : BK * LDA, BSZ = V == 2 ? BN * LDB : BK * LDB;
  __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];
  for (int i = threadIdx.x; i < 2 * (ASZ + BSZ); i += 256) lds[i] = in[(blockIdx.x * 977 + i) & 0xFFFFF];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64, l31 = lane & 31, half = lane >> 5;
  f16v c0 = {}, c1 = {};
  int cur = 0;
  for (int s = 0; s < ksteps; ++s) {
    const float* as = lds + cur * ASZ;
    const float* bs = lds + 2 * ASZ + cur * BSZ;
    if (V == 1) {
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const float a = as[(2 * p + half) * LDA + wm0 + l31];
        const float b0 = bs[(2 * p + half) * LDB + wn0 + l31];
        const float b1 = bs[(2 * p + half) * LDB + wn0 + 32 + l31];
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, c1, 0, 0, 0);
      }
    } else if (V == 3) {
      typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const float a = as[(2 * p + half) * LDA + wm0 + l31];
        const f2v b = *reinterpret_cast<const f2v*>(bs + (2 * p + half) * LDB + wn0 + 2 * l31);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[1], c1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f4v a = *reinterpret_cast<const f4v*>(as + (wm0 + l31) * LDA + 8 * g + 4 * half);
        const f4v b0 = *reinterpret_cast<const f4v*>(bs + (wn0 + l31) * LDB + 8 * g + 4 * half);
        const f4v b1 = *reinterpret_cast<const f4v*>(bs + (wn0 + 32 + l31) * LDB + 8 * g + 4 * half);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[j], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[j], c1, 0, 0, 0);
        }
      }
    }
    if (BARRIER) __syncthreads();
    cur ^= 1;
  }
  float sum = 0; for (int i = 0; i < 16; ++i) sum += c0[i] + c1[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int V, bool BARRIER>
void run(const float* in, float* out, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ksteps = 2000;
  for (int nb : {256, 512}) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int r = 0; r < 10; ++r) k<V, BARRIER><<<nb, 256>>>(in, out, ksteps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = 10.0 * nb * 4 * (double)ksteps * 32 * 4096.0;
    printf("%-34s blocks/CU %d  %.1f TF/s\n", name, nb / 256, flops / (ms * 1e-3) / 1e12);
  }
}

int main() {
  std::vector<float> h(1 << 20);
  srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
  float *in, *out;
  if (hipMalloc(&in, h.size() * 4) != hipSuccess || hipMalloc(&out, 512 * 256 * 4) != hipSuccess) return 1;
  if (hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
  run<1, true>(in, out, "k-major b32 reads, barrier");
  run<1, false>(in, out, "k-major b32 reads, no barrier");
  run<3, true>(in, out, "k-major A b32 + B b64, barrier");
  run<3, false>(in, out, "k-major A b32 + B b64, no barrier");
  run<2, true>(in, out, "m-major b128 reads, barrier");
  run<2, false>(in, out, "m-major b128 reads, no barrier");
  return 0;
}
