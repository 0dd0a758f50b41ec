// Bare fp32 MFMA loops on random operands: which shape holds the higher clock under load?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void __launch_bounds__(256) k(const float* in, float* out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(t * 8 + i) & 0xFFFFF]; b[i] = in[(t * 8 + i + 4096) & 0xFFFFF]; }
  if (SHAPE == 32) {
    f16v c0 = {}, c1 = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[7 - i], c1, 0, 0, 0);
      }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[t] = s;
  } else {
    f4v c[8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // 4 x (16x16x4) = the flops of 2 x (32x32x2)
        c[(2 * i) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c[(2 * i) & 7], 0, 0, 0);
        c[(2 * i + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[7 - i], c[(2 * i + 1) & 7], 0, 0, 0);
        c[(2 * i + 4) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[7 - i], b[i], c[(2 * i + 4) & 7], 0, 0, 0);
        c[(2 * i + 5) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[7 - i], b[7 - i], c[(2 * i + 5) & 7], 0, 0, 0);
      }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[t] = s;
  }
}

int main() {
  const int blocks = 256 * 2, iters = 4000;
  std::vector<float> h(1 << 20);
  srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
  float *in, *out;
  hipMalloc(&in, h.size() * 4); hipMalloc(&out, blocks * 256 * 4);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape : {32, 16, 32, 16}) {
    for (int wpb : {1, 2}) {
      const int nb = 256 * wpb;   // 4 waves per block -> wpb waves per SIMD
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) {
          if (shape == 32) k<32><<<nb, 256>>>(in, out, iters); else k<16><<<nb, 256>>>(in, out, iters);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 20.0 * nb * 4 * (double)iters * 16 * 4096.0;
        if (rep == 2) printf("shape %dx%d  waves/SIMD %d  %.1f TF/s  (%.3f ms per launch)\n", shape, shape, wpb, flops / (ms * 1e-3) / 1e12, ms / 20);
      }
    }
  }
  return 0;
}
