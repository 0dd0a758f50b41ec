// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap?  One workgroup of 512
// threads per CU = 2 waves per SIMD.  Modes: 0 = both waves issue MFMAs (16 independent
// accumulators each); 1 = both issue VALU fmas; 2 = waves 0-3 MFMA, waves 4-7 VALU (one of each per
// SIMD); 3 = every wave alternates blocks of 16 MFMAs and 64 VALU ops; 4 / 5 = only the MFMA / only
// the VALU half of mode 2 (the other four waves exit).
// Build: hipcc -O3 --offload-arch=gfx950 scratch/mfma_valu_overlap.hip -o scratch/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int ITERS = 2000;

__device__ inline void mfma_block(floatx4 (&acc)[16], bf16x8 a, bf16x8 b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
}
__device__ inline void valu_block(float (&v)[16], float m) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, 1.0f);
}

__global__ void __launch_bounds__(512) k(int mode, float* out, float m) {
  const int wave = threadIdx.x >> 6;
  floatx4 acc[16];
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[i] = floatx4{0, 0, 0, 0}; v[i] = threadIdx.x * 0.001f + i; }
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f); b[i] = (__bf16)(i * 0.5f); }
  const bool do_mfma = mode == 0 || mode == 3 || ((mode == 2 || mode == 4) && wave < 4);
  const bool do_valu = mode == 1 || mode == 3 || ((mode == 2 || mode == 5) && wave >= 4);
  if (!do_mfma && !do_valu) return;
#ifdef PRIO      // -DPRIO=n: the MFMA waves of mode 2 raise their issue priority
  if (mode == 2 && do_mfma) __builtin_amdgcn_s_setprio(PRIO);
#endif
  for (int it = 0; it < ITERS; ++it) {
    if (do_mfma) mfma_block(acc, a, b);           // 16 MFMAs = 256 matrix-pipe cycles
    if (do_valu) valu_block(v, m);                // 64 VALU ops = 256 issue cycles (wave64 on 16 lanes)
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[6] = {"MFMA + MFMA", "VALU + VALU", "MFMA wave + VALU wave", "each wave: MFMA block, VALU block",
                          "MFMA wave alone", "VALU wave alone"};
  for (int mode = 0; mode < 6; ++mode) {
    k<<<256, 512>>>(mode, out, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<<<256, 512>>>(mode, out, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 5;
    printf("mode %d %-36s %8.1f us  = %6.1f cycles per iteration at 2.4 GHz (one block: 256)\n", mode, names[mode], us,
           us * 2400.0 / ITERS);
  }
  return 0;
}
