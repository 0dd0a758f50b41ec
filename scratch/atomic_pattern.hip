// Micro-benchmark: fp32 atomic adds of split-K partial tiles (500 workgroups x 128 x 128 floats onto 20
// tiles, 25 workgroups per tile, all at once) with different per-instruction footprints:
//   PATTERN 0: the 16x16 MFMA C layout (lane -> col = lane & 15, row = 4 (lane >> 4) + e): 4 rows x 64 B
//   PATTERN 1: one row x 256 B per instruction (lane -> col = lane)
//   PATTERN 2: two rows x 128 B per instruction
//   PATTERN 3: plain stores in layout 0 (no atomics)
// Build: hipcc -O3 --offload-arch=gfx950 -DPATTERN=0 scratch/atomic_pattern.hip -o /tmp/ab/p0
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef PATTERN
#define PATTERN 0
#endif
constexpr int LD = 1216, TILE = 128;
__global__ void __launch_bounds__(256) k(float* out, int tiles_x) {
#ifndef NTILES
#define NTILES 20
#endif
  const int tile = blockIdx.x % NTILES, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* base = out + (tile / tiles_x) * TILE * LD + (tile % tiles_x) * TILE;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const float v = 1.0f + lane;
#pragma unroll 4
  for (int b = 0; b < 64; ++b) {      // 64 instructions per wave, 64 lanes each: the wave's 64 x 64 quadrant
    int row, col;
    if (PATTERN == 0 || PATTERN == 3) { const int i = b >> 4, j = (b >> 2) & 3, e = b & 3; row = wm0 + i * 16 + 4 * (lane >> 4) + e; col = wn0 + j * 16 + (lane & 15); }
    else if (PATTERN == 1) { row = wm0 + b; col = wn0 + lane; }
    else { row = wm0 + 2 * (b >> 1) + (lane >> 5); col = wn0 + (b & 1) * 32 + (lane & 31); }
    if (PATTERN == 3) base[row * LD + col] = v; else atomicAdd(base + row * LD + col, v);
  }
}
int main() {
  float* d; hipMalloc(&d, (size_t)(NTILES / 9 + 2) * 128 * LD * 4); hipMemset(d, 0, (size_t)(NTILES / 9 + 2) * 128 * LD * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) k<<<500, 256>>>(d, 9);
  hipEventRecord(e0); for (int i = 0; i < 50; ++i) k<<<500, 256>>>(d, 9); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("pattern %d, %d tiles: %.2f us per launch (500 workgroups x 64 KB)\n", PATTERN, NTILES, ms * 20);
  return 0;
}
