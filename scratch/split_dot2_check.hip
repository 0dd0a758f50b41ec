// Checks the dot2-based bf16x3 split (hi + mid + lo == x exactly?) against the shift / mask / subtract one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ inline unsigned cvt_pk_bf16(float a, float b) {
#ifdef ASMCVT
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
#endif
}
__device__ inline void split_old(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(r0, r1);
  const float q0 = r0 - __uint_as_float(mid << 16), q1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk_bf16(q0, q1);
}
__device__ inline void split_new(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
#ifdef REGSEL
  unsigned s0, s1;      // selectors in registers: no inline-constant expansion in the way
  asm volatile("v_mov_b32 %0, 0xbf80" : "=v"(s0));
  asm volatile("v_mov_b32 %0, 0xbf800000" : "=v"(s1));
#else
  const unsigned s0 = 0x8000bf80u, s1 = 0xbf800000u;
#endif
  const bf16x2_t sel0 = __builtin_bit_cast(bf16x2_t, s0);
  const bf16x2_t sel1 = __builtin_bit_cast(bf16x2_t, s1);
  hi = cvt_pk_bf16(x0, x1);
  const bf16x2_t h = __builtin_bit_cast(bf16x2_t, hi);
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(h, sel0, x0, false);
  const float r1 = __builtin_amdgcn_fdot2_f32_bf16(h, sel1, x1, false);
  mid = cvt_pk_bf16(r0, r1);
  const bf16x2_t m = __builtin_bit_cast(bf16x2_t, mid);
  const float q0 = __builtin_amdgcn_fdot2_f32_bf16(m, sel0, r0, false);
  const float q1 = __builtin_amdgcn_fdot2_f32_bf16(m, sel1, r1, false);
  lo = cvt_pk_bf16(q0, q1);
}
__global__ void k(const float* x, unsigned* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  unsigned a, b, c, d, e, f;
  split_old(x[2 * i], x[2 * i + 1], a, b, c);
  split_new(x[2 * i], x[2 * i + 1], d, e, f);
  o[6 * i] = a; o[6 * i + 1] = b; o[6 * i + 2] = c; o[6 * i + 3] = d; o[6 * i + 4] = e; o[6 * i + 5] = f;
}
int main() {
  const int n = 1 << 16;
  std::vector<float> x(n);
  srand(3);
  for (int i = 0; i < n; ++i) {
    const int m = rand() % 8;
    x[i] = m == 0 ? 0.f : (rand() % 200001 - 100000) / 100000.f * (m == 1 ? 1e-3f : m == 2 ? 37.f : m == 3 ? 1e6f : 1.f);
  }
  float* dx; unsigned* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 3 * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(dx, dout, n);
  std::vector<unsigned> o(n * 3);
  hipMemcpy(o.data(), dout, n * 12, hipMemcpyDeviceToHost);
  int diff = 0;
  for (int i = 0; i < n / 2; ++i)
    for (int p = 0; p < 3; ++p)
      if (o[6 * i + p] != o[6 * i + 3 + p]) {
        if (diff < 8) printf("pair %d plane %d: old %08x new %08x  x = %g %g\n", i, p, o[6 * i + p], o[6 * i + 3 + p], x[2 * i], x[2 * i + 1]);
        ++diff;
      }
  printf("%d of %d plane words differ\n", diff, n / 2 * 3);
  return 0;
}
