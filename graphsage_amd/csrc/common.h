// Shared host/device helpers of libsage_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sage_hip.h"

namespace sage {

void set_error(const char* fmt, ...);

// Measurement hook (sage_probe_next_contraction): HIP event pairs that coming contraction launches
// of this thread carry (hipExtLaunchKernelGGL start / stop events: the dispatch's own duration,
// not the time between two markers).  take_probe() is called once per contraction launch.
bool take_probe(hipEvent_t* start, hipEvent_t* stop);

// kernel<<<grid, block, 0, stream>>>(args...) — or the same launch carrying the pending probe events
#define SAGE_LAUNCH_PROBED(kernel, grid, block, stream, ...)                                        \
  do {                                                                                              \
    hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                                      \
    if (sage::take_probe(&pe0_, &pe1_)) {                                                           \
      hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, pe0_, pe1_, 0, __VA_ARGS__); \
    } else {                                                                                        \
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__);                  \
    }                                                                                               \
  } while (0)

#define SAGE_REQUIRE(cond, code, ...) \
  do {                                \
    if (!(cond)) {                    \
      sage::set_error(__VA_ARGS__);   \
      return (code);                  \
    }                                 \
  } while (0)

// Call after a kernel launch: reports launch-configuration errors without synchronising.
#define SAGE_CHECK_LAUNCH(what)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      sage::set_error("%s: launch failed: %s", what, hipGetErrorString(e_)); \
      return SAGE_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

#define SAGE_CHECK_HIP(expr, what)                                    \
  do {                                                                \
    hipError_t e_ = (expr);                                           \
    if (e_ != hipSuccess) {                                           \
      sage::set_error("%s: %s", what, hipGetErrorString(e_));         \
      return SAGE_ELAUNCH;                                            \
    }                                                                 \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Carves aligned sub-buffers out of the caller's workspace.
struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
  size_t used() const { return align_up(off, 256); }
};

constexpr int kWave = 64;

// ---- Philox4x32-10 (Salmon et al., SC'11), the counter-based generator of the sampler.
struct Philox4 {
  uint32_t v[4];
};

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                 uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = static_cast<uint64_t>(M0) * c0;
    uint64_t p1 = static_cast<uint64_t>(M1) * c2;
    uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
    uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n1 = lo1;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    uint32_t n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 out;
  out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
  return out;
}

// Word i (0..3) of a Philox block without a runtime-indexed array (which would go to scratch).
__host__ __device__ inline uint32_t philox_word(const Philox4& p, int i) {
  const uint32_t lo = (i & 1) ? p.v[1] : p.v[0];
  const uint32_t hi = (i & 1) ? p.v[3] : p.v[2];
  return (i & 2) ? hi : lo;
}

// Unbiased-enough range reduction shared with the oracle: floor(u * n / 2^32).
__host__ __device__ inline uint32_t mulhi_range(uint32_t u, uint32_t n) {
  return static_cast<uint32_t>((static_cast<uint64_t>(u) * n) >> 32);
}

// Reads an optional device-side count (see sage_hip.h, `*_dev`).
__device__ inline int64_t dev_count(const int64_t* dev, int64_t host) {
  if (dev == nullptr) return host;
  int64_t v = *dev;
  return v < host ? v : host;
}

}  // namespace sage
