// The weight gradient of a layer with few outputs (n <= 64: the class logits) as a VALU body that
// more than one kernel can host: skinny_dweight_kernel (linear.hip) and the class-logit layer's
// backward launch (logit.hip), where it runs in workgroups of its own beside the seeds' waves.
#pragma once
#include "common.h"

namespace sage {

constexpr int kSkinnyThreads = 256;
constexpr int kDwRows = 8;
struct SkinnyDwProblem {
  const float* a;
  int64_t ld_a;
  const int64_t* rows;
  int k;
  float* dw;
  int64_t ld_dw;
};

// dw[n][k] += sum_m dout[m][n] * a[rows(m)][k]: thread per column k, block (bx, by, bz) per
// (256 columns, group of 8 output rows, slice of m of problem bz / slices); slices are combined
// with fp32 atomics.  256 threads.
__device__ __forceinline__ void skinny_dweight_body(int bx, int by, int bz, const float* __restrict__ dout,
                                                    int64_t ld_dout, const SkinnyDwProblem& p0,
                                                    const SkinnyDwProblem& p1, int slices, int64_t m,
                                                    const int64_t* __restrict__ m_dev, int n, int m_chunk,
                                                    float* __restrict__ dbias) {
  constexpr int kThreads = kSkinnyThreads;

  const int prob = bz >= slices ? 1 : 0;
  const SkinnyDwProblem& pr = prob ? p1 : p0;
  const float* __restrict__ a = pr.a;
  const int64_t ld_a = pr.ld_a;
  const int64_t* __restrict__ rows = pr.rows;
  const int k = pr.k;
  float* __restrict__ dw = pr.dw;
  const int64_t ld_dw = pr.ld_dw;
  if (static_cast<int>(bx) * kThreads >= k) return;
  if (prob) dbias = nullptr;
  int64_t m_lim = m;
  if (m_dev != nullptr) {
    const int64_t v = *m_dev;
    if (v < m_lim) m_lim = v;
  }
  const int64_t m_beg = static_cast<int64_t>(bz - prob * slices) * m_chunk;
  int64_t m_end = m_beg + m_chunk;
  if (m_end > m_lim) m_end = m_lim;
  if (m_beg >= m_end) return;
  const int col = bx * kThreads + threadIdx.x;
  const int n0 = by * kDwRows;
  float acc[kDwRows];
#pragma unroll
  for (int i = 0; i < kDwRows; ++i) acc[i] = 0.f;
  // The block's kDwRows columns of dout are the same for every thread: stage 64 rows of them
  // in LDS (one coalesced pass) and read them back as broadcasts, instead of 8 same-address
  // global loads per row per thread.
  constexpr int kStage = 64, U = 8;
  __shared__ __attribute__((aligned(16))) float sd[kStage][kDwRows];
  for (int64_t mm0 = m_beg; mm0 < m_end; mm0 += kStage) {
    const int cnt = m_end - mm0 < kStage ? static_cast<int>(m_end - mm0) : kStage;
    for (int q = threadIdx.x; q < kStage * kDwRows; q += kThreads) {
      const int r = q / kDwRows, i = q % kDwRows;
      sd[r][i] = (r < cnt && n0 + i < n) ? dout[(mm0 + r) * ld_dout + n0 + i] : 0.f;   // rows past cnt: 0
    }
    __syncthreads();
    for (int u0 = 0; u0 < cnt; u0 += U) {
      float av[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t mc = mm0 + u0 + u < m_end ? mm0 + u0 + u : m_end - 1;   // clamped: times 0 below
        const int64_t r = rows != nullptr ? rows[mc] : mc;
        av[u] = col < k ? a[r * ld_a + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4 d0 = *reinterpret_cast<const float4*>(&sd[u0 + u < kStage ? u0 + u : kStage - 1][0]);
        const float4 d1 = *reinterpret_cast<const float4*>(&sd[u0 + u < kStage ? u0 + u : kStage - 1][4]);
        const float w = u0 + u < cnt ? av[u] : 0.f;
        acc[0] += d0.x * w; acc[1] += d0.y * w; acc[2] += d0.z * w; acc[3] += d0.w * w;
        acc[4] += d1.x * w; acc[5] += d1.y * w; acc[6] += d1.z * w; acc[7] += d1.w * w;
      }
    }
    __syncthreads();
  }
  if (col < k) {
#pragma unroll
    for (int i = 0; i < kDwRows; ++i) {
      if (n0 + i < n) atomicAdd(dw + (n0 + i) * ld_dw + col, acc[i]);
    }
  }
  if (dbias != nullptr && bx == 0 && threadIdx.x < kDwRows && n0 + threadIdx.x < n) {
    float sb = 0.f;
    for (int64_t mm = m_beg; mm < m_end; ++mm) sb += dout[mm * ld_dout + n0 + threadIdx.x];
    atomicAdd(dbias + n0 + threadIdx.x, sb);
  }
}

}  // namespace sage
