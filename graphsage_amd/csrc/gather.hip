// (a-3) row gather and (a-5) fused gather + segment reduce, forward and backward.
//
// Replaces, without materialising the per-edge tensor x_j:
//   filter_data's index_select          reference samplers/uniform_loader.py:238-239
//   SAGE.message  x_j = x[edge_index[0]] reference layers/sage.py:136-137
//   torch_scatter.scatter(x_j, edge_index[1], reduce)  reference layers/sage.py:209-210
//
// Shape of the work (Reddit 25/10, B=512): ~13 k targets with <= 25 / <= 10 neighbours,
// rows of 602 fp32 = 2,408 B read at random from a 561 MB table.  HBM-bound, ~0.25
// flop/byte.  One 64-lane wavefront owns one target: the segment's neighbour ids are
// loaded once by the lanes (one id per lane) and broadcast with v_readlane, so every
// row's base address is wave-uniform and each lane issues 16-byte loads at
// base + 16*lane; ROWS_IN_FLIGHT rows are requested before the first is consumed.  The
// reduction over the segment happens in registers (each lane owns the same feature
// columns of every row), so the forward needs no atomics and no LDS.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <hip/hip_ext.h>

#include "common.h"

namespace sage {
namespace {

constexpr int kBlock = 256;

// used instead of hipMemsetAsync so that every node a caller captures in a HIP graph is a kernel
__global__ void zero_i64_kernel(int64_t* p, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}
__global__ void zero_i32_kernel(int* p, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kRowsInFlight = 4;

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<2> {
  using type = float2;
};
template <>
struct VecT<1> {
  using type = float;
};

template <int VEC>
__device__ inline void vec_to_arr(const typename VecT<VEC>::type& v, float* a);
template <>
__device__ inline void vec_to_arr<4>(const float4& v, float* a) {
  a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
}
template <>
__device__ inline void vec_to_arr<2>(const float2& v, float* a) {
  a[0] = v.x; a[1] = v.y;
}
template <>
__device__ inline void vec_to_arr<1>(const float& v, float* a) {
  a[0] = v;
}
template <int VEC>
__device__ inline typename VecT<VEC>::type arr_to_vec(const float* a);
template <>
__device__ inline float4 arr_to_vec<4>(const float* a) {
  return make_float4(a[0], a[1], a[2], a[3]);
}
template <>
__device__ inline float2 arr_to_vec<2>(const float* a) {
  return make_float2(a[0], a[1]);
}
template <>
__device__ inline float arr_to_vec<1>(const float* a) {
  return a[0];
}

__device__ inline int64_t readlane64(int64_t v, int lane) {
  uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), lane);
  uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32), lane);
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}

// ---------------------------------------------------------------- gather rows
// wide rows: one wave per group of kRowsInFlight rows
template <int VEC>
__global__ void __launch_bounds__(kBlock)
gather_rows_wide_kernel(const uint32_t* __restrict__ table, int64_t ld_table,
                        const int64_t* __restrict__ index, int64_t num_rows,
                        const int64_t* __restrict__ num_rows_dev, int width,
                        uint32_t* __restrict__ out, int64_t ld_out) {
  using V = typename VecT<VEC>::type;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t n = dev_count(num_rows_dev, num_rows);
  const int64_t wave_id = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  const int64_t r0 = wave_id * kRowsInFlight;
  if (r0 >= n) return;
  const int nr = (n - r0) < kRowsInFlight ? static_cast<int>(n - r0) : kRowsInFlight;
  int64_t my = 0;
  if (lane < nr) my = index[r0 + lane];
  const int nvec = (width + VEC - 1) / VEC;  // a ragged last vector stays inside the row stride
  for (int c = lane; c < nvec; c += kWave) {
    V v[kRowsInFlight];
#pragma unroll
    for (int r = 0; r < kRowsInFlight; ++r) {
      if (r < nr) {
        const int64_t rowid = readlane64(my, r);
        v[r] = reinterpret_cast<const V*>(table + rowid * ld_table)[c];
      }
    }
#pragma unroll
    for (int r = 0; r < kRowsInFlight; ++r) {
      if (r < nr) reinterpret_cast<V*>(out + (r0 + r) * ld_out)[c] = v[r];
    }
  }
}

// narrow rows (labels, ids): one thread per 32-bit word
__global__ void gather_rows_narrow_kernel(const uint32_t* __restrict__ table, int64_t ld_table,
                                          const int64_t* __restrict__ index, int64_t num_rows,
                                          const int64_t* __restrict__ num_rows_dev, int width,
                                          uint32_t* __restrict__ out, int64_t ld_out) {
  const int64_t n = dev_count(num_rows_dev, num_rows);
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t r = i / width;
  const int w = static_cast<int>(i - r * width);
  if (r >= n) return;
  out[r * ld_out + w] = table[index[r] * ld_table + w];
}

// ---------------------------------------------------------------- gather + segment reduce
// Where the max came from, per (target, column): `full` = the edge's index into src (or -1),
// int64 [T, F] dense; `compact` = the edge's offset within the target's segment as one byte, 255
// for "no gradient" (empty segment, offset above 254, or a maximum that is not positive — the
// caller's rows are relu outputs, sage_pool_max_fwd), rows ld_compact bytes apart.
struct ArgOut {
  int64_t* full;
  uint8_t* compact;
  int64_t ld_compact;
};

// One wave per (target, feature slab).  A slab is 64*VEC*NV consecutive features.
template <int VEC, int NV, int REDUCE>
__global__ void __launch_bounds__(kBlock)
gather_reduce_fwd_kernel(const float* __restrict__ x, int64_t ld_x,
                         const int64_t* __restrict__ node_map,
                         const int64_t* __restrict__ seg_ptr, const int64_t* __restrict__ src,
                         int64_t num_targets, const int64_t* __restrict__ num_targets_dev,
                         int num_features, float* __restrict__ out, int64_t ld_out,
                         ArgOut argmax) {
  using V = typename VecT<VEC>::type;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t nt = dev_count(num_targets_dev, num_targets);
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (t >= nt) return;
  const int slab0 = blockIdx.y * (kWave * VEC * NV);  // first feature of this slab
  const int64_t s0 = seg_ptr[t];
  const int64_t s1 = seg_ptr[t + 1];
  const int64_t len = s1 - s0;

  // feature column owned by this lane in chunk c: slab0 + (c*64 + lane)*VEC
  bool live[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) live[c] = slab0 + (c * kWave + lane) * VEC < num_features;

  float acc[NV][VEC];
  int arg[NV][VEC];  // max only: edge offset (relative to s0) that supplied the maximum
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      acc[c][j] = 0.f;
      arg[c][j] = -1;
    }

  for (int64_t base = 0; base < len; base += kWave) {
    const int chunk = (len - base) < kWave ? static_cast<int>(len - base) : kWave;
    int64_t my = 0;
    if (lane < chunk) {
      my = src[s0 + base + lane];
      if (node_map != nullptr) my = node_map[my];
    }
    for (int r0 = 0; r0 < chunk; r0 += kRowsInFlight) {
      V v[kRowsInFlight][NV];
#pragma unroll
      for (int r = 0; r < kRowsInFlight; ++r) {
        if (r0 + r < chunk) {
          const int64_t rowid = readlane64(my, r0 + r);
          const V* rowp = reinterpret_cast<const V*>(x + rowid * ld_x + slab0);
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            if (live[c]) v[r][c] = rowp[c * kWave + lane];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kRowsInFlight; ++r) {
        if (r0 + r < chunk) {
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            float a[VEC];
            vec_to_arr<VEC>(v[r][c], a);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              if (REDUCE == SAGE_REDUCE_MAX) {
                const int e = static_cast<int>(base) + r0 + r;
                if (arg[c][j] < 0 || a[j] > acc[c][j]) {
                  acc[c][j] = a[j];
                  arg[c][j] = e;
                }
              } else {
                acc[c][j] += a[j];
              }
            }
          }
        }
      }
    }
  }

  float cnt = len > 0 ? static_cast<float>(len) : 1.f;
  if (REDUCE == SAGE_REDUCE_GCN && len > 0) {
    // one self-loop, appended after the neighbours (reference layers/sage.py:114-117): the
    // target's own row is added last and counted
    const int64_t self = node_map != nullptr ? node_map[t] : t;
    const V* rowp = reinterpret_cast<const V*>(x + self * ld_x + slab0);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      if (live[c]) {
        float a[VEC];
        vec_to_arr<VEC>(rowp[c * kWave + lane], a);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[c][j] += a[j];
      }
    }
    cnt += 1.f;
  }
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    if (!live[c]) continue;
    const int f = slab0 + (c * kWave + lane) * VEC;
    float a[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      a[j] = (REDUCE == SAGE_REDUCE_MEAN || REDUCE == SAGE_REDUCE_GCN) ? acc[c][j] / cnt : acc[c][j];
    }
    *reinterpret_cast<V*>(out + t * ld_out + f) = arr_to_vec<VEC>(a);
    if (REDUCE == SAGE_REDUCE_MAX && argmax.full != nullptr) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (f + j < num_features) {
          argmax.full[t * static_cast<int64_t>(num_features) + f + j] = arg[c][j] < 0 ? -1 : s0 + arg[c][j];
        }
      }
    }
    if (REDUCE == SAGE_REDUCE_MAX && argmax.compact != nullptr) {
      uint8_t* q = argmax.compact + t * argmax.ld_compact + f;
      uint32_t b[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        b[j] = (arg[c][j] < 0 || arg[c][j] > 254 || !(a[j] > 0.f)) ? 255u : static_cast<uint32_t>(arg[c][j]);
      }
      if (VEC == 4 && f + 3 < num_features && (argmax.ld_compact & 3) == 0) {
        *reinterpret_cast<uint32_t*>(q) = b[0] | (b[1] << 8) | (b[2 % VEC] << 16) | (b[3 % VEC] << 24);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (f + j < num_features) q[j] = static_cast<uint8_t>(b[j]);
        }
      }
    }
  }
}

// Backward for sum/mean: one wave per target; lane owns columns lane, lane+64, ... so that
// each atomic wave-instruction adds 256 contiguous bytes of one dx row.
template <int NV>
__global__ void __launch_bounds__(kBlock)
gather_reduce_bwd_kernel(const float* __restrict__ dout, int64_t ld_dout,
                         const int64_t* __restrict__ seg_ptr, const int64_t* __restrict__ src,
                         int64_t num_targets, const int64_t* __restrict__ num_targets_dev,
                         int num_features, int mean, float* __restrict__ dx, int64_t ld_dx) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t nt = dev_count(num_targets_dev, num_targets);
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (t >= nt) return;
  const int slab0 = blockIdx.y * (kWave * NV);
  const int64_t s0 = seg_ptr[t];
  const int64_t len = seg_ptr[t + 1] - s0;
  if (len <= 0) return;
  // mean: 0 = sum, 1 = mean, 2 = gcn (mean over the neighbours and one self-loop)
  const float cnt = static_cast<float>(mean == 2 ? len + 1 : len);
  float g[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int f = slab0 + c * kWave + lane;
    g[c] = f < num_features ? dout[t * ld_dout + f] : 0.f;
    if (mean) g[c] = g[c] / cnt;
  }
  if (mean == 2) {   // the self-loop's share goes to the target's own row
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      if (slab0 + c * kWave + lane < num_features) atomicAdd(dx + t * ld_dx + slab0 + c * kWave + lane, g[c]);
    }
  }
  for (int64_t base = 0; base < len; base += kWave) {
    const int chunk = (len - base) < kWave ? static_cast<int>(len - base) : kWave;
    int64_t my = 0;
    if (lane < chunk) my = src[s0 + base + lane];
    for (int r = 0; r < chunk; ++r) {
      const int64_t rowid = readlane64(my, r);
      float* rowp = dx + rowid * ld_dx + slab0;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        if (slab0 + c * kWave + lane < num_features) atomicAdd(rowp + c * kWave + lane, g[c]);
      }
    }
  }
}

// Backward for max: route dout[t,f] to the source row recorded in argmax.
__global__ void gather_max_bwd_kernel(const float* __restrict__ dout, int64_t ld_dout,
                                      const int64_t* __restrict__ src,
                                      const int64_t* __restrict__ argmax, int64_t num_targets,
                                      const int64_t* __restrict__ num_targets_dev,
                                      int num_features, float* __restrict__ dx, int64_t ld_dx) {
  const int64_t nt = dev_count(num_targets_dev, num_targets);
  const int64_t t = blockIdx.x;
  if (t >= nt) return;
  for (int f = threadIdx.x; f < num_features; f += blockDim.x) {
    const int64_t e = argmax[t * num_features + f];
    if (e >= 0) atomicAdd(dx + src[e] * ld_dx + f, dout[t * ld_dout + f]);
  }
}

// Backward of max over rows of P = relu(pool(x)), straight to the gradient of the pool Linear's
// pre-activation: agg[t,f] = P[arg-max row, f], so agg[t,f] > 0 IS the relu mask of the element
// the gradient is routed to — no pass over P, no dense d P.  One workgroup per target.
__global__ void __launch_bounds__(kBlock)
pool_max_bwd_kernel(const float* __restrict__ dagg, int64_t ld_dagg, const float* __restrict__ agg,
                    int64_t ld_agg, const int64_t* __restrict__ src, const int64_t* __restrict__ argmax,
                    int64_t num_targets, const int64_t* __restrict__ num_targets_dev, int num_features,
                    float* __restrict__ dpre, int64_t ld_dpre) {
  const int64_t nt = dev_count(num_targets_dev, num_targets);
  const int64_t t = blockIdx.x;
  if (t >= nt) return;
  for (int f = threadIdx.x; f < num_features; f += kBlock) {
    const int64_t e = argmax[t * num_features + f];
    const float gv = dagg[t * ld_dagg + f];
    if (e >= 0 && agg[t * ld_agg + f] > 0.f) atomicAdd(dpre + src[e] * ld_dpre + f, gv);
  }
}

// The same routing with the batch's source in-degrees at hand (indeg[s] = number of edges of the
// live targets that start at row s): a row with ONE incoming edge is owned by that edge's target,
// which writes the whole row with plain stores (the routed value or 0) — whole lines, no fetch, no
// atomic; only rows shared by several targets take the atomic route (into zeros).
__global__ void __launch_bounds__(kWave)
pool_indeg_kernel(const int64_t* __restrict__ seg_ptr, const int64_t* __restrict__ src, int64_t num_targets,
                  const int64_t* __restrict__ num_targets_dev, int64_t num_sources, int* __restrict__ indeg) {
  const int64_t nt = dev_count(num_targets_dev, num_targets);
  const int64_t t = blockIdx.x;
  if (t >= nt) return;
  const int64_t s1 = seg_ptr[t + 1];
  for (int64_t e = seg_ptr[t] + threadIdx.x; e < s1; e += kWave) {
    const int64_t s = src[e];
    if (s >= 0 && s < num_sources) atomicAdd(indeg + s, 1);
  }
}

constexpr int kPoolCols = 4;   // columns per thread and pass
__global__ void __launch_bounds__(kBlock)
pool_max_bwd_rows_kernel(const float* __restrict__ dagg, int64_t ld_dagg, const float* __restrict__ agg,
                         int64_t ld_agg, const int64_t* __restrict__ seg_ptr, const int64_t* __restrict__ src,
                         const int64_t* __restrict__ argmax, int64_t num_targets,
                         const int64_t* __restrict__ num_targets_dev, int num_features,
                         int64_t num_sources, const int* __restrict__ indeg, float* __restrict__ dpre,
                         int64_t ld_dpre) {
  const int64_t nt = dev_count(num_targets_dev, num_targets);
  const int64_t t = blockIdx.x;
  if (t >= nt) return;
  const int64_t s0 = seg_ptr[t], s1 = seg_ptr[t + 1];
  for (int f0 = 0; f0 < num_features; f0 += kPoolCols * kBlock) {
    int64_t ef[kPoolCols];
    float gf[kPoolCols];
#pragma unroll
    for (int c = 0; c < kPoolCols; ++c) {
      const int f = f0 + c * kBlock + threadIdx.x;
      ef[c] = -1;
      gf[c] = 0.f;
      if (f < num_features) {
        ef[c] = argmax[t * num_features + f];
        const float gv = dagg[t * ld_dagg + f];
        gf[c] = agg[t * ld_agg + f] > 0.f ? gv : 0.f;
      }
    }
    for (int64_t e = s0; e < s1; ++e) {
      const int64_t s = src[e];                     // workgroup-uniform
      if (s < 0 || s >= num_sources) continue;
      float* row = dpre + s * ld_dpre;
      if (indeg[s] == 1) {
#pragma unroll
        for (int c = 0; c < kPoolCols; ++c) {
          const int f = f0 + c * kBlock + threadIdx.x;
          if (f < num_features) row[f] = ef[c] == e ? gf[c] : 0.f;
        }
      } else {
#pragma unroll
        for (int c = 0; c < kPoolCols; ++c) {
          const int f = f0 + c * kBlock + threadIdx.x;
          if (f < num_features && ef[c] == e && gf[c] != 0.f) atomicAdd(row + f, gf[c]);
        }
      }
    }
  }
}

// mean_pool's counterpart: d agg[t] / deg(t) goes to EVERY source row of the segment, masked by
// that row's own P > 0 (read here, once per edge, instead of in a pass over all of P and d P).
__global__ void __launch_bounds__(kBlock)
pool_mean_bwd_rows_kernel(const float* __restrict__ dagg, int64_t ld_dagg, const float* __restrict__ p,
                          int64_t ld_p, const int64_t* __restrict__ seg_ptr, const int64_t* __restrict__ src,
                          int64_t num_targets, const int64_t* __restrict__ num_targets_dev, int num_features,
                          int64_t num_sources, const int* __restrict__ indeg, float* __restrict__ dpre,
                          int64_t ld_dpre) {
  const int64_t nt = dev_count(num_targets_dev, num_targets);
  const int64_t t = blockIdx.x;
  if (t >= nt) return;
  const int64_t s0 = seg_ptr[t], s1 = seg_ptr[t + 1];
  if (s1 <= s0) return;
  const float inv = 1.f / static_cast<float>(s1 - s0);
  for (int f0 = 0; f0 < num_features; f0 += kPoolCols * kBlock) {
    float gf[kPoolCols];
#pragma unroll
    for (int c = 0; c < kPoolCols; ++c) {
      const int f = f0 + c * kBlock + threadIdx.x;
      gf[c] = f < num_features ? dagg[t * ld_dagg + f] * inv : 0.f;
    }
    for (int64_t e = s0; e < s1; ++e) {
      const int64_t s = src[e];                     // workgroup-uniform
      if (s < 0 || s >= num_sources) continue;
      float* row = dpre + s * ld_dpre;
      const float* prow = p + s * ld_p;
      const bool own = indeg == nullptr ? false : indeg[s] == 1;
#pragma unroll
      for (int c = 0; c < kPoolCols; ++c) {
        const int f = f0 + c * kBlock + threadIdx.x;
        if (f < num_features) {
          const float v = prow[f] > 0.f ? gf[c] : 0.f;
          if (own) row[f] = v;
          else if (v != 0.f) atomicAdd(row + f, v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- the edge pass, pulled by source row
// Second form (layer 0 of the engine's pooling step): the batch's edges listed per SOURCE row
// (sage_pool_reverse_edges, which a caller can run as soon as the batch is sampled — the engine
// does, on the sampler's stream, a step ahead), and one wave per source row that adds up what its
// in-edges route to it and stores the row once: no atomics (a shared row's sparse adds cost one L2
// operation per edge and 128-byte line whatever the number of live lanes: 60 of the max pass's 146
// us), no zero-fill to rely on, P read once per row instead of once per edge (mean), 1 byte
// instead of 8 + 4 per (target, column) to find the arg-max edge (max).
// An entry of a row's list: target * 256 + min(offset of the edge in the target's segment, 255).
__global__ void __launch_bounds__(kWave)
pool_reverse_fill_kernel(const int64_t* __restrict__ seg_ptr, const int64_t* __restrict__ src, int64_t num_targets,
                         const int64_t* __restrict__ num_targets_dev, int64_t num_sources,
                         int* __restrict__ count, const int* __restrict__ rev_ptr, int* __restrict__ rev,
                         int64_t rev_cap) {
  const int64_t nt = dev_count(num_targets_dev, num_targets);
  const int64_t t = blockIdx.x;
  if (t >= nt) return;
  const int64_t s0 = seg_ptr[t], s1 = seg_ptr[t + 1];
  for (int64_t e = s0 + threadIdx.x; e < s1; e += kWave) {
    const int64_t s = src[e];
    if (s < 0 || s >= num_sources) continue;
    // the counts of pool_indeg_kernel are handed back one by one: slot = what was left - 1
    const int64_t pos = static_cast<int64_t>(rev_ptr[s]) + atomicSub(count + s, 1) - 1;
    const int64_t off = e - s0;
    if (pos >= 0 && pos < rev_cap) rev[pos] = static_cast<int>(t * 256 + (off < 255 ? off : 255));
  }
}

constexpr int kPullCols = 4;    // consecutive columns per lane and pass (16-byte loads and stores)
constexpr int kPullPasses = 3;  // passes of 64 lanes held in registers together: 768 columns

template <bool MAX>
__global__ void __launch_bounds__(kBlock)
pool_pull_kernel(const float* __restrict__ dagg, int64_t ld_dagg, const uint8_t* __restrict__ arg8,
                 int64_t ld_arg8, const float* __restrict__ p, int64_t ld_p,
                 const int64_t* __restrict__ seg_ptr, const int* __restrict__ rev_ptr,
                 const int* __restrict__ rev, int rev_cap, int64_t num_sources,
                 const int64_t* __restrict__ num_sources_dev, int num_features, float* __restrict__ dpre,
                 int64_t ld_dpre) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t ns = dev_count(num_sources_dev, num_sources);
  const int64_t s = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (s >= ns) return;
  int b0 = rev_ptr[s], b1 = rev_ptr[s + 1];
  if (b0 > rev_cap) b0 = rev_cap;      // (edges that did not fit the list were dropped by the fill)
  if (b1 > rev_cap) b1 = rev_cap;
  for (int f0 = 0; f0 < num_features; f0 += kWave * kPullCols * kPullPasses) {
    float acc[kPullPasses][kPullCols];
    bool live[kPullPasses];
#pragma unroll
    for (int c = 0; c < kPullPasses; ++c) {
      live[c] = f0 + (c * kWave + lane) * kPullCols < num_features;
#pragma unroll
      for (int j = 0; j < kPullCols; ++j) acc[c][j] = 0.f;
    }
    for (int base = b0; base < b1; base += kWave) {
      const int chunk = (b1 - base) < kWave ? (b1 - base) : kWave;
      int my_t = 0, my_j = 0;
      float my_w = 0.f;
      if (lane < chunk) {
        const int ent = rev[base + lane];
        my_t = ent >> 8;
        my_j = ent & 255;
        if (!MAX) my_w = 1.f / static_cast<float>(seg_ptr[my_t + 1] - seg_ptr[my_t]);
      }
      for (int r = 0; r < chunk; ++r) {
        const int64_t tt = __builtin_amdgcn_readlane(my_t, r);
        const uint32_t jj = static_cast<uint32_t>(__builtin_amdgcn_readlane(my_j, r));
        const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_w), r));
        const float* grow = dagg + tt * ld_dagg + f0 + lane * kPullCols;
        float4 g[kPullPasses];
        uint32_t a[kPullPasses];
#pragma unroll
        for (int c = 0; c < kPullPasses; ++c) {
          if (live[c]) {
            g[c] = *reinterpret_cast<const float4*>(grow + c * kWave * kPullCols);
            if (MAX) a[c] = *reinterpret_cast<const uint32_t*>(arg8 + tt * ld_arg8 + f0 + (c * kWave + lane) * kPullCols);
          }
        }
#pragma unroll
        for (int c = 0; c < kPullPasses; ++c) {
          if (!live[c]) continue;
          if (MAX) {
            acc[c][0] += (a[c] & 255u) == jj ? g[c].x : 0.f;
            acc[c][1] += ((a[c] >> 8) & 255u) == jj ? g[c].y : 0.f;
            acc[c][2] += ((a[c] >> 16) & 255u) == jj ? g[c].z : 0.f;
            acc[c][3] += (a[c] >> 24) == jj ? g[c].w : 0.f;
          } else {
            acc[c][0] += w * g[c].x;
            acc[c][1] += w * g[c].y;
            acc[c][2] += w * g[c].z;
            acc[c][3] += w * g[c].w;
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kPullPasses; ++c) {
      if (!live[c]) continue;
      const int f = f0 + (c * kWave + lane) * kPullCols;
      if (!MAX && b1 > b0) {   // the relu mask of the row itself
        const float4 pv = *reinterpret_cast<const float4*>(p + s * ld_p + f);
        if (!(pv.x > 0.f)) acc[c][0] = 0.f;
        if (!(pv.y > 0.f)) acc[c][1] = 0.f;
        if (!(pv.z > 0.f)) acc[c][2] = 0.f;
        if (!(pv.w > 0.f)) acc[c][3] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < kPullCols; ++j) {
        if (f + j >= num_features) acc[c][j] = 0.f;   // (the rows' padding up to a multiple of 4)
      }
      *reinterpret_cast<float4*>(dpre + s * ld_dpre + f) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    }
  }
}

size_t scan32_temp_bytes(size_t n) {
  size_t bytes = 0;
  int* q = nullptr;
  (void)rocprim::exclusive_scan(nullptr, bytes, q, q, 0, n, rocprim::plus<int>(), nullptr);
  return bytes;
}

// The pool bias gradient of the same step: every (t, f) lands in exactly one row of dpre, so the
// column sums of dpre are the column sums of the masked dagg — 13k target rows instead of 81k
// source rows (an empty segment has agg = 0, so agg > 0 alone says "routed and unmasked").
// blockIdx.x: 256 columns, blockIdx.y: a chunk of rows dealt to the block's 4 row groups
// (threadIdx.y), which are summed through LDS: ONE atomic per column and block — adds to one
// address serialise at ~65 ns each, 416 chunks of 32 rows spent 19 us mostly there.
constexpr int kPoolRowsPerBlock = 128;
constexpr int kPoolRowGroups = 4;
__global__ void __launch_bounds__(kBlock * kPoolRowGroups)
pool_max_dbias_kernel(const float* __restrict__ dagg, int64_t ld_dagg, const float* __restrict__ agg,
                      int64_t ld_agg, int64_t num_targets, const int64_t* __restrict__ num_targets_dev,
                      int num_features, float* __restrict__ dbias) {
  __shared__ float part[kPoolRowGroups - 1][kBlock];
  const int64_t nt = dev_count(num_targets_dev, num_targets);
  const int f = blockIdx.x * kBlock + threadIdx.x;
  const int64_t t0 = static_cast<int64_t>(blockIdx.y) * kPoolRowsPerBlock;
  int64_t t1 = t0 + kPoolRowsPerBlock;
  if (t1 > nt) t1 = nt;
  float s0 = 0.f, s1 = 0.f;
  if (f < num_features) {
    int64_t t = t0 + threadIdx.y;
    for (; t + kPoolRowGroups < t1; t += 2 * kPoolRowGroups) {
      const float g0 = dagg[t * ld_dagg + f], g1 = dagg[(t + kPoolRowGroups) * ld_dagg + f];
      const float a0 = agg[t * ld_agg + f], a1 = agg[(t + kPoolRowGroups) * ld_agg + f];
      s0 += a0 > 0.f ? g0 : 0.f;
      s1 += a1 > 0.f ? g1 : 0.f;
    }
    if (t < t1 && agg[t * ld_agg + f] > 0.f) s0 += dagg[t * ld_dagg + f];
  }
  s0 += s1;
  if (threadIdx.y > 0) part[threadIdx.y - 1][threadIdx.x] = s0;
  __syncthreads();
  if (threadIdx.y == 0 && f < num_features && t0 < t1) {
#pragma unroll
    for (int g = 0; g < kPoolRowGroups - 1; ++g) s0 += part[g][threadIdx.x];
    atomicAdd(dbias + f, s0);
  }
}

// ---------------------------------------------------------------- COO -> target CSR
__global__ void iota_hist_kernel(const int64_t* __restrict__ edge_dst, int64_t num_edges,
                                 int64_t num_targets, int64_t* __restrict__ iota,
                                 unsigned long long* __restrict__ counts) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= num_edges) return;
  iota[e] = e;
  const int64_t d = edge_dst[e];
  if (d >= 0 && d < num_targets) atomicAdd(&counts[d], 1ull);
}

__global__ void permute_src_kernel(const int64_t* __restrict__ edge_src,
                                   const int64_t* __restrict__ perm, int64_t num_edges,
                                   int64_t* __restrict__ src_sorted) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e < num_edges) src_sorted[e] = edge_src[perm[e]];
}

int bits_for(int64_t n) {
  int b = 1;
  while (b < 63 && (1ll << b) < n) ++b;
  return b;
}

size_t sort_temp_bytes(size_t n, int bits) {
  size_t bytes = 0;
  int64_t* p = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, n, 0, bits, nullptr);
  return bytes;
}
size_t scan64_temp_bytes(size_t n) {
  size_t bytes = 0;
  int64_t* p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, bytes, p, p, int64_t(0), n, rocprim::plus<int64_t>(), nullptr);
  return bytes;
}

bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Widest vector width (in 32-bit words) usable for rows of `width` words starting at p with
// stride ld.  A ragged last vector is allowed when it stays inside the row stride (the
// caller pads rows, e.g. 602 -> 608 floats), so it never touches another row.
int pick_vec(const void* p, int64_t ld, int width) {
  if (aligned(p, 16) && ld % 4 == 0 && ld >= (width + 3) / 4 * 4) return 4;
  if (aligned(p, 8) && ld % 2 == 0 && ld >= (width + 1) / 2 * 2) return 2;
  return 1;
}

}  // namespace
}  // namespace sage

using namespace sage;

extern "C" int sage_gather_rows(const void* table, sage_i64 ld_table, const sage_i64* index,
                                sage_i64 num_rows, const sage_i64* num_rows_dev, int width,
                                void* out, sage_i64 ld_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(width > 0 && ld_table >= width && ld_out >= width && num_rows >= 0, SAGE_EINVAL,
               "sage_gather_rows: bad shape (rows=%lld width=%d ld_table=%lld ld_out=%lld)",
               (long long)num_rows, width, (long long)ld_table, (long long)ld_out);
  if (num_rows == 0) return SAGE_OK;
  SAGE_REQUIRE(table && index && out, SAGE_EINVAL, "sage_gather_rows: NULL pointer argument");
  const uint32_t* tp = static_cast<const uint32_t*>(table);
  uint32_t* op = static_cast<uint32_t*>(out);
  if (width < 32) {
    const int64_t total = num_rows * width;
    gather_rows_narrow_kernel<<<static_cast<unsigned>(cdiv(total, kBlock)), kBlock, 0, stream>>>(
        tp, ld_table, index, num_rows, num_rows_dev, width, op, ld_out);
    SAGE_CHECK_LAUNCH("gather_rows_narrow_kernel");
    return SAGE_OK;
  }
  int vec = pick_vec(table, ld_table, width);
  const int vo = pick_vec(out, ld_out, width);
  if (vo < vec) vec = vo;
  const unsigned grid = static_cast<unsigned>(cdiv(cdiv(num_rows, kRowsInFlight), kWavesPerBlock));
  if (vec == 4) {
    gather_rows_wide_kernel<4><<<grid, kBlock, 0, stream>>>(tp, ld_table, index, num_rows,
                                                            num_rows_dev, width, op, ld_out);
  } else if (vec == 2) {
    gather_rows_wide_kernel<2><<<grid, kBlock, 0, stream>>>(tp, ld_table, index, num_rows,
                                                            num_rows_dev, width, op, ld_out);
  } else {
    gather_rows_wide_kernel<1><<<grid, kBlock, 0, stream>>>(tp, ld_table, index, num_rows,
                                                            num_rows_dev, width, op, ld_out);
  }
  SAGE_CHECK_LAUNCH("gather_rows_wide_kernel");
  return SAGE_OK;
}

namespace {

struct LaunchEvents {
  hipEvent_t start, stop;
};

template <int VEC, int NV>
int launch_fwd(const float* x, int64_t ld_x, const int64_t* node_map, const int64_t* seg_ptr,
               const int64_t* src, int64_t nt, const int64_t* nt_dev, int nf, int reduce,
               float* out, int64_t ld_out, ArgOut argmax, const LaunchEvents& ev,
               hipStream_t stream) {
  dim3 grid(static_cast<unsigned>(cdiv(nt, kWavesPerBlock)),
            static_cast<unsigned>(cdiv(nf, kWave * VEC * NV)));
  // with events: the launch itself carries them (they read the dispatch's own start / end
  // timestamps instead of the time between two markers around it)
#define SAGE_LAUNCH_FWD(R)                                                                      \
  do {                                                                                          \
    if (ev.start != nullptr) {                                                                  \
      hipExtLaunchKernelGGL((gather_reduce_fwd_kernel<VEC, NV, R>), grid, dim3(kBlock), 0,      \
                            stream, ev.start, ev.stop, 0, x, ld_x, node_map, seg_ptr, src, nt,  \
                            nt_dev, nf, out, ld_out, argmax);                                   \
    } else {                                                                                    \
      gather_reduce_fwd_kernel<VEC, NV, R><<<grid, kBlock, 0, stream>>>(                        \
          x, ld_x, node_map, seg_ptr, src, nt, nt_dev, nf, out, ld_out, argmax);                \
    }                                                                                           \
  } while (0)
  if (reduce == SAGE_REDUCE_MEAN) {
    SAGE_LAUNCH_FWD(SAGE_REDUCE_MEAN);
  } else if (reduce == SAGE_REDUCE_SUM) {
    SAGE_LAUNCH_FWD(SAGE_REDUCE_SUM);
  } else if (reduce == SAGE_REDUCE_GCN) {
    SAGE_LAUNCH_FWD(SAGE_REDUCE_GCN);
  } else {
    SAGE_LAUNCH_FWD(SAGE_REDUCE_MAX);
  }
#undef SAGE_LAUNCH_FWD
  SAGE_CHECK_LAUNCH("gather_reduce_fwd_kernel");
  return SAGE_OK;
}

template <int VEC>
int dispatch_fwd_nv(int nv, const float* x, int64_t ld_x, const int64_t* node_map,
                    const int64_t* seg_ptr, const int64_t* src, int64_t nt, const int64_t* nt_dev,
                    int nf, int reduce, float* out, int64_t ld_out, ArgOut argmax,
                    const LaunchEvents& ev, hipStream_t stream) {
  switch (nv) {
    case 1:
      return launch_fwd<VEC, 1>(x, ld_x, node_map, seg_ptr, src, nt, nt_dev, nf, reduce, out, ld_out, argmax, ev, stream);
    case 2:
      return launch_fwd<VEC, 2>(x, ld_x, node_map, seg_ptr, src, nt, nt_dev, nf, reduce, out, ld_out, argmax, ev, stream);
    case 3:
      return launch_fwd<VEC, 3>(x, ld_x, node_map, seg_ptr, src, nt, nt_dev, nf, reduce, out, ld_out, argmax, ev, stream);
    default:
      return launch_fwd<VEC, 4>(x, ld_x, node_map, seg_ptr, src, nt, nt_dev, nf, reduce, out, ld_out, argmax, ev, stream);
  }
}

int gather_reduce_fwd_impl(const float* x, sage_i64 ld_x, const sage_i64* node_map,
                           const sage_i64* seg_ptr, const sage_i64* src, sage_i64 num_targets,
                           const sage_i64* num_targets_dev, int num_features, int reduce,
                           float* out, sage_i64 ld_out, ArgOut argmax, const LaunchEvents& ev,
                           void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(reduce == SAGE_REDUCE_SUM || reduce == SAGE_REDUCE_MEAN || reduce == SAGE_REDUCE_MAX ||
                   reduce == SAGE_REDUCE_GCN,
               SAGE_EINVAL, "sage_gather_reduce_fwd: unknown reduce %d", reduce);
  SAGE_REQUIRE(num_features > 0 && ld_x >= num_features && ld_out >= num_features && num_targets >= 0,
               SAGE_EINVAL, "sage_gather_reduce_fwd: bad shape (T=%lld F=%d ld_x=%lld ld_out=%lld)",
               (long long)num_targets, num_features, (long long)ld_x, (long long)ld_out);
  if (num_targets == 0) return SAGE_OK;
  SAGE_REQUIRE(x && seg_ptr && src && out, SAGE_EINVAL, "sage_gather_reduce_fwd: NULL pointer argument");
  int vec = pick_vec(x, ld_x, num_features);
  const int vo = pick_vec(out, ld_out, num_features);
  if (vo < vec) vec = vo;
  const int per = kWave * vec;
  int nv = static_cast<int>(cdiv(num_features, per));
  if (nv > 4) nv = 4;
  if (vec == 4)
    return dispatch_fwd_nv<4>(nv, x, ld_x, node_map, seg_ptr, src, num_targets, num_targets_dev,
                              num_features, reduce, out, ld_out, argmax, ev, stream);
  if (vec == 2)
    return dispatch_fwd_nv<2>(nv, x, ld_x, node_map, seg_ptr, src, num_targets, num_targets_dev,
                              num_features, reduce, out, ld_out, argmax, ev, stream);
  return dispatch_fwd_nv<1>(nv, x, ld_x, node_map, seg_ptr, src, num_targets, num_targets_dev,
                            num_features, reduce, out, ld_out, argmax, ev, stream);
}

}  // namespace

extern "C" int sage_gather_reduce_fwd(const float* x, sage_i64 ld_x, const sage_i64* node_map,
                                      const sage_i64* seg_ptr, const sage_i64* src,
                                      sage_i64 num_targets, const sage_i64* num_targets_dev,
                                      int num_features, int reduce, float* out, sage_i64 ld_out,
                                      sage_i64* argmax, void* stream) {
  return gather_reduce_fwd_impl(x, ld_x, node_map, seg_ptr, src, num_targets, num_targets_dev,
                                num_features, reduce, out, ld_out, ArgOut{argmax, nullptr, 0},
                                LaunchEvents{nullptr, nullptr}, stream);
}

extern "C" int sage_pool_max_fwd(const float* p, sage_i64 ld_p, const sage_i64* seg_ptr, const sage_i64* src,
                                 sage_i64 num_targets, const sage_i64* num_targets_dev, int num_features,
                                 int max_segment, float* out, sage_i64 ld_out, unsigned char* arg8,
                                 sage_i64 ld_arg8, void* stream) {
  SAGE_REQUIRE(arg8 != nullptr && ld_arg8 >= num_features && reinterpret_cast<uintptr_t>(arg8) % 4 == 0,
               SAGE_EINVAL, "sage_pool_max_fwd: arg8 NULL, rows shorter than num_features, or not 4-byte aligned");
  SAGE_REQUIRE(max_segment >= 0 && max_segment <= 255, SAGE_EINVAL,
               "sage_pool_max_fwd: segments of up to %d edges do not fit the one-byte offsets (255 at most); "
               "use sage_gather_reduce_fwd with the int64 argmax",
               max_segment);
  return gather_reduce_fwd_impl(p, ld_p, nullptr, seg_ptr, src, num_targets, num_targets_dev, num_features,
                                SAGE_REDUCE_MAX, out, ld_out, ArgOut{nullptr, arg8, ld_arg8},
                                LaunchEvents{nullptr, nullptr}, stream);
}

extern "C" int sage_gather_reduce_fwd_timed(const float* x, sage_i64 ld_x, const sage_i64* node_map,
                                            const sage_i64* seg_ptr, const sage_i64* src,
                                            sage_i64 num_targets, const sage_i64* num_targets_dev,
                                            int num_features, int reduce, float* out,
                                            sage_i64 ld_out, sage_i64* argmax, void* start_event,
                                            void* stop_event, void* stream) {
  SAGE_REQUIRE(start_event != nullptr && stop_event != nullptr, SAGE_EINVAL,
               "sage_gather_reduce_fwd_timed: NULL event");
  SAGE_REQUIRE(num_targets > 0, SAGE_EINVAL,
               "sage_gather_reduce_fwd_timed: nothing to launch (num_targets = 0), the events would stay unrecorded");
  return gather_reduce_fwd_impl(x, ld_x, node_map, seg_ptr, src, num_targets, num_targets_dev,
                                num_features, reduce, out, ld_out, ArgOut{argmax, nullptr, 0},
                                LaunchEvents{static_cast<hipEvent_t>(start_event),
                                             static_cast<hipEvent_t>(stop_event)},
                                stream);
}

extern "C" int sage_pool_max_bwd(const float* dagg, sage_i64 ld_dagg, const float* agg, sage_i64 ld_agg,
                                 const sage_i64* seg_ptr, const sage_i64* src, const sage_i64* argmax,
                                 sage_i64 num_targets, const sage_i64* num_targets_dev, int num_features,
                                 float* dpre, sage_i64 ld_dpre, sage_i64 num_sources, int* indeg,
                                 float* dbias, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(num_features > 0 && ld_dagg >= num_features && ld_agg >= num_features &&
                   ld_dpre >= num_features && num_targets >= 0 && num_sources >= 0,
               SAGE_EINVAL, "sage_pool_max_bwd: bad shape");
  if (num_targets == 0) return SAGE_OK;
  SAGE_REQUIRE(dagg && agg && src && argmax && dpre && (indeg == nullptr || seg_ptr != nullptr), SAGE_EINVAL,
               "sage_pool_max_bwd: NULL pointer argument");
  if (indeg != nullptr) {
    if (num_sources > 0) {     // (no edge passes the kernels' source-range check otherwise)
      zero_i32_kernel<<<static_cast<unsigned>(cdiv(num_sources, kBlock)), kBlock, 0, stream>>>(indeg, num_sources);
      SAGE_CHECK_LAUNCH("zero_i32_kernel");
    }
    pool_indeg_kernel<<<static_cast<unsigned>(num_targets), kWave, 0, stream>>>(
        seg_ptr, src, num_targets, num_targets_dev, num_sources, indeg);
    SAGE_CHECK_LAUNCH("pool_indeg_kernel");
    // (a 16-byte variant of this kernel measured the same: 142 us against 200 us all-atomic on a
    // Reddit-shaped batch, where 60 % of the edges end in rows shared by several targets)
    // (not latency: staging the sources and flags of a segment's edges in LDS ahead of the stores
    // changed nothing, 146 us; knock-outs on that batch: without the atomics 89, without the plain
    // stores 141, with neither 78 — see pool_pull_kernel)
    pool_max_bwd_rows_kernel<<<static_cast<unsigned>(num_targets), kBlock, 0, stream>>>(
        dagg, ld_dagg, agg, ld_agg, seg_ptr, src, argmax, num_targets, num_targets_dev, num_features,
        num_sources, indeg, dpre, ld_dpre);
    SAGE_CHECK_LAUNCH("pool_max_bwd_rows_kernel");
  } else {
    pool_max_bwd_kernel<<<static_cast<unsigned>(num_targets), kBlock, 0, stream>>>(
        dagg, ld_dagg, agg, ld_agg, src, argmax, num_targets, num_targets_dev, num_features, dpre, ld_dpre);
    SAGE_CHECK_LAUNCH("pool_max_bwd_kernel");
  }
  if (dbias != nullptr) {
    const dim3 grid(static_cast<unsigned>(cdiv(num_features, kBlock)),
                    static_cast<unsigned>(cdiv(num_targets, kPoolRowsPerBlock)));
    pool_max_dbias_kernel<<<grid, dim3(kBlock, kPoolRowGroups), 0, stream>>>(dagg, ld_dagg, agg, ld_agg, num_targets,
                                                       num_targets_dev, num_features, dbias);
    SAGE_CHECK_LAUNCH("pool_max_dbias_kernel");
  }
  return SAGE_OK;
}

extern "C" int sage_pool_mean_bwd(const float* dagg, sage_i64 ld_dagg, const float* p, sage_i64 ld_p,
                                  const sage_i64* seg_ptr, const sage_i64* src, sage_i64 num_targets,
                                  const sage_i64* num_targets_dev, int num_features, float* dpre,
                                  sage_i64 ld_dpre, sage_i64 num_sources, int* indeg, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(num_features > 0 && ld_dagg >= num_features && ld_p >= num_features &&
                   ld_dpre >= num_features && num_targets >= 0 && num_sources >= 0,
               SAGE_EINVAL, "sage_pool_mean_bwd: bad shape");
  if (num_targets == 0) return SAGE_OK;
  SAGE_REQUIRE(dagg && p && seg_ptr && src && dpre && p != dpre, SAGE_EINVAL,
               "sage_pool_mean_bwd: NULL pointer argument or dpre aliasing p");
  if (indeg != nullptr) {
    if (num_sources > 0) {     // (no edge passes the kernels' source-range check otherwise)
      zero_i32_kernel<<<static_cast<unsigned>(cdiv(num_sources, kBlock)), kBlock, 0, stream>>>(indeg, num_sources);
      SAGE_CHECK_LAUNCH("zero_i32_kernel");
    }
    pool_indeg_kernel<<<static_cast<unsigned>(num_targets), kWave, 0, stream>>>(
        seg_ptr, src, num_targets, num_targets_dev, num_sources, indeg);
    SAGE_CHECK_LAUNCH("pool_indeg_kernel");
  }
  pool_mean_bwd_rows_kernel<<<static_cast<unsigned>(num_targets), kBlock, 0, stream>>>(
      dagg, ld_dagg, p, ld_p, seg_ptr, src, num_targets, num_targets_dev, num_features, num_sources, indeg,
      dpre, ld_dpre);
  SAGE_CHECK_LAUNCH("pool_mean_bwd_rows_kernel");
  return SAGE_OK;
}

extern "C" size_t sage_pool_reverse_edges_workspace_bytes(sage_i64 num_sources) {
  if (num_sources < 0) return 0;
  return align_up(scan32_temp_bytes(static_cast<size_t>(num_sources) + 1), 256) + 256;
}

extern "C" int sage_pool_reverse_edges(const sage_i64* seg_ptr, const sage_i64* src, sage_i64 num_targets,
                                       const sage_i64* num_targets_dev, sage_i64 num_sources, int* count,
                                       int* rev_ptr, int* rev, sage_i64 rev_cap, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(num_targets >= 0 && num_sources >= 0 && rev_cap >= 0, SAGE_EINVAL,
               "sage_pool_reverse_edges: negative size");
  SAGE_REQUIRE(num_targets < (1ll << 23) && num_sources < (1ll << 31) - 1 && rev_cap < (1ll << 31), SAGE_EINVAL,
               "sage_pool_reverse_edges: a list entry is target * 256 + offset in 32 bits (targets below 2^23)");
  SAGE_REQUIRE(seg_ptr && src && count && rev_ptr && (rev || rev_cap == 0) && workspace, SAGE_EINVAL,
               "sage_pool_reverse_edges: NULL pointer argument");
  SAGE_REQUIRE(workspace_bytes >= sage_pool_reverse_edges_workspace_bytes(num_sources), SAGE_EWORKSPACE,
               "sage_pool_reverse_edges: workspace too small");
  zero_i32_kernel<<<static_cast<unsigned>(cdiv(num_sources + 1, kBlock)), kBlock, 0, stream>>>(count, num_sources + 1);
  SAGE_CHECK_LAUNCH("zero_i32_kernel");
  if (num_targets > 0 && num_sources > 0) {
    pool_indeg_kernel<<<static_cast<unsigned>(num_targets), kWave, 0, stream>>>(
        seg_ptr, src, num_targets, num_targets_dev, num_sources, count);
    SAGE_CHECK_LAUNCH("pool_indeg_kernel");
  }
  Carver ws(workspace);
  size_t scan_bytes = scan32_temp_bytes(static_cast<size_t>(num_sources) + 1);
  void* scan_tmp = ws.take<char>(scan_bytes);
  SAGE_CHECK_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, count, rev_ptr, 0,
                                         static_cast<size_t>(num_sources) + 1, rocprim::plus<int>(), stream),
                 "scan(in-degrees)");
  if (num_targets > 0 && num_sources > 0) {
    pool_reverse_fill_kernel<<<static_cast<unsigned>(num_targets), kWave, 0, stream>>>(
        seg_ptr, src, num_targets, num_targets_dev, num_sources, count, rev_ptr, rev, rev_cap);
    SAGE_CHECK_LAUNCH("pool_reverse_fill_kernel");
  }
  return SAGE_OK;
}

namespace {
int pool_pull_check(const char* who, const float* dagg, sage_i64 ld_dagg, const int* rev_ptr, const int* rev,
                    sage_i64 rev_cap, sage_i64 num_sources, int num_features, float* dpre, sage_i64 ld_dpre) {
  const sage_i64 padded = (static_cast<sage_i64>(num_features) + 3) / 4 * 4;
  SAGE_REQUIRE(num_features > 0 && num_sources >= 0 && rev_cap >= 0 && rev_cap < (1ll << 31) && ld_dagg >= padded &&
                   ld_dpre >= padded,
               SAGE_EINVAL,
               "%s: bad shape (rows must hold num_features rounded up to a multiple of 4)", who);
  if (num_sources == 0) return SAGE_OK;
  SAGE_REQUIRE(dagg && rev_ptr && rev && dpre, SAGE_EINVAL, "%s: NULL pointer argument", who);
  SAGE_REQUIRE(reinterpret_cast<uintptr_t>(dagg) % 16 == 0 && ld_dagg % 4 == 0 &&
                   reinterpret_cast<uintptr_t>(dpre) % 16 == 0 && ld_dpre % 4 == 0,
               SAGE_EINVAL, "%s: rows of dagg / dpre must be 16-byte aligned (sage_pool_max_bwd / _mean_bwd take any)", who);
  return SAGE_OK;
}
}  // namespace

extern "C" int sage_pool_max_bwd_pull(const float* dagg, sage_i64 ld_dagg, const unsigned char* arg8,
                                      sage_i64 ld_arg8, const int* rev_ptr, const int* rev, sage_i64 rev_cap,
                                      sage_i64 num_sources, const sage_i64* num_sources_dev, int num_features,
                                      float* dpre, sage_i64 ld_dpre, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rc = pool_pull_check("sage_pool_max_bwd_pull", dagg, ld_dagg, rev_ptr, rev, rev_cap, num_sources, num_features, dpre, ld_dpre);
  if (rc != SAGE_OK || num_sources == 0) return rc;
  SAGE_REQUIRE(arg8 && reinterpret_cast<uintptr_t>(arg8) % 4 == 0 && ld_arg8 % 4 == 0 &&
                   ld_arg8 >= (static_cast<sage_i64>(num_features) + 3) / 4 * 4,
               SAGE_EINVAL, "sage_pool_max_bwd_pull: arg8 NULL, or its rows not 4-byte aligned / shorter than the padded width");
  pool_pull_kernel<true><<<static_cast<unsigned>(cdiv(num_sources, kWavesPerBlock)), kBlock, 0, stream>>>(
      dagg, ld_dagg, arg8, ld_arg8, nullptr, 0, nullptr, rev_ptr, rev, static_cast<int>(rev_cap), num_sources,
      num_sources_dev, num_features, dpre, ld_dpre);
  SAGE_CHECK_LAUNCH("pool_pull_kernel(max)");
  return SAGE_OK;
}

extern "C" int sage_pool_mean_bwd_pull(const float* dagg, sage_i64 ld_dagg, const float* p, sage_i64 ld_p,
                                       const sage_i64* seg_ptr, const int* rev_ptr, const int* rev,
                                       sage_i64 rev_cap, sage_i64 num_sources, const sage_i64* num_sources_dev,
                                       int num_features, float* dpre, sage_i64 ld_dpre, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rc = pool_pull_check("sage_pool_mean_bwd_pull", dagg, ld_dagg, rev_ptr, rev, rev_cap, num_sources, num_features, dpre, ld_dpre);
  if (rc != SAGE_OK || num_sources == 0) return rc;
  SAGE_REQUIRE(p && seg_ptr && p != dpre && reinterpret_cast<uintptr_t>(p) % 16 == 0 && ld_p % 4 == 0 &&
                   ld_p >= (static_cast<sage_i64>(num_features) + 3) / 4 * 4,
               SAGE_EINVAL, "sage_pool_mean_bwd_pull: p / seg_ptr NULL, dpre aliasing p, or rows of p not 16-byte aligned");
  pool_pull_kernel<false><<<static_cast<unsigned>(cdiv(num_sources, kWavesPerBlock)), kBlock, 0, stream>>>(
      dagg, ld_dagg, nullptr, 0, p, ld_p, seg_ptr, rev_ptr, rev, static_cast<int>(rev_cap), num_sources,
      num_sources_dev, num_features, dpre, ld_dpre);
  SAGE_CHECK_LAUNCH("pool_pull_kernel(mean)");
  return SAGE_OK;
}

extern "C" int sage_gather_reduce_bwd(const float* dout, sage_i64 ld_dout, const sage_i64* seg_ptr,
                                      const sage_i64* src, const sage_i64* argmax,
                                      sage_i64 num_targets, const sage_i64* num_targets_dev,
                                      int num_features, int reduce, float* dx, sage_i64 ld_dx,
                                      void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(reduce == SAGE_REDUCE_SUM || reduce == SAGE_REDUCE_MEAN || reduce == SAGE_REDUCE_MAX ||
                   reduce == SAGE_REDUCE_GCN,
               SAGE_EINVAL, "sage_gather_reduce_bwd: unknown reduce %d", reduce);
  SAGE_REQUIRE(num_features > 0 && ld_dout >= num_features && ld_dx >= num_features && num_targets >= 0,
               SAGE_EINVAL, "sage_gather_reduce_bwd: bad shape");
  if (num_targets == 0) return SAGE_OK;
  SAGE_REQUIRE(dout && seg_ptr && src && dx, SAGE_EINVAL, "sage_gather_reduce_bwd: NULL pointer argument");
  if (reduce == SAGE_REDUCE_MAX) {
    SAGE_REQUIRE(argmax != nullptr, SAGE_EINVAL, "sage_gather_reduce_bwd: max needs argmax");
    gather_max_bwd_kernel<<<static_cast<unsigned>(num_targets), kBlock, 0, stream>>>(
        dout, ld_dout, src, argmax, num_targets, num_targets_dev, num_features, dx, ld_dx);
    SAGE_CHECK_LAUNCH("gather_max_bwd_kernel");
    return SAGE_OK;
  }
  const int mean = reduce == SAGE_REDUCE_MEAN ? 1 : (reduce == SAGE_REDUCE_GCN ? 2 : 0);
  int nv = static_cast<int>(cdiv(num_features, kWave));
  if (nv > 4) nv = 4;
  dim3 grid(static_cast<unsigned>(cdiv(num_targets, kWavesPerBlock)),
            static_cast<unsigned>(cdiv(num_features, kWave * nv)));
  switch (nv) {
    case 1:
      gather_reduce_bwd_kernel<1><<<grid, kBlock, 0, stream>>>(dout, ld_dout, seg_ptr, src, num_targets, num_targets_dev, num_features, mean, dx, ld_dx);
      break;
    case 2:
      gather_reduce_bwd_kernel<2><<<grid, kBlock, 0, stream>>>(dout, ld_dout, seg_ptr, src, num_targets, num_targets_dev, num_features, mean, dx, ld_dx);
      break;
    case 3:
      gather_reduce_bwd_kernel<3><<<grid, kBlock, 0, stream>>>(dout, ld_dout, seg_ptr, src, num_targets, num_targets_dev, num_features, mean, dx, ld_dx);
      break;
    default:
      gather_reduce_bwd_kernel<4><<<grid, kBlock, 0, stream>>>(dout, ld_dout, seg_ptr, src, num_targets, num_targets_dev, num_features, mean, dx, ld_dx);
      break;
  }
  SAGE_CHECK_LAUNCH("gather_reduce_bwd_kernel");
  return SAGE_OK;
}

extern "C" size_t sage_coo_to_csr_workspace_bytes(sage_i64 num_edges, sage_i64 num_targets) {
  if (num_edges < 0 || num_targets < 0) return 0;
  size_t total = 0;
  total += 2 * align_up(sizeof(int64_t) * (static_cast<size_t>(num_edges) + 1), 256);  // iota, keys_out
  total += align_up(sizeof(int64_t) * (static_cast<size_t>(num_targets) + 2), 256);  // counts
  total += align_up(sort_temp_bytes(static_cast<size_t>(num_edges) + 1, bits_for(num_targets + 1)), 256);
  total += align_up(scan64_temp_bytes(static_cast<size_t>(num_targets) + 2), 256);
  return total + 256;
}

extern "C" int sage_coo_to_csr(const sage_i64* edge_src, const sage_i64* edge_dst,
                               sage_i64 num_edges, sage_i64 num_targets, sage_i64* seg_ptr,
                               sage_i64* src_sorted, sage_i64* perm, void* workspace,
                               size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(num_edges >= 0 && num_targets >= 0, SAGE_EINVAL, "sage_coo_to_csr: negative size");
  SAGE_REQUIRE(seg_ptr && workspace, SAGE_EINVAL, "sage_coo_to_csr: NULL pointer argument");
  SAGE_REQUIRE(workspace_bytes >= sage_coo_to_csr_workspace_bytes(num_edges, num_targets),
               SAGE_EWORKSPACE, "sage_coo_to_csr: workspace too small");
  Carver ws(workspace);
  int64_t* iota = ws.take<int64_t>(num_edges + 1);
  int64_t* keys_out = ws.take<int64_t>(num_edges + 1);
  int64_t* counts = ws.take<int64_t>(num_targets + 2);
  const int bits = bits_for(num_targets + 1);
  size_t sort_bytes = sort_temp_bytes(static_cast<size_t>(num_edges) + 1, bits);
  void* sort_tmp = ws.take<char>(sort_bytes);
  size_t scan_bytes = scan64_temp_bytes(static_cast<size_t>(num_targets) + 2);
  void* scan_tmp = ws.take<char>(scan_bytes);

  {
    const int64_t nz = (num_targets + 1);
    zero_i64_kernel<<<static_cast<unsigned>(cdiv(nz, kBlock)), kBlock, 0, stream>>>(counts, nz);
    SAGE_CHECK_LAUNCH("zero_i64_kernel");
  }
  if (num_edges > 0) {
    SAGE_REQUIRE(edge_src && edge_dst && src_sorted && perm, SAGE_EINVAL,
                 "sage_coo_to_csr: NULL pointer argument");
    iota_hist_kernel<<<static_cast<unsigned>(cdiv(num_edges, kBlock)), kBlock, 0, stream>>>(
        edge_dst, num_edges, num_targets, iota, reinterpret_cast<unsigned long long*>(counts));
    SAGE_CHECK_LAUNCH("iota_hist_kernel");
    SAGE_CHECK_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, edge_dst, keys_out, iota, perm,
                                             static_cast<size_t>(num_edges), 0, bits, stream),
                   "radix_sort_pairs");
    permute_src_kernel<<<static_cast<unsigned>(cdiv(num_edges, kBlock)), kBlock, 0, stream>>>(
        edge_src, perm, num_edges, src_sorted);
    SAGE_CHECK_LAUNCH("permute_src_kernel");
  }
  SAGE_CHECK_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, counts, seg_ptr, int64_t(0),
                                         static_cast<size_t>(num_targets + 1),
                                         rocprim::plus<int64_t>(), stream),
                 "scan(counts)");
  return SAGE_OK;
}
