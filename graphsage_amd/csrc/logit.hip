// The class-logit layer of the supervised step (the last SAGE layer acts on the B seed rows only: 512
// targets, <= 25 sources each, C = 41 outputs — ~0.1 GFLOP and ~30 MB, a chain of short dependent
// kernels when it is run as gather / GEMM / normalise+loss / d-input / scatter).
//
//   forward  (one wave per seed, the [C, 2H] weights staged in LDS once per workgroup):
//     agg = reduce_{e in seg(t)} h[src[e]]   (sum / mean / max / gcn) layers/sage.py:136-137, 195-210
//     z   = agg . W_l^T + h[t] . W_r^T                               layers/sage.py:124-129
//     y   = z / max(|z|, eps);  CE(y, label);  dz = d loss / d z     layers/sage.py:131-132,
//                                                                    fig2a/runners/graphsage_runners.py:43
//   backward (the same wave, right behind its forward in the training step's launch — BWD; or a
//   launch of its own, logit_bwd_kernel, mean / sum only):
//     dh[src[e]] += (dz . W_l) / deg   for the segment's edges,   dh[t] += dz . W_r
//     (what autograd derives for the two Linears and the scatter; rows that only one seed writes
//     are stored instead of added; max sends each column to the row that supplied it)
//   forward only (sage_logit_layer_infer): the arg-max class of every seed row.
//   The training step's common shape (H = 256; mean / sum / max / gcn on pre-activation rows, the
//   pooling aggregators' last layer) runs forward + backward with FOUR waves per seed
//   (logit_fb4_kernel), everything else with one (logit_fwd_kernel).
// The layer's weight gradients and the loss mean need every seed's dz: they ride as extra
// workgroups in front of another launch's grid (skinny_dw.h; rowops.hip's normalise-backward pass
// in the engine's step, or logit_bwd_dw_kernel below).
#include "common.h"
#include "skinny_dw.h"

#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

namespace sage {
namespace {

#ifdef SAGE_LOGIT_PROF      // diagnostic build (scratch/logit_prof.hip): cycle stamps per phase, never in the library
__device__ unsigned long long g_logit_prof[2][16];
#define LOGIT_STAMP(k, i) do { if (threadIdx.x == 0 && blockIdx.x == 3) g_logit_prof[k][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LOGIT_STAMP(k, i) do { } while (0)
#endif

constexpr int kLogitBlock = 256;                       // 4 seeds per workgroup, one wave per SIMD (the products are VALU work)
constexpr int kLogitWaves = kLogitBlock / kWave;
constexpr float kNormEps = 1e-12f;                     // torch.nn.functional.normalize default
constexpr int kRowRegs = 16;         // float4 registers of gathered rows in flight per lane (1 KB rows from L2 /
                                     // MALL, latency-bound: 16 rows of 256 floats at once)

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Sum over the 64 lanes as a wave-uniform value, on the DPP path (no LDS crossbar traffic: the
// butterflies above go through ds_bpermute): xor 1, xor 2, mirror within 8, mirror within 16 give
// every lane its 16-lane row's sum; row_bcast:15 / :31 chain the four rows into lane 63.
template <typename Op>
__device__ inline float wave_reduce_dpp(float v, Op op) {
  auto step = [&](float x, auto ctrl, auto row_mask) {
    // (masked rows and invalid source lanes keep x itself: op(x, x) == x for max, and the sum passes 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, op.neutral(x)), __builtin_bit_cast(int, x),
                                                                 decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false));
  };
  v = op(v, step(v, std::integral_constant<int, 0xb1>{}, std::integral_constant<int, 0xf>{}));     // quad_perm:[1,0,3,2]
  v = op(v, step(v, std::integral_constant<int, 0x4e>{}, std::integral_constant<int, 0xf>{}));     // quad_perm:[2,3,0,1]
  v = op(v, step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{}));    // row_half_mirror
  v = op(v, step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{}));    // row_mirror
  v = op(v, step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}));    // row_bcast:15 into rows 1, 3
  v = op(v, step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{}));    // row_bcast:31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
struct SumOp {
  __device__ float operator()(float a, float b) const { return a + b; }
  __device__ float neutral(float) const { return 0.f; }
};
struct MaxOp {
  __device__ float operator()(float a, float b) const { return fmaxf(a, b); }
  __device__ float neutral(float x) const { return x; }
};
__device__ inline float wave_total(float v) { return wave_reduce_dpp(v, SumOp{}); }
// Sum over the 16 lanes of a row, in every lane of the row (the first four steps of the above).
__device__ inline float row16_total(float v) {
  auto step = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v += step(v, std::integral_constant<int, 0xb1>{});      // quad_perm:[1,0,3,2]
  v += step(v, std::integral_constant<int, 0x4e>{});      // quad_perm:[2,3,0,1]
  v += step(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
  v += step(v, std::integral_constant<int, 0x140>{});     // row_mirror
  return v;
}
// v_permlane32_swap / v_permlane16_swap (gfx950): lanes 32-63 (the odd 16-lane rows) of a change
// places with lanes 0-31 (the even rows) of b — one VALU instruction, no LDS crossbar.  After it
// a + b is, in the lower half (even rows), the sum of the old a over the lane pair, and in the
// upper half (odd rows) the sum of the old b: one step of a transposing reduction.
__device__ inline float swap32_sum(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float swap16_sum(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float wave_top(float v) { return wave_reduce_dpp(v, MaxOp{}); }
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ inline int64_t readlane64(int64_t v, int lane) {
  const int lo = __builtin_amdgcn_readlane(static_cast<int>(v & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), lane);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

// Both kernels are chains of dependent memory round trips (segment bounds -> neighbour ids -> rows
// -> ...), so everything independent is requested up front: the [n, 2h] weights go to LDS as
// lds[c][2 H] (row c = [W_l[c, :] | W_r[c, :]]) through registers, the loads of the first batch
// issued together at the top of the kernel and stored once the chain has something else in
// flight.  A pass covers 512 / (h / 4) whole rows of one matrix, thread -> (row in pass, float4
// in row) fixed for the kernel: one division per thread, no per-load address select (the first
// version computed (row, column, matrix) per load: 1,000 instructions of branches and divisions,
// 7 of the forward kernel's 17.7 us).
constexpr int kStageBatch = 11;      // passes per matrix in flight: 11 x 4 rows covers n = 41 at h = 256
struct WeightStage {
  float4 vl[kStageBatch], vr[kStageBatch];
  int r, q, rpp;                     // row within a pass, float4 within the row, rows per pass
  bool active;
  __device__ inline void init(int quads) {
    rpp = kLogitBlock / quads;
    if (rpp < 1) rpp = 1;            // (quads <= 256 by the entry point's width limit)
    r = threadIdx.x / quads;
    q = threadIdx.x - r * quads;
    active = r < rpp;
  }
  __device__ inline void load(int pass0, int n, const float* __restrict__ w_l, int64_t ld_wl,
                              const float* __restrict__ w_r, int64_t ld_wr) {
#pragma unroll
    for (int b = 0; b < kStageBatch; ++b) {
      int c = (pass0 + b) * rpp + r;
      if (c >= n) c = n - 1;         // clamped, not branched: the loads stay one batch
      vl[b] = *reinterpret_cast<const float4*>(w_l + c * ld_wl + q * 4);
      vr[b] = *reinterpret_cast<const float4*>(w_r + c * ld_wr + q * 4);
    }
  }
  __device__ inline void store(float* lds, int pass0, int n, int h) const {
#pragma unroll
    for (int b = 0; b < kStageBatch; ++b) {
      const int c = (pass0 + b) * rpp + r;
      if (active && c < n) {
        float* row = lds + static_cast<int64_t>(c) * 2 * h + q * 4;
        *reinterpret_cast<float4*>(row) = vl[b];
        *reinterpret_cast<float4*>(row + h) = vr[b];
      }
    }
  }
  // the batches after the first (n above 6 passes of rows)
  __device__ inline void rest(float* lds, int n, int h, const float* __restrict__ w_l, int64_t ld_wl,
                              const float* __restrict__ w_r, int64_t ld_wr) {
    for (int pass0 = kStageBatch; pass0 * rpp < n; pass0 += kStageBatch) {
      load(pass0, n, w_l, ld_wl, w_r, ld_wr);
      store(lds, pass0, n, h);
    }
  }
};

// NV = H / 256 (a lane holds NV float4 of a row: columns 4 (lane + 64 v) ..)
// PRE: hrows holds the previous layer's PRE-activation rows z; a row becomes relu(z / max(|z|, eps))
// in registers on its way in (one DPP reduction per row, 16 rows at a time), so the previous
// layer's normalise pass — 54 MB and a launch — is not run at all; the seeds' own activations
// are written to h_root for the root term's weight gradient.
// BWD: the seed's backward in the same wave, behind its forward (dz never leaves the lanes, the
// weights are staged once): dh[src[e]] += (dz . W_l) (/ deg), dh[t] += dz . W_r as in logit_bwd_body.
// MAX: the neighbour reduce is the element-wise maximum (first edge on ties, 0 for an empty segment:
// gather_reduce_fwd_kernel's rule), and the backward sends each column's gradient to the row that
// supplied it — one add per column instead of one row per edge.
// RED = 2 (gcn, layers/sage.py:114-117): the neighbour rows and ONE self-loop averaged, for targets
// that have neighbours at all; no root weight — z = agg . w_l^T, w_r is not read for the products
// — and the backward hands the target's own row the same share as every neighbour.
template <int NV, bool FULL, bool PRE, bool BWD = false, int RED = 0>      // FULL: h == 256 NV, no lane is ever out of the row
__global__ void __launch_bounds__(kLogitBlock)
logit_fwd_kernel(const float* __restrict__ hrows, int64_t ld_h, const int64_t* __restrict__ seg_ptr,
                 const int64_t* __restrict__ src, int64_t m, int h, int mean,
                 const float* __restrict__ w_l, int64_t ld_wl, const float* __restrict__ w_r, int64_t ld_wr,
                 int n, const int64_t* __restrict__ labels, const int64_t* __restrict__ label_rows,
                 float grad_scale, float* __restrict__ agg, int64_t ld_agg, float* __restrict__ y, int64_t ld_y,
                 float* __restrict__ inv_norm, float* __restrict__ dz, int64_t ld_dz,
                 float* __restrict__ row_loss, float* __restrict__ h_root, int64_t ld_h_root,
                 int64_t* __restrict__ pred, float* __restrict__ dh, int64_t ld_dh,
                 const int* __restrict__ writers, const float* __restrict__ prows, int64_t ld_p,
                 float* __restrict__ dnb, int64_t ld_dnb) {
  // prows != NULL (the pooling aggregators' last layer, layers/sage.py:201-210; never with PRE): the
  // NEIGHBOUR rows come from prows = relu(pool(h)) while the root row stays hrows[t], and the
  // backward sends the neighbours' share to dnb = the gradient of pool(h) — through the relu: a
  // row's element takes it where prows is positive (max: where the maximum is) — and the root's to dh.
  // Forward-only callers (sage_logit_layer_infer) pass labels = agg = dz = row_loss = NULL and
  // take y and / or pred (the arg-max class of each seed row); every one of these tests is
  // wave-uniform.
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  constexpr bool MAX = RED == 1, GCN = RED == 2;
  const bool pool = prows != nullptr;                  // (uniform)
  const float* const nbase = pool ? prows : hrows;     // where the neighbour rows are read
  const int64_t ld_n = pool ? ld_p : ld_h;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kLogitWaves + threadIdx.x / kWave;
  const bool live = t < m;      // wave-uniform; every wave takes part in the staging and the barrier
  const int nq = h / 4;         // float4 per row
  constexpr int kRowsInFlight = kRowRegs / NV;
  auto in_row = [&](int v) { return FULL || lane + kWave * v < nq; };
  auto sumsq = [&](const float4 (&r)[NV]) {       // this lane's share of |row|^2
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (in_row(v)) q += r[v].x * r[v].x + r[v].y * r[v].y + r[v].z * r[v].z + r[v].w * r[v].w;
    }
    return q;
  };
  auto activate = [&](float4 (&r)[NV], float ss) {      // relu(z / max(|z|, eps)), ss = |z|^2 over the wave
    // (times the reciprocal, not a division per element as the normalise pass does it: within an
    // ulp of that, and a quarter of this path's instructions)
    // v_rsq_f32 (1 ulp) and a clamp instead of an IEEE square root and a division: 1 / max(|z|, eps)
    const float inv = fminf(__builtin_amdgcn_rsqf(ss), 1.f / kNormEps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      r[v].x = fmaxf(r[v].x * inv, 0.f);
      r[v].y = fmaxf(r[v].y * inv, 0.f);
      r[v].z = fmaxf(r[v].z * inv, 0.f);
      r[v].w = fmaxf(r[v].w * inv, 0.f);
    }
  };

  LOGIT_STAMP(0, 0);
  float4 acc[NV], root[NV];
  int arg[MAX ? NV : 1][4];          // MAX: the edge (relative to s0) that supplied each maximum, -1 = none yet
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = root[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int v = 0; v < (MAX ? NV : 1); ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) arg[v][j] = -1;
  // requests in the order their results are needed; none waits for another except where noted
  const int64_t tc = live ? t : m - 1;                 // (dead waves of the last workgroup: clamped, results unused)
  const int64_t s0 = seg_ptr[tc];
  const int64_t len = seg_ptr[tc + 1] - s0;
  const int64_t lab_row = (labels != nullptr && label_rows != nullptr) ? label_rows[tc] : tc;
  {
    const float4* rp = reinterpret_cast<const float4*>(hrows + tc * ld_h);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (in_row(v)) root[v] = rp[lane + kWave * v];
    }
  }
  int64_t my0 = 0;
  if (lane < len) my0 = src[s0 + lane];                // needs s0: the chain's second round trip
  // BWD with `writers`: how many waves of this launch add into each row of dh (sage_count_row_writers:
  // its in-edges among the seeds' segments, + 1 for a seed's own row) — a row with ONE writer is
  // stored, not added (dh holds zeros): the scatter is bound by atomic throughput, 1.3 TB/s
  // against several for plain stores, and at the Reddit shape 9 of 10 rows have one writer
  int solo0 = 0, solo_root = 0;
  if constexpr (BWD) {
    if (writers != nullptr) {
      // (pool: the two sinks are different buffers — a seed's own row of dh has that seed as its one
      // writer, a row of dnb the row's in-edges: the count less the seed's own entry)
      if (lane < len) solo0 = writers[my0] - ((pool && my0 < m) ? 1 : 0) == 1 ? 1 : 0;           // needs my0
      solo_root = (pool || writers[tc] == 1) ? 1 : 0;
    }
  }
  WeightStage st;                                      // requested behind it, stored while the rows travel
  st.init(nq);
  st.load(0, n, w_l, ld_wl, w_r, ld_wr);
  const int64_t lab = labels != nullptr ? labels[lab_row] : -1;      // needs lab_row
  LOGIT_STAMP(0, 1);
  bool staged = false;
  if constexpr (PRE) {
    activate(root, wave_total(sumsq(root)));
    if (live && h_root != nullptr) {
      float4* hp = reinterpret_cast<float4*>(h_root + t * ld_h_root);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (in_row(v)) hp[lane + kWave * v] = root[v];
      }
    }
  }
  if (live) {
    for (int64_t base = 0; base < len; base += kWave) {
      const int chunk = (len - base) < kWave ? static_cast<int>(len - base) : kWave;
      int64_t my = my0;
      if (base > 0) {
        my = 0;
        if (lane < chunk) my = src[s0 + base + lane];
      }
      for (int r0 = 0; r0 < chunk; r0 += kRowsInFlight) {
        float4 x[kRowsInFlight][NV];
#pragma unroll
        for (int r = 0; r < kRowsInFlight; ++r) {
          if (r0 + r < chunk) {
            const float4* rowp = reinterpret_cast<const float4*>(nbase + readlane64(my, r0 + r) * ld_n);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              if (in_row(v)) x[r][v] = rowp[lane + kWave * v];
            }
          }
        }
        if (!staged) {       // (first round of rows requested: now the weights, which arrived before them)
          st.store(wlds, 0, n, h);
          staged = true;
        }
        if constexpr (PRE) {
          float ss[kRowsInFlight];
#pragma unroll
          for (int r = 0; r < kRowsInFlight; ++r) ss[r] = wave_total(r0 + r < chunk ? sumsq(x[r]) : 0.f);
#pragma unroll
          for (int r = 0; r < kRowsInFlight; ++r) {
            if (r0 + r < chunk) activate(x[r], ss[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < kRowsInFlight; ++r) {
          if (r0 + r < chunk) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              if (in_row(v)) {
                if constexpr (MAX) {
                  const int e = static_cast<int>(base) + r0 + r;
                  auto upd = [&](float& a, int& g, float val) {
                    if (g < 0 || val > a) {
                      a = val;
                      g = e;
                    }
                  };
                  upd(acc[v].x, arg[v][0], x[r][v].x);
                  upd(acc[v].y, arg[v][1], x[r][v].y);
                  upd(acc[v].z, arg[v][2], x[r][v].z);
                  upd(acc[v].w, arg[v][3], x[r][v].w);
                } else {
                  acc[v].x += x[r][v].x;
                  acc[v].y += x[r][v].y;
                  acc[v].z += x[r][v].z;
                  acc[v].w += x[r][v].w;
                }
              }
            }
          }
        }
      }
    }
    if (GCN && len > 0) {      // the self-loop: the target's own (activated) row, added last and counted
      const float cnt = static_cast<float>(len) + 1.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        acc[v].x = (acc[v].x + root[v].x) / cnt;
        acc[v].y = (acc[v].y + root[v].y) / cnt;
        acc[v].z = (acc[v].z + root[v].z) / cnt;
        acc[v].w = (acc[v].w + root[v].w) / cnt;
      }
    } else if (!GCN && mean) {
      const float cnt = len > 0 ? static_cast<float>(len) : 1.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        acc[v].x /= cnt;
        acc[v].y /= cnt;
        acc[v].z /= cnt;
        acc[v].w /= cnt;
      }
    }
    if (agg != nullptr) {
      float4* ap = reinterpret_cast<float4*>(agg + t * ld_agg);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (in_row(v)) ap[lane + kWave * v] = acc[v];
      }
    }
  }
  if (!staged) st.store(wlds, 0, n, h);      // dead waves and seeds without neighbours
  st.rest(wlds, n, h, w_l, ld_wl, w_r, ld_wr);
  LOGIT_STAMP(0, 2);
  LOGIT_STAMP(0, 3);
  __syncthreads();
  LOGIT_STAMP(0, 4);
  if (!live) return;

  // z[c] = <agg, W_l[c]> + <h[t], W_r[c]>: every lane forms its share of all 64 (n, zero-padded)
  // products, then ONE transposing butterfly sums them: at the step with distance o a lane keeps
  // the half of its values whose index has bit o set like its own lane id and adds the partner's
  // copy of that half — 32 + 16 + ... + 1 = 63 exchanges instead of 6 per output, and lane c ends
  // up with z[c].  (A butterfly per output costs 6 n dependent ds_bpermute: 9 of this kernel's
  // 22 us when it was written that way.)
  float p[64];
#pragma unroll
  for (int c0 = 0; c0 < 64; c0 += 8) {
    if (c0 < n) {       // wave-uniform; inside a group nothing is conditional, so its 16 LDS reads are in flight together
      float4 a[8][NV], b[8][NV];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j < n ? c0 + j : n - 1;
        const float4* wl = reinterpret_cast<const float4*>(wlds + static_cast<int64_t>(c) * 2 * h);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int col = in_row(v) ? lane + kWave * v : 0;
          a[j][v] = wl[col];
          b[j][v] = wl[nq + col];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          float qq = acc[v].x * a[j][v].x + acc[v].y * a[j][v].y + acc[v].z * a[j][v].z + acc[v].w * a[j][v].w;
          if (!GCN) qq += root[v].x * b[j][v].x + root[v].y * b[j][v].y + root[v].z * b[j][v].z + root[v].w * b[j][v].w;
          q += in_row(v) ? qq : 0.f;
        }
        p[c0 + j] = c0 + j < n ? q : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[c0 + j] = 0.f;
    }
  }
  // (on the VALU since the end of round 3: the 63 ds_bpermute exchanges of the butterfly queue on the
  // CU's one LDS pipe behind the other waves' — 4.5 us for 47 of them from eight waves in
  // logit_fb4_kernel.  Two transposing steps across the half-waves and the 16-lane rows
  // (v_permlane32_swap / v_permlane16_swap) leave 16 values per lane, DPP sums them over the row in
  // every lane of it, and lane l picks the one with index l & 15.)
#pragma unroll
  for (int j = 0; j < 32; ++j) p[j] = swap32_sum(p[j], p[j + 32]);
#pragma unroll
  for (int j = 0; j < 16; ++j) p[j] = swap16_sum(p[j], p[j + 16]);
#pragma unroll
  for (int j = 0; j < 16; ++j) p[j] = row16_total(p[j]);
  float zc = p[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) zc = (lane & 15) == j ? p[j] : zc;
  LOGIT_STAMP(0, 5);
  // normalise + cross entropy + d loss / d z on the n <= 64 values, one per lane (the formulas of
  // logits_xent_fused_kernel, rowops.hip)
  const bool has = lane < n;
  const float ss = wave_total(has ? zc * zc : 0.f);
  const float denom = fmaxf(sqrtf(ss), kNormEps);
  const float inv = 1.f / denom;
  const bool clamped = inv >= 1.f / kNormEps;
  const float yv = zc / denom;
  const float mx = wave_top(has ? yv : -INFINITY);
  if (pred != nullptr) {
    // torch.argmax's choice: the first class that attains the maximum
    const unsigned long long at_max = __ballot(has && yv == mx);
    if (lane == 0) pred[t] = at_max != 0ull ? static_cast<int64_t>(__builtin_ctzll(at_max)) : 0;
  }
  if (dz == nullptr) {       // forward only
    if (lane == 0 && inv_norm != nullptr) inv_norm[t] = inv;
    if (has && y != nullptr) y[t * ld_y + lane] = yv;
    return;
  }
  const float se = wave_total(has ? expf(yv - mx) : 0.f);
  const float lse = mx + logf(se);
  const float scale = grad_scale / static_cast<float>(m);
  float g = has ? expf(yv - lse) : 0.f;
  if (has && lane == lab) g -= 1.f;
  g *= scale;
  const float at_label = wave_total(has && lane == lab ? yv : 0.f);
  const float dot = clamped ? 0.f : wave_total(has ? yv * g : 0.f);
  if (lane == 0) {
    if (inv_norm != nullptr) inv_norm[t] = inv;
    row_loss[t] = (lab >= 0 && lab < n) ? lse - at_label : 0.f;
  }
  const float dzv = has ? (g - yv * dot) * inv : 0.f;
  if (has) {
    if (y != nullptr) y[t * ld_y + lane] = yv;
    dz[t * ld_dz + lane] = dzv;
  }
  LOGIT_STAMP(0, 6);
  if constexpr (BWD && MAX) {
    // in the forward's column layout (a lane holds columns 4 (lane + 64 v) .. + 3): the gradient of
    // column f goes to the one row that supplied the maximum — h adds per seed, not h per edge
    float4 da4[NV], dr4[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) da4[v] = dr4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = 0; c0 < n; c0 += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j < n ? c0 + j : n - 1;
        float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dzv), c));   // c is wave-uniform
        if (c0 + j >= n) d = 0.f;
        const float4* wl = reinterpret_cast<const float4*>(wlds + static_cast<int64_t>(c) * 2 * h);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int col = in_row(v) ? lane + kWave * v : 0;
          const float4 a = wl[col], b = wl[nq + col];
          da4[v].x += d * a.x; da4[v].y += d * a.y; da4[v].z += d * a.z; da4[v].w += d * a.w;
          dr4[v].x += d * b.x; dr4[v].y += d * b.y; dr4[v].z += d * b.z; dr4[v].w += d * b.w;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (!in_row(v)) continue;
      float* rootp = dh + t * ld_dh + 4 * (lane + kWave * v);
      const float dr[4] = {dr4[v].x, dr4[v].y, dr4[v].z, dr4[v].w};
      const float da[4] = {da4[v].x, da4[v].y, da4[v].z, da4[v].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (solo_root) rootp[j] = dr[j];      // (wave-uniform)
        else atomicAdd(rootp + j, dr[j]);
      }
      const float mxv[4] = {acc[v].x, acc[v].y, acc[v].z, acc[v].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (arg[v][j] >= 0) {
          const int64_t row = src[s0 + arg[v][j]];
          if (!pool) atomicAdd(dh + row * ld_dh + 4 * (lane + kWave * v) + j, da[j]);
          else if (mxv[j] > 0.f) atomicAdd(dnb + row * ld_dnb + 4 * (lane + kWave * v) + j, da[j]);
        }
      }
    }
    LOGIT_STAMP(0, 7);
  } else if constexpr (BWD) {
    // (lane l owns columns l + 64 k here: every atomic wave-instruction adds 256 contiguous bytes)
    constexpr int NC = 4 * NV;
    float da[NC], dr[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) da[k] = dr[k] = 0.f;
    for (int c0 = 0; c0 < n; c0 += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j < n ? c0 + j : n - 1;
        float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dzv), c));   // c is wave-uniform
        if (c0 + j >= n) d = 0.f;
        const float* wl = wlds + static_cast<int64_t>(c) * 2 * h;
        const float* wr = wl + h;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          if (FULL || lane + kWave * k < h) {
            da[k] += d * wl[lane + kWave * k];
            dr[k] += d * wr[lane + kWave * k];
          }
        }
      }
    }
    float* rootp = dh + t * ld_dh;
    if (GCN) {      // no root weight: the target's row takes the self-loop's share of d agg
      if (len <= 0) return;
      const float cnt = static_cast<float>(len) + 1.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) dr[k] = da[k] = da[k] / cnt;
    }
    if (solo_root) {      // (wave-uniform)
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (FULL || lane + kWave * k < h) rootp[lane + kWave * k] = dr[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (FULL || lane + kWave * k < h) atomicAdd(rootp + lane + kWave * k, dr[k]);
      }
    }
    if (len <= 0) return;
    if (!GCN && mean) {
      const float cnt = static_cast<float>(len);
#pragma unroll
      for (int k = 0; k < NC; ++k) da[k] /= cnt;
    }
    for (int64_t base = 0; base < len; base += kWave) {
      const int chunk = (len - base) < kWave ? static_cast<int>(len - base) : kWave;
      int64_t my = my0;
      int solo = solo0;
      if (base > 0) {
        my = 0;
        solo = 0;
        if (lane < chunk) {
          my = src[s0 + base + lane];
          if (writers != nullptr) solo = writers[my] - ((pool && my < m) ? 1 : 0) == 1 ? 1 : 0;
        }
      }
      if (pool) {
        // through the relu of the rows that were read: their values once more (out of L2), a batch
        // of rows requested together so that the segment is not one round trip per row
        constexpr int kBatch = 8 / NV;
        for (int r0 = 0; r0 < chunk; r0 += kBatch) {
          float pv[kBatch][NC];
#pragma unroll
          for (int i = 0; i < kBatch; ++i) {
            if (r0 + i < chunk) {
              const float* pr = nbase + readlane64(my, r0 + i) * ld_n;
#pragma unroll
              for (int k = 0; k < NC; ++k) {
                if (FULL || lane + kWave * k < h) pv[i][k] = pr[lane + kWave * k];
              }
            }
          }
#pragma unroll
          for (int i = 0; i < kBatch; ++i) {
            if (r0 + i < chunk) {
              float* rowp = dnb + readlane64(my, r0 + i) * ld_dnb;
              const bool one = __builtin_amdgcn_readlane(solo, r0 + i) != 0;      // (wave-uniform)
#pragma unroll
              for (int k = 0; k < NC; ++k) {
                if (FULL || lane + kWave * k < h) {
                  const float v = pv[i][k] > 0.f ? da[k] : 0.f;
                  if (one) rowp[lane + kWave * k] = v;
                  else if (v != 0.f) atomicAdd(rowp + lane + kWave * k, v);
                }
              }
            }
          }
        }
        continue;
      }
      for (int r = 0; r < chunk; ++r) {
        float* rowp = dh + readlane64(my, r) * ld_dh;
        if (__builtin_amdgcn_readlane(solo, r)) {      // (wave-uniform)
#pragma unroll
          for (int k = 0; k < NC; ++k) {
            if (FULL || lane + kWave * k < h) rowp[lane + kWave * k] = da[k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < NC; ++k) {
            if (FULL || lane + kWave * k < h) atomicAdd(rowp + lane + kWave * k, da[k]);
          }
        }
      }
    }
    LOGIT_STAMP(0, 7);
  }
}

// NC = H / 64 (a lane owns columns lane + 64 k)
// The training step's common case once more — H = 256, mean or sum, pre-activation rows, backward in
// the same launch — with FOUR waves per seed and two seeds per workgroup (512 threads, 256
// workgroups: one per CU).  The one-wave kernel above is a chain of latencies (25.6 us for 512
// seeds, cycle stamps of scratch/logit_prof.hip: rows 8.3 us, products 4.7, softmax 1, the
// backward's products 3, its scatter 7) on half of the CUs; here every link of the chain is a
// quarter as long: a wave gathers and normalises every fourth row of its seed's segment, forms the
// products of every fourth class (classes sub, sub + 4, ...) and the backward's partial d agg / d
// root over the same classes, and scatters every fourth row.  The partial aggregates, the logits
// and the partial row gradients cross the four waves through LDS (three barriers).
constexpr int kFb4Block = 512;
constexpr int kFb4H = 256;
constexpr int kFb4Extra = 2 * (2 * 4 * kFb4H) + 2 * 64;      // floats of LDS behind the weights
constexpr int kFb4ExtraMax = 2 * 4 * kFb4H + 2 * kFb4H;      // ... and behind those with MAX (edges: partial, final)

// GCN (layers/sage.py:114-117): the neighbour rows and ONE self-loop averaged for targets that have
// neighbours, no root weight (w_r is not read for the products), the target's own row takes the
// self-loop's share of d agg — as RED = 2 of logit_fwd_kernel.
// POOL (the pooling aggregators' last layer, mean; as `prows` of logit_fwd_kernel): rows as they are
// (no normalisation on the way in), the neighbour rows from prows = relu(pool(h)), their share of the
// backward through the relu into dnb, the root's into dh (stored: its one writer).
// MAX: the neighbour reduce is the element-wise maximum (first edge on ties, 0 for an empty segment);
// every wave keeps (maximum, edge) of its rows, the four pairs meet in LDS (equal maxima: the lower
// edge), and the backward sends each column's gradient to the one row that supplied it (POOL: where
// that maximum is positive) — wave `sub` takes columns lane + 64 sub.
template <bool GCN, bool POOL = false, bool MAX = false>
__global__ void __launch_bounds__(kFb4Block, 2)      // (two waves per SIMD: the register file is the wave's up to 256 VGPRs)
logit_fb4_kernel(const float* __restrict__ hrows, int64_t ld_h, const int64_t* __restrict__ seg_ptr,
                 const int64_t* __restrict__ src, int64_t m, int mean,
                 const float* __restrict__ w_l, int64_t ld_wl, const float* __restrict__ w_r, int64_t ld_wr,
                 int n, const int64_t* __restrict__ labels, const int64_t* __restrict__ label_rows,
                 float grad_scale, float* __restrict__ agg, int64_t ld_agg, float* __restrict__ y, int64_t ld_y,
                 float* __restrict__ inv_norm, float* __restrict__ dz, int64_t ld_dz,
                 float* __restrict__ row_loss, float* __restrict__ h_root, int64_t ld_h_root,
                 float* __restrict__ dh, int64_t ld_dh, const int* __restrict__ writers,
                 const float* __restrict__ prows, int64_t ld_p, float* __restrict__ dnb, int64_t ld_dnb) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  constexpr int H = kFb4H;
  const float* const nbase = POOL ? prows : hrows;
  const int64_t ld_n = POOL ? ld_p : ld_h;
  float* const nsink = POOL ? dnb : dh;
  const int64_t ld_ns = POOL ? ld_dnb : ld_dh;
  float* const part_a = wlds + static_cast<int64_t>(n) * 2 * H;      // [2 seeds][4 waves][H]
  float* const part_r = part_a + 2 * 4 * H;
  float* const zb = part_r + 2 * 4 * H;                               // [2 seeds][64]
  int* const part_g = reinterpret_cast<int*>(zb + 2 * 64);            // MAX: [2 seeds][4 waves][H] edges of the partial maxima
  int* const argc = part_g + 2 * 4 * H;                               // MAX: [2 seeds][H] the edge each column's gradient goes to
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave);
  const int sd = wave >> 2, sub = wave & 3;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 2 + sd;
  const bool live = t < m;      // wave-uniform; every wave takes part in the staging and the barriers
  auto sumsq = [](const float4& r) { return r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w; };
  auto activate = [](float4& r, float ss) {      // relu(z / max(|z|, eps)) as in logit_fwd_kernel
    const float inv = fminf(__builtin_amdgcn_rsqf(ss), 1.f / kNormEps);
    r.x = fmaxf(r.x * inv, 0.f);
    r.y = fmaxf(r.y * inv, 0.f);
    r.z = fmaxf(r.z * inv, 0.f);
    r.w = fmaxf(r.w * inv, 0.f);
  };
  LOGIT_STAMP(0, 0);
  const int64_t tc = live ? t : m - 1;
  const int64_t s0 = seg_ptr[tc];
  const int64_t len = seg_ptr[tc + 1] - s0;
  const int64_t lab_row = label_rows != nullptr ? label_rows[tc] : tc;
  float4 root = reinterpret_cast<const float4*>(hrows + tc * ld_h)[lane];
  int64_t my0 = 0;
  if (lane < len) my0 = src[s0 + lane];
  int solo0 = 0, solo_root = 0;
  if (writers != nullptr) {
    if (lane < len) solo0 = writers[my0] - ((POOL && my0 < m) ? 1 : 0) == 1 ? 1 : 0;
    solo_root = (POOL || writers[tc] == 1) ? 1 : 0;
  }
  // the weights: thread -> (row `wave` of a pass of 8 rows, float4 `lane` of the row); six passes
  // (48 rows of each matrix) are requested here and stored once the rows are on their way
  constexpr int kPasses = 6;
  float4 vl[kPasses], vr[kPasses];
#pragma unroll
  for (int b = 0; b < kPasses; ++b) {
    int c = b * 8 + wave;
    if (c >= n) c = n - 1;
    vl[b] = *reinterpret_cast<const float4*>(w_l + c * ld_wl + lane * 4);
    vr[b] = *reinterpret_cast<const float4*>(w_r + c * ld_wr + lane * 4);
  }
  LOGIT_STAMP(0, 1);
  const int64_t lab = labels[lab_row];
  if (!POOL) {
    activate(root, wave_total(sumsq(root)));
    if (live && sub == 0) reinterpret_cast<float4*>(h_root + t * ld_h_root)[lane] = root;
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // rows r0, r0 + 4, ... of a chunk of 64 edges, eight in flight at once (a fan-out of up to 32 in one round)
  auto load_rows = [&](float4 (&x)[8], int64_t my, int r0, int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + 4 * i;
      if (r < chunk) x[i] = reinterpret_cast<const float4*>(nbase + readlane64(my, r) * ld_n)[lane];
    }
  };
  int arg[4] = {-1, -1, -1, -1};      // MAX: the edge (relative to s0) behind each of this wave's maxima
  auto add_rows = [&](float4 (&x)[8], int r0, int chunk, int base) __attribute__((always_inline)) {
    float ss[8];
    if (!POOL) {
#pragma unroll
      for (int i = 0; i < 8; ++i) ss[i] = r0 + 4 * i < chunk ? wave_total(sumsq(x[i])) : 0.f;      // (wave-uniform test)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (r0 + 4 * i < chunk) {
        if (!POOL) activate(x[i], ss[i]);
        if (MAX) {
          const int e = base + r0 + 4 * i;
          auto upd = [&](float& a, int& g, float val) {
            if (g < 0 || val > a) {
              a = val;
              g = e;
            }
          };
          upd(acc.x, arg[0], x[i].x);
          upd(acc.y, arg[1], x[i].y);
          upd(acc.z, arg[2], x[i].z);
          upd(acc.w, arg[3], x[i].w);
        } else {
          acc.x += x[i].x;
          acc.y += x[i].y;
          acc.z += x[i].z;
          acc.w += x[i].w;
        }
      }
    }
  };
  float4 x[8];
  const int chunk0 = !live ? 0 : (len < kWave ? static_cast<int>(len) : kWave);
  load_rows(x, my0, sub, chunk0);
  // (the first round of rows is requested: now the weights, which arrived before them)
#pragma unroll
  for (int b = 0; b < kPasses; ++b) {
    const int c = b * 8 + wave;
    if (c < n) {
      float* row = wlds + static_cast<int64_t>(c) * 2 * H + lane * 4;
      *reinterpret_cast<float4*>(row) = vl[b];
      *reinterpret_cast<float4*>(row + H) = vr[b];
    }
  }
  add_rows(x, sub, chunk0, 0);
  for (int r0 = sub + 32; r0 < chunk0; r0 += 32) {
    load_rows(x, my0, r0, chunk0);
    add_rows(x, r0, chunk0, 0);
  }
  if (live) {
    for (int64_t base = kWave; base < len; base += kWave) {
      const int chunk = (len - base) < kWave ? static_cast<int>(len - base) : kWave;
      int64_t my = 0;
      if (lane < chunk) my = src[s0 + base + lane];
      for (int r0 = sub; r0 < chunk; r0 += 32) {
        load_rows(x, my, r0, chunk);
        add_rows(x, r0, chunk, static_cast<int>(base));
      }
    }
  }
  for (int c0 = kPasses * 8; c0 < n; c0 += 8) {      // classes 48 .. 63
    const int c = c0 + wave;
    if (c < n) {
      float* row = wlds + static_cast<int64_t>(c) * 2 * H + lane * 4;
      *reinterpret_cast<float4*>(row) = *reinterpret_cast<const float4*>(w_l + c * ld_wl + lane * 4);
      *reinterpret_cast<float4*>(row + H) = *reinterpret_cast<const float4*>(w_r + c * ld_wr + lane * 4);
    }
  }
  LOGIT_STAMP(0, 2);
  *reinterpret_cast<float4*>(part_a + (sd * 4 + sub) * H + 4 * lane) = acc;
  if (MAX) *reinterpret_cast<int4*>(part_g + (sd * 4 + sub) * H + 4 * lane) = make_int4(arg[0], arg[1], arg[2], arg[3]);
  __syncthreads();

  LOGIT_STAMP(0, 3);
  // the seed's aggregate (every wave of the seed adds the four parts up for itself)
  float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MAX) {
    int4 g4 = make_int4(-1, -1, -1, -1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 q = *reinterpret_cast<const float4*>(part_a + (sd * 4 + s) * H + 4 * lane);
      const int4 g = *reinterpret_cast<const int4*>(part_g + (sd * 4 + s) * H + 4 * lane);
      auto pick = [](float& bv, int& bg, float v, int e) {
        if (e >= 0 && (bg < 0 || v > bv || (v == bv && e < bg))) {
          bv = v;
          bg = e;
        }
      };
      pick(a4.x, g4.x, q.x, g.x);
      pick(a4.y, g4.y, q.y, g.y);
      pick(a4.z, g4.z, q.z, g.z);
      pick(a4.w, g4.w, q.w, g.w);
    }
    if (sub == 0) {      // where each column's gradient goes (POOL: only through a positive maximum)
      if (POOL) {
        if (!(a4.x > 0.f)) g4.x = -1;
        if (!(a4.y > 0.f)) g4.y = -1;
        if (!(a4.z > 0.f)) g4.z = -1;
        if (!(a4.w > 0.f)) g4.w = -1;
      }
      *reinterpret_cast<int4*>(argc + sd * H + 4 * lane) = g4;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 q = *reinterpret_cast<const float4*>(part_a + (sd * 4 + s) * H + 4 * lane);
      a4.x += q.x;
      a4.y += q.y;
      a4.z += q.z;
      a4.w += q.w;
    }
  }
  if (GCN) {
    if (len > 0) {
      const float cnt = static_cast<float>(len) + 1.f;
      a4.x = (a4.x + root.x) / cnt;
      a4.y = (a4.y + root.y) / cnt;
      a4.z = (a4.z + root.z) / cnt;
      a4.w = (a4.w + root.w) / cnt;
    }
  } else if (!MAX && mean) {
    const float cnt = len > 0 ? static_cast<float>(len) : 1.f;
    a4.x /= cnt;
    a4.y /= cnt;
    a4.z /= cnt;
    a4.w /= cnt;
  }
  if (live && sub == 0) reinterpret_cast<float4*>(agg + t * ld_agg)[lane] = a4;
  // z[c] for the classes c = sub + 4 j: every lane's share of the 16 products, summed over the 64
  // lanes below
  float p[16];
#pragma unroll
  for (int j0 = 0; j0 < 16; j0 += 8) {
    float4 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int c = sub + 4 * (j0 + j);
      if (c >= n) c = n - 1;
      const float4* wl = reinterpret_cast<const float4*>(wlds + static_cast<int64_t>(c) * 2 * H);
      a[j] = wl[lane];
      if (!GCN) b[j] = wl[H / 4 + lane];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float q = a4.x * a[j].x + a4.y * a[j].y + a4.z * a[j].z + a4.w * a[j].w;
      if (!GCN) q += root.x * b[j].x + root.y * b[j].y + root.z * b[j].z + root.w * b[j].w;
      p[j0 + j] = sub + 4 * (j0 + j) < n ? q : 0.f;
    }
  }
  LOGIT_STAMP(0, 8);
  // 16 values x 64 lanes -> 16 totals, on the VALU (47 ds_bpermute exchanges from eight waves queued
  // on the CU's one LDS pipe for 4.5 us): two transposing steps across the half-waves and the
  // 16-lane rows leave 4 values per lane (index 8 [lane & 32] + 4 [lane & 16] + j), which are summed
  // over the row's 16 lanes by DPP
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = swap32_sum(p[j], p[j + 8]);
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = swap16_sum(p[j], p[j + 4]);
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = row16_total(p[j]);
  LOGIT_STAMP(0, 9);
  if ((lane & 15) == 0) {
    const int v0 = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) zb[sd * 64 + sub + 4 * (v0 + j)] = p[j];
  }
  __syncthreads();

  LOGIT_STAMP(0, 4);
  // normalise + cross entropy + d loss / d z, one class per lane (every wave of the seed for itself)
  const bool has = lane < n;
  const float zc = has ? zb[sd * 64 + lane] : 0.f;
  const float ssz = wave_total(has ? zc * zc : 0.f);
  const float denom = fmaxf(sqrtf(ssz), kNormEps);
  const float inv = 1.f / denom;
  const bool clamped = inv >= 1.f / kNormEps;
  const float yv = zc / denom;
  const float mx = wave_top(has ? yv : -INFINITY);
  const float se = wave_total(has ? expf(yv - mx) : 0.f);
  const float lse = mx + logf(se);
  const float scale = grad_scale / static_cast<float>(m);
  float g = has ? expf(yv - lse) : 0.f;
  if (has && lane == lab) g -= 1.f;
  g *= scale;
  const float at_label = wave_total(has && lane == lab ? yv : 0.f);
  const float dot = clamped ? 0.f : wave_total(has ? yv * g : 0.f);
  const float dzv = has ? (g - yv * dot) * inv : 0.f;
  if (live && sub == 0) {
    if (lane == 0) {
      if (inv_norm != nullptr) inv_norm[t] = inv;
      row_loss[t] = (lab >= 0 && lab < n) ? lse - at_label : 0.f;
    }
    if (has) {
      if (y != nullptr) y[t * ld_y + lane] = yv;
      dz[t * ld_dz + lane] = dzv;
    }
  }

  LOGIT_STAMP(0, 5);
  // the backward's products over this wave's classes: lane l owns columns l + 64 k
  float da[4], dr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) da[k] = dr[k] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = sub + 4 * j;
    if (c < n) {      // wave-uniform
      const float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dzv), c));
      const float* wl = wlds + static_cast<int64_t>(c) * 2 * H;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        da[k] += d * wl[lane + kWave * k];
        if (!GCN) dr[k] += d * wl[H + lane + kWave * k];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    part_a[(sd * 4 + sub) * H + lane + kWave * k] = da[k];
    part_r[(sd * 4 + sub) * H + lane + kWave * k] = dr[k];
  }
  __syncthreads();
  LOGIT_STAMP(0, 6);
  if (!live) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    da[k] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) da[k] += part_a[(sd * 4 + s) * H + lane + kWave * k];
  }
  if (GCN) {
    if (len <= 0) return;
    const float cnt = static_cast<float>(len) + 1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) da[k] /= cnt;
  }
  if (sub == 0) {
    float* rootp = dh + t * ld_dh;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = da[k];      // gcn: the self-loop's share
      if (!GCN) {
        v = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) v += part_r[(sd * 4 + s) * H + lane + kWave * k];
      }
      if (solo_root) rootp[lane + kWave * k] = v;      // (wave-uniform)
      else atomicAdd(rootp + lane + kWave * k, v);
    }
  }
  if (len <= 0) return;
  if (MAX) {      // one add per column: this wave takes columns lane + 64 sub
    const int g = argc[sd * H + lane + kWave * sub];
    const float dv = sub == 0 ? da[0] : (sub == 1 ? da[1] : (sub == 2 ? da[2] : da[3]));      // (wave-uniform selects)
    if (g >= 0) atomicAdd(nsink + src[s0 + g] * ld_ns + lane + kWave * sub, dv);
    LOGIT_STAMP(0, 7);
    return;
  }
  if (!GCN && mean) {
    const float cnt = static_cast<float>(len);
#pragma unroll
    for (int k = 0; k < 4; ++k) da[k] /= cnt;
  }
  for (int64_t base = 0; base < len; base += kWave) {
    const int chunk = (len - base) < kWave ? static_cast<int>(len - base) : kWave;
    int64_t my = my0;
    int solo = solo0;
    if (base > 0) {
      my = 0;
      solo = 0;
      if (lane < chunk) {
        my = src[s0 + base + lane];
        if (writers != nullptr) solo = writers[my] - ((POOL && my < m) ? 1 : 0) == 1 ? 1 : 0;
      }
    }
    if (POOL) {
      // through the relu of the rows that were read: their values once more (out of L2), eight rows of
      // this wave's share requested together
      for (int r0 = sub; r0 < chunk; r0 += 32) {
        float pv[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (r0 + 4 * i < chunk) {
            const float* pr = nbase + readlane64(my, r0 + 4 * i) * ld_n;
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[i][k] = pr[lane + kWave * k];
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (r0 + 4 * i < chunk) {
            float* rowp = nsink + readlane64(my, r0 + 4 * i) * ld_ns;
            const bool one = __builtin_amdgcn_readlane(solo, r0 + 4 * i) != 0;      // (wave-uniform)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float v = pv[i][k] > 0.f ? da[k] : 0.f;
              if (one) rowp[lane + kWave * k] = v;
              else if (v != 0.f) atomicAdd(rowp + lane + kWave * k, v);
            }
          }
        }
      }
      continue;
    }
    for (int r = sub; r < chunk; r += 4) {
      float* rowp = dh + readlane64(my, r) * ld_dh;
      if (__builtin_amdgcn_readlane(solo, r)) {      // (wave-uniform)
#pragma unroll
        for (int k = 0; k < 4; ++k) rowp[lane + kWave * k] = da[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(rowp + lane + kWave * k, da[k]);
      }
    }
  }
  LOGIT_STAMP(0, 7);
}

template <int NC, bool FULL>      // FULL: h == 64 NC
__device__ __forceinline__ void
logit_bwd_body(const float* __restrict__ dz, int64_t ld_dz, const int64_t* __restrict__ seg_ptr,
               const int64_t* __restrict__ src, int64_t m, int h, int mean,
               const float* __restrict__ w_l, int64_t ld_wl, const float* __restrict__ w_r, int64_t ld_wr,
               int n, float* __restrict__ dh, int64_t ld_dh, const float* __restrict__ row_loss,
               float* __restrict__ loss) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kLogitWaves + wave;
  const bool live = t < m;
  LOGIT_STAMP(1, 0);
  const int64_t tc = live ? t : m - 1;
  const int64_t s0 = seg_ptr[tc];
  const int64_t len = seg_ptr[tc + 1] - s0;
  float dzc = 0.f;
  if (lane < n) dzc = dz[tc * ld_dz + lane];
  int64_t my0 = 0;
  if (lane < len) my0 = src[s0 + lane];                // needs s0
  WeightStage st;
  st.init(h / 4);
  st.load(0, n, w_l, ld_wl, w_r, ld_wr);
  st.store(wlds, 0, n, h);
  st.rest(wlds, n, h, w_l, ld_wl, w_r, ld_wr);
  if (loss != nullptr && blockIdx.x == 0 && wave == 0) {
    // loss = mean(row_loss[0 : m]): lane l adds rows l, l + 64, ... in order, then the butterfly
    float s = 0.f;
    for (int64_t i = lane; i < m; i += kWave) s += row_loss[i];
    s = wave_sum(s);
    if (lane == 0) loss[0] = s / static_cast<float>(m);
  }
  LOGIT_STAMP(1, 1);
  __syncthreads();
  LOGIT_STAMP(1, 2);
  if (!live) return;
  float da[NC], dr[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) da[k] = dr[k] = 0.f;
  for (int c0 = 0; c0 < n; c0 += 8) {      // eight outputs at a time: their LDS reads are in flight together
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j < n ? c0 + j : n - 1;
      float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dzc), c));   // c is wave-uniform
      if (c0 + j >= n) d = 0.f;
      const float* wl = wlds + static_cast<int64_t>(c) * 2 * h;
      const float* wr = wl + h;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (FULL || lane + kWave * k < h) {
          da[k] += d * wl[lane + kWave * k];
          dr[k] += d * wr[lane + kWave * k];
        }
      }
    }
  }
  LOGIT_STAMP(1, 3);
  float* rootp = dh + t * ld_dh;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    if (FULL || lane + kWave * k < h) atomicAdd(rootp + lane + kWave * k, dr[k]);
  }
  if (len <= 0) return;
  if (mean) {
    const float cnt = static_cast<float>(len);
#pragma unroll
    for (int k = 0; k < NC; ++k) da[k] /= cnt;
  }
  for (int64_t base = 0; base < len; base += kWave) {
    const int chunk = (len - base) < kWave ? static_cast<int>(len - base) : kWave;
    int64_t my = my0;
    if (base > 0) {
      my = 0;
      if (lane < chunk) my = src[s0 + base + lane];
    }
    for (int r = 0; r < chunk; ++r) {
      float* rowp = dh + readlane64(my, r) * ld_dh;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        if (FULL || lane + kWave * k < h) atomicAdd(rowp + lane + kWave * k, da[k]);
      }
    }
  }
  LOGIT_STAMP(1, 4);
}

template <int NC, bool FULL>
__global__ void __launch_bounds__(kLogitBlock)
logit_bwd_kernel(const float* __restrict__ dz, int64_t ld_dz, const int64_t* __restrict__ seg_ptr,
                 const int64_t* __restrict__ src, int64_t m, int h, int mean,
                 const float* __restrict__ w_l, int64_t ld_wl, const float* __restrict__ w_r, int64_t ld_wr,
                 int n, float* __restrict__ dh, int64_t ld_dh, const float* __restrict__ row_loss,
                 float* __restrict__ loss) {
  logit_bwd_body<NC, FULL>(dz, ld_dz, seg_ptr, src, m, h, mean, w_l, ld_wl, w_r, ld_wr, n, dh, ld_dh, row_loss, loss);
}

// The same launch with the layer's two weight gradients riding along: workgroups 0 .. seed_blocks-1
// are the seeds' (above), the ones behind them run skinny_dweight_body (dW_l += dz^T agg,
// dW_r += dz^T h_root) — 128 seed workgroups leave half of the CUs idle, and both halves read
// nothing the other writes.  (The weights' LDS image is allocated for every workgroup of the
// launch: one workgroup per CU.)
struct LogitDwArgs {
  SkinnyDwProblem p0, p1;
  int slices, m_chunk, gx, gy;
  const int64_t* m_dev;
};
template <int NC, bool FULL>
__global__ void __launch_bounds__(kLogitBlock)
logit_bwd_dw_kernel(const float* __restrict__ dz, int64_t ld_dz, const int64_t* __restrict__ seg_ptr,
                    const int64_t* __restrict__ src, int64_t m, int h, int mean,
                    const float* __restrict__ w_l, int64_t ld_wl, const float* __restrict__ w_r, int64_t ld_wr,
                    int n, float* __restrict__ dh, int64_t ld_dh, const float* __restrict__ row_loss,
                    float* __restrict__ loss, int seed_blocks, LogitDwArgs d) {
  static_assert(kLogitBlock == kSkinnyThreads, "skinny_dweight_body runs on 256 threads");
  if (static_cast<int>(blockIdx.x) >= seed_blocks) {      // (uniform over the workgroup)
    const int id = blockIdx.x - seed_blocks;
    skinny_dweight_body(id % d.gx, id / d.gx % d.gy, id / (d.gx * d.gy), dz, ld_dz, d.p0, d.p1, d.slices, m, d.m_dev, n,
                        d.m_chunk, nullptr);
    return;
  }
  logit_bwd_body<NC, FULL>(dz, ld_dz, seg_ptr, src, m, h, mean, w_l, ld_wl, w_r, ld_wr, n, dh, ld_dh, row_loss, loss);
}

inline bool vec16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_common(const char* who, sage_i64 m, int h, int n, int reduce, const float* w_l, sage_i64 ld_wl,
                 const float* w_r, sage_i64 ld_wr, size_t* lds_bytes, bool max_ok = false) {
  SAGE_REQUIRE(reduce == SAGE_REDUCE_MEAN || reduce == SAGE_REDUCE_SUM ||
                   (max_ok && (reduce == SAGE_REDUCE_MAX || reduce == SAGE_REDUCE_GCN)),
               SAGE_EINVAL, "%s: reduce must be mean or sum%s", who,
               max_ok ? ", max or gcn"
                      : " (max / gcn: the one-launch forward + backward, sage_logit_layer_fwd_bwd, or the forward-only call)");
  SAGE_REQUIRE(m > 0 && n > 0 && h > 0, SAGE_EINVAL, "%s: bad shape (m=%lld h=%d n=%d)", who, (long long)m, h, n);
  SAGE_REQUIRE(n <= kWave && h % 4 == 0 && h <= 1024, SAGE_ELIMIT,
               "%s: needs at most 64 outputs and a width that is a multiple of 4, at most 1024 (n=%d h=%d)", who, n, h);
  SAGE_REQUIRE(w_l && w_r && ld_wl >= h && ld_wr >= h && ld_wl % 4 == 0 && ld_wr % 4 == 0 && vec16(w_l) && vec16(w_r),
               SAGE_EINVAL, "%s: weights must be 16-byte aligned rows of at least h floats", who);
  *lds_bytes = static_cast<size_t>(n) * 2 * h * sizeof(float);
  SAGE_REQUIRE(*lds_bytes <= 160 * 1024, SAGE_ELIMIT, "%s: the [n, 2h] weights (%zu bytes) do not fit the 160 KB of LDS", who,
               *lds_bytes);
  return SAGE_OK;
}

template <typename K>
int allow_lds(K kernel, size_t bytes, const char* who) {
  // above 64 KB of dynamic LDS a kernel has to opt in, once per kernel and device (the attribute
  // is kept at the device's maximum, so the size of a later call never exceeds it).  Keyed by
  // the kernel's ADDRESS: every instantiation of one kernel template has the same type K.
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  if (bytes <= 64 * 1024) return SAGE_OK;
  int dev = 0;
  SAGE_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0, SAGE_ELAUNCH, "%s: cannot query the device", who);
  const std::pair<const void*, int> key(reinterpret_cast<const void*>(kernel), dev);
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& d : done) {
    if (d == key) return SAGE_OK;
  }
  // (static LDS of the kernel — the weight-gradient workgroups' 2 KB stage — counts against the same 160 KB)
  hipFuncAttributes attr;
  SAGE_REQUIRE(hipFuncGetAttributes(&attr, key.first) == hipSuccess, SAGE_ELAUNCH, "%s: cannot query the kernel", who);
  const int max_dyn = 160 * 1024 - static_cast<int>(attr.sharedSizeBytes);
  SAGE_REQUIRE(static_cast<long long>(bytes) <= max_dyn, SAGE_ELIMIT, "%s: %zu bytes of weights beside %zu bytes of static LDS",
               who, bytes, static_cast<size_t>(attr.sharedSizeBytes));
  SAGE_REQUIRE(hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn) == hipSuccess,
               SAGE_ELAUNCH, "%s: cannot raise the dynamic LDS limit", who);
  done.push_back(key);
  return SAGE_OK;
}

}  // namespace
}  // namespace sage

using namespace sage;

namespace {
int launch_logit_fwd(const char* who, bool pre, const float* h_rows, sage_i64 ld_h, const sage_i64* seg_ptr, const sage_i64* src,
                     sage_i64 m, int h, int reduce, const float* w_l, sage_i64 ld_wl, const float* w_r, sage_i64 ld_wr, int n,
                     const sage_i64* labels, const sage_i64* label_rows, float grad_scale, float* agg, sage_i64 ld_agg,
                     float* y, sage_i64 ld_y, float* inv_norm, float* dz, sage_i64 ld_dz, float* row_loss, float* h_root,
                     sage_i64 ld_h_root, sage_i64* pred, float* dh, sage_i64 ld_dh, const int* writers, hipStream_t stream,
                     const float* prows = nullptr, sage_i64 ld_p = 0, float* dnb = nullptr, sage_i64 ld_dnb = 0) {
  size_t lds = 0;
  // (max: its backward needs the arg-max of every column, which only lives in the forward's registers)
  const int rc = check_common(who, m, h, n, reduce, w_l, ld_wl, w_r, ld_wr, &lds, dz == nullptr || dh != nullptr);
  if (rc != SAGE_OK) return rc;
  const unsigned grid = static_cast<unsigned>(cdiv(m, kLogitWaves));
  const int mean = reduce == SAGE_REDUCE_MEAN ? 1 : 0;
  // the training step's common case: four waves per seed (SAGE_LOGIT_ONE_WAVE=1: the general kernel)
  static const bool one_wave = getenv("SAGE_LOGIT_ONE_WAVE") != nullptr;
  const size_t lds4 = lds + kFb4Extra * sizeof(float);
  const size_t lds4m = lds4 + kFb4ExtraMax * sizeof(float);
  if (!one_wave && prows == nullptr && dh != nullptr && pre && h == kFb4H &&
      (reduce == SAGE_REDUCE_MEAN || reduce == SAGE_REDUCE_SUM || reduce == SAGE_REDUCE_GCN ||
       (reduce == SAGE_REDUCE_MAX && lds4m <= 160 * 1024)) &&
      pred == nullptr && dz != nullptr && agg != nullptr && labels != nullptr && row_loss != nullptr && h_root != nullptr &&
      lds4 <= 160 * 1024) {
    const unsigned grid4 = static_cast<unsigned>(cdiv(m, 2));
    if (reduce == SAGE_REDUCE_MAX) {
      const int rc4 = allow_lds(logit_fb4_kernel<false, false, true>, lds4m, who);
      if (rc4 != SAGE_OK) return rc4;
      logit_fb4_kernel<false, false, true><<<grid4, kFb4Block, lds4m, stream>>>(
          h_rows, ld_h, seg_ptr, src, m, mean, w_l, ld_wl, w_r, ld_wr, n, labels, label_rows, grad_scale, agg, ld_agg, y, ld_y,
          inv_norm, dz, ld_dz, row_loss, h_root, ld_h_root, dh, ld_dh, writers, nullptr, 0, nullptr, 0);
    } else if (reduce == SAGE_REDUCE_GCN) {
      const int rc4 = allow_lds(logit_fb4_kernel<true>, lds4, who);
      if (rc4 != SAGE_OK) return rc4;
      logit_fb4_kernel<true><<<grid4, kFb4Block, lds4, stream>>>(
          h_rows, ld_h, seg_ptr, src, m, mean, w_l, ld_wl, w_r, ld_wr, n, labels, label_rows, grad_scale, agg, ld_agg, y, ld_y,
          inv_norm, dz, ld_dz, row_loss, h_root, ld_h_root, dh, ld_dh, writers, nullptr, 0, nullptr, 0);
    } else {
      const int rc4 = allow_lds(logit_fb4_kernel<false>, lds4, who);
      if (rc4 != SAGE_OK) return rc4;
      logit_fb4_kernel<false><<<grid4, kFb4Block, lds4, stream>>>(
          h_rows, ld_h, seg_ptr, src, m, mean, w_l, ld_wl, w_r, ld_wr, n, labels, label_rows, grad_scale, agg, ld_agg, y, ld_y,
          inv_norm, dz, ld_dz, row_loss, h_root, ld_h_root, dh, ld_dh, writers, nullptr, 0, nullptr, 0);
    }
    SAGE_CHECK_LAUNCH("logit_fb4_kernel");
    return SAGE_OK;
  }
  // ... and the pooling aggregators' last layer with the mean (sage_logit_layer_pool_fwd_bwd)
  if (!one_wave && prows != nullptr && dnb != nullptr && dh != nullptr && !pre && h == kFb4H &&
      (reduce == SAGE_REDUCE_MEAN || (reduce == SAGE_REDUCE_MAX && lds4m <= 160 * 1024)) && pred == nullptr && dz != nullptr &&
      agg != nullptr && labels != nullptr && row_loss != nullptr && lds4 <= 160 * 1024) {
    const unsigned grid4 = static_cast<unsigned>(cdiv(m, 2));
    if (reduce == SAGE_REDUCE_MAX) {
      const int rc4 = allow_lds(logit_fb4_kernel<false, true, true>, lds4m, who);
      if (rc4 != SAGE_OK) return rc4;
      logit_fb4_kernel<false, true, true><<<grid4, kFb4Block, lds4m, stream>>>(
          h_rows, ld_h, seg_ptr, src, m, mean, w_l, ld_wl, w_r, ld_wr, n, labels, label_rows, grad_scale, agg, ld_agg, y, ld_y,
          inv_norm, dz, ld_dz, row_loss, nullptr, 0, dh, ld_dh, writers, prows, ld_p, dnb, ld_dnb);
    } else {
      const int rc4 = allow_lds(logit_fb4_kernel<false, true, false>, lds4, who);
      if (rc4 != SAGE_OK) return rc4;
      logit_fb4_kernel<false, true, false><<<grid4, kFb4Block, lds4, stream>>>(
          h_rows, ld_h, seg_ptr, src, m, mean, w_l, ld_wl, w_r, ld_wr, n, labels, label_rows, grad_scale, agg, ld_agg, y, ld_y,
          inv_norm, dz, ld_dz, row_loss, nullptr, 0, dh, ld_dh, writers, prows, ld_p, dnb, ld_dnb);
    }
    SAGE_CHECK_LAUNCH("logit_fb4_kernel(pool)");
    return SAGE_OK;
  }
#define SAGE_LOGIT_FWD_Q(NV, FULL, PRE, BWD, RED)                                                               \
  do {                                                                                                      \
    const int rc2 = allow_lds(logit_fwd_kernel<NV, FULL, PRE, BWD, RED>, lds, who);                         \
    if (rc2 != SAGE_OK) return rc2;                                                                         \
    logit_fwd_kernel<NV, FULL, PRE, BWD, RED><<<grid, kLogitBlock, lds, stream>>>(                          \
        h_rows, ld_h, seg_ptr, src, m, h, mean, w_l, ld_wl, w_r, ld_wr, n, labels, label_rows, grad_scale, agg, ld_agg, y, \
        ld_y, inv_norm, dz, ld_dz, row_loss, h_root, ld_h_root, pred, dh, ld_dh, writers, prows, ld_p, dnb, ld_dnb); \
  } while (0)
#define SAGE_LOGIT_FWD_P(NV, FULL, PRE, BWD)                                                                    \
  do {                                                                                                      \
    if (reduce == SAGE_REDUCE_MAX) SAGE_LOGIT_FWD_Q(NV, FULL, PRE, BWD, 1);                                 \
    else if (reduce == SAGE_REDUCE_GCN) SAGE_LOGIT_FWD_Q(NV, FULL, PRE, BWD, 2);                            \
    else SAGE_LOGIT_FWD_Q(NV, FULL, PRE, BWD, 0);                                                           \
  } while (0)
#define SAGE_LOGIT_FWD(NV, FULL)                                                                                \
  do {                                                                                                      \
    if (dh != nullptr) {                                                                                    \
      if (pre) SAGE_LOGIT_FWD_P(NV, FULL, true, true);                                                      \
      else SAGE_LOGIT_FWD_P(NV, FULL, false, true);                                                         \
    } else {                                                                                                \
      if (pre) SAGE_LOGIT_FWD_P(NV, FULL, true, false);                                                     \
      else SAGE_LOGIT_FWD_P(NV, FULL, false, false);                                                        \
    }                                                                                                       \
  } while (0)
  const int nv = (h / 4 + kWave - 1) / kWave;
  const bool full = h == nv * 4 * kWave;
  if (nv == 1 && full) SAGE_LOGIT_FWD(1, true);
  else if (nv == 1) SAGE_LOGIT_FWD(1, false);
  else if (nv == 2 && full) SAGE_LOGIT_FWD(2, true);
  else if (nv == 2) SAGE_LOGIT_FWD(2, false);
  else SAGE_LOGIT_FWD(4, false);
#undef SAGE_LOGIT_FWD
#undef SAGE_LOGIT_FWD_P
#undef SAGE_LOGIT_FWD_Q
  SAGE_CHECK_LAUNCH("logit_fwd_kernel");
  return SAGE_OK;
}
}  // namespace

extern "C" int sage_logit_layer_fwd(const float* h_rows, sage_i64 ld_h, const sage_i64* seg_ptr, const sage_i64* src,
                                    sage_i64 m, int h, int reduce, const float* w_l, sage_i64 ld_wl,
                                    const float* w_r, sage_i64 ld_wr, int n, const sage_i64* labels,
                                    const sage_i64* label_rows, float grad_scale, float* agg, sage_i64 ld_agg,
                                    float* y, sage_i64 ld_y, float* inv_norm, float* dz, sage_i64 ld_dz,
                                    float* row_loss, float* h_root, sage_i64 ld_h_root, void* stream_) {
  SAGE_REQUIRE(h_rows && seg_ptr && src && labels && agg && dz && row_loss, SAGE_EINVAL,
               "sage_logit_layer_fwd: NULL pointer argument");
  SAGE_REQUIRE(ld_h >= 4 && ld_h % 4 == 0 && vec16(h_rows) && ld_agg >= h && ld_agg % 4 == 0 && vec16(agg) && ld_dz >= n &&
                   (y == nullptr || ld_y >= n) && ld_h >= h,
               SAGE_EINVAL, "sage_logit_layer_fwd: rows of h / agg must be 16-byte aligned, ld_dz / ld_y >= n");
  SAGE_REQUIRE(h_root == nullptr || (ld_h_root >= h && ld_h_root % 4 == 0 && vec16(h_root) && h_root != h_rows), SAGE_EINVAL,
               "sage_logit_layer_fwd: rows of h_root must be 16-byte aligned and at least h wide, and not alias h_rows");
  return launch_logit_fwd("sage_logit_layer_fwd", h_root != nullptr, h_rows, ld_h, seg_ptr, src, m, h, reduce, w_l, ld_wl, w_r,
                          ld_wr, n, labels, label_rows, grad_scale, agg, ld_agg, y, ld_y, inv_norm, dz, ld_dz, row_loss,
                          h_root, ld_h_root, nullptr, nullptr, 0, nullptr, static_cast<hipStream_t>(stream_));
}

extern "C" int sage_logit_layer_fwd_bwd(const float* h_rows, sage_i64 ld_h, const sage_i64* seg_ptr, const sage_i64* src,
                                        sage_i64 m, int h, int reduce, const float* w_l, sage_i64 ld_wl,
                                        const float* w_r, sage_i64 ld_wr, int n, const sage_i64* labels,
                                        const sage_i64* label_rows, float grad_scale, float* agg, sage_i64 ld_agg,
                                        float* y, sage_i64 ld_y, float* inv_norm, float* dz, sage_i64 ld_dz,
                                        float* row_loss, float* h_root, sage_i64 ld_h_root, float* dh, sage_i64 ld_dh,
                                        const int* writers, void* stream_) {
  SAGE_REQUIRE(h_rows && seg_ptr && src && labels && agg && dz && row_loss && dh, SAGE_EINVAL,
               "sage_logit_layer_fwd_bwd: NULL pointer argument");
  SAGE_REQUIRE(ld_h >= 4 && ld_h % 4 == 0 && vec16(h_rows) && ld_agg >= h && ld_agg % 4 == 0 && vec16(agg) && ld_dz >= n &&
                   (y == nullptr || ld_y >= n) && ld_h >= h && ld_dh >= h,
               SAGE_EINVAL, "sage_logit_layer_fwd_bwd: rows of h / agg must be 16-byte aligned, ld_dz / ld_y >= n, ld_dh >= h");
  SAGE_REQUIRE(h_root == nullptr || (ld_h_root >= h && ld_h_root % 4 == 0 && vec16(h_root) && h_root != h_rows), SAGE_EINVAL,
               "sage_logit_layer_fwd_bwd: rows of h_root must be 16-byte aligned and at least h wide, and not alias h_rows");
  SAGE_REQUIRE(dh != h_rows && dh != agg && dh != h_root, SAGE_EINVAL, "sage_logit_layer_fwd_bwd: dh aliases an input");
  return launch_logit_fwd("sage_logit_layer_fwd_bwd", h_root != nullptr, h_rows, ld_h, seg_ptr, src, m, h, reduce, w_l, ld_wl,
                          w_r, ld_wr, n, labels, label_rows, grad_scale, agg, ld_agg, y, ld_y, inv_norm, dz, ld_dz, row_loss,
                          h_root, ld_h_root, nullptr, dh, ld_dh, writers, static_cast<hipStream_t>(stream_));
}

extern "C" int sage_logit_layer_pool_fwd_bwd(const float* h_rows, sage_i64 ld_h, const float* p_rows, sage_i64 ld_p,
                                             const sage_i64* seg_ptr, const sage_i64* src, sage_i64 m, int h, int reduce,
                                             const float* w_l, sage_i64 ld_wl, const float* w_r, sage_i64 ld_wr, int n,
                                             const sage_i64* labels, const sage_i64* label_rows, float grad_scale,
                                             float* agg, sage_i64 ld_agg, float* y, sage_i64 ld_y, float* inv_norm,
                                             float* dz, sage_i64 ld_dz, float* row_loss, float* dh, sage_i64 ld_dh,
                                             float* dpre, sage_i64 ld_dpre, const int* writers, void* stream_) {
  SAGE_REQUIRE(h_rows && p_rows && seg_ptr && src && labels && agg && dz && row_loss && dh && dpre, SAGE_EINVAL,
               "sage_logit_layer_pool_fwd_bwd: NULL pointer argument");
  SAGE_REQUIRE(reduce == SAGE_REDUCE_MEAN || reduce == SAGE_REDUCE_MAX, SAGE_EINVAL,
               "sage_logit_layer_pool_fwd_bwd: reduce must be mean or max (mean_pool / max_pool)");
  SAGE_REQUIRE(ld_h >= h && ld_h % 4 == 0 && vec16(h_rows) && ld_p >= h && ld_p % 4 == 0 && vec16(p_rows) && ld_agg >= h &&
                   ld_agg % 4 == 0 && vec16(agg) && ld_dz >= n && (y == nullptr || ld_y >= n) && ld_dh >= h && ld_dpre >= h,
               SAGE_EINVAL,
               "sage_logit_layer_pool_fwd_bwd: rows of h / p / agg must be 16-byte aligned and at least h wide, ld_dz / ld_y >= n");
  SAGE_REQUIRE(dh != h_rows && dh != agg && dpre != p_rows && dpre != agg && dpre != dh, SAGE_EINVAL,
               "sage_logit_layer_pool_fwd_bwd: a gradient buffer aliases an input or the other one");
  return launch_logit_fwd("sage_logit_layer_pool_fwd_bwd", false, h_rows, ld_h, seg_ptr, src, m, h, reduce, w_l, ld_wl, w_r, ld_wr,
                          n, labels, label_rows, grad_scale, agg, ld_agg, y, ld_y, inv_norm, dz, ld_dz, row_loss, nullptr, 0,
                          nullptr, dh, ld_dh, writers, static_cast<hipStream_t>(stream_), p_rows, ld_p, dpre, ld_dpre);
}

namespace {
__global__ void __launch_bounds__(kLogitBlock)
row_writers_init_kernel(int* __restrict__ writers, int64_t rows, int64_t m) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kLogitBlock + threadIdx.x;
  if (i < rows) writers[i] = i < m ? 1 : 0;
}
__global__ void __launch_bounds__(kLogitBlock)
row_writers_count_kernel(const int64_t* __restrict__ seg_ptr, const int64_t* __restrict__ src, int64_t m, int64_t rows,
                         int* __restrict__ writers) {
  const int64_t e = seg_ptr[0] + static_cast<int64_t>(blockIdx.x) * kLogitBlock + threadIdx.x;
  if (e >= seg_ptr[m]) return;
  const int64_t r = src[e];
  if (r >= 0 && r < rows) atomicAdd(writers + r, 1);
}
}  // namespace

extern "C" int sage_count_row_writers(const sage_i64* seg_ptr, const sage_i64* src, sage_i64 m, sage_i64 max_edges,
                                      int* writers, sage_i64 rows, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m > 0 && rows >= m && max_edges >= 0, SAGE_EINVAL, "sage_count_row_writers: bad sizes");
  SAGE_REQUIRE(seg_ptr && src && writers, SAGE_EINVAL, "sage_count_row_writers: NULL pointer argument");
  row_writers_init_kernel<<<static_cast<unsigned>(cdiv(rows, kLogitBlock)), kLogitBlock, 0, stream>>>(writers, rows, m);
  SAGE_CHECK_LAUNCH("row_writers_init_kernel");
  if (max_edges > 0) {
    row_writers_count_kernel<<<static_cast<unsigned>(cdiv(max_edges, kLogitBlock)), kLogitBlock, 0, stream>>>(seg_ptr, src, m,
                                                                                                            rows, writers);
    SAGE_CHECK_LAUNCH("row_writers_count_kernel");
  }
  return SAGE_OK;
}

extern "C" int sage_logit_layer_infer(const float* h_rows, sage_i64 ld_h, int pre_activation, const sage_i64* seg_ptr,
                                      const sage_i64* src, sage_i64 m, int h, int reduce, const float* w_l,
                                      sage_i64 ld_wl, const float* w_r, sage_i64 ld_wr, int n, float* y, sage_i64 ld_y,
                                      sage_i64* pred, void* stream_) {
  SAGE_REQUIRE(h_rows && seg_ptr && src && (y || pred), SAGE_EINVAL,
               "sage_logit_layer_infer: NULL pointer argument (one of y / pred is required)");
  SAGE_REQUIRE(ld_h >= h && ld_h % 4 == 0 && vec16(h_rows) && (y == nullptr || ld_y >= n), SAGE_EINVAL,
               "sage_logit_layer_infer: rows of h must be 16-byte aligned, ld_y >= n");
  return launch_logit_fwd("sage_logit_layer_infer", pre_activation != 0, h_rows, ld_h, seg_ptr, src, m, h, reduce, w_l, ld_wl,
                          w_r, ld_wr, n, nullptr, nullptr, 1.f, nullptr, 0, y, ld_y, nullptr, nullptr, 0, nullptr, nullptr, 0,
                          pred, nullptr, 0, nullptr, static_cast<hipStream_t>(stream_));
}

namespace {
int launch_logit_bwd(const char* who, const float* dz, sage_i64 ld_dz, const sage_i64* seg_ptr, const sage_i64* src,
                     sage_i64 m, int h, int reduce, const float* w_l, sage_i64 ld_wl, const float* w_r, sage_i64 ld_wr,
                     int n, float* dh, sage_i64 ld_dh, const float* row_loss, float* loss, const LogitDwArgs* dw,
                     hipStream_t stream) {
  size_t lds = 0;
  const int rc = check_common(who, m, h, n, reduce, w_l, ld_wl, w_r, ld_wr, &lds);
  if (rc != SAGE_OK) return rc;
  SAGE_REQUIRE(dz && seg_ptr && src && dh && ld_dz >= n && ld_dh >= h, SAGE_EINVAL,
               "%s: NULL pointer argument or row stride below the width", who);
  SAGE_REQUIRE((loss == nullptr) == (row_loss == nullptr), SAGE_EINVAL, "%s: pass row_loss and loss together (or neither)",
               who);
  const unsigned seed_blocks = static_cast<unsigned>(cdiv(m, kLogitWaves));
  const int mean = reduce == SAGE_REDUCE_MEAN ? 1 : 0;
#define SAGE_LOGIT_BWD(NC, FULL)                                                                                \
  do {                                                                                                      \
    if (dw == nullptr) {                                                                                    \
      const int rc2 = allow_lds(logit_bwd_kernel<NC, FULL>, lds, who);                                      \
      if (rc2 != SAGE_OK) return rc2;                                                                       \
      logit_bwd_kernel<NC, FULL><<<seed_blocks, kLogitBlock, lds, stream>>>(                                \
          dz, ld_dz, seg_ptr, src, m, h, mean, w_l, ld_wl, w_r, ld_wr, n, dh, ld_dh, row_loss, loss);       \
    } else {                                                                                                \
      const int rc2 = allow_lds(logit_bwd_dw_kernel<NC, FULL>, lds, who);                                   \
      if (rc2 != SAGE_OK) return rc2;                                                                       \
      const unsigned grid = seed_blocks + static_cast<unsigned>(dw->gx * dw->gy * 2 * dw->slices);          \
      logit_bwd_dw_kernel<NC, FULL><<<grid, kLogitBlock, lds, stream>>>(                                    \
          dz, ld_dz, seg_ptr, src, m, h, mean, w_l, ld_wl, w_r, ld_wr, n, dh, ld_dh, row_loss, loss,        \
          static_cast<int>(seed_blocks), *dw);                                                              \
    }                                                                                                       \
  } while (0)
  const int nc = (h + kWave - 1) / kWave;
  if (h == 256) SAGE_LOGIT_BWD(4, true);
  else if (nc <= 4) SAGE_LOGIT_BWD(4, false);
  else if (nc <= 8) SAGE_LOGIT_BWD(8, false);
  else SAGE_LOGIT_BWD(16, false);
#undef SAGE_LOGIT_BWD
  SAGE_CHECK_LAUNCH("logit_bwd_kernel");
  return SAGE_OK;
}
}  // namespace

extern "C" int sage_logit_layer_bwd(const float* dz, sage_i64 ld_dz, const sage_i64* seg_ptr, const sage_i64* src,
                                    sage_i64 m, int h, int reduce, const float* w_l, sage_i64 ld_wl,
                                    const float* w_r, sage_i64 ld_wr, int n, float* dh, sage_i64 ld_dh,
                                    const float* row_loss, float* loss, void* stream_) {
  return launch_logit_bwd("sage_logit_layer_bwd", dz, ld_dz, seg_ptr, src, m, h, reduce, w_l, ld_wl, w_r, ld_wr, n, dh,
                          ld_dh, row_loss, loss, nullptr, static_cast<hipStream_t>(stream_));
}

extern "C" int sage_logit_layer_bwd_dw(const float* dz, sage_i64 ld_dz, const sage_i64* seg_ptr, const sage_i64* src,
                                       sage_i64 m, const sage_i64* m_dev, int h, int reduce, const float* w_l,
                                       sage_i64 ld_wl, const float* w_r, sage_i64 ld_wr, int n, float* dh,
                                       sage_i64 ld_dh, const float* row_loss, float* loss, const float* agg,
                                       sage_i64 ld_agg, float* dw_l, sage_i64 ld_dwl, const float* h_root,
                                       sage_i64 ld_h_root, float* dw_r, sage_i64 ld_dwr, void* stream_) {
  SAGE_REQUIRE(agg && dw_l && h_root && dw_r && ld_agg >= h && ld_dwl >= h && ld_h_root >= h && ld_dwr >= h, SAGE_EINVAL,
               "sage_logit_layer_bwd_dw: NULL weight-gradient operand or row stride below the width");
  SAGE_REQUIRE(m > 0 && m < (1ll << 31), SAGE_ELIMIT, "sage_logit_layer_bwd_dw: m out of range");
  if (static_cast<size_t>(n) * 2 * h * sizeof(float) + 4096 > 160 * 1024) {
    // no room for the weight-gradient workgroups' stage beside the weights: two launches
    const int rc = sage_linear_dweight2(dz, ld_dz, agg, ld_agg, nullptr, h, dw_l, ld_dwl, h_root, ld_h_root, nullptr, h,
                                        dw_r, ld_dwr, m, m_dev, n, stream_);
    if (rc != SAGE_OK) return rc;
    return sage_logit_layer_bwd(dz, ld_dz, seg_ptr, src, m, h, reduce, w_l, ld_wl, w_r, ld_wr, n, dh, ld_dh, row_loss,
                                loss, stream_);
  }
  // the grid of sage_linear_dweight2's skinny pair: >= 32 rows of m per block, <= 16 slices
  LogitDwArgs d = {};
  d.p0 = {agg, ld_agg, nullptr, h, dw_l, ld_dwl};
  d.p1 = {h_root, ld_h_root, nullptr, h, dw_r, ld_dwr};
  int64_t sp = cdiv(m, 32);
  if (sp > 16) sp = 16;
  if (sp < 1) sp = 1;
  d.m_chunk = static_cast<int>(cdiv(m, sp));
  d.slices = static_cast<int>(cdiv(m, d.m_chunk));
  d.gx = static_cast<int>(cdiv(h, kSkinnyThreads));
  d.gy = static_cast<int>(cdiv(n, kDwRows));
  d.m_dev = m_dev;
  return launch_logit_bwd("sage_logit_layer_bwd_dw", dz, ld_dz, seg_ptr, src, m, h, reduce, w_l, ld_wl, w_r, ld_wr, n, dh,
                          ld_dh, row_loss, loss, &d, static_cast<hipStream_t>(stream_));
}
