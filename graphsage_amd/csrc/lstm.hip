// (a-7) LSTM aggregator pieces.
//
// The reference's 'lstm' aggregator (reference layers/sage.py:67-69, :147-193) builds, per
// node, the list of its neighbours' features truncated / padded to LSTM_NUM_INPUTS rows
// (padding repeats the first row, :164), FLATTENS it to one vector of n_in*F values (:179) and
// feeds it to ONE step of torch.nn.LSTMCell(n_in*F, F) from a zero state (:187), keeping h.
// Its Python is disabled upstream as broken (settings.py:48-50: dict loops keyed by source ids,
// per-edge tensors indexed by node ids, random.sample truncation), so what is implemented here
// is that computation with the indexing made well-defined (per TARGET, neighbours in edge
// order, the first n_in kept) — see graphsage_amd/layers/lstm.py.  The contraction
// gates = x_flat @ W_ih^T + b runs on the MFMA kernel of linear.hip; this file holds the two
// pieces that are not a GEMM: the slot index builder and the zero-state cell non-linearity.
#include "common.h"

namespace sage {
namespace {

constexpr int kBlock = 256;

// slots[j, t] = row id of the j-th input of target t (slot-major: the n_in row maps of the gates
// contraction's K-segments are contiguous vectors); nonempty[t] = segment has an edge.
__global__ void segment_slots_kernel(const int64_t* __restrict__ seg_ptr,
                                     const int64_t* __restrict__ src,
                                     const int64_t* __restrict__ node_map, int64_t num_targets,
                                     int n_in, int64_t* __restrict__ slots,
                                     float* __restrict__ nonempty) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= num_targets * n_in) return;
  const int j = static_cast<int>(i / num_targets);
  const int64_t t = i - j * num_targets;
  const int64_t s0 = seg_ptr[t];
  const int64_t len = seg_ptr[t + 1] - s0;
  int64_t id = 0;
  if (len > 0) {
    id = src[s0 + (j < len ? j : 0)];  // pad with the first neighbour (reference :164)
    if (node_map != nullptr) id = node_map[id];
  }
  slots[i] = id;
  if (j == 0) nonempty[t] = len > 0 ? 1.f : 0.f;
}

__device__ inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// gates [m, 4F] in torch's (i, f, g, o) order; zero initial state: c = s(i)*tanh(g), h = s(o)*tanh(c)
// bias_a / bias_b (optional, [4F] each): added to the gates on the way in (b_ih + b_hh of the
// cell, when the contraction that produced `gates` ran without a bias epilogue)
__device__ inline float gate_bias(const float* __restrict__ a, const float* __restrict__ b, int c) {
  return (a != nullptr ? a[c] : 0.f) + (b != nullptr ? b[c] : 0.f);
}

__global__ void lstm_cell0_fwd_kernel(const float* __restrict__ gates, int64_t ld_g,
                                      const float* __restrict__ nonempty, int64_t m,
                                      const int64_t* __restrict__ m_dev, int F,
                                      const float* __restrict__ bias_a, const float* __restrict__ bias_b,
                                      float* __restrict__ h, int64_t ld_h) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= dev_count(m_dev, m) * F) return;
  const int64_t r = i / F;
  const int f = static_cast<int>(i - r * F);
  const float* g = gates + r * ld_g;
  const float gi = g[f] + gate_bias(bias_a, bias_b, f);
  const float gg = g[2 * F + f] + gate_bias(bias_a, bias_b, 2 * F + f);
  const float go = g[3 * F + f] + gate_bias(bias_a, bias_b, 3 * F + f);
  const float c = sigmoidf(gi) * tanhf(gg);
  const float out = sigmoidf(go) * tanhf(c);
  h[r * ld_h + f] = (nonempty == nullptr || nonempty[r] != 0.f) ? out : 0.f;
}

__global__ void lstm_cell0_bwd_kernel(const float* __restrict__ gates, int64_t ld_g,
                                      const float* __restrict__ nonempty,
                                      const float* __restrict__ dh, int64_t ld_dh, int64_t m,
                                      const int64_t* __restrict__ m_dev, int F,
                                      const float* __restrict__ bias_a, const float* __restrict__ bias_b,
                                      float* __restrict__ dgates, int64_t ld_dg) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= dev_count(m_dev, m) * F) return;
  const int64_t r = i / F;
  const int f = static_cast<int>(i - r * F);
  const float* g = gates + r * ld_g;
  float* dg = dgates + r * ld_dg;
  float d = dh[r * ld_dh + f];
  if (nonempty != nullptr && nonempty[r] == 0.f) d = 0.f;
  const float si = sigmoidf(g[f] + gate_bias(bias_a, bias_b, f));
  const float tg = tanhf(g[2 * F + f] + gate_bias(bias_a, bias_b, 2 * F + f));
  const float so = sigmoidf(g[3 * F + f] + gate_bias(bias_a, bias_b, 3 * F + f));
  const float c = si * tg;
  const float tc = tanhf(c);
  const float dc = d * so * (1.f - tc * tc);
  dg[f] = dc * tg * si * (1.f - si);
  dg[F + f] = 0.f;  // forget gate multiplies the zero initial cell state
  dg[2 * F + f] = dc * si * (1.f - tg * tg);
  dg[3 * F + f] = d * tc * so * (1.f - so);
}

// x_flat[t, j F : (j+1) F] = x[row(t, j)]: the flattened input of the cell (reference
// layers/sage.py:164, :179) for every target of a layer, slot rule as in segment_slots_kernel —
// the first n_in neighbours in edge order, padded with the first, through node_map if given.
// One workgroup per target, a wave per slot in turn; VEC floats per lane and access (the slot
// offsets j F are only as aligned as F is: F = 602 -> 8 bytes).
template <int VEC>
__global__ void __launch_bounds__(kBlock)
lstm_gather_flat_kernel(const float* __restrict__ x, int64_t ld_x, const int64_t* __restrict__ node_map,
                        const int64_t* __restrict__ seg_ptr, const int64_t* __restrict__ src,
                        int64_t num_targets, const int64_t* __restrict__ m_dev, int n_in, int F,
                        float* __restrict__ xflat, int64_t ld_flat, float* __restrict__ nonempty) {
  const int64_t t = blockIdx.x;
  if (t >= dev_count(m_dev, num_targets)) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int64_t s0 = seg_ptr[t];
  const int64_t len = seg_ptr[t + 1] - s0;
  if (threadIdx.x == 0) nonempty[t] = len > 0 ? 1.f : 0.f;
  float* orow = xflat + t * ld_flat;
  for (int j = wave; j < n_in; j += kBlock / kWave) {
    int64_t id = 0;
    if (len > 0) {
      id = src[s0 + (j < len ? j : 0)];
      if (node_map != nullptr) id = node_map[id];
    }
    const float* irow = x + id * ld_x;
    float* o = orow + static_cast<int64_t>(j) * F;
    if (VEC == 4) {
      for (int c = lane; c < F / 4; c += kWave)
        reinterpret_cast<float4*>(o)[c] = len > 0 ? reinterpret_cast<const float4*>(irow)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (VEC == 2) {
      for (int c = lane; c < F / 2; c += kWave)
        reinterpret_cast<float2*>(o)[c] = len > 0 ? reinterpret_cast<const float2*>(irow)[c] : make_float2(0.f, 0.f);
    } else {
      for (int c = lane; c < F; c += kWave) o[c] = len > 0 ? irow[c] : 0.f;
    }
  }
}

// dx[row(t, j)] += d x_flat[t, j F : (j+1) F] — what autograd derives for the index_select that
// builds x_flat (a padded slot sends its gradient to the first neighbour as well).  fp32 atomics,
// a lane owns columns lane + 64 k (256 contiguous bytes per wave instruction).
__global__ void __launch_bounds__(kBlock)
lstm_scatter_flat_kernel(const float* __restrict__ dflat, int64_t ld_flat, const int64_t* __restrict__ seg_ptr,
                         const int64_t* __restrict__ src, int64_t num_targets,
                         const int64_t* __restrict__ m_dev, int n_in, int F, float* __restrict__ dx,
                         int64_t ld_dx) {
  const int64_t t = blockIdx.x;
  if (t >= dev_count(m_dev, num_targets)) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int64_t s0 = seg_ptr[t];
  const int64_t len = seg_ptr[t + 1] - s0;
  if (len <= 0) return;                 // (its gates received no gradient)
  const float* irow = dflat + t * ld_flat;
  for (int j = wave; j < n_in; j += kBlock / kWave) {
    const int64_t id = src[s0 + (j < len ? j : 0)];
    const float* in = irow + static_cast<int64_t>(j) * F;
    float* o = dx + id * ld_dx;
    for (int c = lane; c < F; c += kWave) atomicAdd(o + c, in[c]);
  }
}

// Targets whose segments are SHORTER than n_in by construction (the rows behind the seeds in a
// sampled batch: their edges come from a hop with a smaller fan-out k) repeat their first
// neighbour in slots k .. n_in-1, so for them
//     x_flat[t] . W_ih^T = sum_{j<k} x_j . W_j^T  +  x_0 . (sum_{j>=k} W_j)^T
// (W_j = columns j F .. (j+1) F of W_ih): the padded slots cost one slot's flops instead of
// n_in - k.  out[r, f] = sum_{j0 <= j < j1} w[r, j F + f], summed in slot order.
__global__ void __launch_bounds__(kBlock)
lstm_pad_weight_sum_kernel(const float* __restrict__ w, int64_t ld_w, int64_t rows, int F, int j0, int j1,
                           float* __restrict__ out, int64_t ld_out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= rows * F) return;
  const int64_t r = i / F;
  const int f = static_cast<int>(i - r * F);
  const float* wr = w + r * ld_w + f;
  float s = 0.f;
  for (int j = j0; j < j1; ++j) s += wr[static_cast<int64_t>(j) * F];
  out[r * ld_out + f] = s;
}

// ... and the backward of that sum: every padded slot's weight gradient receives the gradient of
// the summed block, dw[r, j F + f] += g[r, f] for j0 <= j < j1; g is handed back zeroed (the weight
// gradient kernels accumulate into it).
__global__ void __launch_bounds__(kBlock)
lstm_pad_grad_bcast_kernel(float* __restrict__ g, int64_t ld_g, int64_t rows, int F, int j0, int j1,
                           float* __restrict__ dw, int64_t ld_dw) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= rows * F) return;
  const int64_t r = i / F;
  const int f = static_cast<int>(i - r * F);
  const float v = g[r * ld_g + f];
  g[r * ld_g + f] = 0.f;
  float* dr = dw + r * ld_dw + f;
  for (int j = j0; j < j1; ++j) dr[static_cast<int64_t>(j) * F] += v;
}

}  // namespace
}  // namespace sage

using namespace sage;

extern "C" int sage_lstm_pad_weight_sum(const float* w, sage_i64 ld_w, sage_i64 rows, int num_features, int j0, int j1,
                                        float* out, sage_i64 ld_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(rows >= 0 && num_features > 0 && 0 <= j0 && j0 <= j1 && ld_w >= static_cast<int64_t>(j1) * num_features &&
                   ld_out >= num_features,
               SAGE_EINVAL, "sage_lstm_pad_weight_sum: bad sizes");
  if (rows == 0) return SAGE_OK;
  SAGE_REQUIRE(w && out, SAGE_EINVAL, "sage_lstm_pad_weight_sum: NULL pointer argument");
  lstm_pad_weight_sum_kernel<<<static_cast<unsigned>(cdiv(rows * num_features, kBlock)), kBlock, 0, stream>>>(
      w, ld_w, rows, num_features, j0, j1, out, ld_out);
  SAGE_CHECK_LAUNCH("lstm_pad_weight_sum_kernel");
  return SAGE_OK;
}

extern "C" int sage_lstm_pad_grad_bcast(float* g, sage_i64 ld_g, sage_i64 rows, int num_features, int j0, int j1,
                                        float* dw, sage_i64 ld_dw, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(rows >= 0 && num_features > 0 && 0 <= j0 && j0 <= j1 && ld_dw >= static_cast<int64_t>(j1) * num_features &&
                   ld_g >= num_features,
               SAGE_EINVAL, "sage_lstm_pad_grad_bcast: bad sizes");
  if (rows == 0) return SAGE_OK;
  SAGE_REQUIRE(g && dw, SAGE_EINVAL, "sage_lstm_pad_grad_bcast: NULL pointer argument");
  lstm_pad_grad_bcast_kernel<<<static_cast<unsigned>(cdiv(rows * num_features, kBlock)), kBlock, 0, stream>>>(
      g, ld_g, rows, num_features, j0, j1, dw, ld_dw);
  SAGE_CHECK_LAUNCH("lstm_pad_grad_bcast_kernel");
  return SAGE_OK;
}

extern "C" int sage_lstm_gather_flat(const float* x, sage_i64 ld_x, const sage_i64* node_map,
                                     const sage_i64* seg_ptr, const sage_i64* src, sage_i64 num_targets,
                                     const sage_i64* num_targets_dev, int n_in, int num_features,
                                     float* x_flat, sage_i64 ld_flat, float* nonempty, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(num_targets >= 0 && n_in > 0 && num_features > 0 && ld_x >= num_features &&
                   ld_flat >= static_cast<int64_t>(n_in) * num_features,
               SAGE_EINVAL, "sage_lstm_gather_flat: bad sizes");
  if (num_targets == 0) return SAGE_OK;
  SAGE_REQUIRE(x && seg_ptr && src && x_flat && nonempty, SAGE_EINVAL, "sage_lstm_gather_flat: NULL pointer argument");
  auto aligned = [&](int bytes) {
    return (reinterpret_cast<uintptr_t>(x) % bytes) == 0 && (reinterpret_cast<uintptr_t>(x_flat) % bytes) == 0 &&
           (ld_x * 4) % bytes == 0 && (ld_flat * 4) % bytes == 0 && (num_features * 4) % bytes == 0;
  };
  const unsigned grid = static_cast<unsigned>(num_targets);
#define SAGE_LSTM_GATHER(V)                                                                                        \
  lstm_gather_flat_kernel<V><<<grid, kBlock, 0, stream>>>(x, ld_x, node_map, seg_ptr, src, num_targets, num_targets_dev, \
                                                         n_in, num_features, x_flat, ld_flat, nonempty)
  if (aligned(16)) SAGE_LSTM_GATHER(4);
  else if (aligned(8)) SAGE_LSTM_GATHER(2);
  else SAGE_LSTM_GATHER(1);
#undef SAGE_LSTM_GATHER
  SAGE_CHECK_LAUNCH("lstm_gather_flat_kernel");
  return SAGE_OK;
}

extern "C" int sage_lstm_scatter_flat(const float* d_flat, sage_i64 ld_flat, const sage_i64* seg_ptr,
                                      const sage_i64* src, sage_i64 num_targets, const sage_i64* num_targets_dev,
                                      int n_in, int num_features, float* dx, sage_i64 ld_dx, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(num_targets >= 0 && n_in > 0 && num_features > 0 && ld_dx >= num_features &&
                   ld_flat >= static_cast<int64_t>(n_in) * num_features,
               SAGE_EINVAL, "sage_lstm_scatter_flat: bad sizes");
  if (num_targets == 0) return SAGE_OK;
  SAGE_REQUIRE(d_flat && seg_ptr && src && dx, SAGE_EINVAL, "sage_lstm_scatter_flat: NULL pointer argument");
  lstm_scatter_flat_kernel<<<static_cast<unsigned>(num_targets), kBlock, 0, stream>>>(
      d_flat, ld_flat, seg_ptr, src, num_targets, num_targets_dev, n_in, num_features, dx, ld_dx);
  SAGE_CHECK_LAUNCH("lstm_scatter_flat_kernel");
  return SAGE_OK;
}

extern "C" int sage_segment_slots(const sage_i64* seg_ptr, const sage_i64* src,
                                  const sage_i64* node_map, sage_i64 num_targets, int n_in,
                                  sage_i64* slots, float* nonempty, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(num_targets >= 0 && n_in > 0, SAGE_EINVAL, "sage_segment_slots: bad sizes");
  if (num_targets == 0) return SAGE_OK;
  SAGE_REQUIRE(seg_ptr && src && slots && nonempty, SAGE_EINVAL, "sage_segment_slots: NULL pointer argument");
  const int64_t total = num_targets * n_in;
  segment_slots_kernel<<<static_cast<unsigned>(cdiv(total, kBlock)), kBlock, 0, stream>>>(
      seg_ptr, src, node_map, num_targets, n_in, slots, nonempty);
  SAGE_CHECK_LAUNCH("segment_slots_kernel");
  return SAGE_OK;
}

extern "C" int sage_lstm_cell0_fwd(const float* gates, sage_i64 ld_gates, const float* nonempty,
                                   sage_i64 m, const sage_i64* m_dev, int num_features, const float* bias_a,
                                   const float* bias_b, float* h, sage_i64 ld_h, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && num_features > 0 && ld_gates >= 4 * static_cast<int64_t>(num_features) &&
                   ld_h >= num_features,
               SAGE_EINVAL, "sage_lstm_cell0_fwd: bad shape");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(gates && h, SAGE_EINVAL, "sage_lstm_cell0_fwd: NULL pointer argument");
  lstm_cell0_fwd_kernel<<<static_cast<unsigned>(cdiv(m * num_features, kBlock)), kBlock, 0, stream>>>(
      gates, ld_gates, nonempty, m, m_dev, num_features, bias_a, bias_b, h, ld_h);
  SAGE_CHECK_LAUNCH("lstm_cell0_fwd_kernel");
  return SAGE_OK;
}

extern "C" int sage_lstm_cell0_bwd(const float* gates, sage_i64 ld_gates, const float* nonempty,
                                   const float* dh, sage_i64 ld_dh, sage_i64 m, const sage_i64* m_dev,
                                   int num_features, const float* bias_a, const float* bias_b,
                                   float* dgates, sage_i64 ld_dgates, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && num_features > 0 && ld_gates >= 4 * static_cast<int64_t>(num_features) &&
                   ld_dgates >= 4 * static_cast<int64_t>(num_features) && ld_dh >= num_features,
               SAGE_EINVAL, "sage_lstm_cell0_bwd: bad shape");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(gates && dh && dgates, SAGE_EINVAL, "sage_lstm_cell0_bwd: NULL pointer argument");
  lstm_cell0_bwd_kernel<<<static_cast<unsigned>(cdiv(m * num_features, kBlock)), kBlock, 0, stream>>>(
      gates, ld_gates, nonempty, dh, ld_dh, m, m_dev, num_features, bias_a, bias_b, dgates, ld_dgates);
  SAGE_CHECK_LAUNCH("lstm_cell0_bwd_kernel");
  return SAGE_OK;
}
