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
__global__ void lstm_cell0_fwd_kernel(const float* __restrict__ gates, int64_t ld_g,
                                      const float* __restrict__ nonempty, int64_t m, int F,
                                      float* __restrict__ h, int64_t ld_h) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m * F) return;
  const int64_t r = i / F;
  const int f = static_cast<int>(i - r * F);
  const float* g = gates + r * ld_g;
  const float c = sigmoidf(g[f]) * tanhf(g[2 * F + f]);
  const float out = sigmoidf(g[3 * F + f]) * tanhf(c);
  h[r * ld_h + f] = (nonempty == nullptr || nonempty[r] != 0.f) ? out : 0.f;
}

__global__ void lstm_cell0_bwd_kernel(const float* __restrict__ gates, int64_t ld_g,
                                      const float* __restrict__ nonempty,
                                      const float* __restrict__ dh, int64_t ld_dh, int64_t m, int F,
                                      float* __restrict__ dgates, int64_t ld_dg) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m * F) return;
  const int64_t r = i / F;
  const int f = static_cast<int>(i - r * F);
  const float* g = gates + r * ld_g;
  float* dg = dgates + r * ld_dg;
  float d = dh[r * ld_dh + f];
  if (nonempty != nullptr && nonempty[r] == 0.f) d = 0.f;
  const float si = sigmoidf(g[f]), tg = tanhf(g[2 * F + f]), so = sigmoidf(g[3 * F + f]);
  const float c = si * tg;
  const float tc = tanhf(c);
  const float dc = d * so * (1.f - tc * tc);
  dg[f] = dc * tg * si * (1.f - si);
  dg[F + f] = 0.f;  // forget gate multiplies the zero initial cell state
  dg[2 * F + f] = dc * si * (1.f - tg * tg);
  dg[3 * F + f] = d * tc * so * (1.f - so);
}

}  // namespace
}  // namespace sage

using namespace sage;

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
                                   sage_i64 m, int num_features, float* h, sage_i64 ld_h,
                                   void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && num_features > 0 && ld_gates >= 4 * static_cast<int64_t>(num_features) &&
                   ld_h >= num_features,
               SAGE_EINVAL, "sage_lstm_cell0_fwd: bad shape");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(gates && h, SAGE_EINVAL, "sage_lstm_cell0_fwd: NULL pointer argument");
  lstm_cell0_fwd_kernel<<<static_cast<unsigned>(cdiv(m * num_features, kBlock)), kBlock, 0, stream>>>(
      gates, ld_gates, nonempty, m, num_features, h, ld_h);
  SAGE_CHECK_LAUNCH("lstm_cell0_fwd_kernel");
  return SAGE_OK;
}

extern "C" int sage_lstm_cell0_bwd(const float* gates, sage_i64 ld_gates, const float* nonempty,
                                   const float* dh, sage_i64 ld_dh, sage_i64 m, int num_features,
                                   float* dgates, sage_i64 ld_dgates, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && num_features > 0 && ld_gates >= 4 * static_cast<int64_t>(num_features) &&
                   ld_dgates >= 4 * static_cast<int64_t>(num_features) && ld_dh >= num_features,
               SAGE_EINVAL, "sage_lstm_cell0_bwd: bad shape");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(gates && dh && dgates, SAGE_EINVAL, "sage_lstm_cell0_bwd: NULL pointer argument");
  lstm_cell0_bwd_kernel<<<static_cast<unsigned>(cdiv(m * num_features, kBlock)), kBlock, 0, stream>>>(
      gates, ld_gates, nonempty, dh, ld_dh, m, num_features, dgates, ld_dgates);
  SAGE_CHECK_LAUNCH("lstm_cell0_bwd_kernel");
  return SAGE_OK;
}
