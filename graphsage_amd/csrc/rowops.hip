// Row-wise pieces of the layer and of the training step: L2 normalise (+ReLU) forward and
// backward, softmax cross-entropy, Adam.  All HBM-bound and tiny next to the gathers
// (T1 x 256 fp32 = 13.6 MB per pass at Reddit 25/10), so they are plain one-wave-per-row
// kernels with shuffle reductions.
//
//   F.normalize(out, p=2, dim=-1)         reference layers/sage.py:131-132
//   x.relu_()                             reference models/graphsage.py:47-48
//   F.cross_entropy(y_hat, y)             reference experiments/fig2a/runners/graphsage_runners.py:43
//   torch.optim.Adam(...).step()          reference experiments/fig2a/runners/graphsage_runners.py:45,86
#include "common.h"
#include "skinny_dw.h"

namespace sage {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr float kNormEps = 1e-12f;  // torch.nn.functional.normalize default

inline bool vec16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// V4: n and every row stride are multiples of 4 and the bases 16-byte aligned — each lane moves
// 16-byte vectors (one per lane covers n = 256), twice the bytes per instruction of the scalar form
template <bool V4>
__global__ void __launch_bounds__(kBlock)
l2norm_relu_fwd_kernel(float* __restrict__ z, int64_t ld_z, int64_t m,
                       const int64_t* __restrict__ m_dev, int n, float* __restrict__ y,
                       int64_t ld_y, float* __restrict__ h, int64_t ld_h,
                       float* __restrict__ inv_norm, int clear_z) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t rows = dev_count(m_dev, m);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (r >= rows) return;
  float* zr = z + r * ld_z;
  float ss = 0.f;
  if constexpr (V4) {
    for (int f = lane * 4; f < n; f += kWave * 4) {
      const float4 v = *reinterpret_cast<const float4*>(zr + f);
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  } else {
    for (int f = lane; f < n; f += kWave) {
      const float v = zr[f];
      ss += v * v;
    }
  }
  ss = wave_sum(ss);
  const float denom = fmaxf(sqrtf(ss), kNormEps);
  if (lane == 0 && inv_norm != nullptr) inv_norm[r] = 1.f / denom;
  if constexpr (V4) {
    for (int f = lane * 4; f < n; f += kWave * 4) {
      float4 v = *reinterpret_cast<const float4*>(zr + f);
      v.x /= denom; v.y /= denom; v.z /= denom; v.w /= denom;
      *reinterpret_cast<float4*>(y + r * ld_y + f) = v;
      if (h != nullptr) {
        *reinterpret_cast<float4*>(h + r * ld_h + f) =
            make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
      }
      // the producer adds partial sums (stream-K / split-K) into zeros next time
      if (clear_z) *reinterpret_cast<float4*>(zr + f) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    for (int f = lane; f < n; f += kWave) {
      const float v = zr[f] / denom;
      y[r * ld_y + f] = v;
      if (h != nullptr) h[r * ld_h + f] = v > 0.f ? v : 0.f;
      if (clear_z) zr[f] = 0.f;
    }
  }
}

template <bool V4>
__global__ void __launch_bounds__(kBlock)
l2norm_relu_bwd_kernel(float* __restrict__ dh, int64_t ld_dh, const float* __restrict__ y,
                       int64_t ld_y, const float* __restrict__ inv_norm, int relu_mask, int64_t m,
                       const int64_t* __restrict__ m_dev, int n, float* __restrict__ dz,
                       int64_t ld_dz, int clear_dh) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t rows = dev_count(m_dev, m);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (r >= rows) return;
  const float* yr = y + r * ld_y;
  float* gr = dh + r * ld_dh;
  const float inv = inv_norm[r];
  const bool clamped = inv >= 1.f / kNormEps;  // ||z|| <= eps: the clamp's gradient is zero
  float dot = 0.f;
  if constexpr (V4) {
    for (int f = lane * 4; f < n; f += kWave * 4) {
      const float4 yv = *reinterpret_cast<const float4*>(yr + f);
      float4 g = *reinterpret_cast<const float4*>(gr + f);
      if (relu_mask) {
        if (!(yv.x > 0.f)) g.x = 0.f;
        if (!(yv.y > 0.f)) g.y = 0.f;
        if (!(yv.z > 0.f)) g.z = 0.f;
        if (!(yv.w > 0.f)) g.w = 0.f;
      }
      dot += yv.x * g.x + yv.y * g.y + yv.z * g.z + yv.w * g.w;
    }
  } else {
    for (int f = lane; f < n; f += kWave) {
      const float yv = yr[f];
      float g = gr[f];
      if (relu_mask && !(yv > 0.f)) g = 0.f;
      dot += yv * g;
    }
  }
  dot = clamped ? 0.f : wave_sum(dot);
  if constexpr (V4) {
    for (int f = lane * 4; f < n; f += kWave * 4) {
      const float4 yv = *reinterpret_cast<const float4*>(yr + f);
      float4 g = *reinterpret_cast<const float4*>(gr + f);
      if (relu_mask) {
        if (!(yv.x > 0.f)) g.x = 0.f;
        if (!(yv.y > 0.f)) g.y = 0.f;
        if (!(yv.z > 0.f)) g.z = 0.f;
        if (!(yv.w > 0.f)) g.w = 0.f;
      }
      *reinterpret_cast<float4*>(dz + r * ld_dz + f) =
          make_float4((g.x - yv.x * dot) * inv, (g.y - yv.y * dot) * inv, (g.z - yv.z * dot) * inv,
                      (g.w - yv.w * dot) * inv);
      // consume-and-clear: the next step's scatter-add starts from zeros without a fill kernel
      if (clear_dh) *reinterpret_cast<float4*>(gr + f) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    for (int f = lane; f < n; f += kWave) {
      const float yv = yr[f];
      float g = gr[f];
      if (relu_mask && !(yv > 0.f)) g = 0.f;
      dz[r * ld_dz + f] = (g - yv * dot) * inv;
      if (clear_dh) gr[f] = 0.f;
    }
  }
}

// The same backward from the PRE-normalisation rows z (no y / inv_norm kept by a forward pass:
// the class-logit kernel normalises the rows it gathers on the fly, logit.hip): one wave per
// row, the row of z and of dh held in registers across both reductions (|z|^2, then <y, g>);
// z is handed back zeroed, like dh, for the producers that add into it.
template <int NV>      // float4 per lane: n <= 256 NV
__device__ __forceinline__ void
l2norm_relu_bwd_z_body(int64_t block, float* __restrict__ dh, int64_t ld_dh, float* __restrict__ z, int64_t ld_z,
                       int relu_mask, int64_t m, const int64_t* __restrict__ m_dev, int n,
                       float* __restrict__ dz, int64_t ld_dz, int clear_dh) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t rows = dev_count(m_dev, m);
  const int64_t r = block * kWavesPerBlock + threadIdx.x / kWave;
  if (r >= rows) return;
  const int nq = n / 4;
  float4* zr = reinterpret_cast<float4*>(z + r * ld_z);
  float4* gr = reinterpret_cast<float4*>(dh + r * ld_dh);
  float4* or_ = reinterpret_cast<float4*>(dz + r * ld_dz);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 zv[NV], g[NV];
  float ss = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int q = lane + kWave * v;
    zv[v] = g[v] = zero4;
    if (q < nq) {
      zv[v] = zr[q];
      g[v] = gr[q];
    }
    ss += zv[v].x * zv[v].x + zv[v].y * zv[v].y + zv[v].z * zv[v].z + zv[v].w * zv[v].w;
  }
  ss = wave_sum(ss);
  const float denom = fmaxf(sqrtf(ss), kNormEps);
  const float inv = 1.f / denom;
  const bool clamped = inv >= 1.f / kNormEps;  // ||z|| <= eps: the clamp's gradient is zero
  float dot = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    zv[v].x /= denom; zv[v].y /= denom; zv[v].z /= denom; zv[v].w /= denom;      // y, as the forward pass writes it
    if (relu_mask) {
      if (!(zv[v].x > 0.f)) g[v].x = 0.f;
      if (!(zv[v].y > 0.f)) g[v].y = 0.f;
      if (!(zv[v].z > 0.f)) g[v].z = 0.f;
      if (!(zv[v].w > 0.f)) g[v].w = 0.f;
    }
    dot += zv[v].x * g[v].x + zv[v].y * g[v].y + zv[v].z * g[v].z + zv[v].w * g[v].w;
  }
  dot = clamped ? 0.f : wave_sum(dot);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int q = lane + kWave * v;
    if (q < nq) {
      or_[q] = make_float4((g[v].x - zv[v].x * dot) * inv, (g[v].y - zv[v].y * dot) * inv,
                           (g[v].z - zv[v].z * dot) * inv, (g[v].w - zv[v].w * dot) * inv);
      zr[q] = zero4;
      if (clear_dh) gr[q] = zero4;
    }
  }
}
template <int NV>
__global__ void __launch_bounds__(kBlock)
l2norm_relu_bwd_z_kernel(float* __restrict__ dh, int64_t ld_dh, float* __restrict__ z, int64_t ld_z,
                         int relu_mask, int64_t m, const int64_t* __restrict__ m_dev, int n,
                         float* __restrict__ dz, int64_t ld_dz, int clear_dh) {
  l2norm_relu_bwd_z_body<NV>(blockIdx.x, dh, ld_dh, z, ld_z, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh);
}

// The same pass with the NEXT layer's leftovers riding in front of it (the engine's supervised
// step: the class-logit layer's forward launch has already sent that layer's input gradient into
// dh, logit.hip): workgroups 0 .. rider-1 form that layer's two weight gradients
// (skinny_dweight_body: dw_l += dz2^T agg2, dw_r += dz2^T h_root2) and workgroup `rider` the mean
// of its row losses in a fixed order; the rest is the pass above.  Nothing a rider reads is
// written by this launch.
struct NormDwRider {
  const float* dz2;
  int64_t ld_dz2;
  SkinnyDwProblem p0, p1;
  int slices, m_chunk, gx, gy, n2;
  int64_t m2;
  const int64_t* m2_dev;
  const float* row_loss;
  float* loss;
};
template <int NV>
__global__ void __launch_bounds__(kBlock)
l2norm_relu_bwd_z_dw_kernel(float* __restrict__ dh, int64_t ld_dh, float* __restrict__ z, int64_t ld_z,
                            int relu_mask, int64_t m, const int64_t* __restrict__ m_dev, int n,
                            float* __restrict__ dz, int64_t ld_dz, int clear_dh, int rider, NormDwRider d) {
  static_assert(kBlock == kSkinnyThreads, "skinny_dweight_body runs on 256 threads");
  const int b = blockIdx.x;
  if (b < rider) {      // (uniform over the workgroup)
    skinny_dweight_body(b % d.gx, b / d.gx % d.gy, b / (d.gx * d.gy), d.dz2, d.ld_dz2, d.p0, d.p1, d.slices, d.m2,
                        d.m2_dev, d.n2, d.m_chunk, nullptr);
    return;
  }
  if (b == rider) {
    if (d.loss != nullptr && threadIdx.x < kWave) {
      // loss = mean(row_loss[0 : m2]): lane l adds rows l, l + 64, ... in order, then the butterfly
      float sl = 0.f;
      for (int64_t i = threadIdx.x; i < d.m2; i += kWave) sl += d.row_loss[i];
      sl = wave_sum(sl);
      if (threadIdx.x == 0) d.loss[0] = sl / static_cast<float>(d.m2);
    }
    return;
  }
  l2norm_relu_bwd_z_body<NV>(b - rider - 1, dh, ld_dh, z, ld_z, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh);
}

// One workgroup of 16 waves walks all rows; per-wave partial losses are combined in wave
// order, so the loss is bitwise reproducible.
constexpr int kXentBlock = 1024;
__global__ void __launch_bounds__(kXentBlock)
softmax_xent_kernel(const float* __restrict__ logits, int64_t ld_logits,
                    const int64_t* __restrict__ labels, const int64_t* __restrict__ label_rows,
                    int64_t m, int n, float grad_scale, float* __restrict__ loss,
                    float* __restrict__ dlogits, int64_t ld_dlogits) {
  __shared__ float partial[kXentBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  constexpr int nwaves = kXentBlock / kWave;
  const float inv_m = 1.f / static_cast<float>(m);
  float wave_loss = 0.f;
  if (n <= kWave) {
    // one logit per lane; R rows per iteration so that R independent load chains
    // (label_rows -> labels, logits) are in flight at once
    constexpr int R = 8;
    for (int64_t r0 = wave; r0 < m; r0 += static_cast<int64_t>(nwaves) * R) {
      float v[R];
      int64_t lab[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int64_t r = r0 + static_cast<int64_t>(i) * nwaves;
        const int64_t rc = r < m ? r : m - 1;
        v[i] = lane < n ? logits[rc * ld_logits + lane] : -INFINITY;
        lab[i] = label_rows != nullptr ? label_rows[rc] : rc;
      }
#pragma unroll
      for (int i = 0; i < R; ++i) lab[i] = labels[lab[i]];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int64_t r = r0 + static_cast<int64_t>(i) * nwaves;
        if (r >= m) break;
        const float mx = wave_max(v[i]);
        const float ex = lane < n ? expf(v[i] - mx) : 0.f;
        const float lse = mx + logf(wave_sum(ex));
        const float at_label = wave_sum(lane == lab[i] ? v[i] : 0.f);
        if (lab[i] >= 0 && lab[i] < n) wave_loss += lse - at_label;
        if (dlogits != nullptr && lane < n) {
          float p = expf(v[i] - lse);
          if (lane == lab[i]) p -= 1.f;
          dlogits[r * ld_dlogits + lane] = p * grad_scale * inv_m;
        }
      }
    }
  } else {
    for (int64_t r = wave; r < m; r += nwaves) {
      const float* lr = logits + r * ld_logits;
      const int64_t lab = labels[label_rows != nullptr ? label_rows[r] : r];
      float mx = -INFINITY;
      for (int f = lane; f < n; f += kWave) mx = fmaxf(mx, lr[f]);
      mx = wave_max(mx);
      float se = 0.f;
      for (int f = lane; f < n; f += kWave) se += expf(lr[f] - mx);
      se = wave_sum(se);
      const float lse = mx + logf(se);
      if (lab >= 0 && lab < n) wave_loss += lse - lr[lab];
      if (dlogits != nullptr) {
        for (int f = lane; f < n; f += kWave) {
          float p = expf(lr[f] - lse);
          if (f == lab) p -= 1.f;
          dlogits[r * ld_dlogits + f] = p * grad_scale * inv_m;
        }
      }
    }
  }
  if (lane == 0) partial[wave] = wave_loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < nwaves; ++w) s += partial[w];
    loss[0] = s * inv_m;
  }
}

// Last layer, fused: y = normalize(z); loss_r = CE(y, label); dz = d loss / d z.  Everything is
// row-local, so one wave per row replaces l2norm_fwd + softmax_xent + l2norm_bwd (and the
// single-workgroup loss kernel's serial walk over the rows).
__global__ void __launch_bounds__(kBlock)
logits_xent_fused_kernel(float* __restrict__ z, int64_t ld_z,
                         const int64_t* __restrict__ labels, const int64_t* __restrict__ label_rows,
                         int64_t m, int n, float grad_scale, float* __restrict__ y, int64_t ld_y,
                         float* __restrict__ inv_norm, float* __restrict__ dz, int64_t ld_dz,
                         float* __restrict__ row_loss, int clear_z) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (r >= m) return;
  const int64_t lab = labels[label_rows != nullptr ? label_rows[r] : r];
  float* zr = z + r * ld_z;
  float ss = 0.f;
  for (int f = lane; f < n; f += kWave) {
    const float v = zr[f];
    ss += v * v;
  }
  ss = wave_sum(ss);
  const float denom = fmaxf(sqrtf(ss), kNormEps);
  const float inv = 1.f / denom;
  const bool clamped = inv >= 1.f / kNormEps;
  float mx = -INFINITY;
  for (int f = lane; f < n; f += kWave) mx = fmaxf(mx, zr[f] / denom);
  mx = wave_max(mx);
  float se = 0.f;
  for (int f = lane; f < n; f += kWave) se += expf(zr[f] / denom - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  const float scale = grad_scale / static_cast<float>(m);
  float dot = 0.f, at_label = 0.f;
  for (int f = lane; f < n; f += kWave) {
    const float yv = zr[f] / denom;
    float g = expf(yv - lse);
    if (f == lab) {
      g -= 1.f;
      at_label = yv;
    }
    dot += yv * g * scale;
  }
  dot = clamped ? 0.f : wave_sum(dot);
  at_label = wave_sum(at_label);
  if (lane == 0) {
    if (inv_norm != nullptr) inv_norm[r] = inv;
    row_loss[r] = (lab >= 0 && lab < n) ? lse - at_label : 0.f;
  }
  for (int f = lane; f < n; f += kWave) {
    const float yv = zr[f] / denom;
    float g = expf(yv - lse);
    if (f == lab) g -= 1.f;
    g *= scale;
    if (y != nullptr) y[r * ld_y + f] = yv;
    dz[r * ld_dz + f] = (g - yv * dot) * inv;
    if (clear_z) zr[f] = 0.f;   // the producer accumulates split-K partial sums into zeros
  }
}

// The multi-label form of the kernel above (PPI: 121 independent sigmoid outputs per node,
// reference examples/ppi_graphsage_supervised.py:30 — BCEWithLogitsLoss on the normalised
// logits): y = normalize(z); loss = mean over the m x n elements of
// max(y, 0) - y t + log(1 + exp(-|y|)); dz = d loss / d z.  row_loss[r] = the row's sum / n, so
// that the mean over rows is the loss.  row_stats (optional, int64 [m, 3]): per row the counts
// (true positive, false positive, false negative) of the prediction y > 0 — what micro-F1 is made
// of (graph_level_graphsage_trainers.py:57-58: f1_score(y, out > 0, average='micro')).
// dz == NULL: forward only.
__global__ void __launch_bounds__(kBlock)
logits_bce_fused_kernel(float* __restrict__ z, int64_t ld_z, const float* __restrict__ targets, int64_t ld_t,
                        const int64_t* __restrict__ label_rows, int64_t m, int n, float grad_scale,
                        float* __restrict__ y, int64_t ld_y, float* __restrict__ inv_norm,
                        float* __restrict__ dz, int64_t ld_dz, float* __restrict__ row_loss,
                        int64_t* __restrict__ row_stats, int clear_z) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (r >= m) return;
  const float* tr = targets + (label_rows != nullptr ? label_rows[r] : r) * ld_t;
  float* zr = z + r * ld_z;
  float ss = 0.f;
  for (int f = lane; f < n; f += kWave) {
    const float v = zr[f];
    ss += v * v;
  }
  ss = wave_sum(ss);
  const float denom = fmaxf(sqrtf(ss), kNormEps);
  const float inv = 1.f / denom;
  const bool clamped = inv >= 1.f / kNormEps;
  const float scale = grad_scale / (static_cast<float>(m) * static_cast<float>(n));
  float dot = 0.f, lsum = 0.f;
  int tp = 0, fp = 0, fn = 0;
  for (int f = lane; f < n; f += kWave) {
    const float yv = zr[f] / denom;
    const float t = tr[f];
    lsum += fmaxf(yv, 0.f) - yv * t + log1pf(expf(-fabsf(yv)));
    const float g = (1.f / (1.f + expf(-yv)) - t) * scale;
    dot += yv * g;
    const bool p = yv > 0.f, q = t > 0.5f;
    tp += (p && q) ? 1 : 0;
    fp += (p && !q) ? 1 : 0;
    fn += (!p && q) ? 1 : 0;
  }
  dot = clamped ? 0.f : wave_sum(dot);
  lsum = wave_sum(lsum);
  if (row_stats != nullptr) {
    const float ftp = wave_sum(static_cast<float>(tp)), ffp = wave_sum(static_cast<float>(fp)),
                ffn = wave_sum(static_cast<float>(fn));      // (exact: counts <= n < 2^24)
    if (lane == 0) {
      row_stats[r * 3 + 0] = static_cast<int64_t>(ftp);
      row_stats[r * 3 + 1] = static_cast<int64_t>(ffp);
      row_stats[r * 3 + 2] = static_cast<int64_t>(ffn);
    }
  }
  if (lane == 0) {
    if (inv_norm != nullptr) inv_norm[r] = inv;
    if (row_loss != nullptr) row_loss[r] = lsum / static_cast<float>(n);
  }
  for (int f = lane; f < n; f += kWave) {
    const float yv = zr[f] / denom;
    if (y != nullptr) y[r * ld_y + f] = yv;
    if (dz != nullptr) {
      const float g = (1.f / (1.f + expf(-yv)) - tr[f]) * scale;
      dz[r * ld_dz + f] = (g - yv * dot) * inv;
    }
    if (clear_z) zr[f] = 0.f;
  }
}

// Cross-entropy of m rows with one WAVE PER ROW over as many workgroups as that takes; the row
// losses go to a caller buffer and mean_reduce_kernel sums them in a fixed order (the
// single-workgroup kernel above walks 512 rows in 35 us: 32 dependent reductions per wave).
__global__ void __launch_bounds__(kBlock)
xent_rows_kernel(const float* __restrict__ logits, int64_t ld_logits, const int64_t* __restrict__ labels,
                 const int64_t* __restrict__ label_rows, int64_t m, int n, float grad_scale,
                 float* __restrict__ row_loss, float* __restrict__ dlogits, int64_t ld_dlogits) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (r >= m) return;
  const int64_t lab = labels[label_rows != nullptr ? label_rows[r] : r];
  const float* lr = logits + r * ld_logits;
  float mx = -INFINITY;
  for (int f = lane; f < n; f += kWave) mx = fmaxf(mx, lr[f]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int f = lane; f < n; f += kWave) se += expf(lr[f] - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  if (lane == 0) row_loss[r] = (lab >= 0 && lab < n) ? lse - lr[lab] : 0.f;
  if (dlogits != nullptr) {
    const float scale = grad_scale / static_cast<float>(m);
    for (int f = lane; f < n; f += kWave) {
      float p = expf(lr[f] - lse);
      if (f == lab) p -= 1.f;
      dlogits[r * ld_dlogits + f] = p * scale;
    }
  }
}

// loss = mean(row_loss[0:m]) — one workgroup, fixed order: bitwise reproducible
__global__ void __launch_bounds__(kBlock)
mean_reduce_kernel(const float* __restrict__ v, int64_t m, float* __restrict__ out) {
  __shared__ float partial[kBlock];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < m; i += kBlock) s += v[i];
  partial[threadIdx.x] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) partial[threadIdx.x] += partial[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = partial[0] / static_cast<float>(m);
}

// Unsupervised GraphSAGE loss on row-normalised embeddings (reference trainers/base_trainers.py:
// 179-181): loss = -mean logsigmoid(<out,pos>) - mean logsigmoid(-<out,neg>), and its gradient.
// One wave per row.  d_out is ACCUMULATED (it already holds the supervised term's gradient).
__device__ inline float log_sigmoid(float x) {   // stable: min(x,0) - log1p(exp(-|x|))
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}
__global__ void __launch_bounds__(kBlock)
unsup_loss_kernel(const float* __restrict__ out, const float* __restrict__ pos,
                  const float* __restrict__ neg, int64_t ld, int64_t m, int n, float grad_scale,
                  float* __restrict__ d_out, float* __restrict__ d_pos, float* __restrict__ d_neg,
                  int64_t ld_d, float* __restrict__ row_loss) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (r >= m) return;
  float sp = 0.f, sn = 0.f;
  for (int f = lane; f < n; f += kWave) {
    const float o = out[r * ld + f];
    sp += o * pos[r * ld + f];
    sn += o * neg[r * ld + f];
  }
  sp = wave_sum(sp);
  sn = wave_sum(sn);
  if (lane == 0) row_loss[r] = -log_sigmoid(sp) - log_sigmoid(-sn);
  // d/ds_p of -logsigmoid(s_p) = -sigmoid(-s_p);  d/ds_n of -logsigmoid(-s_n) = sigmoid(s_n)
  const float scale = grad_scale / static_cast<float>(m);
  const float gp = -scale / (1.f + expf(sp));
  const float gn = scale / (1.f + expf(-sn));
  for (int f = lane; f < n; f += kWave) {
    const float o = out[r * ld + f], pv = pos[r * ld + f], nv = neg[r * ld + f];
    d_out[r * ld_d + f] += gp * pv + gn * nv;
    d_pos[r * ld_d + f] = gp * o;
    d_neg[r * ld_d + f] = gn * o;
  }
}

__global__ void adam_kernel(float* __restrict__ param, const float* grad,
                            float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                            int64_t numel, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float grad_scale,
                            const int64_t* __restrict__ step_in, int64_t* __restrict__ step_out,
                            int zero_grad) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t step = *step_in;
  // step_out is a different word than step_in (else it is NULL here and a second kernel
  // bumps the counter): no thread reads what this one writes
  if (i == 0 && step_out != nullptr) *step_out = step + 1;
  if (i >= numel) return;
  const float t = static_cast<float>(step + 1);
  const float bc1 = 1.f - powf(beta1, t);
  const float bc2 = 1.f - powf(beta2, t);
  float g = grad[i] * grad_scale;
  const float p = param[i];
  if (weight_decay != 0.f) g += weight_decay * p;
  const float m1 = beta1 * exp_avg[i] + (1.f - beta1) * g;
  const float v1 = beta2 * exp_avg_sq[i] + (1.f - beta2) * g * g;
  exp_avg[i] = m1;
  exp_avg_sq[i] = v1;
  const float denom = sqrtf(v1) / sqrtf(bc2) + eps;
  param[i] = p - (lr / bc1) * (m1 / denom);
  if (zero_grad) const_cast<float*>(grad)[i] = 0.f;
}

// The same update where the gradient of one contiguous region [off, off + len) of the flat buffer
// comes as `parts` partial sums ws[s * stride + (i - off)], s < parts (sage_linear_dweight2_parts:
// the split-K slices of the layer-0 weight gradients, stored instead of added): summed here in slice
// order, grad[i] is neither read nor written for those elements.
__global__ void __launch_bounds__(kBlock)
adam_parts_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ exp_avg,
                  float* __restrict__ exp_avg_sq, int64_t numel, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float grad_scale, const int64_t* __restrict__ step_in,
                  int64_t* __restrict__ step_out, int zero_grad, float* __restrict__ ws, int64_t stride,
                  int parts, int64_t off, int64_t len) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t step = *step_in;
  if (i == 0 && step_out != nullptr) *step_out = step + 1;
  if (i >= numel) return;
  float g;
  const int64_t r = i - off;
  if (r >= 0 && r < len) {
    float* wp = ws + r;
    g = 0.f;
    int sidx = 0;
    for (; sidx + 4 <= parts; sidx += 4) {      // four slices in flight
      const float a0 = wp[(sidx + 0) * stride], a1 = wp[(sidx + 1) * stride], a2 = wp[(sidx + 2) * stride],
                  a3 = wp[(sidx + 3) * stride];
      g += a0; g += a1; g += a2; g += a3;
    }
    for (; sidx < parts; ++sidx) g += wp[sidx * stride];
    if (parts == 1 && zero_grad) wp[0] = 0.f;      // (one part: the producer ADDS into it, as into grad)
  } else {
    g = grad[i];
    if (zero_grad) grad[i] = 0.f;
  }
  g *= grad_scale;
  const float t = static_cast<float>(step + 1);
  const float bc1 = 1.f - powf(beta1, t);
  const float bc2 = 1.f - powf(beta2, t);
  const float p = param[i];
  if (weight_decay != 0.f) g += weight_decay * p;
  const float m1 = beta1 * exp_avg[i] + (1.f - beta1) * g;
  const float v1 = beta2 * exp_avg_sq[i] + (1.f - beta2) * g * g;
  exp_avg[i] = m1;
  exp_avg_sq[i] = v1;
  const float denom = sqrtf(v1) / sqrtf(bc2) + eps;
  param[i] = p - (lr / bc1) * (m1 / denom);
}

// ... four elements per thread (everything a multiple of 4 floats and 16-byte aligned): the 25
// slices of the headline's layer-0 weight gradients are 31 MB of reads, 16 bytes per lane and load
__global__ void __launch_bounds__(kBlock)
adam_parts4_kernel(float4* __restrict__ param, float4* __restrict__ grad, float4* __restrict__ exp_avg,
                   float4* __restrict__ exp_avg_sq, int64_t numel4, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float grad_scale, const int64_t* __restrict__ step_in,
                   int64_t* __restrict__ step_out, int zero_grad, float4* __restrict__ ws, int64_t stride4,
                   int parts, int64_t off4, int64_t len4) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t step = *step_in;
  if (i == 0 && step_out != nullptr) *step_out = step + 1;
  if (i >= numel4) return;
  float4 g4;
  const int64_t r = i - off4;
  if (r >= 0 && r < len4) {
    float4* wp = ws + r;
    g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](const float4& a) { g4.x += a.x; g4.y += a.y; g4.z += a.z; g4.w += a.w; };
    int sidx = 0;
    for (; sidx + 4 <= parts; sidx += 4) {      // four slices in flight (all of them at once, clamped: 11.0 against 9.1 us)
      const float4 a0 = wp[(sidx + 0) * stride4], a1 = wp[(sidx + 1) * stride4], a2 = wp[(sidx + 2) * stride4],
                   a3 = wp[(sidx + 3) * stride4];
      add(a0); add(a1); add(a2); add(a3);
    }
    for (; sidx < parts; ++sidx) add(wp[sidx * stride4]);
    if (parts == 1 && zero_grad) wp[0] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    g4 = grad[i];
    if (zero_grad) grad[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float t = static_cast<float>(step + 1);
  const float bc1 = 1.f - powf(beta1, t);
  const float bc2 = 1.f - powf(beta2, t);
  const float4 p4 = param[i], m4 = exp_avg[i], v4 = exp_avg_sq[i];
  // (per element the arithmetic of adam_kernel, operation for operation)
  auto one = [&](float gi, float p, float m0, float v0, float& po, float& mo, float& vo) {
    float g = gi * grad_scale;
    if (weight_decay != 0.f) g += weight_decay * p;
    mo = beta1 * m0 + (1.f - beta1) * g;
    vo = beta2 * v0 + (1.f - beta2) * g * g;
    const float denom = sqrtf(vo) / sqrtf(bc2) + eps;
    po = p - (lr / bc1) * (mo / denom);
  };
  float4 po, mo, vo;
  one(g4.x, p4.x, m4.x, v4.x, po.x, mo.x, vo.x);
  one(g4.y, p4.y, m4.y, v4.y, po.y, mo.y, vo.y);
  one(g4.z, p4.z, m4.z, v4.z, po.z, mo.z, vo.z);
  one(g4.w, p4.w, m4.w, v4.w, po.w, mo.w, vo.w);
  exp_avg[i] = mo;
  exp_avg_sq[i] = vo;
  param[i] = po;
}

// The same update, four elements per thread with 16-byte accesses (numel a multiple of 4, all
// four buffers 16-byte aligned: the engine's flat buffers are): the lstm model's 45 M parameters
// are eight streams of 180 MB per step — 479 us one float at a time.  Per element the arithmetic
// is adam_kernel's, operation for operation.
__global__ void __launch_bounds__(kBlock)
adam4_kernel(float4* __restrict__ param, float4* grad, float4* __restrict__ exp_avg,
             float4* __restrict__ exp_avg_sq, int64_t numel4, float lr, float beta1, float beta2, float eps,
             float weight_decay, float grad_scale, const int64_t* __restrict__ step_in,
             int64_t* __restrict__ step_out, int zero_grad) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t step = *step_in;
  if (i == 0 && step_out != nullptr) *step_out = step + 1;
  if (i >= numel4) return;
  const float t = static_cast<float>(step + 1);
  const float bc1 = 1.f - powf(beta1, t);
  const float bc2 = 1.f - powf(beta2, t);
  const float sq_bc2 = sqrtf(bc2), step_size = lr / bc1;
  const float4 g4 = grad[i], p4 = param[i], m4 = exp_avg[i], v4 = exp_avg_sq[i];
  auto one = [&](float gi, float p, float m0, float v0, float& po, float& mo, float& vo) {
    float g = gi * grad_scale;
    if (weight_decay != 0.f) g += weight_decay * p;
    mo = beta1 * m0 + (1.f - beta1) * g;
    vo = beta2 * v0 + (1.f - beta2) * g * g;
    const float denom = sqrtf(vo) / sq_bc2 + eps;
    po = p - step_size * (mo / denom);
  };
  float4 po, mo, vo;
  one(g4.x, p4.x, m4.x, v4.x, po.x, mo.x, vo.x);
  one(g4.y, p4.y, m4.y, v4.y, po.y, mo.y, vo.y);
  one(g4.z, p4.z, m4.z, v4.z, po.z, mo.z, vo.z);
  one(g4.w, p4.w, m4.w, v4.w, po.w, mo.w, vo.w);
  exp_avg[i] = mo;
  exp_avg_sq[i] = vo;
  param[i] = po;
  if (zero_grad) grad[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void bump_step2_kernel(const int64_t* step_in, int64_t* step_out) { *step_out = *step_in + 1; }


// relu backward of the pool Linear (threshold_backward): dz = out > 0 ? dout : 0, one 16-byte
// vector (or one float) per thread; the consumed dout rows are handed back zeroed on request
template <bool V4>
__global__ void __launch_bounds__(kBlock)
relu_bwd_kernel(float* __restrict__ dout, int64_t ld_dout, const float* __restrict__ out, int64_t ld_out,
                int64_t m, const int64_t* __restrict__ m_dev, int n, float* __restrict__ dz, int64_t ld_dz,
                int clear_dout) {
  const int64_t rows = dev_count(m_dev, m);
  const int per_row = V4 ? n / 4 : n;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t r = i / per_row;
  if (r >= rows) return;
  const int c = static_cast<int>(i - r * per_row) * (V4 ? 4 : 1);
  if (V4) {
    float4* gp = reinterpret_cast<float4*>(dout + r * ld_dout + c);
    const float4 g = *gp;
    const float4 o = *reinterpret_cast<const float4*>(out + r * ld_out + c);
    *reinterpret_cast<float4*>(dz + r * ld_dz + c) =
        make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
    if (clear_dout) *gp = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    const float g = dout[r * ld_dout + c];
    dz[r * ld_dz + c] = out[r * ld_out + c] > 0.f ? g : 0.f;
    if (clear_dout) dout[r * ld_dout + c] = 0.f;
  }
}

}  // namespace
}  // namespace sage

using namespace sage;

extern "C" int sage_l2norm_relu_fwd(float* z, sage_i64 ld_z, sage_i64 m,
                                    const sage_i64* m_dev, int n, float* y, sage_i64 ld_y,
                                    float* h, sage_i64 ld_h, float* inv_norm, int clear_z,
                                    void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && n > 0 && ld_z >= n && ld_y >= n && (h == nullptr || ld_h >= n),
               SAGE_EINVAL, "sage_l2norm_relu_fwd: bad shape");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(z && y, SAGE_EINVAL, "sage_l2norm_relu_fwd: NULL pointer argument");
  SAGE_REQUIRE(!clear_z || (y != z && h != z), SAGE_EINVAL,
               "sage_l2norm_relu_fwd: clear_z with an output aliasing z");
  const bool v4 = n % 4 == 0 && ld_z % 4 == 0 && ld_y % 4 == 0 && (h == nullptr || ld_h % 4 == 0) &&
                  vec16(z) && vec16(y) && vec16(h);
  if (v4) {
    l2norm_relu_fwd_kernel<true><<<static_cast<unsigned>(cdiv(m, kWavesPerBlock)), kBlock, 0, stream>>>(
        z, ld_z, m, m_dev, n, y, ld_y, h, ld_h, inv_norm, clear_z);
  } else {
    l2norm_relu_fwd_kernel<false><<<static_cast<unsigned>(cdiv(m, kWavesPerBlock)), kBlock, 0, stream>>>(
        z, ld_z, m, m_dev, n, y, ld_y, h, ld_h, inv_norm, clear_z);
  }
  SAGE_CHECK_LAUNCH("l2norm_relu_fwd_kernel");
  return SAGE_OK;
}

extern "C" int sage_l2norm_relu_bwd(float* dh, sage_i64 ld_dh, const float* y, sage_i64 ld_y,
                                    const float* inv_norm, int relu_mask, sage_i64 m,
                                    const sage_i64* m_dev, int n, float* dz, sage_i64 ld_dz,
                                    int clear_dh, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && n > 0 && ld_dh >= n && ld_y >= n && ld_dz >= n, SAGE_EINVAL,
               "sage_l2norm_relu_bwd: bad shape");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(dh && y && inv_norm && dz, SAGE_EINVAL, "sage_l2norm_relu_bwd: NULL pointer argument");
  const bool v4 = n % 4 == 0 && ld_dh % 4 == 0 && ld_y % 4 == 0 && ld_dz % 4 == 0 && vec16(dh) && vec16(y) &&
                  vec16(dz);
  if (v4) {
    l2norm_relu_bwd_kernel<true><<<static_cast<unsigned>(cdiv(m, kWavesPerBlock)), kBlock, 0, stream>>>(
        dh, ld_dh, y, ld_y, inv_norm, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh);
  } else {
    l2norm_relu_bwd_kernel<false><<<static_cast<unsigned>(cdiv(m, kWavesPerBlock)), kBlock, 0, stream>>>(
        dh, ld_dh, y, ld_y, inv_norm, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh);
  }
  SAGE_CHECK_LAUNCH("l2norm_relu_bwd_kernel");
  return SAGE_OK;
}

extern "C" int sage_l2norm_relu_bwd_z(float* dh, sage_i64 ld_dh, float* z, sage_i64 ld_z, int relu_mask,
                                      sage_i64 m, const sage_i64* m_dev, int n, float* dz, sage_i64 ld_dz,
                                      int clear_dh, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && n > 0 && ld_dh >= n && ld_z >= n && ld_dz >= n, SAGE_EINVAL,
               "sage_l2norm_relu_bwd_z: bad shape");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(dh && z && dz && dz != z && dh != z, SAGE_EINVAL,
               "sage_l2norm_relu_bwd_z: NULL pointer argument or dz / dh aliasing z");
  SAGE_REQUIRE(n % 4 == 0 && n <= 1024 && ld_dh % 4 == 0 && ld_z % 4 == 0 && ld_dz % 4 == 0 && vec16(dh) && vec16(z) &&
                   vec16(dz),
               SAGE_ELIMIT, "sage_l2norm_relu_bwd_z: needs 16-byte aligned rows of at most 1024 floats, a multiple of 4");
  const unsigned grid = static_cast<unsigned>(cdiv(m, kWavesPerBlock));
  if (n <= 256) {
    l2norm_relu_bwd_z_kernel<1><<<grid, kBlock, 0, stream>>>(dh, ld_dh, z, ld_z, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh);
  } else if (n <= 512) {
    l2norm_relu_bwd_z_kernel<2><<<grid, kBlock, 0, stream>>>(dh, ld_dh, z, ld_z, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh);
  } else {
    l2norm_relu_bwd_z_kernel<4><<<grid, kBlock, 0, stream>>>(dh, ld_dh, z, ld_z, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh);
  }
  SAGE_CHECK_LAUNCH("l2norm_relu_bwd_z_kernel");
  return SAGE_OK;
}

extern "C" int sage_l2norm_relu_bwd_z_dw(float* dh, sage_i64 ld_dh, float* z, sage_i64 ld_z, int relu_mask, sage_i64 m,
                                         const sage_i64* m_dev, int n, float* dz, sage_i64 ld_dz, int clear_dh,
                                         const float* dz2, sage_i64 ld_dz2, sage_i64 m2, const sage_i64* m2_dev, int n2,
                                         int h2, const float* agg2, sage_i64 ld_agg2, float* dw_l, sage_i64 ld_dwl,
                                         const float* h_root2, sage_i64 ld_h_root2, float* dw_r, sage_i64 ld_dwr,
                                         const float* row_loss, float* loss, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m > 0 && n > 0 && ld_dh >= n && ld_z >= n && ld_dz >= n, SAGE_EINVAL, "sage_l2norm_relu_bwd_z_dw: bad shape");
  SAGE_REQUIRE(dh && z && dz && dz != z && dh != z, SAGE_EINVAL,
               "sage_l2norm_relu_bwd_z_dw: NULL pointer argument or dz / dh aliasing z");
  SAGE_REQUIRE(n % 4 == 0 && n <= 1024 && ld_dh % 4 == 0 && ld_z % 4 == 0 && ld_dz % 4 == 0 && vec16(dh) && vec16(z) &&
                   vec16(dz),
               SAGE_ELIMIT, "sage_l2norm_relu_bwd_z_dw: needs 16-byte aligned rows of at most 1024 floats, a multiple of 4");
  const bool root = h_root2 != nullptr || dw_r != nullptr;      // (both NULL: a layer without a root weight — gcn)
  SAGE_REQUIRE(dz2 && agg2 && dw_l && m2 > 0 && m2 < (1ll << 31) && n2 > 0 && n2 <= 64 && h2 > 0 && ld_dz2 >= n2 &&
                   ld_agg2 >= h2 && ld_dwl >= h2 && (!root || (h_root2 && dw_r && ld_h_root2 >= h2 && ld_dwr >= h2)),
               SAGE_EINVAL, "sage_l2norm_relu_bwd_z_dw: bad weight-gradient operands (n2 <= 64)");
  SAGE_REQUIRE((loss == nullptr) == (row_loss == nullptr), SAGE_EINVAL,
               "sage_l2norm_relu_bwd_z_dw: pass row_loss and loss together (or neither)");
  SAGE_REQUIRE(dz2 != dz && dz2 != dh && agg2 != dz && (h_root2 == nullptr || h_root2 != dz), SAGE_EINVAL,
               "sage_l2norm_relu_bwd_z_dw: the weight-gradient operands must not be written by this pass");
  // the grid of sage_linear_dweight2's skinny pair: >= 32 rows of m2 per block, <= 16 slices
  NormDwRider d = {};
  d.dz2 = dz2;
  d.ld_dz2 = ld_dz2;
  d.p0 = {agg2, ld_agg2, nullptr, h2, dw_l, ld_dwl};
  d.p1 = {h_root2, ld_h_root2, nullptr, root ? h2 : 0, dw_r, ld_dwr};      // (k = 0: its workgroups return at once)
  int64_t sp = cdiv(m2, 32);
  if (sp > 16) sp = 16;
  if (sp < 1) sp = 1;
  d.m_chunk = static_cast<int>(cdiv(m2, sp));
  d.slices = static_cast<int>(cdiv(m2, d.m_chunk));
  d.gx = static_cast<int>(cdiv(h2, kSkinnyThreads));
  d.gy = static_cast<int>(cdiv(n2, kDwRows));
  d.n2 = n2;
  d.m2 = m2;
  d.m2_dev = m2_dev;
  d.row_loss = row_loss;
  d.loss = loss;
  const int rider = d.gx * d.gy * 2 * d.slices;
  const unsigned grid = static_cast<unsigned>(cdiv(m, kWavesPerBlock)) + rider + 1;
  if (n <= 256) {
    l2norm_relu_bwd_z_dw_kernel<1><<<grid, kBlock, 0, stream>>>(dh, ld_dh, z, ld_z, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh, rider, d);
  } else if (n <= 512) {
    l2norm_relu_bwd_z_dw_kernel<2><<<grid, kBlock, 0, stream>>>(dh, ld_dh, z, ld_z, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh, rider, d);
  } else {
    l2norm_relu_bwd_z_dw_kernel<4><<<grid, kBlock, 0, stream>>>(dh, ld_dh, z, ld_z, relu_mask, m, m_dev, n, dz, ld_dz, clear_dh, rider, d);
  }
  SAGE_CHECK_LAUNCH("l2norm_relu_bwd_z_dw_kernel");
  return SAGE_OK;
}

extern "C" int sage_relu_bwd(float* dout, sage_i64 ld_dout, const float* out, sage_i64 ld_out, sage_i64 m,
                             const sage_i64* m_dev, int n, float* dz, sage_i64 ld_dz, int clear_dout,
                             void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && n > 0 && ld_dout >= n && ld_out >= n && ld_dz >= n, SAGE_EINVAL, "sage_relu_bwd: bad shape");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(dout && out && dz, SAGE_EINVAL, "sage_relu_bwd: NULL pointer argument");
  SAGE_REQUIRE(!clear_dout || dz != dout, SAGE_EINVAL, "sage_relu_bwd: clear_dout excludes dz aliasing dout");
  // 16-byte vectors over the width rounded up to 4 when every row stride covers it (padded rows,
  // e.g. 602 columns in a stride of 608: the pad columns are rewritten with dout's pad values
  // masked by out's — zeros stay zeros)
  const int n4 = (n + 3) / 4 * 4;
  const bool v4 = ld_dout % 4 == 0 && ld_out % 4 == 0 && ld_dz % 4 == 0 && ld_dout >= n4 && ld_out >= n4 &&
                  ld_dz >= n4 && vec16(dout) && vec16(out) && vec16(dz);
  const int64_t per_row = v4 ? n4 / 4 : n;
  const unsigned grid = static_cast<unsigned>(cdiv(m * per_row, kBlock));
  if (v4) {
    relu_bwd_kernel<true><<<grid, kBlock, 0, stream>>>(dout, ld_dout, out, ld_out, m, m_dev, n4, dz, ld_dz, clear_dout);
  } else {
    relu_bwd_kernel<false><<<grid, kBlock, 0, stream>>>(dout, ld_dout, out, ld_out, m, m_dev, n, dz, ld_dz, clear_dout);
  }
  SAGE_CHECK_LAUNCH("relu_bwd_kernel");
  return SAGE_OK;
}

extern "C" int sage_softmax_xent(const float* logits, sage_i64 ld_logits, const sage_i64* labels,
                                 const sage_i64* label_rows, sage_i64 m, int n, float grad_scale,
                                 float* loss, float* dlogits, sage_i64 ld_dlogits, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m > 0 && n > 0 && ld_logits >= n && (dlogits == nullptr || ld_dlogits >= n),
               SAGE_EINVAL, "sage_softmax_xent: bad shape");
  SAGE_REQUIRE(n <= 4096, SAGE_ELIMIT, "sage_softmax_xent: n=%d > 4096", n);
  SAGE_REQUIRE(logits && labels && loss, SAGE_EINVAL, "sage_softmax_xent: NULL pointer argument");
  softmax_xent_kernel<<<1, kXentBlock, 0, stream>>>(logits, ld_logits, labels, label_rows, m, n,
                                                    grad_scale, loss, dlogits, ld_dlogits);
  SAGE_CHECK_LAUNCH("softmax_xent_kernel");
  return SAGE_OK;
}

extern "C" int sage_softmax_xent_rows(const float* logits, sage_i64 ld_logits, const sage_i64* labels,
                                      const sage_i64* label_rows, sage_i64 m, int n, float grad_scale,
                                      float* row_loss, float* loss, float* dlogits, sage_i64 ld_dlogits,
                                      void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m > 0 && n > 0 && ld_logits >= n && (dlogits == nullptr || ld_dlogits >= n),
               SAGE_EINVAL, "sage_softmax_xent_rows: bad shape");
  SAGE_REQUIRE(logits && labels && row_loss && loss, SAGE_EINVAL, "sage_softmax_xent_rows: NULL pointer argument");
  xent_rows_kernel<<<static_cast<unsigned>(cdiv(m, kWavesPerBlock)), kBlock, 0, stream>>>(
      logits, ld_logits, labels, label_rows, m, n, grad_scale, row_loss, dlogits, ld_dlogits);
  SAGE_CHECK_LAUNCH("xent_rows_kernel");
  mean_reduce_kernel<<<1, kBlock, 0, stream>>>(row_loss, m, loss);
  SAGE_CHECK_LAUNCH("mean_reduce_kernel");
  return SAGE_OK;
}

static int adam_step_impl(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                          sage_i64 numel, float lr, float beta1, float beta2, float eps,
                          float weight_decay, float grad_scale, const sage_i64* step_in,
                          sage_i64* step_out, int zero_grad, hipEvent_t done, hipStream_t stream) {
  SAGE_REQUIRE(numel >= 0, SAGE_EINVAL, "sage_adam_step: negative numel");
  SAGE_REQUIRE(step_in != nullptr && step_out != nullptr, SAGE_EINVAL, "sage_adam_step: step pointer is NULL");
  const bool in_place = step_in == step_out;
  if (numel > 0) {
    SAGE_REQUIRE(param && grad && exp_avg && exp_avg_sq, SAGE_EINVAL,
                 "sage_adam_step: NULL pointer argument");
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // (done: attached to the launch as its completion event — the dispatch's own signal, no marker
    // behind it in the queue; only where this launch is the call's last)
    hipEvent_t stop = in_place ? nullptr : done;
    if (numel % 4 == 0 && numel >= (1 << 20) && a16(param) && a16(grad) && a16(exp_avg) && a16(exp_avg_sq)) {
      hipExtLaunchKernelGGL(adam4_kernel, dim3(static_cast<unsigned>(cdiv(numel / 4, kBlock))), dim3(kBlock), 0, stream,
                            nullptr, stop, 0, reinterpret_cast<float4*>(param),
                            reinterpret_cast<float4*>(const_cast<float*>(grad)), reinterpret_cast<float4*>(exp_avg),
                            reinterpret_cast<float4*>(exp_avg_sq), numel / 4, lr, beta1, beta2, eps, weight_decay,
                            grad_scale, step_in, in_place ? nullptr : step_out, zero_grad);
    } else {
      hipExtLaunchKernelGGL(adam_kernel, dim3(static_cast<unsigned>(cdiv(numel, kBlock))), dim3(kBlock), 0, stream,
                            nullptr, stop, 0, param, static_cast<const float*>(grad), exp_avg, exp_avg_sq, numel, lr,
                            beta1, beta2, eps, weight_decay, grad_scale, step_in, in_place ? nullptr : step_out,
                            zero_grad);
    }
    SAGE_CHECK_LAUNCH("adam_kernel");
  }
  if (in_place || numel == 0) {
    hipExtLaunchKernelGGL(bump_step2_kernel, dim3(1), dim3(1), 0, stream, nullptr, done, 0, step_in, step_out);
    SAGE_CHECK_LAUNCH("bump_step_kernel");
  }
  return SAGE_OK;
}

extern "C" int sage_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                              sage_i64 numel, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float grad_scale, const sage_i64* step_in,
                              sage_i64* step_out, int zero_grad, void* stream_) {
  return adam_step_impl(param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, grad_scale, step_in,
                        step_out, zero_grad, nullptr, static_cast<hipStream_t>(stream_));
}

extern "C" int sage_adam_step_parts(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                                    sage_i64 numel, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, float grad_scale, const sage_i64* step_in,
                                    sage_i64* step_out, int zero_grad, float* ws, sage_i64 part_stride,
                                    int parts, sage_i64 region_off, sage_i64 region_len, void* done_event,
                                    void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(numel > 0 && param && grad && exp_avg && exp_avg_sq && step_in && step_out && step_in != step_out,
               SAGE_EINVAL, "sage_adam_step_parts: NULL pointer, empty buffer or step_in == step_out");
  SAGE_REQUIRE(ws && parts >= 1 && region_off >= 0 && region_len > 0 && region_off + region_len <= numel &&
                   part_stride >= region_len,
               SAGE_EINVAL, "sage_adam_step_parts: bad partial-sum region");
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (numel % 4 == 0 && part_stride % 4 == 0 && region_off % 4 == 0 && region_len % 4 == 0 && a16(param) && a16(grad) &&
      a16(exp_avg) && a16(exp_avg_sq) && a16(ws)) {
    hipExtLaunchKernelGGL(adam_parts4_kernel, dim3(static_cast<unsigned>(cdiv(numel / 4, kBlock))), dim3(kBlock), 0, stream,
                          nullptr, static_cast<hipEvent_t>(done_event), 0, reinterpret_cast<float4*>(param),
                          reinterpret_cast<float4*>(grad), reinterpret_cast<float4*>(exp_avg),
                          reinterpret_cast<float4*>(exp_avg_sq), numel / 4, lr, beta1, beta2, eps, weight_decay, grad_scale,
                          step_in, step_out, zero_grad, reinterpret_cast<float4*>(ws), part_stride / 4, parts,
                          region_off / 4, region_len / 4);
  } else {
    hipExtLaunchKernelGGL(adam_parts_kernel, dim3(static_cast<unsigned>(cdiv(numel, kBlock))), dim3(kBlock), 0, stream,
                          nullptr, static_cast<hipEvent_t>(done_event), 0, param, grad, exp_avg, exp_avg_sq, numel, lr,
                          beta1, beta2, eps, weight_decay, grad_scale, step_in, step_out, zero_grad, ws, part_stride, parts,
                          region_off, region_len);
  }
  SAGE_CHECK_LAUNCH("adam_parts_kernel");
  return SAGE_OK;
}

extern "C" int sage_adam_step_signal(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                                     sage_i64 numel, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, float grad_scale, const sage_i64* step_in,
                                     sage_i64* step_out, int zero_grad, void* done_event, void* stream_) {
  SAGE_REQUIRE(done_event != nullptr, SAGE_EINVAL, "sage_adam_step_signal: NULL event");
  return adam_step_impl(param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, grad_scale, step_in,
                        step_out, zero_grad, static_cast<hipEvent_t>(done_event), static_cast<hipStream_t>(stream_));
}

extern "C" int sage_logits_xent_fused(float* z, sage_i64 ld_z, const sage_i64* labels,
                                      const sage_i64* label_rows, sage_i64 m, int n,
                                      float grad_scale, float* y, sage_i64 ld_y, float* inv_norm,
                                      float* dz, sage_i64 ld_dz, float* row_loss, float* loss,
                                      int clear_z, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m > 0 && n > 0 && ld_z >= n && ld_dz >= n && (y == nullptr || ld_y >= n), SAGE_EINVAL,
               "sage_logits_xent_fused: bad shape");
  SAGE_REQUIRE(z && labels && dz && row_loss && loss, SAGE_EINVAL,
               "sage_logits_xent_fused: NULL pointer argument");
  logits_xent_fused_kernel<<<static_cast<unsigned>(cdiv(m, kWavesPerBlock)), kBlock, 0, stream>>>(
      z, ld_z, labels, label_rows, m, n, grad_scale, y, ld_y, inv_norm, dz, ld_dz, row_loss, clear_z);
  SAGE_CHECK_LAUNCH("logits_xent_fused_kernel");
  mean_reduce_kernel<<<1, kBlock, 0, stream>>>(row_loss, m, loss);
  SAGE_CHECK_LAUNCH("mean_reduce_kernel");
  return SAGE_OK;
}

extern "C" int sage_logits_bce_fused(float* z, sage_i64 ld_z, const float* targets, sage_i64 ld_targets,
                                     const sage_i64* label_rows, sage_i64 m, int n, float grad_scale,
                                     float* y, sage_i64 ld_y, float* inv_norm, float* dz, sage_i64 ld_dz,
                                     float* row_loss, float* loss, sage_i64* row_stats, int clear_z,
                                     void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m > 0 && n > 0 && n < (1 << 24) && ld_z >= n && ld_targets >= n && (dz == nullptr || ld_dz >= n) &&
                   (y == nullptr || ld_y >= n),
               SAGE_EINVAL, "sage_logits_bce_fused: bad shape");
  SAGE_REQUIRE(z && targets && (loss == nullptr || row_loss != nullptr), SAGE_EINVAL,
               "sage_logits_bce_fused: NULL pointer argument (loss needs row_loss)");
  logits_bce_fused_kernel<<<static_cast<unsigned>(cdiv(m, kWavesPerBlock)), kBlock, 0, stream>>>(
      z, ld_z, targets, ld_targets, label_rows, m, n, grad_scale, y, ld_y, inv_norm, dz, ld_dz, row_loss, row_stats,
      clear_z);
  SAGE_CHECK_LAUNCH("logits_bce_fused_kernel");
  if (loss != nullptr) {
    mean_reduce_kernel<<<1, kBlock, 0, stream>>>(row_loss, m, loss);
    SAGE_CHECK_LAUNCH("mean_reduce_kernel");
  }
  return SAGE_OK;
}

extern "C" int sage_unsup_loss(const float* out, const float* pos, const float* neg, sage_i64 ld,
                               sage_i64 m, int n, float grad_scale, float* d_out, float* d_pos,
                               float* d_neg, sage_i64 ld_d, float* row_loss, float* loss,
                               void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m > 0 && n > 0 && ld >= n && ld_d >= n, SAGE_EINVAL, "sage_unsup_loss: bad shape");
  SAGE_REQUIRE(out && pos && neg && d_out && d_pos && d_neg && row_loss && loss, SAGE_EINVAL,
               "sage_unsup_loss: NULL pointer argument");
  unsup_loss_kernel<<<static_cast<unsigned>(cdiv(m, kWavesPerBlock)), kBlock, 0, stream>>>(
      out, pos, neg, ld, m, n, grad_scale, d_out, d_pos, d_neg, ld_d, row_loss);
  SAGE_CHECK_LAUNCH("unsup_loss_kernel");
  mean_reduce_kernel<<<1, kBlock, 0, stream>>>(row_loss, m, loss);
  SAGE_CHECK_LAUNCH("mean_reduce_kernel");
  return SAGE_OK;
}
