// (a-2) Multi-hop uniform fan-out neighbour sampling on a device-resident CSC.
//
// Replaces torch.ops.torch_sparse.neighbor_sample<replace=false, directed=true>, called at
// reference samplers/uniform_loader.py:64-72.  The upstream op is a sequential CPU loop
// (hash map for relabelling, one randint per draw).  The design here is parallel and
// deterministic:
//   * one 64-lane wavefront per frontier node; lane t owns Floyd draw t (counter-based
//     Philox, so draws are independent of scheduling); the "already chosen" rule of
//     Robert Floyd's algorithm is resolved across the wave with shuffles + ballots;
//   * output positions come from an exclusive scan of min(deg, k), so the edge order is
//     exactly the sequential op's (frontier order, then draw order);
//   * first-seen relabelling uses a direct-mapped table of num_nodes 64-bit words
//     (cheap in 288 GB of HBM; no hashing, no probing): every sampled neighbour does an
//     atomicMin of its output position, the winners are flagged, and a second scan turns
//     flags into new local ids.  Entries carry a per-call stamp so the table is never
//     cleared.
#include <rocprim/device/device_scan.hpp>

#include <mutex>

#include "common.h"

namespace sage {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxFanout = 2048;

// hop scratch words (int64) snapshotted by count_kernel for the other kernels of the hop
// HI_CTR / HI_STAMP: the call's effective Philox counter and table stamp (host value + the
// optional device control block), resolved once by seed_kernel
enum { HI_NB = 0, HI_EBASE = 1, HI_FB = 2, HI_NF = 3, HI_CTR = 4, HI_STAMP = 5, HI_WORDS = 8 };

__device__ inline uint64_t enc(uint32_t stamp, uint32_t key) {
  return (static_cast<uint64_t>(0xFFFFFFFFu - stamp) << 32) | key;
}

__global__ void zero_words_kernel(int64_t* p, int n) {
  if (threadIdx.x < n) p[threadIdx.x] = 0;
}

__global__ void seed_kernel(const int64_t* __restrict__ seeds, int num_seeds, int64_t num_nodes,
                            uint64_t* __restrict__ table, uint32_t stamp_host, uint64_t ctr_host,
                            const int64_t* __restrict__ ctrl, int64_t* __restrict__ hopinfo,
                            int64_t* __restrict__ node, int64_t* __restrict__ state) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t stamp = stamp_host;
  uint64_t ctr = ctr_host;
  int64_t seed_off = 0;
  if (ctrl != nullptr) {
    ctr += static_cast<uint64_t>(ctrl[SAGE_CTRL_COUNTER]);
    stamp += static_cast<uint32_t>(ctrl[SAGE_CTRL_STAMP]);
    seed_off = ctrl[SAGE_CTRL_SEED_OFFSET];
  }
  if (i == 0) {
    for (int w = 0; w < SAGE_STATE_WORDS; ++w) {
      if (w != SAGE_ST_ERROR) state[w] = 0;
    }
    state[SAGE_ST_NODES] = num_seeds;
    state[SAGE_ST_FRONT_END] = num_seeds;
    state[SAGE_ST_HOP_NODES] = num_seeds;
    hopinfo[HI_CTR] = static_cast<int64_t>(ctr);
    hopinfo[HI_STAMP] = stamp;
  }
  if (i >= num_seeds) return;
  int64_t s = seeds[seed_off + i];
  if (s < 0 || s >= num_nodes) {
    atomicOr(reinterpret_cast<unsigned long long*>(&state[SAGE_ST_ERROR]), 4ull);
    s = 0;
  }
  node[i] = s;
  atomicMin(reinterpret_cast<unsigned long long*>(&table[s]), enc(stamp, static_cast<uint32_t>(i)));
}

// cnt[i] = number of neighbours frontier node i will emit; cnt[fcap] = 0.
__global__ void count_kernel(const int64_t* __restrict__ colptr, const int64_t* __restrict__ node,
                             const int64_t* __restrict__ state, int fanout, int fcap,
                             int* __restrict__ cnt, int64_t* __restrict__ hopinfo) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t fb = state[SAGE_ST_FRONT_BEGIN];
  const int64_t nf = state[SAGE_ST_FRONT_END] - fb;
  if (i == 0) {
    hopinfo[HI_NB] = state[SAGE_ST_NODES];
    hopinfo[HI_EBASE] = state[SAGE_ST_EDGES];
    hopinfo[HI_FB] = fb;
    hopinfo[HI_NF] = nf < fcap ? nf : fcap;
  }
  if (i > fcap) return;
  int c = 0;
  if (i < fcap && i < nf) {
    int64_t w = node[fb + i];
    int64_t deg = colptr[w + 1] - colptr[w];
    c = (fanout < 0 || deg <= fanout) ? static_cast<int>(deg) : fanout;
  }
  cnt[i] = c;
}

// One wave per frontier node.
template <bool BIG>
__global__ void __launch_bounds__(kBlock)
sample_kernel(const int64_t* __restrict__ colptr, const int64_t* __restrict__ row,
              int64_t num_nodes, const int64_t* __restrict__ node,
              const int64_t* __restrict__ hopinfo, const int* __restrict__ off, int fanout,
              int hop, uint32_t seed_lo, uint32_t seed_hi,
              uint64_t* __restrict__ table, int64_t edge_cap,
              int64_t* __restrict__ src, int64_t* __restrict__ dst, int64_t* __restrict__ eid,
              int64_t* __restrict__ seg_ptr, int64_t* __restrict__ state) {
  __shared__ uint32_t sh_chosen[BIG ? kWavesPerBlock * kMaxFanout : 1];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int64_t nf = hopinfo[HI_NF];
  if (i >= nf) return;
  const int64_t nb = hopinfo[HI_NB];
  const int64_t ebase = hopinfo[HI_EBASE];
  const int64_t fb = hopinfo[HI_FB];
  const uint64_t ctr = static_cast<uint64_t>(hopinfo[HI_CTR]);
  const uint32_t ctr_lo = static_cast<uint32_t>(ctr), ctr_hi = static_cast<uint32_t>(ctr >> 32);
  const uint32_t stamp = static_cast<uint32_t>(hopinfo[HI_STAMP]);
  const int64_t t_local = fb + i;
  const int64_t w = node[t_local];
  const int64_t cs = colptr[w];
  const int64_t deg = colptr[w + 1] - cs;
  const int e0 = off[i];
  const int n_out = off[i + 1] - e0;
  const int64_t e_first = ebase + e0;
  if (lane == 0) seg_ptr[t_local] = e_first < edge_cap ? e_first : edge_cap;
  if (n_out == 0) return;
  if (e_first + n_out > edge_cap && lane == 0) {
    atomicOr(reinterpret_cast<unsigned long long*>(&state[SAGE_ST_ERROR]), 2ull);
  }

  auto emit = [&](int t, int64_t offset_in_col) {
    const int64_t e = e_first + t;
    if (e >= edge_cap) return;
    const int64_t pos = cs + offset_in_col;
    int64_t v = row[pos];
    if (v < 0 || v >= num_nodes) {
      atomicOr(reinterpret_cast<unsigned long long*>(&state[SAGE_ST_ERROR]), 8ull);
      v = 0;
    }
    eid[e] = pos;
    src[e] = v;  // global id for now; relabel_kernel rewrites it as a local id
    dst[e] = t_local;
    atomicMin(reinterpret_cast<unsigned long long*>(&table[v]),
              enc(stamp, static_cast<uint32_t>(nb + e0 + t)));
  };

  if (n_out == deg) {  // fan-out < 0 or >= deg: every neighbour, CSC order
    for (int t = lane; t < n_out; t += kWave) emit(t, t);
    return;
  }

  // Robert Floyd: for j = deg-k .. deg-1: r = U[0, j]; if r already chosen take j.
  const int k = n_out;
  const uint32_t base_j = static_cast<uint32_t>(deg - k);
  if (!BIG) {
    const int t = lane;
    uint32_t chosen = 0;
    if (t < k) {
      Philox4 p = philox4x32_10(static_cast<uint32_t>(t >> 2), static_cast<uint32_t>(t_local),
                                ctr_lo, ctr_hi, seed_lo, seed_hi + static_cast<uint32_t>(hop));
      chosen = mulhi_range(philox_word(p, t & 3), base_j + t + 1);
    }
    const uint32_t r = chosen;
    for (int tt = 1; tt < k; ++tt) {
      const uint32_t rt = __shfl(r, tt);
      const bool hit = (lane < tt) && (chosen == rt);
      if (__ballot(hit) != 0ull && lane == tt) chosen = base_j + tt;
    }
    if (t < k) emit(t, chosen);
  } else {
    uint32_t* mine = sh_chosen + wave * kMaxFanout;
    for (int c0 = 0; c0 < k; c0 += kWave) {
      const int t = c0 + lane;
      uint32_t chosen = 0;
      if (t < k) {
        Philox4 p = philox4x32_10(static_cast<uint32_t>(t >> 2), static_cast<uint32_t>(t_local),
                                  ctr_lo, ctr_hi, seed_lo, seed_hi + static_cast<uint32_t>(hop));
        chosen = mulhi_range(philox_word(p, t & 3), base_j + t + 1);
      }
      const uint32_t r = chosen;
      const int lim = (k - c0) < kWave ? (k - c0) : kWave;
      for (int tl = 0; tl < lim; ++tl) {
        const uint32_t rt = __shfl(r, tl);
        bool hit = (lane < tl) && (chosen == rt);
        for (int s = lane; s < c0; s += kWave) hit = hit || (mine[s] == rt);
        if (__ballot(hit) != 0ull && lane == tl) chosen = base_j + c0 + tl;
      }
      if (t < k) {
        mine[t] = chosen;
        emit(t, chosen);
      }
      // the wave's own LDS writes are visible to its later reads after this wait
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// flag[e] = 1 if hop edge e is the first occurrence of a node not seen before; flag[ecap] = 0
__global__ void flag_kernel(const int64_t* __restrict__ hopinfo, const int* __restrict__ off,
                            int fcap, int64_t ecap, int64_t edge_cap,
                            const uint64_t* __restrict__ table, const int64_t* __restrict__ src,
                            int* __restrict__ flag) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e > ecap) return;
  const int64_t nb = hopinfo[HI_NB];
  const int64_t ebase = hopinfo[HI_EBASE];
  int64_t ehop = off[fcap];
  if (ehop > edge_cap - ebase) ehop = edge_cap - ebase;
  int f = 0;
  if (e < ecap && e < ehop) {
    const int64_t v = src[ebase + e];
    f = static_cast<uint32_t>(table[v]) == static_cast<uint32_t>(nb + e);
  }
  flag[e] = f;
}

__global__ void relabel_kernel(const int64_t* __restrict__ hopinfo, const int* __restrict__ off,
                               int fcap, int64_t ecap, int64_t edge_cap, int64_t node_cap,
                               const uint64_t* __restrict__ table, const int* __restrict__ flag,
                               const int* __restrict__ rank, int64_t* __restrict__ src,
                               int64_t* __restrict__ node, int64_t* __restrict__ state) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t nb = hopinfo[HI_NB];
  const int64_t ebase = hopinfo[HI_EBASE];
  int64_t ehop = off[fcap];
  if (ehop > edge_cap - ebase) ehop = edge_cap - ebase;
  if (e >= ecap || e >= ehop) return;
  const int64_t v = src[ebase + e];
  const int64_t key = static_cast<uint32_t>(table[v]);
  int64_t local = key < nb ? key : nb + rank[key - nb];
  if (local >= node_cap) {
    atomicOr(reinterpret_cast<unsigned long long*>(&state[SAGE_ST_ERROR]), 1ull);
    local = 0;
  } else if (flag[e]) {
    node[local] = v;
  }
  src[ebase + e] = local;
}

// table[new node] = its local id; thread 0 advances the state block to the next hop.
__global__ void finalize_kernel(const int64_t* __restrict__ hopinfo, const int* __restrict__ off,
                                int fcap, int64_t ecap, int64_t edge_cap, int64_t node_cap,
                                const int* __restrict__ rank, const int64_t* __restrict__ node,
                                uint64_t* __restrict__ table, int hop,
                                int64_t* __restrict__ state) {
  const uint32_t stamp = static_cast<uint32_t>(hopinfo[HI_STAMP]);
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t nb = hopinfo[HI_NB];
  const int64_t ebase = hopinfo[HI_EBASE];
  int64_t ehop = off[fcap];
  if (ehop > edge_cap - ebase) ehop = edge_cap - ebase;
  int64_t n_new = rank[ecap];
  if (n_new > node_cap - nb) n_new = node_cap - nb;
  if (idx < n_new) {
    const int64_t local = nb + idx;
    table[node[local]] = enc(stamp, static_cast<uint32_t>(local));
  }
  if (idx == 0) {
    state[SAGE_ST_NODES] = nb + n_new;
    state[SAGE_ST_EDGES] = ebase + ehop;
    state[SAGE_ST_FRONT_BEGIN] = hopinfo[HI_FB] + hopinfo[HI_NF];
    state[SAGE_ST_FRONT_END] = nb + n_new;
    state[SAGE_ST_HOP_NODES + hop + 1] = nb + n_new;
    state[SAGE_ST_HOP_EDGES + hop + 1] = ebase + ehop;
  }
}

// Nodes discovered in the last hop are never expanded: their segments are empty.
__global__ void tail_kernel(const int64_t* __restrict__ state, int64_t node_cap,
                            int64_t* __restrict__ seg_ptr) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t > node_cap) return;
  if (t >= state[SAGE_ST_FRONT_BEGIN]) seg_ptr[t] = state[SAGE_ST_EDGES];
}

// negative samples: one uniform node id per slot (reference samplers/utils.py:30-31, torch.randint)
__global__ void random_nodes_kernel(int64_t num_nodes, int64_t n, uint32_t seed_lo, uint32_t seed_hi,
                                    uint32_t ctr_lo, uint32_t ctr_hi, int64_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // counter word 1 is a domain tag ("NE") that separates this stream from the others
  Philox4 p = philox4x32_10(static_cast<uint32_t>(i >> 2), 0x4E450000u + static_cast<uint32_t>(i >> 34),
                            ctr_lo, ctr_hi, seed_lo, seed_hi);
  out[i] = mulhi_range(philox_word(p, static_cast<int>(i & 3)), static_cast<uint32_t>(num_nodes));
}

__global__ void random_neighbor_kernel(const int64_t* __restrict__ colptr,
                                       const int64_t* __restrict__ row, int64_t num_nodes,
                                       const int64_t* __restrict__ start, int64_t n,
                                       const int64_t* __restrict__ n_dev,
                                       uint32_t seed_lo, uint32_t seed_hi, uint32_t ctr_lo,
                                       uint32_t ctr_hi, int64_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= dev_count(n_dev, n)) return;
  const int64_t s = start[i];
  if (s < 0 || s >= num_nodes) {   // never index colptr with an id that is not a node
    out[i] = 0;
    return;
  }
  const int64_t cs = colptr[s];
  const int64_t deg = colptr[s + 1] - cs;
  if (deg == 0) {
    out[i] = s;
    return;
  }
  // counter word 1 is a domain tag that separates this stream from the sampler's
  Philox4 p = philox4x32_10(static_cast<uint32_t>(i >> 2), 0x52570000u + static_cast<uint32_t>(i >> 34),
                            ctr_lo, ctr_hi, seed_lo, seed_hi);
  out[i] = row[cs + mulhi_range(philox_word(p, static_cast<int>(i & 3)), static_cast<uint32_t>(deg))];
}

size_t scan_temp_bytes(size_t n) {
  size_t bytes = 0;
  int* p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, bytes, p, p, 0, n, rocprim::plus<int>(), nullptr);
  return bytes;
}

}  // namespace
}  // namespace sage

using namespace sage;

// Workspace of sage_sample_fanout_static: one array per GPU, resident until the process ends.
// In host code the NAME is a stand-in, not a device address: static_scratch_ptr() resolves the
// address on the current device once and every launch below receives that pointer.
static __device__ __align__(256) unsigned char scratch[SAGE_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};

// Address of `scratch` on the device that is current for the calling thread (each GPU holds its
// own copy at its own address); 0 and an error message on failure.
void* static_scratch_ptr() {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    sage::set_error("sage_sample_fanout_static: hipGetDevice: %s (device %d)",
                    hipGetErrorString(e), dev);
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) {
      sage::set_error("sage_sample_fanout_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
      return nullptr;
    }
    g_scratch_ptr[dev] = p;
  }
  return g_scratch_ptr[dev];
}
}  // namespace

extern "C" size_t sage_sample_workspace_bytes(sage_i64 node_cap, sage_i64 edge_cap) {
  if (node_cap < 0 || edge_cap < 0) return 0;
  size_t n = static_cast<size_t>(node_cap > edge_cap ? node_cap : edge_cap) + 2;
  size_t total = 0;
  total += align_up(sizeof(int64_t) * HI_WORDS, 256);
  total += 2 * align_up(sizeof(int) * (static_cast<size_t>(node_cap) + 2), 256);  // cnt, off
  total += 2 * align_up(sizeof(int) * (static_cast<size_t>(edge_cap) + 2), 256);  // flag, rank
  total += align_up(scan_temp_bytes(n), 256);
  return total + 256;
}

extern "C" int sage_sample_fanout_ctrl(const sage_i64* colptr, const sage_i64* row,
                                       sage_i64 num_nodes, const sage_i64* seeds, int num_seeds,
                                       const int* fanout_host, int hops, uint64_t rng_seed,
                                       uint64_t rng_counter, uint64_t* relabel_table, uint32_t stamp,
                                       const sage_i64* ctrl_dev, sage_i64 node_cap, sage_i64 edge_cap,
                                       sage_i64* node, sage_i64* src, sage_i64* dst, sage_i64* eid,
                                       sage_i64* seg_ptr, sage_i64* state, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(colptr && row && seeds && fanout_host && relabel_table && node && src && dst &&
                   eid && seg_ptr && state && workspace,
               SAGE_EINVAL, "sage_sample_fanout: NULL pointer argument");
  SAGE_REQUIRE(hops > 0 && hops <= SAGE_MAX_HOPS, SAGE_ELIMIT,
               "sage_sample_fanout: hops=%d outside [1,%d]", hops, SAGE_MAX_HOPS);
  SAGE_REQUIRE(num_seeds >= 0 && num_nodes > 0 && num_nodes < (1ll << 32), SAGE_EINVAL,
               "sage_sample_fanout: bad num_seeds=%d / num_nodes=%lld", num_seeds,
               (long long)num_nodes);
  SAGE_REQUIRE(stamp != 0, SAGE_EINVAL, "sage_sample_fanout: stamp must be > 0");
  SAGE_REQUIRE(node_cap >= num_seeds && edge_cap >= 0 && edge_cap < (1ll << 31) &&
                   node_cap < (1ll << 31),
               SAGE_ELIMIT, "sage_sample_fanout: node_cap=%lld / edge_cap=%lld out of range",
               (long long)node_cap, (long long)edge_cap);
  for (int h = 0; h < hops; ++h) {
    SAGE_REQUIRE(fanout_host[h] <= kMaxFanout, SAGE_ELIMIT,
                 "sage_sample_fanout: fan-out %d > limit %d", fanout_host[h], kMaxFanout);
  }
  SAGE_REQUIRE(workspace_bytes >= sage_sample_workspace_bytes(node_cap, edge_cap),
               SAGE_EWORKSPACE, "sage_sample_fanout: workspace too small (%zu < %zu)",
               workspace_bytes, sage_sample_workspace_bytes(node_cap, edge_cap));

  Carver ws(workspace);
  int64_t* hopinfo = ws.take<int64_t>(HI_WORDS);
  int* cnt = ws.take<int>(node_cap + 2);
  int* off = ws.take<int>(node_cap + 2);
  int* flag = ws.take<int>(edge_cap + 2);
  int* rank = ws.take<int>(edge_cap + 2);
  size_t scan_bytes = scan_temp_bytes(static_cast<size_t>(node_cap > edge_cap ? node_cap : edge_cap) + 2);
  void* scan_tmp = ws.take<char>(scan_bytes);

  const uint32_t seed_lo = static_cast<uint32_t>(rng_seed), seed_hi = static_cast<uint32_t>(rng_seed >> 32);

  // a kernel, not hipMemsetAsync: a small memset node replayed from a HIP graph was observed to
  // write a stale fill value on ROCm 7.2 (0x20.. instead of 0)
  zero_words_kernel<<<1, 64, 0, stream>>>(&state[SAGE_ST_ERROR], 1);
  SAGE_CHECK_LAUNCH("zero_words_kernel");
  {
    int grid = static_cast<int>(cdiv(num_seeds > 0 ? num_seeds : 1, kBlock));
    seed_kernel<<<grid, kBlock, 0, stream>>>(seeds, num_seeds, num_nodes, relabel_table, stamp,
                                             rng_counter, ctrl_dev, hopinfo, node, state);
    SAGE_CHECK_LAUNCH("seed_kernel");
  }

  int64_t fcap = num_seeds;
  for (int h = 0; h < hops; ++h) {
    const int k = fanout_host[h];
    int64_t ecap = edge_cap;
    if (k >= 0 && fcap * static_cast<int64_t>(k) < ecap) ecap = fcap * static_cast<int64_t>(k);

    count_kernel<<<static_cast<int>(cdiv(fcap + 1, kBlock)), kBlock, 0, stream>>>(
        colptr, node, state, k, static_cast<int>(fcap), cnt, hopinfo);
    SAGE_CHECK_LAUNCH("count_kernel");
    size_t tb = scan_bytes;
    SAGE_CHECK_HIP(rocprim::exclusive_scan(scan_tmp, tb, cnt, off, 0, static_cast<size_t>(fcap + 1),
                                           rocprim::plus<int>(), stream),
                   "scan(counts)");
    if (fcap > 0) {
      int grid = static_cast<int>(cdiv(fcap, kWavesPerBlock));
      if (k > kWave) {
        sample_kernel<true><<<grid, kBlock, 0, stream>>>(
            colptr, row, num_nodes, node, hopinfo, off, k, h, seed_lo, seed_hi,
            relabel_table, edge_cap, src, dst, eid, seg_ptr, state);
      } else {
        sample_kernel<false><<<grid, kBlock, 0, stream>>>(
            colptr, row, num_nodes, node, hopinfo, off, k, h, seed_lo, seed_hi,
            relabel_table, edge_cap, src, dst, eid, seg_ptr, state);
      }
      SAGE_CHECK_LAUNCH("sample_kernel");
    }
    if (ecap > 0) {
      flag_kernel<<<static_cast<int>(cdiv(ecap + 1, kBlock)), kBlock, 0, stream>>>(
          hopinfo, off, static_cast<int>(fcap), ecap, edge_cap, relabel_table, src, flag);
      SAGE_CHECK_LAUNCH("flag_kernel");
      tb = scan_bytes;
      SAGE_CHECK_HIP(rocprim::exclusive_scan(scan_tmp, tb, flag, rank, 0, static_cast<size_t>(ecap + 1),
                                             rocprim::plus<int>(), stream),
                     "scan(flags)");
      relabel_kernel<<<static_cast<int>(cdiv(ecap, kBlock)), kBlock, 0, stream>>>(
          hopinfo, off, static_cast<int>(fcap), ecap, edge_cap, node_cap, relabel_table, flag, rank,
          src, node, state);
      SAGE_CHECK_LAUNCH("relabel_kernel");
    } else {
      zero_words_kernel<<<1, 64, 0, stream>>>(reinterpret_cast<int64_t*>(rank), 1);  // rank[0] = 0
      SAGE_CHECK_LAUNCH("zero_words_kernel");
    }
    finalize_kernel<<<static_cast<int>(cdiv(ecap > 0 ? ecap : 1, kBlock)), kBlock, 0, stream>>>(
        hopinfo, off, static_cast<int>(fcap), ecap, edge_cap, node_cap, rank, node, relabel_table,
        h, state);
    SAGE_CHECK_LAUNCH("finalize_kernel");

    // frontier of the next hop = nodes first seen in this one
    fcap = ecap < node_cap ? ecap : node_cap;
  }
  tail_kernel<<<static_cast<int>(cdiv(node_cap + 1, kBlock)), kBlock, 0, stream>>>(state, node_cap,
                                                                                    seg_ptr);
  SAGE_CHECK_LAUNCH("tail_kernel");
  return SAGE_OK;
}

extern "C" int sage_sample_fanout(const sage_i64* colptr, const sage_i64* row, sage_i64 num_nodes,
                                  const sage_i64* seeds, int num_seeds, const int* fanout_host,
                                  int hops, uint64_t rng_seed, uint64_t rng_counter,
                                  uint64_t* relabel_table, uint32_t stamp, sage_i64 node_cap,
                                  sage_i64 edge_cap, sage_i64* node, sage_i64* src, sage_i64* dst,
                                  sage_i64* eid, sage_i64* seg_ptr, sage_i64* state,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  return sage_sample_fanout_ctrl(colptr, row, num_nodes, seeds, num_seeds, fanout_host, hops,
                                 rng_seed, rng_counter, relabel_table, stamp, nullptr, node_cap,
                                 edge_cap, node, src, dst, eid, seg_ptr, state, workspace,
                                 workspace_bytes, stream);
}

namespace {
__global__ void state_accumulate_kernel(const int64_t* __restrict__ state, int64_t* __restrict__ sum) {
  sum[threadIdx.x] += state[threadIdx.x];
}

__global__ void ctrl_advance_kernel(int64_t* ctrl, int64_t d_counter, int64_t d_stamp,
                                    int64_t d_offset, int64_t offset_wrap) {
  ctrl[SAGE_CTRL_COUNTER] += d_counter;
  ctrl[SAGE_CTRL_STAMP] += d_stamp;
  int64_t o = ctrl[SAGE_CTRL_SEED_OFFSET] + d_offset;
  if (offset_wrap > 0 && o + d_offset > offset_wrap) o = 0;
  ctrl[SAGE_CTRL_SEED_OFFSET] = o;
}
}  // namespace

extern "C" int sage_ctrl_advance(sage_i64* ctrl_dev, sage_i64 d_counter, sage_i64 d_stamp,
                                 sage_i64 d_offset, sage_i64 offset_wrap, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(ctrl_dev != nullptr, SAGE_EINVAL, "sage_ctrl_advance: ctrl_dev is NULL");
  ctrl_advance_kernel<<<1, 1, 0, stream>>>(ctrl_dev, d_counter, d_stamp, d_offset, offset_wrap);
  SAGE_CHECK_LAUNCH("ctrl_advance_kernel");
  return SAGE_OK;
}

extern "C" int sage_random_neighbor(const sage_i64* colptr, const sage_i64* row, sage_i64 num_nodes,
                                    const sage_i64* start, sage_i64 num_start,
                                    const sage_i64* num_start_dev, uint64_t rng_seed,
                                    uint64_t rng_counter, sage_i64* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(colptr && row && out && (start || num_start == 0), SAGE_EINVAL,
               "sage_random_neighbor: NULL pointer argument");
  SAGE_REQUIRE(num_nodes > 0, SAGE_EINVAL, "sage_random_neighbor: bad num_nodes");
  if (num_start <= 0) return SAGE_OK;
  random_neighbor_kernel<<<static_cast<int>(cdiv(num_start, kBlock)), kBlock, 0, stream>>>(
      colptr, row, num_nodes, start, num_start, num_start_dev, static_cast<uint32_t>(rng_seed),
      static_cast<uint32_t>(rng_seed >> 32), static_cast<uint32_t>(rng_counter),
      static_cast<uint32_t>(rng_counter >> 32), out);
  SAGE_CHECK_LAUNCH("random_neighbor_kernel");
  return SAGE_OK;
}

extern "C" int sage_random_nodes(sage_i64 num_nodes, sage_i64 n, uint64_t rng_seed, uint64_t rng_counter,
                                 sage_i64* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(num_nodes > 0 && num_nodes < (1ll << 31), SAGE_EINVAL, "sage_random_nodes: bad num_nodes");
  SAGE_REQUIRE(out != nullptr || n == 0, SAGE_EINVAL, "sage_random_nodes: NULL pointer argument");
  if (n <= 0) return SAGE_OK;
  random_nodes_kernel<<<static_cast<int>(cdiv(n, kBlock)), kBlock, 0, stream>>>(
      num_nodes, n, static_cast<uint32_t>(rng_seed), static_cast<uint32_t>(rng_seed >> 32),
      static_cast<uint32_t>(rng_counter), static_cast<uint32_t>(rng_counter >> 32), out);
  SAGE_CHECK_LAUNCH("random_nodes_kernel");
  return SAGE_OK;
}

extern "C" int sage_state_accumulate(const sage_i64* state, sage_i64* sum, void* stream) {
  SAGE_REQUIRE(state != nullptr && sum != nullptr, SAGE_EINVAL, "sage_state_accumulate: NULL pointer argument");
  state_accumulate_kernel<<<1, SAGE_STATE_WORDS, 0, static_cast<hipStream_t>(stream)>>>(state, sum);
  SAGE_CHECK_LAUNCH("state_accumulate_kernel");
  return SAGE_OK;
}

extern "C" int sage_sample_fanout_static(const sage_i64* colptr, const sage_i64* row,
                                         sage_i64 num_nodes, const sage_i64* seeds, int num_seeds,
                                         const int* fanout_host, int hops, uint64_t rng_seed,
                                         uint64_t rng_counter, uint64_t* relabel_table,
                                         uint32_t stamp, sage_i64 node_cap, sage_i64 edge_cap,
                                         sage_i64* node, sage_i64* src, sage_i64* dst, sage_i64* eid,
                                         sage_i64* seg_ptr, sage_i64* state, void* stream) {
  SAGE_REQUIRE(node_cap >= 0 && edge_cap >= 0 && node_cap < (1ll << 31) && edge_cap < (1ll << 31),
               SAGE_ELIMIT, "sage_sample_fanout_static: node_cap=%lld / edge_cap=%lld out of range",
               (long long)node_cap, (long long)edge_cap);
  const size_t need = sage_sample_workspace_bytes(node_cap, edge_cap);
  SAGE_REQUIRE(need <= static_cast<size_t>(SAGE_STATIC_SCRATCH_BYTES), SAGE_ELIMIT,
               "sage_sample_fanout_static: node_cap=%lld / edge_cap=%lld need %zu bytes of scratch, the "
               "static array holds %zu: use sage_sample_fanout with a caller workspace",
               (long long)node_cap, (long long)edge_cap, need,
               static_cast<size_t>(SAGE_STATIC_SCRATCH_BYTES));
  void* ws = static_scratch_ptr();
  if (ws == nullptr) return SAGE_ELAUNCH;
  // every word of the workspace that a kernel reads is written earlier in the same call (hop
  // info, counts, offsets, flags, ranks, scan state), so nothing has to be cleared between calls
  return sage_sample_fanout_ctrl(colptr, row, num_nodes, seeds, num_seeds, fanout_host, hops,
                                 rng_seed, rng_counter, relabel_table, stamp, nullptr, node_cap,
                                 edge_cap, node, src, dst, eid, seg_ptr, state, ws,
                                 static_cast<size_t>(SAGE_STATIC_SCRATCH_BYTES), stream);
}
