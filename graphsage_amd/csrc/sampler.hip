// (a-2) Multi-hop uniform fan-out neighbour sampling on a device-resident CSC.
//
// Replaces torch.ops.torch_sparse.neighbor_sample<replace=false, directed=true>, called at
// reference samplers/uniform_loader.py:64-72.  The upstream op is a sequential CPU loop
// (hash map for relabelling, one randint per draw).  The design here is parallel and
// deterministic:
//   * one 64-lane wavefront per frontier node; lane t owns Floyd draw t (counter-based
//     Philox, so draws are independent of scheduling); the "already chosen" rule of
//     Robert Floyd's algorithm is resolved across the wave with shuffles + ballots;
//   * output positions come from an exclusive prefix of min(deg, k) — in-workgroup scan plus one
//     exchange of workgroup totals inside the same launch — so the edge order is exactly the
//     sequential op's (frontier order, then draw order);
//   * first-seen relabelling uses a direct-mapped table of num_nodes 64-bit words
//     (cheap in 288 GB of HBM; no hashing, no probing): every sampled neighbour does an
//     atomicMin of its output position, the winners are flagged, and a second prefix turns
//     flags into new local ids.  Entries carry a per-call stamp so the table is never
//     cleared.
// One call = 1 + 3 x hops launches (7 for the two-hop sampler), no library scans.
#include <mutex>

#include "common.h"

namespace sage {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxFanout = 2048;

// hop scratch words (int64): the pre-hop counters snapshotted by the first workgroup of
// hop_sample_kernel for the other two kernels of the hop (which run after the state block has
// moved on), the hop's edge total, and the call's effective Philox counter and table stamp
// (host value + the optional device control block), resolved once by seed_kernel
enum { HI_NB = 0, HI_EBASE = 1, HI_FB = 2, HI_NF = 3, HI_CTR = 4, HI_STAMP = 5, HI_EHOP = 6, HI_WORDS = 8 };

// Three kernels per hop, each with ONE exchange between workgroups: an exclusive prefix over
// per-workgroup totals (edge counts in hop_sample_kernel, first-occurrence flags in
// hop_flag_kernel).  A workgroup takes a ticket (its logical position: every lower ticket has
// been taken by a workgroup that is already running), publishes its total as one 8-byte word
// {ready bit, total} with an agent-scope store right after counting — before it waits for
// anything — and sums the words of all lower tickets (polling loads, agent scope; <= a few
// hundred words).  seed_kernel clears tickets and words at the start of every call.
// hop kernels 1 and 2 run 1024-thread workgroups, so that few totals are exchanged (hop 2 of the
// 512 x 25/10 sampler: 194 and 125 workgroups).  A frontier node is sampled by a GROUP of G lanes,
// G = 16 / 32 / 64 by the fan-out (k <= G; above 64: one wave with an LDS list), 64 / G nodes per
// wave at once: 1024 / G nodes per workgroup in one pass.
constexpr int kHopBlock = 1024;
constexpr int kHopWaves = kHopBlock / kWave;
constexpr int kMinNodesPerBlock = kHopBlock / kWave;    // G = 64
constexpr unsigned long long kReady = 1ull << 63;

__device__ inline uint64_t enc(uint32_t stamp, uint32_t key) {
  return (static_cast<uint64_t>(0xFFFFFFFFu - stamp) << 32) | key;
}

__device__ inline void publish_total(unsigned long long* word, uint32_t total) {
  __hip_atomic_store(word, kReady | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum of the totals published under tickets 0 .. L-1, by one wave (every lane returns the sum).
// The wait is BOUNDED: the ticket argument (every lower ticket belongs to a workgroup that is
// already resident and publishes before it waits) makes it short — microseconds — but it rests on
// that workgroup making progress; a lane that has polled one word `poll_limit` times (the default
// is of the order of a second) gives up, reports SAGE_ST_ERROR bit 4 and returns what it has, so
// that a broken assumption ends as an error code of the call and not as a grid that never drains.
// What follows then runs on a wrong prefix: every store of the hop kernels is bounds-checked
// against the capacities, so the result is wrong but in bounds, and the error bit says so.
constexpr int kPollLimit = 1 << 20;
__device__ inline uint32_t sum_lower_tickets(unsigned long long* words, int L, int lane, int64_t* state,
                                             int poll_limit) {
  uint32_t sum = 0;
  bool gave_up = false;
  for (int j = lane; j < L; j += kWave) {
    unsigned long long v = __hip_atomic_load(words + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int polls = 0; (v & kReady) == 0 && !gave_up; ++polls) {
      if (polls >= poll_limit) {
        gave_up = true;
        atomicOr(reinterpret_cast<unsigned long long*>(&state[SAGE_ST_ERROR]), 16ull);
        break;
      }
      __builtin_amdgcn_s_sleep(8);     // leave the issue slots and the L2 to the waves that compute
      v = __hip_atomic_load(words + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((v & kReady) != 0) sum += static_cast<uint32_t>(v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  return sum;
}

__global__ void zero_words_kernel(int64_t* p, int n) {
  if (threadIdx.x < n) p[threadIdx.x] = 0;
}

__global__ void seed_kernel(const int64_t* __restrict__ seeds, int num_seeds, int64_t num_nodes,
                            uint64_t* __restrict__ table, uint32_t stamp_host, uint64_t ctr_host,
                            const int64_t* __restrict__ ctrl, int64_t* __restrict__ hopinfo,
                            unsigned long long* __restrict__ exchange, int64_t exchange_words,
                            int64_t* __restrict__ node, int64_t* __restrict__ state) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  uint32_t stamp = stamp_host;
  uint64_t ctr = ctr_host;
  int64_t seed_off = 0;
  if (ctrl != nullptr) {
    ctr += static_cast<uint64_t>(ctrl[SAGE_CTRL_COUNTER]);
    stamp += static_cast<uint32_t>(ctrl[SAGE_CTRL_STAMP]);
    seed_off = ctrl[SAGE_CTRL_SEED_OFFSET];
  }
  if (i < exchange_words) exchange[i] = 0;   // tickets and published totals of every hop
  if (i == 0) {
    for (int w = 0; w < SAGE_STATE_WORDS; ++w) state[w] = 0;
    state[SAGE_ST_NODES] = num_seeds;
    state[SAGE_ST_FRONT_END] = num_seeds;
    state[SAGE_ST_HOP_NODES] = num_seeds;
    hopinfo[HI_CTR] = static_cast<int64_t>(ctr);
    hopinfo[HI_STAMP] = stamp;
  }
  if (i >= num_seeds) return;
  const int64_t s = seeds[seed_off + i];
  if (s < 0 || s >= num_nodes) {
    // not a node: marked here, reported (ST_ERROR bit 2) and replaced by node 0 when hop 0
    // expands it — ST_ERROR is being cleared by thread 0 of this very launch, an OR could be lost
    node[i] = -1;
    return;
  }
  node[i] = s;
  atomicMin(reinterpret_cast<unsigned long long*>(&table[s]), enc(stamp, static_cast<uint32_t>(i)));
}

// One frontier node sampled by a group of G lanes (gl = lane within the group, gbase = first
// lane of the group): n_out = min(deg, k) edges written at e_first ..  Lanes of groups without a
// node run along with n_out = 0.
template <int G, bool BIG>
__device__ inline void sample_node(const int64_t* __restrict__ colptr, const int64_t* __restrict__ row,
                                   int64_t num_nodes, bool have, int64_t w, int64_t t_local, int64_t nb,
                                   int64_t e_first, int e0, int n_out, int fanout, int hop,
                                   uint32_t seed_lo, uint32_t seed_hi, uint32_t ctr_lo, uint32_t ctr_hi,
                                   uint32_t stamp, uint64_t* __restrict__ table, int64_t edge_cap,
                                   int64_t* __restrict__ src, int64_t* __restrict__ dst,
                                   int64_t* __restrict__ eid, int64_t* __restrict__ seg_ptr,
                                   int64_t* __restrict__ state, uint32_t* mine, int gl, int gbase) {
  int64_t cs = 0, deg = 0;
  if (have) {
    cs = colptr[w];
    deg = colptr[w + 1] - cs;
    if (gl == 0) seg_ptr[t_local] = e_first < edge_cap ? e_first : edge_cap;
    if (n_out > 0 && e_first + n_out > edge_cap && gl == 0) {
      atomicOr(reinterpret_cast<unsigned long long*>(&state[SAGE_ST_ERROR]), 2ull);
    }
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
    src[e] = v;  // global id for now; hop_relabel_kernel rewrites it as a local id
    dst[e] = t_local;
    atomicMin(reinterpret_cast<unsigned long long*>(&table[v]),
              enc(stamp, static_cast<uint32_t>(nb + e0 + t)));
  };

  const bool take_all = have && n_out == deg;   // fan-out < 0 or >= deg: every neighbour, CSC order
  if (take_all) {
    for (int t = gl; t < n_out; t += G) emit(t, t);
  }
  // Robert Floyd: for j = deg-k .. deg-1: r = U[0, j]; if r already chosen take j.  A group in
  // this branch has n_out == fanout; the loops below run for the whole wave (shuffles and
  // ballots need every lane), groups that are not drawing carry `floyd` = false.
  const bool floyd = have && !take_all && n_out > 0;
  const int k = fanout;
  const uint32_t base_j = static_cast<uint32_t>(deg - k);
  const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << gbase;
  if (!BIG) {
    if (__ballot(floyd) == 0ull) return;   // wave-uniform
    const int t = gl;
    uint32_t chosen = 0;
    if (floyd && t < k) {
      Philox4 p = philox4x32_10(static_cast<uint32_t>(t >> 2), static_cast<uint32_t>(t_local),
                                ctr_lo, ctr_hi, seed_lo, seed_hi + static_cast<uint32_t>(hop));
      chosen = mulhi_range(philox_word(p, t & 3), base_j + t + 1);
    }
    const uint32_t r = chosen;
    for (int tt = 1; tt < k; ++tt) {
      const uint32_t rt = __shfl(r, gbase + tt);
      const bool hit = floyd && (gl < tt) && (chosen == rt);
      if ((__ballot(hit) & gmask) != 0ull && gl == tt) chosen = base_j + tt;
    }
    if (floyd && t < k) emit(t, chosen);
  } else {   // G == 64, k > 64: the draws so far are kept in LDS
    if (!floyd) return;                     // one node per wave: wave-uniform
    for (int c0 = 0; c0 < k; c0 += kWave) {
      const int t = c0 + gl;
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
        bool hit = (gl < tl) && (chosen == rt);
        for (int q = gl; q < c0; q += kWave) hit = hit || (mine[q] == rt);
        if (__ballot(hit) != 0ull && gl == tl) chosen = base_j + c0 + tl;
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

// Hop kernel 1 of 3: per workgroup 1024 / G frontier nodes — counts min(deg, k), the exclusive
// prefix over all frontier nodes (in-workgroup scan + the ticket exchange), then one lane group
// per node draws the neighbours and claims them in the relabel table.
template <int G, bool BIG>
__global__ void __launch_bounds__(kHopBlock)
hop_sample_kernel(const int64_t* __restrict__ colptr, const int64_t* __restrict__ row,
                  int64_t num_nodes, int64_t* __restrict__ node, int64_t* __restrict__ hopinfo,
                  unsigned long long* __restrict__ ticket, unsigned long long* __restrict__ totals,
                  int fanout, int fcap, int hop, uint32_t seed_lo, uint32_t seed_hi,
                  uint64_t* __restrict__ table, int64_t edge_cap, int64_t* __restrict__ src,
                  int64_t* __restrict__ dst, int64_t* __restrict__ eid, int64_t* __restrict__ seg_ptr,
                  int64_t* __restrict__ state, int poll_limit) {
  constexpr int NPB = kHopBlock / G;          // nodes per workgroup (16, 32 or 64: at most one wave's worth)
  __shared__ uint32_t sh_chosen[BIG ? kHopWaves * kMaxFanout : 1];
  __shared__ int s_L, s_cnt[NPB], s_off[NPB], s_total;
  __shared__ uint32_t s_base;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  if (threadIdx.x == 0) s_L = static_cast<int>(atomicAdd(ticket, 1ull));
  __syncthreads();
  const int L = s_L;
  const int64_t fb = state[SAGE_ST_FRONT_BEGIN];
  int64_t nf = state[SAGE_ST_FRONT_END] - fb;
  if (nf > fcap) nf = fcap;
  const int64_t nb = state[SAGE_ST_NODES];
  const int64_t ebase = state[SAGE_ST_EDGES];
  if (L == 0 && threadIdx.x == 0) {
    hopinfo[HI_NB] = nb;
    hopinfo[HI_EBASE] = ebase;
    hopinfo[HI_FB] = fb;
    hopinfo[HI_NF] = nf;
  }
  const int64_t i0 = static_cast<int64_t>(L) * NPB;
  if (i0 >= nf && L != 0) return;   // (the whole workgroup; no later ticket needs its total)
  if (wave == 0) {
    int c = 0;
    const int64_t i = i0 + lane;
    if (lane < NPB && i < nf) {
      int64_t w = node[fb + i];
      if (w < 0) {   // a seed that is not a node (seed_kernel)
        atomicOr(reinterpret_cast<unsigned long long*>(&state[SAGE_ST_ERROR]), 4ull);
        node[fb + i] = 0;
        w = 0;
      }
      const int64_t deg = colptr[w + 1] - colptr[w];
      c = (fanout < 0 || deg <= fanout) ? static_cast<int>(deg) : fanout;
    }
    int incl = c;     // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    const int total = __shfl(incl, kWave - 1);
    if (lane < NPB) {
      s_cnt[lane] = c;
      s_off[lane] = incl - c;
    }
    if (lane == 0) {
      s_total = total;
      publish_total(totals + L, static_cast<uint32_t>(total));
    }
    const uint32_t base = sum_lower_tickets(totals, L, lane, state, poll_limit);
    if (lane == 0) s_base = base;
  }
  __syncthreads();
  const int64_t base = s_base;
  const bool last = nf > 0 ? (L == (nf - 1) / NPB) : (L == 0);
  if (last && threadIdx.x == 0) {   // edges of the hop, clipped to what fits
    int64_t ehop = base + s_total;
    if (ehop > edge_cap - ebase) ehop = edge_cap - ebase;
    hopinfo[HI_EHOP] = ehop > 0 ? ehop : 0;
  }
  const uint64_t ctr = static_cast<uint64_t>(hopinfo[HI_CTR]);
  const uint32_t ctr_lo = static_cast<uint32_t>(ctr), ctr_hi = static_cast<uint32_t>(ctr >> 32);
  const uint32_t stamp = static_cast<uint32_t>(hopinfo[HI_STAMP]);
  uint32_t* mine = sh_chosen + (BIG ? wave * kMaxFanout : 0);
  const int gl = lane % G, gbase = lane - gl;
  const int li = wave * (kWave / G) + lane / G;     // node of this lane's group
  const int64_t i = i0 + li;
  const bool have = i < nf;
  const int64_t t_local = fb + i;
  int64_t w = have ? node[t_local] : 0;
  if (w < 0) w = 0;
  const int e0 = have ? static_cast<int>(base) + s_off[li] : 0;
  sample_node<G, BIG>(colptr, row, num_nodes, have, w, t_local, nb, ebase + e0, e0, have ? s_cnt[li] : 0,
                      fanout, hop, seed_lo, seed_hi, ctr_lo, ctr_hi, stamp, table, edge_cap, src, dst, eid,
                      seg_ptr, state, mine, gl, gbase);
}

// Hop kernel 2 of 3: an edge is the FIRST occurrence of a new node if the table still holds its
// own position; the exclusive prefix over those flags is the node's rank among the new nodes,
// i.e. its local id.  The winner writes node[local] and replaces the position in the table by
// the local id (every other edge of that node reads it there in kernel 3).  The workgroup that
// holds the hop's last edge advances the state block.
__global__ void __launch_bounds__(kHopBlock)
hop_flag_kernel(const int64_t* __restrict__ hopinfo_c,
                unsigned long long* __restrict__ ticket, unsigned long long* __restrict__ totals,
                int64_t node_cap, uint64_t* __restrict__ table, const int64_t* __restrict__ src,
                int64_t* __restrict__ node, int hop, int64_t* __restrict__ state, int poll_limit) {
  __shared__ int s_L;
  __shared__ uint32_t s_wave[kHopWaves], s_base;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  if (threadIdx.x == 0) s_L = static_cast<int>(atomicAdd(ticket, 1ull));
  __syncthreads();
  const int L = s_L;
  const int64_t nb = hopinfo_c[HI_NB];
  const int64_t ebase = hopinfo_c[HI_EBASE];
  const int64_t ehop = hopinfo_c[HI_EHOP];
  const uint32_t stamp = static_cast<uint32_t>(hopinfo_c[HI_STAMP]);
  const int64_t e0 = static_cast<int64_t>(L) * kHopBlock;
  if (e0 >= ehop && L != 0) return;
  const int64_t e = e0 + threadIdx.x;
  int64_t v = 0;
  bool f = false;
  if (e < ehop) {
    v = src[ebase + e];
    f = static_cast<uint32_t>(table[v]) == static_cast<uint32_t>(nb + e);
  }
  const unsigned long long ball = __ballot(f);
  const uint32_t below = static_cast<uint32_t>(__popcll(ball & ((1ull << lane) - 1ull)));
  if (lane == 0) s_wave[wave] = static_cast<uint32_t>(__popcll(ball));
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int q = 0; q < kHopWaves; ++q) {
    const uint32_t c = s_wave[q];
    if (q < wave) before += c;
    total += c;
  }
  if (wave == 0) {
    if (lane == 0) publish_total(totals + L, total);
    const uint32_t base = sum_lower_tickets(totals, L, lane, state, poll_limit);
    if (lane == 0) s_base = base;
  }
  __syncthreads();
  const int64_t base = s_base;
  if (f) {
    const int64_t local = nb + base + before + below;
    if (local < node_cap) {
      node[local] = v;
      table[v] = enc(stamp, static_cast<uint32_t>(local));
    } else {   // node capacity exceeded: the output is truncated, its edges point at local 0
      atomicOr(reinterpret_cast<unsigned long long*>(&state[SAGE_ST_ERROR]), 1ull);
      table[v] = enc(stamp, 0u);
    }
  }
  const bool last = ehop > 0 ? (L == (ehop - 1) / kHopBlock) : (L == 0);
  if (last && threadIdx.x == 0) {
    int64_t n_new = base + total;
    if (n_new > node_cap - nb) n_new = node_cap - nb;
    state[SAGE_ST_NODES] = nb + n_new;
    state[SAGE_ST_EDGES] = ebase + ehop;
    state[SAGE_ST_FRONT_BEGIN] = hopinfo_c[HI_FB] + hopinfo_c[HI_NF];
    state[SAGE_ST_FRONT_END] = nb + n_new;
    state[SAGE_ST_HOP_NODES + hop + 1] = nb + n_new;
    state[SAGE_ST_HOP_EDGES + hop + 1] = ebase + ehop;
  }
}

// Hop kernel 3 of 3: every edge's source becomes a local id (read from the table).  In the last
// hop the same launch closes the segment pointers: nodes discovered in the last hop are never
// expanded, their segments are empty.
__global__ void __launch_bounds__(kBlock)
hop_relabel_kernel(const int64_t* __restrict__ hopinfo, const uint64_t* __restrict__ table,
                   int64_t* __restrict__ src, int close_segments, int64_t node_cap,
                   const int64_t* __restrict__ state, int64_t* __restrict__ seg_ptr) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t ebase = hopinfo[HI_EBASE];
  const int64_t ehop = hopinfo[HI_EHOP];
  if (e < ehop) {
    const int64_t v = src[ebase + e];
    src[ebase + e] = static_cast<int64_t>(static_cast<uint32_t>(table[v]));
  }
  if (close_segments && e <= node_cap && e >= state[SAGE_ST_FRONT_BEGIN]) seg_ptr[e] = state[SAGE_ST_EDGES];
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

// exchange area: per hop one ticket word per kernel kind + the published totals of its workgroups
struct ExchangeLayout {
  int64_t n_a, n_b;       // workgroups of hop_sample_kernel / hop_flag_kernel (upper bounds)
  int64_t per_hop;        // words per hop: 2 tickets + n_a + n_b
};
ExchangeLayout exchange_layout(int64_t node_cap, int64_t edge_cap) {
  ExchangeLayout x;
  x.n_a = cdiv(node_cap, kMinNodesPerBlock) + 1;
  x.n_b = cdiv(edge_cap, kHopBlock) + 1;
  x.per_hop = 2 + x.n_a + x.n_b;
  return x;
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
  const ExchangeLayout x = exchange_layout(node_cap, edge_cap);
  size_t total = align_up(sizeof(int64_t) * HI_WORDS, 256);
  total += align_up(sizeof(unsigned long long) * static_cast<size_t>(x.per_hop) * SAGE_MAX_HOPS, 256);
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

  const ExchangeLayout x = exchange_layout(node_cap, edge_cap);
  Carver ws(workspace);
  int64_t* hopinfo = ws.take<int64_t>(HI_WORDS);
  unsigned long long* exchange = ws.take<unsigned long long>(static_cast<size_t>(x.per_hop) * SAGE_MAX_HOPS);
  const int64_t exchange_words = x.per_hop * hops;

  const uint32_t seed_lo = static_cast<uint32_t>(rng_seed), seed_hi = static_cast<uint32_t>(rng_seed >> 32);

  // 1 + 3 * hops launches, kernels only (no memset nodes: a small memset replayed from a HIP
  // graph was observed to write a stale fill value on ROCm 7.2)
  {
    const int64_t work = exchange_words > num_seeds ? exchange_words : (num_seeds > 0 ? num_seeds : 1);
    seed_kernel<<<static_cast<int>(cdiv(work, kBlock)), kBlock, 0, stream>>>(
        seeds, num_seeds, num_nodes, relabel_table, stamp, rng_counter, ctrl_dev, hopinfo, exchange,
        exchange_words, node, state);
    SAGE_CHECK_LAUNCH("seed_kernel");
  }

  int64_t fcap = num_seeds;
  for (int h = 0; h < hops; ++h) {
    const int k = fanout_host[h];
    int64_t ecap = edge_cap;
    if (k >= 0 && fcap * static_cast<int64_t>(k) < ecap) ecap = fcap * static_cast<int64_t>(k);
    unsigned long long* ticket_a = exchange + x.per_hop * h;
    unsigned long long* ticket_b = ticket_a + 1;
    unsigned long long* totals_a = ticket_a + 2;
    unsigned long long* totals_b = totals_a + x.n_a;
    const bool last = h == hops - 1;

    {
      // lane-group width by the fan-out: k <= G lanes draw one node's neighbours
      const int G = (k >= 0 && k <= 16) ? 16 : (k >= 0 && k <= 32) ? 32 : 64;
      const int grid = static_cast<int>(cdiv(fcap > 0 ? fcap : 1, kHopBlock / G));   // <= n_a
#define SAGE_HOP_SAMPLE(GG, BB)                                                                       \
  hop_sample_kernel<GG, BB><<<grid, kHopBlock, 0, stream>>>(                                            \
      colptr, row, num_nodes, node, hopinfo, ticket_a, totals_a, k, static_cast<int>(fcap), h, seed_lo, \
      seed_hi, relabel_table, edge_cap, src, dst, eid, seg_ptr, state, kPollLimit)
      if (k > kWave) SAGE_HOP_SAMPLE(64, true);
      else if (G == 64) SAGE_HOP_SAMPLE(64, false);
      else if (G == 32) SAGE_HOP_SAMPLE(32, false);
      else SAGE_HOP_SAMPLE(16, false);
#undef SAGE_HOP_SAMPLE
      SAGE_CHECK_LAUNCH("hop_sample_kernel");
    }
    hop_flag_kernel<<<static_cast<int>(cdiv(ecap > 0 ? ecap : 1, kHopBlock)), kHopBlock, 0, stream>>>(   // <= n_b
        hopinfo, ticket_b, totals_b, node_cap, relabel_table, src, node, h, state, kPollLimit);
    SAGE_CHECK_LAUNCH("hop_flag_kernel");
    {
      int64_t work = ecap;
      if (last && node_cap + 1 > work) work = node_cap + 1;
      if (work > 0) {
        hop_relabel_kernel<<<static_cast<int>(cdiv(work, kBlock)), kBlock, 0, stream>>>(
            hopinfo, relabel_table, src, last ? 1 : 0, node_cap, state, seg_ptr);
        SAGE_CHECK_LAUNCH("hop_relabel_kernel");
      }
    }
    // frontier of the next hop = nodes first seen in this one
    fcap = ecap < node_cap ? ecap : node_cap;
  }
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

namespace {
// One wave of hop_flag_kernel's exchange on a prepared totals array (sage_debug_exchange_wait).
__global__ void exchange_wait_kernel(unsigned long long* words, int L, int64_t* state, int poll_limit,
                                     unsigned long long* out) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t base = sum_lower_tickets(words, L, lane, state, poll_limit);
  if (lane == 0) out[0] = base;
}
}  // namespace

extern "C" int sage_debug_exchange_wait(unsigned long long* words, int num_words, sage_i64* state, int poll_limit,
                                        unsigned long long* sum_out, void* stream) {
  SAGE_REQUIRE(words && state && sum_out && num_words >= 0 && num_words <= 4096 && poll_limit >= 0 &&
                   poll_limit <= kPollLimit,
               SAGE_EINVAL, "sage_debug_exchange_wait: bad argument");
  exchange_wait_kernel<<<1, kWave, 0, static_cast<hipStream_t>(stream)>>>(words, num_words, state, poll_limit, sum_out);
  SAGE_CHECK_LAUNCH("exchange_wait_kernel");
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
  // the array keeps the last call's contents: every call starts by clearing the tickets and
  // published totals it will use (seed_kernel), and the hop-info words are written before they
  // are read, so nothing else has to be cleared between calls
  return sage_sample_fanout_ctrl(colptr, row, num_nodes, seeds, num_seeds, fanout_host, hops,
                                 rng_seed, rng_counter, relabel_table, stamp, nullptr, node_cap,
                                 edge_cap, node, src, dst, eid, seg_ptr, state, ws,
                                 static_cast<size_t>(SAGE_STATIC_SCRATCH_BYTES), stream);
}
