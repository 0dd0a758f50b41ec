/* neighbor_sample restated in plain C — TEST INFRASTRUCTURE / CPU BASELINE ONLY.
 *
 * Same algorithm and same Philox counters as oracle/sampler.py (which documents the
 * upstream op it restates: torch_sparse::neighbor_sample<replace=false, directed=true>,
 * called at reference samplers/uniform_loader.py:64-72): sequential over frontier nodes,
 * Robert Floyd selection, hash map for first-seen relabelling.  Written separately from the
 * HIP sampler (graphsage_amd/csrc/sampler.hip) on purpose: sequential, one hash map, no
 * scans — the structure of the upstream CPU op.  tests/test_oracle_c.py checks it bit for
 * bit against the Python restatement; bench.py times it as the CPU baseline's sampler.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

/* open-addressing map global id -> local id */
typedef struct { int64_t* keys; int64_t* vals; uint64_t mask; } map_t;
static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }
static int64_t* map_slot(map_t* m, int64_t key, int* found) {
  uint64_t i = mix((uint64_t)key) & m->mask;
  for (;;) {
    if (m->keys[i] == -1) { *found = 0; m->keys[i] = key; return &m->vals[i]; }
    if (m->keys[i] == key) { *found = 1; return &m->vals[i]; }
    i = (i + 1) & m->mask;
  }
}

/* Returns 0 on success, -1 if a capacity is exceeded, -2 on allocation failure.
 * counts[0] = n_nodes, counts[1] = n_edges, counts[2+h] = nodes after hop h, counts[18+h] = edges after hop h */
int oracle_neighbor_sample(const int64_t* colptr, const int64_t* row, const int64_t* seeds, int num_seeds,
                           const int* fanout, int hops, uint64_t rng_seed, uint64_t rng_counter,
                           int64_t node_cap, int64_t edge_cap, int64_t* node, int64_t* src, int64_t* dst,
                           int64_t* eid, int64_t* counts) {
  uint64_t cap = 16;
  while (cap < (uint64_t)node_cap * 2 + 16) cap <<= 1;
  map_t m;
  m.mask = cap - 1;
  m.keys = (int64_t*)malloc(cap * sizeof(int64_t));
  m.vals = (int64_t*)malloc(cap * sizeof(int64_t));
  int kmax = 1;
  for (int h = 0; h < hops; ++h) if (fanout[h] > kmax) kmax = fanout[h];
  uint32_t* chosen = (uint32_t*)malloc((size_t)kmax * sizeof(uint32_t));
  if (!m.keys || !m.vals || !chosen) { free(m.keys); free(m.vals); free(chosen); return -2; }
  memset(m.keys, 0xFF, cap * sizeof(int64_t));
  int rc = 0;
  int64_t n_nodes = 0, n_edges = 0;
  if (num_seeds > node_cap) { rc = -1; goto done; }
  for (int i = 0; i < num_seeds; ++i) {
    int found;
    int64_t* slot = map_slot(&m, seeds[i], &found);
    if (!found) *slot = i;
    node[n_nodes++] = seeds[i];
  }
  counts[2] = n_nodes; counts[18] = 0;
  int64_t begin = 0, end = n_nodes;
  for (int h = 0; h < hops; ++h) {
    const int k = fanout[h];
    const uint32_t k0 = (uint32_t)rng_seed, k1 = (uint32_t)(rng_seed >> 32) + (uint32_t)h;
    for (int64_t i = begin; i < end; ++i) {
      const int64_t w = node[i];
      const int64_t cs = colptr[w], deg = colptr[w + 1] - cs;
      if (deg == 0) continue;
      const int take_all = (k < 0 || k >= deg);
      const int64_t n_out = take_all ? deg : k;
      if (n_edges + n_out > edge_cap) { rc = -1; goto done; }
      for (int64_t t = 0; t < n_out; ++t) {
        int64_t c;
        if (take_all) {
          c = t;
        } else {
          const int64_t j = deg - k + t;
          uint32_t ctr[4] = {(uint32_t)(t >> 2), (uint32_t)i, (uint32_t)rng_counter, (uint32_t)(rng_counter >> 32)};
          philox4x32_10(ctr, k0, k1);
          c = (int64_t)(((uint64_t)ctr[t & 3] * (uint64_t)(j + 1)) >> 32);
          for (int64_t s = 0; s < t; ++s) if (chosen[s] == (uint32_t)c) { c = j; break; }
          chosen[t] = (uint32_t)c;
        }
        const int64_t off = cs + c;
        const int64_t v = row[off];
        int found;
        int64_t* slot = map_slot(&m, v, &found);
        if (!found) {
          if (n_nodes >= node_cap) { rc = -1; goto done; }
          *slot = n_nodes;
          node[n_nodes++] = v;
        }
        src[n_edges] = *slot; dst[n_edges] = i; eid[n_edges] = off;
        ++n_edges;
      }
    }
    begin = end; end = n_nodes;
    counts[2 + h + 1] = n_nodes; counts[18 + h + 1] = n_edges;
  }
done:
  counts[0] = n_nodes; counts[1] = n_edges;
  free(m.keys); free(m.vals); free(chosen);
  return rc;
}
