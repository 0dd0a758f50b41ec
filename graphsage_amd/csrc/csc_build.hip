// (a-1) One-off CSC build on the device.
//
// Replaces torch_geometric.loader.utils.to_csc(data, device='cpu'), called at reference
// samplers/uniform_loader.py:32:
//     perm = (col * N + row).argsort();  row = row[perm];  colptr = ind2ptr(col[perm], N)
// The reference does this argsort on the CPU once per loader (three loaders per trainer:
// trainers/node_level_graphsage_trainers.py:18-25).  Here it is one stable LSD radix sort
// of the 64-bit keys col*N+row (only the bits N*N needs) plus a histogram + scan.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace sage {
namespace {

constexpr int kBlock = 256;

// used instead of hipMemsetAsync so that every node a caller captures in a HIP graph is a kernel
__global__ void zero_i64_kernel(int64_t* p, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

__global__ void make_keys_kernel(const int64_t* __restrict__ edge_row,
                                 const int64_t* __restrict__ edge_col, int64_t num_edges,
                                 int64_t num_nodes, int64_t* __restrict__ keys,
                                 int64_t* __restrict__ iota,
                                 unsigned long long* __restrict__ counts,
                                 unsigned long long* __restrict__ bad) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= num_edges) return;
  int64_t r = edge_row[e], c = edge_col[e];
  if (r < 0 || r >= num_nodes || c < 0 || c >= num_nodes) {
    atomicAdd(bad, 1ull);
    r = 0;
    c = 0;
  }
  keys[e] = c * num_nodes + r;
  iota[e] = e;
  atomicAdd(&counts[c], 1ull);
}

__global__ void decode_rows_kernel(const int64_t* __restrict__ keys_sorted, int64_t num_edges,
                                   int64_t num_nodes, int64_t* __restrict__ row_out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e < num_edges) row_out[e] = keys_sorted[e] % num_nodes;
}

int key_bits(int64_t num_nodes) {
  // keys are < N*N
  const unsigned __int128 lim = static_cast<unsigned __int128>(num_nodes) * num_nodes;
  int b = 1;
  while (b < 63 && (static_cast<unsigned __int128>(1) << b) < lim) ++b;
  return b;
}

size_t sort_bytes_for(size_t n, int bits) {
  size_t bytes = 0;
  int64_t* p = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, n, 0, bits, nullptr);
  return bytes;
}
size_t scan_bytes_for(size_t n) {
  size_t bytes = 0;
  int64_t* p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, bytes, p, p, int64_t(0), n, rocprim::plus<int64_t>(), nullptr);
  return bytes;
}

}  // namespace
}  // namespace sage

using namespace sage;

extern "C" size_t sage_csc_build_workspace_bytes(sage_i64 num_edges, sage_i64 num_nodes) {
  if (num_edges < 0 || num_nodes <= 0) return 0;
  const size_t e = static_cast<size_t>(num_edges) + 1;
  size_t total = 0;
  total += 3 * align_up(sizeof(int64_t) * e, 256);  // keys, keys_sorted, iota
  total += align_up(sizeof(int64_t) * (static_cast<size_t>(num_nodes) + 2), 256);  // counts (+bad)
  total += align_up(sort_bytes_for(e, key_bits(num_nodes)), 256);
  total += align_up(scan_bytes_for(static_cast<size_t>(num_nodes) + 2), 256);
  return total + 256;
}

extern "C" int sage_csc_build(const sage_i64* edge_row, const sage_i64* edge_col,
                              sage_i64 num_edges, sage_i64 num_nodes, sage_i64* colptr,
                              sage_i64* row_out, sage_i64* perm, void* workspace,
                              size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(num_edges >= 0 && num_nodes > 0, SAGE_EINVAL, "sage_csc_build: bad sizes");
  SAGE_REQUIRE(num_nodes < (1ll << 31), SAGE_ELIMIT,
               "sage_csc_build: num_nodes=%lld: col*N+row keys would not fit 63 bits",
               (long long)num_nodes);
  SAGE_REQUIRE(colptr && workspace, SAGE_EINVAL, "sage_csc_build: NULL pointer argument");
  SAGE_REQUIRE(workspace_bytes >= sage_csc_build_workspace_bytes(num_edges, num_nodes),
               SAGE_EWORKSPACE, "sage_csc_build: workspace too small");
  Carver ws(workspace);
  const size_t e = static_cast<size_t>(num_edges) + 1;
  int64_t* keys = ws.take<int64_t>(e);
  int64_t* keys_sorted = ws.take<int64_t>(e);
  int64_t* iota = ws.take<int64_t>(e);
  int64_t* counts = ws.take<int64_t>(num_nodes + 2);
  const int bits = key_bits(num_nodes);
  size_t sort_bytes = sort_bytes_for(e, bits);
  void* sort_tmp = ws.take<char>(sort_bytes);
  size_t scan_bytes = scan_bytes_for(static_cast<size_t>(num_nodes) + 2);
  void* scan_tmp = ws.take<char>(scan_bytes);

  // counts[0..N-1] histogram, counts[N] = 0 (scan total slot), counts[N+1] = bad-edge counter
  {
    const int64_t nz = (num_nodes + 2);
    zero_i64_kernel<<<static_cast<unsigned>(cdiv(nz, kBlock)), kBlock, 0, stream>>>(counts, nz);
    SAGE_CHECK_LAUNCH("zero_i64_kernel");
  }
  if (num_edges > 0) {
    SAGE_REQUIRE(edge_row && edge_col && row_out && perm, SAGE_EINVAL,
                 "sage_csc_build: NULL pointer argument");
    const unsigned grid = static_cast<unsigned>(cdiv(num_edges, kBlock));
    make_keys_kernel<<<grid, kBlock, 0, stream>>>(
        edge_row, edge_col, num_edges, num_nodes, keys, iota,
        reinterpret_cast<unsigned long long*>(counts),
        reinterpret_cast<unsigned long long*>(counts + num_nodes + 1));
    SAGE_CHECK_LAUNCH("make_keys_kernel");
    SAGE_CHECK_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys, keys_sorted, iota, perm,
                                             static_cast<size_t>(num_edges), 0, bits, stream),
                   "radix_sort_pairs");
    decode_rows_kernel<<<grid, kBlock, 0, stream>>>(keys_sorted, num_edges, num_nodes, row_out);
    SAGE_CHECK_LAUNCH("decode_rows_kernel");
  }
  SAGE_CHECK_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, counts, colptr, int64_t(0),
                                         static_cast<size_t>(num_nodes + 1),
                                         rocprim::plus<int64_t>(), stream),
                 "scan(counts)");
  return SAGE_OK;
}
