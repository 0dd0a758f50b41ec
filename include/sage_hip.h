/*
 * sage_hip.h — C ABI of libsage_hip.so, the MI355X (gfx950) implementation of
 * GraphSAGE's sampled-neighbourhood mini-batch hot path.
 *
 * Every entry point replaces one third-party native call made by the reference
 * (wilcoln/graphsage); the call site it replaces is cited as  <file>:<line>  relative to
 * the reference tree.  The reference binds those natives through PyTorch custom ops
 * (torch.ops.torch_sparse.*, torch_scatter.scatter, torch_geometric Linear); a maintainer
 * binds this library through ctypes exactly as graphsage_amd/_lib.py does (see
 * INTEGRATION.md).
 *
 * Conventions
 *   - All pointers are DEVICE pointers (hipMalloc'ed / torch CUDA tensors) unless the
 *     parameter name ends in `_host`.
 *   - Index tensors are int64 (`sage_i64`), as in the reference's torch.LongTensor
 *     (samplers/uniform_loader.py:60-61).  Features/weights/gradients are IEEE fp32.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued on it; no entry point synchronises, allocates or frees device memory or
 *     retains a pointer after it returns, so every call may be captured in a HIP graph.
 *   - Return value 0 = success; otherwise a negative SAGE_E* code and a message in
 *     sage_last_error() (thread-local).  The Python binding raises RuntimeError, the
 *     convention of the natives it replaces (TORCH_CHECK).
 *   - `*_dev` count arguments are optional device pointers to a row/edge count that is
 *     only known on the device (a sampled batch's size).  When non-NULL the kernels read
 *     the count from device memory and the host-side count is the launch upper bound.
 */
#ifndef SAGE_HIP_H
#define SAGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t sage_i64;

#define SAGE_OK 0
#define SAGE_EINVAL (-1)    /* bad argument (shape, alignment, NULL pointer)          */
#define SAGE_EWORKSPACE (-2) /* workspace too small                                     */
#define SAGE_ELAUNCH (-3)   /* HIP runtime reported an error at launch                 */
#define SAGE_ELIMIT (-4)    /* size beyond a documented limit of this library          */

/* reduce ops of sage_gather_reduce_* — the `reduce` strings of
 * layers/sage.py:195-210 ('mean' | 'max' | 'sum'; 'gcn'/'mean_pool' map to mean,
 * 'max_pool' to max). */
#define SAGE_REDUCE_SUM 0
#define SAGE_REDUCE_MEAN 1
#define SAGE_REDUCE_MAX 2
#define SAGE_REDUCE_GCN 3 /* mean over the neighbours and ONE self-loop, for targets that have an
                             in-edge (aggregator 'gcn', layers/sage.py:62-65, :114-117) */

/* epilogue flags of sage_linear_concat_fwd */
#define SAGE_EPI_NONE 0
#define SAGE_EPI_BIAS 1 /* + bias[n]            (pool Linear, layers/sage.py:76)       */
#define SAGE_EPI_RELU 2 /* relu                 (layers/sage.py:202,206)               */
#define SAGE_EPI_ACCUM 4 /* out += product (alone): `out += lin_r(x_r)` of layers/sage.py:129; the
                          * caller provides the addend (zeros for a plain product)        */
#define SAGE_EPI_ZEROED 8 /* with SAGE_EPI_ACCUM: the caller promises that the addend is all zeros
                           * (a buffer its consumer hands back cleared).  Shapes whose output
                           * elements have one owner are then stored instead of load-add-stored;
                           * shapes that share an element between workgroups add into it as before,
                           * without a fill pass.  No counterpart in the reference.          */

/* words of the sampler's device state block (sage_i64 state[SAGE_STATE_WORDS]) */
#define SAGE_ST_NODES 0      /* number of sampled nodes so far  (len(samples))           */
#define SAGE_ST_EDGES 1      /* number of sampled edges so far                           */
#define SAGE_ST_FRONT_BEGIN 2 /* local id range of the nodes the next hop expands       */
#define SAGE_ST_FRONT_END 3
#define SAGE_ST_ERROR 4      /* bit0: node capacity exceeded, bit1: edge capacity,       */
                             /* bit2: a seed is not a node, bit3: a CSC row id is not a  */
                             /* node, bit4: the in-launch prefix exchange gave up        */
                             /* waiting for another workgroup (result wrong, in bounds)  */
#define SAGE_ST_HOP_NODES 8  /* [8+h]  = #nodes after hop h (h=0: the seeds)             */
#define SAGE_ST_HOP_EDGES 24 /* [24+h] = #edges after hop h (h=0: 0)                     */
#define SAGE_STATE_WORDS 40
#define SAGE_MAX_HOPS 15

const char* sage_last_error(void);
int sage_abi_version(void);
/* Measurement hook (no counterpart in the reference; bench.py's roofline_mfma): an MFMA
 * contraction kernel that this thread launches through sage_linear_concat_fwd /
 * sage_pool_linear_fwd / sage_linear_dinput(2) / sage_linear_dweight(2) carries this HIP event
 * pair (hipExtLaunchKernelGGL start / stop: their elapsed time is that dispatch's own duration)
 * — the (skip+1)-th such launch from now.  One-shot per pair, at most 8 pending per thread
 * (SAGE_ELIMIT); (NULL, NULL) cancels all pending pairs.  The skinny VALU kernels and the
 * zero-fill launches do not count. */
int sage_probe_next_contraction(void* start_event, void* stop_event, int skip);
/* Cross-stream ordering for callers that replay recorded call sequences over several streams
 * (graphsage_amd/engine.py: the forked variants of the three-forward steps): hipEventRecord /
 * hipStreamWaitEvent on caller-owned handles, so that the fork and the join are entries of the
 * same call tape as the kernels between them (2.6 us and 0.3 us of host time; the host
 * framework's stream-to-stream wait creates, records and destroys an event per call: 15 us).
 * No counterpart in the reference.  SAGE_EINVAL for a NULL event, SAGE_ELAUNCH if HIP refuses. */
int sage_event_record(void* event, void* stream);
int sage_stream_wait_event(void* stream, void* event);

/* ------------------------------------------------------------------------------------
 * (a-1) CSC build.  Replaces torch_geometric.loader.utils.to_csc(data, device='cpu')
 * called at samplers/uniform_loader.py:32:
 *     perm = argsort(col * N + row);  row_out = row[perm];  colptr = ind2ptr(col[perm], N)
 * Stable sort: equal (col,row) pairs keep their input order in `perm`.
 * Limits: N*N must fit in 63 bits.
 * ---------------------------------------------------------------------------------- */
size_t sage_csc_build_workspace_bytes(sage_i64 num_edges, sage_i64 num_nodes);
int sage_csc_build(const sage_i64* edge_row, const sage_i64* edge_col, sage_i64 num_edges,
                   sage_i64 num_nodes, sage_i64* colptr /*[N+1]*/, sage_i64* row_out /*[E]*/,
                   sage_i64* perm /*[E]*/, void* workspace, size_t workspace_bytes,
                   void* stream);

/* ------------------------------------------------------------------------------------
 * (a-2) Multi-hop uniform fan-out neighbour sampling without replacement.  Replaces
 *     torch.ops.torch_sparse.neighbor_sample(colptr, row, index, num_neighbors,
 *                                            replace=False, directed=True)
 * called at samplers/uniform_loader.py:64-72.  Outputs follow that op:
 *     node[0:n_nodes]  global ids in first-seen order, node[0:B] == seeds
 *     src[e] (the op's `row`) local id of the sampled neighbour,
 *     dst[e] (the op's `col`) local id of the target, eid[e] offset into `row`
 * and in addition seg_ptr[0:node_cap+1]: edges whose target is local node t are
 * [seg_ptr[t], seg_ptr[t+1]) — the op emits edges grouped by target in frontier order,
 * so this CSR comes for free and is what the aggregation kernels consume.
 *
 * Randomness: Philox4x32-10, key = (rng_seed lo, rng_seed hi + hop),
 * counter = (draw>>2, local id of target, rng_counter lo, rng_counter hi); draw t of a
 * target with deg > k picks r = mulhi32(word[t&3], j+1), j = deg-k+t (Robert Floyd),
 * r replaced by j if r was already chosen.  Bit-exact against oracle/sampler.py for the
 * same (rng_seed, rng_counter); the upstream op's CPU generator stream cannot be
 * reproduced on any device, so equivalence to it is distributional.
 *
 * relabel_table: sage_u64[num_nodes] scratch owned by the caller, initialised ONCE to
 * all-ones bytes (0xFF); `stamp` must be > 0 and larger than in any earlier call that
 * used the same table (stale entries are recognised by their stamp, no clearing pass).
 * Concurrent calls must not share a table.
 * state: sage_i64[SAGE_STATE_WORDS], written by the call (see SAGE_ST_*).
 * Capacity overflow truncates the output and sets SAGE_ST_ERROR; with all fan-outs >= 0,
 * node_cap = B*(1 + k1 + k1*k2 + ...) and edge_cap = B*(k1 + k1*k2 + ...) never overflow.
 * Limits: 0 < hops <= SAGE_MAX_HOPS, fan-out <= 2048 (or < 0 = all neighbours),
 * edge_cap < 2^31.
 * ---------------------------------------------------------------------------------- */
size_t sage_sample_workspace_bytes(sage_i64 node_cap, sage_i64 edge_cap);
int sage_sample_fanout(const sage_i64* colptr, const sage_i64* row, sage_i64 num_nodes,
                       const sage_i64* seeds, int num_seeds, const int* fanout_host, int hops,
                       uint64_t rng_seed, uint64_t rng_counter, uint64_t* relabel_table,
                       uint32_t stamp, sage_i64 node_cap, sage_i64 edge_cap,
                       sage_i64* node /*[node_cap]*/, sage_i64* src /*[edge_cap]*/,
                       sage_i64* dst /*[edge_cap]*/, sage_i64* eid /*[edge_cap]*/,
                       sage_i64* seg_ptr /*[node_cap+1]*/, sage_i64* state, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Same call with a DEVICE control block, for callers that replay the sampler from a HIP graph
 * (host scalars are frozen at capture time): ctrl_dev[SAGE_CTRL_COUNTER] is added to
 * rng_counter, ctrl_dev[SAGE_CTRL_STAMP] to stamp, and the seeds are read from
 * seeds + ctrl_dev[SAGE_CTRL_SEED_OFFSET].  ctrl_dev = NULL gives sage_sample_fanout.
 * sage_ctrl_advance (one tiny kernel, capturable) adds (d_counter, d_stamp, d_offset) to the
 * block; the seed offset restarts at 0 when another batch of d_offset seeds would pass
 * offset_wrap (> 0).  No counterpart in the reference: its sampler runs on the host. */
#define SAGE_CTRL_COUNTER 0
#define SAGE_CTRL_STAMP 1
#define SAGE_CTRL_SEED_OFFSET 2
#define SAGE_CTRL_WORDS 4
int sage_sample_fanout_ctrl(const sage_i64* colptr, const sage_i64* row, sage_i64 num_nodes,
                            const sage_i64* seeds, int num_seeds, const int* fanout_host, int hops,
                            uint64_t rng_seed, uint64_t rng_counter, uint64_t* relabel_table,
                            uint32_t stamp, const sage_i64* ctrl_dev, sage_i64 node_cap,
                            sage_i64 edge_cap, sage_i64* node, sage_i64* src, sage_i64* dst,
                            sage_i64* eid, sage_i64* seg_ptr, sage_i64* state, void* workspace,
                            size_t workspace_bytes, void* stream);
/* Diagnostic (tests only; no counterpart in the reference): ONE wave runs the sampler's bounded
 * wait — the sum of the totals published in words[0 .. num_words) as {bit 63 = ready, low 32 bits
 * = total} — with `poll_limit` polls per word instead of the sampler's own limit.  Words whose
 * ready bit never appears are skipped after the limit and bit4 is set in state[SAGE_ST_ERROR];
 * sum_out[0] receives the sum of the words that were ready. */
int sage_debug_exchange_wait(unsigned long long* words, int num_words, sage_i64* state, int poll_limit,
                             unsigned long long* sum_out, void* stream);
int sage_ctrl_advance(sage_i64* ctrl_dev, sage_i64 d_counter, sage_i64 d_stamp, sage_i64 d_offset,
                      sage_i64 offset_wrap, void* stream);

/* sage_sample_fanout for callers that cannot allocate device memory (code under HIP-graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments, outputs and results as sage_sample_fanout, without the workspace:
 * the hop counters, the tickets and the per-workgroup totals of its prefix exchanges live in a file-scope device array of
 * SAGE_STATIC_SCRATCH_BYTES inside the library (one copy per GPU, held until the process ends;
 * its address is resolved once per device on the first call made with that device current).
 * The call allocates and frees nothing.  node_cap / edge_cap whose
 * sage_sample_workspace_bytes() exceeds SAGE_STATIC_SCRATCH_BYTES are rejected with SAGE_ELIMIT
 * (batch 512 at fan-out 25/10 — node_cap 141,312, edge_cap 140,800 — needs 1.1 MB).  All calls on one GPU share that GPU's array: two calls that overlap on
 * one GPU (different streams or threads) race; serialise them or use sage_sample_fanout. */
#define SAGE_STATIC_SCRATCH_BYTES (8u << 20)
int sage_sample_fanout_static(const sage_i64* colptr, const sage_i64* row, sage_i64 num_nodes,
                              const sage_i64* seeds, int num_seeds, const int* fanout_host,
                              int hops, uint64_t rng_seed, uint64_t rng_counter,
                              uint64_t* relabel_table, uint32_t stamp, sage_i64 node_cap,
                              sage_i64 edge_cap, sage_i64* node /*[node_cap]*/,
                              sage_i64* src /*[edge_cap]*/, sage_i64* dst /*[edge_cap]*/,
                              sage_i64* eid /*[edge_cap]*/, sage_i64* seg_ptr /*[node_cap+1]*/,
                              sage_i64* state, void* stream);

/* (f-2) one uniform random node id per slot: the negative samples of
 * samplers/utils.py:30-31 (torch.randint(0, num_nodes, (n,))).  Philox key = rng_seed,
 * counter = (i>>2, 0x4E450000, rng_counter lo, hi), range reduction mulhi32(u, num_nodes). */
int sage_random_nodes(sage_i64 num_nodes, sage_i64 n, uint64_t rng_seed, uint64_t rng_counter,
                      sage_i64* out, void* stream);

/* sum[0:SAGE_STATE_WORDS] += state[...]: running totals of the batch sizes the sampler
 * reported (the device-side counterpart of `total_examples += batch.batch_size`,
 * experiments/fig2a/runners/graphsage_runners.py:48, without reading the sizes back). */
int sage_state_accumulate(const sage_i64* state, sage_i64* sum, void* stream);

/* (f-2) one uniform random in-neighbour per start node; replaces
 * torch_cluster.random_walk(row, col, start, walk_length=1)[:,1] at samplers/utils.py:22.
 * A start with no neighbour returns itself (torch_cluster's behaviour); a start id outside
 * [0, num_nodes) yields 0 and touches no memory.  Philox key =
 * (rng_seed), counter = (i>>2, 0x52570000, rng_counter lo, hi). */
int sage_random_neighbor(const sage_i64* colptr, const sage_i64* row, sage_i64 num_nodes,
                         const sage_i64* start, sage_i64 num_start,
                         const sage_i64* num_start_dev /* optional device-side count, limits num_start */,
                         uint64_t rng_seed, uint64_t rng_counter, sage_i64* out, void* stream);

/* ------------------------------------------------------------------------------------
 * (a-3) Row gather.  Replaces torch_geometric.loader.utils.filter_data's
 * index_select of node-level attributes (samplers/uniform_loader.py:238-239):
 *     out[i, 0:width] = table[index[i], 0:width]     (rows of `width` 32-bit words)
 * `ld_table`/`ld_out` are row strides in 32-bit words.  Serves fp32 features
 * (width = F) and int64 labels / ids (width = 2).
 * ---------------------------------------------------------------------------------- */
int sage_gather_rows(const void* table, sage_i64 ld_table, const sage_i64* index,
                     sage_i64 num_rows, const sage_i64* num_rows_dev, int width, void* out,
                     sage_i64 ld_out, void* stream);

/* ------------------------------------------------------------------------------------
 * (a-5) Fused gather + segment reduce.  Replaces SAGE.message (x_j = x[edge_index[0]],
 * layers/sage.py:136-137) followed by torch_scatter.scatter(x_j, edge_index[1], dim=-2,
 * dim_size, reduce) at layers/sage.py:209-210, without materialising x_j:
 *     out[t, :] = reduce_{e in [seg_ptr[t], seg_ptr[t+1])} X[row_of(src[e]), :]
 * row_of(s) = node_map ? node_map[s] : s   (node_map = the batch's `node` vector, so the
 * gather reads the global feature table directly instead of a materialised batch.x).
 * Empty segments give 0 for every reduce; mean divides by max(count,1).  SAGE_REDUCE_GCN adds the
 * target's own row X[row_of(t), :] after its neighbours and divides by count + 1 (count > 0).
 * max also writes argmax[t, f] = the edge index e that supplied the maximum (first such
 * edge in segment order on ties), or -1 for an empty segment; argmax may be NULL.
 * ---------------------------------------------------------------------------------- */
int sage_gather_reduce_fwd(const float* x, sage_i64 ld_x, const sage_i64* node_map,
                           const sage_i64* seg_ptr, const sage_i64* src, sage_i64 num_targets,
                           const sage_i64* num_targets_dev, int num_features, int reduce,
                           float* out, sage_i64 ld_out, sage_i64* argmax, void* stream);

/* The same launch carrying two HIP events (hipEvent_t, created by the caller with timing
 * enabled): the kernel is dispatched with hipExtLaunchKernelGGL, so that
 * hipEventElapsedTime(start_event, stop_event) is the duration of THIS kernel (the dispatch's
 * own start / end timestamps) rather than the distance between two markers recorded around it.
 * Measurement only (bench.py's roofline probe); no counterpart in the reference.  Fails with
 * SAGE_EINVAL when nothing would be launched (num_targets == 0). */
int sage_gather_reduce_fwd_timed(const float* x, sage_i64 ld_x, const sage_i64* node_map,
                                 const sage_i64* seg_ptr, const sage_i64* src,
                                 sage_i64 num_targets, const sage_i64* num_targets_dev,
                                 int num_features, int reduce, float* out, sage_i64 ld_out,
                                 sage_i64* argmax, void* start_event, void* stop_event,
                                 void* stream);

/* Backward of the above w.r.t. x (autograd of scatter + index_select):
 *   sum : dx[src[e], :] += dout[t, :]            mean: += dout[t, :] / max(count_t, 1)
 *   max : dx[src[argmax[t,f]], f] += dout[t, f]  (argmax >= 0)
 *   gcn : += dout[t, :] / (count_t + 1) to every dx[src[e], :] AND to dx[t, :]  (count_t > 0)
 * dx must be pre-initialised (zero, or a gradient to accumulate into); adds are fp32
 * atomics, so the summation order — not the set of addends — may vary between runs. */
int sage_gather_reduce_bwd(const float* dout, sage_i64 ld_dout, const sage_i64* seg_ptr,
                           const sage_i64* src, const sage_i64* argmax, sage_i64 num_targets,
                           const sage_i64* num_targets_dev, int num_features, int reduce,
                           float* dx, sage_i64 ld_dx, void* stream);

/* COO -> target-CSR for an arbitrary edge_index (any order): stable counting sort of the
 * edges by target.  Needed because SAGE.forward accepts any [2,E] tensor
 * (layers/sage.py:99-100) while the kernels above want segments.  perm[i] = input edge
 * placed at sorted position i, src_sorted[i] = edge_src[perm[i]]. */
size_t sage_coo_to_csr_workspace_bytes(sage_i64 num_edges, sage_i64 num_targets);
int sage_coo_to_csr(const sage_i64* edge_src, const sage_i64* edge_dst, sage_i64 num_edges,
                    sage_i64 num_targets, sage_i64* seg_ptr /*[T+1]*/,
                    sage_i64* src_sorted /*[E]*/, sage_i64* perm /*[E]*/, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * (a-4) self||neighbour Linear.  Replaces lin_l(out) + lin_r(x_r) of
 * layers/sage.py:124-129 (two bias-free torch_geometric Linear, :78-81) and the pool
 * Linear (+bias, +ReLU) of layers/sage.py:76,202,206:
 *     out[m, n] = sum_k a1[r1(m), k] * w1[n, k]  +  sum_k a2[r2(m), k] * w2[n, k]
 *                 (+ bias[n]) (relu)
 * r1/r2: optional row maps (NULL = identity) so the root term can be read straight from
 * the global feature table through the batch's `node` vector.  a2/w2 may be NULL
 * (k2 = 0: 'gcn' aggregator / root_weight=False / pool).
 * Alignment: every pointer and row stride must be a multiple of 8 bytes.
 * Arithmetic — two kernels, chosen per call by the shape:
 *   * LARGE contractions run in "bf16x3" mode BY DEFAULT: n > 64, every operand 16-byte
 *     aligned (pointers and row strides), and at least 6 K-steps of 32 per workgroup slot of the
 *     chip (about 1 GFLOP on MI355X); epilogues NONE, ACCUM, and BIAS / RELU on whole tiles.
 *     Every fp32 operand is split round-to-nearest into three bf16 terms (x = hi + mid + lo
 *     exactly) while it is staged, six of the nine cross products go through the bf16 MFMA
 *     (v_mfma_f32_16x16x32_bf16) into fp32 accumulators; the three dropped products are
 *     <= 2^-24 of the product.  The error against an exact result is that of an fp32 fma chain
 *     on the same operands or smaller; the BITS are not those of an fp32 fma chain.  Domain:
 *     an Inf operand gives NaN (Inf - Inf in the residuals) in every output it reaches, a NaN
 *     operand NaN — the same outputs the fp32 kernel makes non-finite; operands below about
 *     2^-110 lose their mid / lo terms to bf16 underflow (absolute error below 2^-126 per
 *     product: invisible next to any normal-sized term of the same sum).
 *     SAGE_GEMM_X3=0 in the environment (read at every call) switches the mode off.
 *   * everything else — and every call under SAGE_GEMM_X3=0 — runs exact fp32 arithmetic on
 *     the matrix cores (v_mfma_f32_32x32x2_f32): each output is an fp32 fma chain over k.
 * Few-tile shapes take a stream-K / split-K schedule whose partial tiles are added with fp32
 * atomics (summation order varies between runs; out must then hold zeros or the addend:
 * SAGE_EPI_ACCUM).  The same routing covers the large weight-gradient contractions of
 * sage_linear_dweight (with dbias) / sage_linear_dweight2 below (k > 64, m >= 2048 rows,
 * 16-byte aligned operands).  No counterpart in the reference.
 * ---------------------------------------------------------------------------------- */
int sage_linear_concat_fwd(const float* a1, sage_i64 ld_a1, const sage_i64* rows1, int k1,
                           const float* w1, sage_i64 ld_w1, const float* a2, sage_i64 ld_a2,
                           const sage_i64* rows2, int k2, const float* w2, sage_i64 ld_w2,
                           const float* bias, int epilogue, sage_i64 m,
                           const sage_i64* m_dev, int n, float* out, sage_i64 ld_out,
                           void* stream);

/* Input gradient:  da[m, k] = sum_n dout[m, n] * w[n, k]   (da = dout @ w), optionally
 * accumulated into da (beta = 1) instead of overwriting it (beta = 0). */
int sage_linear_dinput(const float* dout, sage_i64 ld_dout, const float* w, sage_i64 ld_w,
                       sage_i64 m, const sage_i64* m_dev, int n, int k, float* da,
                       sage_i64 ld_da, int accumulate, void* stream);

/* Both input gradients of a layer in one launch: da1 = dout . w1, da2 = dout . w2 (what autograd
 * derives for out = lin_l(agg) + lin_r(x_r), layers/sage.py:124-129: d agg and d x_r).  Same
 * shapes as sage_linear_dinput; both outputs are overwritten (rows at and past *m_dev stay
 * untouched).  n <= 64: one VALU grid over both products.  Otherwise, when both products together
 * have at most 128 output tiles of 64 x 128 (the 512 seed rows of a wide last layer), ONE MFMA
 * grid works on both with the contraction split over blockIdx.z: a fill launch clears the outputs
 * and the slices add into them with fp32 atomics (summation order varies between runs); larger
 * shapes run as two whole-K launches. */
int sage_linear_dinput2(const float* dout, sage_i64 ld_dout, const float* w1, sage_i64 ld_w1,
                        float* da1, sage_i64 ld_da1, const float* w2, sage_i64 ld_w2, float* da2,
                        sage_i64 ld_da2, sage_i64 m, const sage_i64* m_dev, int n, int k,
                        void* stream);

/* Weight gradient: dw[n, k] = sum_m dout[m, n] * a[r(m), k]  (dw = dout^T @ a), split
 * over m across workgroups and combined with fp32 atomics into dw, which must be
 * pre-initialised (zero, or a gradient to accumulate into).  dbias[n] = sum_m dout[m,n]
 * if dbias != NULL. */
int sage_linear_dweight(const float* dout, sage_i64 ld_dout, const float* a, sage_i64 ld_a,
                        const sage_i64* rows, sage_i64 m, const sage_i64* m_dev, int n, int k,
                        float* dw, sage_i64 ld_dw, float* dbias, void* stream);

/* The layer's two weight gradients in one launch (they share dout and come in pairs: the
 * backward of sage_linear_concat_fwd):  dw1 += dout^T @ a1[rows1],  dw2 += dout^T @ a2[rows2].
 * One grid over both problems fills the chip more evenly than two launches. */
int sage_linear_dweight2(const float* dout, sage_i64 ld_dout, const float* a1, sage_i64 ld_a1,
                         const sage_i64* rows1, int k1, float* dw1, sage_i64 ld_dw1,
                         const float* a2, sage_i64 ld_a2, const sage_i64* rows2, int k2,
                         float* dw2, sage_i64 ld_dw2, sage_i64 m, const sage_i64* m_dev, int n,
                         void* stream);
/* sage_linear_dweight2 WITHOUT the atomics where it can: when the launch takes the sliced bf16x3
 * path (n >= 256, m >= 2048, aligned operands) with 2 .. max_parts slices, slice z STORES its
 * partial sums at dw1 + z * part_stride / dw2 + z * part_stride and *parts = the number of
 * slices (host int, a function of the shapes only): the true gradient is the sum over z, which
 * sage_adam_step_parts forms while it reads it.  Otherwise *parts = 1 and the sums are ADDED into
 * dw1 / dw2 as by sage_linear_dweight2.  *parts on entry: 0, or the number an earlier call
 * returned — a different result is then an error (SAGE_EINVAL), for callers that replay recorded
 * calls.  No counterpart in the reference. */
int sage_linear_dweight2_parts(const float* dout, sage_i64 ld_dout, const float* a1, sage_i64 ld_a1,
                               const sage_i64* rows1, int k1, float* dw1, sage_i64 ld_dw1,
                               const float* a2, sage_i64 ld_a2, const sage_i64* rows2, int k2,
                               float* dw2, sage_i64 ld_dw2, sage_i64 m, const sage_i64* m_dev, int n,
                               sage_i64 part_stride, int max_parts, int* parts, void* stream);

/* ------------------------------------------------------------------------------------
 * (a-7) LSTM aggregator (layers/sage.py:67-69, :147-193): per target, the first n_in
 * neighbours' rows (padded by repeating the first, :164) are flattened to one vector (:179)
 * and fed to ONE LSTMCell(n_in*F, F) step from a zero state (:187).  The contraction
 * x_flat @ W_ih^T + (b_ih + b_hh) is sage_linear_concat_fwd; these are the other pieces.
 *   sage_segment_slots: slots[j*T + t] = row id (through node_map if given) of input j of
 *     target t — slot-major, so that slots + j*T is the row map of K-segment j of the gates
 *     contraction (x_flat is never materialised: sage_linear_concat_fwd reads segment j through
 *     it); nonempty[t] = 1.0 if the target has an in-edge else 0.0 (its output is 0, the F.pad of
 *     layers/sage.py:121-122).
 *   sage_lstm_cell0_fwd: gates [m, 4F] in torch's (i,f,g,o) order ->
 *     h = sigmoid(o) * tanh(sigmoid(i) * tanh(g)), zeroed where nonempty == 0.
 *   sage_lstm_cell0_bwd: dgates from dh (the forget-gate gradient is 0: c0 = 0).
 *     Both: bias_a / bias_b (NULL or [4F]) are added to the gates on the way in — the cell's
 *     b_ih + b_hh when the contraction ran without a bias epilogue; m_dev: optional device-side
 *     row count.
 *   sage_lstm_gather_flat: x_flat[t, j*F : (j+1)*F] = x[row(t, j)] for t < num_targets, j < n_in,
 *     row(t, j) as in sage_segment_slots (zeros for a target without in-edges), and nonempty[t]:
 *     the cell's flattened input (:179) materialised once per layer and step (0.8 GB at the
 *     Reddit shape, in 288 GB of HBM), so that the gates contraction and its weight gradient are
 *     ONE launch each over K = n_in*F with both operands contiguous in k — the static-shape
 *     engine's route (graphsage_amd/engine.py); ld_flat >= n_in*F.
 *   sage_lstm_scatter_flat: its backward, dx[row(t, j)] += d_flat[t, j*F : (j+1)*F] (fp32 atomics).
 * ---------------------------------------------------------------------------------- */
int sage_segment_slots(const sage_i64* seg_ptr, const sage_i64* src, const sage_i64* node_map,
                       sage_i64 num_targets, int n_in, sage_i64* slots /*[n_in*T]*/,
                       float* nonempty /*[T]*/, void* stream);
int sage_lstm_cell0_fwd(const float* gates, sage_i64 ld_gates, const float* nonempty, sage_i64 m,
                        const sage_i64* m_dev, int num_features, const float* bias_a,
                        const float* bias_b, float* h, sage_i64 ld_h, void* stream);
int sage_lstm_cell0_bwd(const float* gates, sage_i64 ld_gates, const float* nonempty,
                        const float* dh, sage_i64 ld_dh, sage_i64 m, const sage_i64* m_dev,
                        int num_features, const float* bias_a, const float* bias_b,
                        float* dgates, sage_i64 ld_dgates, void* stream);
int sage_lstm_gather_flat(const float* x, sage_i64 ld_x, const sage_i64* node_map,
                          const sage_i64* seg_ptr, const sage_i64* src, sage_i64 num_targets,
                          const sage_i64* num_targets_dev, int n_in, int num_features,
                          float* x_flat, sage_i64 ld_flat, float* nonempty /*[T]*/, void* stream);
int sage_lstm_scatter_flat(const float* d_flat, sage_i64 ld_flat, const sage_i64* seg_ptr,
                           const sage_i64* src, sage_i64 num_targets, const sage_i64* num_targets_dev,
                           int n_in, int num_features, float* dx, sage_i64 ld_dx, void* stream);
/* Padded slots (layers/sage.py:164: a target with fewer than n_in neighbours repeats its FIRST one).
 * The rows behind the seeds of a sampled batch have at most k < n_in neighbours by construction
 * (their edges come from a hop with fan-out k), so for them slots k .. n_in-1 all hold the first
 * neighbour's row and
 *     x_flat[t] . W_ih^T = sum_{j<k} x_j . W_j^T + x_0 . (sum_{j>=k} W_j)^T,   W_j = W_ih[:, jF:(j+1)F]:
 * the engine contracts those rows over k+1 slots instead of n_in (the same sum, re-associated).
 *   sage_lstm_pad_weight_sum:  out[r, f] = sum_{j0<=j<j1} w[r, jF + f]      (slot order, rows x F)
 *   sage_lstm_pad_grad_bcast:  dw[r, jF + f] += g[r, f] for j0<=j<j1, then g[r, f] = 0
 *     (the backward of that sum; g is the weight gradient against the summed block). */
int sage_lstm_pad_weight_sum(const float* w, sage_i64 ld_w, sage_i64 rows, int num_features, int j0, int j1,
                             float* out, sage_i64 ld_out, void* stream);
int sage_lstm_pad_grad_bcast(float* g, sage_i64 ld_g, sage_i64 rows, int num_features, int j0, int j1,
                             float* dw, sage_i64 ld_dw, void* stream);

/* ------------------------------------------------------------------------------------
 * Row-wise epilogues of the layer and the loss.
 * ---------------------------------------------------------------------------------- */
/* y = z / max(||z||_2, 1e-12)  (F.normalize(p=2, dim=-1), layers/sage.py:131-132) and,
 * if h != NULL, h = relu(y) (the in-place x.relu_() of models/graphsage.py:47-48).
 * inv_norm[m] = 1 / max(||z||, eps) is kept for the backward.  y may alias z. */
int sage_l2norm_relu_fwd(float* z, sage_i64 ld_z, sage_i64 m, const sage_i64* m_dev,
                         int n, float* y, sage_i64 ld_y, float* h, sage_i64 ld_h,
                         float* inv_norm,
                         int clear_z /* 1: z rows are zeroed after use (consume-and-clear: the
                                        producer may then run with SAGE_EPI_ACCUM); excludes aliasing */,
                         void* stream);
/* dz = (g - y * <y, g>) * inv_norm, with g = relu_mask ? dh * [y > 0] : dh.
 * Rows whose norm was clamped by eps (inv_norm == 1e12) get dz = g * inv_norm, the
 * gradient torch's clamp_min gives.  dz may alias dh. */
int sage_l2norm_relu_bwd(float* dh, sage_i64 ld_dh, const float* y, sage_i64 ld_y,
                         const float* inv_norm, int relu_mask, sage_i64 m,
                         const sage_i64* m_dev, int n, float* dz, sage_i64 ld_dz,
                         int clear_dh /* 1: dh rows are zeroed after use (consume-and-clear) */,
                         void* stream);
/* The same gradient from the PRE-normalisation rows z, for a forward that kept neither y nor
 * inv_norm (sage_logit_layer_fwd with h_root: the rows were normalised on their way into the
 * class-logit kernel; models/graphsage.py:47-48 + layers/sage.py:131-132 differentiated):
 * y = z / max(||z||, 1e-12) is recomputed per row, then dz as above.  z is handed back ZEROED
 * (its producer adds stream-K partial sums into it), dh too if clear_dh.  n a multiple of 4,
 * at most 1024, rows 16-byte aligned (else SAGE_ELIMIT); dz and dh must not alias z. */
int sage_l2norm_relu_bwd_z(float* dh, sage_i64 ld_dh, float* z, sage_i64 ld_z, int relu_mask,
                           sage_i64 m, const sage_i64* m_dev, int n, float* dz, sage_i64 ld_dz,
                           int clear_dh, void* stream);
/* sage_l2norm_relu_bwd_z of the hidden layer with the leftovers of the class-logit layer above
 * it in the same launch (the supervised step after sage_logit_layer_fwd_bwd; what
 * loss.backward() leaves in the last SAGE layer's lin_l.weight.grad / lin_r.weight.grad,
 * layers/sage.py:129-131, and the `loss` of fig2a/runners/graphsage_runners.py:43):
 *   dw_l[n2, h2] += dz2[0:m2]^T . agg2[0:m2],   dw_r[n2, h2] += dz2[0:m2]^T . h_root2[0:m2]
 *   (= sage_linear_dweight2 for n2 <= 64; rows limited by *m2_dev if given), and
 *   loss[0] = mean(row_loss[0:m2]) in a fixed order if row_loss / loss are given —
 * in workgroups of their own in front of the pass's: nothing they read is written by it (dz2,
 * agg2, h_root2 must not alias dz / dh).  h_root2 = dw_r = NULL: a layer without a root weight
 * (gcn, layers/sage.py:62-65) — dw_l only. */
int sage_l2norm_relu_bwd_z_dw(float* dh, sage_i64 ld_dh, float* z, sage_i64 ld_z, int relu_mask, sage_i64 m,
                              const sage_i64* m_dev, int n, float* dz, sage_i64 ld_dz, int clear_dh,
                              const float* dz2, sage_i64 ld_dz2, sage_i64 m2, const sage_i64* m2_dev, int n2,
                              int h2, const float* agg2, sage_i64 ld_agg2, float* dw_l, sage_i64 ld_dwl,
                              const float* h_root2, sage_i64 ld_h_root2, float* dw_r, sage_i64 ld_dwr,
                              const float* row_loss, float* loss, void* stream);

/* Gradient through the ReLU of the pool Linear, relu(pool(x_j)) at layers/sage.py:202,206 — what
 * autograd's threshold_backward computes:  dz[r, :] = out[r, :] > 0 ? dout[r, :] : 0  for r < m
 * (m_dev: optional device-side row count).  clear_dout != 0 also zeroes the consumed rows of
 * dout (consume-and-clear: dout is the buffer the next step's scatter-add accumulates into);
 * it excludes dz == dout.  Without it dz may alias dout. */
int sage_relu_bwd(float* dout, sage_i64 ld_dout, const float* out, sage_i64 ld_out, sage_i64 m,
                  const sage_i64* m_dev, int n, float* dz, sage_i64 ld_dz, int clear_dout,
                  void* stream);

/* The pool Linear of the pooling aggregators, relu(pool(x_j)) at layers/sage.py:76,202,206, over the
 * m rows x[rows[r], :] (rows NULL: x[r, :]; m_dev: optional device-side row count):
 *   p[r, :] = relu(x[rows[r], :] . w^T + bias)        w is [n, k]
 * and, in the same stores, dpre[r, :] = 0 for r < m when dpre != NULL: dpre is the buffer
 * sage_pool_max_bwd scatters the pre-activation gradient into later in the step, so the step needs
 * no fill pass over it.  sage_linear_concat_fwd with SAGE_EPI_BIAS | SAGE_EPI_RELU otherwise. */
int sage_pool_linear_fwd(const float* x, sage_i64 ld_x, const sage_i64* rows, int k, const float* w,
                         sage_i64 ld_w, const float* bias, sage_i64 m, const sage_i64* m_dev, int n,
                         float* p, sage_i64 ld_p, float* dpre, sage_i64 ld_dpre, void* stream);

/* Backward of the max_pool aggregator's reduce AND of the pool Linear's ReLU in one pass
 * (autograd of x_j = relu(pool(x_j)) -> scatter(..., reduce='max'), layers/sage.py:201-207):
 *   dpre[src[e], f] += dagg[t, f]   for e = argmax[t, f] >= 0 and agg[t, f] > 0,
 * agg being the forward's reduce output (agg[t, f] = p[src[e], f], so agg > 0 is the ReLU mask
 * of the element the gradient is routed to; no pass over p, no dense gradient of p), and
 *   dbias[f] += the column sums of what was scattered   (dbias NULL: skipped)
 * — the pool bias gradient.  dpre must hold zeros in rows < num_sources on entry (e.g. from
 * sage_pool_linear_fwd).  argmax is the [num_targets, num_features] edge-index array
 * sage_gather_reduce_fwd wrote; seg_ptr / src are that call's segments.
 * indeg: optional int32[num_sources] workspace.  With it the call first counts the live targets'
 * edges per source row; a row with ONE incoming edge is then written whole with plain stores by
 * that edge's target (the routed value or 0) and only rows shared by several targets take fp32
 * atomics (sums over targets that share an arg-max row are order-dependent in the last bit).
 * indeg NULL: every routed element is an atomic add (seg_ptr may then be NULL). */
int sage_pool_max_bwd(const float* dagg, sage_i64 ld_dagg, const float* agg, sage_i64 ld_agg,
                      const sage_i64* seg_ptr, const sage_i64* src, const sage_i64* argmax,
                      sage_i64 num_targets, const sage_i64* num_targets_dev, int num_features,
                      float* dpre, sage_i64 ld_dpre, sage_i64 num_sources, int* indeg, float* dbias,
                      void* stream);

/* mean_pool's counterpart (scatter(..., reduce='mean') over relu(pool(x_j)), layers/sage.py:201-210):
 *   dpre[src[e], f] += dagg[t, f] / deg(t)   for every edge e of target t with p[src[e], f] > 0,
 * p being the pool Linear's output (its own sign is the ReLU mask: read once per edge here instead
 * of a pass over all of p and a dense gradient of p).  dpre holds zeros in rows < num_sources on
 * entry; indeg as in sage_pool_max_bwd (rows with one incoming edge are written whole, NULL: all
 * atomics).  The pool bias gradient is the column sums of dpre (sage_linear_dweight's dbias). */
int sage_pool_mean_bwd(const float* dagg, sage_i64 ld_dagg, const float* p, sage_i64 ld_p,
                       const sage_i64* seg_ptr, const sage_i64* src, sage_i64 num_targets,
                       const sage_i64* num_targets_dev, int num_features, float* dpre, sage_i64 ld_dpre,
                       sage_i64 num_sources, int* indeg, void* stream);

/* The same two edge passes PULLED by the source rows (layer 0 of the engine's pooling step; the
 * reference's scatter of layers/sage.py:209-210 differentiated, as the two entry points above).
 * sage_pool_reverse_edges lists the live targets' edges per source row: count [num_sources + 1]
 * (workspace, left zero), rev_ptr [num_sources + 1] (row s's entries are rev[rev_ptr[s] ..
 * rev_ptr[s + 1])), an entry = target * 256 + min(offset of the edge in the target's segment, 255);
 * entries of one row come in no fixed order (atomic slots).  It needs nothing but the sampled
 * batch, so a caller can run it ahead of the step.  Limits: num_targets < 2^23, rev_cap < 2^31;
 * edges that would not fit rev_cap are dropped.  workspace: sage_pool_reverse_edges_workspace_bytes.
 * sage_pool_max_fwd: sage_gather_reduce_fwd(max) over rows of p = relu(...) that records the
 * arg-max edge of (target, column) as ONE byte in arg8 (rows ld_arg8 bytes apart, 4-byte aligned):
 * the edge's offset in the segment, or 255 where no gradient flows (empty segment, maximum <= 0 —
 * the relu mask of the routed element); max_segment = the caller's bound on a segment's length,
 * refused above 255.
 * sage_pool_max_bwd_pull / sage_pool_mean_bwd_pull: one wave per source row adds up what its
 * in-edges route to it (max: dagg[t, f] where arg8[t, f] == the entry's offset; mean: dagg[t] /
 * deg(t), masked by the row's own p > 0) and STORES rows [0, num_sources) (device-side count
 * num_sources_dev limits them) of dpre, rows without in-edges as zeros: no atomics, no zero-fill
 * expected; rev_cap as given to sage_pool_reverse_edges (list positions beyond it are not read:
 * the edges it dropped contribute nothing).  Rows of dagg, p, dpre 16-byte aligned and padded to
 * a multiple of 4 columns (the padding of dpre is written with zeros). */
size_t sage_pool_reverse_edges_workspace_bytes(sage_i64 num_sources);
int sage_pool_reverse_edges(const sage_i64* seg_ptr, const sage_i64* src, sage_i64 num_targets,
                            const sage_i64* num_targets_dev, sage_i64 num_sources, int* count,
                            int* rev_ptr, int* rev, sage_i64 rev_cap, void* workspace,
                            size_t workspace_bytes, void* stream);
int sage_pool_max_fwd(const float* p, sage_i64 ld_p, const sage_i64* seg_ptr, const sage_i64* src,
                      sage_i64 num_targets, const sage_i64* num_targets_dev, int num_features,
                      int max_segment, float* out, sage_i64 ld_out, unsigned char* arg8,
                      sage_i64 ld_arg8, void* stream);
int sage_pool_max_bwd_pull(const float* dagg, sage_i64 ld_dagg, const unsigned char* arg8,
                           sage_i64 ld_arg8, const int* rev_ptr, const int* rev, sage_i64 rev_cap,
                           sage_i64 num_sources, const sage_i64* num_sources_dev, int num_features,
                           float* dpre, sage_i64 ld_dpre, void* stream);
int sage_pool_mean_bwd_pull(const float* dagg, sage_i64 ld_dagg, const float* p, sage_i64 ld_p,
                            const sage_i64* seg_ptr, const int* rev_ptr, const int* rev,
                            sage_i64 rev_cap, sage_i64 num_sources, const sage_i64* num_sources_dev,
                            int num_features, float* dpre, sage_i64 ld_dpre, void* stream);

/* Mean softmax cross-entropy over m rows and its gradient (F.cross_entropy of
 * experiments/fig2a/runners/graphsage_runners.py:43 / nn.CrossEntropyLoss of
 * trainers/node_level_graphsage_trainers.py:41).  labels are gathered through
 * label_rows (NULL = identity): label of row i = labels[label_rows[i]].
 * loss[0] = mean_i( logsumexp(logits[i]) - logits[i, label_i] );
 * dlogits = (softmax - onehot) * grad_scale / m.  loss is overwritten (one workgroup
 * reduces deterministically).  Limit: n <= 4096. */
int sage_softmax_xent(const float* logits, sage_i64 ld_logits, const sage_i64* labels,
                      const sage_i64* label_rows, sage_i64 m, int n, float grad_scale,
                      float* loss, float* dlogits, sage_i64 ld_dlogits, void* stream);

/* The same loss and gradient with one wave per row over many workgroups (sage_softmax_xent walks
 * the rows with ONE workgroup: 35 us for 512 rows): the per-row losses go to row_loss[0 .. m)
 * (caller buffer) and are averaged into loss[0] in a fixed order, so the result is bitwise
 * reproducible and equal to a sequential sum of the row losses / m.  Same label handling
 * (label_rows optional; a label outside [0, n) contributes 0 to the loss). */
int sage_softmax_xent_rows(const float* logits, sage_i64 ld_logits, const sage_i64* labels,
                           const sage_i64* label_rows, sage_i64 m, int n, float grad_scale,
                           float* row_loss, float* loss, float* dlogits, sage_i64 ld_dlogits,
                           void* stream);

/* The three calls above fused for the LAST layer of the training step (everything is
 * row-local): y = normalize(z), loss = mean CE(y, labels), dz = d loss / d z (through the
 * soft-max and the normalisation).  row_loss[m] receives the per-row losses (workspace and
 * diagnostic), loss[0] their mean (fixed-order reduction).  y / inv_norm may be NULL. */
int sage_logits_xent_fused(float* z, sage_i64 ld_z, const sage_i64* labels,
                           const sage_i64* label_rows, sage_i64 m, int n, float grad_scale,
                           float* y, sage_i64 ld_y, float* inv_norm, float* dz, sage_i64 ld_dz,
                           float* row_loss /*[m]*/, float* loss /*[1]*/,
                           int clear_z /* 1: z rows are zeroed after use */, void* stream);

/* The multi-label form (PPI: n independent sigmoid outputs per node, BCEWithLogitsLoss on the
 * normalised logits — examples/ppi_graphsage_supervised.py:30, trainers/
 * graph_level_graphsage_trainers.py:36-37): y = normalize(z), loss = mean over the m x n elements
 * of max(y,0) - y*t + log(1 + exp(-|y|)) with t = targets[label_rows ? label_rows[r] : r, f]
 * (float 0 / 1), dz = d loss / d z.  row_loss[r] = the row's sum / n.  row_stats (optional,
 * sage_i64 [m, 3]): per row (true positive, false positive, false negative) counts of the
 * prediction y > 0 against t > 0.5 — micro-F1's ingredients (graph_level_graphsage_trainers.py:
 * 57-58).  dz == NULL: forward only (evaluation); loss == NULL: no mean; y / inv_norm optional. */
int sage_logits_bce_fused(float* z, sage_i64 ld_z, const float* targets, sage_i64 ld_targets,
                          const sage_i64* label_rows, sage_i64 m, int n, float grad_scale,
                          float* y, sage_i64 ld_y, float* inv_norm, float* dz, sage_i64 ld_dz,
                          float* row_loss /*[m]*/, float* loss /*[1]*/, sage_i64* row_stats /*[m,3]*/,
                          int clear_z, void* stream);

/* The class-logit layer of the supervised step — the last SAGE layer on the m seed rows
 * (layers/sage.py:99-134 called from models/graphsage.py:44-49, sliced [:batch_size] at
 * fig2a/runners/graphsage_runners.py:41-43) — in one launch each way, for reduce = mean or sum
 * with a root weight, n <= 64 outputs, width h a multiple of 4 and <= 1024, n * 2h * 4 bytes of
 * weights within the 160 KB of LDS (else SAGE_ELIMIT: run the separate entry points).
 *   forward:  agg[t] = reduce_{e in [seg_ptr[t], seg_ptr[t+1])} h_rows[src[e]]   (kept for the
 *             weight gradients);  z = agg[t] . w_l^T + h_rows[t] . w_r^T;  y = z / max(|z|, 1e-12);
 *             row_loss[t] = CE(y, labels[label_rows ? label_rows[t] : t]);
 *             dz[t] = grad_scale / m * d CE / d z   (exactly sage_gather_reduce_fwd +
 *             sage_linear_concat_fwd + sage_logits_xent_fused without the loss mean; y, inv_norm
 *             optional)
 *   backward: dh[src[e]] += (dz[t] . w_l) (/ segment length for mean),  dh[t] += dz[t] . w_r
 *             (sage_linear_dinput2 + sage_gather_reduce_bwd; dh is accumulated with fp32 atomics
 *             and must hold zeros or a gradient to add to), and, if row_loss / loss are given,
 *             loss[0] = mean(row_loss[0:m]) in a fixed order. */
int sage_logit_layer_fwd(const float* h_rows, sage_i64 ld_h, const sage_i64* seg_ptr, const sage_i64* src,
                         sage_i64 m, int h, int reduce, const float* w_l, sage_i64 ld_wl,
                         const float* w_r, sage_i64 ld_wr, int n, const sage_i64* labels,
                         const sage_i64* label_rows, float grad_scale, float* agg, sage_i64 ld_agg,
                         float* y, sage_i64 ld_y, float* inv_norm, float* dz, sage_i64 ld_dz,
                         float* row_loss,
                         float* h_root /* NULL, or [m, h]: h_rows then holds the previous layer's
                                          PRE-activation rows — every row read goes through
                                          relu(z / max(||z||, 1e-12)) first (what
                                          sage_l2norm_relu_fwd would have stored), and the seeds'
                                          own activations are written here for the root term's
                                          weight gradient; must not alias h_rows */,
                         sage_i64 ld_h_root, void* stream);
int sage_logit_layer_bwd(const float* dz, sage_i64 ld_dz, const sage_i64* seg_ptr, const sage_i64* src,
                         sage_i64 m, int h, int reduce, const float* w_l, sage_i64 ld_wl,
                         const float* w_r, sage_i64 ld_wr, int n, float* dh, sage_i64 ld_dh,
                         const float* row_loss, float* loss, void* stream);
/* Forward AND the input gradient of sage_logit_layer_bwd in ONE launch: every seed's wave runs
 * its backward right behind its forward (a seed's input gradient needs nothing but its own dz;
 * the weights are staged in LDS once).  Arguments as sage_logit_layer_fwd + dh; dz, agg, h_root
 * and row_loss are still written for the weight gradients and the loss mean, which
 * sage_l2norm_relu_bwd_z_dw (or sage_linear_dweight2 + a mean) supplies.  This entry (and
 * sage_logit_layer_infer) also takes reduce = SAGE_REDUCE_MAX (layers/sage.py:195-199, scatter max):
 * element-wise maximum, first edge on ties, 0 for an empty segment; the backward sends every
 * column's gradient to the row that supplied the maximum (the arg-max never leaves the wave);
 * and reduce = SAGE_REDUCE_GCN (layers/sage.py:114-117: neighbours and one self-loop averaged, no
 * root weight): z = agg . w_l^T — w_r is ignored (pass w_l) — and the target's own row receives
 * the self-loop's share of the gradient. */
int sage_logit_layer_fwd_bwd(const float* h_rows, sage_i64 ld_h, const sage_i64* seg_ptr, const sage_i64* src,
                             sage_i64 m, int h, int reduce, const float* w_l, sage_i64 ld_wl,
                             const float* w_r, sage_i64 ld_wr, int n, const sage_i64* labels,
                             const sage_i64* label_rows, float grad_scale, float* agg, sage_i64 ld_agg,
                             float* y, sage_i64 ld_y, float* inv_norm, float* dz, sage_i64 ld_dz,
                             float* row_loss, float* h_root, sage_i64 ld_h_root, float* dh, sage_i64 ld_dh,
                             const int* writers /* NULL, or sage_count_row_writers' output for these segments:
                                                   rows of dh with ONE writer are stored instead of added
                                                   (dh must then hold zeros on entry) */,
                             void* stream);
/* The same one-launch forward + backward for the last layer of a POOLING aggregator (reference
 * layers/sage.py:201-210 with :124-132 behind it): the neighbour rows are read from p_rows =
 * relu(pool(h)) (one row per source node, indexed like h_rows), the root row from h_rows[t]; both
 * as they are (no normalisation on the way in).  reduce = mean or max.  Backward: dh[t] (rows <
 * m) = dz . W_r, stored; dpre[src] += the neighbours' share of dz . W_l THROUGH the relu — mean:
 * (dz . W_l) / deg where p_rows[src] > 0, max: each column to the row that supplied the maximum,
 * where that maximum is positive.  dpre holds zeros on entry.  writers as in
 * sage_logit_layer_fwd_bwd (sage_count_row_writers over the seeds' segments): a row of dpre whose
 * only in-edge is this seed's is stored instead of added.  Weight gradients and the loss mean:
 * from dz / agg / h_rows[:m] / row_loss, as behind sage_logit_layer_fwd_bwd (riding on
 * sage_l2norm_relu_bwd_z_dw, or sage_linear_dweight2 + a mean of row_loss).  h = 256: four waves
 * per seed; other widths: one. */
int sage_logit_layer_pool_fwd_bwd(const float* h_rows, sage_i64 ld_h, const float* p_rows, sage_i64 ld_p,
                                  const sage_i64* seg_ptr, const sage_i64* src, sage_i64 m, int h, int reduce,
                                  const float* w_l, sage_i64 ld_wl, const float* w_r, sage_i64 ld_wr, int n,
                                  const sage_i64* labels, const sage_i64* label_rows, float grad_scale,
                                  float* agg, sage_i64 ld_agg, float* y, sage_i64 ld_y, float* inv_norm,
                                  float* dz, sage_i64 ld_dz, float* row_loss, float* dh, sage_i64 ld_dh,
                                  float* dpre, sage_i64 ld_dpre, const int* writers, void* stream);

/* writers[r] (int32, r < rows) = the number of waves of sage_logit_layer_fwd_bwd that add into row r
 * of dh: the edges e in [seg_ptr[0], seg_ptr[m]) with src[e] == r, plus one for r < m (a seed's
 * own row).  max_edges bounds seg_ptr[m] - seg_ptr[0] (the launch is sized by it).  Two short
 * launches; the engine runs them behind the sampler on its stream.  No counterpart in the
 * reference (torch_scatter adds every edge atomically). */
int sage_count_row_writers(const sage_i64* seg_ptr, const sage_i64* src, sage_i64 m, sage_i64 max_edges,
                           int* writers, sage_i64 rows, void* stream);
/* sage_logit_layer_bwd with the layer's two weight gradients in the same launch (what
 * loss.backward() leaves in lin_l.weight.grad / lin_r.weight.grad of the last SAGE layer,
 * layers/sage.py:129-131):  dw_l[n, h] += dz[0:m]^T . agg[0:m],  dw_r[n, h] += dz[0:m]^T . h_root[0:m]
 * (= sage_linear_dweight2 for n <= 64; rows limited by *m_dev if given) — they run in workgroups
 * of their own behind the seeds' workgroups, which leave half of the CUs idle. */
int sage_logit_layer_bwd_dw(const float* dz, sage_i64 ld_dz, const sage_i64* seg_ptr, const sage_i64* src,
                            sage_i64 m, const sage_i64* m_dev, int h, int reduce, const float* w_l,
                            sage_i64 ld_wl, const float* w_r, sage_i64 ld_wr, int n, float* dh,
                            sage_i64 ld_dh, const float* row_loss, float* loss, const float* agg,
                            sage_i64 ld_agg, float* dw_l, sage_i64 ld_dwl, const float* h_root,
                            sage_i64 ld_h_root, float* dw_r, sage_i64 ld_dwr, void* stream);
/* The same layer forward-only, for the inference pass (GraphSAGE.inference,
 * models/graphsage.py:51-68, and the `y_hat.argmax(dim=-1)` of the runners' test(),
 * fig2a/runners/graphsage_runners.py:57-66 / trainers/node_level_graphsage_trainers.py:56):
 *   y[t]    = normalised logits of seed row t as above (optional),
 *   pred[t] = the first class attaining max_c y[t, c] — torch.argmax's choice (optional; one of
 *             y / pred is required).
 * No labels, loss, gradient or kept aggregate.  pre_activation != 0: h_rows holds the previous
 * layer's pre-activation rows, normalised and rectified on their way in (sage_logit_layer_fwd's
 * h_root mode without the stored seed rows).  Same shape limits (SAGE_ELIMIT). */
int sage_logit_layer_infer(const float* h_rows, sage_i64 ld_h, int pre_activation, const sage_i64* seg_ptr,
                           const sage_i64* src, sage_i64 m, int h, int reduce, const float* w_l,
                           sage_i64 ld_wl, const float* w_r, sage_i64 ld_wr, int n, float* y, sage_i64 ld_y,
                           sage_i64* pred, void* stream);

/* (f-2) unsupervised loss of trainers/base_trainers.py:179-181 on m rows of width n:
 *   loss[0] = -mean logsigmoid(<out_i, pos_i>) - mean logsigmoid(-<out_i, neg_i>)
 * and its gradient times grad_scale: d_out is ACCUMULATED (it may already hold the supervised
 * term's gradient, node_level_graphsage_trainers.py:41), d_pos / d_neg are overwritten.
 * row_loss[m] is scratch for the deterministic mean. */
int sage_unsup_loss(const float* out, const float* pos, const float* neg, sage_i64 ld, sage_i64 m,
                    int n, float grad_scale, float* d_out, float* d_pos, float* d_neg,
                    sage_i64 ld_d, float* row_loss, float* loss, void* stream);

/* Adam (torch.optim.Adam defaults semantics, no amsgrad, L2 weight_decay added to the
 * gradient) on one flat parameter buffer (optimizer.step() of
 * experiments/fig2a/runners/graphsage_runners.py:45).  The step count t-1 is read from the
 * device word step_in and t is written to step_out (so a captured graph advances it on
 * replay); when the two are different words one kernel does both, when they are the same a
 * second one-thread kernel bumps the counter.  zero_grad = 1 clears grad after use
 * (optimizer.zero_grad() of :40 for the next step). */
int sage_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                   sage_i64 numel, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float grad_scale, const sage_i64* step_in,
                   sage_i64* step_out, int zero_grad, void* stream);
/* The same step with a HIP event (hipEvent_t) attached to its launch as the completion event
 * (hipExtLaunchKernelGGL's stop event): other streams can hipStreamWaitEvent on it as if it had
 * been recorded behind the kernel, without a marker in this stream's queue — the engine's
 * end-of-step signal for the sampler / gather streams.  No counterpart in the reference. */
int sage_adam_step_signal(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                   sage_i64 numel, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float grad_scale, const sage_i64* step_in,
                   sage_i64* step_out, int zero_grad, void* done_event, void* stream);
/* Adam where the gradient of flat elements [region_off, region_off + region_len) is the sum of
 * `parts` partial buffers ws + s * part_stride (sage_linear_dweight2_parts; parts = 1: the
 * gradient itself, kept outside `grad` — its producer ADDS into it, so it is cleared after use
 * if zero_grad); grad is not touched for those elements.  step_in and
 * step_out must be different words.  done_event: NULL or as in sage_adam_step_signal. */
int sage_adam_step_parts(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                         sage_i64 numel, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float grad_scale, const sage_i64* step_in,
                         sage_i64* step_out, int zero_grad, float* ws, sage_i64 part_stride,
                         int parts, sage_i64 region_off, sage_i64 region_len, void* done_event,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAGE_HIP_H */
