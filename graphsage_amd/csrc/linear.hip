// (a-4) self||neighbour Linear on the matrix cores, forward and backward.
//
// Replaces the two bias-free torch_geometric Linear calls of reference layers/sage.py:124-129
// (out = lin_l(agg); out += lin_r(x_r)) by ONE contraction over the concatenated K
// dimension [agg || x_r] . [W_l || W_r]^T, and the pool Linear (+bias, ReLU) of
// layers/sage.py:76,202,206; the backward provides what autograd derives for them.
//
// Arithmetic is exact fp32 on the MFMA pipe: v_mfma_f32_32x32x2_f32 (gfx950 has no
// xf32/TF32 path).  Every output element is an fp32 fma chain over k, so results agree
// with an fp32 reference to summation-order rounding.  (Opt-in, SAGE_GEMM_X3=1: the large forward
// and weight-gradient contractions run on the bf16 MFMA with each fp32 operand split into three
// bf16 terms — gemm_x3_kernel / gemm_x3_dw_kernel below; fp32-accurate, other bits.)
//
// One tile kernel serves the three contractions.  In "C[M][N] = sum_k A(m,k) B(k,n)" form:
//   forward   : A(m,k) = a[rows(m)][k]   (k contiguous)   B(k,n) = w[n][k]   (k contiguous)
//   d input   : A(m,k) = dout[m][k]      (k contiguous)   B(k,n) = w[k][n]   (n contiguous)
//   d weight  : A(m,k) = dout[k][m]      (m contiguous)   B(k,n) = a[rows(k)][n]
// Tiles are staged global -> registers -> LDS, double buffered, one barrier per K-step of 32.
// The LDS image of an operand keeps the operand's own contiguous dimension contiguous:
//   k contiguous in memory (forward, dout of d-input): [m][BK+4] rows, written with one 16-byte
//     store per loaded vector and read back as ONE ds_read_b128 per lane per 8 k — lanes 0-31
//     take k = 8g..8g+3, lanes 32-63 k = 8g+4..8g+7 of their row, which feeds four MFMAs.
//     An LDS-fed 32x32x2 MFMA loop sustains 142-148 TF/s with these reads against 106-124 TF/s
//     with one ds_read_b32 per operand per MFMA (scratch/lds_mfma.hip, MI355X);
//   m/n contiguous in memory (d-weight): k-major [BK][MN+4] rows, ds_read_b32 per k.
// Both operands of one MFMA use the same k slot assignment, so any pairing of the two layouts
// is valid.  The optional row map lets the root operand come straight out of the global
// feature table (no materialised batch.x).
#include "common.h"
#include "skinny_dw.h"
#include <type_traits>

namespace sage {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int BK = 32;

// Split-K over blockIdx.z with a DEVICE-side contraction length: the host sized the grid for the
// capacity (its k_chunk), the actual count can be much smaller (the pooling aggregators contract
// over all sampled nodes: capacity 141k rows, ~81k sampled) — deal the actual length evenly over
// the gridDim.z slices instead of leaving the upper ones empty.
__device__ inline int split_chunk(int host_chunk, bool device_count, int k_lim) {
  if (!device_count) return host_chunk;
  const int per = (k_lim + static_cast<int>(gridDim.z) - 1) / static_cast<int>(gridDim.z);
  return (per + BK - 1) / BK * BK;
}

struct Operand {
  const float* p;          // base pointer
  int64_t ld;              // row stride in floats
  const int64_t* rowmap;   // optional map applied to the operand's global row index
  int vec;                 // widest safe vector width for its rows (4, 2 or 1)
};

struct Segment {
  Operand a, b;
  int k;  // contraction length of this segment
};

struct GemmArgs {
  Segment seg[2];
  int nseg;
  int64_t m;               // rows of C (upper bound when m_dev != NULL)
  const int64_t* m_dev;    // optional device-side row count (limits M)
  const int64_t* k_dev;    // optional device-side contraction length (limits K of seg 0)
  int n;                   // cols of C
  float* out;
  int64_t ld_out;
  const float* bias;
  int epilogue;            // SAGE_EPI_* flags
  int atomic_out;          // 1: atomicAdd into out (split-K / accumulate)
  int rmw_out;             // 1: out += product by plain load-add-store (whole tiles: one owner per element)
  int whole;               // bf16x3 forward: workgroups take WHOLE tiles (no partial pieces, no atomics),
                           // so the bias / relu / rmw epilogues apply
  int k_chunk;             // split-K: contraction elements per blockIdx.z (multiple of BK)
  // stream-K (forward only): a 1-D grid of gridDim.x workgroups shares the K-steps of ALL tiles
  // evenly; a workgroup walks a contiguous range of (tile, K-step) pairs, stores a tile it
  // covers completely and adds the others with fp32 atomics (output zeroed beforehand)
  int streamk;
  float* colsum;           // optional: colsum[m] += sum_k A(m,k) (dbias), only blockIdx.y == 0
  // batched mode (d-weight pair): seg[0] and seg[1] are two INDEPENDENT problems that share the
  // A operand; blockIdx.y < ny0 works on problem 0 (out, n), the rest on problem 1 (out2, n2)
  int batched;
  int ny0;
  int n2;
  float* out2;
  int64_t ld_out2;
  // whole-tile epilogues only (one owner per element): a second buffer of out's shape whose
  // rows < m are handed back zeroed by the same stores (sage_pool_linear_fwd)
  float* zero2;
  int64_t ld_zero2;
  // 1: workgroups that run on one XCD (dispatch order: linear workgroup id modulo 8) take
  // neighbouring pieces of the problem, so that operand pieces several of them need are fetched
  // into that XCD's L2 once (SAGE_GEMM_XCD=0 keeps the plain order: A/B switch)
  int xcd_map;
  // weight gradients (gemm_x3_dw_kernel) only: > 0 — split-K slice z STORES its partial tile at
  // out + z * part_stride (out2 likewise) instead of adding it into out with fp32 atomics; the
  // consumer sums the slices (sage_adam_step_parts).  fp32 atomics run at ~1.2 TB/s chip-wide, a
  // sixth of plain stores: 25 slices x 1.2 MB cost the headline's launch 10 of its 64 us.
  int64_t part_stride;
};

// ---- global -> register staging of one operand tile ---------------------------------
// KC  (contraction dim contiguous in memory): tile = MN rows x BK cols, row index = m/n.
// !KC (m/n contiguous in memory):             tile = BK rows x MN cols, row index = k.
// Everything about the thread -> (row, col) mapping is compile-time, and nothing consumes a
// loaded value before store(): the loads of K-step s+1 stay in flight while the MFMAs of
// step s run.  Row-map lookups never sit in front of a data load on the critical path: for KC
// operands the mapped row pointers are fixed for the whole segment (begin()), for !KC
// operands the row ids of step s+1 are fetched while step s is being loaded.
template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float __attribute__((ext_vector_type(4))); };
template <> struct VecT<2> { using type = float __attribute__((ext_vector_type(2))); };
template <> struct VecT<1> { using type = float; };

template <bool KC, int MN, int VEC>
struct TileLoader {
  static constexpr int UNITS = MN * BK / VEC;
  static constexpr int PASSES = UNITS / kThreads;
  static constexpr int UPR = (KC ? BK : MN) / VEC;  // vector units per tile row
  static_assert(PASSES >= 1, "tile too small for the thread block");
  using VT = typename VecT<VEC>::type;

  VT v[2][PASSES];            // two register sets: the loads of two K-steps are in flight
  int64_t rowoff[PASSES];     // KC: offset of the (clamped, mapped) row in floats from op.p
  int rows[PASSES];           // !KC: (mapped) row ids of the K-step issued next (< 2^31)
  int64_t coff[PASSES];       // !KC: clamped column offset of the pass
  Operand op;
  int64_t mn0, mn_lim;
  int k_end;

  __device__ inline int trow(int p) const { return (threadIdx.x + p * kThreads) / UPR; }
  __device__ inline int tcol(int p) const { return ((threadIdx.x + p * kThreads) % UPR) * VEC; }

  // Rows / columns of the NON-contracted dimension beyond the limit are clamped to valid
  // addresses and NOT zeroed: they only feed output rows / columns that the epilogue does not
  // store.  Only the tail of the contraction is zeroed (store_pass).
  __device__ inline void begin(const Operand& o, int64_t mn0_, int64_t mn_lim_, int k_beg, int k_end_) {
    op = o; mn0 = mn0_; mn_lim = mn_lim_; k_end = k_end_;
    // all row-map lookups are issued back to back (clamped indices, no branch between them)
    int64_t ids[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int64_t r = KC ? (mn0 + trow(p)) : (static_cast<int64_t>(k_beg) + trow(p));
      const int64_t lim = KC ? mn_lim : static_cast<int64_t>(k_end);
      ids[p] = r < lim ? r : lim - 1;
    }
    if (op.rowmap != nullptr) {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) ids[p] = op.rowmap[ids[p]];
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      if (KC) {
        rowoff[p] = ids[p] * op.ld;
      } else {
        rows[p] = static_cast<int>(ids[p]);
        const int64_t cmax = (mn_lim - 1) / VEC * VEC;
        const int64_t c = mn0 + tcol(p);
        coff[p] = c < cmax ? c : cmax;
      }
    }
  }

  // Request pass p of the tile whose first contraction index is k0.  The load is
  // UNCONDITIONAL (clamped address).  Straight-line loads (no exec-masked branches) let the
  // compiler count vmcnt exactly, so the store of one register set does not wait for the
  // younger set.  Safe for any k0; passes must be issued for consecutive K-steps.
  template <int SET>
  __device__ inline void issue_pass(int p, int k0) {
    typedef const VT __attribute__((address_space(1)))* GlobalVec;   // global, not flat
    const float* src;
    if (KC) {
      const int kmax = (k_end - 1) / VEC * VEC;
      const int kk = k0 + tcol(p);
      src = op.p + rowoff[p] + (kk < kmax ? kk : kmax);
    } else {
      src = op.p + static_cast<int64_t>(rows[p]) * op.ld + coff[p];   // one 32x32->64 multiply-add
    }
    v[SET][p] = *((GlobalVec)(src));
    if (!KC) {
      // row id of the NEXT K-step; it arrives while this step's data is in flight
      const int r = k0 + BK + trow(p);
      const int rc = r < k_end ? r : k_end - 1;
      rows[p] = op.rowmap != nullptr ? static_cast<int>(op.rowmap[rc]) : rc;
    }
  }
  template <int SET>
  __device__ inline void issue(int k0) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) issue_pass<SET>(p, k0);
  }

  // write pass p of the tile requested by issue(k0) into LDS.  TAIL = false: the step lies
  // entirely below k_end, nothing to zero.
  template <int LDS_LD, int SET, bool TAIL = true>
  __device__ inline void store_pass(int p, int k0, float* lds) const {
    float e[VEC];
    if constexpr (VEC == 4) { e[0] = v[SET][p].x; e[1] = v[SET][p].y; e[2] = v[SET][p].z; e[3] = v[SET][p].w; }
    else if constexpr (VEC == 2) { e[0] = v[SET][p].x; e[1] = v[SET][p].y; }
    else { e[0] = v[SET][p]; }
    if constexpr (TAIL) {
      if (KC) {
        const int c0 = k0 + tcol(p);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (c0 + j >= k_end) e[j] = 0.f;
        }
      } else {
        const bool row_ok = k0 + trow(p) < k_end;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (!row_ok) e[j] = 0.f;
        }
      }
    }
    // both layouts keep the loaded vector contiguous: [mn][k] for KC, [k][mn] otherwise
    float* dst = lds + trow(p) * LDS_LD + tcol(p);
    if constexpr (VEC == 4) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      *reinterpret_cast<f4v*>(dst) = f4v{e[0], e[1], e[2], e[3]};
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) dst[j] = e[j];
    }
  }
  template <int LDS_LD, int SET, bool TAIL = true>
  __device__ inline void store(int k0, float* lds) const {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) store_pass<LDS_LD, SET, TAIL>(p, k0, lds);
  }

  // bf16x3 mode, k-major operand: the tile as three bf16 planes [k][row_bytes] with this
  // operand's columns starting at column col0 of the plane (plane q at lds + q * plane_bytes);
  // rows of the contraction tail are zeroed.  Defined after split_bf16x3 below.
  template <int SET>
  __device__ inline void store_x3(int k0, unsigned char* lds, int col0, int row_bytes, int plane_bytes) const;

  // k-major operand: add the rows (< k_end) of the tile requested by issue(k0) into the thread's
  // column sums — every pass of a thread covers the same 4 columns (kThreads is a multiple of UPR)
  template <int SET>
  __device__ inline void colsum_add(int k0, float (&cs)[4]) const {
    static_assert(!KC && VEC == 4 && kThreads % UPR == 0, "k-major operand in 16-byte vectors");
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const bool row_ok = k0 + trow(p) < k_end;
      const VT x = v[SET][p];
      cs[0] += row_ok ? x.x : 0.f;
      cs[1] += row_ok ? x.y : 0.f;
      cs[2] += row_ok ? x.z : 0.f;
      cs[3] += row_ok ? x.w : 0.f;
    }
  }
};

// ---- bf16x3 mode (opt-in, SAGE_GEMM_X3=1): an fp32 value as three bf16 terms ------------
// hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), all round-to-nearest: the residuals
// are exact in fp32 and hi + mid + lo == x (24 significand bits in three 8-bit pieces), so
// a.b = sum of nine exact bf16 x bf16 products; the kernel accumulates six and drops mid.lo, lo.mid
// and lo.lo: at most 2^-24 of the product (|mid| <= 2^-8 |x|, |lo| <= 2^-17 |x|), 2^-28 on average
// (tests/test_x3_split_cpu.py).  Two values per call, packed as the MFMA wants them (even k low half).
__device__ inline unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline void split_bf16x3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(r0, r1);
  const float q0 = r0 - __uint_as_float(mid << 16), q1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk_bf16(q0, q1);
}
// k-contiguous planes (forward): one LDS row = BK bf16 = 64 bytes, no padding; the four 16-byte
// slots of row r are stored at slot ^ ((-((r & 15) >> 2)) & 3), which makes the b128 fragment
// reads of v_mfma_f32_16x16x32_bf16 (lane = row r & 15, slot = lane >> 4) conflict-free
constexpr int kX3RowBytes = 2 * BK;
__device__ inline int x3_slot(int row, int slot) { return ((slot ^ (-((row & 15) >> 2))) & 3) << 4; }

// k-major planes (weight gradients): byte offset of column `col` (a multiple of 4) in row k.  The
// 32-byte blocks (16 columns) of rows with (k >> 3) odd are swapped in pairs, so that the two
// 16-lane groups of a half-wave — k = 8 kg .. for kg even and odd, same 16 columns — hit
// different banks in the transposing fragment reads of the 16x16x32 MFMA.
__device__ inline int x3_dw_offset(int k, int col, int row_bytes) {
  return k * row_bytes + (((col >> 4) ^ ((k >> 3) & 1)) << 5) + ((col * 2) & 31);
}

template <bool KC, int MN, int VEC>
template <int SET>
__device__ inline void TileLoader<KC, MN, VEC>::store_x3(int k0, unsigned char* lds, int col0, int row_bytes,
                                                         int plane_bytes) const {
  static_assert(!KC && VEC == 4, "k-major operand staged in 16-byte vectors");
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const VT x = v[SET][p];
    float e[4] = {x.x, x.y, x.z, x.w};
    const bool row_ok = k0 + trow(p) < k_end;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!row_ok) e[j] = 0.f;
    }
    unsigned h0, m0, l0, h1, m1, l1;
    split_bf16x3(e[0], e[1], h0, m0, l0);
    split_bf16x3(e[2], e[3], h1, m1, l1);
    unsigned char* dst = lds + x3_dw_offset(trow(p), col0 + tcol(p), row_bytes);
    *reinterpret_cast<u2v*>(dst) = u2v{h0, h1};
    *reinterpret_cast<u2v*>(dst + plane_bytes) = u2v{m0, m1};
    *reinterpret_cast<u2v*>(dst + 2 * plane_bytes) = u2v{l0, l1};
  }
}

// Forward operand (contraction dim contiguous) with up to TWO K-segments behind one pipeline:
// the K-loop runs over "unified" steps 0 .. steps0+steps1-1, the first steps0 of them in segment
// 0, so the loads of segment 1 are already in flight while the last steps of segment 0 multiply
// (no second pipeline fill, and the row-map lookups of both segments are issued together before
// the loop).  Everything that depends on the segment is a select on a wave-uniform condition —
// no branch in the loop (see the K-loop notes in gemm_kernel).
template <int MN, int VEC>
struct DualLoader {
  static constexpr int UNITS = MN * BK / VEC;
  static constexpr int PASSES = UNITS / kThreads;
  static constexpr int UPR = BK / VEC;
  static_assert(PASSES >= 1, "tile too small for the thread block");
  using VT = typename VecT<VEC>::type;

  // (every member is indexed with compile-time constants only, and a segment-dependent value
  // is selected AFTER both candidates were read into locals: a conditional expression over two
  // array elements becomes a load through a selected pointer, which sends the whole struct to
  // scratch memory)
  VT va[PASSES], vb[PASSES];           // the two register sets
  int64_t rowoff0[PASSES], rowoff1[PASSES];   // offset of the (clamped, mapped) row, per segment
  const float* base0;
  const float* base1;
  int kbeg0, kend0, kbeg1, kend1;
  int steps0;                          // K-steps of this piece that lie in segment 0

  __device__ inline int trow(int p) const { return (threadIdx.x + p * kThreads) / UPR; }
  __device__ inline int tcol(int p) const { return ((threadIdx.x + p * kThreads) % UPR) * VEC; }

  // rows of the tile: fixed for every K-range of this tile
  __device__ inline void begin(const Operand& o0, const Operand& o1, int64_t mn0, int64_t mn_lim) {
    int64_t ids[PASSES], id0[PASSES], id1[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int64_t r = mn0 + trow(p);
      ids[p] = r < mn_lim ? r : mn_lim - 1;
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) id0[p] = o0.rowmap != nullptr ? o0.rowmap[ids[p]] : ids[p];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) id1[p] = o1.rowmap != nullptr ? o1.rowmap[ids[p]] : ids[p];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      rowoff0[p] = id0[p] * o0.ld;
      rowoff1[p] = id1[p] * o1.ld;
    }
    base0 = o0.p;
    base1 = o1.p;
  }

  // contraction ranges [kb, ke) of the two segments for this piece; returns the number of K-steps
  __device__ inline int range(int kb0, int ke0, int kb1, int ke1) {
    const int n0 = ke0 > kb0 ? (ke0 - kb0 + BK - 1) / BK : 0;
    const int n1 = ke1 > kb1 ? (ke1 - kb1 + BK - 1) / BK : 0;
    const bool unused1 = n1 == 0;
    // segment 1 unused: requests past the last step are addressed through it, keep it valid
    base1 = unused1 ? base0 : base1;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int64_t r0 = rowoff0[p], r1 = rowoff1[p];
      rowoff1[p] = unused1 ? r0 : r1;
    }
    kbeg0 = kb0; kend0 = ke0;
    kbeg1 = unused1 ? kb0 : kb1;
    kend1 = unused1 ? ke0 : ke1;
    steps0 = n0;
    return n0 + n1;
  }

  // Request the tile of unified step `step` (any step >= 0: addresses are clamped, the values of
  // a step past the end are never stored).
  template <int SET>
  __device__ inline void issue(int step) {
    typedef const VT __attribute__((address_space(1)))* GlobalVec;   // global, not flat
    const bool s1 = step >= steps0;
    const int k0a = kbeg0 + step * BK, k0b = kbeg1 + (step - steps0) * BK;
    const int k0 = s1 ? k0b : k0a;
    const int kea = kend0, keb = kend1;
    const int kmax = ((s1 ? keb : kea) - 1) / VEC * VEC;
    const float* ba = base0;
    const float* bb = base1;
    const float* b = s1 ? bb : ba;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int kk = k0 + tcol(p);
      const int64_t r0 = rowoff0[p], r1 = rowoff1[p];
      const float* src = b + (s1 ? r1 : r0) + (kk < kmax ? kk : kmax);
      if constexpr (SET == 0) va[p] = *((GlobalVec)(src));
      else vb[p] = *((GlobalVec)(src));
    }
  }

  // write the tile requested by issue<SET>(step) into LDS as [mn][k]; the contraction tail of
  // the step's segment is zeroed
  template <int LDS_LD, int SET>
  __device__ inline void store(int step, float* lds) const {
    const bool s1 = step >= steps0;
    const int k0a = kbeg0 + step * BK, k0b = kbeg1 + (step - steps0) * BK;
    const int k0 = s1 ? k0b : k0a;
    const int kea = kend0, keb = kend1;
    const int ke = s1 ? keb : kea;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      VT x;
      if constexpr (SET == 0) x = va[p];
      else x = vb[p];
      float e[VEC];
      if constexpr (VEC == 4) { e[0] = x.x; e[1] = x.y; e[2] = x.z; e[3] = x.w; }
      else if constexpr (VEC == 2) { e[0] = x.x; e[1] = x.y; }
      else { e[0] = x; }
      const int c0 = k0 + tcol(p);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (c0 + j >= ke) e[j] = 0.f;
      }
      float* dst = lds + trow(p) * LDS_LD + tcol(p);
      if constexpr (VEC == 4) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        *reinterpret_cast<f4v*>(dst) = f4v{e[0], e[1], e[2], e[3]};
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = e[j];
      }
    }
  }

  // bf16x3 mode: the same tile as three bf16 planes [mn][BK] (plane p at lds + p * plane_bytes)
  template <int SET>
  __device__ inline void store_x3(int step, unsigned char* lds, int plane_bytes) const {
    static_assert(VEC == 4, "bf16x3 staging takes 16-byte vectors");
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const bool s1 = step >= steps0;
    const int k0a = kbeg0 + step * BK, k0b = kbeg1 + (step - steps0) * BK;
    const int k0 = s1 ? k0b : k0a;
    const int kea = kend0, keb = kend1;
    const int ke = s1 ? keb : kea;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      VT x;
      if constexpr (SET == 0) x = va[p];
      else x = vb[p];
      float e[4] = {x.x, x.y, x.z, x.w};
      const int c0 = k0 + tcol(p);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c0 + j >= ke) e[j] = 0.f;
      }
      unsigned h0, m0, l0, h1, m1, l1;
      split_bf16x3(e[0], e[1], h0, m0, l0);
      split_bf16x3(e[2], e[3], h1, m1, l1);
      // (tile rows start at a multiple of 16 in the plane: MN and the operand's row offset are)
      unsigned char* dst = lds + trow(p) * kX3RowBytes + x3_slot(trow(p), tcol(p) >> 3) + ((tcol(p) >> 2) & 1) * 8;
      *reinterpret_cast<u2v*>(dst) = u2v{h0, h1};
      *reinterpret_cast<u2v*>(dst + plane_bytes) = u2v{m0, m1};
      *reinterpret_cast<u2v*>(dst + 2 * plane_bytes) = u2v{l0, l1};
    }
  }
};

// 4 waves as 2 (M) x 2 (N); each wave owns (BM/2) x (BN/2) = TM x TN tiles of 32x32.
template <bool A_KC, bool B_KC, int BM, int BN, int VEC>
__global__ void __launch_bounds__(kThreads, 2) gemm_kernel(GemmArgs g) {
  constexpr int TM = BM / 64, TN = BN / 64;
  // n-contiguous B with an even number of column tiles per wave: the tiles are interleaved in
  // pairs so that one 8-byte LDS read serves two MFMAs (see the K-loop); ncol() is the map
  constexpr bool kPairN = !B_KC && (TN % 2 == 0);
  // row stride and size of one LDS buffer per operand: [BM][BK+4] (k contiguous) or [BK][BM+4]
  constexpr int LDA = A_KC ? BK + 4 : BM + 4;
  constexpr int LDB = B_KC ? BK + 4 : BN + 4;
  constexpr int ASZ = A_KC ? BM * LDA : BK * LDA;
  constexpr int BSZ = B_KC ? BN * LDB : BK * LDB;
  __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];
  float* const As = lds;
  float* const Bs = lds + 2 * ASZ;

  int64_t m_lim = g.m;
  if (g.m_dev != nullptr) {
    const int64_t v = *g.m_dev;
    if (v < m_lim) m_lim = v;
  }
  // forward (both operands k-contiguous): both K-segments behind one pipeline, optional stream-K
  constexpr bool kDual = A_KC && B_KC;
  const unsigned bx = blockIdx.x, by_ = blockIdx.y, bz = blockIdx.z;
  int64_t m0 = static_cast<int64_t>(bx) * BM;
  int prob = 0, by = by_;
  if (g.batched && by >= g.ny0) {
    prob = 1;
    by -= g.ny0;
  }
  int64_t n0 = static_cast<int64_t>(by) * BN;
  // stream-K state: this workgroup's range [sk_it, sk_end) of global K-steps, sk_S steps per tile
  int64_t sk_it = 0, sk_end = 0;
  int sk_S = 1, sk_S0 = 0, sk_tiles_n = 1, sk_u0 = 0, sk_u1 = 0;
  bool sk_partial = false;
  if constexpr (kDual) {
    if (g.streamk) {
      sk_S0 = (g.seg[0].k + BK - 1) / BK;
      sk_S = sk_S0 + (g.nseg > 1 ? (g.seg[1].k + BK - 1) / BK : 0);
      sk_tiles_n = (g.n + BN - 1) / BN;
      const int64_t total = (m_lim + BM - 1) / BM * sk_tiles_n * sk_S;
      // consecutive ranges (the two column halves of a row block, read from the same A rows)
      // go to workgroups of the SAME XCD, which share an L2: workgroup ids are dealt to the 8
      // XCDs round-robin
      unsigned lb = blockIdx.x;
      if (gridDim.x % 8 == 0) lb = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
      sk_it = total * lb / gridDim.x;
      sk_end = total * (lb + 1) / gridDim.x;
      if (sk_it >= sk_end) return;
    } else if (m0 >= m_lim) {
      return;
    }
  } else {
    if (m0 >= m_lim) return;
  }
  const int64_t n_lim = prob ? g.n2 : g.n;
  float* const out_p = prob ? g.out2 : g.out;
  const int64_t ld_out_p = prob ? g.ld_out2 : g.ld_out;
  const int s_begin = g.batched ? prob : 0;
  const int s_end = g.batched ? prob + 1 : g.nseg;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * (BM / 2);
  const int wn0 = (wave & 1) * (BN / 2);
  const int l31 = lane & 31;
  const int half = lane >> 5;
  auto ncol = [&](int j) -> int {   // tile-local column of accumulator tile j held by this lane
    return wn0 + j * 32 + l31;      // (after the epilogue's un-interleave when kPairN)
  };

  floatx16 acc[TM][TN];
  float csum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) csum[i] = 0.f;

  std::conditional_t<kDual, DualLoader<BM, VEC>, TileLoader<A_KC, BM, VEC>> la;
  std::conditional_t<kDual, DualLoader<BN, VEC>, TileLoader<B_KC, BN, VEC>> lb;

  typedef float f4v __attribute__((ext_vector_type(4)));
  constexpr int NG = BK / 8;
  // PAR = parity of the (unified) step = register set that holds step + 2: the step computes
  // from (as, bs), stores set PAR^1 (step + 1) into (as_next, bs_next) and requests step + 3
  // into set PAR^1.
  auto k_step = [&](auto par, int step, const float* as, const float* bs, float* as_next, float* bs_next) {
    constexpr int PAR = decltype(par)::value;
    float af[NG][TM][4], bf[NG][TN][4];
    auto read_group = [&](int g) {
      const int g8 = 8 * g;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (A_KC) {
          const f4v v = *reinterpret_cast<const f4v*>(as + (wm0 + i * 32 + l31) * LDA + g8 + 4 * half);
          af[g][i][0] = v[0]; af[g][i][1] = v[1]; af[g][i][2] = v[2]; af[g][i][3] = v[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) af[g][i][j] = as[(g8 + 4 * half + j) * LDA + wm0 + i * 32 + l31];
        }
      }
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        if constexpr (B_KC) {
          const f4v v = *reinterpret_cast<const f4v*>(bs + (wn0 + jn * 32 + l31) * LDB + g8 + 4 * half);
          bf[g][jn][0] = v[0]; bf[g][jn][1] = v[1]; bf[g][jn][2] = v[2]; bf[g][jn][3] = v[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[g][jn][j] = bs[(g8 + 4 * half + j) * LDB + wn0 + jn * 32 + l31];
        }
      }
    };
    read_group(0);
    read_group(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 2 < NG) read_group(g + 2);
      // k slot of MFMA j in this group: 8g + 4*half + j, for both operands
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          csum[i] += af[g][i][j];
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) {
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i][j], bf[g][jn][j], acc[i][jn], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (g == 1) {
        la.template store<LDA, PAR ^ 1>(step + 1, as_next);
        lb.template store<LDB, PAR ^ 1>(step + 1, bs_next);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (g == 2) {
        la.template issue<PAR ^ 1>(step + 3);
        lb.template issue<PAR ^ 1>(step + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  };

  for (;;) {   // one pass per tile (piece); a single pass unless stream-K
  if constexpr (kDual) {
    if (g.streamk) {
      const int64_t tile = sk_it / sk_S;
      sk_u0 = static_cast<int>(sk_it - tile * sk_S);
      const int64_t left = sk_end - sk_it;
      sk_u1 = left < sk_S - sk_u0 ? sk_u0 + static_cast<int>(left) : sk_S;
      sk_partial = !(sk_u0 == 0 && sk_u1 == sk_S);
      m0 = tile / sk_tiles_n * BM;
      n0 = tile % sk_tiles_n * BN;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if constexpr (kDual) {
    // contraction range of each segment for this workgroup (split-K: slice blockIdx.z of both).
    // (g.seg[] is only indexed with constants: a dynamic index makes the compiler copy the
    // kernel arguments to scratch memory, and scratch traffic counts on vmcnt)
    auto seg_range = [&](int k_lim, int& kb_out, int& ke_out) {
      int k_beg = 0, k_end = k_lim;
      if (g.k_chunk > 0) {
        k_beg = bz * g.k_chunk;
        k_end = k_beg + g.k_chunk < k_lim ? k_beg + g.k_chunk : k_lim;
      }
      kb_out = k_beg < k_end ? k_beg : 0;
      ke_out = k_beg < k_end ? k_end : 0;
    };
    int kb0, ke0, kb1 = 0, ke1 = 0;
    int k_lim0 = g.seg[0].k;
    if (g.k_dev != nullptr) {
      const int64_t v = *g.k_dev;
      if (v < k_lim0) k_lim0 = static_cast<int>(v);
    }
    seg_range(k_lim0, kb0, ke0);
    Operand a1 = g.seg[0].a, b1 = g.seg[0].b;
    if (g.nseg > 1) {
      seg_range(g.seg[1].k, kb1, ke1);
      a1 = g.seg[1].a;
      b1 = g.seg[1].b;
    }
    if (g.streamk) {
      // unified steps [sk_u0, sk_u1) of this tile: the first sk_S0 lie in segment 0
      const int a_lo = sk_u0 < sk_S0 ? sk_u0 : sk_S0, a_hi = sk_u1 < sk_S0 ? sk_u1 : sk_S0;
      const int b_lo = (sk_u0 > sk_S0 ? sk_u0 : sk_S0) - sk_S0, b_hi = (sk_u1 > sk_S0 ? sk_u1 : sk_S0) - sk_S0;
      const int k_lim1 = g.nseg > 1 ? g.seg[1].k : 0;
      kb0 = a_hi > a_lo ? a_lo * BK : 0;
      ke0 = a_hi > a_lo ? (a_hi * BK < k_lim0 ? a_hi * BK : k_lim0) : 0;
      kb1 = b_hi > b_lo ? b_lo * BK : 0;
      ke1 = b_hi > b_lo ? (b_hi * BK < k_lim1 ? b_hi * BK : k_lim1) : 0;
    }
    la.begin(g.seg[0].a, a1, m0, m_lim);
    lb.begin(g.seg[0].b, b1, n0, n_lim);
    const int nsteps = la.range(kb0, ke0, kb1, ke1);
    lb.range(kb0, ke0, kb1, ke1);
    if (nsteps > 0) {
      // (the scheduling barriers keep the program order of the loads: vmcnt counts in issue
      // order, and the loop's waits are derived from the order seen here)
      la.template issue<0>(0);
      lb.template issue<0>(0);
      __builtin_amdgcn_sched_barrier(0);
      la.template issue<1>(1);   // requests past the last step are clamped and never stored
      lb.template issue<1>(1);
      __builtin_amdgcn_sched_barrier(0);
      la.template store<LDA, 0>(0, As);
      lb.template store<LDB, 0>(0, Bs);
      __builtin_amdgcn_sched_barrier(0);
      la.template issue<0>(2);
      lb.template issue<0>(2);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();

      // One K-step = 4 groups of 8 k.  The fragments of group g+2 are read from LDS while group
      // g multiplies.  Global loads run TWO steps ahead in two register sets (a 128-byte piece
      // of a gathered feature row is a fresh HBM access every step, and one step of MFMAs does
      // not cover its latency): step s stores the tile of step s+1 (requested two steps ago,
      // set (s+1)&1) into the other LDS buffer behind group 1 and requests step s+3 into the
      // same set behind group 2.  The loop is unrolled by two so the set index is static.
      // Pairs of steps in a branch-free loop body (any branch inside it gives the structurised
      // CFG a path on which the younger load set is the one being stored, and every vmcnt in the
      // loop degrades to a full drain); an odd last step runs from a copy after the loop.  The
      // step that would be stored past the last one is a clamped load: its LDS image is written
      // but never read.
      int st = 0;
      for (; st + 1 < nsteps; st += 2) {
        k_step(std::integral_constant<int, 0>{}, st, As, Bs, As + ASZ, Bs + BSZ);
        k_step(std::integral_constant<int, 1>{}, st + 1, As + ASZ, Bs + BSZ, As, Bs);
      }
      if (nsteps & 1) k_step(std::integral_constant<int, 0>{}, st, As, Bs, As + ASZ, Bs + BSZ);
    }
  } else {
  for (int s = s_begin; s < s_end; ++s) {
    const Segment& sg = g.seg[s];
    int k_lim = sg.k;
    if ((s == 0 || g.batched) && g.k_dev != nullptr) {
      const int64_t v = *g.k_dev;
      if (v < k_lim) k_lim = static_cast<int>(v);
    }
    int k_beg = 0, k_end = k_lim;
    if (g.k_chunk > 0) {
      const int chunk = split_chunk(g.k_chunk, g.k_dev != nullptr, k_lim);
      k_beg = bz * chunk;
      k_end = k_beg + chunk < k_lim ? k_beg + chunk : k_lim;
    }
    if (k_beg >= k_end) continue;

    la.begin(sg.a, m0, m_lim, k_beg, k_end);
    lb.begin(sg.b, n0, n_lim, k_beg, k_end);
    {
      // d-input / d-weight: one register set, request step k+1 before the MFMAs of step k and
      // store it behind them (the deeper pipeline above does not fit the register budget of
      // the k-major fragment reads)
      la.template issue<0>(k_beg);
      lb.template issue<0>(k_beg);
      la.template store<LDA, 0>(k_beg, As);
      lb.template store<LDB, 0>(k_beg, Bs);
      __syncthreads();
      int cur = 0;
      for (int k0 = k_beg; k0 < k_end; k0 += BK) {
        const bool has_next = k0 + BK < k_end;
        if (has_next) {
          la.template issue<0>(k0 + BK);
          lb.template issue<0>(k0 + BK);
        }
        const float* as = As + cur * ASZ;
        const float* bs = Bs + cur * BSZ;
#pragma unroll
        for (int g8 = 0; g8 < BK; g8 += 8) {
          float af[TM][4], bf[TN][4];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            if constexpr (A_KC) {
              const f4v v = *reinterpret_cast<const f4v*>(as + (wm0 + i * 32 + l31) * LDA + g8 + 4 * half);
              af[i][0] = v[0]; af[i][1] = v[1]; af[i][2] = v[2]; af[i][3] = v[3];
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) af[i][j] = as[(g8 + 4 * half + j) * LDA + wm0 + i * 32 + l31];
            }
          }
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) {
            if constexpr (B_KC) {
              const f4v v = *reinterpret_cast<const f4v*>(bs + (wn0 + jn * 32 + l31) * LDB + g8 + 4 * half);
              bf[jn][0] = v[0]; bf[jn][1] = v[1]; bf[jn][2] = v[2]; bf[jn][3] = v[3];
            } else if constexpr (kPairN) {
              // column tiles jn, jn+1 interleaved (n = 64*(jn/2) + 2*lane + (jn&1)): one
              // ds_read_b64 per k feeds both (142 vs 125 TF/s LDS-fed, scratch/lds_mfma.hip)
              if ((jn & 1) == 0) {
                typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const f2v v = *reinterpret_cast<const f2v*>(bs + (g8 + 4 * half + j) * LDB + wn0 + jn * 32 + 2 * l31);
                  bf[jn][j] = v[0];
                  bf[jn + 1][j] = v[1];
                }
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) bf[jn][j] = bs[(g8 + 4 * half + j) * LDB + wn0 + jn * 32 + l31];
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              csum[i] += af[i][j];
#pragma unroll
              for (int jn = 0; jn < TN; ++jn) {
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j], bf[jn][j], acc[i][jn], 0, 0, 0);
              }
            }
          }
        }
        if (k0 + 2 * BK <= k_end) {          // the next step is a full one: nothing to zero
          la.template store<LDA, 0, false>(k0 + BK, As + (cur ^ 1) * ASZ);
          lb.template store<LDB, 0, false>(k0 + BK, Bs + (cur ^ 1) * BSZ);
        } else if (has_next) {
          la.template store<LDA, 0>(k0 + BK, As + (cur ^ 1) * ASZ);
          lb.template store<LDB, 0>(k0 + BK, Bs + (cur ^ 1) * BSZ);
        }
        __syncthreads();
        cur ^= 1;
      }
    }
  }
  }  // !kDual

  if constexpr (kPairN) {
    // The K-loop computed column tiles (j, j+1) interleaved: lane c holds columns 2c and 2c+1
    // of the 64-column pair.  Give every lane columns c and 32+c instead (4 lane shuffles per
    // accumulator register), so that the stores / atomics below stay 128-byte contiguous.
    const int s0 = (lane & 32) | (l31 >> 1), s1 = s0 + 16;
    const bool odd = (l31 & 1) != 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float x0 = acc[i][j][e], x1 = acc[i][j + 1][e];
          const float a0 = __shfl(x0, s0), a1 = __shfl(x1, s0);
          const float b0 = __shfl(x0, s1), b1 = __shfl(x1, s1);
          acc[i][j][e] = odd ? a1 : a0;
          acc[i][j + 1][e] = odd ? b1 : b0;
        }
      }
    }
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3)+8*(e>>2)+4*half
  const bool relu = (g.epilogue & SAGE_EPI_RELU) != 0;
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int64_t col = n0 + ncol(j);
    bv[j] = ((g.epilogue & SAGE_EPI_BIAS) && col < n_lim) ? g.bias[col] : 0.f;
    // consume the load here so that the stores below do not each wait on vmcnt(0)
    asm volatile("" : "+v"(bv[j]));
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + ncol(j);
      const int64_t row0 = m0 + wm0 + i * 32 + 4 * half;
      float* base = out_p + row0 * ld_out_p + col;
      if (g.atomic_out || sk_partial) {  // block-uniform
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = (e & 3) + 8 * (e >> 2);
          if (col < n_lim && row0 + r < m_lim) atomicAdd(base + r * ld_out_p, acc[i][j][e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = (e & 3) + 8 * (e >> 2);
          float v = acc[i][j][e] + bv[j];
          if (relu) v = fmaxf(v, 0.f);
          if (col < n_lim && row0 + r < m_lim) {
            if (g.rmw_out) v += base[r * ld_out_p];
            base[r * ld_out_p] = v;
            if (g.zero2 != nullptr) g.zero2[(row0 + r) * g.ld_zero2 + col] = 0.f;
          }
        }
      }
    }
  }
  if (g.colsum != nullptr && by_ == 0 && (wave & 1) == 0) {
    // the two lane halves summed disjoint k slots of row (wm0 + i*32 + l31)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float tot = csum[i] + __shfl_xor(csum[i], 32);
      const int64_t row = m0 + wm0 + i * 32 + l31;
      if (half == 0 && row < m_lim) atomicAdd(g.colsum + row, tot);
    }
  }
  if constexpr (kDual) {
    if (g.streamk) {
      sk_it += sk_u1 - sk_u0;
      if (sk_it < sk_end) continue;
    }
  }
  break;
  }  // tile pieces
}

// Forward in bf16x3 mode (opt-in): the fp32 operands are split into three bf16 terms while they
// are staged into LDS and six cross products per k16 go through v_mfma_f32_32x32x16_bf16 into
// fp32 accumulators.  Same operand handling as the stream-K forward above (two K-segments, row
// maps, device-side row count, K / N tails, partial tiles added with fp32 atomics into a zeroed
// or accumulated-into output); 64 x 128 tile, one LDS stage, three resident workgroups per CU,
// loads two steps ahead.  scratch/bf16x3_gemm.hip is the stand-alone study of this kernel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// BM = 64: three resident workgroups per CU (the stream-K pieces of few-tile shapes, headline);
// BM = 128 (wave tile 64 x 64): two per CU, a third fewer operand bytes and LDS fragment reads
// per MFMA — 10 % faster where there are tiles enough to fill the chip several times over
// (scratch/bf16x3_gemm_16.hip -DBMV=128 at the pool Linear's shape: 462 vs 511 us)
template <int BM>
__global__ void __launch_bounds__(kThreads, BM == 64 ? 3 : 2) gemm_x3_kernel(GemmArgs g) {
  constexpr int BN = 128;
  constexpr int TI = BM / 32;        // 16-row blocks per wave tile
  constexpr int PLANE = (BM + BN) * kX3RowBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PLANE];
  int64_t m_lim = g.m;
  if (g.m_dev != nullptr) {
    const int64_t v = *g.m_dev;
    if (v < m_lim) m_lim = v;
  }
  int k_lim0 = g.seg[0].k;
  if (g.k_dev != nullptr) {
    const int64_t v = *g.k_dev;
    if (v < k_lim0) k_lim0 = static_cast<int>(v);
  }
  const int k_lim1 = g.nseg > 1 ? g.seg[1].k : 0;
  const int S0 = (k_lim0 + BK - 1) / BK;
  const int S = S0 + (k_lim1 + BK - 1) / BK;
  const int tiles_n = (g.n + BN - 1) / BN;
  const int64_t total = (m_lim + BM - 1) / BM * tiles_n * S;
  unsigned lb_ = blockIdx.x;
  if (gridDim.x % 8 == 0) lb_ = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
  int64_t it = total * lb_ / gridDim.x;
  int64_t end_ = total * (lb_ + 1) / gridDim.x;
  int64_t tile_skip = 0;   // K-steps skipped after every tile
  if (g.whole) {   // whole tiles only: the range boundaries are tile boundaries
    const int64_t ntiles = total / S;
    it = ntiles * lb_ / gridDim.x * S;
    end_ = ntiles * (lb_ + 1) / gridDim.x * S;
    if (g.xcd_map && gridDim.x % 8 == 0) {
      // the XCD's share of the tiles dealt round-robin to its workgroups: at any time they work
      // on ~gridDim.x / 8 CONSECUTIVE tiles, i.e. on all column tiles of a few row tiles, whose A
      // rows (random table rows through a row map: the pool Linear) then come out of L2 for all
      // but the first of them
      const unsigned xcd = blockIdx.x % 8, j = blockIdx.x / 8, per = gridDim.x / 8;
      it = (ntiles * xcd / 8 + j) * S;
      end_ = ntiles * (xcd + 1) / 8 * S;
      tile_skip = static_cast<int64_t>(per - 1) * S;
    }
  }
  const int64_t end = end_;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;     // 16x16x32 operand: row r16, k = 8 kg .. 8 kg + 7
  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * 64;
  const int offa = (wm0 + r16) * kX3RowBytes + x3_slot(r16, kg);          // + 16 i rows
  const int offb = (BM + wn0 + r16) * kX3RowBytes + x3_slot(r16, kg);     // + 16 j rows
  typedef float floatx4 __attribute__((ext_vector_type(4)));
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  auto frag = [&](const unsigned char* p) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4v*>(p));
  };
  Operand a1 = g.seg[0].a, b1 = g.seg[0].b;
  if (g.nseg > 1) {
    a1 = g.seg[1].a;
    b1 = g.seg[1].b;
  }
  DualLoader<BM, 4> la;
  DualLoader<BN, 4> lb;
  using S0_ = std::integral_constant<int, 0>;
  using S1_ = std::integral_constant<int, 1>;

  while (it < end) {   // block-uniform
    const int64_t tile = it / S;
    const int u0 = static_cast<int>(it - tile * S);
    const int64_t left = end - it;
    const int u1 = left < S - u0 ? u0 + static_cast<int>(left) : S;
    it += u1 - u0 + tile_skip;
    const int64_t m0 = tile / tiles_n * BM, n0 = tile % tiles_n * BN;
    const int a_lo = u0 < S0 ? u0 : S0, a_hi = u1 < S0 ? u1 : S0;
    const int b_lo = (u0 > S0 ? u0 : S0) - S0, b_hi = (u1 > S0 ? u1 : S0) - S0;
    const int kb0 = a_hi > a_lo ? a_lo * BK : 0;
    const int ke0 = a_hi > a_lo ? (a_hi * BK < k_lim0 ? a_hi * BK : k_lim0) : 0;
    const int kb1 = b_hi > b_lo ? b_lo * BK : 0;
    const int ke1 = b_hi > b_lo ? (b_hi * BK < k_lim1 ? b_hi * BK : k_lim1) : 0;
    la.begin(g.seg[0].a, a1, m0, m_lim);
    lb.begin(g.seg[0].b, b1, n0, g.n);
    const int nsteps = la.range(kb0, ke0, kb1, ke1);
    lb.range(kb0, ke0, kb1, ke1);

    floatx4 acc[TI][4];      // the wave's (BM/2) x 64 tile as TI x 4 blocks of 16 x 16
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    // v_mfma_f32_16x16x32_bf16 (one K-step of 32 per instruction) rather than 32x32x16: the same
    // cycles per flop at lower power, and this kernel is power-limited (scratch/bf16x3_gemm_16.hip:
    // 57.9 vs 61.8 us sustained)
    auto mfmas = [&]() {
      // the A fragments stay, the B fragments of one 16-column block at a time (register budget:
      // 168 VGPRs at three workgroups per CU)
      bf16x8 a[TI][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < TI; ++i) a[i][p] = frag(lds + p * PLANE + offa + i * 16 * kX3RowBytes);
      // (a term, b term) of the six products, smallest first: lo.hi hi.lo mid.mid mid.hi hi.mid hi.hi
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8 b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = frag(lds + p * PLANE + offb + j * 16 * kX3RowBytes);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int i = 0; i < TI; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA[q]], b[PB[q]], acc[i][j], 0, 0, 0);
      }
    };
    // one step: LDS holds step t, register set HOLD holds step t+1, set FREE takes step t+2
    // (requests past the last step are clamped by the loader; their LDS image is never read)
    auto step = [&](auto hold, auto free_, int t) {
      constexpr int HOLD = decltype(hold)::value, FREE = decltype(free_)::value;
      la.template issue<FREE>(t + 2);
      lb.template issue<FREE>(t + 2);
      __builtin_amdgcn_sched_barrier(0);
      mfmas();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      la.template store_x3<HOLD>(t + 1, lds, PLANE);
      lb.template store_x3<HOLD>(t + 1, lds + BM * kX3RowBytes, PLANE);
      __syncthreads();
    };
    if (nsteps > 0) {
      la.template issue<0>(0);
      lb.template issue<0>(0);
      __builtin_amdgcn_sched_barrier(0);
      la.template issue<1>(1);
      lb.template issue<1>(1);
      __builtin_amdgcn_sched_barrier(0);
      la.template store_x3<0>(0, lds, PLANE);
      lb.template store_x3<0>(0, lds + BM * kX3RowBytes, PLANE);
      __syncthreads();
      int t = 0;
      for (; t + 1 < nsteps; t += 2) {
        step(S1_{}, S0_{}, t);
        step(S0_{}, S1_{}, t + 1);
      }
      if (t < nsteps) step(S1_{}, S0_{}, t);
    }
    // C/D of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = n0 + wn0 + j * 16 + r16;
        const int64_t row0 = m0 + wm0 + i * 16 + 4 * kg;
        float* base = g.out + row0 * g.ld_out + col;
        if (!g.whole) {   // block-uniform
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (col < g.n && row0 + e < m_lim) atomicAdd(base + e * g.ld_out, acc[i][j][e]);
          }
        } else {
          const float bv = ((g.epilogue & SAGE_EPI_BIAS) && col < g.n) ? g.bias[col] : 0.f;
          const bool relu = (g.epilogue & SAGE_EPI_RELU) != 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (col < g.n && row0 + e < m_lim) {
              float v = acc[i][j][e] + bv;
              if (relu) v = fmaxf(v, 0.f);
              if (g.rmw_out) v += base[e * g.ld_out];
              base[e * g.ld_out] = v;
              if (g.zero2 != nullptr) g.zero2[(row0 + e) * g.ld_zero2 + col] = 0.f;
            }
          }
        }
      }
  }
}

// Weight gradients in bf16x3 mode (opt-in): dw[n][k] += sum_m dout[m][n] * a[rows(m)][k], both
// operands with the contraction index as their ROW index.  The tiles are staged as they lie in
// memory — three bf16 planes [k][64 + 128 columns] — and the MFMA fragments (8 consecutive k per
// lane) come out of LDS through the transposing read ds_read_b64_tr_b16.  Same grid and
// arguments as the fp32 d-weight launch (split-K over blockIdx.z, the two problems of a layer
// over blockIdx.y, device-side contraction length, fp32 atomics into dw).
// scratch/bf16x3_dweight.hip is the stand-alone study.
// LDS row of one plane: (BM + 128) bf16 columns + 64 B.  448 B = 112 dwords = 48 (mod 64) for BM = 64,
// 576 B = 144 dwords = 16 (mod 64) for BM = 128: either way the 4 rows x 16 dwords of a half-wave's
// transposing read tile the 64 banks.  BM = 128 (64 x 64 wave tiles, two workgroups per CU): fewer
// operand bytes and fragment reads per MFMA, for weight matrices with tiles enough (the pool Linear's).
template <int BM>
__global__ void __launch_bounds__(kThreads, BM == 64 ? 3 : 2) gemm_x3_dw_kernel(GemmArgs g) {
  constexpr int BN = 128;
  constexpr int TI = BM / 32;
  constexpr int kX3DwRowBytes = (BM + BN) * 2 + 64;
  constexpr int PLANE = BK * kX3DwRowBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PLANE];
  unsigned bx = blockIdx.x, by_ = blockIdx.y, bz = blockIdx.z;
  if (g.xcd_map) {
    // XCD c runs the workgroups with linear id = c (mod 8): give it a CONTIGUOUS range of
    // (slice, column tile, row tile) triples, i.e. whole split-K slices — every tile of a slice
    // streams the same rows of both operands, which then cross HBM -> L2 once instead of once
    // per XCD that holds one of the slice's tiles
    const unsigned gx = gridDim.x, gy = gridDim.y, W = gx * gy * gridDim.z;
    const unsigned L = bx + gx * (by_ + gy * bz);
    const unsigned xcd = L % 8, qn = W / 8, rn = W % 8;
    const unsigned Lp = xcd * qn + (xcd < rn ? xcd : rn) + L / 8;
    bx = Lp % gx;
    by_ = Lp / gx % gy;
    bz = Lp / (gx * gy);
  }
  int by = static_cast<int>(by_);
  const bool prob = g.batched && by >= g.ny0;
  if (prob) by -= g.ny0;
  const int64_t m0 = static_cast<int64_t>(bx) * BM, n0 = static_cast<int64_t>(by) * BN;
  const int64_t m_lim = g.m;
  const int64_t n_lim = prob ? g.n2 : g.n;
  float* const out_p = prob ? g.out2 : g.out;
  const int64_t ld_out_p = prob ? g.ld_out2 : g.ld_out;
  // (g.seg[] indexed with constants only, field by field: see gemm_kernel)
  Operand oa, ob;
  oa.p = prob ? g.seg[1].a.p : g.seg[0].a.p;
  oa.ld = prob ? g.seg[1].a.ld : g.seg[0].a.ld;
  oa.rowmap = prob ? g.seg[1].a.rowmap : g.seg[0].a.rowmap;
  oa.vec = 4;
  ob.p = prob ? g.seg[1].b.p : g.seg[0].b.p;
  ob.ld = prob ? g.seg[1].b.ld : g.seg[0].b.ld;
  ob.rowmap = prob ? g.seg[1].b.rowmap : g.seg[0].b.rowmap;
  ob.vec = 4;
  int k_lim = prob ? g.seg[1].k : g.seg[0].k;
  if (g.k_dev != nullptr) {
    const int64_t v = *g.k_dev;
    if (v < k_lim) k_lim = static_cast<int>(v);
  }
  int k_beg = 0, k_end = k_lim;
  if (g.k_chunk > 0) {
    const int chunk = split_chunk(g.k_chunk, g.k_dev != nullptr, k_lim);
    k_beg = bz * chunk;
    k_end = k_beg + chunk < k_lim ? k_beg + chunk : k_lim;
  }
  if (m0 >= m_lim) return;                     // block-uniform: EXEC stays full for the transposing reads
  if (k_beg >= k_end) {
    if (g.part_stride > 0) {                   // an empty slice still owns its partial tile: zeros
      float* zp = out_p + static_cast<int64_t>(bz) * g.part_stride;
      for (int idx = threadIdx.x; idx < BM * BN; idx += kThreads) {
        const int64_t r = m0 + idx / BN, c = n0 + idx % BN;
        if (r < m_lim && c < n_lim) zp[r * ld_out_p + c] = 0.f;
      }
    }
    return;
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;     // 16x16x32 operand: column r16, k = 8 kg .. 8 kg + 7
  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * 64;
  // transposing read: lane 4q+p of a 16-lane group addresses row (k) q, columns 4p..4p+3 of a
  // 4 x 16 block and receives column (lane & 15) of it, row q in element q.  Group kg reads rows
  // 8 kg + q (and + 4 in a second read) of the 16 columns of block cb.
  const int q = r16 >> 2, p4 = (lane & 3) * 4;
  const int tr_row = 8 * kg + q;
  typedef float floatx4 __attribute__((ext_vector_type(4)));
  typedef short s4v __attribute__((ext_vector_type(4)));
  typedef short s8v __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s4v* LdsS4;
  auto frag = [&](const unsigned char* plane, int col) -> bf16x8 {   // col: first of the block's 16 columns
    const unsigned char* base = plane + x3_dw_offset(tr_row, col + p4, kX3DwRowBytes);   // (tr_row + 4) >> 3 == tr_row >> 3
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsS4)(base));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsS4)(base + 4 * kX3DwRowBytes));
    const s8v w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, w);
  };

  TileLoader<false, BM, 4> la;
  TileLoader<false, BN, 4> lb;
  la.begin(oa, m0, m_lim, k_beg, k_end);
  lb.begin(ob, n0, n_lim, k_beg, k_end);
  floatx4 acc[TI][4];      // the wave's (BM/2) x 64 tile as TI x 4 blocks of 16 x 16
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // v_mfma_f32_16x16x32_bf16, one K-step of 32 per instruction (see gemm_x3_kernel); the A
  // fragments stay, the B fragments of one 16-column block at a time
  auto mfmas = [&]() {
    bf16x8 a[TI][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < TI; ++i) a[i][p] = frag(lds + p * PLANE, wm0 + 16 * i);
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = frag(lds + p * PLANE, BM + wn0 + 16 * j);
#pragma unroll
      for (int qq = 0; qq < 6; ++qq)
#pragma unroll
        for (int i = 0; i < TI; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA[qq]], b[PB[qq]], acc[i][j], 0, 0, 0);
    }
  };
  // bias gradient as a by-product (colsum[n] += sum_m dout[m][n], the column sums of the A operand):
  // the workgroups of the first column tile add the rows they stage anyway, in registers; one
  // atomic per column and split-K slice at the end (adds to one address serialise at ~65 ns each:
  // a separate pass over dout with hundreds of row chunks spends its time there)
  const bool do_cs = g.colsum != nullptr && by == 0 && !prob;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  using S0_ = std::integral_constant<int, 0>;
  using S1_ = std::integral_constant<int, 1>;
  // one step: LDS holds the tile at k0, set HOLD the one at k0 + BK, set FREE takes k0 + 2 BK
  // (requests past k_end are clamped by the loader; their LDS image is zeros and never read)
  auto step = [&](auto hold, auto free_, int k0) {
    constexpr int HOLD = decltype(hold)::value, FREE = decltype(free_)::value;
    la.template issue<FREE>(k0 + 2 * BK);
    lb.template issue<FREE>(k0 + 2 * BK);
    __builtin_amdgcn_sched_barrier(0);
    mfmas();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    la.template store_x3<HOLD>(k0 + BK, lds, 0, kX3DwRowBytes, PLANE);
    lb.template store_x3<HOLD>(k0 + BK, lds, BM, kX3DwRowBytes, PLANE);
    if (do_cs) la.template colsum_add<HOLD>(k0 + BK, cs);     // block-uniform; registers only
    __syncthreads();
  };
  la.template issue<0>(k_beg);
  lb.template issue<0>(k_beg);
  __builtin_amdgcn_sched_barrier(0);
  la.template issue<1>(k_beg + BK);
  lb.template issue<1>(k_beg + BK);
  __builtin_amdgcn_sched_barrier(0);
  la.template store_x3<0>(k_beg, lds, 0, kX3DwRowBytes, PLANE);
  lb.template store_x3<0>(k_beg, lds, BM, kX3DwRowBytes, PLANE);
  if (do_cs) la.template colsum_add<0>(k_beg, cs);
  __syncthreads();
  int k0 = k_beg;
  for (; k0 + BK < k_end; k0 += 2 * BK) {
    step(S1_{}, S0_{}, k0);
    step(S0_{}, S1_{}, k0 + BK);
  }
  if (k0 < k_end) step(S1_{}, S0_{}, k0);
  if (do_cs) {   // (the last step ended with a barrier and nothing reads its LDS image)
    constexpr int UPR = BM / 4;                 // threads that share a row of the A tile = distinct column groups
    float* part = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < 4; ++j) part[threadIdx.x * 4 + j] = cs[j];
    __syncthreads();
    if (threadIdx.x < UPR) {
      float tot[4] = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < kThreads / UPR; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[j] += part[(r * UPR + threadIdx.x) * 4 + j];
      const int64_t c0 = m0 + threadIdx.x * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < m_lim) atomicAdd(g.colsum + c0 + j, tot[j]);
    }
  }
  // C/D of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = n0 + wn0 + j * 16 + r16;
      const int64_t row0 = m0 + wm0 + i * 16 + 4 * kg;
      float* base = out_p + row0 * ld_out_p + col;
      if (g.part_stride > 0) {      // (uniform) this slice's own copy of the tile: plain stores
        base += static_cast<int64_t>(bz) * g.part_stride;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (col < n_lim && row0 + e < m_lim) base[e * ld_out_p] = acc[i][j][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (col < n_lim && row0 + e < m_lim) atomicAdd(base + e * ld_out_p, acc[i][j][e]);
        }
      }
    }
}

// bf16x3 mode is the default for the large contractions; SAGE_GEMM_X3=0 in the environment
// (read at every call) keeps them on the exact-fp32 MFMA
bool xcd_mode() {
  const char* v = getenv("SAGE_GEMM_XCD");
  return v == nullptr || v[0] != '0';
}

// Split-K factor of a weight-gradient launch with `tiles` output tiles on `slots` resident
// workgroups.  Few tiles: as many slices as fill one round.  MORE tiles than slots (the lstm
// cell's W_ih gradient: 590 and 1,180 tiles of 128 x 128 on 512 slots — 2 and 3 rounds of which
// the last is 15 / 30 % full, SQ_BUSY_CU_CYCLES 70 % of the launch): the factor in 1..8 whose
// slices x tiles fill whole rounds best (590 x 6 = 3,540 on 7 x 512 = 3,584), if that wins at
// least 5 % over no split — the partial sums are fp32 atomics into dw either way.
int64_t balanced_splits(int64_t tiles, int64_t slots, int64_t max_splits) {
  int64_t sp = slots / tiles;
  if (tiles > slots) {
    auto fill = [&](int64_t f) { return static_cast<double>(tiles * f) / static_cast<double>(cdiv(tiles * f, slots) * slots); };
    sp = 1;
    double best = fill(1) * 1.05;
    for (int64_t f = 2; f <= 8; ++f) {
      if (fill(f) > best) {
        best = fill(f);
        sp = f;
      }
    }
  }
  if (sp > max_splits) sp = max_splits;
  return sp < 1 ? 1 : sp;
}

// K-steps per split-K slice of a weight-gradient launch, at least: binds for the 512 seed rows of a
// last layer, whose few output tiles then get 8 slices of 2 steps instead of 4 of 4 (PPI 0.1221 ->
// 0.1197 ms per step, the pooling and three-forward steps unchanged; SAGE_DW_MIN_STEPS=4: the
// earlier rule, the A/B switch)
int dw_min_steps() {
  static const char* v = getenv("SAGE_DW_MIN_STEPS");
  const int s = v ? atoi(v) : 2;
  return s < 1 ? 1 : s;
}

bool x3_mode() {
  const char* v = getenv("SAGE_GEMM_X3");
  return v == nullptr || v[0] != '0';
}

// CUs of the current device (0 if the query fails: callers then keep whole-tile grids)
int num_cus() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cached[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cached[dev] = v;
  }
  return cached[dev];
}

bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Widest vector (floats) for loading `width` contiguous floats per row of stride ld,
// starting at multiples of BK (row-contiguous dim = contraction) or of 64 (m/n).
int op_vec(const float* p, int64_t ld, int64_t width) {
  if (aligned(p, 16) && ld % 4 == 0 && ld >= (width + 3) / 4 * 4) return 4;
  if (aligned(p, 8) && ld % 2 == 0 && ld >= (width + 1) / 2 * 2) return 2;
  return 1;
}

template <bool A_KC, bool B_KC, int VEC>
int launch_gemm_vec(const GemmArgs& g, int splits, hipStream_t stream, const char* what) {
  if (g.batched) {  // 64x128 tiles; y covers problem 0 then problem 1
    dim3 grid(static_cast<unsigned>(cdiv(g.m, 64)),
              static_cast<unsigned>(g.ny0 + cdiv(g.n2, 128)), static_cast<unsigned>(splits));
    SAGE_LAUNCH_PROBED((gemm_kernel<A_KC, B_KC, 64, 128, VEC>), grid, kThreads, stream, g);
    SAGE_CHECK_LAUNCH(what);
    return SAGE_OK;
  }
  if (g.n > 64) {
    dim3 grid(static_cast<unsigned>(cdiv(g.m, 64)), static_cast<unsigned>(cdiv(g.n, 128)),
              static_cast<unsigned>(splits));
    SAGE_LAUNCH_PROBED((gemm_kernel<A_KC, B_KC, 64, 128, VEC>), grid, kThreads, stream, g);
  } else {
    dim3 grid(static_cast<unsigned>(cdiv(g.m, 64)), static_cast<unsigned>(cdiv(g.n, 64)),
              static_cast<unsigned>(splits));
    SAGE_LAUNCH_PROBED((gemm_kernel<A_KC, B_KC, 64, 64, VEC>), grid, kThreads, stream, g);
  }
  SAGE_CHECK_LAUNCH(what);
  return SAGE_OK;
}

template <bool A_KC, bool B_KC>
int launch_gemm(const GemmArgs& g, int splits, hipStream_t stream, const char* what) {
  if (g.m <= 0 || g.n <= 0) return SAGE_OK;
  // one vector width for the whole kernel: the narrowest any operand allows
  int vec = 4;
  for (int s = 0; s < g.nseg; ++s) {
    if (g.seg[s].a.vec < vec) vec = g.seg[s].a.vec;
    if (g.seg[s].b.vec < vec) vec = g.seg[s].b.vec;
  }
  if (vec == 4) return launch_gemm_vec<A_KC, B_KC, 4>(g, splits, stream, what);
  if (vec == 2) return launch_gemm_vec<A_KC, B_KC, 2>(g, splits, stream, what);
  return launch_gemm_vec<A_KC, B_KC, 1>(g, splits, stream, what);
}


// ======================================================================================
// Skinny contractions: one of the dimensions is <= 64 (the class logits: 41 on Reddit, 7 on
// Cora).  An MFMA tile grid has only M/64 workgroups there and every K-step is a latency-bound
// round trip, so these run as plain VALU kernels with one wave per row instead: the whole
// weight matrix (<= 64 rows) stays in L1/L2 and every lane streams 16-byte vectors.
// ======================================================================================
__global__ void zero_rows_kernel(float* out, int64_t ld, int64_t m, int n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < m * n) out[(i / n) * ld + (i % n)] = 0.f;
}

__global__ void zero_rows2_kernel(float* out1, int64_t ld1, float* out2, int64_t ld2, int64_t m,
                                  const int64_t* m_dev, int n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (m_dev != nullptr) {      // rows past the device-side count stay untouched, as in the whole-K launches
    const int64_t v = *m_dev;
    if (v < m) m = v;
  }
  if (i < m * n) {
    float* const o = blockIdx.y ? out2 : out1;
    o[(i / n) * (blockIdx.y ? ld2 : ld1) + (i % n)] = 0.f;
  }
}

struct SkinnyArgs {
  Segment seg[2];
  int nseg;
  int64_t m;
  const int64_t* m_dev;
  int n;
  float* out;
  int64_t ld_out;
  const float* bias;
  int epilogue;
  int accumulate;
  float* out2;       // d-input pair: blockIdx.y == 1 multiplies by seg[1].b into out2
  int64_t ld_out2;
};

__device__ inline float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ inline float4 load4_guard(const float* p, int k0, int k_lim) {
  float4 v = *reinterpret_cast<const float4*>(p + k0);  // stays inside the padded row stride
  if (k0 + 1 >= k_lim) v.y = 0.f;
  if (k0 + 2 >= k_lim) v.z = 0.f;
  if (k0 + 3 >= k_lim) v.w = 0.f;
  return v;
}

// da[m][k] (+)= sum_n dout[m][n] * w[n][k]  (n <= 64): one wave per row m, lanes over k
__global__ void __launch_bounds__(kThreads) skinny_dinput_kernel(SkinnyArgs s) {
  int64_t m_lim = s.m;
  if (s.m_dev != nullptr) {
    const int64_t v = *s.m_dev;
    if (v < m_lim) m_lim = v;
  }
  const int lane = threadIdx.x & 63;
  const int64_t m = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + (threadIdx.x >> 6);
  if (m >= m_lim) return;
  const Segment& g = s.seg[blockIdx.y];  // a = dout [m][n], b = w [n][k]; g.k = n (contraction)
  float* const outp = blockIdx.y ? s.out2 : s.out;
  const int64_t ld_outp = blockIdx.y ? s.ld_out2 : s.ld_out;
  const float* drow = g.a.p + m * g.a.ld;
  const int K = s.n;                 // output columns
  for (int k0 = lane * 4; k0 < K; k0 += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = 0; n < g.k; ++n) {
      const float d = drow[n];
      const float4 wv = *reinterpret_cast<const float4*>(g.b.p + n * g.b.ld + k0);
      acc.x += d * wv.x; acc.y += d * wv.y; acc.z += d * wv.z; acc.w += d * wv.w;
    }
    float* o = outp + m * ld_outp + k0;
    const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (k0 + j < K) o[j] = s.accumulate ? o[j] + v[j] : v[j];
    }
  }
}

// skinny_dweight_body (skinny_dw.h): dw[n][k] += sum_m dout[m][n] * a[rows(m)][k]  (n <= 64).
// blockIdx.z = problem * slices + slice of m: the two weight gradients of a layer (same dout)
// run as one grid
__global__ void __launch_bounds__(kThreads)
skinny_dweight_kernel(const float* __restrict__ dout, int64_t ld_dout, SkinnyDwProblem p0,
                      SkinnyDwProblem p1, int slices, int64_t m,
                      const int64_t* __restrict__ m_dev, int n, int m_chunk,
                      float* __restrict__ dbias) {
  skinny_dweight_body(blockIdx.x, blockIdx.y, blockIdx.z, dout, ld_dout, p0, p1, slices, m, m_dev, n, m_chunk, dbias);
}

}  // namespace
}  // namespace sage

using namespace sage;

static int linear_concat_fwd_impl(const float* a1, sage_i64 ld_a1, const sage_i64* rows1, int k1,
                                  const float* w1, sage_i64 ld_w1, const float* a2, sage_i64 ld_a2,
                                  const sage_i64* rows2, int k2, const float* w2, sage_i64 ld_w2,
                                  const float* bias, int epilogue, sage_i64 m, const sage_i64* m_dev,
                                  int n, float* out, sage_i64 ld_out, float* zero2, sage_i64 ld_zero2,
                                  void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // (zero2 rides on the plain stores of a whole-tile epilogue: bias / relu launches never split K)
  SAGE_REQUIRE(zero2 == nullptr || ((epilogue & (SAGE_EPI_BIAS | SAGE_EPI_RELU)) && ld_zero2 >= n),
               SAGE_EINVAL, "sage_pool_linear_fwd: bad companion buffer");
  SAGE_REQUIRE(m >= 0 && n > 0 && k1 > 0 && k2 >= 0 && ld_out >= n, SAGE_EINVAL,
               "sage_linear_concat_fwd: bad shape (m=%lld n=%d k1=%d k2=%d)", (long long)m, n, k1, k2);
  SAGE_REQUIRE(ld_a1 >= k1 && ld_w1 >= k1 && (k2 == 0 || (ld_a2 >= k2 && ld_w2 >= k2)), SAGE_EINVAL,
               "sage_linear_concat_fwd: row stride smaller than k");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(a1 && w1 && out && (k2 == 0 || (a2 && w2)), SAGE_EINVAL,
               "sage_linear_concat_fwd: NULL pointer argument");
  SAGE_REQUIRE(!(epilogue & SAGE_EPI_BIAS) || bias != nullptr, SAGE_EINVAL,
               "sage_linear_concat_fwd: SAGE_EPI_BIAS without bias");
  SAGE_REQUIRE(!(epilogue & SAGE_EPI_ACCUM) || (epilogue & ~SAGE_EPI_ZEROED) == SAGE_EPI_ACCUM, SAGE_EINVAL,
               "sage_linear_concat_fwd: SAGE_EPI_ACCUM excludes bias / relu");
  SAGE_REQUIRE(!(epilogue & SAGE_EPI_ZEROED) || (epilogue & SAGE_EPI_ACCUM), SAGE_EINVAL,
               "sage_linear_concat_fwd: SAGE_EPI_ZEROED qualifies SAGE_EPI_ACCUM");
  // the addend is all zeros (the caller's promise): paths whose elements have one owner store
  // instead of load-add-store, the others add into it without a fill pass as before
  const bool zeroed = (epilogue & SAGE_EPI_ZEROED) != 0;
  epilogue &= ~SAGE_EPI_ZEROED;
  GemmArgs g = {};
  g.seg[0].a = {a1, ld_a1, rows1, op_vec(a1, ld_a1, k1)};
  g.seg[0].b = {w1, ld_w1, nullptr, op_vec(w1, ld_w1, k1)};
  g.seg[0].k = k1;
  g.nseg = 1;
  if (k2 > 0) {
    g.seg[1].a = {a2, ld_a2, rows2, op_vec(a2, ld_a2, k2)};
    g.seg[1].b = {w2, ld_w2, nullptr, op_vec(w2, ld_w2, k2)};
    g.seg[1].k = k2;
    g.nseg = 2;
  }
  g.m = m;
  g.m_dev = m_dev;
  g.n = n;
  g.out = out;
  g.ld_out = ld_out;
  g.bias = bias;
  g.epilogue = epilogue;
  g.zero2 = zero2;
  g.ld_zero2 = ld_zero2;
  // Few output tiles (the class-logit layer: M = 512, N = 41): split the contraction over
  // blockIdx.z so that more than M/64 workgroups share the latency-bound K-steps; the partial
  // sums are combined with fp32 atomics into a zeroed output.
  const int64_t tiles = cdiv(m, 64) * cdiv(n, n > 64 ? 128 : 64);
  const int kmax = k1 > k2 ? k1 : k2;
  const bool accum = (epilogue & SAGE_EPI_ACCUM) != 0;
  if ((epilogue & ~SAGE_EPI_ACCUM) == SAGE_EPI_NONE && tiles <= 32 && kmax >= 4 * BK) {
    int splits = kmax / (2 * BK);
    if (splits > 8) splits = 8;
    g.k_chunk = static_cast<int>(cdiv(cdiv(kmax, splits), BK) * BK);
    splits = static_cast<int>(cdiv(kmax, g.k_chunk));
    g.atomic_out = 1;
    if (!accum) zero_rows_kernel<<<static_cast<unsigned>(cdiv(m * n, kThreads)), kThreads, 0, stream>>>(out, ld_out, m, n);
    SAGE_CHECK_LAUNCH("zero_rows_kernel");
    return launch_gemm<true, true>(g, splits, stream, "gemm_kernel(fwd, split-K)");
  }
  // Whole tiles that do not fill the last round of workgroup slots (two resident workgroups per
  // CU) leave CUs idle while others still hold two tiles — layer 0 on the Reddit shape: 410 tiles
  // on 512 slots, critical path 2 tile-times for 1.6 of work.  Stream-K: one workgroup per slot,
  // every one takes the same number of K-steps, partial tiles are added with fp32 atomics into a
  // zeroed output.
  const int slots = 2 * num_cus();
  // opt-in bf16x3 mode (see gemm_x3_kernel): fp32-accurate, not the bit pattern of an fp32 FMA
  // chain; large contractions with 16-byte aligned operands only, everything else as before
  {
    bool vec4 = true;
    for (int sgi = 0; sgi < g.nseg; ++sgi) vec4 = vec4 && g.seg[sgi].a.vec == 4 && g.seg[sgi].b.vec == 4;
    const int64_t steps = tiles * (cdiv(k1, BK) + cdiv(k2, BK));
    const bool plain = (epilogue & ~SAGE_EPI_ACCUM) == SAGE_EPI_NONE;
    const int64_t grid_x3 = slots / 2 * 3;      // three resident workgroups per CU
    // many tiles per workgroup: whole tiles (plain stores, bias / relu epilogue, out += by
    // load-add-store), the uneven last round costs little; few: stream-K pieces with atomics
    // (a bias / relu epilogue needs whole tiles whatever their number: one owner per element;
    // short contractions — F = 100: 8 K-steps per tile — gain nothing from sharing K-steps out:
    // 4 steps and an atomic epilogue per workgroup took 36 us for 1.3 GFLOP)
    // 128-row tiles (two workgroups per CU) pay whenever they come as whole tiles in well-filled
    // rounds: two rounds and more, or a last round at least 90 % full (the lstm gates: 485 and
    // 970 tiles on 512 slots; the headline's 208 are not, they stay stream-K pieces of 64 x 128)
    const char* bm_env = getenv("SAGE_GEMM_BM");      // "64": keep the 64-row tile (A/B switch)
    const int64_t t128 = cdiv(m, 128) * cdiv(n, 128);
    // (a capacity that overshoots a whole number of rounds by at most 3 % counts as filling it:
    // the row count of a sampled batch is below its capacity whenever two seeds share a
    // neighbour — the lstm cell's i-gate block, 520 tiles for the capacity of 13,312 rows, has
    // 485-495 for the 12.4-12.6 k rows a Reddit-shaped batch really holds)
    const bool fill128 = slots > 0 && !(bm_env && bm_env[0] == '6') &&
                         (t128 >= 2ll * slots || t128 * 10 >= cdiv(t128, slots) * slots * 9 ||
                          (t128 > slots && t128 % slots * 100 <= 3 * slots && m_dev != nullptr));
    // SAGE_FWD_WHOLE=64 / 128: whole tiles of that many rows whatever the rule says (the A/B switch
    // of DESIGN.md 8.0: the headline forward without stream-K's atomic epilogue)
    static const char* whole_env = getenv("SAGE_FWD_WHOLE");
    const bool force128 = whole_env && whole_env[0] == '1';
    const bool whole = tiles >= 8 * grid_x3 || !plain || cdiv(k1, BK) + cdiv(k2, BK) <= 12 || fill128 || whole_env;
    if (x3_mode() && (plain || whole) && n > 64 && slots > 0 && vec4 && steps >= 6ll * slots) {
      if (whole) {
        g.whole = 1;
        g.rmw_out = (accum && !zeroed) ? 1 : 0;
      } else if (!accum) {
        zero_rows_kernel<<<static_cast<unsigned>(cdiv(m * n, kThreads)), kThreads, 0, stream>>>(out, ld_out, m, n);
        SAGE_CHECK_LAUNCH("zero_rows_kernel");
      }
      g.xcd_map = xcd_mode() ? 1 : 0;
      if (whole && (whole_env ? force128 : fill128)) {
        SAGE_LAUNCH_PROBED((gemm_x3_kernel<128>), static_cast<unsigned>(slots), kThreads, stream, g);
      } else {
        SAGE_LAUNCH_PROBED((gemm_x3_kernel<64>), static_cast<unsigned>(grid_x3), kThreads, stream, g);
      }
      SAGE_CHECK_LAUNCH("gemm_x3_kernel");
      return SAGE_OK;
    }
  }
  // (short contractions — PPI's 50 + 50 features: 4 K-steps per tile — stay whole tiles, as in the
  // bf16x3 mode above: a piece of 3 steps with an atomic epilogue is no faster than a tile of 4 with
  // plain stores; SAGE_FWD_STREAMK_SHORT=1: stream-K for them too, the A/B switch)
  static const char* sk_short_env = getenv("SAGE_FWD_STREAMK_SHORT");
  const bool short_k = cdiv(k1, BK) + cdiv(k2, BK) <= 12 && !(sk_short_env && sk_short_env[0] == '1');
  if ((epilogue & ~SAGE_EPI_ACCUM) == SAGE_EPI_NONE && n > 64 && slots > 0 && tiles > slots / 2 &&
      tiles * 10 < cdiv(tiles, slots) * slots * 9 && !short_k) {
    g.streamk = 1;
    g.atomic_out = accum ? 1 : 0;   // out += product: whole tiles are added too
    if (!accum) zero_rows_kernel<<<static_cast<unsigned>(cdiv(m * n, kThreads)), kThreads, 0, stream>>>(out, ld_out, m, n);
    SAGE_CHECK_LAUNCH("zero_rows_kernel");
    int vec = 4;
    for (int sgi = 0; sgi < g.nseg; ++sgi) {
      if (g.seg[sgi].a.vec < vec) vec = g.seg[sgi].a.vec;
      if (g.seg[sgi].b.vec < vec) vec = g.seg[sgi].b.vec;
    }
    const dim3 grid(static_cast<unsigned>(slots));
    if (vec == 4) SAGE_LAUNCH_PROBED((gemm_kernel<true, true, 64, 128, 4>), grid, kThreads, stream, g);
    else if (vec == 2) SAGE_LAUNCH_PROBED((gemm_kernel<true, true, 64, 128, 2>), grid, kThreads, stream, g);
    else SAGE_LAUNCH_PROBED((gemm_kernel<true, true, 64, 128, 1>), grid, kThreads, stream, g);
    SAGE_CHECK_LAUNCH("gemm_kernel(fwd, stream-K)");
    return SAGE_OK;
  }
  if (accum && !zeroed) g.rmw_out = 1;   // out += product: every element has one owner, plain load-add-store
  return launch_gemm<true, true>(g, 1, stream, "gemm_kernel(fwd)");
}

extern "C" int sage_linear_concat_fwd(const float* a1, sage_i64 ld_a1, const sage_i64* rows1,
                                      int k1, const float* w1, sage_i64 ld_w1, const float* a2,
                                      sage_i64 ld_a2, const sage_i64* rows2, int k2,
                                      const float* w2, sage_i64 ld_w2, const float* bias,
                                      int epilogue, sage_i64 m, const sage_i64* m_dev, int n,
                                      float* out, sage_i64 ld_out, void* stream) {
  return linear_concat_fwd_impl(a1, ld_a1, rows1, k1, w1, ld_w1, a2, ld_a2, rows2, k2, w2, ld_w2, bias,
                                epilogue, m, m_dev, n, out, ld_out, nullptr, 0, stream);
}

extern "C" int sage_pool_linear_fwd(const float* x, sage_i64 ld_x, const sage_i64* rows, int k,
                                    const float* w, sage_i64 ld_w, const float* bias, sage_i64 m,
                                    const sage_i64* m_dev, int n, float* p, sage_i64 ld_p, float* dpre,
                                    sage_i64 ld_dpre, void* stream) {
  SAGE_REQUIRE(bias != nullptr && p != dpre, SAGE_EINVAL, "sage_pool_linear_fwd: NULL bias or dpre aliasing p");
  return linear_concat_fwd_impl(x, ld_x, rows, k, w, ld_w, nullptr, 0, nullptr, 0, nullptr, 0, bias,
                                SAGE_EPI_BIAS | SAGE_EPI_RELU, m, m_dev, n, p, ld_p, dpre, ld_dpre, stream);
}

extern "C" int sage_linear_dinput(const float* dout, sage_i64 ld_dout, const float* w,
                                  sage_i64 ld_w, sage_i64 m, const sage_i64* m_dev, int n, int k,
                                  float* da, sage_i64 ld_da, int accumulate, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && n > 0 && k > 0 && ld_dout >= n && ld_w >= k && ld_da >= k, SAGE_EINVAL,
               "sage_linear_dinput: bad shape (m=%lld n=%d k=%d)", (long long)m, n, k);
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(dout && w && da, SAGE_EINVAL, "sage_linear_dinput: NULL pointer argument");
  // C[m][kk] = sum_nn dout[m][nn] * w[nn][kk]: contraction over n, output columns = k
  GemmArgs g = {};
  g.seg[0].a = {dout, ld_dout, nullptr, op_vec(dout, ld_dout, n)};
  g.seg[0].b = {w, ld_w, nullptr, op_vec(w, ld_w, k)};
  g.seg[0].k = n;
  g.nseg = 1;
  g.m = m;
  g.m_dev = m_dev;
  g.n = k;
  g.out = da;
  g.ld_out = ld_da;
  g.atomic_out = accumulate ? 1 : 0;
  if (n <= 64 && g.seg[0].b.vec == 4) {
    SkinnyArgs s = {};
    s.seg[0] = g.seg[0];
    s.nseg = 1;
    s.m = m; s.m_dev = m_dev; s.n = k; s.out = da; s.ld_out = ld_da; s.accumulate = accumulate ? 1 : 0;
    skinny_dinput_kernel<<<static_cast<unsigned>(cdiv(m, kThreads / 64)), kThreads, 0, stream>>>(s);
    SAGE_CHECK_LAUNCH("skinny_dinput_kernel");
    return SAGE_OK;
  }
  return launch_gemm<true, false>(g, 1, stream, "gemm_kernel(dinput)");
}

extern "C" int sage_linear_dinput2(const float* dout, sage_i64 ld_dout, const float* w1,
                                   sage_i64 ld_w1, float* da1, sage_i64 ld_da1, const float* w2,
                                   sage_i64 ld_w2, float* da2, sage_i64 ld_da2, sage_i64 m,
                                   const sage_i64* m_dev, int n, int k, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && n > 0 && k > 0 && ld_dout >= n && ld_w1 >= k && ld_w2 >= k && ld_da1 >= k &&
                   ld_da2 >= k,
               SAGE_EINVAL, "sage_linear_dinput2: bad shape (m=%lld n=%d k=%d)", (long long)m, n, k);
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(dout && w1 && w2 && da1 && da2, SAGE_EINVAL, "sage_linear_dinput2: NULL pointer argument");
  if (n <= 64 && op_vec(w1, ld_w1, k) == 4 && op_vec(w2, ld_w2, k) == 4) {
    SkinnyArgs s = {};
    s.seg[0].a = {dout, ld_dout, nullptr, op_vec(dout, ld_dout, n)};
    s.seg[0].b = {w1, ld_w1, nullptr, 4};
    s.seg[0].k = n;
    s.seg[1] = s.seg[0];
    s.seg[1].b = {w2, ld_w2, nullptr, 4};
    s.nseg = 2;
    s.m = m; s.m_dev = m_dev; s.n = k; s.out = da1; s.ld_out = ld_da1; s.out2 = da2; s.ld_out2 = ld_da2;
    dim3 grid(static_cast<unsigned>(cdiv(m, kThreads / 64)), 2);
    skinny_dinput_kernel<<<grid, kThreads, 0, stream>>>(s);
    SAGE_CHECK_LAUNCH("skinny_dinput_kernel(pair)");
    return SAGE_OK;
  }
  // Few output tiles (the 512 seed rows of a wide last layer: 8 x 2 tiles per product, whose 8
  // dependent K-steps took 17 us per launch): both products in one grid (they share dout), the
  // contraction split over blockIdx.z, partial sums added with fp32 atomics into zeroed outputs
  // (SAGE_DINPUT2_SERIAL=1: the two whole-K launches, the A/B switch)
  static const char* serial_env = getenv("SAGE_DINPUT2_SERIAL");
  const int64_t tiles = cdiv(m, 64) * 2 * cdiv(k, 128);
  if (!(serial_env && serial_env[0] == '1') && tiles <= 128 && n > 2 * BK) {
    GemmArgs g = {};
    g.seg[0].a = {dout, ld_dout, nullptr, op_vec(dout, ld_dout, n)};
    g.seg[0].b = {w1, ld_w1, nullptr, op_vec(w1, ld_w1, k)};
    g.seg[0].k = n;
    g.seg[1].a = g.seg[0].a;
    g.seg[1].b = {w2, ld_w2, nullptr, op_vec(w2, ld_w2, k)};
    g.seg[1].k = n;
    g.nseg = 2;
    g.batched = 1;
    g.m = m;
    g.m_dev = m_dev;
    g.n = k;
    g.out = da1;
    g.ld_out = ld_da1;
    g.n2 = k;
    g.out2 = da2;
    g.ld_out2 = ld_da2;
    g.ny0 = static_cast<int>(cdiv(k, 128));
    g.atomic_out = 1;
    int splits = static_cast<int>(cdiv(n, 2 * BK));      // two K-steps per slice, more beyond 8 slices
    if (splits > 8) splits = 8;
    g.k_chunk = static_cast<int>(cdiv(cdiv(n, splits), BK) * BK);
    splits = static_cast<int>(cdiv(n, g.k_chunk));
    const dim3 zgrid(static_cast<unsigned>(cdiv(m * k, kThreads)), 2);
    zero_rows2_kernel<<<zgrid, kThreads, 0, stream>>>(da1, ld_da1, da2, ld_da2, m, m_dev, k);
    SAGE_CHECK_LAUNCH("zero_rows2_kernel");
    return launch_gemm<true, false>(g, splits, stream, "gemm_kernel(dinput pair, split-K)");
  }
  int rc = sage_linear_dinput(dout, ld_dout, w1, ld_w1, m, m_dev, n, k, da1, ld_da1, 0, stream_);
  if (rc != SAGE_OK) return rc;
  return sage_linear_dinput(dout, ld_dout, w2, ld_w2, m, m_dev, n, k, da2, ld_da2, 0, stream_);
}

extern "C" int sage_linear_dweight(const float* dout, sage_i64 ld_dout, const float* a,
                                   sage_i64 ld_a, const sage_i64* rows, sage_i64 m,
                                   const sage_i64* m_dev, int n, int k, float* dw, sage_i64 ld_dw,
                                   float* dbias, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SAGE_REQUIRE(m >= 0 && n > 0 && k > 0 && ld_dout >= n && ld_a >= k && ld_dw >= k, SAGE_EINVAL,
               "sage_linear_dweight: bad shape (m=%lld n=%d k=%d)", (long long)m, n, k);
  SAGE_REQUIRE(m < (1ll << 31), SAGE_ELIMIT, "sage_linear_dweight: m too large");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(dout && a && dw, SAGE_EINVAL, "sage_linear_dweight: NULL pointer argument");
  // C[nn][kk] = sum_mm dout[mm][nn] * a[rows(mm)][kk]: contraction over m
  GemmArgs g = {};
  g.seg[0].a = {dout, ld_dout, nullptr, op_vec(dout, ld_dout, n)};
  g.seg[0].b = {a, ld_a, rows, op_vec(a, ld_a, k)};
  g.seg[0].k = static_cast<int>(m);
  g.nseg = 1;
  g.m = n;
  g.k_dev = m_dev;
  g.n = k;
  g.out = dw;
  g.ld_out = ld_dw;
  g.atomic_out = 1;
  g.colsum = dbias;
  if (n <= 64) {
    const int64_t col_blocks = cdiv(k, kThreads);
    int64_t sp = cdiv(m, 32);       // >= 32 rows of m per block, <= 16 slices (atomic contention)
    if (sp > 16) sp = 16;
    if (sp < 1) sp = 1;
    const int m_chunk = static_cast<int>(cdiv(m, sp));
    const int slices = static_cast<int>(cdiv(m, m_chunk));
    dim3 grid(static_cast<unsigned>(col_blocks), static_cast<unsigned>(cdiv(n, kDwRows)),
              static_cast<unsigned>(slices));
    const SkinnyDwProblem p0 = {a, ld_a, rows, k, dw, ld_dw};
    skinny_dweight_kernel<<<grid, kThreads, 0, stream>>>(dout, ld_dout, p0, p0, slices, m, m_dev, n,
                                                         m_chunk, dbias);
    SAGE_CHECK_LAUNCH("skinny_dweight_kernel");
    return SAGE_OK;
  }
  // split the contraction so that the grid covers the chip about twice
  const int64_t tiles = cdiv(n, 64) * cdiv(k, k > 64 ? 128 : 64);
  int64_t splits = 768 / tiles;   // one full round at 3 resident workgroups per CU
  const int64_t max_splits = cdiv(m, dw_min_steps() * BK);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  g.k_chunk = static_cast<int>(cdiv(cdiv(m, splits), BK) * BK);
  splits = cdiv(m, g.k_chunk);
  // (m >= 2,048 contracted rows: the lstm cell's weight gradient over the 512 seed rows — 1,204 x
  // 15,050 outputs, 18 GFLOP — was tried here on its flop count and is slower than on the exact-fp32
  // kernel, 5.07 against 5.00 ms per lstm step: 16 K-steps per tile do not pay for the split)
  if (x3_mode() && k > 64 && g.seg[0].a.vec == 4 && g.seg[0].b.vec == 4 && m >= 64 * BK) {
    g.xcd_map = xcd_mode() ? 1 : 0;
    const char* bm_env = getenv("SAGE_GEMM_BM");      // "64": keep the 64-row tile (A/B switch)
    if (n >= 256 && !(bm_env && bm_env[0] == '6')) {
      // 128 x 128 tiles, two workgroups per CU: one full round of 512
      const int64_t tiles2 = cdiv(n, 128) * cdiv(k, 128);
      int64_t sp = balanced_splits(tiles2, 512, max_splits);
      g.k_chunk = static_cast<int>(cdiv(cdiv(m, sp), BK) * BK);
      sp = cdiv(m, g.k_chunk);
      const dim3 grid(static_cast<unsigned>(cdiv(n, 128)), static_cast<unsigned>(cdiv(k, 128)),
                      static_cast<unsigned>(sp));
      SAGE_LAUNCH_PROBED((gemm_x3_dw_kernel<128>), grid, kThreads, stream, g);
    } else {
      const dim3 grid(static_cast<unsigned>(cdiv(n, 64)), static_cast<unsigned>(cdiv(k, 128)),
                      static_cast<unsigned>(splits));
      SAGE_LAUNCH_PROBED((gemm_x3_dw_kernel<64>), grid, kThreads, stream, g);
    }
    SAGE_CHECK_LAUNCH("gemm_x3_dw_kernel");
    return SAGE_OK;
  }
  return launch_gemm<false, false>(g, static_cast<int>(splits), stream, "gemm_kernel(dweight)");
}

static int linear_dweight2_impl(const float* dout, sage_i64 ld_dout, const float* a1,
                                sage_i64 ld_a1, const sage_i64* rows1, int k1, float* dw1,
                                sage_i64 ld_dw1, const float* a2, sage_i64 ld_a2,
                                const sage_i64* rows2, int k2, float* dw2, sage_i64 ld_dw2,
                                sage_i64 m, const sage_i64* m_dev, int n, sage_i64 part_stride, int max_parts,
                                int* parts_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (parts_out != nullptr) *parts_out = 1;       // unless the sliced path below is taken: sums in dw1 / dw2
  SAGE_REQUIRE(m >= 0 && n > 0 && k1 > 0 && k2 > 0 && ld_dout >= n && ld_a1 >= k1 && ld_a2 >= k2 &&
                   ld_dw1 >= k1 && ld_dw2 >= k2,
               SAGE_EINVAL, "sage_linear_dweight2: bad shape (m=%lld n=%d k1=%d k2=%d)", (long long)m,
               n, k1, k2);
  SAGE_REQUIRE(m < (1ll << 31), SAGE_ELIMIT, "sage_linear_dweight2: m too large");
  if (m == 0) return SAGE_OK;
  SAGE_REQUIRE(dout && a1 && a2 && dw1 && dw2, SAGE_EINVAL, "sage_linear_dweight2: NULL pointer argument");
  if (n <= 64) {  // skinny: both problems in one VALU grid
    const int kmax = k1 > k2 ? k1 : k2;
    int64_t sp = cdiv(m, 32);       // >= 32 rows of m per block, <= 16 slices (atomic contention)
    if (sp > 16) sp = 16;
    if (sp < 1) sp = 1;
    const int m_chunk = static_cast<int>(cdiv(m, sp));
    const int slices = static_cast<int>(cdiv(m, m_chunk));
    dim3 grid(static_cast<unsigned>(cdiv(kmax, kThreads)), static_cast<unsigned>(cdiv(n, kDwRows)),
              static_cast<unsigned>(2 * slices));
    const SkinnyDwProblem p0 = {a1, ld_a1, rows1, k1, dw1, ld_dw1};
    const SkinnyDwProblem p1 = {a2, ld_a2, rows2, k2, dw2, ld_dw2};
    skinny_dweight_kernel<<<grid, kThreads, 0, stream>>>(dout, ld_dout, p0, p1, slices, m, m_dev, n,
                                                         m_chunk, nullptr);
    SAGE_CHECK_LAUNCH("skinny_dweight_kernel(pair)");
    return SAGE_OK;
  }
  static const char* narrow_env = getenv("SAGE_DW2_NARROW_SERIAL");      // "1": two launches (the A/B switch)
  if (k1 <= 64 && k2 <= 64 && !(narrow_env && narrow_env[0] == '1')) {
    // narrow inputs on both sides (PPI's 50 features): both problems in one grid of 64 x 64 tiles
    // (two launches of 416 workgroups took 12 + 9.5 us, one of 768 takes 12)
    GemmArgs g = {};
    g.seg[0].a = {dout, ld_dout, nullptr, op_vec(dout, ld_dout, n)};
    g.seg[0].b = {a1, ld_a1, rows1, op_vec(a1, ld_a1, k1)};
    g.seg[0].k = static_cast<int>(m);
    g.seg[1].a = g.seg[0].a;
    g.seg[1].b = {a2, ld_a2, rows2, op_vec(a2, ld_a2, k2)};
    g.seg[1].k = static_cast<int>(m);
    g.nseg = 2;
    g.batched = 1;
    g.m = n;
    g.k_dev = m_dev;
    g.n = k1;
    g.out = dw1;
    g.ld_out = ld_dw1;
    g.n2 = k2;
    g.out2 = dw2;
    g.ld_out2 = ld_dw2;
    g.ny0 = 1;
    g.atomic_out = 1;
    const int64_t tiles = cdiv(n, 64) * 2;
    int64_t splits = 768 / tiles;   // one full round at 3 resident workgroups per CU
    const int64_t max_splits = cdiv(m, dw_min_steps() * BK);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    g.k_chunk = static_cast<int>(cdiv(cdiv(m, splits), BK) * BK);
    splits = cdiv(m, g.k_chunk);
    int vec = 4;
    for (int sgi = 0; sgi < 2; ++sgi) {
      if (g.seg[sgi].a.vec < vec) vec = g.seg[sgi].a.vec;
      if (g.seg[sgi].b.vec < vec) vec = g.seg[sgi].b.vec;
    }
    const dim3 grid(static_cast<unsigned>(cdiv(n, 64)), 2, static_cast<unsigned>(splits));
    if (vec == 4) SAGE_LAUNCH_PROBED((gemm_kernel<false, false, 64, 64, 4>), grid, kThreads, stream, g);
    else if (vec == 2) SAGE_LAUNCH_PROBED((gemm_kernel<false, false, 64, 64, 2>), grid, kThreads, stream, g);
    else SAGE_LAUNCH_PROBED((gemm_kernel<false, false, 64, 64, 1>), grid, kThreads, stream, g);
    SAGE_CHECK_LAUNCH("gemm_kernel(dweight2, narrow pair)");
    return SAGE_OK;
  }
  if (k1 <= 64 || k2 <= 64) {  // one narrow input: the single-problem tile shapes are better
    int rc = sage_linear_dweight(dout, ld_dout, a1, ld_a1, rows1, m, m_dev, n, k1, dw1, ld_dw1, nullptr, stream_);
    if (rc != SAGE_OK) return rc;
    return sage_linear_dweight(dout, ld_dout, a2, ld_a2, rows2, m, m_dev, n, k2, dw2, ld_dw2, nullptr, stream_);
  }
  GemmArgs g = {};
  g.seg[0].a = {dout, ld_dout, nullptr, op_vec(dout, ld_dout, n)};
  g.seg[0].b = {a1, ld_a1, rows1, op_vec(a1, ld_a1, k1)};
  g.seg[0].k = static_cast<int>(m);
  g.seg[1].a = g.seg[0].a;
  g.seg[1].b = {a2, ld_a2, rows2, op_vec(a2, ld_a2, k2)};
  g.seg[1].k = static_cast<int>(m);
  g.nseg = 2;
  g.batched = 1;
  g.m = n;
  g.k_dev = m_dev;
  g.n = k1;
  g.out = dw1;
  g.ld_out = ld_dw1;
  g.n2 = k2;
  g.out2 = dw2;
  g.ld_out2 = ld_dw2;
  g.ny0 = static_cast<int>(cdiv(k1, 128));
  g.atomic_out = 1;
  // both problems in one grid: aim at <= 4 workgroups per CU
  const int64_t tiles = cdiv(n, 64) * (cdiv(k1, 128) + cdiv(k2, 128));
  // 3 workgroups per CU are resident (51 KB LDS, 168 VGPRs each): 768 = one full round
  int64_t splits = 768 / tiles;
  const int64_t max_splits = cdiv(m, dw_min_steps() * BK);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  g.k_chunk = static_cast<int>(cdiv(cdiv(m, splits), BK) * BK);
  splits = cdiv(m, g.k_chunk);
  if (x3_mode() && g.seg[0].a.vec == 4 && g.seg[0].b.vec == 4 && g.seg[1].b.vec == 4 && m >= 64 * BK) {
    g.xcd_map = xcd_mode() ? 1 : 0;
    const char* bm_env = getenv("SAGE_GEMM_BM");      // "64": keep the 64-row tile (A/B switch)
    if (n >= 256 && !(bm_env && bm_env[0] == '6')) {
      const int64_t tiles2 = cdiv(n, 128) * (cdiv(k1, 128) + cdiv(k2, 128));
      int64_t sp = balanced_splits(tiles2, 512, max_splits);
      // a narrow input (F = 100: 4 tiles) would be cut into 128 slices of 4 K-steps whose every
      // output element then takes 128 atomics: half as many slices, twice as long, are faster
      // (40 against 51 us alone, 0.499 against 0.506 ms per products-scale unsupervised step;
      // SAGE_DW_SPLIT_SLOTS=512: the old rule, the A/B switch)
      static const char* sp_env = getenv("SAGE_DW_SPLIT_SLOTS");
      if (tiles2 <= 4) sp = balanced_splits(tiles2, sp_env ? atoi(sp_env) : 256, max_splits);
      g.k_chunk = static_cast<int>(cdiv(cdiv(m, sp), BK) * BK);
      sp = cdiv(m, g.k_chunk);
      if (parts_out != nullptr && sp > 1 && sp <= max_parts) {
        g.part_stride = part_stride;
        *parts_out = static_cast<int>(sp);
      }
      const dim3 grid(static_cast<unsigned>(cdiv(n, 128)), static_cast<unsigned>(g.ny0 + cdiv(k2, 128)),
                      static_cast<unsigned>(sp));
      SAGE_LAUNCH_PROBED((gemm_x3_dw_kernel<128>), grid, kThreads, stream, g);
    } else {
      const dim3 grid(static_cast<unsigned>(cdiv(n, 64)), static_cast<unsigned>(g.ny0 + cdiv(k2, 128)),
                      static_cast<unsigned>(splits));
      SAGE_LAUNCH_PROBED((gemm_x3_dw_kernel<64>), grid, kThreads, stream, g);
    }
    SAGE_CHECK_LAUNCH("gemm_x3_dw_kernel(pair)");
    return SAGE_OK;
  }
  return launch_gemm<false, false>(g, static_cast<int>(splits), stream, "gemm_kernel(dweight2)");
}

extern "C" int sage_linear_dweight2(const float* dout, sage_i64 ld_dout, const float* a1,
                                    sage_i64 ld_a1, const sage_i64* rows1, int k1, float* dw1,
                                    sage_i64 ld_dw1, const float* a2, sage_i64 ld_a2,
                                    const sage_i64* rows2, int k2, float* dw2, sage_i64 ld_dw2,
                                    sage_i64 m, const sage_i64* m_dev, int n, void* stream_) {
  return linear_dweight2_impl(dout, ld_dout, a1, ld_a1, rows1, k1, dw1, ld_dw1, a2, ld_a2, rows2, k2, dw2, ld_dw2, m, m_dev,
                              n, 0, 0, nullptr, stream_);
}

extern "C" int sage_linear_dweight2_parts(const float* dout, sage_i64 ld_dout, const float* a1,
                                          sage_i64 ld_a1, const sage_i64* rows1, int k1, float* dw1,
                                          sage_i64 ld_dw1, const float* a2, sage_i64 ld_a2,
                                          const sage_i64* rows2, int k2, float* dw2, sage_i64 ld_dw2,
                                          sage_i64 m, const sage_i64* m_dev, int n, sage_i64 part_stride,
                                          int max_parts, int* parts, void* stream_) {
  SAGE_REQUIRE(parts != nullptr && max_parts >= 1 && part_stride > 0, SAGE_EINVAL,
               "sage_linear_dweight2_parts: parts is NULL, max_parts < 1 or part_stride <= 0");
  // *parts on entry: 0, or what an earlier call with these shapes returned — a caller that replays
  // recorded calls (and has told its consumer that number) learns here if the route has changed
  // under it (SAGE_GEMM_X3 / SAGE_GEMM_BM are read at every call)
  const int expected = *parts;
  const int rc = linear_dweight2_impl(dout, ld_dout, a1, ld_a1, rows1, k1, dw1, ld_dw1, a2, ld_a2, rows2, k2, dw2, ld_dw2, m,
                                      m_dev, n, part_stride, max_parts, parts, stream_);
  if (rc != SAGE_OK) return rc;
  SAGE_REQUIRE(expected == 0 || expected == *parts, SAGE_EINVAL,
               "sage_linear_dweight2_parts: %d parts now, %d when the caller first asked", *parts, expected);
  return SAGE_OK;
}
