// Graph structure kernels: stable COO(int64) -> CSR(int32), degree
// normalisation, argmax assignment, dense adjacency.
//
// Integer/byte work, HBM- (at LRGB batch sizes: latency-) bound.  No float
// atomics: the only atomics are int32 counters whose final values do not depend
// on arrival order, and exact +1.0f adds in to_dense_adj.
#include "hscn_common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048 entries per block

// ---- histogram: cnt[key+1] += 1 -------------------------------------------------
__global__ void k_hist(const int64_t* __restrict__ key, const int64_t* __restrict__ other, int64_t E,
                       int64_t num_rows, int64_t num_cols, int32_t* __restrict__ cnt,
                       int32_t* __restrict__ flag) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t k = key[e], o = other[e];
  if (k < 0 || k >= num_rows || o < 0 || o >= num_cols) {
    if (flag) atomicOr(flag, 1);
    return;
  }
  atomicAdd(&cnt[k + 1], 1);
}

// ---- block-wise inclusive scan of cnt[0..n] (n+1 entries) in place ---------------
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int& total) {
  // wave scan then cross-wave
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) lds[w] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
    int s = lds[i];
    if (i < w) woff += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return woff + incl - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_partial(const int32_t* __restrict__ a, int64_t n,
                                                               int32_t* __restrict__ blocksum) {
  __shared__ int lds[SCAN_THREADS / 64];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    if (idx < n) s += a[idx];
  }
  int total;
  block_exclusive_scan(s, lds, total);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) k_scan_blocksums(int32_t* __restrict__ blocksum, int nblocks) {
  __shared__ int lds[16];
  int carry = 0;
  for (int base = 0; base < nblocks; base += 1024) {
    int idx = base + threadIdx.x;
    int v = idx < nblocks ? blocksum[idx] : 0;
    int total;
    int ex = block_exclusive_scan(v, lds, total);
    if (idx < nblocks) blocksum[idx] = carry + ex;
    carry += total;
  }
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_final(int32_t* __restrict__ a, int64_t n,
                                                             const int32_t* __restrict__ blockoff) {
  __shared__ int lds[SCAN_THREADS / 64];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    v[i] = idx < n ? a[idx] : 0;
    s += v[i];
  }
  int total;
  int ex = block_exclusive_scan(s, lds, total) + blockoff[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    ex += v[i];
    if (idx < n) a[idx] = ex;  // inclusive
  }
}

// Small-n path: a single block scans everything (one launch instead of three).
__global__ void __launch_bounds__(1024) k_scan_single(int32_t* __restrict__ a, int64_t n) {
  __shared__ int lds[16];
  int carry = 0;
  for (int64_t base = 0; base < n; base += 1024 * 4) {
    int64_t i0 = base + (int64_t)threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = (i0 + i) < n ? a[i0 + i] : 0;
      s += v[i];
    }
    int total;
    int ex = block_exclusive_scan(s, lds, total) + carry;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ex += v[i];
      if ((i0 + i) < n) a[i0 + i] = ex;
    }
    carry += total;
  }
}

// ---- fill: unordered placement through an int cursor ----------------------------
__global__ void k_fill(const int64_t* __restrict__ key, const int64_t* __restrict__ other, int64_t E,
                       int64_t num_rows, int64_t num_cols, const int32_t* __restrict__ rowptr,
                       int32_t* __restrict__ cursor, int32_t* __restrict__ tmp) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t k = key[e], o = other[e];
  if (k < 0 || k >= num_rows || o < 0 || o >= num_cols) return;
  int p = atomicAdd(&cursor[k], 1);
  tmp[rowptr[k] + p] = (int32_t)e;
}

// ---- rank inside the row by edge id => stable, arrival-order independent ---------
__global__ void k_rank(const int64_t* __restrict__ key, const int64_t* __restrict__ other, int64_t E,
                       int64_t num_rows, int64_t num_cols, const int32_t* __restrict__ rowptr,
                       const int32_t* __restrict__ tmp, int32_t* __restrict__ col, int32_t* __restrict__ eid) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t k = key[e], o = other[e];
  if (k < 0 || k >= num_rows || o < 0 || o >= num_cols) return;
  int s = rowptr[k], t = rowptr[k + 1];
  int rank = 0;
  int me = (int)e;
  for (int q = s; q < t; ++q) rank += (tmp[q] < me) ? 1 : 0;
  col[s + rank] = (int32_t)o;
  eid[s + rank] = me;
}

// ---- both CSRs of one edge list (hscn_csr_build_pair): side 0 keyed by dst, side 1 keyed by src ----------------
struct CsrSide {
  int64_t nrows;            // rows of this side's CSR
  int32_t* rowptr;          // [nrows + 1]
  int32_t* cursor;          // [nrows]
  int32_t* tmp;             // [E]
  int32_t* col;             // [E]
  int32_t* eid;             // [E]
  int32_t* blocksum;        // scan scratch
};
struct CsrPair {
  CsrSide s[2];
};

__global__ void k_zero_pair(const CsrPair P) {
  const int64_t n0 = P.s[0].nrows + 1, c0 = P.s[0].nrows, n1 = P.s[1].nrows + 1, c1 = P.s[1].nrows;
  const int64_t total = n0 + c0 + n1 + c1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n0) P.s[0].rowptr[i] = 0;
    else if (i < n0 + c0) P.s[0].cursor[i - n0] = 0;
    else if (i < n0 + c0 + n1) P.s[1].rowptr[i - n0 - c0] = 0;
    else P.s[1].cursor[i - n0 - c0 - n1] = 0;
  }
}

__global__ void k_hist_pair(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, const CsrPair P,
                            int32_t* __restrict__ flag) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t u = src[e], v = dst[e];
  if (v < 0 || v >= P.s[0].nrows || u < 0 || u >= P.s[1].nrows) {
    if (flag) atomicOr(flag, 1);
    return;
  }
  atomicAdd(&P.s[0].rowptr[v + 1], 1);
  atomicAdd(&P.s[1].rowptr[u + 1], 1);
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_partial_pair(const CsrPair P) {
  __shared__ int lds[SCAN_THREADS / 64];
  const CsrSide& S = P.s[blockIdx.y];
  const int64_t n = S.nrows + 1;
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  if ((int64_t)blockIdx.x * SCAN_TILE >= n) return;      // (the grid is sized for the longer side)
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    if (idx < n) s += S.rowptr[idx];
  }
  int total;
  block_exclusive_scan(s, lds, total);
  if (threadIdx.x == 0) S.blocksum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) k_scan_blocksums_pair(const CsrPair P) {
  __shared__ int lds[16];
  const CsrSide& S = P.s[blockIdx.x];
  const int nblocks = (int)((S.nrows + 1 + SCAN_TILE - 1) / SCAN_TILE);
  int carry = 0;
  for (int base = 0; base < nblocks; base += 1024) {
    int idx = base + threadIdx.x;
    int v = idx < nblocks ? S.blocksum[idx] : 0;
    int total;
    int ex = block_exclusive_scan(v, lds, total);
    if (idx < nblocks) S.blocksum[idx] = carry + ex;
    carry += total;
  }
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_final_pair(const CsrPair P) {
  __shared__ int lds[SCAN_THREADS / 64];
  const CsrSide& S = P.s[blockIdx.y];
  const int64_t n = S.nrows + 1;
  if ((int64_t)blockIdx.x * SCAN_TILE >= n) return;
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    v[i] = idx < n ? S.rowptr[idx] : 0;
    s += v[i];
  }
  int total;
  int ex = block_exclusive_scan(s, lds, total) + S.blocksum[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    ex += v[i];
    if (idx < n) S.rowptr[idx] = ex;  // inclusive
  }
}

__global__ void k_fill_pair(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, const CsrPair P) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t u = src[e], v = dst[e];
  if (v < 0 || v >= P.s[0].nrows || u < 0 || u >= P.s[1].nrows) return;
  const int p0 = atomicAdd(&P.s[0].cursor[v], 1);
  P.s[0].tmp[P.s[0].rowptr[v] + p0] = (int32_t)e;
  const int p1 = atomicAdd(&P.s[1].cursor[u], 1);
  P.s[1].tmp[P.s[1].rowptr[u] + p1] = (int32_t)e;
}

__global__ void k_rank_pair(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, const CsrPair P) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t u = src[e], v = dst[e];
  if (v < 0 || v >= P.s[0].nrows || u < 0 || u >= P.s[1].nrows) return;
  const CsrSide& S = P.s[blockIdx.y];
  const int64_t k = blockIdx.y == 0 ? v : u, o = blockIdx.y == 0 ? u : v;
  const int s = S.rowptr[k], t = S.rowptr[k + 1];
  int rank = 0;
  const int me = (int)e;
  for (int q = s; q < t; ++q) rank += (S.tmp[q] < me) ? 1 : 0;
  S.col[s + rank] = (int32_t)o;
  S.eid[s + rank] = me;
}

__global__ void k_inv_pos(const int32_t* __restrict__ eid, int64_t E, int32_t* __restrict__ inv) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < E) inv[eid[p]] = (int32_t)p;
}
__global__ void k_pos_t(const int32_t* __restrict__ eid_t, const int32_t* __restrict__ inv, int64_t E,
                        int32_t* __restrict__ pos_t) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < E) pos_t[q] = inv[eid_t[q]];
}

__global__ void k_dinv(const int32_t* __restrict__ rowptr, int64_t n, float* __restrict__ dinv) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int d = rowptr[i + 1] - rowptr[i];
  // torch: deg.pow(-0.5) == 1/sqrt(deg) (both correctly rounded); inf -> 0
  dinv[i] = d > 0 ? 1.0f / sqrtf((float)d) : 0.f;
}

__global__ void k_wdeg_dinv(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ eid,
                            const float* __restrict__ w, int64_t n, float* __restrict__ dinv) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float deg = 0.f;
  for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) deg = add_rn(deg, w ? w[eid[p]] : 1.0f);
  float r = 1.0f / sqrtf(deg);
  dinv[i] = isinf(r) ? 0.f : r;
}

__global__ void k_wnorm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                        const int32_t* __restrict__ eid, const float* __restrict__ w,
                        const float* __restrict__ dinv, int64_t n, float* __restrict__ wn) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float di = dinv[i];
  for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
    int e = eid[p];
    float we = w ? w[e] : 1.0f;
    wn[e] = mul_rn(mul_rn(dinv[col[p]], we), di);  // dis[row]*w*dis[col]
  }
}

__global__ void k_argmax(const float* __restrict__ S, int64_t* __restrict__ ids, int64_t n, int K) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = S + i * K;
  float best = r[0];
  int bi = 0;
  for (int k = 1; k < K; ++k) {
    float v = r[k];
    if (v > best) { best = v; bi = k; }
  }
  ids[i] = bi;
}

__global__ void k_dense_adj(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t E,
                            int64_t n, float* __restrict__ adj) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t r = row[e], c = col[e];
  if (r < 0 || r >= n || c < 0 || c >= n) return;
  atomicAdd(&adj[r * n + c], 1.0f);  // exact: integer-valued, order independent
}

// block-diagonal batch of B graphs with n nodes each -> [B, n, n]; an edge that leaves its block is dropped
__global__ void k_dense_adj_batched(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t E,
                                    int64_t B, int64_t n, float* __restrict__ adj) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t r = row[e], c = col[e];
  if (r < 0 || r >= B * n || c < 0 || c >= B * n) return;
  const int64_t b = r / n;
  if (c / n != b) return;
  atomicAdd(&adj[(b * n + (r - b * n)) * n + (c - b * n)], 1.0f);
}

// block-diagonal batch of graphs of DIFFERENT sizes -> [B, nmax, nmax] (zero beyond a graph's n_b).  gid[i] = graph of
// node i, nptr[b] = first node of graph b.  mode 0: adj[b, r, c] += 1 for every edge (to_dense_adj on the list as
// given); mode 1: self loops of the list are skipped and the identity is added -- the adjacency
// add_remaining_self_loops + to_dense_adj produce from RAW edges (every node ends with exactly one loop).
// Threads [0, E) take the edges, threads [E, E + N) the diagonal.
__global__ void k_dense_adj_ragged(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t E,
                                   const int32_t* __restrict__ nptr, const int32_t* __restrict__ gid, int64_t N,
                                   int64_t nmax, int mode, float* __restrict__ adj) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < E) {
    const int64_t r = row[t], c = col[t];
    if (r < 0 || r >= N || c < 0 || c >= N) return;
    if (mode == 1 && r == c) return;
    const int b = gid[r];
    if (gid[c] != b) return;
    const int64_t r0 = nptr[b];
    atomicAdd(&adj[((int64_t)b * nmax + (r - r0)) * nmax + (c - r0)], 1.0f);
  } else if (mode == 1 && t < E + N) {
    const int64_t i = t - E;
    const int b = gid[i];
    const int64_t li = i - nptr[b];
    atomicAdd(&adj[((int64_t)b * nmax + li) * nmax + li], 1.0f);
  }
}

// the same counts as BYTES: adj8 [B, nmax, lda8], lda8 = nmax rounded up to 32 (16-byte loads of a row never leave
// it), zero on entry -- a quarter of the float adjacency's bytes for the matrix-core products that stream it
// (csrc/dense.hip k_adj_s).  A byte is bumped through an atomic add on its aligned word; a count that would pass
// 255 (more than 255 parallel edges) raises flag bit 16 instead of carrying into the neighbour.
__global__ void k_dense_adj_ragged_u8(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t E,
                                      const int32_t* __restrict__ nptr, const int32_t* __restrict__ gid, int64_t N,
                                      int64_t nmax, int64_t lda8, int mode, unsigned int* __restrict__ adjw,
                                      int32_t* __restrict__ flag) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t o;
  if (t < E) {
    const int64_t r = row[t], c = col[t];
    if (r < 0 || r >= N || c < 0 || c >= N) return;
    if (mode == 1 && r == c) return;
    const int b = gid[r];
    if (gid[c] != b) return;
    const int64_t r0 = nptr[b];
    o = ((int64_t)b * nmax + (r - r0)) * lda8 + (c - r0);
  } else if (mode == 1 && t < E + N) {
    const int64_t i = t - E;
    const int b = gid[i];
    const int64_t li = i - nptr[b];
    o = ((int64_t)b * nmax + li) * lda8 + li;
  } else {
    return;
  }
  const unsigned sh = (unsigned)(o & 3) * 8u;
  const unsigned old = atomicAdd(&adjw[o >> 2], 1u << sh);
  if (((old >> sh) & 0xffu) == 0xffu && flag) atomicOr(flag, 16);
}

// gcn_norm's self-loop bookkeeping (PyG add_remaining_self_loops) with a STATIC output shape, so that
// it can sit inside a captured stream: out = the E input edges IN PLACE followed by one loop per node.  An input edge
// that is a self loop keeps its slot with weight 0 (PyG removes it from the front part: a zero-weight term adds
// nothing to a degree and +0 to an aggregation, at the position the removed edge had), its weight moves to the
// node's loop in the tail (several loops on one node: one of them wins, as with torch's index_put_); nodes without
// one get `fill`.  Launch 1 (tail = false) writes the edge part and initialises the tail, launch 2 (tail = true) moves
// the existing loops' weights.
__global__ void k_gcn_norm_self_loops(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                      const float* __restrict__ w, int64_t E, int64_t N, float fill,
                                      int64_t* __restrict__ row_out, int64_t* __restrict__ col_out,
                                      float* __restrict__ w_out, bool tail) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (!tail) {
    if (t < E) {
      const int64_t r = row[t], c = col[t];
      row_out[t] = r;
      col_out[t] = c;
      w_out[t] = r == c ? 0.f : (w ? w[t] : 1.f);
    } else if (t < E + N) {
      row_out[t] = t - E;
      col_out[t] = t - E;
      w_out[t] = fill;
    }
  } else if (t < E) {
    const int64_t r = row[t];
    if (r == col[t] && r >= 0 && r < N) w_out[E + r] = w ? w[t] : 1.f;
  }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

int hscn_abi_version(void) { return HSCN_ABI_VERSION; }

const char* hscn_strerror(int code) {
  if (code == 0) return "ok";
  if (code == HSCN_E_BADARG) return "hscn: bad argument";
  if (code == HSCN_E_WORKSPACE) return "hscn: workspace too small";
  if (code == HSCN_E_UNSUPPORTED) return "hscn: unsupported shape";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "hscn: unknown error";
}

size_t hscn_csr_workspace_bytes(int64_t E, int64_t num_rows) {
  size_t cursor = align_up((size_t)(num_rows > 0 ? num_rows : 1) * 4, 256);
  size_t tmp = align_up((size_t)(E > 0 ? E : 1) * 4, 256);
  size_t nblk = (size_t)((num_rows + 1 + SCAN_TILE - 1) / SCAN_TILE);
  size_t bs = align_up((nblk + 1) * 4, 256);
  return cursor + tmp + bs;
}

int hscn_csr_build(const int64_t* key, const int64_t* other, int64_t E, int64_t num_rows, int64_t num_cols,
                   int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* flag, void* workspace,
                   size_t workspace_bytes, void* stream_) {
  if (E < 0 || num_rows < 0 || !rowptr || (E > 0 && (!key || !other || !col || !eid)) || !workspace)
    return HSCN_E_BADARG;
  if (E > INT32_MAX || num_rows >= INT32_MAX || num_cols > INT32_MAX) return HSCN_E_UNSUPPORTED;
  if (workspace_bytes < hscn_csr_workspace_bytes(E, num_rows)) return HSCN_E_WORKSPACE;
  hipStream_t st = hscn_stream(stream_);
  char* ws = (char*)workspace;
  size_t cursor_b = align_up((size_t)(num_rows > 0 ? num_rows : 1) * 4, 256);
  size_t tmp_b = align_up((size_t)(E > 0 ? E : 1) * 4, 256);
  int32_t* cursor = (int32_t*)ws;
  int32_t* tmp = (int32_t*)(ws + cursor_b);
  int32_t* blocksum = (int32_t*)(ws + cursor_b + tmp_b);

  hipError_t err = hipMemsetAsync(rowptr, 0, (size_t)(num_rows + 1) * 4, st);
  if (err != hipSuccess) return (int)err;
  err = hipMemsetAsync(cursor, 0, cursor_b, st);
  if (err != hipSuccess) return (int)err;
  if (E > 0) {
    k_hist<<<hscn_blocks(E, 256), 256, 0, st>>>(key, other, E, num_rows, num_cols, rowptr, flag);
    HSCN_RETURN_IF_LAUNCH_FAILED();
  }
  int64_t n1 = num_rows + 1;
  // (one block walks 4096 entries per trip, ~2.5 us each: beyond two trips the three-launch scan is faster --
  // 60 k rows of a PascalVOC-SP batch took 38 us in the single block)
  if (n1 <= 8 * 1024) {
    k_scan_single<<<1, 1024, 0, st>>>(rowptr, n1);
  } else {
    int nblk = (int)((n1 + SCAN_TILE - 1) / SCAN_TILE);
    k_scan_partial<<<nblk, SCAN_THREADS, 0, st>>>(rowptr, n1, blocksum);
    k_scan_blocksums<<<1, 1024, 0, st>>>(blocksum, nblk);
    k_scan_final<<<nblk, SCAN_THREADS, 0, st>>>(rowptr, n1, blocksum);
  }
  HSCN_RETURN_IF_LAUNCH_FAILED();
  if (E > 0) {
    k_fill<<<hscn_blocks(E, 256), 256, 0, st>>>(key, other, E, num_rows, num_cols, rowptr, cursor, tmp);
    k_rank<<<hscn_blocks(E, 256), 256, 0, st>>>(key, other, E, num_rows, num_cols, rowptr, tmp, col, eid);
    HSCN_RETURN_IF_LAUNCH_FAILED();
  }
  return 0;
}

size_t hscn_csr_pair_workspace_bytes(int64_t E, int64_t num_src, int64_t num_dst) {
  size_t b = 0;
  b += align_up((size_t)(num_dst > 0 ? num_dst : 1) * 4, 256) + align_up((size_t)(num_src > 0 ? num_src : 1) * 4, 256);
  b += 2 * align_up((size_t)(E > 0 ? E : 1) * 4, 256);
  b += align_up((size_t)((num_dst + 1 + SCAN_TILE - 1) / SCAN_TILE + 1) * 4, 256);
  b += align_up((size_t)((num_src + 1 + SCAN_TILE - 1) / SCAN_TILE + 1) * 4, 256);
  return b;
}

int hscn_csr_build_pair(const int64_t* src, const int64_t* dst, int64_t E, int64_t num_src, int64_t num_dst,
                        int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* rowptr_t, int32_t* col_t, int32_t* eid_t,
                        int32_t* flag, void* workspace, size_t workspace_bytes, void* stream_) {
  if (E < 0 || num_src < 0 || num_dst < 0 || !rowptr || !rowptr_t || !workspace ||
      (E > 0 && (!src || !dst || !col || !eid || !col_t || !eid_t)))
    return HSCN_E_BADARG;
  if (E > INT32_MAX || num_src >= INT32_MAX || num_dst >= INT32_MAX) return HSCN_E_UNSUPPORTED;
  if (workspace_bytes < hscn_csr_pair_workspace_bytes(E, num_src, num_dst)) return HSCN_E_WORKSPACE;
  hipStream_t st = hscn_stream(stream_);
  char* ws = (char*)workspace;
  auto take = [&](size_t bytes) { char* p = ws; ws += align_up(bytes, 256); return (int32_t*)p; };
  CsrPair P;
  P.s[0] = CsrSide{num_dst, rowptr, nullptr, nullptr, col, eid, nullptr};
  P.s[1] = CsrSide{num_src, rowptr_t, nullptr, nullptr, col_t, eid_t, nullptr};
  P.s[0].cursor = take((size_t)(num_dst > 0 ? num_dst : 1) * 4);
  P.s[1].cursor = take((size_t)(num_src > 0 ? num_src : 1) * 4);
  P.s[0].tmp = take((size_t)(E > 0 ? E : 1) * 4);
  P.s[1].tmp = take((size_t)(E > 0 ? E : 1) * 4);
  const int nblk0 = (int)((num_dst + 1 + SCAN_TILE - 1) / SCAN_TILE), nblk1 = (int)((num_src + 1 + SCAN_TILE - 1) / SCAN_TILE);
  P.s[0].blocksum = take((size_t)(nblk0 + 1) * 4);
  P.s[1].blocksum = take((size_t)(nblk1 + 1) * 4);
  const int64_t zt = 2 * (num_dst + num_src) + 2;
  unsigned zb = hscn_blocks(zt, 256);
  if (zb > 2048) zb = 2048;
  k_zero_pair<<<zb, 256, 0, st>>>(P);
  if (E > 0) k_hist_pair<<<hscn_blocks(E, 256), 256, 0, st>>>(src, dst, E, P, flag);
  const int nblk = nblk0 > nblk1 ? nblk0 : nblk1;
  k_scan_partial_pair<<<dim3((unsigned)nblk, 2), SCAN_THREADS, 0, st>>>(P);
  k_scan_blocksums_pair<<<2, 1024, 0, st>>>(P);
  k_scan_final_pair<<<dim3((unsigned)nblk, 2), SCAN_THREADS, 0, st>>>(P);
  if (E > 0) {
    k_fill_pair<<<hscn_blocks(E, 256), 256, 0, st>>>(src, dst, E, P);
    k_rank_pair<<<dim3(hscn_blocks(E, 256), 2), 256, 0, st>>>(src, dst, E, P);
  }
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_csr_cross_positions(const int32_t* eid, const int32_t* eid_t, int64_t E, int32_t* inv, int32_t* pos_t,
                             void* stream_) {
  if (E < 0 || (E > 0 && (!eid || !eid_t || !inv || !pos_t))) return HSCN_E_BADARG;
  if (E == 0) return 0;
  hipStream_t st = hscn_stream(stream_);
  k_inv_pos<<<hscn_blocks(E, 256), 256, 0, st>>>(eid, E, inv);
  k_pos_t<<<hscn_blocks(E, 256), 256, 0, st>>>(eid_t, inv, E, pos_t);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_gcn_dinv(const int32_t* rowptr, int64_t n, float* dinv, void* stream_) {
  if (n < 0 || (n > 0 && (!rowptr || !dinv))) return HSCN_E_BADARG;
  if (n == 0) return 0;
  k_dinv<<<hscn_blocks(n, 256), 256, 0, hscn_stream(stream_)>>>(rowptr, n, dinv);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_gcn_norm_weights(const int32_t* rowptr, const int32_t* col, const int32_t* eid, const float* w,
                          int64_t n, float* dinv, float* w_norm, void* stream_) {
  if (n < 0 || (n > 0 && (!rowptr || !col || !eid || !dinv || !w_norm))) return HSCN_E_BADARG;
  if (n == 0) return 0;
  hipStream_t st = hscn_stream(stream_);
  k_wdeg_dinv<<<hscn_blocks(n, 256), 256, 0, st>>>(rowptr, eid, w, n, dinv);
  k_wnorm<<<hscn_blocks(n, 256), 256, 0, st>>>(rowptr, col, eid, w, dinv, n, w_norm);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_assign_argmax(const float* S, int64_t* ids, int64_t n, int K, void* stream_) {
  if (n < 0 || K < 1 || (n > 0 && (!S || !ids))) return HSCN_E_BADARG;
  if (n == 0) return 0;
  k_argmax<<<hscn_blocks(n, 256), 256, 0, hscn_stream(stream_)>>>(S, ids, n, K);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_to_dense_adj(const int64_t* row, const int64_t* col, int64_t E, int64_t n, float* adj, void* stream_) {
  if (E < 0 || n < 0 || (E > 0 && (!row || !col || !adj))) return HSCN_E_BADARG;
  if (E == 0) return 0;
  k_dense_adj<<<hscn_blocks(E, 256), 256, 0, hscn_stream(stream_)>>>(row, col, E, n, adj);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_to_dense_adj_batched(const int64_t* row, const int64_t* col, int64_t E, int64_t B, int64_t n, float* adj,
                              void* stream_) {
  if (E < 0 || n < 0 || B < 0 || (E > 0 && (!row || !col || !adj))) return HSCN_E_BADARG;
  if (E == 0 || B == 0) return 0;
  k_dense_adj_batched<<<hscn_blocks(E, 256), 256, 0, hscn_stream(stream_)>>>(row, col, E, B, n, adj);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_to_dense_adj_ragged(const int64_t* row, const int64_t* col, int64_t E, const int32_t* nptr, const int32_t* gid,
                             int64_t N, int64_t B, int64_t nmax, int mode, float* adj, void* stream_) {
  if (E < 0 || N < 0 || B < 0 || nmax < 0 || (mode != 0 && mode != 1)) return HSCN_E_BADARG;
  if ((E > 0 && (!row || !col)) || !nptr || !gid || !adj) return HSCN_E_BADARG;
  const int64_t work = E + (mode == 1 ? N : 0);
  if (work == 0 || B == 0) return 0;
  k_dense_adj_ragged<<<hscn_blocks(work, 256), 256, 0, hscn_stream(stream_)>>>(row, col, E, nptr, gid, N, nmax, mode, adj);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_to_dense_adj_ragged_u8(const int64_t* row, const int64_t* col, int64_t E, const int32_t* nptr,
                                const int32_t* gid, int64_t N, int64_t B, int64_t nmax, int mode, uint8_t* adj8,
                                int32_t* flag, void* stream_) {
  if (E < 0 || N < 0 || B < 0 || nmax < 0 || (mode != 0 && mode != 1)) return HSCN_E_BADARG;
  if ((E > 0 && (!row || !col)) || !nptr || !gid || !adj8) return HSCN_E_BADARG;
  const int64_t work = E + (mode == 1 ? N : 0);
  if (work == 0 || B == 0) return 0;
  k_dense_adj_ragged_u8<<<hscn_blocks(work, 256), 256, 0, hscn_stream(stream_)>>>(
      row, col, E, nptr, gid, N, nmax, (nmax + 31) & ~(int64_t)31, mode, reinterpret_cast<unsigned int*>(adj8), flag);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_gcn_norm_self_loops(const int64_t* row, const int64_t* col, const float* w, int64_t E, int64_t N, float fill,
                             int64_t* row_out, int64_t* col_out, float* w_out, void* stream_) {
  if (E < 0 || N < 0 || (E > 0 && (!row || !col)) || !row_out || !col_out || !w_out) return HSCN_E_BADARG;
  if (E + N == 0) return 0;
  hipStream_t st = hscn_stream(stream_);
  k_gcn_norm_self_loops<<<hscn_blocks(E + N, 256), 256, 0, st>>>(row, col, w, E, N, fill, row_out, col_out, w_out, false);
  if (E > 0)
    k_gcn_norm_self_loops<<<hscn_blocks(E, 256), 256, 0, st>>>(row, col, w, E, N, fill, row_out, col_out, w_out, true);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
