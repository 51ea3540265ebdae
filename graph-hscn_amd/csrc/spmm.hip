// CSR gather-reduce (SpMM) kernels: the GCNConv / GraphConv propagate step of the
// hot path (reference model/hscn.py:32,40,88-93).
//
// HBM-bound: per row one rowptr pair, its column indices, one gathered feature
// row per neighbour, one written row.  A row is owned by LPR = width/VEC
// consecutive lanes (VEC=4: 16-byte loads/stores), so a wave covers 64/LPR rows
// and every global access of a lane group is one contiguous row segment.  Rows
// accumulate in CSR slot order with separately rounded multiply and add: the
// order and rounding of the CPU reference's gather -> scale -> index_add_.
#include "hscn_common.h"
#include <cstdlib>

namespace {

constexpr int SP_THREADS = 256;

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  using T = float4;
  static __device__ __forceinline__ T load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void store(float* p, T v) { *reinterpret_cast<float4*>(p) = v; }
  static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ T axpy(float w, T x, T a) {
    return make_float4(add_rn(a.x, mul_rn(w, x.x)), add_rn(a.y, mul_rn(w, x.y)),
                       add_rn(a.z, mul_rn(w, x.z)), add_rn(a.w, mul_rn(w, x.w)));
  }
  static __device__ __forceinline__ T add(T a, T b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
  static __device__ __forceinline__ T act(T a, int k) {
    return make_float4(apply_act(a.x, k), apply_act(a.y, k), apply_act(a.z, k), apply_act(a.w, k));
  }
};
template <>
struct Vec<1> {
  using T = float;
  static __device__ __forceinline__ T load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, T v) { *p = v; }
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ T axpy(float w, T x, T a) { return add_rn(a, mul_rn(w, x)); }
  static __device__ __forceinline__ T add(T a, T b) { return a + b; }
  static __device__ __forceinline__ T act(T a, int k) { return apply_act(a, k); }
};

// MODE 0: w = dinv_c[col]*dinv_r[row] (GCN)   MODE 1: w = wts[eid ? eid[p] : p] or 1
// A lane owns NV consecutive VEC-wide pieces of its row (NV=2 at width >= 64: twice the bytes in
// flight per lane -- the gather is latency-bound, three dependent loads deep: rowptr -> col -> h).
template <int VEC, int MODE, int NV>
__global__ void __launch_bounds__(SP_THREADS)
k_spmm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
       const float* __restrict__ dinv_r, const float* __restrict__ dinv_c, const float* __restrict__ wts,
       const float* __restrict__ h, const float* __restrict__ bias, float* __restrict__ out,
       int64_t num_rows, int width, int LPR, int RPB, int accumulate, int act, int64_t rows_per_block,
       int xcd_map) {
  using V = Vec<VEC>;
  const int rl = threadIdx.x / LPR;
  const int f = (threadIdx.x - rl * LPR) * VEC * NV;
  if (rl >= RPB) return;
  // rows_per_block > 0: every block owns one contiguous row range; with xcd_map the ranges of the
  // blocks that share an XCD (blockIdx % 8, round-robin dispatch) are adjacent, so a gathered
  // neighbour row is usually in that XCD's L2.  rows_per_block == 0: grid-stride over rows.
  int64_t r_begin, r_end, r_step;
  if (rows_per_block > 0) {
    const int64_t nb = gridDim.x;
    const int64_t vb = (xcd_map && nb % 8 == 0) ? ((int64_t)(blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8)
                                                 : (int64_t)blockIdx.x;
    r_begin = vb * rows_per_block + rl;
    r_end = vb * rows_per_block + rows_per_block;
    if (r_end > num_rows) r_end = num_rows;
    r_step = RPB;
  } else {
    r_begin = (int64_t)blockIdx.x * RPB + rl;
    r_end = num_rows;
    r_step = (int64_t)gridDim.x * RPB;
  }
  for (int64_t r = r_begin; r < r_end; r += r_step) {
    const int s = rowptr[r], t = rowptr[r + 1];
    typename V::T acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = V::zero();
    float dr = 0.f;
    if (MODE == 0) dr = dinv_r[r];
#pragma unroll 2
    for (int p = s; p < t; ++p) {
      const int j = col[p];
      float w;
      if (MODE == 0) w = mul_rn(dinv_c[j], dr);
      else w = wts ? wts[eid ? eid[p] : p] : 1.0f;
      const float* hj = h + (size_t)j * width + f;
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = V::axpy(w, V::load(hj + q * VEC), acc[q]);
    }
    float* o = out + (size_t)r * width + f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      typename V::T a = acc[q];
      if (bias) a = V::add(a, V::load(bias + f + q * VEC));
      if (accumulate) a = V::add(a, V::load(o + q * VEC));
      V::store(o + q * VEC, V::act(a, act));
    }
  }
}

// The same gather-reduce with the INDEX CHAIN TAKEN OFF THE CRITICAL PATH (VEC = 4).  k_spmm walks a row through
// three dependent loads -- rowptr -> col -> h -- so a lane group has data in flight for a third of the time.  Here
// a lane group carries its current row's first four column indices in registers and, while that row's (up to four)
// neighbour rows are in flight, fetches the NEXT row's rowptr pair and then its first four columns: per row one
// exposed latency instead of three.  Loads of absent edges are predicated off, not clamped (a row has ~2 edges: clamped
// slots would double the gather traffic).  Rows longer than four edges finish in the plain loop.  Accumulation is in
// CSR slot order with separately rounded multiply and add, exactly as k_spmm: bit-identical results.
template <int MODE, int NV>
__global__ void __launch_bounds__(SP_THREADS)
k_spmm_pipe(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
            const float* __restrict__ dinv_r, const float* __restrict__ dinv_c, const float* __restrict__ wts,
            const float* __restrict__ h, const float* __restrict__ bias, float* __restrict__ out,
            int64_t num_rows, int width, int LPR, int RPB, int accumulate, int act, int nt_store) {
  using V = Vec<4>;
  constexpr int D = 4;
  const int rl = threadIdx.x / LPR;
  const int f = (threadIdx.x - rl * LPR) * 4 * NV;
  if (rl >= RPB) return;
  int64_t r = (int64_t)blockIdx.x * RPB + rl;
  const int64_t step = (int64_t)gridDim.x * RPB;
  if (r >= num_rows) return;
  int s = rowptr[r], t = rowptr[r + 1];
  int jc[D];
#pragma unroll
  for (int u = 0; u < D; ++u) { jc[u] = 0; if (s + u < t) jc[u] = col[s + u]; }
  float4 bq[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) bq[q] = bias ? V::load(bias + f + q * 4) : V::zero();
  for (;;) {
    const int64_t rn = r + step;
    const bool more = rn < num_rows;
    int sn = 0, tn = 0;
    if (more) { sn = rowptr[rn]; tn = rowptr[rn + 1]; }
    float dr = 0.f;
    if (MODE == 0) dr = dinv_r[r];
    float w[D];
    float4 x[D][NV];
#pragma unroll
    for (int u = 0; u < D; ++u) {
      w[u] = 0.f;
#pragma unroll
      for (int q = 0; q < NV; ++q) x[u][q] = V::zero();
      if (s + u < t) {
        const int j = jc[u];
        if (MODE == 0) w[u] = dinv_c[j];
        else w[u] = wts ? wts[eid ? eid[s + u] : s + u] : 1.0f;
        const float* hj = h + (size_t)j * width + f;
#pragma unroll
        for (int q = 0; q < NV; ++q) x[u][q] = V::load(hj + q * 4);
      }
    }
    int jn[D];
#pragma unroll
    for (int u = 0; u < D; ++u) { jn[u] = 0; if (more && sn + u < tn) jn[u] = col[sn + u]; }
    float4 acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = V::zero();
#pragma unroll
    for (int u = 0; u < D; ++u) {
      if (s + u < t) {
        const float ww = MODE == 0 ? mul_rn(w[u], dr) : w[u];
#pragma unroll
        for (int q = 0; q < NV; ++q) acc[q] = V::axpy(ww, x[u][q], acc[q]);
      }
    }
    for (int p = s + D; p < t; ++p) {      // (rows of more than four edges)
      const int j = col[p];
      float ww;
      if (MODE == 0) ww = mul_rn(dinv_c[j], dr);
      else ww = wts ? wts[eid ? eid[p] : p] : 1.0f;
      const float* hj = h + (size_t)j * width + f;
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = V::axpy(ww, V::load(hj + q * 4), acc[q]);
    }
    float* o = out + (size_t)r * width + f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      float4 a = acc[q];
      if (bias) a = V::add(a, bq[q]);
      if (accumulate) a = V::add(a, V::load(o + q * 4));
      a = V::act(a, act);
      if (nt_store) {
        __builtin_nontemporal_store(a.x, o + q * 4 + 0); __builtin_nontemporal_store(a.y, o + q * 4 + 1);
        __builtin_nontemporal_store(a.z, o + q * 4 + 2); __builtin_nontemporal_store(a.w, o + q * 4 + 3);
      } else {
        V::store(o + q * 4, a);
      }
    }
    if (!more) break;
    r = rn; s = sn; t = tn;
#pragma unroll
    for (int u = 0; u < D; ++u) jc[u] = jn[u];
  }
}

template <int MODE>
int launch_spmm(const int32_t* rowptr, const int32_t* col, const int32_t* eid, const float* dinv_r,
                const float* dinv_c, const float* wts, const float* h, const float* bias, float* out,
                int64_t num_rows, int width, int accumulate, int act, hipStream_t st) {
  static const int xcd_map = getenv("HSCN_SPMM_XCD") ? atoi(getenv("HSCN_SPMM_XCD")) : 0;
  static const int passes = getenv("HSCN_SPMM_PASSES") ? atoi(getenv("HSCN_SPMM_PASSES")) : 0;  // 0 = grid-stride
  static const int nv_env = getenv("HSCN_SPMM_NV") ? atoi(getenv("HSCN_SPMM_NV")) : 0;
  const int VEC = (width % 4 == 0) ? 4 : 1;
  // measured on MI355X (tools/bench_spmm.py, 4096 Peptides graphs): NV=2 lifts H=128 from 3.0 to
  // 4.1 TB/s; NV=4 falls back to 3.2 TB/s; contiguous / XCD-grouped row ranges do not beat grid-stride
  int NV = (VEC == 4 && width % 8 == 0 && width >= 64) ? 2 : 1;
  if (nv_env == 4 && width % 16 == 0 && width >= 128) NV = 4;
  else if (nv_env == 1) NV = 1;
  const int LPR = width / (VEC * NV);
  if (LPR > SP_THREADS) return HSCN_E_UNSUPPORTED;
  const int RPB = SP_THREADS / LPR;
  static const int pipe = getenv("HSCN_SPMM_PIPE") ? atoi(getenv("HSCN_SPMM_PIPE")) : 1;
  static const int nt = getenv("HSCN_SPMM_NT") ? atoi(getenv("HSCN_SPMM_NT")) : 0;
  static const int nbmax = getenv("HSCN_SPMM_BLOCKS") ? atoi(getenv("HSCN_SPMM_BLOCKS")) : 8192;
  // measured (tools/ab_spmm.sh, 658 k rows): the pipelined form wins where a row is a few lanes -- H = 16: 3.55 -> 4.17
  // TB/s -- and loses where a lane group already keeps 1 KB in flight and registers decide the occupancy -- H = 128:
  // 4.42 -> 2.95 TB/s (4.05 with one piece per lane); HSCN_SPMM_PIPE=2 forces it at every width
  if ((pipe == 2 || (pipe == 1 && width <= 32)) && VEC == 4 && NV <= 2 && passes == 0) {
    int64_t nbp = (num_rows + RPB - 1) / RPB;
    if (nbp > nbmax) nbp = nbmax;
    if (NV == 2)
      k_spmm_pipe<MODE, 2><<<(unsigned)nbp, SP_THREADS, 0, st>>>(rowptr, col, eid, dinv_r, dinv_c, wts, h, bias, out, num_rows,
                                                               width, LPR, RPB, accumulate, act, nt);
    else
      k_spmm_pipe<MODE, 1><<<(unsigned)nbp, SP_THREADS, 0, st>>>(rowptr, col, eid, dinv_r, dinv_c, wts, h, bias, out, num_rows,
                                                               width, LPR, RPB, accumulate, act, nt);
    HSCN_RETURN_IF_LAUNCH_FAILED();
    return 0;
  }
  int64_t nb, rpb = 0;
  if (passes > 0) {
    nb = (num_rows + (int64_t)RPB * passes - 1) / ((int64_t)RPB * passes);
    if (nb > 65536) nb = 65536;
    nb = (nb + 7) / 8 * 8;
    rpb = (num_rows + nb - 1) / nb;
    rpb = (rpb + RPB - 1) / RPB * RPB;
  } else {
    nb = (num_rows + RPB - 1) / RPB;
    if (nb > 8192) nb = 8192;
  }
#define HSCN_SPMM_LAUNCH(V_, N_)                                                                              \
  k_spmm<V_, MODE, N_><<<(unsigned)nb, SP_THREADS, 0, st>>>(rowptr, col, eid, dinv_r, dinv_c, wts, h, bias, out, \
                                                            num_rows, width, LPR, RPB, accumulate, act, rpb, xcd_map)
  if (VEC == 4 && NV == 4) HSCN_SPMM_LAUNCH(4, 4);
  else if (VEC == 4 && NV == 2) HSCN_SPMM_LAUNCH(4, 2);
  else if (VEC == 4) HSCN_SPMM_LAUNCH(4, 1);
  else HSCN_SPMM_LAUNCH(1, 1);
#undef HSCN_SPMM_LAUNCH
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // namespace

extern "C" {

int hscn_spmm_csr_gcn(const int32_t* rowptr, const int32_t* col, const float* dinv_r, const float* dinv_c,
                      const float* h, const float* bias, float* out, int64_t num_rows, int width,
                      int accumulate, int act, void* stream_) {
  if (num_rows < 0 || width < 1) return HSCN_E_BADARG;
  if (num_rows == 0) return 0;
  if (!rowptr || !col || !dinv_r || !dinv_c || !h || !out) return HSCN_E_BADARG;
  return launch_spmm<0>(rowptr, col, nullptr, dinv_r, dinv_c, nullptr, h, bias, out, num_rows, width,
                        accumulate, act, hscn_stream(stream_));
}

int hscn_spmm_csr_weighted(const int32_t* rowptr, const int32_t* col, const int32_t* eid, const float* w,
                           const float* x, float* out, int64_t num_rows, int width, void* stream_) {
  if (num_rows < 0 || width < 1) return HSCN_E_BADARG;
  if (num_rows == 0) return 0;
  if (!rowptr || !col || !x || !out) return HSCN_E_BADARG;
  return launch_spmm<1>(rowptr, col, eid, nullptr, nullptr, w, x, nullptr, out, num_rows, width, 0,
                        HSCN_ACT_IDENTITY, hscn_stream(stream_));
}

}  // extern "C"
