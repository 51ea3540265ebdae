// dense_mincut_pool on the edge-list route (reference model/hscn.py:61-63;
// SURVEY.md A.4).  A is never densified:
//   tr(S^T A S) = sum_i S_i . (A S)_i,  (A S)_i = sum_{p in row i} S[col[p]]
//   tr(S^T D S) = sum_i d_i |S_i|^2,    d_i = |row i|
// One workgroup per graph: node tiles of S / (A S) / X are staged in LDS and the
// K x K (and K x Fx) contractions accumulate in LDS cells each owned by one
// thread, so all sums are ordered (node order) and reproducible.
#include "hscn_common.h"

namespace {

constexpr int MC_THREADS = 256;
constexpr int MC_T = 16;  // nodes per LDS tile

__global__ void k_softmax_rows(const float* __restrict__ logits, float* __restrict__ S, int64_t n, int K) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = logits + i * K;
  float m = r[0];
  for (int k = 1; k < K; ++k) m = fmaxf(m, r[k]);
  float sum = 0.f;
  for (int k = 0; k < K; ++k) sum += expf(r[k] - m);
  for (int k = 0; k < K; ++k) S[i * K + k] = expf(r[k] - m) / sum;
}

// wave-0 ordered reduction of f(idx) over idx in [0,n)
template <typename F>
__device__ __forceinline__ float wave0_reduce(int n, F f) {
  float v = 0.f;
  for (int idx = threadIdx.x; idx < n; idx += 64) v += f(idx);
  return wave_sum(v);
}

__global__ void __launch_bounds__(MC_THREADS)
k_mincut_stats(const float* __restrict__ S, const float* __restrict__ x, const int32_t* __restrict__ rowptr,
               const int32_t* __restrict__ col, const int32_t* __restrict__ node_ptr,
               float* __restrict__ stats, float* __restrict__ ss_out, float* __restrict__ pooled_x,
               float* __restrict__ pooled_adj, int K, int Fx) {
  extern __shared__ __align__(16) float lds[];
  const int KK = K * K;
  float* acc_ss = lds;                 // [K][K]
  float* acc_oa = acc_ss + KK;         // [K][K]
  float* acc_px = acc_oa + KK;         // [K][Fx]
  float* S_t = acc_px + K * Fx;        // [T][K]
  float* AS_t = S_t + MC_T * K;        // [T][K]
  float* x_t = AS_t + MC_T * K;        // [T][Fx]
  float* d_t = x_t + MC_T * Fx;        // [T]
  float* den_a = d_t + MC_T;           // [K]
  float* dn = den_a + K;               // [K]
  float* red = dn + K;                 // [4]

  const int g = blockIdx.x;
  const int n0 = node_ptr[g], n1 = node_ptr[g + 1];
  for (int idx = threadIdx.x; idx < 2 * KK + K * Fx; idx += MC_THREADS) lds[idx] = 0.f;
  for (int idx = threadIdx.x; idx < K; idx += MC_THREADS) den_a[idx] = 0.f;

  for (int base = n0; base < n1; base += MC_T) {
    const int nt = (n1 - base) < MC_T ? (n1 - base) : MC_T;
    __syncthreads();
    for (int idx = threadIdx.x; idx < nt * K; idx += MC_THREADS) {
      const int t = idx / K, k = idx - t * K;
      const int i = base + t;
      S_t[idx] = S[(size_t)i * K + k];
      float a = 0.f;
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) a += S[(size_t)col[p] * K + k];
      AS_t[idx] = a;
    }
    if (x)
      for (int idx = threadIdx.x; idx < nt * Fx; idx += MC_THREADS) x_t[idx] = x[(size_t)base * Fx + idx];
    for (int t = threadIdx.x; t < nt; t += MC_THREADS)
      d_t[t] = (float)(rowptr[base + t + 1] - rowptr[base + t]);
    __syncthreads();
    for (int idx = threadIdx.x; idx < KK; idx += MC_THREADS) {
      const int a = idx / K, b = idx - a * K;
      float s1 = acc_ss[idx], s2 = acc_oa[idx];
      for (int t = 0; t < nt; ++t) {
        const float sa = S_t[t * K + a];
        s1 = fmaf(sa, S_t[t * K + b], s1);
        s2 = fmaf(sa, AS_t[t * K + b], s2);
      }
      acc_ss[idx] = s1;
      acc_oa[idx] = s2;
    }
    if (x)
      for (int idx = threadIdx.x; idx < K * Fx; idx += MC_THREADS) {
        const int a = idx / Fx, f = idx - a * Fx;
        float s = acc_px[idx];
        for (int t = 0; t < nt; ++t) s = fmaf(S_t[t * K + a], x_t[t * Fx + f], s);
        acc_px[idx] = s;
      }
    for (int a = threadIdx.x; a < K; a += MC_THREADS) {
      float s = den_a[a];
      for (int t = 0; t < nt; ++t) {
        const float sa = S_t[t * K + a];
        s = fmaf(d_t[t], sa * sa, s);
      }
      den_a[a] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const float num = wave0_reduce(K, [&](int a) { return acc_oa[a * K + a]; });
    const float den = wave0_reduce(K, [&](int a) { return den_a[a]; });
    const float nrm2 = wave0_reduce(KK, [&](int i) { return acc_ss[i] * acc_ss[i]; });
    const float nrm = sqrtf(nrm2);
    const float isk = 1.0f / sqrtf((float)K);
    const float o2 = wave0_reduce(KK, [&](int i) {
      const int a = i / K, b = i - a * K;
      const float q = acc_ss[i] / nrm - (a == b ? isk : 0.f);
      return q * q;
    });
    if (threadIdx.x == 0) {
      stats[g * 4 + 0] = num;
      stats[g * 4 + 1] = den;
      stats[g * 4 + 2] = nrm;
      stats[g * 4 + 3] = sqrtf(o2);
    }
  }
  for (int idx = threadIdx.x; idx < KK; idx += MC_THREADS) ss_out[(size_t)g * KK + idx] = acc_ss[idx];
  if (pooled_x && x)
    for (int idx = threadIdx.x; idx < K * Fx; idx += MC_THREADS) pooled_x[(size_t)g * K * Fx + idx] = acc_px[idx];
  if (pooled_adj) {
    for (int a = threadIdx.x; a < K; a += MC_THREADS) {
      float s = 0.f;
      for (int b = 0; b < K; ++b) s += (a == b) ? 0.f : acc_oa[a * K + b];
      dn[a] = sqrtf(s) + 1e-15f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < KK; idx += MC_THREADS) {
      const int a = idx / K, b = idx - a * K;
      pooled_adj[(size_t)g * KK + idx] = (a == b) ? 0.f : (acc_oa[idx] / dn[b]) / dn[a];
    }
  }
  (void)red;
}

__global__ void k_mincut_losses(const float* __restrict__ stats, float* __restrict__ losses, int G) {
  // single wave, ordered
  float mc = 0.f, o = 0.f;
  for (int g = threadIdx.x; g < G; g += 64) {
    mc += -(stats[g * 4 + 0] / stats[g * 4 + 1]);
    o += stats[g * 4 + 3];
  }
  mc = wave_sum(mc);
  o = wave_sum(o);
  if (threadIdx.x == 0) {
    losses[0] = mc / (float)G;
    losses[1] = o / (float)G;
  }
}

__global__ void __launch_bounds__(MC_THREADS)
k_mincut_bwd(const float* __restrict__ S, const float* __restrict__ stats, const float* __restrict__ ss,
             const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
             const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ col_t,
             const int32_t* __restrict__ node_ptr, const float* __restrict__ g_losses,
             float* __restrict__ g_logits, int K, int G) {
  extern __shared__ __align__(16) float lds[];
  const int KK = K * K;
  float* Gss = lds;            // [K][K]  d ortho / d (S^T S)
  float* S_t = Gss + KK;       // [T][K]
  float* dS_t = S_t + MC_T * K;  // [T][K]
  float* red = dS_t + MC_T * K;  // [2]
  const int g = blockIdx.x;
  const int n0 = node_ptr[g], n1 = node_ptr[g + 1];
  const float num = stats[g * 4 + 0], den = stats[g * 4 + 1], nrm = stats[g * 4 + 2], o = stats[g * 4 + 3];
  const float gmc = g_losses[0] / (float)G, go = g_losses[1] / (float)G;
  const float isk = 1.0f / sqrtf((float)K);
  const float* ssg = ss + (size_t)g * KK;
  // inner = <Gq, ss>,  Gq = (ss/nrm - I/sqrt(K)) / o
  if (threadIdx.x < 64) {
    float v = 0.f;
    if (o > 0.f)
      for (int i = threadIdx.x; i < KK; i += 64) {
        const int a = i / K, b = i - a * K;
        v += ((ssg[i] / nrm - (a == b ? isk : 0.f)) / o) * ssg[i];
      }
    v = wave_sum(v);
    if (threadIdx.x == 0) red[0] = v;
  }
  __syncthreads();
  const float inner = red[0];
  for (int i = threadIdx.x; i < KK; i += MC_THREADS) {
    const int a = i / K, b = i - a * K;
    const float gq = o > 0.f ? (ssg[i] / nrm - (a == b ? isk : 0.f)) / o : 0.f;
    Gss[i] = (gq - inner / (nrm * nrm) * ssg[i]) / nrm;
  }
  const float c_num = -gmc / den;
  const float c_den = gmc * num / (den * den);
  for (int base = n0; base < n1; base += MC_T) {
    const int nt = (n1 - base) < MC_T ? (n1 - base) : MC_T;
    __syncthreads();
    for (int idx = threadIdx.x; idx < nt * K; idx += MC_THREADS) S_t[idx] = S[(size_t)base * K + idx];
    __syncthreads();
    for (int idx = threadIdx.x; idx < nt * K; idx += MC_THREADS) {
      const int t = idx / K, k = idx - t * K;
      const int i = base + t;
      float as = 0.f;
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) as += S[(size_t)col[p] * K + k];
      for (int q = rowptr_t[i]; q < rowptr_t[i + 1]; ++q) as += S[(size_t)col_t[q] * K + k];
      const float d_i = (float)(rowptr[i + 1] - rowptr[i]);
      float orth = 0.f;
      for (int a = 0; a < K; ++a) orth = fmaf(S_t[t * K + a], Gss[a * K + k], orth);
      // Gss is symmetric: d o / dS = S (Gss + Gss^T) = 2 S Gss
      dS_t[idx] = c_num * as + c_den * 2.f * d_i * S_t[idx] + go * 2.f * orth;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nt * K; idx += MC_THREADS) {
      const int t = idx / K;
      float dot = 0.f;
      for (int a = 0; a < K; ++a) dot = fmaf(dS_t[t * K + a], S_t[t * K + a], dot);
      g_logits[(size_t)base * K + idx] = S_t[idx] * (dS_t[idx] - dot);
    }
  }
}

}  // namespace

extern "C" {

int hscn_mincut_sparse_fwd(const float* logits, const float* x, const int32_t* rowptr, const int32_t* col,
                           const int32_t* node_ptr, float* S, float* stats, float* ss, float* pooled_x,
                           float* pooled_adj, float* losses, int64_t num_nodes, int64_t num_graphs, int K,
                           int Fx, void* stream_) {
  if (num_nodes < 0 || num_graphs < 1 || K < 1 || K > 256 || Fx < 0) return HSCN_E_BADARG;
  if (!logits || !rowptr || !col || !node_ptr || !S || !stats || !ss || !losses) return HSCN_E_BADARG;
  if (!x) Fx = 0;
  hipStream_t st = hscn_stream(stream_);
  if (num_nodes > 0) {
    k_softmax_rows<<<hscn_blocks(num_nodes, 256), 256, 0, st>>>(logits, S, num_nodes, K);
    HSCN_RETURN_IF_LAUNCH_FAILED();
  }
  size_t lds = (size_t)(2 * K * K + K * Fx + 2 * MC_T * K + MC_T * Fx + MC_T + 2 * K + 4) * 4;
  if (lds > 160 * 1024) return HSCN_E_UNSUPPORTED;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k_mincut_stats, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_mincut_stats<<<(unsigned)num_graphs, MC_THREADS, lds, st>>>(S, x, rowptr, col, node_ptr, stats, ss,
                                                                pooled_x, pooled_adj, K, Fx);
  k_mincut_losses<<<1, 64, 0, st>>>(stats, losses, (int)num_graphs);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_mincut_sparse_bwd(const float* S, const float* stats, const float* ss, const int32_t* rowptr,
                           const int32_t* col, const int32_t* rowptr_t, const int32_t* col_t,
                           const int32_t* node_ptr, const float* g_losses, float* g_logits, int64_t num_nodes,
                           int64_t num_graphs, int K, void* stream_) {
  if (num_nodes < 0 || num_graphs < 1 || K < 1 || K > 256) return HSCN_E_BADARG;
  if (!S || !stats || !ss || !rowptr || !col || !rowptr_t || !col_t || !node_ptr || !g_losses || !g_logits)
    return HSCN_E_BADARG;
  size_t lds = (size_t)(K * K + 2 * MC_T * K + 4) * 4;
  if (lds > 160 * 1024) return HSCN_E_UNSUPPORTED;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k_mincut_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_mincut_bwd<<<(unsigned)num_graphs, MC_THREADS, lds, hscn_stream(stream_)>>>(
      S, stats, ss, rowptr, col, rowptr_t, col_t, node_ptr, g_losses, g_logits, K, (int)num_graphs);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
