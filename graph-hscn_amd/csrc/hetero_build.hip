// Cluster assignments -> heterogeneous (local / virtual) batch, on the device.
//
// Reference: graph_hscn/loader/hetero_data.py:42-87 runs a Python loop per NODE with
// `.tolist()` round trips; here one workgroup per graph does the integer bookkeeping, bit-exact:
//   * np.unique remap of the raw cluster ids to 0..U-1 (:46-51): a K-bit presence mask, the
//     remapped id is the popcount of the lower bits;
//   * virtual node v carries the float64 mean features of remapped cluster (v+1) mod U
//     (slot index clusters[ix]-1 at :53 + dropped empty slots at :55), summed in node order like
//     np.mean over the per-cluster lists, cast to float32 (:59,66);
//   * lv edges [ix, clusters[ix]] (:80-86), vv edges {(i -> j): i + j <= U-1} in the reference's
//     order (:68-79); both offset by the batch's cumulative node counts (PyG collate).
// Two launches around one host read of the totals (the outputs have data-dependent sizes, as in
// the reference): count (U per graph, remapped ids, cluster means) and emit.
#include "hscn_common.h"

namespace {

constexpr int HB_T = 256;
constexpr int HB_KMAX = 64;

template <typename XT>
__global__ void __launch_bounds__(HB_T)
k_hetero_count(const XT* __restrict__ x, const int64_t* __restrict__ clusters, const int32_t* __restrict__ nptr,
               int F, int K, int32_t* __restrict__ U_out, int32_t* __restrict__ lvl, float* __restrict__ means,
               int32_t* __restrict__ flag) {
  __shared__ unsigned long long mask_s;
  __shared__ int cnt[HB_KMAX];
  __shared__ int remap[HB_KMAX];
  const int g = blockIdx.x;
  const int n0 = nptr[g], n = nptr[g + 1] - n0;
  if (threadIdx.x == 0) mask_s = 0ull;
  for (int k = threadIdx.x; k < HB_KMAX; k += HB_T) cnt[k] = 0;
  __syncthreads();
  unsigned long long m = 0ull;
  bool bad = false;
  for (int i = threadIdx.x; i < n; i += HB_T) {
    const int64_t c = clusters[n0 + i];
    if (c < 0 || c >= K) bad = true;
    else m |= 1ull << c;
  }
  if (bad && flag) atomicOr(flag, 8);
  // wave OR, then one atomic per wave
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)(m & 0xffffffffull), o, 64);
    const unsigned hi = __shfl_xor((unsigned)(m >> 32), o, 64);
    m |= ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicOr(&mask_s, m);
  __syncthreads();
  const unsigned long long mask = mask_s;
  const int U = __popcll(mask);
  for (int k = threadIdx.x; k < K; k += HB_T) remap[k] = __popcll(mask & ((1ull << k) - 1ull));
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += HB_T) {
    const int64_t c = clusters[n0 + i];
    const int r = (c >= 0 && c < K) ? remap[c] : 0;
    lvl[n0 + i] = r;
    atomicAdd(&cnt[r], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) U_out[g] = U;
  // float64 sums in node order, one thread per (virtual slot, feature)
  for (int idx = threadIdx.x; idx < U * F; idx += HB_T) {
    const int v = idx / F, f = idx - v * F;
    const int u = (v + 1) % U;             // virtual v <- remapped cluster (v+1) mod U
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
      const int64_t c = clusters[n0 + i];
      const int r = (c >= 0 && c < K) ? remap[c] : 0;
      if (r == u) s += (double)x[(size_t)(n0 + i) * F + f];
    }
    means[((size_t)g * K + v) * F + f] = (float)(s / (double)cnt[u]);
  }
}

// vv edge e of a graph with U clusters: the reference concatenates, for i = 0..U-1,
//   source side: [i] * (U - i)      target side: range(U - i)
__global__ void __launch_bounds__(HB_T)
k_hetero_emit(const int32_t* __restrict__ U_in, const int64_t* __restrict__ vptr, const int64_t* __restrict__ evptr,
              const int32_t* __restrict__ nptr, const int32_t* __restrict__ lvl, const float* __restrict__ means,
              int F, int K, int64_t N, int64_t Evv, float* __restrict__ virtual_x, int64_t* __restrict__ ei_lv,
              int64_t* __restrict__ ei_vv) {
  const int g = blockIdx.x;
  const int n0 = nptr[g], n = nptr[g + 1] - n0;
  const int U = U_in[g];
  const int64_t v0 = vptr[g], e0 = evptr[g];
  for (int idx = threadIdx.x; idx < U * F; idx += HB_T)
    virtual_x[(size_t)v0 * F + idx] = means[(size_t)g * K * F + idx];
  for (int i = threadIdx.x; i < n; i += HB_T) {
    ei_lv[n0 + i] = n0 + i;
    ei_lv[N + n0 + i] = v0 + lvl[n0 + i];
  }
  // block i starts at i*U - i*(i-1)/2 and has U - i entries
  for (int i = threadIdx.x; i < U; i += HB_T) {
    const int start = i * U - (i * (i - 1)) / 2;
    for (int j = 0; j < U - i; ++j) {
      ei_vv[e0 + start + j] = v0 + i;
      ei_vv[Evv + e0 + start + j] = v0 + j;
    }
  }
}

}  // namespace

extern "C" {

int hscn_build_hetero_count(const void* x, int x_is_int64, const int64_t* clusters, const int32_t* nptr, int64_t B,
                            int F, int K, int32_t* U_out, int32_t* lvl, float* means, int32_t* flag,
                            void* stream_) {
  if (B < 1 || F < 1 || K < 1 || K > HB_KMAX) return HSCN_E_BADARG;
  if (!x || !clusters || !nptr || !U_out || !lvl || !means) return HSCN_E_BADARG;
  hipStream_t st = hscn_stream(stream_);
  if (x_is_int64)
    k_hetero_count<int64_t><<<(unsigned)B, HB_T, 0, st>>>((const int64_t*)x, clusters, nptr, F, K, U_out, lvl, means, flag);
  else
    k_hetero_count<float><<<(unsigned)B, HB_T, 0, st>>>((const float*)x, clusters, nptr, F, K, U_out, lvl, means, flag);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_build_hetero_emit(const int32_t* U, const int64_t* vptr, const int64_t* evptr, const int32_t* nptr,
                           const int32_t* lvl, const float* means, int64_t B, int F, int K, int64_t N, int64_t Evv,
                           float* virtual_x, int64_t* ei_lv, int64_t* ei_vv, void* stream_) {
  if (B < 1 || F < 1 || K < 1 || K > HB_KMAX || N < 0 || Evv < 0) return HSCN_E_BADARG;
  if (!U || !vptr || !evptr || !nptr || !lvl || !means || !virtual_x || !ei_lv || !ei_vv) return HSCN_E_BADARG;
  k_hetero_emit<<<(unsigned)B, HB_T, 0, hscn_stream(stream_)>>>(U, vptr, evptr, nptr, lvl, means, F, K, N, Evv,
                                                                virtual_x, ei_lv, ei_vv);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
