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
constexpr int HB_FMAX = 16;                            // feature columns (9 atom / 14 superpixel features)
constexpr int HB_CH = 128;                             // nodes staged per chunk
constexpr int HB_VFPT = HB_KMAX * HB_FMAX / HB_T;      // (virtual slot, feature) pairs per thread

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
  // float64 sums in node order, one thread per (virtual slot, feature) -- the order is what makes the
  // means bit-equal to np.mean over the per-cluster lists, so the walk over the nodes stays serial; the
  // nodes are staged through LDS in chunks (coalesced loads by all threads), the serial walk reads LDS
  __shared__ double xs[HB_CH * HB_FMAX];
  __shared__ int rs[HB_CH];
  const int nvf = U * F;
  double acc[HB_VFPT];
#pragma unroll
  for (int q = 0; q < HB_VFPT; ++q) acc[q] = 0.0;
  for (int c0 = 0; c0 < n; c0 += HB_CH) {
    const int cn = (n - c0) < HB_CH ? (n - c0) : HB_CH;
    __syncthreads();
    for (int i = threadIdx.x; i < cn; i += HB_T) rs[i] = lvl[n0 + c0 + i];   // written above by this workgroup
    for (int idx = threadIdx.x; idx < cn * F; idx += HB_T)
      xs[idx] = (double)x[(size_t)(n0 + c0) * F + idx];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < HB_VFPT; ++q) {
      const int idx = threadIdx.x + q * HB_T;
      if (idx < nvf) {
        const int v = idx / F, f = idx - v * F;
        const int u = (v + 1) % U;             // virtual v <- remapped cluster (v+1) mod U
        double s = acc[q];
        for (int i = 0; i < cn; ++i)
          if (rs[i] == u) s += xs[i * F + f];
        acc[q] = s;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < HB_VFPT; ++q) {
    const int idx = threadIdx.x + q * HB_T;
    if (idx < nvf) {
      const int v = idx / F, f = idx - v * F;
      const int u = (v + 1) % U;
      means[((size_t)g * K + v) * F + f] = (float)(acc[q] / (double)cnt[u]);
    }
  }
}

// offsets of the collated batch from the per-graph cluster counts, one workgroup: vptr[g] = sum_{h<g} U_h,
// evptr[g] = sum_{h<g} U_h (U_h + 1) / 2 (int64 for the PyG-style ptr, int32 for the resident kernels), and
// the four numbers the host needs to size the outputs: {V, E_vv, flag, max U}.  Serial chunks of 256 graphs
// with a running carry: B is a few hundred.
__global__ void __launch_bounds__(HB_T)
k_hetero_scan(const int32_t* __restrict__ U, int B, const int32_t* __restrict__ flag, int64_t* __restrict__ vptr,
              int64_t* __restrict__ evptr, int32_t* __restrict__ vptr32, int32_t* __restrict__ evptr32,
              int64_t* __restrict__ totals) {
  __shared__ long long sv[HB_T], se[HB_T];
  __shared__ int smax[HB_T];
  long long cv = 0, ce = 0;
  int mx = 0;
  for (int base = 0; base < B; base += HB_T) {
    const int g = base + threadIdx.x;
    const long long u = g < B ? U[g] : 0;
    sv[threadIdx.x] = u;
    se[threadIdx.x] = u * (u + 1) / 2;
    smax[threadIdx.x] = (int)u;
    __syncthreads();
    if (threadIdx.x == 0) {   // B is small: a serial pass keeps the order obvious
      long long a = cv, b = ce;
      for (int i = 0; i < HB_T && base + i < B; ++i) {
        vptr[base + i] = a; evptr[base + i] = b;
        vptr32[base + i] = (int32_t)a; evptr32[base + i] = (int32_t)b;
        a += sv[i]; b += se[i];
        if (smax[i] > mx) mx = smax[i];
      }
      cv = a; ce = b;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    vptr[B] = cv; evptr[B] = ce;
    vptr32[B] = (int32_t)cv; evptr32[B] = (int32_t)ce;
    totals[0] = cv; totals[1] = ce; totals[2] = flag ? flag[0] : 0; totals[3] = mx;
  }
}

// vv edge e of a graph with U clusters: the reference concatenates, for i = 0..U-1,
//   source side: [i] * (U - i)      target side: range(U - i)
__global__ void __launch_bounds__(HB_T)
k_hetero_emit(const int32_t* __restrict__ U_in, const int64_t* __restrict__ vptr, const int64_t* __restrict__ evptr,
              const int32_t* __restrict__ nptr, const int32_t* __restrict__ lvl, const float* __restrict__ means,
              int F, int K, int64_t N, int64_t Evv, float* __restrict__ virtual_x, int64_t* __restrict__ ei_lv,
              int64_t* __restrict__ ei_vv, int64_t* __restrict__ vbatch) {
  const int g = blockIdx.x;
  const int n0 = nptr[g], n = nptr[g + 1] - n0;
  const int U = U_in[g];
  const int64_t v0 = vptr[g], e0 = evptr[g];
  for (int idx = threadIdx.x; idx < U * F; idx += HB_T)
    virtual_x[(size_t)v0 * F + idx] = means[(size_t)g * K * F + idx];
  if (vbatch)
    for (int v = threadIdx.x; v < U; v += HB_T) vbatch[v0 + v] = g;   // graph id of every virtual node
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
  if (F > HB_FMAX) return HSCN_E_UNSUPPORTED;
  if (!x || !clusters || !nptr || !U_out || !lvl || !means) return HSCN_E_BADARG;
  hipStream_t st = hscn_stream(stream_);
  if (x_is_int64)
    k_hetero_count<int64_t><<<(unsigned)B, HB_T, 0, st>>>((const int64_t*)x, clusters, nptr, F, K, U_out, lvl, means, flag);
  else
    k_hetero_count<float><<<(unsigned)B, HB_T, 0, st>>>((const float*)x, clusters, nptr, F, K, U_out, lvl, means, flag);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_build_hetero_scan(const int32_t* U, int64_t B, const int32_t* flag, int64_t* vptr, int64_t* evptr,
                           int32_t* vptr32, int32_t* evptr32, int64_t* totals, void* stream_) {
  if (B < 1 || !U || !vptr || !evptr || !vptr32 || !evptr32 || !totals) return HSCN_E_BADARG;
  k_hetero_scan<<<1, HB_T, 0, hscn_stream(stream_)>>>(U, (int)B, flag, vptr, evptr, vptr32, evptr32, totals);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int hscn_build_hetero_emit(const int32_t* U, const int64_t* vptr, const int64_t* evptr, const int32_t* nptr,
                           const int32_t* lvl, const float* means, int64_t B, int F, int K, int64_t N, int64_t Evv,
                           float* virtual_x, int64_t* ei_lv, int64_t* ei_vv, int64_t* vbatch, void* stream_) {
  if (B < 1 || F < 1 || K < 1 || K > HB_KMAX || N < 0 || Evv < 0) return HSCN_E_BADARG;
  if (!U || !vptr || !evptr || !nptr || !lvl || !means || !virtual_x || !ei_lv || !ei_vv) return HSCN_E_BADARG;
  k_hetero_emit<<<(unsigned)B, HB_T, 0, hscn_stream(stream_)>>>(U, vptr, evptr, nptr, lvl, means, F, K, N, Evv,
                                                                virtual_x, ei_lv, ei_vv, vbatch);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
