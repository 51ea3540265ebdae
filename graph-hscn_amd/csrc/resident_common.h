// Device helpers shared by the graph-resident kernels (resident.hip: HSCN, resident_scn.hip: SCN):
// wave groups, LDS scans, stable CSR construction in LDS, DPP row sums, the per-graph
// parameter-gradient reduction, and the diagnostic phase stamps.
#pragma once
#include "hscn_common.h"

namespace {

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// ---- storage type of node features and inter-layer activations in HBM: float, or IEEE half (BASELINE.json
// configs[4]: "fp16 feat").  Arithmetic is float either way: values are widened on load, every sum (gather-reduce,
// MFMA, pool, gradient partials) accumulates in float registers, and an activation is rounded to the storage type
// ONCE, where it is produced -- the copy that stays in LDS for the next layer holds the same rounded value as the
// copy that goes to HBM, so forward and backward see one and the same activation.
typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ldf(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const half_t* p, size_t i) { return (float)p[i]; }
__device__ __forceinline__ float4 ldf4(const float* p, size_t i4) { return reinterpret_cast<const float4*>(p)[i4]; }
__device__ __forceinline__ float4 ldf4(const half_t* p, size_t i4) {
  const half4_t v = reinterpret_cast<const half4_t*>(p)[i4];
  return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
__device__ __forceinline__ void stf(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void stf(half_t* p, size_t i, float v) { p[i] = (half_t)v; }
__device__ __forceinline__ void stf4(float* p, size_t i4, float4 v) { reinterpret_cast<float4*>(p)[i4] = v; }
__device__ __forceinline__ void stf4(half_t* p, size_t i4, float4 v) {
  half4_t h;
  h.x = (half_t)v.x; h.y = (half_t)v.y; h.z = (half_t)v.z; h.w = (half_t)v.w;
  reinterpret_cast<half4_t*>(p)[i4] = h;
}
// write-through (sc1) forms for bytes that another workgroup of the SAME launch reads (resident_step.h): every store
// of such bytes is one of these, every load of them one of the sc1 buffer loads below (cdna_hip_programming.md
// Guideline 16: with both, the consumer needs no agent-scope acquire)
__device__ __forceinline__ void stf_sc1(float* p, size_t i, float v) {
  __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stf_sc1(half_t* p, size_t i, float v) {
  const half_t h = (half_t)v;
  unsigned short b;
  __builtin_memcpy(&b, &h, 2);
  __hip_atomic_store(reinterpret_cast<unsigned short*>(p) + i, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename TS>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const TS* p, int elems) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<TS*>(p), 0, elems * (int)sizeof(TS), 0x00020000);
}
template <typename TS>
__device__ __forceinline__ float4 ldf4_sc1(__amdgpu_buffer_rsrc_t rs, int i4) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  if constexpr (sizeof(TS) == 4) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, i4 * 16, 0, 16 /* sc1 */);
    return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
  } else {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, i4 * 8, 0, 16 /* sc1 */);
    half4_t h;
    __builtin_memcpy(&h, &w, 8);
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
  }
}
// the value an activation has after it was stored as TS (round to nearest even; float: unchanged)
template <typename TS> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<half_t>(float v) { return (float)(half_t)v; }

// Phase stamps exist only in the diagnostic build (make diag -> libhscn_diag.so); the
// shipped kernels execute none.  Values go to a buffer nothing else reads.
#ifdef HSCN_STAMPS
__device__ long long* g_stamp_buf = nullptr;  // [grid][64]
#define STAMP(k)                                                                                 \
  do {                                                                                           \
    if (threadIdx.x == 0 && g_stamp_buf) g_stamp_buf[(size_t)blockIdx.x * 64 + (k)] = clock64(); \
  } while (0)
#define STAMP_T(k, tid)                                                                           \
  do {                                                                                           \
    if ((int)threadIdx.x == (tid) && g_stamp_buf) g_stamp_buf[(size_t)blockIdx.x * 64 + (k)] = clock64(); \
  } while (0)
#else
#define STAMP(k) do {} while (0)
#define STAMP_T(k, tid) do {} while (0)
#endif

// A by-value kernel argument struct (at byte `offset` of the kernel-argument segment) as an object in the constant
// address space behind a LAUNDERED pointer: its fields are scalar loads issued where they are used, not preloaded in
// the kernel's entry block and carried (or spilled) from there.
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* late_args(int offset) {
  typedef const __attribute__((address_space(4))) char* P;
  P p = (P)__builtin_amdgcn_kernarg_segment_ptr() + offset;
  asm volatile("" : "+s"(p));
  return (const __attribute__((address_space(4))) T*)p;
}

// Workgroup barrier for phases that exchange data through LDS only.  __syncthreads() carries a
// workgroup-scope fence, which on gfx9 drains the vector-memory counter as well: every barrier
// would wait for the global stores (activations, CSR export, gradient partials) and prefetch loads
// still in flight.  Nothing in these kernels is exchanged between the waves of a workgroup through
// global memory, so waiting for the LDS / scalar counter is enough.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A wave group: a contiguous range of waves of the workgroup that works on its own arrays
// between the workgroup-wide barriers (every group executes the same barrier sequence).
struct Grp {
  int t;   // thread index inside the group
  int nt;  // threads in the group
  int w;   // wave index inside the group
  int nw;  // waves in the group
};

// sum over the 16 lanes of a DPP row (all 16 lanes receive the total): VALU only, no LDS
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
  return v;
}
// Whole-wave reductions for a fully active wave: DPP inside the four rows, then the four row
// results through readlane (scalar), folded in row order.  ~10 VALU ops instead of six dependent
// ds_bpermute round trips.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  const int r = __float_as_int(row16_sum(v));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(r, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(r, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(r, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(r, 48));
  return ((r0 + r1) + r2) + r3;
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  const int r = __float_as_int(row16_max(v));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(r, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(r, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(r, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(r, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ int wave_max_int(int v) {   // fully active wave
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// reductions over aligned groups of KP consecutive lanes (KP a power of two; every lane of the group
// receives the result): DPP inside a 16-lane row (xor 1, xor 2, mirror inside 8, mirror inside 16),
// bpermute only across rows
__device__ __forceinline__ float seg_sum(float v, int KP) {
  if (KP >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  if (KP >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  if (KP >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  if (KP >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  if (KP >= 32) v += __shfl_xor(v, 16, 64);
  if (KP >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float seg_max(float v, int KP) {
  if (KP >= 2) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
  if (KP >= 4) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
  if (KP >= 8) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
  if (KP >= 16) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
  if (KP >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
  if (KP >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// v + (the lane `off` away inside the 16-lane DPP row, rotation): with off = 8 then 4 every lane ends
// with the sum over the four lanes congruent to it mod 4 (the same pairs as an xor butterfly)
template <int OFF>
__device__ __forceinline__ float row_ror_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + OFF, 0xF, 0xF, true));
}

// the value lane K of each 16-lane DPP row holds, in every lane of that row (row_newbcast: gfx90a and later): a vector
// element crosses lanes in ONE VALU operation -- no LDS round trip, no v_readlane -> SGPR -> VALU hazard slots
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xF, 0xF, true));
}
// the value lane U of each aligned group of four lanes holds, in all four (DPP quad_perm [U,U,U,U])
template <int U>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), U * 0x55, 0xF, 0xF, true));
}
template <int U>
__device__ __forceinline__ int quad_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, U * 0x55, 0xF, 0xF, true);
}
// out[r] = acc + sum_k z[k] Wt[k * 16 + r], k ascending, for a 16-vector z that lives four features per lane
// (lane u of a DPP row holds z[4u .. 4u+3], u = 0 .. 3) and a 16 x 16 matrix in LDS; r = this lane's index in its row
template <int KQ = 0>
__device__ __forceinline__ float row_matvec16(const float4& z, const float* Wt, int r, float acc) {
  if constexpr (KQ < 4) {
    acc = fmaf(row_bcast<KQ>(z.x), Wt[(4 * KQ + 0) * 16 + r], acc);
    acc = fmaf(row_bcast<KQ>(z.y), Wt[(4 * KQ + 1) * 16 + r], acc);
    acc = fmaf(row_bcast<KQ>(z.z), Wt[(4 * KQ + 2) * 16 + r], acc);
    acc = fmaf(row_bcast<KQ>(z.w), Wt[(4 * KQ + 3) * 16 + r], acc);
    return row_matvec16<KQ + 1>(z, Wt, r, acc);
  } else {
    return acc;
  }
}
// acc = fmaf(x_k, w[k], acc) for k = 0 .. N-1 in ascending order, x_k = the value of lane k of this lane's DPP row:
// one row of a small matrix-vector product whose vector lives one element per lane
template <int N, int K = 0>
__device__ __forceinline__ float row_dot(float x, const float (&w)[N], float acc) {
  if constexpr (K < N) {
    acc = fmaf(row_bcast<K>(x), w[K], acc);
    return row_dot<N, K + 1>(x, w, acc);
  } else {
    return acc;
  }
}
// x_0 + x_1 + ... + x_{N-1} added in that order (the lanes of this lane's DPP row)
template <int N, int K = 0>
__device__ __forceinline__ float row_seq_sum(float x, float acc) {
  if constexpr (K < N) return row_seq_sum<N, K + 1>(x, acc + row_bcast<K>(x));
  else return acc;
}

// inclusive scan over the 64 lanes of a fully active wave: Hillis-Steele inside the four DPP rows
// (row_shr, zeros shifted in), then the three row totals through readlane.  VALU only.
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31);
  const int t2 = __builtin_amdgcn_readlane(v, 47);
  const int row = (threadIdx.x & 63) >> 4;
  return v + (row == 0 ? 0 : (row == 1 ? t0 : (row == 2 ? t0 + t1 : t0 + t1 + t2)));
}

// ---- group inclusive scan of a[0..n) in LDS, in place (two workgroup barriers) -----------------
__device__ void scan_inclusive_lds(int* a, int n, int* wsum /*[nw]*/, const Grp& G) {
  const int per = (n + G.nt - 1) / G.nt;
  const int b = G.t * per;
  int s = 0;
  for (int i = 0; i < per; ++i)
    if (b + i < n) s += a[b + i];
  const int lane = threadIdx.x & 63;
  const int incl = wave_incl_scan(s);
  if (lane == 63) wsum[G.w] = incl;
  lds_barrier();
  int off = 0;
  for (int i = 0; i < G.w; ++i) off += wsum[i];
  int run = off + incl - s;
  for (int i = 0; i < per; ++i)
    if (b + i < n) {
      run += a[b + i];
      a[b + i] = run;
    }
  lds_barrier();
}

// ---- stable CSR from staged edges (local ids; key -1 = dropped): low-degree rows --------------
// rowptr[0..nrows], col[ne]; rows keep ascending edge order.  Counting and placement by LDS int
// atomics on ONE counter array (counted up, then taken back down to zero by the placement: the slot
// an edge receives is arbitrary, the final order is not), then every edge ranks itself inside its row
// by edge number: O(degree) per edge, meant for rows of a few edges.  The prefix sum needs no barrier
// of its own: a wave re-adds the counts in front of its slice instead of waiting for the waves that
// own them.  cursor: [nrows+1], ZERO on entry unless zero_first, zero again on exit; tmp: [ne].
// Four workgroup barriers (five with zero_first).
constexpr int CSR_BUILD_BARRIERS = 4;
__device__ void build_csr_lds(const int* ek, const int* eo, int ne, int nrows, int* rowptr, int* col, int* cursor,
                              int* tmp, const Grp& G, bool zero_first) {
  if (zero_first) {
    for (int i = G.t; i <= nrows; i += G.nt) cursor[i] = 0;
    lds_barrier();
  }
  for (int e = G.t; e < ne; e += G.nt) {
    const int k = ek[e];
    if (k >= 0) atomicAdd(&cursor[k], 1);
  }
  lds_barrier();
  {  // rowptr[i] = sum of the counts of rows < i  (exclusive scan), rowptr[nrows] = total
    const int lane = threadIdx.x & 63;
    const int per = (nrows + G.nt - 1) / G.nt;
    const int wb = G.w * 64 * per;                         // first row of this wave's slice
    int front = 0;
    for (int j = lane; j < wb && j < nrows; j += 64) front += cursor[j];
    front = __builtin_amdgcn_readlane(wave_incl_scan(front), 63);
    const int b = G.t * per;
    int s = 0;
    for (int i = 0; i < per; ++i)
      if (b + i < nrows) s += cursor[b + i];
    int run = front + wave_incl_scan(s) - s;
    for (int i = 0; i < per; ++i)
      if (b + i < nrows) {
        rowptr[b + i] = run;
        run += cursor[b + i];
        if (b + i == nrows - 1) rowptr[nrows] = run;
      }
    if (nrows == 0 && G.t == 0) rowptr[0] = 0;
  }
  lds_barrier();
  for (int e = G.t; e < ne; e += G.nt) {
    const int k = ek[e];
    if (k < 0) continue;
    const int p = atomicAdd(&cursor[k], -1) - 1;
    tmp[rowptr[k] + p] = e;
  }
  lds_barrier();
  for (int e = G.t; e < ne; e += G.nt) {
    const int k = ek[e];
    if (k < 0) continue;
    const int s = rowptr[k], t = rowptr[k + 1];
    int rank = 0;
    for (int q = s; q < t; ++q) rank += (tmp[q] < e) ? 1 : 0;
    col[s + rank] = eo[e];
  }
  lds_barrier();
}

// ---- BOTH stable CSRs of one edge list (keyed by ek: rowptr_a / col_a = eo; keyed by eo: rowptr_t / col_t = ek)
// for rows of at most ELL_D edges, in TWO workgroup barriers (the pair of build_csr_lds calls takes four + four side
// by side).  Phase 1: every edge draws a slot in its row of either table from a returning LDS atomic and leaves its
// edge number there (the slot order is arbitrary, the final order is not).  Phase 2: a thread owns a row: it reads the
// row's <= ELL_D edge numbers, ranks them against each other in registers and stores the columns in ascending edge
// order; the row's offset comes from a prefix sum of the counts for which a wave re-adds the counts in front of its
// slice (no barrier of its own).  cnt_a, cnt_t: [nrows + 1], ovf: [1], all ZERO on entry; ell_a, ell_t: [nrows * ELL_D].
// dinv (optional) = in-degree^-1/2 of the rows keyed by ek.  Returns false -- for every thread alike -- when some row
// holds more than ELL_D edges; nothing usable has been written then and the caller takes the general build.
constexpr int ELL_D = 6;
__device__ bool build_csr_pair_ell(const int* ek, const int* eo, int ne, int nrows, int* rowptr_a, int* col_a,
                                   int* rowptr_t, int* col_t, float* dinv, int* cnt_a, int* ell_a, int* cnt_t,
                                   int* ell_t, int* ovf, int RTn, int wave, int NWn) {
  for (int e = threadIdx.x; e < ne; e += RTn) {
    const int k = ek[e], o = eo[e];
    if (k >= 0) {
      const int pa = atomicAdd(&cnt_a[k], 1), pt = atomicAdd(&cnt_t[o], 1);
      if (pa < ELL_D) ell_a[k * ELL_D + pa] = e; else ovf[0] = 1;
      if (pt < ELL_D) ell_t[o * ELL_D + pt] = e; else ovf[0] = 1;
    }
  }
  lds_barrier();
  if (ovf[0] != 0) return false;
  const int NA = NWn > 1 ? NWn / 2 : 1;
  auto rows = [&](const int* cnt, const int* ell, const int* other, int* rowptr, int* col, float* dv, int gw, int gnw) {
    const int lane = threadIdx.x & 63, gt = gw * 64 + lane, gnt = gnw * 64;
    const int per = (nrows + gnt - 1) / gnt;
    const int wb = gw * 64 * per;                          // first row of this wave's slice
    int front = 0;
    for (int j = lane; j < wb && j < nrows; j += 64) front += cnt[j];
    front = __builtin_amdgcn_readlane(wave_incl_scan(front), 63);
    const int b = gt * per;
    int s = 0;
    for (int i = 0; i < per; ++i)
      if (b + i < nrows) s += cnt[b + i];
    int run = front + wave_incl_scan(s) - s;
    for (int i = 0; i < per; ++i) {
      const int r = b + i;
      if (r >= nrows) break;
      const int c = cnt[r];
      int ev[ELL_D];
#pragma unroll
      for (int j = 0; j < ELL_D; ++j) ev[j] = j < c ? ell[r * ELL_D + j] : 0x7fffffff;
      int ov[ELL_D];
#pragma unroll
      for (int j = 0; j < ELL_D; ++j) ov[j] = other[j < c ? ev[j] : 0];
#pragma unroll
      for (int j = 0; j < ELL_D; ++j) {
        int rank = 0;
#pragma unroll
        for (int m = 0; m < ELL_D; ++m) rank += ev[m] < ev[j] ? 1 : 0;
        if (j < c) col[run + rank] = ov[j];
      }
      rowptr[r] = run;
      if (dv) dv[r] = c > 0 ? 1.0f / sqrtf((float)c) : 0.f;
      run += c;
      if (r == nrows - 1) rowptr[nrows] = run;
    }
    if (nrows == 0 && gt == 0) rowptr[0] = 0;
  };
  if (NWn == 1) {
    rows(cnt_a, ell_a, eo, rowptr_a, col_a, dinv, 0, 1);
    rows(cnt_t, ell_t, ek, rowptr_t, col_t, nullptr, 0, 1);
  } else if (wave < NA) {
    rows(cnt_a, ell_a, eo, rowptr_a, col_a, dinv, wave, NA);
  } else {
    rows(cnt_t, ell_t, ek, rowptr_t, col_t, nullptr, wave - NA, NWn - NA);
  }
  lds_barrier();
  return true;
}

// ---- the same two structures as 16-byte ROW RECORDS (rows of at most ELL_D = 6 edges, at most 65 535 rows) ------------
// rec[r] = {id0 | id1 << 16, id2 | id3 << 16, id4 | id5 << 16, r | count << 16}: the row's neighbours in ascending edge
// order as uint16 (unused slots hold r itself: a valid row to read with weight zero), the count in the top half of the
// last word.  A consumer reads a row's whole neighbourhood with ONE 16-byte LDS load, where rowptr -> col takes two
// dependent ones -- and the build needs no prefix sum: phase 1 as build_csr_pair_ell (slots drawn by LDS atomics), phase 2
// a thread per row (rank the <= 6 edge numbers in registers, store the record).  rec_a: rows keyed by ek, neighbours eo;
// rec_t: rows keyed by eo, neighbours ek.  cnt_*, ovf zero on entry.  Returns false (for every thread alike) when a row
// holds more than ELL_D edges: the caller takes the general CSR build then.
__device__ bool build_ell16_pair(const int* ek, const int* eo, int ne, int nrows, uint4* rec_a, uint4* rec_t, float* dinv,
                                 int* cnt_a, int* ell_a, int* cnt_t, int* ell_t, int* ovf, int RTn) {
  for (int e = threadIdx.x; e < ne; e += RTn) {
    const int k = ek[e], o = eo[e];
    if (k >= 0) {
      const int pa = atomicAdd(&cnt_a[k], 1), pt = atomicAdd(&cnt_t[o], 1);
      if (pa < ELL_D) ell_a[k * ELL_D + pa] = e; else ovf[0] = 1;
      if (pt < ELL_D) ell_t[o * ELL_D + pt] = e; else ovf[0] = 1;
    }
  }
  lds_barrier();
  if (ovf[0] != 0) return false;
  for (int idx = threadIdx.x; idx < 2 * nrows; idx += RTn) {
    const bool tside = idx >= nrows;
    const int r = tside ? idx - nrows : idx;
    const int* cnt = tside ? cnt_t : cnt_a;
    const int* ell = tside ? ell_t : ell_a;
    const int* other = tside ? ek : eo;
    const int c = cnt[r];
    int ev[ELL_D], ov[ELL_D];
#pragma unroll
    for (int j = 0; j < ELL_D; ++j) ev[j] = j < c ? ell[r * ELL_D + j] : 0x7fffffff;
#pragma unroll
    for (int j = 0; j < ELL_D; ++j) ov[j] = other[j < c ? ev[j] : 0];
    unsigned id[ELL_D];
#pragma unroll
    for (int j = 0; j < ELL_D; ++j) id[j] = (unsigned)r;
#pragma unroll
    for (int j = 0; j < ELL_D; ++j) {
      int rank = 0;
#pragma unroll
      for (int m = 0; m < ELL_D; ++m) rank += ev[m] < ev[j] ? 1 : 0;
      // (edge numbers are distinct, so the ranks of the c real entries are 0 .. c-1)
#pragma unroll
      for (int q = 0; q < ELL_D; ++q)
        if (j < c && rank == q) id[q] = (unsigned)ov[j];
    }
    const uint4 rec = make_uint4(id[0] | (id[1] << 16), id[2] | (id[3] << 16), id[4] | (id[5] << 16),
                                 (unsigned)r | ((unsigned)c << 16));
    (tside ? rec_t : rec_a)[r] = rec;
    if (!tside && dinv) dinv[r] = c > 0 ? 1.0f / sqrtf((float)c) : 0.f;
  }
  lds_barrier();
  return true;
}

// ---- stable CSR, few rows of high degree (local -> virtual: rows are clusters) ----------------
// Wave-ballot multisplit: edges are cut into 64-edge chunks (one wave each, in edge order);
// cnt[row][chunk] by ballot, one scan over (row-major, chunk-minor) gives every
// (row, chunk) its base slot, the rank inside the chunk is popcount(ballot & lanes below).
// cnt: [nrows * ceil(ne/64)] ints, tmp: [ne].  Five barriers.
constexpr int CSR_MULTISPLIT_BARRIERS = 5;
__device__ void build_csr_multisplit_lds(const int* ek, const int* eo, int ne, int nrows, int* rowptr, int* col,
                                         int* cnt, int* tmp, int* wsum, const Grp& G) {
  const int nchunk = (ne + 63) >> 6;
  const int lane = threadIdx.x & 63;
  for (int i = G.t; i < nrows * nchunk; i += G.nt) cnt[i] = 0;
  lds_barrier();
  for (int c = G.w; c < nchunk; c += G.nw) {
    const int e = c * 64 + lane;
    const int k = e < ne ? ek[e] : -1;
    unsigned long long todo = __ballot(k >= 0);
    int rank = 0;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      // (leader is wave-uniform: v_readlane, one VALU operation -- a ds_bpermute round trip per distinct key made this
      // loop the long pole of a 32-cluster build)
      const int k0 = __builtin_amdgcn_readlane(k, leader);
      const unsigned long long m = __ballot(k == k0);
      if (k == k0) rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == leader) cnt[k0 * nchunk + c] = __popcll(m);
      todo &= ~m;
    }
    if (e < ne) tmp[e] = rank;
  }
  lds_barrier();
  scan_inclusive_lds(cnt, nrows * nchunk, wsum, G);
  for (int r = G.t; r <= nrows; r += G.nt) rowptr[r] = (r * nchunk > 0) ? cnt[r * nchunk - 1] : 0;
  for (int e = G.t; e < ne; e += G.nt) {
    const int k = ek[e];
    if (k < 0) continue;
    const int idx = k * nchunk + (e >> 6);
    const int base = idx > 0 ? cnt[idx - 1] : 0;
    col[base + tmp[e]] = eo[e];
  }
  lds_barrier();
}

__device__ __forceinline__ void dinv_from_rowptr(const int* rowptr, int n, float* dinv, const Grp& G) {
  for (int i = G.t; i < n; i += G.nt) {
    const int d = rowptr[i + 1] - rowptr[i];
    dinv[i] = d > 0 ? 1.0f / sqrtf((float)d) : 0.f;
  }
}


// consumer side: one lane polls (relaxed, agent scope: an sc1 load) until the producer's counter reaches `want`;
// bounded.  The wave's loads of the published bytes follow the poll in program order and are sc1 buffer loads to
// registers (ldf4_sc1), which bypass this CU's L1: no agent-scope acquire (its buffer_inv + wait cost ~1.7 us per
// hand-off on the virtual branch's chain); ACQ = true issues one anyway, for consumers that use plain loads.
// Returns false on timeout.
template <bool ACQ>
__device__ __forceinline__ bool wait_published(const uint32_t* word, uint32_t want, int32_t* flag) {
  bool ok = true;
  if ((threadIdx.x & 63) == 0) {
    unsigned spins = 0;
    while ((int32_t)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) { ok = false; break; }   // ~ seconds: the producer is not coming
    }
    if (!ok && flag) atomicOr(flag, 8);
  }
  ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;
  if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the loads below the poll)
  return ok;
}


// out[p] = sum_g partials[g][p]: a block owns 32 parameters x 8 contiguous graph slices (coalesced
// over p), each slice summed in graph order, slices folded in slice order -> fixed summation tree
// (column p_scaled, if any, is multiplied by `scale`: the loss column of a step whose loss tail rode
// on the backward launch -- sum of the per-graph loss terms times 1/count)
// epoch (optional): the step counter of the one-launch training step (resident_step.h), advanced here -- after every
// workgroup of that launch has finished, before the next step's launch starts
__global__ void __launch_bounds__(256) k_param_reduce(const float* __restrict__ partials, float* __restrict__ out,
                                                      int B, int P, int p_scaled, float scale,
                                                      uint32_t* epoch = nullptr) {
  if (epoch && blockIdx.x == 0 && threadIdx.x == 0) epoch[0] = epoch[0] + 1u;
  __shared__ float red[8][32];
  const int pl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int p = blockIdx.x * 32 + pl;
  const int per = (B + 7) / 8;
  const int g0 = sl * per, g1 = (g0 + per) < B ? (g0 + per) : B;
  float s = 0.f;
  if (p < P) {
    int g = g0;
    for (; g + 16 <= g1; g += 16) {  // sixteen loads in flight (a whole slice of a 128-graph batch), added in graph order
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = partials[(size_t)(g + u) * P + p];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; g + 4 <= g1; g += 4) {
      const float a = partials[(size_t)g * P + p], b = partials[(size_t)(g + 1) * P + p];
      const float c = partials[(size_t)(g + 2) * P + p], d = partials[(size_t)(g + 3) * P + p];
      s += a; s += b; s += c; s += d;
    }
    for (; g < g1; ++g) s += partials[(size_t)g * P + p];
  }
  red[sl][pl] = s;
  lds_barrier();
  if (sl == 0 && p < P) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][pl];
    out[p] = p == p_scaled ? t * scale : t;
  }
}


}  // namespace
