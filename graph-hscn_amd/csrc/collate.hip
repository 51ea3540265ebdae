// Collate on the device: a training batch gathered out of a hetero dataset that lives in HBM
// (reference: PyG DataLoader -> Batch.from_data_list on the host for every step, loader/loader.py:48-60,
// loader/hetero_data.py:91-106).  The whole Peptides hetero dataset is ~250 MB -- it stays
// resident; a step's batch is `ids[B]` (a slice of the epoch's permutation, on the device) and ONE launch
// that copies the chosen graphs' features, edge lists (re-based to batch node numbering), targets and
// writes the per-graph segment tables, straight into the fixed-capacity buffers a captured step replays on.
// No host collate, no PCIe traffic, no synchronisation.
//
// grid = (B, 6): block (j, part) handles one array family of graph ids[j].  Every block derives the
// destination offsets it needs itself -- a prefix sum over the sizes of the j graphs before it, B <= a few
// thousand 8-byte loads spread over 256 threads, all the tables a block needs in ONE pass (slot_ranges) -- so there is
// no scan launch and no inter-block dependency.
#include "hscn_common.h"

namespace {

constexpr int CT = 256;

// (offset, size) of graph slot j in NT tables of per-graph ranges AT ONCE: one pass over the ids before j (each id is
// loaded once, its NT range lengths in parallel) and one block reduction for all of them.  (Measured: the launch stays
// at 6.7 us either way -- cursor -> ids -> ranges -> data is four dependent trips to memory plus the launch itself.)
template <int NT>
__device__ __forceinline__ void slot_ranges(const int64_t* const (&p)[NT], const int64_t* __restrict__ ids, int j,
                                            long long* red /*[NT][CT / 64]*/, long long (&off)[NT], long long (&size)[NT]) {
  long long acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = 0;
  for (int q = threadIdx.x; q < j; q += CT) {
    const int64_t g = ids[q];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] += p[t][g + 1] - p[t][g];
  }
  const int64_t gj = ids[j];
#pragma unroll
  for (int t = 0; t < NT; ++t) size[t] = p[t][gj + 1] - p[t][gj];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    long long v = acc[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[t * (CT / 64) + (threadIdx.x >> 6)] = v;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    long long s_ = 0;
#pragma unroll
    for (int w = 0; w < CT / 64; ++w) s_ += red[t * (CT / 64) + w];
    off[t] = s_;
  }
}

__global__ void __launch_bounds__(CT) k_collate_gather(const hscn_hetero_dataset D, const int64_t* __restrict__ ids,
                                                       int B, const hscn_hetero_batch_out O, int32_t* __restrict__ flag,
                                                       const int32_t* __restrict__ cursor,
                                                       const int32_t* __restrict__ cursor_base) {
  __shared__ long long red[5 * (CT / 64)];
  const int j = blockIdx.x, part = blockIdx.y;
  // batch number `cursor` (- `cursor_base`) of a permutation that lives on the device
  if (cursor) ids += (int64_t)(cursor[0] - (cursor_base ? cursor_base[0] : 0)) * B;
  const int64_t g = ids[j];
  if (g < 0 || g >= D.G) {              // (uniform per block)
    if (threadIdx.x == 0 && flag) atomicOr(flag, 8);
    return;
  }
  // every range this part needs, in one pass: [0] local nodes, [1] virtual nodes, [2 ..] edge tables
  long long on = 0, n = 0, ov = 0, nv = 0;
  long long oe[3] = {0, 0, 0}, ne[3] = {0, 0, 0};
  if (part <= 1) {
    const int64_t* const tb[1] = {part == 0 ? D.nptr : D.vptr};
    long long o1[1], s1[1];
    slot_ranges<1>(tb, ids, j, red, o1, s1);
    if (part == 0) { on = o1[0]; n = s1[0]; } else { ov = o1[0]; nv = s1[0]; }
  } else if (part <= 4) {
    const int r = part - 2;
    const int64_t* const tb[3] = {D.nptr, D.vptr, D.eptr[r]};
    long long o3[3], s3[3];
    slot_ranges<3>(tb, ids, j, red, o3, s3);
    on = o3[0]; n = s3[0]; ov = o3[1]; nv = s3[1]; oe[r] = o3[2]; ne[r] = s3[2];
  } else {
    const int64_t* const tb[5] = {D.nptr, D.vptr, D.eptr[0], D.eptr[1], D.eptr[2]};
    long long o5[5], s5[5];
    slot_ranges<5>(tb, ids, j, red, o5, s5);
    on = o5[0]; n = s5[0]; ov = o5[1]; nv = s5[1];
    for (int r = 0; r < 3; ++r) { oe[r] = o5[2 + r]; ne[r] = s5[2 + r]; }
  }
  const bool fits_n = on + n <= O.ncap, fits_v = ov + nv <= O.vcap;
  if (part == 0) {                      // local features + batch vector
    if (!fits_n) { if (threadIdx.x == 0 && flag) atomicOr(flag, 8); return; }
    const float* s = D.x_local + (size_t)D.nptr[g] * D.F;
    float* d = O.x_local + (size_t)on * D.F;
    for (long long i = threadIdx.x; i < n * D.F; i += CT) d[i] = s[i];
    for (long long i = threadIdx.x; i < n; i += CT) O.batch_local[on + i] = j;
  } else if (part == 1) {               // virtual features + batch vector
    if (!fits_v) { if (threadIdx.x == 0 && flag) atomicOr(flag, 8); return; }
    const float* s = D.x_virtual + (size_t)D.vptr[g] * D.F;
    float* d = O.x_virtual + (size_t)ov * D.F;
    for (long long i = threadIdx.x; i < nv * D.F; i += CT) d[i] = s[i];
    for (long long i = threadIdx.x; i < nv; i += CT) O.batch_virtual[ov + i] = j;
  } else if (part <= 4) {               // relation r = ll, vv, lv: local ids -> batch ids
    const int r = part - 2;
    if (oe[r] + ne[r] > O.ecap[r]) { if (threadIdx.x == 0 && flag) atomicOr(flag, 8); return; }
    const long long os = r == 1 ? ov : on, od = r == 0 ? on : ov;
    const int64_t e0 = D.eptr[r][g];
    const int32_t* ss = D.src[r] + e0;
    const int32_t* dd = D.dst[r] + e0;
    int64_t* es = O.ei[r] + oe[r];
    int64_t* ed = O.ei[r] + O.ecap[r] + oe[r];
    const long long ner = ne[r];
    for (long long e = threadIdx.x; e < ner; e += CT) {
      es[e] = (int64_t)ss[e] + os;
      ed[e] = (int64_t)dd[e] + od;
    }
  } else {                              // targets + the five segment tables (entry j; the last slot also writes the totals)
    if (D.y && O.y)
      for (int c = threadIdx.x; c < D.C; c += CT) O.y[(size_t)j * D.C + c] = D.y[(size_t)g * D.C + c];
    if (threadIdx.x == 0) {
      O.ptr_local[j] = on;    O.ptr32_local[j] = (int32_t)on;
      O.ptr_virtual[j] = ov;  O.ptr32_virtual[j] = (int32_t)ov;
      for (int r = 0; r < 3; ++r) O.eptr32[r][j] = (int32_t)oe[r];
      if (j == B - 1) {
        O.ptr_local[B] = on + n;    O.ptr32_local[B] = (int32_t)(on + n);
        O.ptr_virtual[B] = ov + nv; O.ptr32_virtual[B] = (int32_t)(ov + nv);
        for (int r = 0; r < 3; ++r) O.eptr32[r][B] = (int32_t)(oe[r] + ne[r]);
      }
    }
  }
}

// structure slices (include/hscn.h: hscn_structure; graph-local ids, so they are copied as they are) of the graphs
// ids[0..B) into batch-level arrays: grid = (B, 4), part 0: ll by target + degree norm, 1: ll by source, 2: lv, 3: vv
__global__ void __launch_bounds__(CT) k_collate_structure(const hscn_hetero_dataset D, const hscn_structure S,
                                                          const int64_t* __restrict__ ids, int B,
                                                          const hscn_hetero_batch_out O, const hscn_structure T,
                                                          int32_t* __restrict__ flag, const int32_t* __restrict__ cursor,
                                                          const int32_t* __restrict__ cursor_base) {
  __shared__ long long red[2 * (CT / 64)];
  const int j = blockIdx.x, part = blockIdx.y;
  if (cursor) ids += (int64_t)(cursor[0] - (cursor_base ? cursor_base[0] : 0)) * B;
  const int64_t g = ids[j];
  if (g < 0 || g >= D.G) {
    if (threadIdx.x == 0 && flag) atomicOr(flag, 8);
    return;
  }
  long long on = 0, n = 0, ov = 0, nv = 0, oe = 0, ne = 0;
  const int r = part <= 1 ? 0 : (part == 2 ? 2 : 1);          // relation order of the dataset: ll, vv, lv
  {
    const int64_t* const tb[2] = {part <= 1 ? D.nptr : D.vptr, D.eptr[r]};
    long long o2[2], s2[2];
    slot_ranges<2>(tb, ids, j, red, o2, s2);
    if (part <= 1) { on = o2[0]; n = s2[0]; } else { ov = o2[0]; nv = s2[0]; }
    oe = o2[1]; ne = s2[1];
  }
  if (on + n > O.ncap || ov + nv > O.vcap || oe + ne > O.ecap[r]) {
    if (threadIdx.x == 0 && flag) atomicOr(flag, 8);
    return;
  }
  const int64_t e0 = D.eptr[r][g];
  if (part <= 1) {
    const int64_t rs = D.nptr[g] + g;
    const int32_t* rp = (part == 0 ? S.ll_rowptr_d : S.ll_rowptr_s) + rs;
    const int32_t* cl = (part == 0 ? S.ll_col_d : S.ll_col_s) + e0;
    int32_t* rpo = (part == 0 ? T.ll_rowptr_d : T.ll_rowptr_s) + on + j;
    int32_t* clo = (part == 0 ? T.ll_col_d : T.ll_col_s) + oe;
    for (long long i = threadIdx.x; i <= n; i += CT) rpo[i] = rp[i];
    for (long long i = threadIdx.x; i < ne; i += CT) clo[i] = cl[i];
    if (part == 0)
      for (long long i = threadIdx.x; i < n; i += CT) T.ll_dinv[on + i] = S.ll_dinv[D.nptr[g] + i];
  } else {
    const int64_t rs = D.vptr[g] + g;
    const int32_t* rp = (part == 2 ? S.lv_rowptr : S.vv_rowptr) + rs;
    const int32_t* cl = (part == 2 ? S.lv_col : S.vv_col) + e0;
    int32_t* rpo = (part == 2 ? T.lv_rowptr : T.vv_rowptr) + ov + j;
    int32_t* clo = (part == 2 ? T.lv_col : T.vv_col) + oe;
    for (long long i = threadIdx.x; i <= nv; i += CT) rpo[i] = rp[i];
    for (long long i = threadIdx.x; i < ne; i += CT) clo[i] = cl[i];
    if (part == 3)
      for (long long i = threadIdx.x; i < nv; i += CT) T.vv_dinv[ov + i] = S.vv_dinv[D.vptr[g] + i];
  }
}

// after the gather of batch `cursor`: the next replay of the same captured launches takes the next slice.
// (A launch of its own: letting the gather's last block advance the counter -- sign-off counter, device-scope
// fence -- was measured at +23 us per step over 768 blocks; this costs ~4.  With cursor_base the counter is one that
// SOMEBODY ELSE advances once per step -- the epoch word of the one-launch training step's sync buffer -- and the
// slice is counter - base: no launch at all.)
__global__ void k_cursor_advance(int32_t* cursor) { cursor[0] += 1; }

}  // namespace

extern "C" int hscn_collate_gather(const hscn_hetero_dataset* ds, const int64_t* ids, int64_t B,
                                   const hscn_hetero_batch_out* out, int32_t* flag, int32_t* cursor,
                                   const int32_t* cursor_base, void* stream_) {
  if (!ds || !out || B < 0 || B > 65535) return HSCN_E_BADARG;
  if (B == 0) return 0;
  if (!ids || !ds->x_local || !ds->x_virtual || !ds->nptr || !ds->vptr || ds->F < 1 || ds->G < 1) return HSCN_E_BADARG;
  if (!out->x_local || !out->x_virtual || !out->ptr_local || !out->ptr_virtual || !out->ptr32_local ||
      !out->ptr32_virtual || !out->batch_local || !out->batch_virtual)
    return HSCN_E_BADARG;
  for (int r = 0; r < 3; ++r)
    if (!ds->src[r] || !ds->dst[r] || !ds->eptr[r] || !out->ei[r] || !out->eptr32[r] || out->ecap[r] < 1) return HSCN_E_BADARG;
  if ((ds->y != nullptr) != (out->y != nullptr)) return HSCN_E_BADARG;
  k_collate_gather<<<dim3((unsigned)B, 6), CT, 0, hscn_stream(stream_)>>>(*ds, ids, (int)B, *out, flag, cursor, cursor_base);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  if (cursor && !cursor_base) {
    k_cursor_advance<<<1, 1, 0, hscn_stream(stream_)>>>(cursor);
    HSCN_RETURN_IF_LAUNCH_FAILED();
  }
  return 0;
}

extern "C" int hscn_collate_gather_structure(const hscn_hetero_dataset* ds, const hscn_structure* dss, const int64_t* ids,
                                             int64_t B, const hscn_hetero_batch_out* out, const hscn_structure* outs,
                                             int32_t* flag, const int32_t* cursor, const int32_t* cursor_base,
                                             void* stream_) {
  if (!ds || !dss || !out || !outs || B < 0 || B > 65535) return HSCN_E_BADARG;
  if (B == 0) return 0;
  if (!ids || !ds->nptr || !ds->vptr || ds->G < 1) return HSCN_E_BADARG;
  for (int r = 0; r < 3; ++r)
    if (!ds->eptr[r] || out->ecap[r] < 1) return HSCN_E_BADARG;
  const hscn_structure* both[2] = {dss, outs};
  for (const hscn_structure* q : both)
    if (!q->ll_rowptr_d || !q->ll_col_d || !q->ll_rowptr_s || !q->ll_col_s || !q->ll_dinv || !q->lv_rowptr ||
        !q->lv_col || !q->vv_rowptr || !q->vv_col || !q->vv_dinv)
      return HSCN_E_BADARG;
  k_collate_structure<<<dim3((unsigned)B, 4), CT, 0, hscn_stream(stream_)>>>(*ds, *dss, ids, (int)B, *out, *outs, flag, cursor, cursor_base);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
