// One-shot peer-to-peer all-reduce of the flat gradient buffer (the ONE exchange step of the data-parallel hot path:
// it sits between loss.backward() and optimizer.step(), reference graph_hscn/train/train.py:87-94; the reference
// itself is single process).
//
// The message is 4.6 KB (headline model) to 640 KB: pure latency.  A ring all-reduce walks 2(G-1) dependent hops;
// the 8 GPUs of a node reach each other in ONE xGMI hop, so every rank
//   (a) stores its buffer into slot `rank` of EVERY rank's slot buffer (peer memory mapped through hipIpc),
//   (b) raises flag `rank` on every rank with the step's epoch number (system-scope release),
//   (c) waits until its own G flags carry the epoch (bounded spin), acquires,
//   (d) adds its G slots IN RANK ORDER (the same summation order on every rank: replicas stay bit-identical)
//       and writes scale * sum back over its buffer.
// One launch, one hop, capturable (no host interaction; the epoch lives in device memory and is advanced here).
//
// Slot reuse: slots are double-buffered by epoch parity.  A rank leaves epoch e only after it has seen every peer's
// flag e, and a peer raises flag e + 1 only after it has left epoch e; so when rank r overwrites parity (e & 1) at
// epoch e + 2, every peer has raised e + 1, i.e. has finished reading the epoch-e data.  A flag may therefore read e
// or e + 1 while a rank waits for e: the test is (int32)(flag - e) >= 0.
//
// Two forms of the same protocol, chosen by size:
//   * granules (count <= AR_GRANULE_MAX, the headline's 1 146 floats): a slot element is ONE naturally aligned 8-byte
//     word {float bits, epoch}, stored and polled with relaxed system-scope 64-bit atomics.  Value and tag travel in
//     one single-copy-atomic object, so there is no flag, no release and no acquire fence: the reader spins on the
//     element itself until its tag is the epoch.  One fabric round trip in all.
//   * slabs + flags (larger buffers): 16-byte stores of plain floats, per-(source, chunk) epoch flags, one
//     system-scope release on the writer and one acquire on the reader per workgroup (two L2 maintenance operations,
//     ~1.7 us each: measured +4.7 us per step at one rank against +2.0 for the granule form's target).
//
// Memory: slots and flags are fine-grained device allocations (hscn_comm_alloc), the only kind for which the HSA
// memory model promises system-scope release/acquire between agents inside a running kernel.
#include "hscn_common.h"
#include <cstdlib>
#include <cstring>

namespace {

constexpr int AR_MAXG = 8;
constexpr int AR_THREADS = 256;
constexpr int AR_CHUNK = 2048;   // floats per workgroup (8 KB: two 16-byte pieces per lane and peer)
constexpr int AR_GRANULE_MAX = 16384;   // counts up to this use the granule form (one element per thread)

struct ArArgs {
  float* slots[AR_MAXG];      // slots[p]: rank p's slot buffer as mapped here: [2 parities][G sources][stride]
  uint32_t* flags[AR_MAXG];   // flags[p]: rank p's flag words: [G sources][nchunks]
  float* flat;                // [count] in: this rank's values; out: scale * sum over ranks
  uint32_t* epoch;            // [nchunks] local: last completed epoch of each chunk (0 before the first call)
  uint32_t* status;           // [2] local: [0] bit 0 = a wait timed out, [1] = mask of the sources that were missing
  int64_t count, stride;
  float scale;
  int rank, G, nchunks;
  uint32_t spin_limit;
};

__global__ void __launch_bounds__(AR_THREADS) k_allreduce_oneshot(const ArArgs A) {
  __shared__ uint32_t s_e;
  __shared__ int s_ok;
  const int c = blockIdx.x, t = threadIdx.x;
  if (t == 0) { s_e = A.epoch[c] + 1u; s_ok = 1; }
  __syncthreads();
  const uint32_t e = s_e;
  const int64_t par = (int64_t)(e & 1u) * A.G;
  const int64_t lo = (int64_t)c * AR_CHUNK;
  const int64_t hi = lo + AR_CHUNK < A.count ? lo + AR_CHUNK : A.count;
  const bool vec = ((reinterpret_cast<uintptr_t>(A.flat) & 15) == 0);
  const int64_t hi4 = vec ? lo + ((hi - lo) & ~int64_t(3)) : lo;   // [lo, hi4) by 16-byte pieces, [hi4, hi) by words

  // (a) my values into slot `rank` of every rank (my own included: the sum below then reads G slots of ONE buffer)
  for (int64_t i = lo + 4 * t; i < hi4; i += 4 * AR_THREADS) {
    const float4 v = *reinterpret_cast<const float4*>(A.flat + i);
#pragma unroll
    for (int p = 0; p < AR_MAXG; ++p)
      if (p < A.G) *reinterpret_cast<float4*>(A.slots[p] + (par + A.rank) * A.stride + i) = v;
  }
  for (int64_t i = hi4 + t; i < hi; i += AR_THREADS) {
    const float v = A.flat[i];
#pragma unroll
    for (int p = 0; p < AR_MAXG; ++p)
      if (p < A.G) A.slots[p][(par + A.rank) * A.stride + i] = v;
  }
  // (b) every storing wave drains, the workgroup meets, ONE wave releases at system scope and raises the flags
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t < AR_MAXG) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (ROCm 7.2 may drop the fence's own wait: cdna_hip_programming.md G16)
    if (t < A.G)
      __hip_atomic_store(A.flags[t] + (int64_t)A.rank * A.nchunks + c, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (c) lane q waits for source q's flag on THIS rank; bounded
    bool have = t >= A.G;
    const uint32_t* mine = A.flags[A.rank] + (int64_t)(t < A.G ? t : 0) * A.nchunks + c;
    uint32_t spins = 0;
    while (true) {
      if (!have) have = (int32_t)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) >= 0;
      const unsigned long long missing = __ballot(!have) & 0xffull;
      if (missing == 0) break;
      if (++spins > A.spin_limit) {
        if (t == 0) {
          s_ok = 0;
          atomicOr(A.status, 1u);
          atomicOr(A.status + 1, (uint32_t)missing);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // (d) rank-ordered sum of my G slots; a timed-out chunk keeps this rank's own values (the host raises on status)
  if (s_ok) {
    const float* base = A.slots[A.rank] + par * A.stride;
    for (int64_t i = lo + 4 * t; i < hi4; i += 4 * AR_THREADS) {
      float4 s = *reinterpret_cast<const float4*>(base + i);
      for (int q = 1; q < A.G; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(base + q * A.stride + i);
        s.x = add_rn(s.x, v.x); s.y = add_rn(s.y, v.y); s.z = add_rn(s.z, v.z); s.w = add_rn(s.w, v.w);
      }
      s.x = mul_rn(s.x, A.scale); s.y = mul_rn(s.y, A.scale); s.z = mul_rn(s.z, A.scale); s.w = mul_rn(s.w, A.scale);
      *reinterpret_cast<float4*>(A.flat + i) = s;
    }
    for (int64_t i = hi4 + t; i < hi; i += AR_THREADS) {
      float s = base[i];
      for (int q = 1; q < A.G; ++q) s = add_rn(s, base[q * A.stride + i]);
      A.flat[i] = mul_rn(s, A.scale);
    }
  }
  if (t == 0) A.epoch[c] = e;
}

// Granule form: thread t of workgroup c owns element i = c * 256 + t.  It stores {value, e} into slot `rank` of every
// rank, then polls its element in each of its own G slots until the tag reads e (all G loads in flight per round),
// adds the G values in rank order, scales and writes back.  No LDS, no barrier except the one that keeps the epoch
// word from being advanced before every thread has read it.
__global__ void __launch_bounds__(AR_THREADS) k_allreduce_granules(const ArArgs A) {
  typedef unsigned long long u64;
  const int c = blockIdx.x, t = threadIdx.x;
  const uint32_t e = A.epoch[c] + 1u;
  const int64_t par = (int64_t)(e & 1u) * A.G;
  const int64_t i = (int64_t)c * AR_THREADS + t;
  if (i < A.count) {
    const float mine = A.flat[i];
    const u64 g = ((u64)e << 32) | (u64)__float_as_uint(mine);
#pragma unroll
    for (int p = 0; p < AR_MAXG; ++p)
      if (p < A.G)
        __hip_atomic_store(reinterpret_cast<u64*>(A.slots[p]) + (par + A.rank) * A.stride + i, g, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    const u64* base = reinterpret_cast<const u64*>(A.slots[A.rank]) + par * A.stride + i;
    u64 v[AR_MAXG];
    unsigned have = 0;
    const unsigned want = (1u << A.G) - 1u;
    uint32_t spins = 0;
    while (true) {
#pragma unroll
      for (int q = 0; q < AR_MAXG; ++q)
        if (q < A.G && !((have >> q) & 1u))
          v[q] = __hip_atomic_load(base + (int64_t)q * A.stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
      for (int q = 0; q < AR_MAXG; ++q)
        if (q < A.G && (uint32_t)(v[q] >> 32) == e) have |= 1u << q;
      if (have == want) break;
      if (++spins > A.spin_limit) {
        atomicOr(A.status, 1u);
        atomicOr(A.status + 1, want & ~have);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    if (have == want) {
      float s = __uint_as_float((uint32_t)v[0]);
#pragma unroll
      for (int q = 1; q < AR_MAXG; ++q)
        if (q < A.G) s = add_rn(s, __uint_as_float((uint32_t)v[q]));
      A.flat[i] = mul_rn(s, A.scale);
    }
  }
  __syncthreads();
  if (t == 0) A.epoch[c] = e;
}

inline bool ar_granules(int64_t count) {
  static const int forced = [] { const char* v = getenv("HSCN_ALLREDUCE_FORM"); return v ? (v[0] == 'g' ? 1 : (v[0] == 's' ? 2 : 0)) : 0; }();
  if (forced == 1) return true;
  if (forced == 2) return false;
  return count <= AR_GRANULE_MAX;
}
inline int64_t ar_stride(int64_t count) { return (count + 3) & ~int64_t(3); }
inline int ar_chunks(int64_t count) {
  return (int)(ar_granules(count) ? (count + AR_THREADS - 1) / AR_THREADS : (count + AR_CHUNK - 1) / AR_CHUNK);
}

}  // namespace

extern "C" {

int hscn_comm_alloc(size_t bytes, int kind, void** out_ptr_host) {
  if (!out_ptr_host || bytes == 0 || kind < 0 || kind > 2) return HSCN_E_BADARG;
  void* p = nullptr;
  hipError_t e = kind == 2 ? hipMalloc(&p, bytes)
                           : hipExtMallocWithFlags(&p, bytes, kind == 0 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
  e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
  *out_ptr_host = p;
  return 0;
}

int hscn_comm_free(void* ptr) {
  if (!ptr) return HSCN_E_BADARG;
  return (int)hipFree(ptr);
}

int hscn_comm_ipc_export(void* ptr, void* handle64_host) {
  if (!ptr || !handle64_host) return HSCN_E_BADARG;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI carries the handle as 64 bytes");
  hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64_host), ptr);
  if (e != hipSuccess) (void)hipGetLastError();
  return (int)e;
}

int hscn_comm_ipc_open(const void* handle64_host, void** out_ptr_host) {
  if (!handle64_host || !out_ptr_host) return HSCN_E_BADARG;
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64_host, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(out_ptr_host, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) (void)hipGetLastError();
  return (int)e;
}

int hscn_comm_ipc_close(void* ptr) {
  if (!ptr) return HSCN_E_BADARG;
  return (int)hipIpcCloseMemHandle(ptr);
}

size_t hscn_allreduce_oneshot_slot_bytes(int64_t count, int G) {
  if (count <= 0 || G < 1 || G > AR_MAXG) return 0;
  return (size_t)2 * G * ar_stride(count) * (ar_granules(count) ? sizeof(uint64_t) : sizeof(float));
}

size_t hscn_allreduce_oneshot_flag_bytes(int64_t count, int G) {
  if (count <= 0 || G < 1 || G > AR_MAXG) return 0;
  return (size_t)G * ar_chunks(count) * sizeof(uint32_t);
}

int64_t hscn_allreduce_oneshot_chunks(int64_t count) { return count > 0 ? ar_chunks(count) : 0; }

int hscn_allreduce_oneshot(float* flat, int64_t count, void* const* peer_slots_host, void* const* peer_flags_host,
                           uint32_t* epoch, uint32_t* status, int rank, int G, float scale, uint32_t spin_limit,
                           void* stream) {
  if (!flat || count <= 0 || !peer_slots_host || !peer_flags_host || !epoch || !status || G < 1 || G > AR_MAXG ||
      rank < 0 || rank >= G)
    return HSCN_E_BADARG;
  if (ar_chunks(count) > 1024) return HSCN_E_UNSUPPORTED;   // every workgroup must be resident while it waits
  ArArgs A;
  for (int p = 0; p < AR_MAXG; ++p) {
    A.slots[p] = p < G ? static_cast<float*>(peer_slots_host[p]) : nullptr;
    A.flags[p] = p < G ? static_cast<uint32_t*>(peer_flags_host[p]) : nullptr;
    if (p < G && (!A.slots[p] || !A.flags[p])) return HSCN_E_BADARG;
  }
  A.flat = flat; A.epoch = epoch; A.status = status;
  A.count = count; A.stride = ar_stride(count);
  A.scale = scale; A.rank = rank; A.G = G; A.nchunks = ar_chunks(count);
  A.spin_limit = spin_limit ? spin_limit : (1u << 21);
  if (ar_granules(count))
    hipLaunchKernelGGL(k_allreduce_granules, dim3(A.nchunks), dim3(AR_THREADS), 0, hscn_stream(stream), A);
  else
    hipLaunchKernelGGL(k_allreduce_oneshot, dim3(A.nchunks), dim3(AR_THREADS), 0, hscn_stream(stream), A);
  HSCN_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

}  // extern "C"
