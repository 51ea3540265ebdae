"""Host emulation of csrc/allreduce.hip's slot / flag / epoch protocol (TEST INFRASTRUCTURE: never imported by the
product).  Every rank is a process; the G slot buffers and flag arrays live in ONE POSIX shared-memory segment laid
out exactly as the kernel lays out a rank's allocation:

    rank p's slots : float32 [2 parities][G sources][stride]     stride = count rounded up to 4
    rank p's flags : uint32  [G sources][nchunks]                chunks of 2048 floats

and a call walks the kernel's steps chunk by chunk: epoch e = last + 1, parity e & 1, store into slot `rank` of
every rank, raise flag `rank` there with e, wait (bounded) until the own flags read e or e + 1
((int32)(flag - e) >= 0), add the own G slots in rank order with float32 roundings, scale, advance the epoch.
(The kernel's granule form for small buffers differs only in carrying the epoch inside each 8-byte element instead
of in a flag word; the parity / epoch reasoning checked here is common to both.)
It pins the index arithmetic and the two-parity reuse argument under real concurrency on a CPU box; the memory
ordering of the device code is what the -m gpu tests (two processes sharing one GPU) are for."""
import time
from multiprocessing import shared_memory

import numpy as np

CHUNK = 2048


def layout(count, G):
    stride = (count + 3) & ~3
    nch = (count + CHUNK - 1) // CHUNK
    slot_words = 2 * G * stride
    flag_words = G * nch
    per_rank = slot_words + flag_words
    return stride, nch, slot_words, flag_words, per_rank


class HostSlotAllReduce:
    def __init__(self, count, rank, G, shm_name, spin_limit=2_000_000, jitter=None):
        self.count, self.rank, self.G = int(count), int(rank), int(G)
        self.stride, self.nch, sw, fw, per = layout(self.count, self.G)
        self.shm = shared_memory.SharedMemory(name=shm_name)
        words = np.ndarray((self.G * per,), dtype=np.uint32, buffer=self.shm.buf)
        self.slots = [words[p * per: p * per + sw].view(np.float32).reshape(2, self.G, self.stride) for p in range(self.G)]
        self.flags = [words[p * per + sw: (p + 1) * per].reshape(self.G, self.nch) for p in range(self.G)]
        self.epoch = np.zeros(self.nch, dtype=np.uint32)
        self.status = np.zeros(2, dtype=np.uint32)
        self.spin_limit = spin_limit
        self.jitter = jitter

    @staticmethod
    def create(count, G):
        per = layout(count, G)[4]
        shm = shared_memory.SharedMemory(create=True, size=4 * G * per)
        np.ndarray((G * per,), dtype=np.uint32, buffer=shm.buf)[:] = 0
        return shm

    def __call__(self, flat, scale):
        """``flat``: torch float32 CPU tensor (reduced in place) or numpy float32 array."""
        a = flat.numpy() if hasattr(flat, "numpy") else flat
        assert a.dtype == np.float32 and a.size == self.count
        scale = np.float32(scale)
        for c in range(self.nch):
            e = np.uint32(self.epoch[c] + np.uint32(1))
            par = int(e & 1)
            lo, hi = c * CHUNK, min(self.count, (c + 1) * CHUNK)
            for p in range(self.G):
                self.slots[p][par, self.rank, lo:hi] = a[lo:hi]
            if self.jitter is not None:
                time.sleep(self.jitter())
            for p in range(self.G):
                self.flags[p][self.rank, c] = e
            ok, spins = False, 0
            while not ok:
                seen = self.flags[self.rank][:, c].copy()
                ok = bool(np.all((seen - e).astype(np.int32) >= 0))
                spins += 1
                if not ok and spins > self.spin_limit:
                    self.status[0] |= 1
                    late = (seen - e).astype(np.int32) < 0
                    self.status[1] |= int(sum(1 << q for q in range(self.G) if late[q]))
                    break
            if ok:
                s = self.slots[self.rank][par, 0, lo:hi].copy()
                for q in range(1, self.G):
                    s = (s + self.slots[self.rank][par, q, lo:hi]).astype(np.float32)
                a[lo:hi] = (s * scale).astype(np.float32)
            self.epoch[c] = e

    def check(self):
        if self.status[0]:
            raise RuntimeError(f"emulated one-shot all-reduce timed out (sources mask {int(self.status[1]):#x})")

    def close(self):
        self.slots = self.flags = None
        self.shm.close()
