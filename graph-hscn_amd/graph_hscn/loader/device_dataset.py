"""A hetero dataset that lives in HBM, and batches collated on the device.

The reference hands every step a batch collated on the host (PyG ``DataLoader`` ->
``Batch.from_data_list``; loader/hetero_data.py:91-106, loader/loader.py:48-60): ~10 ms of Python for a
128-graph batch against a 47 us training step, plus a PCIe copy.  An MI355X has 288 GB: the whole Peptides
hetero dataset is ~250 MB.  ``DeviceHeteroDataset`` keeps it resident as concatenated arrays (edge lists with
per-graph LOCAL node ids, int32) and ``gather(ids)`` writes the batch made of graphs ``ids`` -- a slice of the
epoch's permutation, itself a device tensor -- into the fixed-capacity buffers of a ``StaticHeteroBatch`` with
ONE launch (``hscn_collate_gather``), bit for bit what ``HeteroBatch.from_data_list`` builds for the same list.
An epoch is then ``perm = torch.randperm(G, device=...)`` and, per step, ``ds.gather(perm[i:i+B]);
step.replay(); optimizer.step()`` with no host work that scales with the batch.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch
from torch import Tensor

from .. import _hip
from ..data import HeteroBatch, HeteroData
from ..replay import LL, LV, VV, StaticHeteroBatch

_RELS = (LL, VV, LV)     # include/hscn.h: relation order of hscn_hetero_dataset / hscn_hetero_batch_out


class _Dataset(ctypes.Structure):
    _fields_ = [("x_local", ctypes.c_void_p), ("x_virtual", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("nptr", ctypes.c_void_p), ("vptr", ctypes.c_void_p), ("src", ctypes.c_void_p * 3),
                ("dst", ctypes.c_void_p * 3), ("eptr", ctypes.c_void_p * 3), ("G", ctypes.c_int64),
                ("F", ctypes.c_int32), ("C", ctypes.c_int32)]


class _BatchOut(ctypes.Structure):
    _fields_ = [("x_local", ctypes.c_void_p), ("x_virtual", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("ptr_local", ctypes.c_void_p), ("ptr_virtual", ctypes.c_void_p), ("ptr32_local", ctypes.c_void_p),
                ("ptr32_virtual", ctypes.c_void_p), ("batch_local", ctypes.c_void_p), ("batch_virtual", ctypes.c_void_p),
                ("ei", ctypes.c_void_p * 3), ("eptr32", ctypes.c_void_p * 3), ("ncap", ctypes.c_int64),
                ("vcap", ctypes.c_int64), ("ecap", ctypes.c_int64 * 3)]


def _top_sum(sizes: Tensor, k: int) -> int:
    """Largest total any ``k`` graphs can have."""
    return int(torch.topk(sizes, min(k, sizes.numel())).values.sum()) if sizes.numel() else 0


class DeviceHeteroDataset:
    """``graphs``: the list ``generate_hetero_data`` returns (or any sequence of ``HeteroData`` with the
    local / virtual node types and the ll / vv / lv relations).  ``batch_size`` graphs per step."""

    def __init__(self, graphs: Sequence[HeteroData], device, batch_size: int, resident_structure: bool = False):
        """``resident_structure``: also keep every graph's CSRs (ll by target and by source, lv, vv; graph-local
        ids) and degree norms in HBM -- built ONCE here by ``hscn_resident_structure`` over the dataset laid out as
        one batch -- and gather them with every batch (``static.batch.structure``), so that a step can load its
        structure instead of rebuilding it (``CapturedStep(..., structure="batch")``).  +~40 % dataset bytes."""
        if not graphs:
            raise ValueError("empty dataset")
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.num_graphs = len(graphs)
        self.batch_size = int(batch_size)
        whole = HeteroBatch.from_data_list(graphs)            # once, on the host: global ids + per-graph ranges
        nptr, vptr = whole["local"].ptr, whole["virtual"].ptr
        base = {"local": nptr, "virtual": vptr}
        self.F = int(whole["local"].x.size(1))
        y = whole["local"].y if "y" in whole["local"] else None
        self.C = None if y is None else int(y.size(1))
        dev = self.device
        self._t = {"x_local": whole["local"].x.float().contiguous().to(dev),
                   "x_virtual": whole["virtual"].x.float().contiguous().to(dev),
                   "y": None if y is None else y.float().contiguous().to(dev),
                   "nptr": nptr.to(dev), "vptr": vptr.to(dev)}
        sizes = {"local": nptr[1:] - nptr[:-1], "virtual": vptr[1:] - vptr[:-1]}
        esizes = {}
        for r, et in enumerate(_RELS):
            ei = whole[et].edge_index
            eptr = whole[et].ptr32.to(torch.int64)
            esizes[et] = eptr[1:] - eptr[:-1]
            owner = torch.repeat_interleave(torch.arange(self.num_graphs), esizes[et])
            s, _, d = et
            self._t[f"src{r}"] = (ei[0] - base[s][owner]).to(torch.int32).contiguous().to(dev)
            self._t[f"dst{r}"] = (ei[1] - base[d][owner]).to(torch.int32).contiguous().to(dev)
            self._t[f"eptr{r}"] = eptr.contiguous().to(dev)
        B = self.batch_size
        self.static = StaticHeteroBatch.from_capacities(
            B, dev, _top_sum(sizes["local"], B), _top_sum(sizes["virtual"], B),
            {et: _top_sum(esizes[et], B) for et in _RELS},
            {"local": int(sizes["local"].max()), "virtual": int(sizes["virtual"].max())},
            {et: int(esizes[et].max()) if esizes[et].numel() else 0 for et in _RELS}, self.F, self.C)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.structure = None
        if resident_structure:
            from ..engine import BatchStructure, build_structure
            self.structure = build_structure(whole.to(dev))              # one launch over all G graphs
            st0 = self.static
            self._out_structure = BatchStructure(dev, st0.N, st0.V, B, st0.E[LL], st0.E[LV], st0.E[VV])
            st0.batch.structure = self._out_structure
            torch.cuda.synchronize(dev)                                   # (the int64 COO copy of `whole` may go now)
        self._perm: Optional[Tensor] = None
        self._cursor: Optional[Tensor] = None
        self._counters: dict = {}      # id(step) -> (the step's per-step device counter, its value at epoch start, step)
        t = self._t
        p = _hip.ptr
        self._ds = _Dataset(p(t["x_local"]), p(t["x_virtual"]), p(t["y"]), p(t["nptr"]), p(t["vptr"]),
                            (ctypes.c_void_p * 3)(*[p(t[f"src{r}"]) for r in range(3)]),
                            (ctypes.c_void_p * 3)(*[p(t[f"dst{r}"]) for r in range(3)]),
                            (ctypes.c_void_p * 3)(*[p(t[f"eptr{r}"]) for r in range(3)]),
                            self.num_graphs, self.F, self.C or 0)
        hb, st = self.static.batch, self.static
        self._out = _BatchOut(p(hb["local"].x), p(hb["virtual"].x), p(hb["local"].y) if self.C else None,
                              p(hb["local"].ptr), p(hb["virtual"].ptr), p(hb["local"].ptr32), p(hb["virtual"].ptr32),
                              p(hb["local"].batch), p(hb["virtual"].batch),
                              (ctypes.c_void_p * 3)(*[p(hb[et].edge_index) for et in _RELS]),
                              (ctypes.c_void_p * 3)(*[p(hb[et].ptr32) for et in _RELS]),
                              st.N, st.V, (ctypes.c_int64 * 3)(*[st.E[et] for et in _RELS]))

    @property
    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._t.values() if t is not None)

    def gather(self, ids: Tensor) -> HeteroBatch:
        """Make ``self.static`` hold the batch of graphs ``ids`` (int64 ``[batch_size]`` on the device; order
        kept).  Asynchronous on the current stream, capturable; returns ``self.static.batch``."""
        if ids.dtype != torch.int64 or ids.device != self.device or ids.numel() != self.batch_size:
            raise ValueError(f"ids must be int64 [{self.batch_size}] on {self.device}")
        ids = ids.contiguous()
        if self.structure is not None:
            _hip.call("hscn_collate_gather_structure", ctypes.byref(self._ds), ctypes.byref(self.structure.c),
                      _hip.ptr(ids), self.batch_size, ctypes.byref(self._out), ctypes.byref(self._out_structure.c),
                      _hip.ptr(self.flag), None, None, _hip.stream())
        _hip.call("hscn_collate_gather", ctypes.byref(self._ds), _hip.ptr(ids), self.batch_size,
                  ctypes.byref(self._out), _hip.ptr(self.flag), None, None, _hip.stream())
        return self.static.batch

    # ---- an epoch that walks by itself: the permutation and a batch counter live on the device ----------------
    def new_epoch(self, generator: Optional[torch.Generator] = None) -> Tensor:
        """Draw the epoch's permutation into the dataset's own buffer and rewind the batch counter.  Returns the
        permutation (a device tensor; ``gather_next`` serves its first ``num_graphs // batch_size`` slices)."""
        if self._perm is None:
            # (one spare batch of valid ids behind the permutation: a replay too many reads those, not stray memory)
            self._perm_buf = torch.zeros(self.num_graphs + self.batch_size, dtype=torch.int64, device=self.device)
            self._perm = self._perm_buf[: self.num_graphs]
            self._cursor = torch.zeros(1, dtype=torch.int32, device=self.device)
        torch.randperm(self.num_graphs, device=self.device, generator=generator, out=self._perm)
        self._cursor.zero_()
        for counter, base, _ in self._counters.values():   # steps with a counter of their own: this epoch's slices
            base.copy_(counter)                            # count from its value now
        return self._perm

    def gather_next(self, step=None) -> HeteroBatch:
        """Gather the next batch of the current epoch's permutation and advance the device-side counter, with no
        host argument that changes from step to step, so the launches can be CAPTURED in front of the training step
        (``CapturedStep(..., pre=ds.gather_next)``) -- a replay is then "next batch + iteration".
        ``step``: the ``ResidentTrainStep`` the gather is captured with (``CapturedStep`` passes it).  When that step
        keeps a per-step counter on the device (word 0 of the one-launch step's sync buffer, advanced by its gradient
        fold), the gather reads its slice number off THAT counter (minus its value when the epoch began) and the
        launch that would advance a counter of our own (4.5 us per iteration) does not exist."""
        if self._perm is None:
            raise RuntimeError("call new_epoch() first")
        sync = getattr(step, "_sync", None) if step is not None else None
        if sync is not None and getattr(step, "advances_sync", False):
            ent = self._counters.get(id(step))
            if ent is None:                      # (uint32 counter read as int32; the base is this epoch's start)
                ent = self._counters[id(step)] = (sync[:1], sync[:1].clone(), step)
            cur, base = _hip.ptr(ent[0]), _hip.ptr(ent[1])
        else:
            cur, base = _hip.ptr(self._cursor), None
        if self.structure is not None:        # (before the gather proper: that call advances the cursor)
            _hip.call("hscn_collate_gather_structure", ctypes.byref(self._ds), ctypes.byref(self.structure.c),
                      _hip.ptr(self._perm), self.batch_size, ctypes.byref(self._out),
                      ctypes.byref(self._out_structure.c), _hip.ptr(self.flag), cur, base, _hip.stream())
        _hip.call("hscn_collate_gather", ctypes.byref(self._ds), _hip.ptr(self._perm), self.batch_size,
                  ctypes.byref(self._out), _hip.ptr(self.flag), cur, base, _hip.stream())
        return self.static.batch

    def check(self) -> None:
        """Synchronising validity check of the gathers issued so far."""
        if int(self.flag.item()) & 8:
            raise IndexError("a graph id was outside the dataset (or a batch exceeded the static capacity)")
