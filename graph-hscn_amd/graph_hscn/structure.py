"""Device-side graph structure for the hot-path kernels.

PyG's MessagePassing gathers by ``edge_index[0]`` and scatters by
``edge_index[1]`` on every call (reference model/hscn.py:32,40,85-93).  The HIP
kernels instead walk a CSR keyed by the target node, built once per
``edge_index`` tensor on the device (stable: a row keeps its edges in edge
order, the order torch's CPU ``index_add_`` sums in) and cached for the layers
and steps that reuse the same tensor.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _hip


@dataclass
class CSR:
    rowptr: Tensor   # int32 [num_rows+1]
    col: Tensor      # int32 [max(E,1)]
    eid: Tensor      # int32 [max(E,1)]  original edge number of each slot
    num_rows: int
    num_cols: int
    num_edges: int
    flag: Tensor     # int32 [1]; 1 if an edge was out of range and skipped

    def check(self) -> None:
        """Synchronising validity check (out-of-range indices)."""
        if int(self.flag.item()) != 0:
            raise IndexError("edge_index holds node ids outside [0, num_nodes)")


def build_csr(key: Tensor, other: Tensor, num_rows: int, num_cols: int) -> CSR:
    """COO(int64) -> CSR(int32) through hscn_csr_build."""
    if key.dtype != torch.int64 or other.dtype != torch.int64:
        raise TypeError("edge_index must be int64")
    dev = key.device
    E = int(key.numel())
    key = key.contiguous()
    other = other.contiguous()
    rowptr = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    eid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    L = _hip.lib()
    ws_bytes = int(L.hscn_csr_workspace_bytes(E, num_rows))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _hip.call("hscn_csr_build", _hip.ptr(key) if E else None, _hip.ptr(other) if E else None, E, num_rows,
              num_cols, _hip.ptr(rowptr), _hip.ptr(col), _hip.ptr(eid), _hip.ptr(flag), _hip.ptr(ws),
              ws_bytes, _hip.stream())
    return CSR(rowptr, col, eid, num_rows, num_cols, E, flag)


def build_csr_pair(src: Tensor, dst: Tensor, num_src: int, num_dst: int):
    """Both stable CSRs of one edge list through hscn_csr_build_pair: (keyed by target, keyed by source) --
    bit-identical to build_csr(dst, src, num_dst, num_src) and build_csr(src, dst, num_src, num_dst), 7 launches
    instead of 16."""
    if src.dtype != torch.int64 or dst.dtype != torch.int64:
        raise TypeError("edge_index must be int64")
    dev = src.device
    E = int(src.numel())
    src = src.contiguous()
    dst = dst.contiguous()
    i32 = dict(dtype=torch.int32, device=dev)
    rowptr, rowptr_t = torch.empty(num_dst + 1, **i32), torch.empty(num_src + 1, **i32)
    col, eid = torch.empty(max(E, 1), **i32), torch.empty(max(E, 1), **i32)
    col_t, eid_t = torch.empty(max(E, 1), **i32), torch.empty(max(E, 1), **i32)
    flag = torch.zeros(1, **i32)
    L = _hip.lib()
    ws_bytes = int(L.hscn_csr_pair_workspace_bytes(E, num_src, num_dst))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _hip.call("hscn_csr_build_pair", _hip.ptr(src) if E else None, _hip.ptr(dst) if E else None, E, num_src, num_dst,
              _hip.ptr(rowptr), _hip.ptr(col), _hip.ptr(eid), _hip.ptr(rowptr_t), _hip.ptr(col_t), _hip.ptr(eid_t),
              _hip.ptr(flag), _hip.ptr(ws), ws_bytes, _hip.stream())
    return (CSR(rowptr, col, eid, num_dst, num_src, E, flag), CSR(rowptr_t, col_t, eid_t, num_src, num_dst, E, flag))


class Relation:
    """One edge type ``src -> dst``: forward CSR (keyed by target), and lazily
    the transposed CSR (keyed by source), cross positions and GCN degree norm."""

    def __init__(self, edge_index: Tensor, num_src: int, num_dst: int, both: Optional[bool] = None,
                 csr: Optional[CSR] = None):
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("edge_index must be [2, E]")
        self.edge_index = edge_index
        self.num_src = int(num_src)
        self.num_dst = int(num_dst)
        self.num_edges = int(edge_index.size(1))
        self._csr_t: Optional[CSR] = None
        # both = True: the caller knows a backward through the propagate is coming (it needs the source-keyed CSR): the
        # two CSRs then come out of ONE build (hscn_csr_build_pair); otherwise the source-keyed one is built on first use
        if csr is not None:        # the caller has the target-keyed CSR of this very list already (adopt_relation)
            self.csr = csr
        elif both:
            self.csr, self._csr_t = build_csr_pair(edge_index[0], edge_index[1], self.num_src, self.num_dst)
        else:
            self.csr = build_csr(edge_index[1], edge_index[0], self.num_dst, self.num_src)
        self._pos_t: Optional[Tensor] = None
        self._dinv: Optional[Tensor] = None

    @property
    def csr_t(self) -> CSR:
        if self._csr_t is None:
            self._csr_t = build_csr(self.edge_index[0], self.edge_index[1], self.num_src, self.num_dst)
        return self._csr_t

    @property
    def pos_t(self) -> Tensor:
        """CSR slot of the edge stored at each transposed-CSR slot."""
        if self._pos_t is None:
            E = self.num_edges
            dev = self.edge_index.device
            inv = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
            pos = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
            _hip.call("hscn_csr_cross_positions", _hip.ptr(self.csr.eid), _hip.ptr(self.csr_t.eid), E,
                      _hip.ptr(inv), _hip.ptr(pos), _hip.stream())
            self._pos_t = pos
        return self._pos_t

    @property
    def dinv(self) -> Tensor:
        """PyG gcn_norm degree term for unit weights, no self loops: in-degree^-1/2."""
        if self._dinv is None:
            d = torch.empty(max(self.num_dst, 1), dtype=torch.float32, device=self.edge_index.device)
            _hip.call("hscn_gcn_dinv", _hip.ptr(self.csr.rowptr), self.num_dst, _hip.ptr(d), _hip.stream())
            self._dinv = d
        return self._dinv

    def check(self) -> None:
        self.csr.check()


_CACHE: "OrderedDict[Tuple, Relation]" = OrderedDict()
_CACHE_MAX = 64


def relation_of(edge_index: Tensor, num_src: int, num_dst: int, cache: bool = True, both: bool = False) -> Relation:
    """Structure for ``edge_index``; cached per tensor (pointer, shape, version) so
    the L layers of one forward and repeated epochs over the same batch build it once.
    both: build the source-keyed CSR together with the target-keyed one (see Relation)."""
    if not cache:
        return Relation(edge_index, num_src, num_dst, both)
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, str(edge_index.device),
           int(num_src), int(num_dst))
    rel = _CACHE.get(key)
    if rel is not None and rel.edge_index is edge_index:
        _CACHE.move_to_end(key)
        return rel
    rel = Relation(edge_index, num_src, num_dst, both)
    _CACHE[key] = rel
    while len(_CACHE) > _CACHE_MAX:
        _CACHE.popitem(last=False)
    return rel


def adopt_relation(edge_index: Tensor, num_src: int, num_dst: int, csr: CSR) -> Relation:
    """Enter a target-keyed CSR that was built for ``edge_index`` anyway (gcn_norm builds one for the degrees) into the
    relation cache, so that the conv layer that receives this tensor next does not build it a second time."""
    rel = Relation(edge_index, num_src, num_dst, csr=csr)
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, str(edge_index.device),
           int(num_src), int(num_dst))
    _CACHE[key] = rel
    while len(_CACHE) > _CACHE_MAX:
        _CACHE.popitem(last=False)
    return rel


def with_self_loops(edge_index: Tensor, num_nodes: int) -> Tensor:
    """PyG ``add_remaining_self_loops`` for unit weights: existing self loops are taken out,
    one loop per node is appended after the remaining edges (so a target-keyed stable CSR ends every row
    with its loop, the order the reference's scatter-add sums in).  Index plumbing on the device."""
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(int(num_nodes), dtype=torch.int64, device=edge_index.device)
    return torch.cat([edge_index[:, keep], loops.unsqueeze(0).expand(2, -1)], 1).contiguous()


_LOOP_CACHE: "OrderedDict[Tuple, Tuple[Tensor, Relation]]" = OrderedDict()


def self_loop_relation_of(edge_index: Tensor, num_nodes: int, both: bool = False) -> Relation:
    """Relation over ``edge_index`` + self loops (what GCNConv(add_self_loops=True) normalises and
    propagates over), cached per tensor like ``relation_of``."""
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, str(edge_index.device), int(num_nodes))
    hit = _LOOP_CACHE.get(key)
    if hit is not None and hit[0] is edge_index:
        _LOOP_CACHE.move_to_end(key)
        return hit[1]
    rel = Relation(with_self_loops(edge_index, num_nodes), num_nodes, num_nodes, both)
    _LOOP_CACHE[key] = (edge_index, rel)
    while len(_LOOP_CACHE) > _CACHE_MAX:
        _LOOP_CACHE.popitem(last=False)
    return rel


def clear_cache() -> None:
    _LOOP_CACHE.clear()
    _CACHE.clear()


def segments_from_batch(batch: Tensor, num_segments: Optional[int] = None) -> CSR:
    """CSR over graphs from a PyG ``batch`` vector (any order): row g lists the
    nodes of graph g.  ``num_segments=None`` reads ``batch.max()+1`` (one sync),
    as global_mean_pool does."""
    if num_segments is None:
        num_segments = int(batch.max().item()) + 1 if batch.numel() else 0
    node = torch.arange(batch.numel(), dtype=torch.int64, device=batch.device)
    return build_csr(batch, node, int(num_segments), int(batch.numel()))
