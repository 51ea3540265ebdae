"""Minimal graph containers with the protocol the reference's loops touch.

torch_geometric is not available on the target boxes, so the package owns the
few container behaviours the hot path's callers use
(/root/reference/graph_hscn/train/train.py:73-77, train_clustering.py:36-47,
loader/hetero_data.py:62-87, loader/loader.py:48-60):

  Data(x, edge_index, edge_weight, y, num_nodes), ``.to(device)``
  HeteroData: ``h["local"].x``, ``h["local","to","local"].edge_index``,
              ``.x_dict``, ``.edge_index_dict``
  Batch / HeteroBatch ``.from_data_list`` with PyG collate semantics
              (concat per type, offset edge indices, ``batch`` / ``ptr`` vectors)
  DataLoader(dataset, batch_size, shuffle)
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor

EdgeType = Tuple[str, str, str]


class Store:
    """Attribute bag for one node type or one edge type."""

    def __init__(self, **kw):
        self.__dict__["_d"] = {}
        for k, v in kw.items():
            self._d[k] = v

    def __getattr__(self, k):
        try:
            return self.__dict__["_d"][k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self._d[k] = v

    def __contains__(self, k):
        return k in self._d

    def keys(self):
        return self._d.keys()

    def items(self):
        return self._d.items()

    def to(self, device, non_blocking: bool = False) -> "Store":
        out = Store()
        for k, v in self._d.items():
            out._d[k] = v.to(device, non_blocking=non_blocking) if isinstance(v, Tensor) else v
        return out

    @property
    def num_nodes(self) -> int:
        if "num_nodes" in self._d:
            return int(self._d["num_nodes"])
        return int(self._d["x"].size(0))


class Data(Store):
    """Homogeneous graph.  ``edge_index`` is int64 ``[2, E]`` = [source; target]."""

    def __init__(self, x: Optional[Tensor] = None, edge_index: Optional[Tensor] = None,
                 y: Optional[Tensor] = None, edge_weight: Optional[Tensor] = None,
                 num_nodes: Optional[int] = None, **kw):
        super().__init__()
        self._d.update(dict(x=x, edge_index=edge_index, y=y, edge_weight=edge_weight))
        if num_nodes is not None:
            self._d["num_nodes"] = int(num_nodes)
        self._d.update(kw)

    @property
    def num_features(self) -> int:
        return int(self.x.size(1))

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.size(1))

    def to(self, device, non_blocking: bool = False) -> "Data":
        out = self.__class__.__new__(self.__class__)
        out.__dict__["_d"] = {
            k: (v.to(device, non_blocking=non_blocking) if isinstance(v, Tensor) else v)
            for k, v in self._d.items()
        }
        return out


def _cat_targets(ys: Sequence[Tensor]) -> Tensor:
    """PyG collate of ``y``: concatenate along dim 0 AS IS (a graph-level class index ``y=[c]`` of five graphs
    gives ``[5]``, node labels ``y[N_i]`` give ``[sum N_i]``, ``[1, C]`` rows give ``[B, C]``); only 0-dim labels
    are unsqueezed.  The reference's multiclass branch (``pred.ndim > 1 and true.ndim == 1``, loss.py:11)
    depends on 1-D targets staying 1-D."""
    return torch.cat([y.reshape(1) if y.dim() == 0 else y for y in ys], 0)


class Batch(Data):
    """Block-diagonal union of ``Data`` graphs (PyG ``Batch.from_data_list``)."""

    @classmethod
    def from_data_list(cls, graphs: Sequence[Data]) -> "Batch":
        ns = [g.num_nodes for g in graphs]
        ptr = torch.zeros(len(ns) + 1, dtype=torch.long)
        ptr[1:] = torch.cumsum(torch.as_tensor(ns, dtype=torch.long), 0)
        out = cls(
            x=torch.cat([g.x for g in graphs], 0),
            edge_index=torch.cat([g.edge_index + int(ptr[i]) for i, g in enumerate(graphs)], 1),
            num_nodes=int(ptr[-1]),
        )
        if all(g.y is not None for g in graphs):
            out.y = _cat_targets([g.y for g in graphs])
        if all(g.edge_weight is not None for g in graphs):
            out.edge_weight = torch.cat([g.edge_weight for g in graphs], 0)
        out.batch = torch.repeat_interleave(torch.arange(len(ns)), torch.as_tensor(ns))
        out.ptr = ptr
        out.num_graphs = len(graphs)
        # per-graph node / edge ranges + maxima for the graph-resident kernels
        es = [int(g.edge_index.size(1)) for g in graphs]
        eptr = torch.zeros(len(es) + 1, dtype=torch.int32)
        eptr[1:] = torch.cumsum(torch.as_tensor(es, dtype=torch.int64), 0).to(torch.int32)
        out.ptr32 = ptr.to(torch.int32)
        out.eptr32 = eptr
        out.max_nodes = int(max(ns)) if ns else 0
        out.max_edges = int(max(es)) if es else 0
        return out

    @property
    def batch_size(self) -> int:
        return int(self.num_graphs)


class HeteroData:
    """Typed graph: node stores keyed by ``str``, edge stores by ``(src, rel, dst)``."""

    def __init__(self):
        self._nodes: Dict[str, Store] = {}
        self._edges: Dict[EdgeType, Store] = {}

    def __getitem__(self, key: Union[str, EdgeType]) -> Store:
        if isinstance(key, tuple):
            return self._edges.setdefault(tuple(key), Store())
        return self._nodes.setdefault(key, Store())

    @property
    def node_types(self) -> List[str]:
        return list(self._nodes)

    @property
    def edge_types(self) -> List[EdgeType]:
        return list(self._edges)

    @property
    def x_dict(self) -> Dict[str, Tensor]:
        return {k: s.x for k, s in self._nodes.items() if "x" in s}

    @property
    def edge_index_dict(self) -> Dict[EdgeType, Tensor]:
        return {k: s.edge_index for k, s in self._edges.items() if "edge_index" in s}

    def to(self, device, non_blocking: bool = False):
        out = self.__class__()
        out._nodes = {k: s.to(device, non_blocking) for k, s in self._nodes.items()}
        out._edges = {k: s.to(device, non_blocking) for k, s in self._edges.items()}
        for k, v in self.__dict__.items():
            if k not in ("_nodes", "_edges"):
                out.__dict__[k] = v
        return out


class HeteroBatch(HeteroData):
    """PyG collate for ``HeteroData``: relation insertion order
    of the first graph is kept (ll, vv, lv for loader/hetero_data.py:67,77,84)."""

    num_graphs: int = 0

    @classmethod
    def from_data_list(cls, graphs: Sequence[HeteroData]) -> "HeteroBatch":
        out = cls()
        out.num_graphs = len(graphs)
        ptrs: Dict[str, Tensor] = {}
        for nt in graphs[0].node_types:
            ns = [g[nt].num_nodes for g in graphs]
            ptr = torch.zeros(len(ns) + 1, dtype=torch.long)
            ptr[1:] = torch.cumsum(torch.as_tensor(ns, dtype=torch.long), 0)
            ptrs[nt] = ptr
            st = out[nt]
            st.x = torch.cat([g[nt].x for g in graphs], 0)
            st.batch = torch.repeat_interleave(torch.arange(len(ns)), torch.as_tensor(ns))
            st.ptr = ptr
            st.ptr32 = ptr.to(torch.int32)          # per-graph node ranges for the graph-resident kernels
            st.max_nodes = int(max(ns)) if ns else 0
            st.num_nodes = int(ptr[-1])
            if all("y" in g[nt] and g[nt].y is not None for g in graphs):
                st.y = _cat_targets([g[nt].y for g in graphs])
        for et in graphs[0].edge_types:
            s, _, d = et
            off = torch.stack([ptrs[s][:-1], ptrs[d][:-1]], 0)  # [2, B]
            out[et].edge_index = torch.cat(
                [g[et].edge_index + off[:, i : i + 1] for i, g in enumerate(graphs)], 1)
            es = [int(g[et].edge_index.size(1)) for g in graphs]
            eptr = torch.zeros(len(es) + 1, dtype=torch.int32)
            eptr[1:] = torch.cumsum(torch.as_tensor(es, dtype=torch.int64), 0).to(torch.int32)
            out[et].ptr32 = eptr                    # edges of graph g are the slice [ptr32[g], ptr32[g+1])
            out[et].max_edges = int(max(es)) if es else 0
        return out

    @property
    def batch_size(self) -> int:
        return int(self.num_graphs)

    def with_feature_dtype(self, dtype) -> "HeteroBatch":
        """The same batch with the node features of every type stored as ``dtype`` (``torch.float16``: the
        half-storage mode of the graph-resident engine, BASELINE.json configs[4]; integer-valued atom features
        below 2048 are exact in half).  Everything else is shared, not copied."""
        out = self.__class__()
        out._nodes = {}
        for k, st in self._nodes.items():
            ns = Store()
            ns._d.update(st._d)
            if "x" in st:
                ns._d["x"] = st.x.to(dtype)
            out._nodes[k] = ns
        out._edges = self._edges
        for k, v in self.__dict__.items():
            if k not in ("_nodes", "_edges", "_resident_meta"):
                out.__dict__[k] = v
        return out


class DataLoader:
    """List-backed mini-batch iterator (reference: loader/loader.py:48-60 wraps the
    PyG DataLoader; workers default to 0, defaults.py:3)."""

    def __init__(self, dataset: Sequence, batch_size: int = 1, shuffle: bool = False,
                 num_workers: int = 0, persistent_workers: bool = False,
                 generator: Optional[torch.Generator] = None, **_):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.generator = generator

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator:
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for i in range(0, n, self.batch_size):
            yield collate([self.dataset[j] for j in order[i : i + self.batch_size]])


def collate(graphs: Sequence):
    if isinstance(graphs[0], HeteroData):
        return HeteroBatch.from_data_list(graphs)
    return Batch.from_data_list(graphs)
