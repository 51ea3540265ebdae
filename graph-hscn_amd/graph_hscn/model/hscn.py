"""SCN (MinCUT spectral clustering net) and HSCN (heterogeneous local/virtual
message passing) with the reference's class names, constructor and ``forward``
signatures (/root/reference/graph_hscn/model/hscn.py:19-140), computing on
MI355X through the HIP C ABI (include/hscn.h).

Differences a caller can observe, all documented in DESIGN.md:
  * parameters are materialised eagerly (the reference's lazy ``-1`` dims
    materialise after the optimizer was built, train/train.py:155-159);
  * ``SCN.forward`` accepts an optional ``node_ptr`` to process a block-diagonal
    batch of graphs in one call (losses = mean over graphs); without it the call
    is the reference's single-graph step;
  * the dense ``[1,n,n]`` adjacency the reference returns and both callers drop
    (train/train_clustering.py:45,65) is built on the device only for
    single-graph calls; batched calls return ``None`` in that slot.
"""
from __future__ import annotations

import os

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from ..config.config import ACT_DICT, CONV_DICT, HSCNConfig
from ..nn import GATConv, GCNConv, GraphConv, HeteroConv, Linear
from ..nn import functional as Fh
from ..nn.pool import (global_mean_pool, mincut_pool_sparse, to_dense_adj, to_dense_adj_batched,
                       to_dense_adj_ragged)
from ..structure import Relation, relation_of
from .. import engine as _engine

LL = ("local", "to", "local")
VV = ("virtual", "to", "virtual")
LV = ("local", "to", "virtual")


def _act_name(act) -> Optional[str]:
    if isinstance(act, str):
        return act.lower()
    return getattr(act, "hscn_name", None)


class _MessagePassingStack(nn.Module):
    """Stand-in for PyG ``Sequential('x, edge_index, edge_weight', [...])``
    (hscn.py:28-45): GraphConv children are registered as ``module_{2i}`` so the
    ``state_dict`` keys equal the reference's."""

    def __init__(self, num_features: int, mp_units: list, act: str):
        super().__init__()
        self.act = act
        self.num = len(mp_units)
        dims = [num_features] + list(mp_units)
        for i in range(self.num):
            setattr(self, f"module_{2 * i}", GraphConv(dims[i], dims[i + 1]))

    def forward(self, x: Tensor, rel: Relation, edge_weight: Optional[Tensor]) -> Tensor:
        for i in range(self.num):
            x = getattr(self, f"module_{2 * i}")(x, rel, edge_weight, act=self.act)
        return x


class SCN(nn.Module):
    def __init__(self, mp_units: list, mp_act: str, num_features: int, num_clusters: int,
                 mlp_units: list = [], mlp_act: str = "identity", mincut_route: str = "sparse"):
        """``mincut_route`` (extension): how ``dense_mincut_pool``'s contractions are evaluated.
        "sparse" -- on the edge list, A never densified (tr(S^T A S) = sum over edges of s_i . s_j; the fused
        graph-resident launch when the model has the reference's shape);
        "dense"  -- the reference's literal sequence ``to_dense_adj`` -> ``dense_mincut_pool`` (model/hscn.py:61-63)
        with the [B,n,n] adjacency materialised and A S, S^T (A S), S^T S, S^T X on the matrix cores (csrc/dense.hip,
        exact-fp32 MFMA): BASELINE.json configs[3], PascalVOC-SP with 64 clusters;
        "auto"   -- dense from 64 clusters on.  Same values either way (tests/test_gpu_models.py)."""
        super().__init__()
        if mincut_route not in ("sparse", "dense", "auto"):
            raise ValueError(f"mincut_route must be 'sparse', 'dense' or 'auto', not {mincut_route!r}")
        self.mincut_route = mincut_route
        self.num_clusters = int(num_clusters)
        if _act_name(mp_act) not in ACT_DICT or _act_name(mlp_act) not in ACT_DICT:
            raise KeyError(f"unknown activation {mp_act!r}/{mlp_act!r}")  # ACT_DICT[...] at hscn.py:34,53
        self.mp = _MessagePassingStack(num_features, mp_units, _act_name(mp_act))
        out_channels = mp_units[-1]
        self.mlp_act = _act_name(mlp_act)
        self.mlp = nn.Sequential()
        # hscn.py:50-53 keeps `out_channels` as the input width of every layer (the
        # notebook's `out_chan = units` got inverted); restated literally.
        for units in mlp_units:
            self.mlp.append(Linear(out_channels, units))
            self.mlp.append(nn.Identity())
        self.mlp.append(Linear(out_channels, num_clusters))

    def _dense(self) -> bool:
        return self.mincut_route == "dense" or (self.mincut_route == "auto" and self.num_clusters >= 64)

    def resident_ok(self, data) -> bool:
        """Can ``forward_graphs`` take the fused graph-resident path for this input?"""
        if self._dense():
            return False          # the dense route is the layered operators + the MFMA contractions
        layers = list(self.mlp)
        if self.mp.num != 1 or len(layers) != 1:
            return False
        conv = self.mp.module_0
        H, F = conv.lin_rel.weight.shape
        K = layers[0].weight.shape[0]
        meta = _engine.scn_meta(data, conv.lin_rel.weight.device)
        from .._hip import lib
        return bool(lib().hscn_scn_resident_supported(F, H, K, meta.max_n, meta.max_e))

    def forward_graphs(self, data, with_total: bool = False):
        """The body of the reference's clustering loop (train/train_clustering.py:37-47) for one
        ``Data`` graph or a block-diagonal ``Batch`` of RAW graphs: gcn_norm(add_self_loops=True),
        forward, MinCUT + orthogonality losses -- one fused launch when the model is the
        reference's shape (mp_units=[H], mlp_units=[]), else the layered operators.
        Returns ``(softmax [N,K], mc_loss, o_loss)`` (losses = mean over the graphs); with
        ``with_total`` also ``mc_loss + o_loss`` (train_clustering.py:48), which the fused launch
        produces itself -- no add launch, and its backward reaches the kernel as one scalar."""
        dev = self.mp.module_0.lin_rel.weight.device
        if self.resident_ok(data):
            conv, lin = self.mp.module_0, list(self.mlp)[0]
            meta = _engine.scn_meta(data, dev)
            x = data.x if data.x.is_cuda else data.x.to(dev)
            ei = data.edge_index if data.edge_index.is_cuda else data.edge_index.to(dev)
            if x.dtype != torch.float16:      # (half features stay half: include/hscn.h, hscn_scn_resident_*_f16)
                x = x.float()
            S, mc, o, total = _engine.SCNResidentFn.apply(x, ei, meta, _engine.ACT[self.mp.act],
                                                          conv.lin_rel.weight, conv.lin_rel.bias, conv.lin_root.weight,
                                                          lin.weight, lin.bias)
            self.last_engine = "resident"
            return (S, mc, o, total) if with_total else (S, mc, o)
        from ..nn.pool import gcn_norm, gcn_norm_static
        self.last_engine = "layered"
        if data.x.dtype == torch.float16:
            raise RuntimeError("half-precision feature storage runs on the fused stage-A launch only (mp_units=[H], "
                               "mlp_units=[], graphs that fit one CU's LDS)")
        raw_ei = data.edge_index if data.edge_index.is_cuda else data.edge_index.to(dev)
        N = int(data.num_nodes)
        batched = "ptr" in data and data.ptr is not None
        if not self._dense():
            ei, ew = gcn_norm(raw_ei, None, N, add_self_loops=True)
            node_ptr = data.ptr.to(dev).to(torch.int32) if batched else None
            S, mc, o, _ = self.forward(data.x.to(dev).float(), ei, ew, node_ptr=node_ptr)
            return (S, mc, o, mc + o) if with_total else (S, mc, o)
        # dense route: gcn_norm(add_self_loops=True) in its static-shape form and A + I straight from the raw edges --
        # every launch has a shape known on the host, so the whole step (forward, losses, backward) can be captured
        # and replayed as one hipGraph; graphs of any sizes share a batch
        ei, ew = gcn_norm_static(raw_ei, None, N)
        cached = getattr(data, "_dense_seg", None)
        if cached is None or cached[0].device != dev:
            if batched:
                cached = (data.ptr.to(dev).to(torch.int32).contiguous(), data.batch.to(dev).to(torch.int32).contiguous(),
                          int(data.max_nodes))
            else:
                cached = (torch.tensor([0, N], dtype=torch.int32, device=dev),
                          torch.zeros(N, dtype=torch.int32, device=dev), N)
            try:
                data._dense_seg = cached
            except AttributeError:
                pass
        node_ptr, gid, nmax = cached
        S, mc, o, _ = self.forward(data.x.to(dev).float(), ei, ew, node_ptr=node_ptr, nodes_per_graph=nmax,
                                   node_graph=gid, raw_edge_index=raw_ei)
        return (S, mc, o, mc + o) if with_total else (S, mc, o)

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor],
                node_ptr: Optional[Tensor] = None, nodes_per_graph: Optional[int] = None,
                node_graph: Optional[Tensor] = None, raw_edge_index: Optional[Tensor] = None):
        """Batched calls of the dense route (extensions; the reference's own call is one graph at a time):
        ``nodes_per_graph`` = the LARGEST node count of the batch's graphs (the common one when they are equal),
        ``node_graph`` int32 [N] = graph of every node -- needed when the graphs differ in size: the adjacency is then
        [B, nmax, nmax] with zeros beyond each graph, node-indexed tensors stay flat, and every graph's loss terms
        equal the single-graph call's.  ``raw_edge_index``: the edge list BEFORE gcn_norm added self loops, when
        ``edge_index`` came from ``gcn_norm_static`` (whose zero-weight placeholders must not be counted)."""
        n = x.size(0)
        rel = relation_of(edge_index, n, n)
        x = self.mp(x.float(), rel, edge_weight)
        s = x
        layers = list(self.mlp)
        for i, m in enumerate(layers):
            if isinstance(m, Linear):
                last = i == len(layers) - 1
                s = m(s, act="identity" if last else self.mlp_act)
        if self._dense():
            # model/hscn.py:61-63 literally: adj = to_dense_adj(edge_index); dense_mincut_pool(x, adj, s)
            if node_ptr is None:
                Bg, ng = 1, n
            else:
                Bg = int(node_ptr.numel()) - 1
                ng = int(nodes_per_graph) if nodes_per_graph else 0
                if ng <= 0:
                    raise ValueError("the dense MinCUT route on a batch needs nodes_per_graph (the largest graph's node "
                                     "count: it sizes the [B,n,n] adjacency)")
            if node_ptr is not None and (Bg * ng != n or raw_edge_index is not None):
                if node_graph is None:
                    raise ValueError("graphs of different sizes on the dense MinCUT route need node_graph (int32 [N])")
                # ragged batch: A + I of every graph in its own [nmax, nmax] block, node-indexed tensors flat
                # (the adjacency as bytes: nobody outside this call sees it -- batched calls return None in its slot --
                # and the products that stream it move a quarter of the bytes; HSCN_DENSE_ADJ=f32 keeps floats)
                as_bytes = os.environ.get("HSCN_DENSE_ADJ", "u8") != "f32"
                want_sym = as_bytes and os.environ.get("HSCN_DENSE_SYM", "1") != "0"     # (A/B: 0 = always compute A^T S)
                if raw_edge_index is not None:
                    adj = to_dense_adj_ragged(raw_edge_index, node_ptr, node_graph, Bg, ng, raw=True, as_bytes=as_bytes,
                                              symmetry=want_sym)
                else:
                    adj = to_dense_adj_ragged(edge_index, node_ptr, node_graph, Bg, ng, as_bytes=as_bytes, symmetry=want_sym)
                # byte route: one pass flags the graphs whose adjacency is not symmetric; the others (undirected graphs:
                # the norm) take the backward's A^T S from the forward's A S
                asym = None
                if want_sym:
                    adj, asym = adj
                S, mc_loss, o_loss, _, _ = Fh.MinCutDenseRaggedFn.apply(s, x, adj, node_ptr, node_graph, asym)
                self.last_route = "dense-ragged"
                return S, mc_loss, o_loss, None
            adj = to_dense_adj_batched(edge_index, Bg, ng)
            S, mc_loss, o_loss, _, _ = Fh.MinCutDenseFn.apply(s.view(Bg, ng, -1), x.view(Bg, ng, -1), adj)
            self.last_route = "dense"
            return S.view(n, -1), mc_loss, o_loss, (adj if node_ptr is None else None)
        self.last_route = "sparse"
        S, _, _, mc_loss, o_loss = mincut_pool_sparse(x, rel, s, node_ptr)
        adj = to_dense_adj(edge_index, n) if node_ptr is None else None
        return S, mc_loss, o_loss, adj


def build_conv_relation(conv_type: str, hidden_channels: int, in_channels=None) -> nn.Module:
    """hscn.py:117-125.  ``in_channels`` (extension) materialises the lazy ``-1``."""
    if conv_type == "GAT":
        dim = (-1, -1) if in_channels is None else (in_channels, in_channels)
    else:
        dim = -1 if in_channels is None else in_channels
    return CONV_DICT[conv_type.lower()](dim, hidden_channels, add_self_loops=False, cached=False)


class HSCN(nn.Module):
    def __init__(self, lv_conv: str, ll_conv: str, vv_conv: str, activation: Callable, num_features: int,
                 hidden_channels: int, num_classes: int, num_layers: int) -> None:
        super().__init__()
        self.activation = activation
        self.convs = nn.ModuleList()
        for layer in range(num_layers):
            fin = num_features if layer == 0 else hidden_channels
            conv = HeteroConv(
                {
                    LV: build_conv_relation(lv_conv, hidden_channels, fin),
                    LL: build_conv_relation(ll_conv, hidden_channels, fin),
                    VV: build_conv_relation(vv_conv, hidden_channels, fin),
                },
                aggr="sum",
            )
            self.convs.append(conv)
        self.lin_1 = Linear(hidden_channels, hidden_channels)
        self.lin_2 = Linear(hidden_channels, num_classes)
        # execution engine: "auto" picks the graph-resident fused kernels when the batch
        # qualifies (engine.py), else the per-operator (layered) path; both are HIP.
        self.engine = "auto"
        self.compute_virtual = True   # the reference evaluates the virtual branch although pred ignores it
        self.keep_virtual = False     # expose the final virtual features as self.last_virtual
        # resident engine: run the virtual branch as its own launch beside loss + backward (engine.py)
        self.overlap_virtual = os.environ.get("HSCN_OVERLAP_VIRTUAL", "1") != "0"
        self.last_virtual: Optional[Tensor] = None
        self.last_engine: Optional[str] = None

    def _resident_plan(self, x_dict, edge_index_dict, batch):
        if self.engine == "layered":
            return None
        name = _act_name(self.activation)
        ok = (name in ACT_DICT and set(edge_index_dict) == {LL, VV, LV} and "local" in x_dict
              and "virtual" in x_dict and x_dict["local"].is_cuda)
        ok = ok and self._resident_params() is not None
        meta = _engine.meta_from_batch(batch, x_dict["local"].device) if ok else None
        H, C = self.lin_1.out_channels, self.lin_2.out_channels
        if meta is None or not _engine.supported(x_dict["local"].size(1), H, len(self.convs), C, meta,
                                                 x_dict["local"].dtype):
            if self.engine == "resident":
                raise RuntimeError("engine='resident' requested but the batch/model does not qualify "
                                   "(needs a graph_hscn HeteroBatch, GAT/GCN/GCN relations, H in {16,32,64}, "
                                   "F <= H and graphs that fit one CU's LDS)")
            return None
        return meta, name

    def _resident_params(self):
        """The parameter order the resident launches take, or None when the relations are not the
        GAT / GCN / GCN combination; looked up once (the eager path is host-bound), rebuilt when the module
        tree changes (load_state_dict keeps the Parameter objects, assigning new modules does not)."""
        key = tuple(id(m) for conv in self.convs for m in conv.convs.values()) + (id(self.lin_1.weight), id(self.lin_2.weight))
        cached = getattr(self, "_resident_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        params = []
        for conv in self.convs:
            c = conv.convs
            ll, vv, lv = (c["__".join(k)] for k in (LL, VV, LV))
            if not (isinstance(ll, GCNConv) and isinstance(vv, GCNConv) and isinstance(lv, GATConv)):
                params = None
                break
            params += [ll.lin.weight, ll.bias, vv.lin.weight, vv.bias, lv.lin_src.weight, lv.lin_dst.weight,
                       lv.att_src, lv.att_dst, lv.bias]
        if params is not None:
            params += [self.lin_1.weight, self.lin_1.bias, self.lin_2.weight, self.lin_2.bias]
        object.__setattr__(self, "_resident_cache", (key, params))
        return params

    def _forward_resident(self, x_dict, edge_index_dict, meta, act_name) -> Tensor:
        params = self._resident_params()
        slope = self.convs[0].convs["__".join(LV)].negative_slope
        cfg = (_engine.ACT[act_name], slope, self.compute_virtual, self.keep_virtual, self.overlap_virtual)
        out = _engine.HSCNResidentFn.apply(x_dict["local"], x_dict["virtual"], edge_index_dict[LL],
                                           edge_index_dict[VV], edge_index_dict[LV], meta, cfg, *params)
        out, xv, score = out
        if xv is not None:
            self.last_virtual = xv
        if score is not None:
            # loss.criterion recognises a prediction that comes with its score and lets the loss tail ride on
            # the backward launch (no launch of its own)
            out._hscn_score = (score, out._version)
        return out

    def forward(self, x_dict: Dict[str, Tensor], edge_index_dict: Dict[Tuple[str, str, str], Tensor],
                batch) -> Tensor:
        plan = self._resident_plan(x_dict, edge_index_dict, batch)
        if plan is not None:
            self.last_engine = "resident"
            return self._forward_resident(x_dict, edge_index_dict, *plan)
        self.last_engine = "layered"
        if x_dict["local"].dtype == torch.float16:
            raise RuntimeError("half-precision feature storage runs on the graph-resident engine only (H in {16, 32}, "
                               "a graph_hscn HeteroBatch, graphs that fit one CU's LDS)")
        relu = ACT_DICT["relu"]
        for conv in self.convs:
            x_dict = conv(x_dict, edge_index_dict)
            x_dict = {key: relu(x) for key, x in x_dict.items()}           # hscn.py:110 (hard-coded ReLU)
        local = batch["local"]
        size = getattr(batch, "num_graphs", None) or None
        x = global_mean_pool(x_dict["local"], local.batch, size)           # hscn.py:111
        name = _act_name(self.activation)
        if name is not None:
            x = self.lin_1(x, act=name)                                    # hscn.py:112 fused epilogue
        else:
            x = self.activation(self.lin_1(x))
        return self.lin_2(x)                                               # hscn.py:113


def build_hscn(model_cfg: HSCNConfig, num_features: int, num_classes: int) -> HSCN:
    return HSCN(
        model_cfg.lv_conv_type,
        model_cfg.ll_conv_type,
        model_cfg.vv_conv_type,
        ACT_DICT[model_cfg.activation.lower()],
        num_features,
        model_cfg.hidden_channels,
        num_classes,
        model_cfg.num_layers,
    )
