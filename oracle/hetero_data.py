"""Restatement of the cluster-ids -> heterogeneous graph transform and of the
PyG mini-batch collate the training loop relies on.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

Follows /root/reference/graph_hscn/loader/hetero_data.py:42-87 literally,
quirks included:
  * slot index is ``clusters[ix] - 1`` (:53) so remapped cluster 0 lands in the
    LAST slot; after empty slots are dropped (:55) virtual node v carries the
    mean of cluster (v+1) mod U while lv edges point at ``clusters[ix]`` (:81-83);
  * vv edges are {(i -> j): i + j <= U-1} (:68-79), U(U+1)/2 of them;
  * means are float64 numpy means of Python float lists cast to float32 (:56-59,66).
Pure-Python per-node loops: use on small cases only.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

LL = ("local", "to", "local")
VV = ("virtual", "to", "virtual")
LV = ("local", "to", "virtual")


def hetero_from_clusters(x: torch.Tensor, edge_index: torch.Tensor, y, clusters_raw: Sequence[int],
                         num_clusters: int) -> Dict:
    """One iteration of the loop body hetero_data.py:42-87 for a single graph.

    ``x`` is the graph's raw node feature tensor (int64 atom features for
    Peptides, cast with ``.float()`` at :64), ``clusters_raw`` the argmax ids
    from stage A (train_clustering.py:68)."""
    num_nodes = x.size(0)
    clust_node: List[list] = [[] for _ in range(num_clusters)]            # :44
    clusters_raw = np.asarray(clusters_raw)
    unique_clusters = np.unique(clusters_raw)                              # :46
    clust_map = {unique_clusters[i]: i for i in range(len(unique_clusters))}
    clusters = [clust_map[v] for v in clusters_raw]                        # :51
    for ix in range(num_nodes):                                            # :52-54
        clust_num = clusters[ix] - 1
        clust_node[clust_num].append(x[ix].tolist())
    clust_node = [lst for lst in clust_node if len(lst) != 0]              # :55
    clust_mean = np.array([np.mean(lst, axis=0) for lst in clust_node])    # :56-59
    num_clust = len(clust_mean)

    col = np.concatenate([[i] * (num_clust - i) for i in range(num_clust)])           # :68-70
    row = np.concatenate([[i for i in range(num_clust - ix)] for ix in range(num_clust)])  # :71-76
    vv = torch.LongTensor([list(col), list(row)])                          # :77-79
    lv = torch.LongTensor([[ix, clusters[ix]] for ix in range(len(clusters))]).T       # :80-86
    return {
        "local_x": x.float(),                                              # :64
        "local_y": y,                                                      # :65
        "virtual_x": torch.FloatTensor(clust_mean),                        # :66
        LL: edge_index,                                                    # :67
        VV: vv,
        LV: lv,
        "num_local": num_nodes,
        "num_virtual": num_clust,
    }


def collate_hetero(graphs: List[Dict]) -> Dict:
    """PyG ``Batch.from_data_list`` for HeteroData: concat x
    per node type, offset every relation's rows/cols by the cumulative node
    counts of its src/dst type, emit per-type ``batch`` vectors and ``ptr``."""
    nl = [g["num_local"] for g in graphs]
    nv = [g["num_virtual"] for g in graphs]
    off_l = np.concatenate([[0], np.cumsum(nl)])
    off_v = np.concatenate([[0], np.cumsum(nv)])
    out = {
        "x_dict": {
            "local": torch.cat([g["local_x"] for g in graphs], 0),
            "virtual": torch.cat([g["virtual_x"] for g in graphs], 0),
        },
        "edge_index_dict": {
            LL: torch.cat([g[LL] + int(off_l[i]) for i, g in enumerate(graphs)], 1),
            VV: torch.cat([g[VV] + int(off_v[i]) for i, g in enumerate(graphs)], 1),
            LV: torch.cat([
                g[LV] + torch.tensor([[int(off_l[i])], [int(off_v[i])]], dtype=torch.long)
                for i, g in enumerate(graphs)], 1),
        },
        "batch_local": torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(nl)]),
        "batch_virtual": torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(nv)]),
        "ptr_local": torch.as_tensor(off_l, dtype=torch.long),
        "ptr_virtual": torch.as_tensor(off_v, dtype=torch.long),
        "num_graphs": len(graphs),
    }
    ys = [g["local_y"] for g in graphs]
    if all(y is not None for y in ys):
        out["y"] = torch.cat([y.view(1, -1) if y.dim() < 2 else y for y in ys], 0)
    return out
