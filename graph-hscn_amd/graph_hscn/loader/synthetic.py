"""Seeded synthetic LRGB-shaped graphs.

The reference's datasets (graph_hscn/loader/dataset/peptides_functional.py:21-115,
peptides_structural.py:21-121) need network + rdkit + ogb; none exist on the
build or GPU boxes.  These generators reproduce only the *shape* statistics of
the LRGB sets: node-count distribution, directed edge
count, feature width/type, label width.  Edge lists follow the OGB
``smiles2graph`` layout: undirected bonds emitted as adjacent (i,j),(j,i) pairs,
no self loops, no duplicates.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from ..data import Data

# OGB atom feature cardinalities (9 integer columns)
_ATOM_CARD = np.array([119, 5, 12, 12, 10, 6, 6, 2, 2])


@dataclass(frozen=True)
class Shape:
    name: str
    n_mean: float
    n_std: float
    n_min: int
    n_max: int
    und_per_node: float      # undirected edges per node (directed = 2x)
    num_features: int
    feature_kind: str        # "atom" (int columns) | "normal" (float)
    num_classes: int
    task: str                # "multilabel" | "regression"


SHAPES = {
    # n ~ 150.94, e ~ 307.30 directed
    "peptides_func": Shape("peptides_func", 151.0, 84.0, 8, 444, 1.018, 9, "atom", 10, "multilabel"),
    "peptides_struct": Shape("peptides_struct", 151.0, 84.0, 8, 444, 1.018, 9, "atom", 11, "regression"),
    # n ~ 479.40, e ~ 2710.48 directed
    "pascalvoc_sp": Shape("pascalvoc_sp", 479.0, 60.0, 395, 500, 2.827, 14, "normal", 21, "multilabel"),
    # n ~ 30.14, e ~ 61.09 directed
    "pcqm_contact": Shape("pcqm_contact", 30.0, 8.0, 9, 53, 1.013, 9, "atom", 1, "regression"),
}


def _molecule_edges(rng: np.random.Generator, n: int, n_und: int) -> np.ndarray:
    """Random tree with SMILES-like locality plus ring closures -> [2, 2*m]."""
    und = set()
    for i in range(1, n):
        if rng.random() < 0.8:
            p = i - 1
        else:
            p = int(rng.integers(max(0, i - 12), i))
        und.add((p, i))
    tries = 0
    while len(und) < n_und and tries < 20 * n_und and n > 3:
        tries += 1
        i = int(rng.integers(0, n - 2))
        j = i + int(rng.integers(2, min(8, n - i)))
        if j < n:
            und.add((i, j))
    und = sorted(und, key=lambda e: (e[1], e[0]))
    ei = np.empty((2, 2 * len(und)), dtype=np.int64)
    for k, (i, j) in enumerate(und):
        ei[:, 2 * k] = (i, j)
        ei[:, 2 * k + 1] = (j, i)
    return ei


def _lattice_edges(rng: np.random.Generator, n: int, n_und: int) -> np.ndarray:
    """Superpixel-adjacency-like graph: nodes on a ~sqrt(n) wide strip, edges to
    near neighbours in index space."""
    w = max(2, int(round(np.sqrt(n))))
    und = set()
    for i in range(n):
        if i + 1 < n and (i + 1) % w:
            und.add((i, i + 1))
        if i + w < n:
            und.add((i, i + w))
    tries = 0
    while len(und) < n_und and tries < 20 * n_und:
        tries += 1
        i = int(rng.integers(0, n - 1))
        j = i + int(rng.choice([w - 1, w + 1, 2, 2 * w]))
        if j < n:
            und.add((i, j))
    und = sorted(und)
    ei = np.empty((2, 2 * len(und)), dtype=np.int64)
    for k, (i, j) in enumerate(und):
        ei[:, 2 * k] = (i, j)
        ei[:, 2 * k + 1] = (j, i)
    return ei


def make_graph(rng: np.random.Generator, shape: Shape, n: Optional[int] = None) -> Data:
    if n is None:
        n = int(np.clip(round(rng.normal(shape.n_mean, shape.n_std)), shape.n_min, shape.n_max))
    n_und = max(n - 1, int(round(shape.und_per_node * n)))
    if shape.feature_kind == "atom":
        ei = _molecule_edges(rng, n, n_und)
        x = torch.from_numpy(
            np.stack([rng.integers(0, c, size=n) for c in _ATOM_CARD[: shape.num_features]], 1).astype(np.int64))
    else:
        ei = _lattice_edges(rng, n, n_und)
        x = torch.from_numpy(rng.normal(size=(n, shape.num_features)).astype(np.float32))
    if shape.task == "multilabel":
        y = torch.from_numpy((rng.random((1, shape.num_classes)) < 0.2).astype(np.float32))
    else:
        y = torch.from_numpy(rng.normal(size=(1, shape.num_classes)).astype(np.float32))
    return Data(x=x, edge_index=torch.from_numpy(ei), y=y, num_nodes=n)


def make_dataset(name: str, num_graphs: int, seed: int = 0) -> List[Data]:
    """``num_graphs`` seeded graphs of the named LRGB shape."""
    shape = SHAPES[name]
    rng = np.random.default_rng(seed)
    return [make_graph(rng, shape) for _ in range(num_graphs)]
