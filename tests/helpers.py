"""Shared builders for the parity tests (seeded inputs, oracle <-> product weights)."""
import numpy as np
import torch

from graph_hscn.loader.synthetic import make_dataset
from oracle import hetero_data as OH

DEV = "cuda"
ATOL = 1e-5      # north_star: float activations within 1e-5 of the CPU path
RTOL = 1e-5


def rand_graph(n, e, seed, symmetric=False, self_loops=False):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    if not self_loops:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    ei = torch.stack([src, dst])
    if symmetric:
        ei = torch.cat([ei, ei.flip(0)], 1)
    return ei


def hetero_batch(name, num_graphs, K, seed):
    graphs = make_dataset(name, num_graphs, seed=seed)
    rng = np.random.default_rng(seed)
    hs = [OH.hetero_from_clusters(g.x, g.edge_index, g.y, rng.integers(0, K, g.num_nodes), K) for g in graphs]
    return OH.collate_hetero(hs), graphs


def close(a, b, atol=ATOL, rtol=RTOL):
    a = a.detach().cpu()
    b = b.detach().cpu()
    ok = torch.allclose(a, b, atol=atol, rtol=rtol)
    if not ok:
        d = (a - b).abs()
        print("max abs diff", float(d.max()), "at", int(d.argmax()), "ref", float(b.flatten()[d.argmax()]))
    return ok
