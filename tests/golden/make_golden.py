#!/usr/bin/env python3
"""Generate the golden vectors in this directory FROM THE CPU ORACLE.

The reference has no tests, fixtures or golden vectors for this path and its
third-party arithmetic (torch_geometric) cannot be imported here (ordinary
ModuleNotFoundError), so these files pin the oracle against regressions only
("parity unpinned", oracle/__init__.py).  Closed-form cases in
tests/test_oracle_kat.py pin its semantics.

  python tests/golden/make_golden.py      # rewrites *.npz next to this file
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]

from graph_hscn.loader.synthetic import make_dataset  # noqa: E402
from oracle import hetero_data as OH  # noqa: E402
from oracle import models as OM  # noqa: E402


def flat_state(m):
    return {f"w::{k}": v.detach().numpy() for k, v in m.state_dict().items()}


def scn_case(seed=0, K=16):
    torch.manual_seed(seed)
    g = make_dataset("peptides_func", 1, seed=seed + 10)[0]
    m = OM.SCN([16], "elu", 9, K)
    S, mc, o, adj, ei, ew = OM.scn_step_single_graph(m, g.x, g.edge_index)
    (mc + o).backward()
    out = dict(x=g.x.numpy(), edge_index=g.edge_index.numpy(), ei_norm=ei.numpy(), ew_norm=ew.detach().numpy(),
               S=S.detach().numpy(), mc=mc.detach().numpy(), o=o.detach().numpy(),
               clusters=OM.assign_clusters(S), K=np.int64(K))
    out.update(flat_state(m))
    out.update({f"g::{k}": p.grad.numpy() for k, p in m.named_parameters()})
    np.savez_compressed(os.path.join(HERE, "scn_peptides_k16.npz"), **out)


def hscn_case(seed=1, B=4, K=8, H=16, L=3, C=10):
    torch.manual_seed(seed)
    graphs = make_dataset("peptides_func", B, seed=seed + 20)
    rng = np.random.default_rng(seed)
    ids = [rng.integers(0, K, g.num_nodes) for g in graphs]
    hs = [OH.hetero_from_clusters(g.x, g.edge_index, g.y, i, K) for g, i in zip(graphs, ids)]
    b = OH.collate_hetero(hs)
    m = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, H, C, L)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("bias"):
                p.normal_(0, 0.1)
    pred = m(b["x_dict"], b["edge_index_dict"], b["batch_local"], B)
    loss, _ = OM.criterion("cross_entropy", pred, b["y"])
    loss.backward()
    xo = b["x_dict"]
    for conv in m.convs:
        xo = {k: v.relu() for k, v in conv(xo, b["edge_index_dict"]).items()}
    out = dict(num_nodes=np.array([g.num_nodes for g in graphs]),
               x=np.concatenate([g.x.numpy() for g in graphs]),
               edge_index=np.concatenate([g.edge_index.numpy() for g in graphs], 1),   # per-graph local ids
               num_edges=np.array([g.num_edges for g in graphs]),
               y=b["y"].numpy(), clusters=np.concatenate(ids), K=np.int64(K),
               virtual_x=b["x_dict"]["virtual"].numpy(), ei_vv=b["edge_index_dict"][OH.VV].numpy(),
               ei_lv=b["edge_index_dict"][OH.LV].numpy(), pred=pred.detach().numpy(), loss=loss.detach().numpy(),
               final_local=xo["local"].detach().numpy(), final_virtual=xo["virtual"].detach().numpy())
    out.update(flat_state(m))
    out.update({f"g::{k}": p.grad.numpy() for k, p in m.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(HERE, "hscn_peptides_b4.npz"), **out)


def mpnn_case(seed=2, B=4, H=16, L=3, C=10):
    """The GCN baseline of BASELINE config 1 (model/mpnn.py) in eval mode on a 4-graph Peptides batch."""
    torch.manual_seed(seed)
    graphs = make_dataset("peptides_func", B, seed=seed + 30)
    ns = [g.num_nodes for g in graphs]
    off = np.concatenate([[0], np.cumsum(ns)])
    x = torch.cat([g.x for g in graphs]).float()
    ei = torch.cat([g.edge_index + int(off[i]) for i, g in enumerate(graphs)], 1)
    batch = torch.repeat_interleave(torch.arange(B), torch.as_tensor(ns))
    y = torch.cat([g.y.view(1, -1) for g in graphs]).float()
    m = OM.MPNN(OM.ACT["relu"], 9, H, C, L, dropout=0.2).eval()
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("bias"):
                p.normal_(0, 0.1)
    pred = m(x, ei, batch, B)
    loss, _ = OM.criterion("cross_entropy", pred, y)
    loss.backward()
    out = dict(num_nodes=np.array(ns), x=x.numpy(), edge_index=ei.numpy(), batch=batch.numpy(), y=y.numpy(),
               pred=pred.detach().numpy(), loss=loss.detach().numpy())
    out.update(flat_state(m))
    out.update({f"g::{k}": p.grad.numpy() for k, p in m.named_parameters()})
    np.savez_compressed(os.path.join(HERE, "mpnn_gcn_peptides_b4.npz"), **out)


if __name__ == "__main__":
    only = sys.argv[1:]                     # e.g. `make_golden.py mpnn` rewrites that file alone
    if not only or "scn" in only:
        scn_case()
    if not only or "hscn" in only:
        hscn_case()
    if not only or "mpnn" in only:
        mpnn_case()
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])
