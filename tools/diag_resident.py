#!/usr/bin/env python3
"""Phase timeline of the graph-resident forward kernel (diagnostic build only).

  make -C graph-hscn_amd diag && HSCN_LIB=graph-hscn_amd/graph_hscn/lib/libhscn_diag.so python tools/diag_resident.py

Reads the per-workgroup clock64() stamps the -DHSCN_STAMPS build writes and prints,
for the slowest workgroup and the median one, where the cycles go.  Read SHARES,
not totals: the stamped build is not the shipped kernel.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import numpy as np
import torch

import bench
from graph_hscn import _hip
from graph_hscn.config.config import ACT_DICT
from graph_hscn.model.hscn import HSCN


def main():
    dev = torch.device("cuda:0")
    L = _hip.lib()
    hb_host, graphs, _ = bench.build_hetero_batch("peptides_func", 128, 16, 0, dev)
    hb = hb_host.to(dev)
    torch.manual_seed(0)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(dev)
    model.engine = "resident"
    B = hb.num_graphs
    buf = torch.zeros(B, 64, dtype=torch.int64, device=dev)
    L.hscn_diag_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    assert L.hscn_diag_set_stamp_buffer(buf.data_ptr()) == 0
    for _ in range(5):
        with torch.no_grad():
            model(hb.x_dict, hb.edge_index_dict, hb)
    torch.cuda.synchronize()
    st = buf.cpu().numpy()
    sizes = np.diff(hb_host["local"].ptr.numpy())
    total = st[:, 63] - st[:, 0]
    order = np.argsort(total)
    names = {0: "start", 1: "prologue loads", 2: "4 CSRs side by side", 3: "barrier + export"}
    for l in range(3):
        names.update({4 + 4 * l: f"L{l} begin", 5 + 4 * l: f"L{l} transforms (A: ll | B: lv,vv)",
                      6 + 4 * l: f"L{l} reduce (A: ll | B: gat+vv)"})
    names[63] = "pool+head"
    keys = sorted(names)
    for tag, g in (("slowest", order[-1]), ("median", order[len(order) // 2]), ("fastest", order[0])):
        print(f"--- {tag}: graph {g} n={sizes[g]} total {total[g]} cycles")
        prev = st[g, 0]
        for k in keys[1:]:
            d = st[g, k] - prev
            prev = st[g, k]
            print(f"   {names[k]:22s} {d:8d} cyc  {100.0 * d / total[g]:5.1f}%")
    # ---- backward kernel (its stamps overwrite the forward's slots 0..2, 3+6l..6+6l, 63) ----
    buf.zero_()
    out = model(hb.x_dict, hb.edge_index_dict, hb)
    out.sum().backward()
    torch.cuda.synchronize()
    st = buf.cpu().numpy()
    bn = {0: "start", 1: "prologue loads", 2: "dinv + csr^T", 3: "head bwd"}
    for l in (2, 1, 0):
        bn.update({4 + 4 * l: f"L{l} load X,W + bias + A^T G", 5 + 4 * l: f"L{l} gW", 6 + 4 * l: f"L{l} gX"})
    bn[63] = "end"
    order_keys = [0, 1, 2, 3] + [k + 4 * l for l in (2, 1, 0) for k in (4, 5, 6)] + [63]
    total = st[:, 63] - st[:, 0]
    order = np.argsort(total)
    for tag, g in (("bwd slowest", order[-1]), ("bwd median", order[len(order) // 2])):
        print(f"--- {tag}: graph {g} n={sizes[g]} total {total[g]} cycles")
        prev = st[g, 0]
        for k in order_keys[1:]:
            d = st[g, k] - prev
            prev = st[g, k]
            print(f"   {bn[k]:24s} {d:8d} cyc  {100.0 * d / total[g]:5.1f}%")


if __name__ == "__main__":
    main()
