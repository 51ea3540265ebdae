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
    buf = torch.zeros(2 * B, 64, dtype=torch.int64, device=dev)
    L.hscn_diag_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    assert L.hscn_diag_set_stamp_buffer(buf.data_ptr()) == 0
    sizes = np.diff(hb_host["local"].ptr.numpy())
    fn = {0: "start", 1: "prologue loads", 2: "CSRs side by side", 3: "barrier + export"}
    for l in range(3):
        fn.update({4 + 4 * l: f"L{l} begin", 5 + 4 * l: f"L{l} transforms", 6 + 4 * l: f"L{l} reduce"})
    fn[62] = "xv_out"
    fkeys = sorted(fn)

    def show(st, rows, names, keys, last, tag):
        total = st[rows, last] - st[rows, 0]
        order = np.argsort(total)
        for t, i in ((tag + " slowest", order[-1]), (tag + " median", order[len(order) // 2])):
            r = rows[i]
            print(f"--- {t}: graph {i} n={sizes[i]} total {total[i]} cycles")
            prev = st[r, 0]
            for k in keys[1:]:
                if k > last:
                    break
                d = st[r, k] - prev
                prev = st[r, k]
                print(f"   {names[k]:26s} {d:8d} cyc  {100.0 * d / total[i]:5.1f}%")

    rows = np.arange(B)
    # ---- one-launch forward (both branches) ----
    model.overlap_virtual = False
    for _ in range(3):
        with torch.no_grad():
            model(hb.x_dict, hb.edge_index_dict, hb)
    torch.cuda.synchronize()
    show(buf.cpu().numpy(), rows, {**fn, 63: "pool+head"}, fkeys + [63], 63, "fwd fused")
    # ---- training step: local-only forward, then backward carrying the virtual branch ----
    model.overlap_virtual = True
    buf.zero_()
    out = model(hb.x_dict, hb.edge_index_dict, hb)
    torch.cuda.synchronize()
    st = buf.cpu().numpy()
    show(st, 2 * rows, {**fn, 63: "pool+head"}, fkeys + [63], 63, "fwd local chain (even blocks of the forward launch)")
    show(st, 2 * rows + 1, fn, fkeys, 62, "virtual part 1: CSRs + layer 0 (odd blocks of the forward launch)")
    lr = 2 * rows
    tot = st[lr, 63] - st[lr, 0]
    for tag, i in (("slowest", np.argsort(tot)[-1]), ("median", np.argsort(tot)[len(tot) // 2])):
        r = lr[i]
        print(f"--- local chain {tag} n={sizes[i]}: per layer, cycles since the layer's begin stamp")
        for l in range(3):
            b0 = st[r, 4 + 4 * l]
            print(f"   L{l}: wave0 transforms done {st[r, 41 + 4 * l] - b0:6d} | weights parked {st[r, 40 + 4 * l] - b0:6d} | barrier {st[r, 5 + 4 * l] - b0:6d} | "
                  f"wave0 reduce done {st[r, 43 + 4 * l] - b0:6d} | layer end {st[r, 6 + 4 * l] - b0:6d}")
    buf.zero_()
    out.sum().backward()
    torch.cuda.synchronize()
    st = buf.cpu().numpy()
    show(st, 2 * rows + 1, fn, fkeys, 62, "virtual part 2: layers 1.. (odd blocks of the backward launch)")
    vr = 2 * rows + 1
    tot = st[vr, 62] - st[vr, 0]
    for tag, i in (("slowest", np.argsort(tot)[-1]), ("median", np.argsort(tot)[len(tot) // 2])):
        r = vr[i]
        print(f"--- virtual-only {tag} n={sizes[i]}: per layer, cycles since the layer's begin stamp")
        for l in range(1, 3):
            b0 = st[r, 4 + 4 * l]
            print(f"   L{l}: B transforms done {st[r, 40 + 4 * l] - b0:6d} | A phase-1 done {st[r, 41 + 4 * l] - b0:6d} | "
                  f"barrier {st[r, 5 + 4 * l] - b0:6d} | B(wave0) reduce done {st[r, 42 + 4 * l] - b0:6d} | "
                  f"A phase-2 done {st[r, 43 + 4 * l] - b0:6d} | layer end {st[r, 6 + 4 * l] - b0:6d}")
    bn = {0: "start", 1: "prologue loads", 2: "dinv + csr^T", 3: "head bwd"}
    for l in (2, 1, 0):
        bn.update({4 + 4 * l: f"L{l} load X,W + bias + A^T G", 5 + 4 * l: f"L{l} gW", 6 + 4 * l: f"L{l} gX"})
    bn[63] = "end"
    order_keys = [0, 1, 2, 3] + [k + 4 * l for l in (2, 1, 0) for k in (4, 5, 6)] + [63]
    show(st, 2 * rows, bn, order_keys, 63, "bwd (even blocks)")


if __name__ == "__main__":
    main()
