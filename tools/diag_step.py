#!/usr/bin/env python3
"""Phase timeline of the one-launch training step (diagnostic build only).

  make -C graph-hscn_amd diag && HSCN_LIB=graph-hscn_amd/graph_hscn/lib/libhscn_diag.so python tools/diag_step.py [uniform]

Reads the per-workgroup clock64() stamps the -DHSCN_STAMPS build writes (csrc/resident_step.h for the local
workgroups = blocks [0, B); csrc/resident_kernels.h hscn_fwd_body for the virtual workgroups = blocks [B, 2B)) and
prints where the cycles go for the slowest and the median workgroup of each kind.  Read SHARES, not totals.
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
from graph_hscn.step import ResidentTrainStep


def main():
    dev = torch.device("cuda:0")
    lib = _hip.lib()
    ids = "uniform" if "uniform" in sys.argv[1:] else "scn_untrained"
    wl = next((w for w in bench.WORKLOADS if w in sys.argv[1:]), "peptides_func")      # e.g. peptides_struct, pcqm_contact
    shape, B0, K0, C0, loss0 = bench.WORKLOADS[wl]
    hb_host, graphs, _ = bench.build_hetero_batch(shape, B0, K0, 0, dev, ids)
    hb = hb_host.to(dev)
    if "f16" in sys.argv[1:]:
        hb = hb.with_feature_dtype(torch.float16)
    torch.manual_seed(0)
    Lyr = 3
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], hb_host["local"].x.size(1), 16, C0, Lyr).to(dev)
    B = hb.num_graphs
    buf = torch.zeros(2 * B, 64, dtype=torch.int64, device=dev)
    lib.hscn_diag_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    assert lib.hscn_diag_set_stamp_buffer(buf.data_ptr()) == 0
    sizes = np.diff(hb_host["local"].ptr.numpy())
    structure = None
    if "resident" in sys.argv[1:]:
        from graph_hscn.engine import build_structure
        structure = build_structure(hb)
    rs = ResidentTrainStep(model, hb, loss0, one_launch=True, structure=structure)
    for _ in range(3):
        rs.run()
    torch.cuda.synchronize()
    buf.zero_()
    rs.run()
    torch.cuda.synchronize()
    st = buf.cpu().numpy()
    names = {0: "start", 1: "prologue + stage", 3: "two CSRs (two-barrier build)"}
    keys = [0, 1, 3]
    for l in range(Lyr):
        names[4 + l] = f"fwd L{l - 1} (+ publish)" if l else "(-)"
        keys.append(4 + l)
    names[12] = f"fwd L{Lyr - 1}"
    names[13] = "head + loss row (wave 0)"
    names[14] = "head bwd + mask"
    keys += [12, 13, 14]
    for l in range(Lyr - 1, -1, -1):          # H = 16: one phase per backward layer
        names[16 + 3 * l] = f"bwd L{l} (bias, A^T G, gX, gW)"
        keys += [16 + 3 * l]
    names[61] = "last barrier + fold"
    keys.append(61)
    rows = np.arange(B)
    total = st[rows, 61] - st[rows, 0]
    order = np.argsort(total)
    t0 = st[:, 0][st[:, 0] > 0].min()
    for tag, i in (("slowest", order[-1]), ("median", order[len(order) // 2])):
        print(f"--- local workgroup {tag}: graph {i} n={sizes[i]} total {total[i]} cycles (start +{st[i, 0] - t0})")
        prev = st[i, 0]
        for k in keys[1:]:
            d = st[i, k] - prev
            prev = st[i, k]
            print(f"   {names[k]:26s} {d:8d} cyc  {100.0 * d / total[i]:5.1f}%")
    vr = B + rows
    vt = st[vr, 62] - st[vr, 0]
    vo = np.argsort(vt)
    fn = {0: "start", 1: "prologue loads", 2: "CSRs side by side", 3: "barrier"}
    for l in range(Lyr):
        fn.update({4 + 4 * l: f"L{l} begin", 5 + 4 * l: f"L{l} phase 1", 6 + 4 * l: f"L{l} phase 2"})
    fn[62] = "xv_out"
    for tag, i in (("slowest", vo[-1]), ("median", vo[len(vo) // 2])):
        r = vr[i]
        print(f"--- virtual workgroup {tag}: graph {i} n={sizes[i]} total {vt[i]} cycles (start +{st[r, 0] - t0}, "
              f"ends {st[r, 62] - st[i, 61]:+d} cycles relative to its local workgroup's end)")
        prev = st[r, 0]
        for k in sorted(fn)[1:]:
            d = st[r, k] - prev
            prev = st[r, k]
            print(f"   {fn[k]:26s} {d:8d} cyc  {100.0 * d / vt[i]:5.1f}%")
    print(f"launch span: first start -> last local end {st[rows, 61].max() - t0} cycles, -> last virtual end {st[vr, 62].max() - t0} cycles")


if __name__ == "__main__":
    main()
