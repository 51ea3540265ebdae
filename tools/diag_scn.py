#!/usr/bin/env python3
"""Phase timeline of the stage-A resident kernels (diagnostic build only).

  make -C graph-hscn_amd diag && HSCN_LIB=graph-hscn_amd/graph_hscn/lib/libhscn_diag.so python tools/diag_scn.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-hscn_amd")]
import numpy as np
import torch

from graph_hscn import _hip
from graph_hscn.data import Batch
from graph_hscn.loader.synthetic import make_dataset
from graph_hscn.model.hscn import SCN


def main():
    dev = torch.device("cuda:0")
    L = _hip.lib()
    graphs = make_dataset("peptides_func", 128, seed=0)
    big = Batch.from_data_list(graphs).to(dev)
    big.x = big.x.float()
    torch.manual_seed(1)
    scn = SCN([16], "elu", 9, 16).to(dev)
    B = 128
    buf = torch.zeros(B, 64, dtype=torch.int64, device=dev)
    L.hscn_diag_set_stamp_buffer_scn.argtypes = [ctypes.c_void_p]
    assert L.hscn_diag_set_stamp_buffer_scn(buf.data_ptr()) == 0
    sizes = np.diff(big.ptr.cpu().numpy())

    def show(st, names, tag):
        keys = sorted(names)
        total = st[:, 63] - st[:, 0]
        order = np.argsort(total)
        for t, i in ((tag + " slowest", order[-1]), (tag + " median", order[len(order) // 2])):
            print(f"--- {t}: graph {i} n={sizes[i]} total {total[i]} cycles")
            prev = st[i, 0]
            for k in keys[1:]:
                d = st[i, k] - prev
                prev = st[i, k]
                print(f"   {names[k]:34s} {d:8d} cyc  {100.0 * d / total[i]:5.1f}%")

    if "step" in sys.argv[1:]:
        from graph_hscn.step import ScnTrainStep
        pool = None
        if "cached" in sys.argv[1:]:
            from graph_hscn.step import ScnStructurePool
            pool = ScnStructurePool(dev, int(big.num_nodes), int(big.edge_index.size(1)), B)
        st = ScnTrainStep(scn, big, one_launch=True, structure_pool=pool)
        for _ in range(3):
            st.run()
        torch.cuda.synchronize()
        buf.zero_()
        st.run()
        torch.cuda.synchronize()
        show(buf.cpu().numpy(), {0: "start", 1: "requests + park", 2: "two CSRs + degrees", 3: "aggregate",
                                 4: "y = act(W agg + W x), x/agg -> regs", 5: "logits + softmax",
                                 6: "neighbour terms + S^T S", 7: "norms", 13: "Gss",
                                 14: "backward tiles (dS..dW partials)", 15: "park partials", 63: "fold -> HBM"}, "scn step")
        return
    for _ in range(3):
        scn.zero_grad(set_to_none=True)
        t = scn.forward_graphs(big, with_total=True)[3]
    torch.cuda.synchronize()
    show(buf.cpu().numpy(), {0: "start", 1: "requests + park", 2: "two CSRs + degrees", 3: "aggregate + export",
                             4: "y = act(W agg + W x)", 5: "logits + softmax", 6: "mincut stats + S^T S",
                             63: "norms + ticket"}, "scn fwd")
    buf.zero_()
    t.backward()
    torch.cuda.synchronize()
    show(buf.cpu().numpy(), {0: "start", 1: "requests + park", 12: "<Gq, ss>", 13: "Gss", 14: "dS -> dlogits",
                             15: "dW_mlp + db_mlp", 16: "dz", 63: "dW_rel, dW_root, db_rel"}, "scn bwd")


if __name__ == "__main__":
    main()
