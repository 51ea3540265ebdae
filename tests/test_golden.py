"""Committed golden vectors (tests/golden/*.npz, written by the CPU oracle with
tests/golden/make_golden.py): the oracle must still reproduce them (CPU), and the
HIP path must match them on the GPU box (bit-exact indices, 1e-5 activations)."""
import os

import numpy as np
import pytest
import torch

from oracle import hetero_data as OH
from oracle import models as OM

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(HERE, name)))


def _graphs(z):
    from graph_hscn.data import Data
    out, no, eo = [], 0, 0
    for n, e in zip(z["num_nodes"], z["num_edges"]):
        out.append(Data(x=torch.from_numpy(z["x"][no:no + n]), edge_index=torch.from_numpy(z["edge_index"][:, eo:eo + e]),
                        y=None, num_nodes=int(n)))
        no += n
        eo += e
    return out


def test_oracle_reproduces_scn_golden():
    z = _load("scn_peptides_k16.npz")
    m = OM.SCN([16], "elu", 9, int(z["K"]))
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("w::")})
    S, mc, o, adj, ei, ew = OM.scn_step_single_graph(m, torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"]))
    (mc + o).backward()
    assert np.array_equal(ei.numpy(), z["ei_norm"]) and np.array_equal(OM.assign_clusters(S), z["clusters"])
    np.testing.assert_allclose(S.detach().numpy(), z["S"], atol=1e-6)
    np.testing.assert_allclose(ew.numpy(), z["ew_norm"], atol=0, rtol=0)
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), z[f"g::{k}"], atol=1e-6, rtol=1e-5)


def test_oracle_reproduces_hscn_golden_and_transform():
    z = _load("hscn_peptides_b4.npz")
    graphs = _graphs(z)
    K = int(z["K"])
    ids = np.split(z["clusters"], np.cumsum(z["num_nodes"])[:-1])
    b = OH.collate_hetero([OH.hetero_from_clusters(g.x, g.edge_index, torch.zeros(1, 10), i, K)
                           for g, i in zip(graphs, ids)])
    assert np.array_equal(b["x_dict"]["virtual"].numpy(), z["virtual_x"])          # bit-exact transform
    assert np.array_equal(b["edge_index_dict"][OH.VV].numpy(), z["ei_vv"])
    assert np.array_equal(b["edge_index_dict"][OH.LV].numpy(), z["ei_lv"])
    m = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 3)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("w::")})
    pred = m(b["x_dict"], b["edge_index_dict"], b["batch_local"], len(graphs))
    np.testing.assert_allclose(pred.detach().numpy(), z["pred"], atol=1e-6, rtol=1e-5)


def test_oracle_reproduces_mpnn_golden():
    z = _load("mpnn_gcn_peptides_b4.npz")
    m = OM.MPNN(OM.ACT["relu"], 9, 16, 10, 3, dropout=0.2).eval()
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("w::")})
    pred = m(torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"]), torch.from_numpy(z["batch"]), 4)
    np.testing.assert_allclose(pred.detach().numpy(), z["pred"], atol=1e-6, rtol=1e-5)


@pytest.mark.gpu
def test_hip_mpnn_matches_golden():
    from graph_hscn.config.config import ACT_DICT, CONV_DICT
    from graph_hscn.data import Batch
    from graph_hscn.loss import criterion
    from graph_hscn.model.mpnn import MPNN
    z = _load("mpnn_gcn_peptides_b4.npz")
    b = Batch(x=torch.from_numpy(z["x"]), edge_index=torch.from_numpy(z["edge_index"]), y=torch.from_numpy(z["y"]))
    b.batch, b.num_graphs = torch.from_numpy(z["batch"]), 4
    b = b.to("cuda")
    m = MPNN(CONV_DICT["gcn"], ACT_DICT["relu"], 9, 16, 10, 3, dropout=0.2).to("cuda").eval()
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("w::")})
    pred = m(b)
    loss, _ = criterion("cross_entropy", pred, b.y)
    loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), z["pred"], atol=1e-5, rtol=1e-5)
    assert abs(loss.item() - float(z["loss"])) < 1e-6
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), z[f"g::{k}"], atol=1e-5, rtol=1e-3)


@pytest.mark.gpu
def test_hip_scn_matches_golden():
    from graph_hscn import _hip
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn import gcn_norm
    z = _load("scn_peptides_k16.npz")
    K = int(z["K"])
    m = SCN([16], "elu", 9, K).to("cuda")
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("w::")})
    x = torch.from_numpy(z["x"]).to("cuda")
    ei, ew = gcn_norm(torch.from_numpy(z["edge_index"]).to("cuda"), None, x.size(0), add_self_loops=True)
    assert np.array_equal(ei.cpu().numpy(), z["ei_norm"])
    assert np.array_equal(ew.cpu().numpy(), z["ew_norm"])                          # same order, same rounding
    S, mc, o, _ = m(x.float(), ei, ew)
    (mc + o).backward()
    ids = torch.empty(x.size(0), dtype=torch.int64, device="cuda")
    _hip.call("hscn_assign_argmax", _hip.ptr(S), _hip.ptr(ids), x.size(0), K, _hip.stream())
    assert np.array_equal(ids.cpu().numpy(), z["clusters"])                        # bit-exact cluster indices
    np.testing.assert_allclose(S.detach().cpu().numpy(), z["S"], atol=1e-5)
    assert abs(mc.item() - float(z["mc"])) < 1e-5 and abs(o.item() - float(z["o"])) < 1e-5
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), z[f"g::{k}"], atol=1e-4, rtol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["resident", "layered"])
def test_hip_hscn_matches_golden(engine):
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loss import criterion
    from graph_hscn.model.hscn import HSCN
    z = _load("hscn_peptides_b4.npz")
    graphs = _graphs(z)
    K = int(z["K"])
    ids = np.split(z["clusters"], np.cumsum(z["num_nodes"])[:-1])
    for g, yrow in zip(graphs, z["y"]):
        g.y = torch.from_numpy(yrow).view(1, -1)
    hb = HeteroBatch.from_data_list([hetero_from_clusters(g, i, K) for g, i in zip(graphs, ids)])
    assert np.array_equal(hb["virtual"].x.numpy(), z["virtual_x"])
    assert np.array_equal(hb.edge_index_dict[("local", "to", "virtual")].numpy(), z["ei_lv"])
    hb = hb.to("cuda")
    m = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to("cuda")
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("w::")})
    m.engine, m.keep_virtual = engine, True
    pred = m(hb.x_dict, hb.edge_index_dict, hb)
    assert m.last_engine == engine
    loss, _ = criterion("cross_entropy", pred, hb["local"].y)
    loss.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), z["pred"], atol=1e-5, rtol=1e-5)
    assert abs(loss.item() - float(z["loss"])) < 1e-6
    if engine == "resident":
        fv = z["final_virtual"]        # 1e-5 relative to the tensor's magnitude (tests/helpers.py: scale_close)
        assert float(np.abs(m.last_virtual.cpu().numpy() - fv).max()) <= 1e-5 * max(1.0, float(np.abs(fv).max()))
    for k, p in m.named_parameters():
        if f"g::{k}" in z:
            np.testing.assert_allclose(p.grad.cpu().numpy(), z[f"g::{k}"], atol=1e-5, rtol=1e-3)
        else:
            assert p.grad is None
