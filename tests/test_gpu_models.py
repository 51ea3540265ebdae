"""SCN / HSCN through the HIP path vs the CPU oracle with identical weights:
activations within 1e-5, cluster indices bit-exact (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import hetero_data as OH
from oracle import models as OM
from oracle import pyg_ops as P
from tests.helpers import ATOL, DEV, close, scale_close, hetero_batch

pytestmark = pytest.mark.gpu


def _to_dev(d):
    return {k: v.to(DEV) for k, v in d.items()}


class _Batch(dict):
    def __init__(self, b):
        super().__init__()
        self.num_graphs = b["num_graphs"]

        class L:
            batch = b["batch_local"].to(DEV)
        self["local"] = L()


@pytest.mark.parametrize("name,B,K,H,L,C", [("peptides_func", 6, 16, 16, 3, 10), ("peptides_struct", 5, 32, 32, 2, 11),
                                            ("pcqm_contact", 9, 16, 16, 3, 1), ("pascalvoc_sp", 2, 64, 16, 2, 21)])
def test_hscn_forward_backward_matches_oracle(name, B, K, H, L, C):
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.model.hscn import HSCN
    b, _ = hetero_batch(name, B, K, seed=B)
    F = b["x_dict"]["local"].size(1)
    torch.manual_seed(B)
    om = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], F, H, C, L)
    with torch.no_grad():
        for n_, p in om.named_parameters():
            if n_.endswith("bias"):
                p.normal_(0, 0.1)
    pm = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], F, H, C, L).to(DEV)
    assert sorted(pm.state_dict()) == sorted(om.state_dict())
    pm.load_state_dict(om.state_dict())
    out_o = om(b["x_dict"], b["edge_index_dict"], b["batch_local"], B)
    out_d = pm(_to_dev(b["x_dict"]), _to_dev(b["edge_index_dict"]), _Batch(b))
    assert close(out_d, out_o, atol=ATOL, rtol=1e-5)
    g = torch.randn(B, C, generator=torch.Generator().manual_seed(1))
    out_o.backward(g)
    out_d.backward(g.to(DEV))
    for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
        if po.grad is None:
            assert pp.grad is None, n_
        else:
            assert close(pp.grad, po.grad, atol=1e-4, rtol=1e-3), n_


def test_hscn_virtual_branch_activations_match():
    """The virtual branch never reaches the prediction, so check it directly:
    one HeteroConv layer's 'virtual' output (vv GCN + lv GAT, summed)."""
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.config.config import ACT_DICT
    b, _ = hetero_batch("peptides_func", 5, 16, seed=3)
    torch.manual_seed(0)
    om = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 2)
    pm = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 2).to(DEV)
    pm.load_state_dict(om.state_dict())
    xo, xd = b["x_dict"], _to_dev(b["x_dict"])
    eo, ed = b["edge_index_dict"], _to_dev(b["edge_index_dict"])
    for lo, lp in zip(om.convs, pm.convs):
        xo = {k: v.relu() for k, v in lo(xo, eo).items()}
        xd = {k: v.relu() for k, v in lp(xd, ed).items()}
        assert scale_close(xd["virtual"], xo["virtual"]) and close(xd["local"], xo["local"])


@pytest.mark.parametrize("K,act,units", [(16, "elu", [16]), (4, "tanh", [16]), (32, "relu", [16, 16])])
def test_scn_single_graph_step_matches_oracle(K, act, units):
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn import gcn_norm
    torch.manual_seed(K)
    om = OM.SCN(units, act, 9, K)
    pm = SCN(units, act, 9, K).to(DEV)
    assert sorted(pm.state_dict()) == sorted(om.state_dict())
    pm.load_state_dict(om.state_dict())
    for g in make_dataset("peptides_func", 3, seed=K):
        om.zero_grad(); pm.zero_grad()
        S_o, mc_o, o_o, adj_o, ei_o, ew_o = OM.scn_step_single_graph(om, g.x, g.edge_index)
        (mc_o + o_o).backward()
        ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
        S_d, mc_d, o_d, adj_d = pm(g.x.to(DEV).float(), ei, ew)
        (mc_d + o_d).backward()
        assert close(S_d, S_o)
        assert abs(mc_d.item() - mc_o.item()) < ATOL and abs(o_d.item() - o_o.item()) < ATOL
        assert torch.equal(adj_d.cpu(), adj_o)
        for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
            assert close(pp.grad, po.grad, atol=1e-4, rtol=2e-3), n_


def test_cluster_assignment_bit_exact_over_many_graphs():
    """argmax ids from the HIP path == oracle ids on EVERY node of 64 graphs (train_clustering.py:68), through the
    layered operators and through the fused stage-A launch; no flip is tolerated on these committed seeds.  Prints
    the histogram of the oracle's top-2 margins (how close the nearest tie is)."""
    from graph_hscn import _hip
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn import gcn_norm
    torch.manual_seed(7)
    K = 16
    om = OM.SCN([16], "elu", 9, K)
    pm = SCN([16], "elu", 9, K).to(DEV)
    pm.load_state_dict(om.state_dict())
    margins = []
    flips = {"layered": 0, "resident": 0}
    with torch.no_grad():
        for g in make_dataset("peptides_func", 64, seed=11):
            S_o, *_ = OM.scn_step_single_graph(om, g.x, g.edge_index)
            top2 = S_o.topk(2, dim=1).values
            margins.append((top2[:, 0] - top2[:, 1]))
            want = OM.assign_clusters(S_o)
            ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
            S_l, *_ = pm(g.x.to(DEV).float(), ei, ew)
            S_r, *_ = pm.forward_graphs(g.to(DEV))
            assert pm.last_engine == "resident"
            for tag, S_d in (("layered", S_l), ("resident", S_r)):
                ids = torch.empty(g.num_nodes, dtype=torch.int64, device=DEV)
                _hip.call("hscn_assign_argmax", _hip.ptr(S_d.contiguous()), _hip.ptr(ids), g.num_nodes, K, _hip.stream())
                flips[tag] += int((want != ids.cpu().numpy()).sum())
    m = torch.cat(margins)
    edges = [0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0]
    hist = torch.histogram(m.clamp(max=1.0), bins=torch.tensor(edges)).hist.int().tolist()
    print(f"top-2 margin histogram over {m.numel()} nodes, bin edges {edges}: {hist}; min {m.min():.3e}; flips {flips}")
    assert flips == {"layered": 0, "resident": 0}, flips


def test_scn_batched_equals_mean_of_single_graph_losses():
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn import gcn_norm
    torch.manual_seed(0)
    K = 16
    pm = SCN([16], "elu", 9, K).to(DEV)
    graphs = make_dataset("peptides_func", 7, seed=4)
    mcs, oos, Ss = [], [], []
    with torch.no_grad():
        for g in graphs:
            ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
            S, mc, o, _ = pm(g.x.to(DEV).float(), ei, ew)
            mcs.append(mc); oos.append(o); Ss.append(S)
        big = Batch.from_data_list(graphs)
        ei, ew = gcn_norm(big.edge_index.to(DEV), None, big.num_nodes, add_self_loops=True)
        S, mc, o, adj = pm(big.x.to(DEV).float(), ei, ew, node_ptr=big.ptr.to(DEV).to(torch.int32))
    assert adj is None
    assert close(S, torch.cat(Ss), atol=1e-6)
    assert abs(mc.item() - torch.stack(mcs).mean().item()) < 1e-6
    assert abs(o.item() - torch.stack(oos).mean().item()) < 1e-6


def test_hetero_container_batch_through_model():
    """HeteroData -> DataLoader -> batch.to(device) -> HSCN, the protocol of
    train/train.py:73-77."""
    from graph_hscn.config.config import ACT_DICT, HSCNConfig
    from graph_hscn.data import DataLoader
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import HSCN, build_hscn
    graphs = make_dataset("peptides_func", 8, seed=9)
    rng = np.random.default_rng(0)
    hs = [hetero_from_clusters(g, rng.integers(0, 16, g.num_nodes), 16) for g in graphs]
    model = build_hscn(HSCNConfig(activation="ReLU"), 9, 10).to(DEV)
    assert isinstance(model, HSCN)
    for batch in DataLoader(hs, batch_size=4, shuffle=False):
        batch = batch.to(DEV)
        pred = model(batch.x_dict, batch.edge_index_dict, batch)
        assert pred.shape == (4, 10) and batch["local"].y.shape == (4, 10)
        pred.sum().backward()


@pytest.mark.parametrize("route,K", [("dense", 64), ("auto", 64), ("dense", 16)])
def test_scn_dense_mfma_route_matches_oracle_through_the_model(route, K):
    """BASELINE.json configs[3]: ``SCN(..., mincut_route="dense")`` runs the reference's literal sequence
    to_dense_adj -> dense_mincut_pool (model/hscn.py:61-63) with the contractions on the matrix cores.  Through the
    MODEL, against the oracle (which is that sequence in plain torch): assignments, both losses, the returned dense
    adjacency, every parameter gradient; cluster ids equal on every node whose top-2 margin exceeds 1e-5.  Then a
    batch of equally sized graphs == the mean over single-graph calls."""
    from graph_hscn.loader.synthetic import SHAPES, make_graph
    from graph_hscn.model.hscn import SCN
    from graph_hscn.nn import gcn_norm
    torch.manual_seed(K)
    rng = np.random.default_rng(K)
    graphs = [make_graph(rng, SHAPES["pascalvoc_sp"], n=nn) for nn in (479, 401, 479, 479)]
    F = graphs[0].x.size(1)
    om = OM.SCN([16], "elu", F, K)
    pm = SCN([16], "elu", F, K, mincut_route=route).to(DEV)
    pm.load_state_dict(om.state_dict())
    assert not pm.resident_ok(graphs[0])
    singles = []
    for g in graphs[:2]:
        om.zero_grad(); pm.zero_grad()
        S_o, mc_o, o_o, adj_o, _, _ = OM.scn_step_single_graph(om, g.x, g.edge_index)
        (mc_o + o_o).backward()
        ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
        S_d, mc_d, o_d, adj_d = pm(g.x.to(DEV).float(), ei, ew)
        assert pm.last_route == "dense"
        (mc_d + o_d).backward()
        assert close(S_d, S_o)
        assert abs(mc_d.item() - mc_o.item()) < ATOL and abs(o_d.item() - o_o.item()) < ATOL
        assert torch.equal(adj_d.cpu(), adj_o)
        top = S_o.topk(2, 1).values
        sure = (top[:, 0] - top[:, 1]) > 1e-5
        assert torch.equal(S_d.max(1)[1].cpu()[sure], S_o.max(1)[1][sure])
        for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
            assert close(pp.grad, po.grad, atol=1e-4, rtol=2e-3), n_
    # equally sized graphs as one [B,n,n] batch through forward_graphs: losses = mean over the graphs
    from graph_hscn.data import Batch
    same = [graphs[0], graphs[2], graphs[3]]
    with torch.no_grad():
        for g in same:
            ei, ew = gcn_norm(g.edge_index.to(DEV), None, g.num_nodes, add_self_loops=True)
            S1, mc1, o1, _ = pm(g.x.to(DEV).float(), ei, ew)
            singles.append((S1, float(mc1), float(o1)))
        Sb, mcb, ob = pm.forward_graphs(Batch.from_data_list(same).to(DEV))
    assert pm.last_route == "dense" and pm.last_engine == "layered"
    assert close(Sb, torch.cat([s[0] for s in singles]), atol=1e-6)
    assert abs(float(mcb) - np.mean([s[1] for s in singles])) < 1e-6
    assert abs(float(ob) - np.mean([s[2] for s in singles])) < 1e-6
    with pytest.raises(ValueError):
        pm.forward_graphs(Batch.from_data_list(graphs[:2]).to(DEV))        # 479 and 401 nodes: no common n
