"""Closed-form known-answer tests pinning the CPU oracle.

The reference ships no tests or vectors for this path (parity unpinned), so
these analytic cases are what pins the oracle's semantics.
"""
import math

import numpy as np
import pytest
import torch

from oracle import pyg_ops as P
from oracle import hetero_data as OH
from oracle import models as OM


def _und(pairs):
    e = []
    for i, j in pairs:
        e += [(i, j), (j, i)]
    return torch.tensor(e, dtype=torch.long).T


def _hard(assign, K):
    # logits so that softmax is (numerically) one-hot
    s = torch.full((len(assign), K), -1e4)
    s[torch.arange(len(assign)), torch.tensor(assign)] = 1e4
    return s


def _mincut(ei, s, n, self_loops=False):
    if self_loops:
        ei, _ = P.add_remaining_self_loops(ei, None, 1.0, n)
    adj = P.to_dense_adj(ei, n)
    x = torch.zeros(n, 3)
    _, _, mc, o = P.dense_mincut_pool(x, adj, s)
    return float(mc), float(o)


def test_c1_two_triangles():
    ei = _und([(0, 1), (1, 2), (0, 2), (3, 4), (4, 5), (3, 5)])
    s = _hard([0, 0, 0, 1, 1, 1], 2)
    for sl in (False, True):
        mc, o = _mincut(ei, s, 6, sl)
        assert mc == pytest.approx(-1.0, abs=1e-6)
        assert o == pytest.approx(0.0, abs=1e-6)


@pytest.mark.parametrize("K,expect", [(4, 1.0), (16, 1.2247449), (32, 1.2831394), (64, 1.3228757)])
def test_c2_uniform_assignment(K, expect):
    ei = _und([(0, 1), (1, 2), (2, 3), (3, 4), (0, 4), (1, 3)])
    s = torch.zeros(5, K)
    mc, o = _mincut(ei, s, 5, True)
    assert mc == pytest.approx(-1.0, abs=1e-6)
    assert o == pytest.approx(expect, abs=1e-5)
    assert o == pytest.approx(math.sqrt(2 - 2 / math.sqrt(K)), abs=1e-5)


def test_c3_path_two_pairs():
    ei = _und([(0, 1), (1, 2), (2, 3)])
    s = _hard([0, 0, 1, 1], 2)
    mc, o = _mincut(ei, s, 4, False)
    assert mc == pytest.approx(-2.0 / 3.0, abs=1e-6) and o == pytest.approx(0.0, abs=1e-6)
    mc, o = _mincut(ei, s, 4, True)   # the reference's case: A + I (hscn.py:61 after gcn_norm self loops)
    assert mc == pytest.approx(-0.8, abs=1e-6) and o == pytest.approx(0.0, abs=1e-6)


def test_c4_path_unbalanced():
    ei = _und([(0, 1), (1, 2), (2, 3)])
    s = _hard([0, 1, 1, 1], 2)
    mc, o = _mincut(ei, s, 4, True)
    assert mc == pytest.approx(-0.8, abs=1e-6)
    assert o == pytest.approx(0.4595058, abs=1e-6)


def test_c5_hetero_transform():
    raw = [4, 4, 2, 5, 2, 2, 0, 5]
    x = torch.arange(8 * 3).view(8, 3)
    ei = _und([(i, i + 1) for i in range(7)])
    h = OH.hetero_from_clusters(x, ei, torch.zeros(1, 2), raw, 6)
    assert h[OH.LV][1].tolist() == [2, 2, 1, 3, 1, 1, 0, 3]
    assert h[OH.LV][0].tolist() == list(range(8))
    assert h[OH.VV][0].tolist() == [0, 0, 0, 0, 1, 1, 1, 2, 2, 3]
    assert h[OH.VV][1].tolist() == [0, 1, 2, 3, 0, 1, 2, 0, 1, 0]
    src = [[2, 4, 5], [0, 1], [3, 7], [6]]          # virtual v carries cluster (v+1) mod U
    for v, nodes in enumerate(src):
        want = x[nodes].double().mean(0).float()
        assert torch.equal(h["virtual_x"][v], want)
    assert h["virtual_x"].dtype == torch.float32 and h["num_virtual"] == 4


def test_hetero_single_cluster():
    x = torch.arange(12).view(4, 3)
    h = OH.hetero_from_clusters(x, _und([(0, 1), (1, 2), (2, 3)]), None, [7, 7, 7, 7], 16)
    assert h["num_virtual"] == 1 and h[OH.VV].tolist() == [[0], [0]] and h[OH.LV][1].tolist() == [0] * 4


def test_gcn_norm_self_loops_and_weights():
    ei = torch.tensor([[0, 1, 1, 2, 2], [1, 0, 2, 1, 2]])        # one existing self loop (2,2)
    w = torch.tensor([1.0, 1.0, 2.0, 2.0, 5.0])
    ei2, w2 = P.gcn_norm(ei, w, 3, add_self_loops=True)
    # non-loop edges first, then arange(N) loops; existing loop keeps weight 5, others 1
    assert ei2.tolist() == [[0, 1, 1, 2, 0, 1, 2], [1, 0, 2, 1, 0, 1, 2]]
    deg = torch.tensor([1 + 1.0, 1 + 2 + 1.0, 2 + 5.0])
    dis = deg.pow(-0.5)
    raw = torch.tensor([1, 1, 2, 2, 1, 1, 5.0])
    assert torch.allclose(w2, dis[ei2[0]] * raw * dis[ei2[1]])


def test_gcn_star_no_self_loops():
    # centre 0, leaves 1..4; add_self_loops=False: centre in-degree 4, leaves 1
    ei = _und([(0, i) for i in range(1, 5)])
    conv = P.GCNConv(3, 2, add_self_loops=False)
    x = torch.randn(5, 3, generator=torch.Generator().manual_seed(1))
    out = conv(x, ei)
    h = x @ conv.lin.weight.T
    assert torch.allclose(out[0], h[1:].sum(0) / 2.0 + conv.bias, atol=1e-6)
    for leaf in range(1, 5):
        assert torch.allclose(out[leaf], h[0] / 2.0 + conv.bias, atol=1e-6)


def test_gcn_isolated_node_gets_bias_only():
    ei = _und([(0, 1)])
    conv = P.GCNConv(3, 2, add_self_loops=False)
    with torch.no_grad():
        conv.bias.copy_(torch.tensor([0.5, -0.25]))
    out = conv(torch.randn(3, 3), ei)
    assert torch.equal(out[2], conv.bias.detach())


def test_gat_single_in_edge_alpha_is_one():
    conv = P.GATConv((3, 3), 4)
    xs, xd = torch.randn(5, 3), torch.randn(2, 3)
    ei = torch.tensor([[3], [1]])
    out = conv((xs, xd), ei)
    hs = xs @ conv.lin_src.weight.T
    assert torch.allclose(out[1], hs[3] + conv.bias, atol=1e-6)
    assert torch.allclose(out[0], conv.bias, atol=1e-7)       # no in-edges


def test_gat_softmax_weights_sum_to_one_and_match_formula():
    g = torch.Generator().manual_seed(3)
    conv = P.GATConv((3, 3), 4)
    xs, xd = torch.randn(6, 3, generator=g), torch.randn(2, 3, generator=g)
    ei = torch.tensor([[0, 1, 2, 3, 4, 5], [0, 0, 0, 1, 1, 1]])
    out = conv((xs, xd), ei)
    hs = xs @ conv.lin_src.weight.T
    hd = xd @ conv.lin_dst.weight.T
    a_s = (hs * conv.att_src.view(-1)).sum(-1)
    a_d = (hd * conv.att_dst.view(-1)).sum(-1)
    for v in range(2):
        js = [3 * v, 3 * v + 1, 3 * v + 2]
        e = torch.nn.functional.leaky_relu(a_s[js] + a_d[v], 0.2)
        al = torch.softmax(e, 0)
        assert torch.allclose(out[v], (al[:, None] * hs[js]).sum(0) + conv.bias, atol=1e-6)


def test_global_mean_pool_ragged():
    x = torch.arange(12.0).view(6, 2)
    b = torch.tensor([0, 0, 0, 1, 2, 2])
    out = P.global_mean_pool(x, b)
    assert torch.allclose(out, torch.stack([x[:3].mean(0), x[3], x[4:].mean(0)]))


def test_to_dense_adj_counts_duplicates():
    ei = torch.tensor([[0, 0, 1], [1, 1, 0]])
    adj = P.to_dense_adj(ei)
    assert adj.shape == (1, 2, 2) and adj[0].tolist() == [[0, 2], [1, 0]]


def test_sparse_identities_match_dense_formulas():
    """tr(S^T A S) = sum_e s_row . s_col and tr(S^T D S) = sum_i d_i |s_i|^2
    (the identities the HIP sparse route relies on)."""
    g = torch.Generator().manual_seed(0)
    n, K = 9, 4
    ei = _und([(i, (i + 1) % n) for i in range(n)] + [(0, 4), (2, 7)])
    ei, _ = P.add_remaining_self_loops(ei, None, 1.0, n)
    S = torch.softmax(torch.randn(n, K, generator=g), -1)
    A = P.to_dense_adj(ei, n)[0]
    num = torch.trace(S.T @ A @ S)
    den = torch.trace(S.T @ torch.diag(A.sum(-1)) @ S)
    num_s = (S[ei[0]] * S[ei[1]]).sum()
    deg = torch.bincount(ei[0], minlength=n).float()
    den_s = (deg * (S * S).sum(-1)).sum()
    assert torch.allclose(num, num_s, atol=1e-5) and torch.allclose(den, den_s, atol=1e-5)


def test_hscn_batch_of_two_equals_two_singles():
    from graph_hscn.loader.synthetic import make_dataset

    torch.manual_seed(0)
    graphs = make_dataset("peptides_func", 2, seed=5)
    hs = []
    rng = np.random.default_rng(0)
    for gph in graphs:
        raw = rng.integers(0, 6, size=gph.num_nodes)
        hs.append(OH.hetero_from_clusters(gph.x, gph.edge_index, gph.y, raw, 6))
    model = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 8, 10, 2)
    both = OH.collate_hetero(hs)
    out_b = model(both["x_dict"], both["edge_index_dict"], both["batch_local"], 2)
    for i, h in enumerate(hs):
        one = OH.collate_hetero([h])
        out_1 = model(one["x_dict"], one["edge_index_dict"], one["batch_local"], 1)
        assert torch.allclose(out_b[i], out_1[0], atol=1e-5)


def test_hscn_virtual_branch_never_reaches_prediction():
    """Quirk of the reference architecture (hscn.py:83-96): 'local' only receives
    the local->local relation, so the prediction ignores every virtual-branch
    parameter; their gradients are None."""
    from graph_hscn.loader.synthetic import make_dataset

    torch.manual_seed(0)
    gph = make_dataset("peptides_func", 1, seed=2)[0]
    raw = np.random.default_rng(1).integers(0, 4, size=gph.num_nodes)
    b = OH.collate_hetero([OH.hetero_from_clusters(gph.x, gph.edge_index, gph.y, raw, 4)])
    model = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 8, 10, 3)
    out = model(b["x_dict"], b["edge_index_dict"], b["batch_local"], 1)
    out.sum().backward()
    for name, p in model.named_parameters():
        if "local__to__virtual" in name or "virtual__to__virtual" in name:
            assert p.grad is None, name
        else:
            assert p.grad is not None, name


def test_scn_state_dict_keys_follow_pyg_names():
    m = OM.SCN([16], "elu", 9, 16)
    assert sorted(m.state_dict()) == sorted([
        "mp.module_0.lin_rel.weight", "mp.module_0.lin_rel.bias", "mp.module_0.lin_root.weight",
        "mlp.0.weight", "mlp.0.bias"])
    h = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 3)
    assert sum(p.numel() for p in h.parameters()) == 3306
    assert "convs.0.convs.local__to__virtual.att_src" in h.state_dict()
