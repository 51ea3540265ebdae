"""Host-side logic: containers / collate, the cluster->hetero transform, synthetic
generators, sharding.  CPU only."""
import numpy as np
import pytest
import torch

from graph_hscn.data import Batch, DataLoader, HeteroBatch
from graph_hscn.distributed import shard_bounds, shard_list
from graph_hscn.loader.hetero_data import LL, LV, VV, generate_hetero_data, hetero_from_clusters, hetero_loaders
from graph_hscn.loader.synthetic import SHAPES, make_dataset
from oracle import hetero_data as OH


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_synthetic_graphs_have_the_lrgb_shape(name):
    gs = make_dataset(name, 40, seed=1)
    gs2 = make_dataset(name, 40, seed=1)
    sh = SHAPES[name]
    for a, b in zip(gs, gs2):
        assert torch.equal(a.x, b.x) and torch.equal(a.edge_index, b.edge_index)      # seeded
    n = np.array([g.num_nodes for g in gs])
    e = np.array([g.num_edges for g in gs])
    assert n.min() >= sh.n_min and n.max() <= sh.n_max
    assert abs(e.sum() / n.sum() - 2 * sh.und_per_node) < 0.25
    for g in gs:
        ei = g.edge_index
        assert (ei[0] != ei[1]).all()                                                  # no self loops
        assert torch.equal(ei[:, 0::2], ei[:, 1::2].flip(0))                           # (i,j),(j,i) pairs
        assert len({tuple(c) for c in ei.T.tolist()}) == ei.size(1)                    # no duplicates
        assert g.x.shape == (g.num_nodes, sh.num_features) and g.y.shape == (1, sh.num_classes)


@pytest.mark.parametrize("K", [1, 4, 16, 64])
def test_hetero_transform_is_bit_identical_to_the_oracle(K):
    rng = np.random.default_rng(K)
    for name in ("peptides_func", "pascalvoc_sp"):
        for g in make_dataset(name, 6, seed=K):
            raw = rng.integers(0, K, g.num_nodes)
            raw[rng.integers(0, g.num_nodes)] = K - 1
            want = OH.hetero_from_clusters(g.x, g.edge_index, g.y, raw, K)
            got = hetero_from_clusters(g, raw, K)
            assert torch.equal(got["virtual"].x, want["virtual_x"]) and got["virtual"].x.dtype == torch.float32
            assert torch.equal(got["local"].x, want["local_x"])
            for et, key in ((LL, OH.LL), (VV, OH.VV), (LV, OH.LV)):
                assert torch.equal(got[et].edge_index, want[key])
            assert got["virtual"].num_nodes == want["num_virtual"]


def test_generate_hetero_data_orders_train_val_test_and_loaders_reindex():
    from graph_hscn.config.config import DataConfig, HSCNConfig
    graphs = make_dataset("peptides_func", 10, seed=3)
    rng = np.random.default_rng(0)
    clusters = [rng.integers(0, 4, g.num_nodes) for g in graphs]
    split = {"train": torch.tensor([4, 0, 7, 2]), "val": torch.tensor([1, 9, 5]), "test": torch.tensor([3, 6, 8])}
    dc, mc = DataConfig("peptides_func", batch_size=2), HSCNConfig("relu", num_clusters=4)
    hs = generate_hetero_data(clusters, graphs, split, dc, mc)
    order = [4, 0, 7, 2, 1, 9, 5, 3, 6, 8]
    assert len(hs) == 10
    for h, i in zip(hs, order):
        assert h["local"].x.shape[0] == graphs[i].num_nodes
    loaders = hetero_loaders(dc, hs, split)
    assert [len(l) for l in loaders] == [2, 2, 2]
    first_val = next(iter(loaders[1]))
    # quirk B.1-8: hetero_loaders indexes the split-ordered list by ORIGINAL ids
    assert first_val["local"].x.shape[0] == hs[1]["local"].x.shape[0] + hs[9]["local"].x.shape[0]


def test_hetero_collate_matches_oracle_collate_and_carries_segmentation():
    graphs = make_dataset("peptides_func", 5, seed=2)
    rng = np.random.default_rng(1)
    ids = [rng.integers(0, 8, g.num_nodes) for g in graphs]
    ob = OH.collate_hetero([OH.hetero_from_clusters(g.x, g.edge_index, g.y, i, 8) for g, i in zip(graphs, ids)])
    pb = HeteroBatch.from_data_list([hetero_from_clusters(g, i, 8) for g, i in zip(graphs, ids)])
    assert list(pb.edge_index_dict) == [LL, VV, LV]                       # insertion order of hetero_data.py:67,77,84
    for k in ("local", "virtual"):
        assert torch.equal(pb.x_dict[k], ob["x_dict"][k])
    for et in (LL, VV, LV):
        assert torch.equal(pb.edge_index_dict[et], ob["edge_index_dict"][et])
    assert torch.equal(pb["local"].batch, ob["batch_local"]) and torch.equal(pb["local"].ptr, ob["ptr_local"])
    assert torch.equal(pb["local"].y, ob["y"]) and pb.num_graphs == 5
    # resident-engine metadata: int32 node / edge ranges + maxima
    assert pb["local"].ptr32.dtype == torch.int32 and pb["local"].max_nodes == max(g.num_nodes for g in graphs)
    e = pb[LL].ptr32
    for g_i, g in enumerate(graphs):
        sl = pb[LL].edge_index[:, e[g_i]:e[g_i + 1]] - int(pb["local"].ptr[g_i])
        assert torch.equal(sl, g.edge_index)


def test_batch_and_loader_for_homogeneous_graphs():
    graphs = make_dataset("pcqm_contact", 7, seed=0)
    b = Batch.from_data_list(graphs)
    assert b.num_graphs == 7 and b.x.shape[0] == sum(g.num_nodes for g in graphs)
    assert int(b.edge_index.max()) < b.num_nodes and b.ptr.tolist()[-1] == b.num_nodes
    dl = DataLoader(graphs, batch_size=3, shuffle=False)
    assert len(dl) == 3 and [bb.num_graphs for bb in dl] == [3, 3, 1]
    g = torch.Generator().manual_seed(0)
    a = [bb.x.shape[0] for bb in DataLoader(graphs, 3, shuffle=True, generator=g)]
    assert sum(a) == b.x.shape[0]


def test_shard_bounds_are_contiguous_balanced_and_non_empty():
    sizes = [g.num_nodes for g in make_dataset("peptides_func", 128, seed=0)]
    for ws in (1, 2, 4, 8):
        b = shard_bounds(sizes, ws)
        assert b[0] == 0 and b[-1] == len(sizes) and all(x < y for x, y in zip(b, b[1:]))
        loads = [sum(sizes[b[r]:b[r + 1]]) for r in range(ws)]
        assert max(loads) <= 1.25 * sum(sizes) / ws + max(sizes)
        assert sum(len(shard_list(sizes, sizes, r, ws)) for r in range(ws)) == len(sizes)
    assert shard_bounds([5, 5], 4) == [0, 1, 2, 2, 2] or shard_bounds([5, 5], 4)[-1] == 2


def test_product_ops_refuse_cpu_tensors_instead_of_falling_back():
    from graph_hscn.nn import GCNConv
    conv = GCNConv(4, 4, add_self_loops=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        conv(torch.randn(3, 4), torch.tensor([[0, 1], [1, 2]]))


def test_state_dict_keys_equal_the_reference_naming():
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.model.hscn import HSCN, SCN
    from oracle import models as OM
    assert sorted(SCN([16], "elu", 9, 16).state_dict()) == sorted(OM.SCN([16], "elu", 9, 16).state_dict())
    a = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3)
    b = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 3)
    assert sorted(a.state_dict()) == sorted(b.state_dict())
    assert {k: tuple(v.shape) for k, v in a.state_dict().items()} == {k: tuple(v.shape) for k, v in b.state_dict().items()}
    assert sum(p.numel() for p in a.parameters()) == 3306


def test_epoch_metrics_equal_sklearn():
    """graph_hscn.metrics (torch ops, device-resident in training) == the reference's sklearn calls
    (metrics.py:6-36), including tied scores, NaN labels and single-class columns."""
    import numpy as np
    import pytest
    import torch
    from sklearn.metrics import average_precision_score, mean_absolute_error
    from graph_hscn.metrics import eval_ap, eval_mae
    rng = np.random.default_rng(0)
    n, C = 257, 7
    y = (rng.random((n, C)) < 0.3).astype(np.float64)
    s = np.round(rng.random((n, C)), 2)          # many ties
    y[:, 2] = 1.0                                # a column without negatives is skipped
    y[rng.integers(0, n, 20), 4] = np.nan        # unlabeled entries are ignored
    ref = []
    for i in range(C):
        if np.sum(y[:, i] == 1) > 0 and np.sum(y[:, i] == 0) > 0:
            m = y[:, i] == y[:, i]
            ref.append(average_precision_score(y[m, i], s[m, i]))
    got = eval_ap(torch.from_numpy(y).float(), torch.from_numpy(s).float())
    ref32 = []
    y32, s32 = y.astype(np.float32), s.astype(np.float32)
    for i in range(C):
        if np.sum(y32[:, i] == 1) > 0 and np.sum(y32[:, i] == 0) > 0:
            m = y32[:, i] == y32[:, i]
            ref32.append(average_precision_score(y32[m, i], s32[m, i]))
    assert abs(got - sum(ref32) / len(ref32)) < 1e-12
    assert abs(got - sum(ref) / len(ref)) < 1e-6
    with pytest.raises(RuntimeError):
        eval_ap(torch.ones(5, 2), torch.rand(5, 2))
    t, p = torch.randn(40, 11), torch.randn(40, 11)
    assert abs(eval_mae(t, p) - mean_absolute_error(t.numpy(), p.numpy())) < 1e-6
    p[3, 3] = float("nan")
    with pytest.raises(Exception):
        eval_mae(t, p)


def test_mpnn_config_and_factory():
    """config/config.py:49-73 validation; model/mpnn.py:65-78 factory; parameter names of PyG's GCNConv."""
    from graph_hscn.config.config import MPNNConfig
    from graph_hscn.model.mpnn import MPNN, build_mpnn
    from oracle import models as OM
    cfg = MPNNConfig("GCN", "relu")
    assert (cfg.hidden_channels, cfg.num_layers, cfg.dropout) == (16, 3, 0.2)
    with pytest.raises(ValueError):
        MPNNConfig("gcn", "relu", dropout=1.5)
    with pytest.raises(ValueError):
        MPNNConfig("gcn", "relu", num_layers=-1)
    m = build_mpnn(cfg, 9, 10)
    assert isinstance(m, MPNN) and len(m.conv_layers) == 3 and m.dropout == 0.2
    assert sorted(m.state_dict()) == sorted(OM.MPNN(OM.ACT["relu"], 9, 16, 10, 3).state_dict())
    assert [tuple(c.lin.weight.shape) for c in m.conv_layers] == [(16, 9), (16, 16), (10, 16)]
    # normalisation layers as the reference builds them (mpnn.py:34-44): BOTH lists under use_layer_norm, none under
    # use_batch_norm alone; names and buffers are torch's
    mn = build_mpnn(MPNNConfig("gcn", "relu", use_layer_norm=True), 9, 10)
    ref = OM.MPNN(OM.ACT["relu"], 9, 16, 10, 3, use_layer_norm=True)
    assert sorted(mn.state_dict()) == sorted(ref.state_dict())
    assert len(mn.bns) == len(mn.lns) == 2 and isinstance(mn.bns[0], torch.nn.BatchNorm1d)
    assert not hasattr(build_mpnn(MPNNConfig("gcn", "relu", use_batch_norm=True), 9, 10), "bns")
    with pytest.raises(KeyError):
        build_mpnn(MPNNConfig("gin", "relu"), 9, 10)          # GINConv(dim, dim) is a TypeError in PyG


def test_static_batch_buffers_pack_and_load_on_the_host():
    """replay.StaticHeteroBatch is plain tensor bookkeeping: capacities are the maxima over the batches, every
    field is a view of one flat buffer, pack() + load() reproduce a batch inside the valid ranges, and a batch
    beyond a capacity is refused."""
    from graph_hscn.replay import LL, LV, VV, StaticHeteroBatch
    graphs = make_dataset("peptides_func", 12, seed=4)
    rng = np.random.default_rng(0)
    hs = [hetero_from_clusters(g, rng.integers(0, 8, g.num_nodes), 8) for g in graphs]
    batches = [HeteroBatch.from_data_list(hs[i:i + 4]) for i in (0, 4, 8)]
    st = StaticHeteroBatch(batches, "cpu")
    assert st.N == max(b["local"].num_nodes for b in batches) and st.num_graphs == 4
    assert st.flat.dtype == torch.uint8 and st.flat.numel() == st.nbytes
    for b in batches:
        flat = st.pack(b)
        hb = st.load(flat)
        n, v = b["local"].num_nodes, b["virtual"].num_nodes
        assert torch.equal(hb["local"].x[:n], b["local"].x.float()) and torch.equal(hb["virtual"].x[:v], b["virtual"].x)
        assert torch.equal(hb["local"].ptr32, b["local"].ptr32) and torch.equal(hb["local"].y, b["local"].y.float())
        for et in (LL, VV, LV):
            e = b[et].edge_index.size(1)
            assert torch.equal(hb[et].edge_index[:, :e], b[et].edge_index) and torch.equal(hb[et].ptr32, b[et].ptr32)
        assert hb["local"].x.data_ptr() == st.batch["local"].x.data_ptr()        # same buffers every time
    big = HeteroBatch.from_data_list(hs[:4] + hs[4:8])
    with pytest.raises(ValueError):
        st.load(big)                                                             # 8 graphs into 4-graph buffers
    with pytest.raises(ValueError):
        st.load(torch.zeros(3, dtype=torch.uint8))


def test_collate_keeps_target_rank_like_pyg():
    """PyG's collate concatenates ``y`` along dim 0 as is: graph-level class indices stay 1-D (the reference's
    multiclass branch, loss.py:11, needs ``true.ndim == 1``), node labels stay per node, [1, C] rows stack."""
    from graph_hscn.data import Batch, Data, DataLoader
    from graph_hscn.loss import criterion
    ei = torch.tensor([[0, 1], [1, 0]])
    gl = [Data(x=torch.zeros(2, 3), edge_index=ei, y=torch.tensor([c])) for c in (0, 2, 1, 1, 0)]
    b = Batch.from_data_list(gl)
    assert b.y.shape == (5,) and b.y.tolist() == [0, 2, 1, 1, 0]
    nl = [Data(x=torch.zeros(n, 3), edge_index=ei, y=torch.arange(n)) for n in (2, 3, 4)]
    assert Batch.from_data_list(nl).y.shape == (9,)
    ml = [Data(x=torch.zeros(2, 3), edge_index=ei, y=torch.ones(1, 4)) for _ in range(3)]
    assert Batch.from_data_list(ml).y.shape == (3, 4)
    zl = [Data(x=torch.zeros(2, 3), edge_index=ei, y=torch.tensor(1.5)) for _ in range(3)]
    assert Batch.from_data_list(zl).y.shape == (3,)
    # through the loader into the reference's multiclass branch
    batch = next(iter(DataLoader(gl, batch_size=5)))
    pred = torch.randn(5, 3)
    loss, score = criterion("cross_entropy", pred, batch.y)
    want = torch.nn.functional.nll_loss(torch.log_softmax(pred, -1), batch.y)
    assert torch.allclose(loss, want) and score.shape == (5, 3)
