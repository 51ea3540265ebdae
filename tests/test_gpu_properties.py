"""Size-independent properties of the hot path at BASELINE.json's full size (Peptides-func-shaped, B = 128,
K = 16, H = 16, L = 3, C = 10 -- the bench workload), where an element-for-element oracle comparison would
take too long: a batch is a set of independent graphs (reference: PyG block-diagonal collation), so

* a graph's prediction does not depend on what else is in the batch, nor on its position in it  (bit-exact)
* the batch gradient is the mean of the per-graph gradients                                      (1e-6)
* relabelling the nodes inside a graph changes nothing but summation order                       (1e-5)
* the CSR gather-reduce is linear, and its column checksum equals the checksum of its input     (scaled shape)
"""
import numpy as np
import pytest
import torch

from tests.helpers import DEV, close

pytestmark = pytest.mark.gpu

B, K, H, L, C = 128, 16, 16, 3, 10


@pytest.fixture(scope="module")
def full():
    import bench
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.model.hscn import HSCN
    dev = torch.device("cuda:0")
    hb, graphs, ids = bench.build_hetero_batch("peptides_func", B, K, 0, dev)
    ptr = hb["local"].ptr.numpy()
    hs = [hetero_from_clusters(g, ids[ptr[i]:ptr[i + 1]], K) for i, g in enumerate(graphs)]
    torch.manual_seed(0)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, H, C, L).to(dev)
    model.engine = "resident"
    return model, hs, graphs, ids, ptr


def _batch(hs):
    from graph_hscn.data import HeteroBatch
    return HeteroBatch.from_data_list(hs).to(DEV)


def _pred(model, hs):
    d = _batch(hs)
    with torch.no_grad():
        out = model(d.x_dict, d.edge_index_dict, d)
    assert model.last_engine == "resident"
    return out


def _grads(model, hs, scale=1.0):
    from graph_hscn.loss import criterion
    d = _batch(hs)
    for p in model.parameters():
        p.grad = None
    loss, _ = criterion("cross_entropy", model(d.x_dict, d.edge_index_dict, d), d["local"].y)
    (loss * scale).backward()
    return torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).double()


def test_prediction_of_a_graph_is_independent_of_its_batch(full):
    model, hs, *_ = full
    whole = _pred(model, hs)
    assert whole.shape == (B, C) and bool(torch.isfinite(whole).all())
    sizes = [h["local"].num_nodes for h in hs]
    pick = sorted({int(np.argmax(sizes)), int(np.argmin(sizes)), 0, 1, 17, 63, 64, 100, 127})
    for g in pick:                                                     # alone
        assert torch.equal(_pred(model, [hs[g]])[0], whole[g]), g
    perm = np.random.default_rng(0).permutation(B)                     # anywhere in the batch
    assert torch.equal(_pred(model, [hs[i] for i in perm]), whole[torch.as_tensor(perm)])
    half = _pred(model, hs[40:90])                                     # with other neighbours
    assert torch.equal(half, whole[40:90])


def test_changing_one_graph_leaves_the_others_untouched(full):
    import copy
    model, hs, *_ = full
    whole = _pred(model, hs)
    other = list(hs)
    other[5] = copy.deepcopy(hs[5])
    other[5]["local"].x = hs[5]["local"].x + 1.0
    changed = _pred(model, other)
    keep = torch.ones(B, dtype=torch.bool)
    keep[5] = False
    assert torch.equal(changed[keep], whole[keep.to(whole.device)]) and not torch.equal(changed[5], whole[5])


def test_batch_gradient_is_the_mean_of_per_graph_gradients(full):
    model, hs, *_ = full
    whole = _grads(model, hs)
    acc = torch.zeros_like(whole)
    for lo in range(0, B, 16):                                         # 8 sub-batches of 16: mean of means
        acc += _grads(model, hs[lo:lo + 16]) / 8
    assert float((whole - acc).abs().max()) < 1e-6
    perm = np.random.default_rng(1).permutation(B)
    shuffled = _grads(model, [hs[i] for i in perm])                    # only the reduction order moves
    assert float((whole - shuffled).abs().max()) < 1e-6


def test_relabelling_nodes_inside_graphs(full):
    """Permute the node numbering of every graph (features, cluster ids and edge endpoints follow; edge order
    kept): predictions move only by summation order of the mean pool / neighbour sums."""
    from graph_hscn.data import Data
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    model, hs, graphs, ids, ptr = full
    rng = np.random.default_rng(2)
    rel = []
    for i, g in enumerate(graphs[:32]):
        n = g.num_nodes
        new_of_old = torch.as_tensor(rng.permutation(n))
        x = torch.empty_like(g.x)
        x[new_of_old] = g.x
        cid = np.empty(n, dtype=np.int64)
        cid[new_of_old.numpy()] = ids[ptr[i]:ptr[i + 1]]
        rel.append(hetero_from_clusters(Data(x=x, edge_index=new_of_old[g.edge_index], y=g.y, num_nodes=n), cid, K))
    a, b = _pred(model, hs[:32]), _pred(model, rel)
    assert close(b, a, atol=1e-5, rtol=1e-5)


def test_stage_a_batch_equals_graph_by_graph(full):
    """SCN on the 128 raw graphs in one launch: assignments S and cluster ids of every graph equal the
    one-graph launches bit for bit; the batch losses are the means of the per-graph losses."""
    from graph_hscn.data import Batch
    from graph_hscn.model.hscn import SCN
    _, _, graphs, *_ = full
    torch.manual_seed(3)
    scn = SCN([16], "elu", 9, K).to(DEV)

    def run(gs):
        d = Batch.from_data_list(gs).to(DEV)
        d.x = d.x.float()
        assert scn.resident_ok(d)
        with torch.no_grad():
            S, mc, o = scn.forward_graphs(d)[:3]
        return S, float(mc), float(o)

    S, mc, o = run(graphs)
    off, mcs, os_ = 0, [], []
    for g in graphs:
        s1, m1, o1 = run([g])
        assert torch.equal(s1, S[off:off + g.num_nodes])
        off += g.num_nodes
        mcs.append(m1)
        os_.append(o1)
    assert abs(mc - float(np.mean(mcs))) < 1e-6 and abs(o - float(np.mean(os_))) < 1e-6


def test_spmm_linearity_and_checksum_at_the_streaming_shape():
    """GCN gather-reduce at the bandwidth-resident shape of the bench (4096 graphs, H = 128, 685 MB):
    linear in its input, and the column sums of the output equal the weighted column sums of the input
    (sum_i out_i = sum_j (sum_i w_ij) x_j), both accumulated in float64 on the device."""
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.nn import functional as Fh
    from graph_hscn.structure import Relation
    big = Batch.from_data_list(make_dataset("peptides_func", 4096, seed=0))
    N = big.num_nodes
    ei = big.edge_index.to(DEV)
    rel = Relation(ei, N, N)
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(N, 128, device=DEV, generator=g)
    y = torch.randn(N, 128, device=DEV, generator=g)
    sx, sy = Fh.spmm_gcn_raw(rel.csr, rel.dinv, rel.dinv, x), Fh.spmm_gcn_raw(rel.csr, rel.dinv, rel.dinv, y)
    sxy = Fh.spmm_gcn_raw(rel.csr, rel.dinv, rel.dinv, 0.5 * x - 2.0 * y)
    assert close(sxy, 0.5 * sx - 2.0 * sy, atol=2e-5, rtol=1e-5)
    w = (rel.dinv[ei[0]] * rel.dinv[ei[1]]).double()                   # per-edge weight, as gcn_norm defines it
    colw = torch.zeros(N, dtype=torch.float64, device=DEV).index_add_(0, ei[0], w)
    want = (colw.unsqueeze(1) * x.double()).sum(0)
    got = sx.double().sum(0)
    # every fp32 output carries <= a few ulp of rounding; over N rows of mixed sign that adds up like a random walk
    bound = 8.0 * (N ** 0.5) * 2.0 ** -24 * float(sx.double().pow(2).mean().sqrt())
    assert float((got - want).abs().max()) < bound


def test_stage_b_on_the_device_at_full_size(full):
    """generate_hetero_data's transform for the whole 128-graph batch on the device: identical to the host
    transform graph by graph, and structurally what loader/hetero_data.py:42-87 defines -- one virtual node
    per cluster id in use, every local node tied to exactly one of its own graph's virtual nodes, U(U+1)/2
    virtual-virtual pairs."""
    from graph_hscn.data import Batch, HeteroBatch
    from graph_hscn.loader.hetero_data import LL, LV, VV, hetero_batch_on_device
    _, hs, graphs, ids, ptr = full
    host = HeteroBatch.from_data_list(hs)
    dev = hetero_batch_on_device(Batch.from_data_list(graphs).to(DEV), torch.from_numpy(ids).to(DEV), K)
    for nt in ("local", "virtual"):
        assert torch.equal(dev[nt].x.cpu(), host[nt].x) and torch.equal(dev[nt].ptr32.cpu(), host[nt].ptr32)
        assert torch.equal(dev[nt].batch.cpu(), host[nt].batch)
    for et in (LL, VV, LV):
        assert torch.equal(dev[et].edge_index.cpu(), host[et].edge_index) and torch.equal(dev[et].ptr32.cpu(), host[et].ptr32)
    vptr = dev["virtual"].ptr32.cpu().numpy()
    U = np.array([len(np.unique(ids[ptr[g]:ptr[g + 1]])) for g in range(B)])
    assert np.array_equal(np.diff(vptr), U)
    lv = dev[LV].edge_index.cpu().numpy()
    assert np.array_equal(lv[0], np.arange(ptr[-1]))
    g_of_node = np.repeat(np.arange(B), np.diff(ptr))
    assert np.all(lv[1] >= vptr[g_of_node]) and np.all(lv[1] < vptr[g_of_node + 1])
    assert np.array_equal(np.diff(dev[VV].ptr32.cpu().numpy()), U * (U + 1) // 2)
    # nodes that share a cluster id share a virtual node, and only they do
    for g in (0, 31, 127):
        sl = slice(ptr[g], ptr[g + 1])
        same_id = ids[sl][:, None] == ids[sl][None, :]
        same_v = lv[1][sl][:, None] == lv[1][sl][None, :]
        assert np.array_equal(same_id, same_v)
