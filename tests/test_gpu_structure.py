"""structure_build = "dataset-resident": the graphs' CSRs and degree norms built once (hscn_resident_structure),
gathered with the batch (hscn_collate_gather_structure) and LOADED by the one-launch step instead of rebuilt in LDS
every step.  Graph structure is epoch-invariant; results must not change by a bit."""
import numpy as np
import pytest
import torch

from tests.helpers import DEV

pytestmark = pytest.mark.gpu


def _hetero(name, G, K, seed, C=10):
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    graphs = make_dataset(name, G, seed=seed)
    rng = np.random.default_rng(seed)
    hs = [hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs]
    for h in hs:
        h["local"].y = torch.from_numpy((rng.random((1, C)) < 0.3).astype(np.float32))
    return hs


def _stable_csr(key, other, rows):
    order = np.argsort(key, kind="stable")
    rowptr = np.zeros(rows + 1, dtype=np.int64)
    np.add.at(rowptr, key + 1, 1)
    return np.cumsum(rowptr), other[order]


def test_structure_builder_equals_a_host_stable_sort():
    """Every graph's four CSRs (rows keep ascending edge order) and two degree norms against numpy's stable argsort
    on the same edge lists, including a relation with repeated edges (the virtual -> virtual pattern has self loops)."""
    from graph_hscn.data import HeteroBatch
    from graph_hscn.engine import build_structure
    LL, VV, LV = ("local", "to", "local"), ("virtual", "to", "virtual"), ("local", "to", "virtual")
    hs = _hetero("peptides_func", 9, 16, seed=3)
    hb = HeteroBatch.from_data_list(hs)
    st = build_structure(hb.to(DEV))
    torch.cuda.synchronize()
    t = {k: v.cpu().numpy() for k, v in st.t.items()}
    lp, vp = hb["local"].ptr.numpy(), hb["virtual"].ptr.numpy()
    for g in range(9):
        n0, n = lp[g], lp[g + 1] - lp[g]
        v0, nv = vp[g], vp[g + 1] - vp[g]
        for et, (rk, ck, dk, r0, nr, kb, ob, by_dst) in {
                "lld": ("ll_rowptr_d", "ll_col_d", "ll_dinv", n0, n, n0, n0, True),
                "lls": ("ll_rowptr_s", "ll_col_s", None, n0, n, n0, n0, False),
                "vv": ("vv_rowptr", "vv_col", "vv_dinv", v0, nv, v0, v0, True),
                "lv": ("lv_rowptr", "lv_col", None, v0, nv, v0, n0, True)}.items():
            rel = LL if et.startswith("ll") else (VV if et == "vv" else LV)
            e0, e1 = hb[rel].ptr32.numpy()[g], hb[rel].ptr32.numpy()[g + 1]
            ei = hb[rel].edge_index.numpy()[:, e0:e1]
            key = (ei[1] if by_dst else ei[0]) - kb
            oth = (ei[0] if by_dst else ei[1]) - ob
            rp, col = _stable_csr(key, oth, nr)
            got_rp = t[rk][r0 + g: r0 + g + nr + 1]
            assert np.array_equal(got_rp, rp), (g, et)
            assert np.array_equal(t[ck][e0:e1], col), (g, et)
            if dk:
                deg = np.diff(rp).astype(np.float64)
                want = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0.0).astype(np.float32)
                np.testing.assert_allclose(t[dk][r0:r0 + nr], want, rtol=1e-6)


@pytest.mark.parametrize("name,B,K,H,L,C,dtype", [("peptides_func", 24, 16, 16, 3, 10, torch.float32),
                                                  ("pcqm_contact", 40, 16, 32, 2, 1, torch.float16),
                                                  ("peptides_struct", 12, 32, 16, 1, 11, torch.float32)])
def test_step_on_resident_structure_is_bit_identical(name, B, K, H, L, C, dtype):
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import HeteroBatch
    from graph_hscn.engine import build_structure
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.step import ResidentTrainStep
    hs = _hetero(name, B, K, seed=B, C=C)
    d = HeteroBatch.from_data_list(hs).to(DEV).with_feature_dtype(dtype)
    torch.manual_seed(2)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], d["local"].x.size(1), H, C, L).to(DEV)
    a = ResidentTrainStep(model, d, "l1" if C == 1 else "cross_entropy", one_launch=True)
    a.run()
    st = build_structure(d)
    b = ResidentTrainStep(model, d, "l1" if C == 1 else "cross_entropy", one_launch=True, structure="batch")
    assert b.structure is st
    b.run()
    torch.cuda.synchronize()
    a.check()
    for f in ("pred", "score", "grads", "virtual"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f


def test_device_dataset_gathers_structure_and_replays_on_it():
    """DeviceHeteroDataset(resident_structure=True): the gathered slices equal the structure built on the
    host-collated batch of the same graphs, and an epoch of captured replays that LOAD their structure produces the
    losses and parameters of the epoch that rebuilds it every step, bit for bit (optimizer in the graph)."""
    import copy
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import HeteroBatch
    from graph_hscn.engine import build_structure
    from graph_hscn.loader.device_dataset import DeviceHeteroDataset
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.replay import CapturedStep
    G, B = 40, 8
    hs = _hetero("peptides_func", G, 8, seed=4)
    ds = DeviceHeteroDataset(hs, DEV, B, resident_structure=True)
    ids = torch.tensor([5, 31, 2, 2, 17, 39, 0, 11], device=DEV)
    hb = ds.gather(ids)
    torch.cuda.synchronize()
    ds.check()
    ref = build_structure(HeteroBatch.from_data_list([hs[i] for i in ids.tolist()]).to(DEV))
    got = hb.structure
    for k, v in ref.t.items():
        assert torch.equal(got.t[k][: v.numel()], v) or v.numel() == 1, k
    torch.manual_seed(0)
    m1 = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(DEV)
    m2 = copy.deepcopy(m1)
    mk = lambda m: torch.optim.AdamW(m.parameters(), lr=1e-2, capturable=True, fused=True)
    o1, o2 = mk(m1), mk(m2)
    gen = torch.Generator(device=DEV).manual_seed(1)
    ds.new_epoch(gen)
    s1 = CapturedStep(m1, ds.static, "cross_entropy", optimizer=o1, pre=ds.gather_next)
    s2 = CapturedStep(m2, ds.static, "cross_entropy", optimizer=o2, pre=ds.gather_next, structure="batch")
    assert s2.step.structure is ds.static.batch.structure and s1.step.structure is None
    for step in (s1, s2):
        ds.new_epoch(torch.Generator(device=DEV).manual_seed(7))
        losses = []
        for _ in range(G // B):
            losses.append(step.replay().clone())
        step.losses = torch.stack(losses)
    torch.cuda.synchronize()
    ds.check()
    assert torch.equal(s1.losses, s2.losses)
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), n
