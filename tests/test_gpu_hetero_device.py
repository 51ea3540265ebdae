"""Device-side cluster -> hetero batch transform vs the oracle's per-node Python loop: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import hetero_data as OH
from tests.helpers import DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,B,K", [("peptides_func", 9, 16), ("peptides_func", 3, 1), ("pascalvoc_sp", 4, 64),
                                      ("pcqm_contact", 17, 5)])
def test_device_transform_is_bit_identical(name, B, K):
    from graph_hscn.data import Batch, HeteroBatch
    from graph_hscn.loader.hetero_data import LL, LV, VV, hetero_batch_on_device, hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    graphs = make_dataset(name, B, seed=B)
    rng = np.random.default_rng(K)
    ids = [rng.integers(0, K, g.num_nodes) for g in graphs]
    want = OH.collate_hetero([OH.hetero_from_clusters(g.x, g.edge_index, g.y, i, K) for g, i in zip(graphs, ids)])
    big = Batch.from_data_list(graphs).to(DEV)
    hb = hetero_batch_on_device(big, torch.from_numpy(np.concatenate(ids)).to(DEV), K)
    assert torch.equal(hb["virtual"].x.cpu(), want["x_dict"]["virtual"])          # f64 means cast to f32, shifted
    assert torch.equal(hb["local"].x.cpu(), want["x_dict"]["local"])
    for et, key in ((LL, OH.LL), (VV, OH.VV), (LV, OH.LV)):
        assert torch.equal(hb[et].edge_index.cpu(), want["edge_index_dict"][key])
    assert torch.equal(hb["virtual"].batch.cpu(), want["batch_virtual"])
    host = HeteroBatch.from_data_list([hetero_from_clusters(g, i, K) for g, i in zip(graphs, ids)])
    for et in (LL, VV, LV):
        assert torch.equal(hb[et].ptr32.cpu(), host[et].ptr32)
    assert hb["virtual"].max_nodes == host["virtual"].max_nodes and torch.equal(hb["virtual"].ptr32.cpu(), host["virtual"].ptr32)


def test_device_batch_feeds_the_resident_engine_like_the_host_batch():
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import Batch, HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_batch_on_device, hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import HSCN
    graphs = make_dataset("peptides_func", 8, seed=2)
    rng = np.random.default_rng(0)
    ids = [rng.integers(0, 16, g.num_nodes) for g in graphs]
    torch.manual_seed(0)
    m = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(DEV)
    m.engine = "resident"
    host = HeteroBatch.from_data_list([hetero_from_clusters(g, i, 16) for g, i in zip(graphs, ids)]).to(DEV)
    dev = hetero_batch_on_device(Batch.from_data_list(graphs).to(DEV), torch.from_numpy(np.concatenate(ids)).to(DEV), 16)
    a = m(host.x_dict, host.edge_index_dict, host)
    b = m(dev.x_dict, dev.edge_index_dict, dev)
    assert torch.equal(a, b)


def test_device_transform_rejects_bad_cluster_ids():
    from graph_hscn.data import Batch
    from graph_hscn.loader.hetero_data import hetero_batch_on_device
    from graph_hscn.loader.synthetic import make_dataset
    big = Batch.from_data_list(make_dataset("pcqm_contact", 2, seed=0)).to(DEV)
    ids = torch.zeros(big.num_nodes, dtype=torch.int64, device=DEV)
    ids[3] = 99
    with pytest.raises(IndexError):
        hetero_batch_on_device(big, ids, 16)
