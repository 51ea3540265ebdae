"""CPU oracle for the Graph-HSCN hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-torch / numpy *restatement* of what the reference
(`/root/reference/graph_hscn`, pure Python on top of un-vendored
torch_geometric 2.2/2.3 + torch_scatter) computes on its CPU path for

  * MinCUT spectral-clustering coarsening  (graph_hscn/model/hscn.py:19-64,
    graph_hscn/train/train_clustering.py:36-69)
  * heterogeneous local/virtual message passing (graph_hscn/model/hscn.py:67-140)
  * the cluster-ids -> HeteroData transform   (graph_hscn/loader/hetero_data.py:14-88)

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
this path, and its third-party arithmetic (torch_geometric, unpinned in
requirements-cpu.txt:10) is not installed in the build container, so the
reference cannot be run here (ordinary ModuleNotFoundError, nothing was
denied).  The oracle is therefore pinned only by (i) closed-form known-answer
tests (tests/test_oracle_kat.py) and (ii) seeded
fixtures it generated itself (tests/golden/, regression pins).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this package.  The product (`graph-hscn_amd/`) never does.
"""
