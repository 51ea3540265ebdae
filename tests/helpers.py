"""Shared builders for the parity tests (seeded inputs, oracle <-> product weights)."""
import numpy as np
import torch

from graph_hscn.loader.synthetic import make_dataset
from oracle import hetero_data as OH

DEV = "cuda"
ATOL = 1e-5      # north_star: float activations within 1e-5 of the CPU path
RTOL = 1e-5


def rand_graph(n, e, seed, symmetric=False, self_loops=False):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    if not self_loops:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    ei = torch.stack([src, dst])
    if symmetric:
        ei = torch.cat([ei, ei.flip(0)], 1)
    return ei


def hetero_batch(name, num_graphs, K, seed):
    graphs = make_dataset(name, num_graphs, seed=seed)
    rng = np.random.default_rng(seed)
    hs = [OH.hetero_from_clusters(g.x, g.edge_index, g.y, rng.integers(0, K, g.num_nodes), K) for g in graphs]
    return OH.collate_hetero(hs), graphs


def close(a, b, atol=ATOL, rtol=RTOL):
    a = a.detach().cpu()
    b = b.detach().cpu()
    ok = torch.allclose(a, b, atol=atol, rtol=rtol)
    if not ok:
        d = (a - b).abs()
        print("max abs diff", float(d.max()), "at", int(d.argmax()), "ref", float(b.flatten()[d.argmax()]))
    return ok


def scale_close(a, b, rel=1e-5):
    """max|a - b| <= rel * max(1, max|b|): the 1e-5 bar read relative to the magnitude of the tensor.  Used for
    outputs whose elements are sums of terms of size ~max|b| that cancel (the virtual-node features: a ReLU of
    GAT + GCN rows whose terms reach 10^1..10^2 on integer atom features) -- there an absolute 1e-5 on an element
    near zero asks for more than float32 holds (2^-23 * 67 = 8e-6 per rounding).  That HIP is no further from
    the float64 value than the float32 oracle is shown separately
    (tests/test_gpu_resident.py::test_hip_is_as_close_to_float64_as_the_float32_oracle)."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    d = float((a - b).abs().max()) if a.numel() else 0.0
    lim = rel * max(1.0, float(b.abs().max()) if b.numel() else 0.0)
    if d > lim:
        print("max abs diff", d, "limit", lim)
    return d <= lim


def grads_close(a, b, rel=2e-6):
    """Two float32 evaluations of the same gradient that group their partial sums differently (the one-launch step
    sums a weight gradient per 16-row tile, the launch pair per strided row chunk): equal to a few ulp of the
    gradient's magnitude."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    lim = rel * max(1e-6, float(b.abs().max()))
    d = float((a - b).abs().max()) if a.numel() else 0.0
    if d > lim:
        print("max abs diff", d, "limit", lim)
    return d <= lim


def pool_order_close(a, b, rel=1e-5):
    """Outputs of the one-launch step's head against the launch pair's: the same head applied to a mean pool whose
    partial sums are grouped differently -- per wave out of the last layer's accumulators (rows of a tile, tiles of a
    wave, waves) instead of a strided pass over the finished layer.  n <= 444 non-negative (post-ReLU) terms: either
    grouping is within (n - 1) 2^-24 <= 2.7e-5 of the exact sum in the worst case and ~sqrt(n) 2^-24 ~ 1e-6 typically;
    the bar applied is north_star's 1e-5, relative to the tensor's magnitude.  Returns the verdict; prints the distance
    when it fails."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    d = float((a - b).abs().max()) if a.numel() else 0.0
    lim = rel * max(1.0, float(b.abs().max()) if b.numel() else 0.0)
    if d > lim:
        print("pool-order distance", d, "limit", lim)
    return d <= lim
