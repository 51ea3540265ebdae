"""A short run of the randomised parity tool (tools/fuzz_resident.py: random sizes, self loops, repeated edges,
hubs, isolated nodes, random K / H / L / C / activation / loss) inside the suite; the long runs are manual."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_graphs_match_the_oracle():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_resident.py")
    spec = importlib.util.spec_from_file_location("fuzz_resident", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, refused = mod.run(24, seed=11, verbose=False)
    assert bad == 0 and refused < 24
