"""INTEGRATION.md shows the ctypes stub a maintainer of the reference would add (`junctiontree/_hip.py`).  This test
EXECUTES that block as written - only the library path is substituted - on the reference's own networks: a
JunctionTree-shaped object (same attributes as the reference's: `.tree`, `.separators`, `.clique_tree.{maxcliques,
factor_to_maxclique, factor_graph.{factors, sizes}, evaluate}`) goes in, factor marginals come out, and they must
equal the outputs captured from the unmodified reference (tests/golden/networks.npz)."""
import os
import re

import numpy as np
import pytest

import junctiontree_amd as jt
from junctiontree_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_the_stub_in_integration_md_runs_and_matches_the_reference(golden):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "def propagate(jt, xs)" in b)
    assert 'C.CDLL("libjtprop.so")' in code
    code = code.replace('C.CDLL("libjtprop.so")', "C.CDLL(%r)" % _capi.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md:_hip.py", "exec"), ns)
    g = golden("networks.npz")
    for name, net in g.meta["networks"].items():
        tree = jt.create_junction_tree(net["factors"], dict(net["sizes"]))
        values = g.arrs(net["values"])
        out = ns["propagate"](tree, values)
        assert len(out) == len(values)
        for o, v, r in zip(out, values, g.arrs(net["ref_propagate"])):
            assert o.shape == np.shape(v)
            np.testing.assert_allclose(o, r, rtol=1e-11, atol=1e-300, err_msg=name)
