"""Diagnostic: what the refusal of a multi-set plan costs.  A chain of cliques of three cardinality-64 variables (4096-entry separators:
32 KiB per evidence set, beyond the 16 KiB region a multi-set pass gives a set) under S evidence sets: the multi-set plan is refused,
`propagate_evidence_sets` then runs one pass per set over shared tables - timed here against S times a single set's propagate.

    python tools/evidence_fallback_time.py [n_cliques] [sets ...]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import engine, synthetic, _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sets = [int(a) for a in sys.argv[2:]] or [8, 64]
spec = synthetic.chain_tree(n_cliques=n, card=64, width=3)
labels = sorted(spec["sizes"])


def timed(plan, S, steps=10):
    for _ in range(2):
        plan.propagate(0, S)
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.propagate(0, S, sync=False)
    plan.sync()
    return (time.perf_counter() - t0) / steps * 1e3


one = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64")
one.fill_synthetic(1, spec["scales"])
t1 = timed(one, 1)
print("chain of %d cliques, 64^3 doubles each (%.0f MB of tables), separators of 4096 doubles: one evidence set %.3f ms per propagate" % (n, n * 64 ** 3 * 8 / 1e6, t1))
one.close()
for S in sets:
    try:
        engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", n_batch=S, multiset=True).close()
        print("%d sets: the multi-set plan was made (unexpected here)" % S)
        continue
    except _capi.UnsupportedStructure as exc:
        why = str(exc)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", n_batch=S, share_potentials=True)
    plan.fill_synthetic(1, spec["scales"])
    rng = np.random.default_rng(5)
    for b in range(S):
        plan.set_evidence({labels[i]: int(rng.integers(0, 64)) for i in rng.choice(len(labels), size=4, replace=False)}, batch=b)
    t = timed(plan, S)
    print("%3d sets: multi-set plan refused (%s); one pass per set over shared tables, a stream each: %.3f ms per step = %.3f ms per set = %.2f x a single set's propagate"
          % (S, why[:80], t, t / S, t / S / t1))
    plan.close()
