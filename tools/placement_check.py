import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import engine, synthetic
spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
plan.fill_synthetic(1, spec["scales"])
for _ in range(10): plan.propagate(sync=False)
plan.sync()
t0 = time.perf_counter()
for _ in range(200): plan.propagate(sync=False)
plan.sync()
print("%.4f" % ((time.perf_counter() - t0) / 200 * 1e3))
