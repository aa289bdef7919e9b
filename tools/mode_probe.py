"""Diagnostic: is the few-per-cent spread of config 4 between processes a property of the process or of the allocation?
Creates the width-20 plan several times in ONE process (destroying it in between, or keeping the old ones alive so that
the new arenas land elsewhere) and times 60 propagates of each."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import engine, synthetic
spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
keep = []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
    plan.fill_synthetic(1, spec["scales"])
    for _ in range(10): plan.propagate(sync=False)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(60): plan.propagate(sync=False)
    plan.sync()
    ms = (time.perf_counter() - t0) / 60 * 1e3
    print("plan %d: %.4f ms/propagate  (%d older plans alive)" % (rnd, ms, len(keep)))
    if len(sys.argv) > 2 and sys.argv[2] == "keep" and len(keep) < 40: keep.append(plan)
    else: plan.close()
