"""PCIe-inclusive rate of the C4 workload: host potentials in (jtp_set_potential), propagate, beliefs out
(jtp_get_belief).  `value` of bench.py never includes this; DESIGN.md quotes it beside the resident rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import engine, synthetic

spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
n = spec["n_cliques"]
plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
rng = np.random.default_rng(0)
pots = [(rng.random([2] * 20, dtype=np.float32) + 0.5) * np.float32(spec["scales"][c]) for c in range(8)]
for rep in range(2):
    t0 = time.perf_counter()
    for c in range(n):
        plan.set_potential(c, pots[c % 8])
    plan.sync(); t1 = time.perf_counter()
    plan.propagate(); t2 = time.perf_counter()
    outs = [plan.belief(c, dtype=np.float32) for c in range(n)]
    t3 = time.perf_counter()
pinned = [engine.pinned_empty([2] * 20, np.float32) for _ in range(n)]
for c in range(n):
    pinned[c][...] = pots[c % 8]
t4 = time.perf_counter()
for c in range(n):
    plan.set_potential(c, pinned[c])
plan.sync(); t5 = time.perf_counter()
plan.propagate(); t6 = time.perf_counter()
for c in range(n):
    plan.belief(c, out=pinned[c])
t7 = time.perf_counter()
gb = n * 4 * 2**20 / 1e9
print("pinned host arrays: upload %.1f ms (%.1f GB/s)   read back %.1f ms (%.1f GB/s)   end to end %.1f ms" % (
    (t5 - t4) * 1e3, n * 4 * 2**20 / 1e9 / (t5 - t4), (t7 - t6) * 1e3, n * 4 * 2**20 / 1e9 / (t7 - t6), (t7 - t4) * 1e3))
print("upload %d x 4 MiB: %.1f ms (%.1f GB/s)   propagate %.2f ms   read back: %.1f ms (%.1f GB/s)   end to end %.1f ms" % (
    n, (t1 - t0) * 1e3, gb / (t1 - t0), (t2 - t1) * 1e3, (t3 - t2) * 1e3, gb / (t3 - t2), (t3 - t0) * 1e3))
