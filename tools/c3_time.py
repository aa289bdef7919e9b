"""Diagnostic: device time of collect + distribute on BASELINE config 3 (6 x W lattice, cardinality 8, float32), tables
filled on the device; nothing is checked (use with the JTP_DEBUG experiments, whose results are wrong by design)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
import junctiontree_amd as jt
from junctiontree_amd import engine
H, W, K = 6, int(sys.argv[1]) if len(sys.argv) > 1 else 167, 8
names = {(i, j): i * W + j for i in range(H) for j in range(W)}
factors = []
for i in range(H):
    for j in range(W):
        if i + 1 < H: factors.append([names[i, j], names[i + 1, j]])
        if j + 1 < W: factors.append([names[i, j], names[i, j + 1]])
sizes = {v: K for v in names.values()}
order = [names[i, j] for j in range(W) for i in range(H)] if os.environ.get("C3_SWEEP") else None      # the column-sweep tree of SURVEY.md 8d
tree = jt.create_junction_tree(factors, sizes, order=order)
node_vars = [list(c) for c in tree.clique_tree.maxcliques] + [list(s) for s in tree.separators]
# (C3_NO_COVER=1: every clique keeps a full table, as in rounds 1-4)
plan = engine.Plan(tree.tree, node_vars, sizes, dtype="f32", cover=None if os.environ.get("C3_NO_COVER") else tree.cover(),
                   lds_budget=int(os.environ.get("C3_LDS_BUDGET", "0")), block_log2=int(os.environ.get("C3_BLOCK_LOG2", "0")))
plan.fill_synthetic(1, [8.0 ** -(len(c) - 1) for c in tree.clique_tree.maxcliques])
for _ in range(2):
    plan.propagate()
plan.set_profiling(3)
for _ in range(5):
    plan.propagate()
st = plan.stats()
d = plan.describe()
print("launches %d blocks %d max_lds %d: collect %.2f ms distribute %.2f ms  total %.2f ms" % (
    st["n_launches"], d["n_blocks"], d["max_lds"], st["collect_ms"], st["distribute_ms"], st["collect_ms"] + st["distribute_ms"]))
