"""Diagnostic: per-launch device times of BASELINE config 3 (6 x W lattice), aggregated by the size of the level."""
import os, sys, collections
import numpy as np
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
tree = jt.create_junction_tree(factors, sizes)
node_vars = [list(c) for c in tree.clique_tree.maxcliques] + [list(s) for s in tree.separators]
plan = engine.Plan(tree.tree, node_vars, sizes, dtype="f32", cover=None if os.environ.get("C3_NO_COVER") else tree.cover())
plan.fill_synthetic(1, [8.0 ** -(len(c) - 1) for c in tree.clique_tree.maxcliques])
for _ in range(2):
    plan.propagate()
plan.set_profiling(3, per_launch=True)
for _ in range(3):
    plan.propagate()
L = plan.launch_ms()
d = plan.describe()
tot = sum(x["ms"] for x in L)
print("launches %d, sum of launch times %.2f ms" % (len(L), tot))
widths = collections.Counter(len(c) for c in tree.clique_tree.maxcliques)
print("clique widths:", dict(widths))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for x in L:
    key = (x["phase"], "reduce" if x["variant"] == 16 else ("<=512 blocks" if x["nblocks"] <= 512 else "<=1024 blocks" if x["nblocks"] <= 1024 else "<=2048 blocks" if x["nblocks"] <= 2048 else ">2048 blocks"))
    a = agg[key]
    a[0] += 1; a[1] += x["ms"]; a[2] += x["alg_bytes"]
for key in sorted(agg):
    n, ms, b = agg[key]
    print("phase %d %-14s: %4d launches %8.3f ms (%4.1f %%)  %8.1f MB  %7.0f GB/s  avg %.1f us" % (key[0], key[1], n, ms, 100 * ms / tot, b / 1e6, b / max(ms, 1e-9) / 1e6, ms / n * 1e3))
big = sorted(L, key=lambda x: -x["ms"])[:8]
for x in big:
    print("   slowest: phase %d level %d  %5d blocks %3d tasks %.1f MB  %.3f ms  %.0f GB/s" % (x["phase"], x["level"], x["nblocks"], x["ntasks"], x["alg_bytes"] / 1e6, x["ms"], x["alg_bytes"] / x["ms"] / 1e6))
