"""BASELINE config 3 as restated in SURVEY.md 8d: 2-D lattice 6 x W (default 167 -> 1002 variables),
pairwise factors on the lattice edges, cardinality 8, float32.  Builds the junction tree with this
repo's own constructor, runs propagate() on the GPU, checks size-independent properties."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
import junctiontree_amd as jt
from junctiontree_amd import engine

H = 6
W = int(sys.argv[1]) if len(sys.argv) > 1 else 167
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
names = {(i, j): i * W + j for i in range(H) for j in range(W)}
factors = []
for i in range(H):
    for j in range(W):
        if i + 1 < H: factors.append([names[i, j], names[i + 1, j]])
        if j + 1 < W: factors.append([names[i, j], names[i, j + 1]])
sizes = {v: K for v in names.values()}
rng = np.random.default_rng(0)
scale = K ** (-len(names) / len(factors))
values = [(rng.uniform(0.5, 1.5, (K, K)) * scale).astype(np.float32) for _ in factors]
t0 = time.perf_counter(); tree = jt.create_junction_tree(factors, sizes); t1 = time.perf_counter()
widths = [len(c) for c in tree.clique_tree.maxcliques]
print("variables %d factors %d cliques %d max width %d  construction %.2f s" % (len(names), len(factors), len(widths), max(widths), t1 - t0))
plan = tree.plan("f32"); t2 = time.perf_counter()
d = plan.describe()
print("plan %.2f s: arena %.2f GiB, %d launches, %d blocks, max LDS %d" % (t2 - t1, d["arena_elems"] * 4 / 2**30, len(d["launches"]), d["n_blocks"], d["max_lds"]))
t3 = time.perf_counter(); out = tree.propagate(values); t4 = time.perf_counter()
print("propagate() end to end (H2D factors, device evaluate, collect+distribute, %d device marginals, D2H) %.2f s" % (len(factors), t4 - t3))
ct = tree.clique_tree
ta = time.perf_counter()
for c, members in enumerate(ct._members()):
    plan.set_potential_product(c, jt.take(values, members), jt.take(ct.factor_graph.factors, members))
plan.sync(); tb = time.perf_counter()
plan.propagate(); tc = time.perf_counter()
plan.marginals([(mc, list(fvars)) for fvars, mc in zip(ct.factor_graph.factors, ct.factor_to_maxclique)]); td = time.perf_counter()
print("   of which: device evaluate of %d cliques %.3f s, collect+distribute %.3f s, %d marginals %.3f s" % (len(widths), tb - ta, tc - tb, len(factors), td - tc))
plan.set_profiling(3)
for _ in range(3): plan.propagate()
st = plan.stats()
ms = st["collect_ms"] + st["distribute_ms"]
print("device collect+distribute: %.2f ms  -> %.0f GB/s algorithmic, %.0f messages/s" % (ms, st["algorithmic_bytes"] / ms / 1e6, st["n_messages"] / ms * 1e3))
z = plan.z()
# properties: every factor marginal sums to Z; the single-variable marginals implied by different factors agree
sums = np.array([o.sum() for o in out])
print("Z %.6g ; factor marginals sum to Z within %.2e" % (z, np.max(np.abs(sums - z)) / z))
marg = {}
worst = 0.0
for f, o in zip(factors, out):
    for ax, v in enumerate(f):
        m = o.sum(axis=1 - ax)
        if v in marg: worst = max(worst, float(np.max(np.abs(m - marg[v])) / np.max(marg[v])))
        else: marg[v] = m
print("single-variable marginals from different factors agree within %.2e" % worst)
assert np.isfinite(z) and worst < 2e-5
