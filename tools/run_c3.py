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
# argv[3] = "sweep": the column-by-column elimination order of SURVEY.md 8d (a chain of width-(H+1) cliques) instead of min-fill
ORDER = [names[i, j] for j in range(W) for i in range(H)] if len(sys.argv) > 3 and sys.argv[3] == "sweep" else None
t0 = time.perf_counter(); tree = jt.create_junction_tree(factors, sizes, order=ORDER); t1 = time.perf_counter()
widths = [len(c) for c in tree.clique_tree.maxcliques]
print("variables %d factors %d cliques %d max width %d  construction %.2f s" % (len(names), len(factors), len(widths), max(widths), t1 - t0))
plan = tree.plan("f32"); t2 = time.perf_counter()
d = plan.describe()
print("plan %.2f s: arena %.2f GiB, %d launches, %d blocks, max LDS %d" % (t2 - t1, d["arena_elems"] * 4 / 2**30, len(d["launches"]), d["n_blocks"], d["max_lds"]))
t3 = time.perf_counter(); out = tree.propagate(values); t4 = time.perf_counter()
print("propagate() first call (H2D factors, device evaluate, collect+distribute, %d device marginals, D2H) %.3f s" % (len(factors), t4 - t3))
ct = tree.clique_tree
labels, f2c = ct.factor_graph.factors, ct.factor_to_maxclique
# steady state, every factor table changed (the worst case for evaluate): end to end, then stage by stage
reps = 5
e2e, ev, pr, mg = [], [], [], []
for r in range(reps):
    vals = [v * np.float32(1.0 + 1e-3 * (r + 1)) for v in values]
    plan.sync()
    t0 = time.perf_counter(); out = tree.propagate(vals); t1 = time.perf_counter()
    e2e.append(t1 - t0)
    vals = [v * np.float32(1.0 - 1e-3 * (r + 1)) for v in values]
    ta = time.perf_counter(); n_staged = plan.stage_factors(labels, f2c, vals); plan.sync(); tb = time.perf_counter()
    plan.propagate(); tc = time.perf_counter()
    out = plan.factor_marginals(labels, f2c); td = time.perf_counter()
    ev.append(tb - ta); pr.append(tc - tb); mg.append(td - tc)
d2h = sum(o.nbytes for o in out)
# what the device stores of the potentials since round 5: the tables of the cliques covered (nearly) whole and the static tables of
# the others at the shape their factors cover - a clique without factors stores nothing (plan.stats(): n_unit_cliques)
st5 = plan.stats()
phys = {p["real"]: (0 if p["unit"] else p["phys_elems"]) for p in d["pnodes"] if p["real"] >= 0}
with_factors = sorted(set(f2c))
staged_bytes = sum(phys[c] for c in with_factors) * 4 + st5["fixed_bytes"]
print("cliques %d, of which %d keep no table (%d of those hold factors: static tables, %.1f MB); full-shape tables would be %.2f GiB"
      % (len(widths), st5["n_unit_cliques"], st5["n_static_tables"], st5["fixed_bytes"] / 1e6, sum(8.0 ** w for w in widths) * 4 / 2**30 if K == 8 else 0))
# every clique (also the ones no factor is assigned to: all-ones tables, formed once per plan)
cold = []
for r in range(3):
    plan._factor_tables.prev = None
    plan.sync()
    ta = time.perf_counter(); n_all = plan.stage_factors(labels, f2c, values); plan.sync(); cold.append(time.perf_counter() - ta)
all_bytes = sum(phys.values()) * 4 + st5["fixed_bytes"]
print("evaluate of ALL %d cliques (first call of a plan): %.2f ms for %.2f GiB written -> %.2f TB/s" % (n_all, min(cold) * 1e3, all_bytes / 2**30, all_bytes / min(cold) / 1e12))
n_fold = sum(1 for t in plan.describe()["tasks"] if t.get("fold"))
print("(the API's plan: %d marginal tasks of cliques without a table are folded into its propagate - the \"collect+distribute\" stage below includes them; the planner folds where the distribute levels leave the chip's slots idle, JTP_FOLD=1 / 0: wherever possible / nowhere)" % n_fold)
print("propagate() steady state, all %d factor tables new each call: end to end %.2f ms (min of %d; median %.2f)" % (len(factors), min(e2e) * 1e3, reps, sorted(e2e)[reps // 2] * 1e3))
named = []
for r in range(reps):           # the caller says what changed (round 6): nothing compared, the factor lists not looked at again
    vals = [v * np.float32(1.0 + 3e-3 * (r + 1)) for v in values]
    plan.sync()
    t0 = time.perf_counter(); out = tree.propagate(vals, changed="all"); named.append(time.perf_counter() - t0)
print("   the same with propagate(values, changed=\"all\"): %.2f ms (min of %d; median %.2f)" % (min(named) * 1e3, reps, sorted(named)[reps // 2] * 1e3))
print("   stage by stage (each synchronised, min of %d): evaluate of the %d cliques that have factors %.2f ms (%.2f GiB written -> %.2f TB/s), collect+distribute %.2f ms, %d factor marginals %.2f ms (%.2f GiB of potentials / belief tables behind them; D2H %.2f MB)"
      % (reps, n_staged, min(ev) * 1e3, staged_bytes / 2**30, staged_bytes / min(ev) / 1e12, min(pr) * 1e3, len(factors), min(mg) * 1e3, staged_bytes / 2**30, d2h / 1e6))
out = tree.propagate(vals)
t0 = time.perf_counter(); out = tree.propagate(vals); t1 = time.perf_counter()
print("propagate() with unchanged factor tables (nothing staged: %d cliques): %.2f ms" % (plan.staged_cliques, (t1 - t0) * 1e3))
out = tree.propagate(values)
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
