"""Soak test of the dataflow launches: many propagates, every result compared bit for bit with the first
(the engine is deterministic), the fallback counter must stay 0.   python tools/soak.py [propagates]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import engine, synthetic

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cases = [("c4", synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0), "f32", reps),
         ("mid", synthetic.wide_binary_tree(n_cliques=63, width=16, sep=8, card=2, seed=3), "f64", reps),
         ("random", synthetic.random_tree(n_cliques=40, width=14, sep=6, card=2, seed=5), "f32", reps),
         ("chain", synthetic.chain_tree(n_cliques=60, card=16, width=3), "f64", reps),
         ("odd", synthetic.wide_binary_tree(n_cliques=15, width=10, sep=5, card=3, seed=2), "f32", reps),     # mixed-radix rows, compact form (kernels *_mix<T, true>: LDS adds in wave order)
         ("odd5", synthetic.wide_binary_tree(n_cliques=15, width=7, sep=3, card=5, seed=4), "f64", reps),
         ("odd6", synthetic.wide_binary_tree(n_cliques=15, width=7, sep=3, card=6, seed=6), "f32", reps)]     # mixed-radix rows, one row per step (*_mix<T, false>)
for name, spec, dt, n in cases:
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dt)
    plan.fill_synthetic(1, spec["scales"])
    plan.propagate()
    probe = [c for c in (0, spec["n_cliques"] // 2, spec["n_cliques"] - 1)]
    ref = [plan.marginal(c, list(spec["node_vars"][c])[:2]) for c in probe]
    z0 = plan.z()
    t0 = time.perf_counter()
    bad = 0
    for i in range(n):
        plan.propagate(sync=False)
        if i % 50 == 49:
            plan.sync()
            got = [plan.marginal(c, list(spec["node_vars"][c])[:2]) for c in probe]
            if plan.z() != z0 or any(not np.array_equal(a, b) for a, b in zip(got, ref)):
                bad += 1
    plan.sync()
    st = plan.stats()
    print("%-7s %5d propagates  %.1f s  mismatching checks %d  fallbacks %d  Z %.12g" % (name, n, time.perf_counter() - t0, bad, st["flow_fallbacks"], z0))
    assert bad == 0 and st["flow_fallbacks"] == 0
    plan.close()
# several evidence sets in flight on their own streams (their dataflow kernels share the GPU)
spec = synthetic.wide_binary_tree(n_cliques=127, width=17, sep=8, card=2, seed=7)
plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", n_batch=6)
for b in range(6):
    plan.fill_synthetic(10 + b, spec["scales"], batch=b)
plan.propagate(0, 6)
z0 = [plan.z(batch=b) for b in range(6)]
t0 = time.perf_counter()
bad = 0
for i in range(reps):
    plan.propagate(0, 6, sync=False)
    if i % 100 == 99:
        plan.sync()
        bad += sum(plan.z(batch=b) != z0[b] for b in range(6))
plan.sync()
print("batch6  %5d x 6 propagates  %.1f s  mismatching checks %d  fallbacks %d" % (reps, time.perf_counter() - t0, bad, plan.stats()["flow_fallbacks"]))
assert bad == 0 and plan.stats()["flow_fallbacks"] == 0
plan.close()
# two PLANS whose propagates are in flight together, each on its own stream: the second one to start takes its workgroups in
# ticket order (jtp_propagate: another plan's dataflow propagate is in flight), the kernels share the GPU
specs = [synthetic.wide_binary_tree(n_cliques=63, width=17, sep=8, card=2, seed=11), synthetic.chain_tree(n_cliques=40, card=16, width=3),
         synthetic.wide_binary_tree(n_cliques=15, width=10, sep=5, card=3, seed=12)]
plans = [engine.Plan(sp["tree"], sp["node_vars"], sp["sizes"], dtype=dt) for sp, dt in zip(specs, ("f32", "f64", "f32"))]
for pl, sp in zip(plans, specs):
    pl.fill_synthetic(21, sp["scales"])
    pl.propagate()
z0 = [pl.z() for pl in plans]
m0 = [pl.marginal(0, list(sp["node_vars"][0])[:2]) for pl, sp in zip(plans, specs)]
t0 = time.perf_counter()
bad = 0
for i in range(reps):
    for pl in plans:
        pl.propagate(sync=False)
    if i % 100 == 99:
        for k, (pl, sp) in enumerate(zip(plans, specs)):
            pl.sync()
            bad += pl.z() != z0[k] or not np.array_equal(pl.marginal(0, list(sp["node_vars"][0])[:2]), m0[k])
for pl in plans:
    pl.sync()
st = [pl.stats() for pl in plans]
print("3plans  %5d x 3 propagates  %.1f s  mismatching checks %d  fallbacks %d  propagates in ticket order %r" % (
    reps, time.perf_counter() - t0, bad, sum(x["flow_fallbacks"] for x in st), [x["tickets_used"] for x in st]))
assert bad == 0 and all(x["flow_fallbacks"] == 0 for x in st)
for pl in plans:
    pl.close()
print("soak ok")
# (round 6) the lean unit pass: a lattice whose junction tree is mostly cliques that keep no table, through the public API
import junctiontree_amd as jt
factors, sizes, values = synthetic.lattice_mrf(6, 40, 8)
# ... on its min-fill tree (marginals by the read-out: the planner folds where the distribute levels leave slots idle), on its column-sweep
# tree (marginal tasks folded into the propagate by the planner's own choice), and on the min-fill tree with the folded tasks forced
for name, order, env in (("lean", None, None), ("folded", synthetic.lattice_column_order(6, 40), None), ("folded+", None, "1")):
    if env is not None:
        os.environ["JTP_FOLD"] = env
    tree = jt.create_junction_tree(factors, sizes, order=order)
    first = tree.propagate(values)
    plan = tree.plan("f32")
    n_lean = sum(1 for t in plan.describe()["tasks"] if t["lean_off"] > 0)
    n_fold = sum(1 for t in plan.describe()["tasks"] if t["fold"])
    t0 = time.perf_counter()
    bad = 0
    for i in range(reps):
        plan.propagate(sync=False)
        if i % 100 == 99:
            out = plan.factor_marginals(tree.clique_tree.factor_graph.factors, tree.clique_tree.factor_to_maxclique)
            bad += sum(not np.array_equal(a, b) for a, b in zip(out, first))
    plan.sync()
    print("%-8s%5d propagates  %.1f s  mismatching checks %d  fallbacks %d  (%d lean tasks, %d of them folded marginal tasks, %d cliques without a table)"
          % (name, reps, time.perf_counter() - t0, bad, plan.stats()["flow_fallbacks"], n_lean, n_fold, plan.stats()["n_unit_cliques"]))
    assert bad == 0 and plan.stats()["flow_fallbacks"] == 0 and n_lean > 0 and (n_fold > 0) == (name != "lean")
    engine.clear_plan_cache()
    os.environ.pop("JTP_FOLD", None)
# (round 6) the active lists of a multi-set plan: 64 evidence sets, an evidence-free set, the same evidence every propagate
spec = synthetic.wide_binary_tree(n_cliques=63, width=16, sep=8, card=2, seed=21)
plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", n_batch=64, multiset=True)
plan.fill_synthetic(3, spec["scales"])
labels = sorted(spec["sizes"])
for b in range(64):
    rng = np.random.default_rng(500 + b)
    plan.set_evidence({labels[i]: int(rng.integers(0, 2)) for i in rng.choice(len(labels), size=6, replace=False)}, batch=b)
plan.propagate(0, 64)
z0 = [plan.z(batch=b) for b in range(64)]
t0 = time.perf_counter()
bad = 0
for i in range(reps // 4):
    plan.propagate(0, 64, sync=False)
    if i % 50 == 49:
        plan.sync()
        bad += sum(plan.z(batch=b) != z0[b] for b in range(0, 64, 5))
plan.sync()
print("multi64 %5d x 64 propagates  %.1f s  mismatching checks %d  fallbacks %d" % (reps // 4, time.perf_counter() - t0, bad, plan.stats()["flow_fallbacks"]))
assert bad == 0 and plan.stats()["flow_fallbacks"] == 0
plan.close()
