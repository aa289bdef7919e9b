"""GPU fuzz (round 6) of the factor marginals formed inside the propagate (`jtp_tree_desc.fold_*`): random lattices (height, width,
cardinality, storage type, min-fill or column-sweep tree, some with unary factors and mixed cardinalities) through the public API -
whose plan carries the folded tasks - against the SAME tree's plan without them (the read-out, itself checked against the oracle and
the brute-force joint by the other batteries), and 40 propagates queued back to back that must return the first call's bits.
    python3 tools/gpu_fuzz_fold.py [count] [first seed]          (JTP_TINY_LEVEL_ELEMS=0 and JTP_FOLD=1 are set here: small lattices plan as large ones do, and fold wherever the plan's form allows)"""
import os, sys, time
os.environ.setdefault("JTP_TINY_LEVEL_ELEMS", "0")
os.environ.setdefault("JTP_FOLD", "1")                  # (wherever the plan's form allows, not only where the planner would by itself)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
import numpy as np
import junctiontree_amd as jt
from junctiontree_amd import engine, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
n_tasks = n_plans = n_copies = n_levels = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(2, 7)), int(rng.integers(4, 41))
    card = int(rng.choice([2, 2, 3, 4, 4, 5, 8]))
    if h >= 6 and card > 4:
        card = 4
    f32 = bool(seed % 2)
    factors, sizes, values = synthetic.lattice_mrf(h, w, card, seed=seed, dtype=np.float32 if f32 else np.float64)
    if seed % 3 == 0:          # mixed cardinalities: some variables one state fewer / more, the tables cut to match
        for v in rng.choice(h * w, size=max(1, h * w // 5), replace=False):
            sizes[int(v)] = max(2, card + int(rng.integers(-1, 2)))
        values = [(rng.uniform(0.5, 1.5, [sizes[a], sizes[b]]) * card ** (-len(sizes) / len(factors))).astype(values[0].dtype) for a, b in factors]
    if seed % 4 == 1:          # unary factors on some variables
        for v in rng.choice(h * w, size=max(1, h * w // 4), replace=False):
            factors.append([int(v)])
            values.append(rng.uniform(0.5, 1.5, [sizes[int(v)]]).astype(values[0].dtype))
    kw = {}
    if seed % 5 == 2:
        kw["order"] = synthetic.lattice_column_order(h, w)
    tree = jt.create_junction_tree(factors, sizes, **kw)
    if seed % 7 == 3:
        tree._opts["level_launches"] = True
        n_levels += 1
    got = tree.propagate(values)
    dt = "f32" if f32 else "f64"
    plan = tree.plan(dt)
    d = plan.describe()
    folded = [t for t in d["tasks"] if t["fold"]]
    n_tasks += len(folded)
    n_plans += bool(folded)
    n_copies += sum(1 for t in folded if any(m["npart"] > 1 for m in t["in"]))
    plain = tree.plan(dt, fold=False)
    assert not any(t["fold"] for t in plain.describe()["tasks"])
    plain.stage_factors(factors, tree.clique_tree.factor_to_maxclique, values)
    plain.propagate()
    want = plain.factor_marginals(factors, tree.clique_tree.factor_to_maxclique)
    for k, (g, x) in enumerate(zip(got, want)):
        assert g.shape == x.shape
        np.testing.assert_allclose(g, x, rtol=3e-6 if f32 else 1e-12, atol=1e-30, err_msg="seed %d factor %d %r" % (seed, k, factors[k]))
    for i in range(40):
        plan.propagate(sync=False)
        if i % 8 == 7:
            for k, (a, b) in enumerate(zip(plan.factor_marginals(factors, tree.clique_tree.factor_to_maxclique), got)):
                assert np.array_equal(a, b), "seed %d: propagate %d changed factor %d" % (seed, i, k)
    assert plan.stats()["flow_fallbacks"] == 0
    engine.clear_plan_cache()
    if (seed - first) % 20 == 19:
        print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("%d random lattices: folded marginals agree with the read-out of the same tree; plans with folded tasks %d (%d tasks, %d of them reading partial copies), %d trees launched level by level; %.0f s"
      % (n, n_plans, n_tasks, n_copies, n_levels, time.time() - t0))
