"""Diagnostic (round 6): propagates queued back to back on a plan with folded marginals, the factor marginals compared bit for bit with
the first call's; every mismatch is printed with its clique.   python3 tools/fold_diag.py [reps] [every]"""
import os, sys
os.environ.setdefault("JTP_FOLD", "1")      # (the planner by itself folds only where the distribute levels leave slots idle: not on this tree)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
import numpy as np
import junctiontree_amd as jt
from junctiontree_amd import synthetic

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 20
factors, sizes, values = synthetic.lattice_mrf(6, 40, 8)
tree = jt.create_junction_tree(factors, sizes)
first = tree.propagate(values)
plan = tree.plan("f32")
d = plan.describe()
f2c = tree.clique_tree.factor_to_maxclique
folded_cliques = {t["pnode"] for t in d["tasks"] if t["fold"]}
print("mode", d.get("mode"), "fold tasks", sum(1 for t in d["tasks"] if t["fold"]), "launches", len(d["launches"]), "env",
      {k: v for k, v in os.environ.items() if k.startswith("JTP_")})
bad = 0
for i in range(reps):
    plan.propagate(sync=False)
    if i % every == every - 1:
        out = plan.factor_marginals(tree.clique_tree.factor_graph.factors, f2c)
        again = None
        for k, (a, b) in enumerate(zip(out, first)):
            if not np.array_equal(a, b):
                bad += 1
                if again is None:
                    again = plan.factor_marginals(tree.clique_tree.factor_graph.factors, f2c)
                p = d["pnodes"][f2c[k]]
                print("propagate %d factor %d %r clique %d unit %d folded %d depth %d: %d of %d entries differ, max rel %.3g; a second read-out %s; nan %d"
                      % (i, k, factors[k], f2c[k], p["unit"], f2c[k] in folded_cliques, p["depth"], int((a != b).sum()), a.size,
                         float(np.nanmax(np.abs(a - b) / np.abs(b))), "agrees with the first call" if np.array_equal(again[k], b) else
                         ("repeats the mismatch" if np.array_equal(again[k], a, equal_nan=True) else "differs from both"), int(np.isnan(a).sum())))
plan.sync()
print("mismatching marginals", bad, "fallbacks", plan.stats()["flow_fallbacks"])
