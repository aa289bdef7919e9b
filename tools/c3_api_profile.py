"""Diagnostic: where the host time of `tree.propagate(values)` goes on BASELINE config 3 (6 x 167 lattice, 1831 factor tables, all new
on every call): cProfile of ten calls, cumulative times of the package's own functions."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
import junctiontree_amd as jt
from junctiontree_amd import synthetic
factors, sizes, values = synthetic.lattice_mrf(6, int(sys.argv[1]) if len(sys.argv) > 1 else 167, 8)
tree = jt.create_junction_tree(factors, sizes)
tree.propagate(values)
plan = tree.plan("f32")
sets = [[v * np.float32(1.0 + 1e-3 * (r + 1)) for v in values] for r in range(10)]
kw = {"changed": "all"} if os.environ.get("C3_CHANGED") else {}
for vals in sets[:2]:
    tree.propagate(vals, **kw)
times = []
for vals in sets:
    plan.sync()
    t0 = time.perf_counter()
    tree.propagate(vals, **kw)
    times.append((time.perf_counter() - t0) * 1e3)
print("tree.propagate, all tables new: min %.2f ms median %.2f ms" % (min(times), sorted(times)[len(times) // 2]))
plan.set_profiling(3)
for _ in range(4):
    plan.propagate()
st = plan.stats()
print("hot path (device): collect %.2f + distribute %.2f ms" % (st["collect_ms"], st["distribute_ms"]))
pr = cProfile.Profile()
pr.enable()
for vals in sets:
    tree.propagate(vals, **kw)
pr.disable()
ps = pstats.Stats(pr, stream=sys.stdout)
ps.sort_stats("cumulative").print_stats(28)
