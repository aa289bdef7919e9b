"""Diagnostic (round 6): would timing a few placements of the arenas at plan creation and keeping the fastest pay?  Config 4's plan created
up to N times in one process - the fastest so far kept alive, every slower one closed at once (so that its pages can be handed out
again) - 40 propagates of each timed.   python3 tools/placement_probe.py [tries]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import engine, synthetic
spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
best, best_ms, seq = None, None, []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
    for _ in range(4): plan.propagate(sync=False)          # (zeroed tables: the time does not depend on the values)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(40): plan.propagate(sync=False)
    plan.sync()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    seq.append(round(ms, 4))
    if best is None or ms < best_ms:
        if best is not None: best.close()
        best, best_ms = plan, ms
    else:
        plan.close()
print("tries", seq, "-> kept %.4f" % best_ms)
best.fill_synthetic(1, spec["scales"])
for _ in range(10): best.propagate(sync=False)
best.sync()
t0 = time.perf_counter()
for _ in range(200): best.propagate(sync=False)
best.sync()
print("the kept plan with synthetic tables, 200 propagates: %.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
