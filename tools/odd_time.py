"""Diagnostic: device time of one propagate on a tree of odd-cardinality cliques, thread part stored at the true
cardinalities (mixed-radix rows, kernels *_mix) against the padded power-of-two thread part (JTP_NO_TMIX=1).
    python tools/odd_time.py [card width sep n_cliques dtype]
Prints arena size, HBM bytes of the tables and ms per propagate for both layouts; nothing is checked."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import engine, synthetic
card, width, sep, n = (int(x) for x in (sys.argv[1:5] + ["3", "13", "6", "63"][len(sys.argv) - 1:4]))
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
spec = synthetic.wide_binary_tree(n_cliques=n, width=width, sep=sep, card=card, seed=0)
host = n * card ** width
for name, env in (("padded", "1"), ("mixed", "")):
    os.environ["JTP_NO_TMIX"] = env
    if not env: del os.environ["JTP_NO_TMIX"]
    levels = bool(os.environ.get("ODD_LEVELS"))         # ODD_LEVELS=1: one launch per level, every launch timed
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, level_launches=levels)
    plan.fill_synthetic(1, [float(card) ** -(width - 1)] * n)
    for _ in range(3): plan.propagate()
    plan.set_profiling(3, per_launch=levels, stride=1) if levels else plan.set_profiling(3)
    for _ in range(10): plan.propagate()
    st, d = plan.stats(), plan.describe()
    ms = st["collect_ms"] + st["distribute_ms"]
    print("%-6s tmix %d arena %.3f x host (%d MB)  launches %d  %.3f ms/propagate  %.2f TB/s of host-table bytes (3 passes)" % (
        name, d["tmix"], d["arena_elems"] / host, d["arena_elems"] * (4 if dtype == "f32" else 8) >> 20, st["n_launches"], ms,
        3 * host * (4 if dtype == "f32" else 8) / ms / 1e9))
    if levels:
        for L in plan.launch_ms():
            print("   %s level %d: %3d tasks %5d blocks %8.4f ms %7.0f GB/s" % ("collect   " if L["phase"] == 0 else "distribute", L["level"], L["ntasks"], L["nblocks"],
                                                                           L["ms"], L["alg_bytes"] / max(L["ms"], 1e-9) / 1e6))
    plan.close()
