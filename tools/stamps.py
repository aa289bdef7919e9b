"""Diagnostic: JTP_DEBUG=2 python tools/stamps.py  -> per-level stage timings of one propagate."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import _capi, engine, synthetic
if len(sys.argv) > 1 and sys.argv[1] == "c3":
    import junctiontree_amd as jt
    H, W, K = 6, int(sys.argv[2]) if len(sys.argv) > 2 else 60, 8
    names = {(i, j): i * W + j for i in range(H) for j in range(W)}
    factors = []
    for i in range(H):
        for j in range(W):
            if i + 1 < H: factors.append([names[i, j], names[i + 1, j]])
            if j + 1 < W: factors.append([names[i, j], names[i, j + 1]])
    sizes = {v: K for v in names.values()}
    tree = jt.create_junction_tree(factors, sizes)
    node_vars = [list(c) for c in tree.clique_tree.maxcliques] + [list(s) for s in tree.separators]
    plan = engine.Plan(tree.tree, node_vars, sizes, dtype="f32")
    spec = {"scales": [8.0 ** -(len(c) - 1) for c in tree.clique_tree.maxcliques]}
elif len(sys.argv) > 1 and sys.argv[1] == "c2":
    spec = synthetic.chain_tree(n_cliques=int(sys.argv[2]) if len(sys.argv) > 2 else 64, card=64, width=3)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64")
elif len(sys.argv) > 1 and sys.argv[1] == "multi":        # multi-set plan on the width-20 tree: [sets] [layout policy]
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", multiset=True,
                       n_batch=int(sys.argv[2]) if len(sys.argv) > 2 else 8, layout_policy=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
else:
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
plan.fill_synthetic(1, spec["scales"])
for _ in range(3):
    plan.propagate()
d = plan.describe()
base, nb = d["dbg_base"], d["n_blocks"]
buf = np.empty(nb * 8)
_capi.check(plan._lib.jtp_debug_read_msg(plan._handle, 0, base, nb * 8, buf.ctypes.data_as(C.POINTER(C.c_double))))
st = buf.reshape(nb, 8)[:, :6] * 0.01      # 100 MHz ticks -> microseconds
attempts = buf.reshape(nb, 8)[:, 6]
phase_t0 = {}
only_big = len(sys.argv) > 1 and sys.argv[1] == "c3"
for L in d["launches"]:
    if only_big and L["nblocks"] < 4096:
        continue
    if L["variant"] == 16:          # reduce tasks carry no time stamps
        continue
    s = st[L["blk_off"]:L["blk_off"] + L["nblocks"]]
    phase_t0.setdefault(L["phase"], s[:, 0].min())
    print("   [since phase start: first block in %.1f us, last block in %.1f, last block out %.1f]" % (
        s[:, 0].min() - phase_t0[L["phase"]], s[:, 0].max() - phase_t0[L["phase"]], s[:, 5].max() - phase_t0[L["phase"]]))
    att = attempts[L["blk_off"]:L["blk_off"] + L["nblocks"]]
    print("   [staging attempts: mean %.2f max %d]" % (att.mean(), att.max()))
    t0 = s[:, 0].min()
    rel = s - t0
    names = ["entry", "loads issued", "staged", "consts", "loop done", "flushed"]
    print("%s level %d (%d blocks): kernel span %.1f us; first/last block start %.1f/%.1f" % (
        "collect" if L["phase"] == 0 else "distrib", L["level"], L["nblocks"], rel[:, 5].max(), rel[:, 0].min(), rel[:, 0].max()))
    dur = np.diff(s, axis=1)
    print("     median stage us: " + "  ".join("%s %.2f" % (n, v) for n, v in zip(
        ["rec+issue", "table+staging", "consts", "loop", "epilogue+flush"], np.median(dur, axis=0))) +
        "   block total median %.2f max %.2f" % (np.median(s[:, 5] - s[:, 0]), (s[:, 5] - s[:, 0]).max()))
