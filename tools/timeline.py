"""Diagnostic: JTP_DEBUG=2 python tools/timeline.py [multi SETS | single] -> what the resident workgroups of a dataflow
launch are doing over time (10 us bins): staging / waiting for producers, looping, flushing; per phase."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamps
from junctiontree_amd import engine, synthetic
multi = len(sys.argv) > 1 and sys.argv[1] == "multi"
if len(sys.argv) > 1 and sys.argv[1] == "c3":          # BASELINE config 3, shortened: 6 x W lattice, cardinality 8
    import junctiontree_amd as jt
    H, W, K = 6, int(sys.argv[2]) if len(sys.argv) > 2 else 40, 8
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
else:
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", multiset=multi, n_batch=int(sys.argv[2]) if multi else 1)
plan.fill_synthetic(1, spec["scales"])
for _ in range(3):
    plan.propagate()
d, full = _stamps.read(plan)
st = _stamps.coarse(full)
kind = np.array([d["tasks"][b[0]]["kind"] for b in d["blocks"]])
BIN = float(os.environ.get("TIMELINE_BIN", "10"))
for ph in (0, 1):
    Ls = [L for L in d["launches"] if L["phase"] == ph]
    lo, hi = min(L["blk_off"] for L in Ls), max(L["blk_off"] + L["nblocks"] for L in Ls)
    s, k = st[lo:hi], kind[lo:hi]
    ok = (k == 0) & (s[:, 5] > 0)
    t0 = s[ok, 0].min()
    s = s - t0
    end = s[ok, 5].max()
    level = np.zeros(hi - lo, dtype=int)
    for L in Ls:
        level[L["blk_off"] - lo:L["blk_off"] - lo + L["nblocks"]] = L["level"]
    print("phase %d: %d pass blocks, %d reduce blocks, span %.0f us" % (ph, ok.sum(), (k != 0).sum(), end))
    print("   t(us)  resident  staging/waiting  looping  flushing | levels looping")
    for b in range(int(end / BIN) + 1):
        a, z = b * BIN, (b + 1) * BIN
        def overlap(x0, x1):
            return np.clip(np.minimum(x1, z) - np.maximum(x0, a), 0, None)[ok].sum() / BIN
        res, stg, lp, fl = overlap(s[:, 0], s[:, 5]), overlap(s[:, 0], s[:, 2]), overlap(s[:, 2], s[:, 4]), overlap(s[:, 4], s[:, 5])
        per = {}
        for lv in np.unique(level[ok]):
            m = ok & (level == lv)
            v = np.clip(np.minimum(s[m, 4], z) - np.maximum(s[m, 2], a), 0, None).sum() / BIN
            if v >= 1: per[int(lv)] = int(v)
        print("   %5.0f  %8.0f  %15.0f  %7.0f  %8.0f | %s" % (a, res, stg, lp, fl, per))
