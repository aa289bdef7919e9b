"""Diagnostic: JTP_DEBUG=2 python tools/stamps_blocks.py [sets] -> the slowest workgroups of the multi-set leaf level."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamps
from junctiontree_amd import engine, synthetic
spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
multi = len(sys.argv) > 1
plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", multiset=multi, n_batch=int(sys.argv[1]) if multi else 1)
plan.fill_synthetic(1, spec["scales"])
for _ in range(3):
    plan.propagate()
d, full = _stamps.read(plan)
st = _stamps.coarse(full)
for L in d["launches"][:4]:
    if L["variant"] == 16:
        continue
    s = st[L["blk_off"]:L["blk_off"] + L["nblocks"]]
    t0 = s[:, 0].min()
    dur = s[:, 5] - s[:, 0]
    order = np.argsort(-dur)[:12]
    print("phase %d level %d: %d blocks; durations: median %.1f p90 %.1f p99 %.1f max %.1f us" % (L["phase"], L["level"], L["nblocks"], np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
    for i in order:
        print("   block %5d: start %.1f  stages %s" % (i, s[i, 0] - t0, " ".join("%.1f" % x for x in np.diff(s[i]))))
    hist, edges = np.histogram(dur, bins=10)
    print("   histogram:", list(zip(np.round(edges[:-1], 0), hist)))
