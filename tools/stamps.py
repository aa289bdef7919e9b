"""Diagnostic: JTPROP_LIB=<build with -DJT_STAMPS> JTP_DEBUG=2 python tools/stamps.py [c2 N | c3 W | multi SETS | ranks WORLD RANK]
-> per-level stage timings of one propagate, and how long after the previous level's last workgroup each level's last one ends."""
import os, sys, warnings
import numpy as np
warnings.filterwarnings("ignore", category=RuntimeWarning)      # medians over stage slots a short loop never reaches
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _stamps
from junctiontree_amd import engine, partition, synthetic
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
    order = [names[i, j] for j in range(W) for i in range(H)] if os.environ.get("STAMPS_SWEEP") else None
    tree = jt.create_junction_tree(factors, sizes, order=order)
    node_vars = [list(c) for c in tree.clique_tree.maxcliques] + [list(s) for s in tree.separators]
    # (STAMPS_NO_COVER=1: every clique keeps a full table, as in rounds 1-4)
    plan = engine.Plan(tree.tree, node_vars, sizes, dtype="f32", cover=None if os.environ.get("STAMPS_NO_COVER") else tree.cover())
    spec = {"scales": [8.0 ** -(len(c) - 1) for c in tree.clique_tree.maxcliques]}
elif len(sys.argv) > 1 and sys.argv[1] == "c2":
    spec = synthetic.chain_tree(n_cliques=int(sys.argv[2]) if len(sys.argv) > 2 else 64, card=64, width=3)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64")
elif len(sys.argv) > 1 and sys.argv[1] == "multi":        # multi-set plan on the width-20 tree: [sets] [layout policy]
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", multiset=True,
                       n_batch=int(sys.argv[2]) if len(sys.argv) > 2 else 8, layout_policy=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
elif len(sys.argv) > 1 and sys.argv[1] == "ranks":        # one rank's share of the sharded width-20 tree (JTP_FAKE_COMM=1)
    os.environ.setdefault("JTP_FAKE_COMM", "1")
    world, rank = int(sys.argv[2]), int(sys.argv[3])
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    owner = partition.subtree_owners(spec["parent"], [1.0] * spec["n_cliques"], world, replicate_top=True)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", n_ranks=world, rank=rank, owner=owner)
else:
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
plan.fill_synthetic(1, spec["scales"])
for _ in range(3):
    plan.propagate()
d, full = _stamps.read(plan)
st = _stamps.coarse(full)
attempts = full[:, 13]
phase_t0, prev_out = {}, {}
summary = {0: [], 1: []}            # STAMPS_SUMMARY=1: medians over the levels of a phase instead of a line per level
only_big = len(sys.argv) > 1 and sys.argv[1] == "c3"
fine_names = ["rec+issue", "1st attempt", "wait+restage", "consts", "step0", "step1", "step2", "step3", "more steps", "epilogues", "flush issue", "flush retire"]
fine_cols = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
for L in d["launches"]:
    if only_big and L["nblocks"] < 1000:
        continue
    if L["variant"] == 16:          # reduce tasks carry no time stamps
        continue
    s = st[L["blk_off"]:L["blk_off"] + L["nblocks"]]
    f = full[L["blk_off"]:L["blk_off"] + L["nblocks"]]
    ok = s[:, 5] > 0
    if not ok.any():
        continue
    s, f = s[ok], f[ok]
    phase_t0.setdefault(L["phase"], s[:, 0].min())
    t0p = phase_t0[L["phase"]]
    out = s[:, 5].max() - t0p
    print("%s level %d (%d blocks, %d tasks): first block in %.1f us, last in %.1f, LAST OUT %.1f (%+.1f after the level before); attempts mean %.2f max %d" % (
        "collect" if L["phase"] == 0 else "distrib", L["level"], L["nblocks"], len(L["tasks"]), s[:, 0].min() - t0p, s[:, 0].max() - t0p, out,
        out - prev_out.get(L["phase"], 0.0), attempts[L["blk_off"]:L["blk_off"] + L["nblocks"]][ok].mean(), attempts[L["blk_off"]:L["blk_off"] + L["nblocks"]][ok].max()))
    fd0 = np.diff(f[:, fine_cols], axis=1)
    fd0[f[:, fine_cols][:, 1:] <= 0] = np.nan
    w8 = f[:, 14] > 0
    with np.errstate(all="ignore"):
        summary[L["phase"]].append([out - prev_out.get(L["phase"], 0.0), np.median(f[w8, 3] - f[w8, 14]) if w8.any() else np.nan,
                                    (s[:, 5].max() - f[w8, 14].max()) if w8.any() else np.nan] + list(np.nanmedian(fd0, axis=0)))
    prev_out[L["phase"]] = out
    if os.environ.get("STAMPS_SUMMARY"):
        continue
    dur = np.diff(s, axis=1)
    print("     median stage us: " + "  ".join("%s %.2f" % (n, v) for n, v in zip(
        ["rec+issue", "table+staging", "consts", "loop", "epilogue+flush"], np.median(dur, axis=0))) +
        "   block total median %.2f max %.2f" % (np.median(s[:, 5] - s[:, 0]), (s[:, 5] - s[:, 0]).max()))
    # the workgroups that end last: where did THEIR time go after their inputs arrived
    fd = np.diff(f[:, fine_cols], axis=1)
    fd[f[:, fine_cols][:, 1:] <= 0] = np.nan          # (slots a short loop never reaches)
    waited = f[:, 14] > 0
    if waited.any():                                   # producer arrival seen -> staged: the re-staging after the last wait
        print("     after the last wait (%d of %d blocks waited): re-stage median %.2f us (p90 %.2f); last wait ended %.2f us before the level's last block out" % (
            waited.sum(), len(f), np.median(f[waited, 3] - f[waited, 14]), np.percentile(f[waited, 3] - f[waited, 14], 90), s[:, 5].max() - f[waited, 14].max()))
    last = np.argsort(-s[:, 5])[:max(1, len(s) // 20)]
    with np.errstate(all="ignore"):
        print("     fine, median of all  : " + "  ".join("%s %.2f" % (n, v) for n, v in zip(fine_names, np.nanmedian(fd, axis=0))))
        print("     fine, last 5%% to end : " + "  ".join("%s %.2f" % (n, v) for n, v in zip(fine_names, np.nanmedian(fd[last], axis=0))))

if os.environ.get("STAMPS_SUMMARY"):
    for ph in (0, 1):
        a = np.array(summary[ph][2:-2] if len(summary[ph]) > 8 else summary[ph])
        if len(a) == 0:
            continue
        with np.errstate(all="ignore"):
            med = np.nanmedian(a, axis=0)
        print("%s: %d levels; median over levels: level increment %.2f us, re-stage after the last wait %.2f, last wait -> last block out %.2f" % (
            "collect" if ph == 0 else "distribute", len(a), med[0], med[1], med[2]))
        print("     stages: " + "  ".join("%s %.2f" % (n, v) for n, v in zip(fine_names, med[3:])))
