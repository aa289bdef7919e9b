"""Diagnostic: JTP_DEBUG=2 python tools/dispatch_probe.py [W] -> config 3 (6 x W lattice): how workgroups of one distribute
level start and end over time, per XCD (block index mod 8), to see what keeps CU slots empty."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
import junctiontree_amd as jt
from junctiontree_amd import _capi, engine
H, W, K = 6, int(sys.argv[1]) if len(sys.argv) > 1 else 40, 8
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
plan.fill_synthetic(1, [8.0 ** -(len(c) - 1) for c in tree.clique_tree.maxcliques])
for _ in range(3):
    plan.propagate()
d = plan.describe()
base, nb = d["dbg_base"], d["n_blocks"]
buf = np.empty(nb * 8)
_capi.check(plan._lib.jtp_debug_read_msg(plan._handle, 0, base, nb * 8, buf.ctypes.data_as(C.POINTER(C.c_double))))
st = buf.reshape(nb, 8)[:, :6] * 0.01
kind = np.array([d["tasks"][b[0]]["kind"] for b in d["blocks"]])
task = np.array([b[0] for b in d["blocks"]])
Ls = [L for L in d["launches"] if L["phase"] == 1 and L["variant"] != 16]
L = sorted(Ls, key=lambda L: L["nblocks"])[-max(1, len(Ls) // 4)]        # one of the large levels
lo, hi = L["blk_off"], L["blk_off"] + L["nblocks"]
print("distribute level %d: %d blocks, tasks %s" % (L["level"], L["nblocks"], sorted(set(task[lo:hi].tolist()))))
s = st[lo:hi]; k = kind[lo:hi]
ok = (k == 0) & (s[:, 5] > 0)
t0 = s[ok, 0].min()
for t in sorted(set(task[lo:hi].tolist())):
    m = ok & (task[lo:hi] == t)
    if not m.any(): continue
    tk = d["tasks"][t]
    dur = s[m, 5] - s[m, 0]
    print("  task %d (%d in, %d out, %d iterations, lds %d): %d blocks, start %.0f..%.0f us, duration median %.1f p90 %.1f max %.1f; stages median %s" % (
        t, tk["n_in"], tk["n_out"], tk["total"], tk["lds_bytes"], m.sum(), s[m, 0].min() - t0, s[m, 0].max() - t0, np.median(dur), np.percentile(dur, 90), dur.max(),
        " ".join("%.1f" % x for x in np.median(np.diff(s[m], axis=1), axis=0))))
# residency per XCD over time
idx = np.arange(lo, hi)[ok]
for x in range(8):
    m = (idx % 8) == x
    ss = s[ok][m]
    span = ss[:, 5].max() - ss[:, 0].min()
    print("  XCD %d: %d blocks, busy slot-time / span = %.0f resident on average (of 96-128 slots)" % (x, m.sum(), (ss[:, 5] - ss[:, 0]).sum() / span))
# start-time gaps along the block order
starts = s[ok, 0] - t0
order = np.argsort(idx)
gaps = np.diff(starts[order])
print("  start gaps along the block order: median %.3f us, p99 %.2f, max %.1f; blocks starting later than a successor: %d" % (np.median(gaps), np.percentile(gaps, 99), gaps.max(), (gaps < 0).sum()))
