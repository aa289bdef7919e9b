import os, sys
import numpy as np
sys.path.insert(0, "junction-tree_amd")
import junctiontree_amd as jt
from junctiontree_amd import engine
nv, card, dt = 6, 5, np.float32
rng = np.random.default_rng(0)
names = list("abcdefghi")[:nv]
sizes = {v: card for v in names}
factors = [names, names[:2], names[-3:]]
values = [rng.uniform(0.2, 1.0, [sizes[v] for v in f]).astype(dt) for f in factors]
tree = jt.create_junction_tree(factors, sizes)
got = tree.propagate(values)
plan = tree.plan("f32")
ax = {v: i for i, v in enumerate(names)}
ops = []
for f, val in zip(factors, values):
    ops += [np.asarray(val, dtype=np.float64), [ax[v] for v in f]]
joint = np.einsum(*ops, list(range(nv)))
clique = tree.clique_tree.maxcliques[0]
bel = plan.belief(0)
want = np.einsum(joint, list(range(nv)), [ax[v] for v in clique])
print("env", {k: v for k, v in os.environ.items() if k.startswith("JTP_")})
print("belief (= psi, evaluate) max rel err", float(np.max(np.abs(bel - want) / want)))
for f, g in zip(factors, got):
    w = np.einsum(joint, list(range(nv)), [ax[v] for v in f])
    print("marginal", f, "max rel err", float(np.max(np.abs(g - w) / w)))
d = plan.describe(); pk = d["pack"][0]
bad = np.argwhere(np.abs(bel - want) / want > 1e-5)
print("bad entries", len(bad), "of", bel.size)
# host digits (clique axis order = pack variable order) -> device place
def place(digs):
    x = 0
    for i, dg in enumerate(digs):
        if i == pk["split_var"]:
            x += (dg & ((1 << pk["split_lb"]) - 1)) * pk["dstride"][i] + (dg >> pk["split_lb"]) * pk["split_ds2"]
        else:
            x += dg * pk["dstride"][i]
    return x
import collections
rows = collections.Counter(); ts = collections.Counter()
for b in bad:
    x = place(list(b)); rows[x // 252] += 1; ts[(x % 252) // 125] += 1
print("bad by row", sorted(rows.items())[:80]); print("bad by low digit of split var", ts)
for b in bad[:12]:
    x = place(list(b)); print(list(b), "x", x, "row", x // 252, "t", x % 252, "got/want", bel[tuple(b)] / want[tuple(b)])
