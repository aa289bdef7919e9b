"""Synthetic junction trees and counter-based potentials.

Host-side helpers shared by `bench.py`, the tests and the golden-vector generator.  The
value generator is counter based (splitmix64 of a per-node key plus the element's linear
index in the node's *host* C-order layout), so the device can regenerate exactly the same
numbers in `jtp_fill_synthetic` (csrc/jtp_kernels.hip: `synth_value`) without shipping
GiB-sized fixtures, and numpy can regenerate them for the oracle.

Tree recipes follow BASELINE.json's configs:
  * `chain_tree`        - C2: clique i = variables {i, .., i+w-1}, separator = the w-1 shared
  * `wide_binary_tree`  - C4: parent(i) = (i-1)//2, each child shares `sep` variables drawn
                          without replacement from its parent's and adds `width-sep` fresh
  * `random_tree`       - secondary shape: parent(i) uniform in [0, i)
The nested-list tree / node-list format is the reference's (`README.md:50-77`).
"""

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_MIX1 = np.uint64(0xBF58476D1CE4E5B9)
_MIX2 = np.uint64(0x94D049BB133111EB)
_KEYMUL = np.uint64(0x100000001B3)


def splitmix64(x):
    """One splitmix64 output step for an array of uint64 states (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _MIX1
        z = (z ^ (z >> np.uint64(27))) * _MIX2
        return z ^ (z >> np.uint64(31))


def node_key(seed, node):
    with np.errstate(over="ignore"):
        return splitmix64(np.array([np.uint64(seed) * _KEYMUL + np.uint64(node)],
                                   dtype=np.uint64))[0]


def synth_values(seed, node, shape, scale=1.0, dtype=np.float64):
    """psi[idx] = (0.5 + u(idx)) * scale with u in [0,1), idx the C-order linear index."""
    n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    key = node_key(seed, node)
    with np.errstate(over="ignore"):
        bits = splitmix64(key + np.arange(n, dtype=np.uint64))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return ((0.5 + u) * float(scale)).astype(dtype).reshape(shape)


# --------------------------------------------------------------------------- tree shapes

def _nest(parent, child_lists, sep_of):
    """Build the nested-list tree bottom-up (no recursion, so chains of 1000 are fine)."""
    n = len(parent)
    sub = [None] * n
    for c in range(n - 1, -1, -1):          # children always have larger indices
        sub[c] = [c] + [(sep_of[k], sub[k]) for k in child_lists[c]]
    return sub[0]


def _assemble(parent, clique_vars, sizes):
    n = len(parent)
    child_lists = [[] for _ in range(n)]
    for c in range(1, n):
        child_lists[parent[c]].append(c)
    sep_of, sep_vars = {}, []
    for c in range(1, n):
        pset = set(clique_vars[parent[c]])
        sep_of[c] = n + len(sep_vars)
        sep_vars.append([v for v in clique_vars[c] if v in pset])
    tree = _nest(parent, child_lists, sep_of)
    scales = []
    for c in range(n):
        shared = set(sep_vars[sep_of[c] - n]) if c > 0 else set()
        free = [v for v in clique_vars[c] if v not in shared]
        scales.append(1.0 / float(np.prod([sizes[v] for v in free], dtype=np.float64)))
    return {
        "tree": tree,
        "n_cliques": n,
        "node_vars": [list(v) for v in clique_vars] + sep_vars,
        "sizes": dict(sizes),
        "parent": list(parent),
        "scales": scales,
    }


def chain_tree(n_cliques=1000, card=64, width=3):
    """C2: sliding-window chain rooted at clique 0."""
    clique_vars = [list(range(i, i + width)) for i in range(n_cliques)]
    sizes = {v: card for v in range(n_cliques + width - 1)}
    parent = [-1] + list(range(n_cliques - 1))
    return _assemble(parent, clique_vars, sizes)


def _grow(parent, width, sep, card, rng):
    clique_vars = [list(range(width))]
    next_var = width
    for c in range(1, len(parent)):
        pv = clique_vars[parent[c]]
        shared = [pv[i] for i in rng.choice(len(pv), size=sep, replace=False)]
        fresh = list(range(next_var, next_var + width - sep))
        next_var += width - sep
        clique_vars.append(shared + fresh)
    sizes = {v: card for v in range(next_var)}
    return _assemble(parent, clique_vars, sizes)


def wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0):
    """C4: balanced binary tree of wide cliques, parent(i) = (i-1)//2."""
    rng = np.random.default_rng(seed)
    parent = [-1] + [(i - 1) // 2 for i in range(1, n_cliques)]
    return _grow(parent, width, sep, card, rng)


def random_tree(n_cliques=64, width=12, sep=6, card=2, seed=0):
    """Random recursive tree: parent(i) uniform in [0, i)."""
    rng = np.random.default_rng(seed)
    parent = [-1] + [int(rng.integers(0, i)) for i in range(1, n_cliques)]
    return _grow(parent, width, sep, card, rng)


def renumber(spec, perm):
    """The same junction tree with clique c renamed perm[c] (separators keep their numbers): the
    numbering `construct_junction_tree` produces is arbitrary, the recipes above number breadth first.
    `parent` and `scales` follow the new numbering; the root is perm[0]."""
    n = spec["n_cliques"]
    perm = [int(x) for x in perm]
    assert sorted(perm) == list(range(n))
    old_parent = spec["parent"]
    kids = [[] for _ in range(n)]
    for c in range(1, n):
        kids[old_parent[c]].append(c)
    sub = [None] * n
    for c in range(n - 1, -1, -1):          # children have larger OLD indices
        sub[c] = [perm[c]] + [(n + k - 1, sub[k]) for k in kids[c]]
    node_vars = [None] * n + [list(v) for v in spec["node_vars"][n:]]
    parent, scales = [None] * n, [None] * n
    for c in range(n):
        node_vars[perm[c]] = list(spec["node_vars"][c])
        parent[perm[c]] = perm[old_parent[c]] if old_parent[c] >= 0 else -1
        scales[perm[c]] = spec["scales"][c]
    return {"tree": sub[0], "n_cliques": n, "node_vars": node_vars, "sizes": dict(spec["sizes"]),
            "parent": parent, "scales": scales}


def potentials_for(spec, seed=1, dtype=np.float64):
    """Clique potentials for a spec from the recipes above (scaled so Z is O(1)) followed
    by all-ones separators, i.e. the `potentials` list `compute_beliefs` expects."""
    n = spec["n_cliques"]
    out = []
    for node, labels in enumerate(spec["node_vars"]):
        shape = tuple(spec["sizes"][v] for v in labels)
        if node < n:
            out.append(synth_values(seed, node, shape, spec["scales"][node], dtype))
        else:
            out.append(np.ones(shape))
    return out


def algorithmic_bytes(spec, itemsize):
    """Algorithmic HBM bytes of one propagate: every clique table read once
    in collect (root excepted) and once in distribute, every belief written once, every
    separator message read once per direction and written as up, down and belief."""
    n = spec["n_cliques"]
    sz = [int(np.prod([spec["sizes"][v] for v in labels], dtype=np.int64)) if labels else 1
          for labels in spec["node_vars"]]
    cl, sp = sz[:n], sz[n:]
    reads = (2 * sum(cl) - cl[0]) * itemsize + 2 * sum(sp) * itemsize
    writes = sum(cl) * itemsize + 3 * sum(sp) * itemsize
    return {"read": reads, "write": writes, "total": reads + writes, "messages": 2 * (n - 1)}


def lattice_mrf(h=6, w=167, card=8, seed=0, dtype=np.float32):
    """BASELINE configs[2] as restated in SURVEY.md 8d: a 2-D lattice of h x w variables of one cardinality with a
    pairwise factor on every lattice edge (6 x 167: 1002 variables, 1831 factors).  Returns (factors, sizes,
    values); the values are U(0.5, 1.5) scaled by card^(-V/E) so that Z stays O(1)."""
    names = {(i, j): i * w + j for i in range(h) for j in range(w)}
    factors = []
    for i in range(h):
        for j in range(w):
            if i + 1 < h:
                factors.append([names[i, j], names[i + 1, j]])
            if j + 1 < w:
                factors.append([names[i, j], names[i, j + 1]])
    sizes = {v: card for v in names.values()}
    rng = np.random.default_rng(seed)
    scale = card ** (-len(names) / len(factors))
    values = [(rng.uniform(0.5, 1.5, (card, card)) * scale).astype(dtype) for _ in factors]
    return factors, sizes, values


def lattice_column_order(h=6, w=167):
    """The column-by-column elimination order of `lattice_mrf`'s variables ("column-sweep elimination"):
    every elimination clique is a variable, what is left of its column and the frontier in the next - h + 1 variables,
    one maximal clique per variable but the last h, a chain."""
    return [i * w + j for j in range(w) for i in range(h)]
