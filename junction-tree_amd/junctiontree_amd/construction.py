"""Compile phase: factor graph -> maximal cliques -> junction tree (structure only).

The reference does this in `junctiontree/construction.py` (`find_triangulation` :176-353,
`construct_junction_tree` :522-578).  It is outside the accelerated hot path (SURVEY.md
section 2 row 5: combinatorial, once per model), but `create_junction_tree` needs *a*
correct builder, and marginals do not depend on which valid junction tree is used.  This
is an independent implementation:

* `triangulate`  greedy min-fill elimination (ties: smaller clique weight, then first seen)
                 with incremental fill-count maintenance; maximal elimination cliques only.
* `construct_junction_tree`  Kruskal maximum-weight spanning tree over sepset sizes
                 (ties: smaller separator table), union-find, components joined by empty
                 separators as the reference does (`construction.py:530`); the result is the
                 reference's nested-list format (`README.md:50-77`) rooted at clique 0, with
                 separator node indices `len(cliques) + k`.
Everything is iterative (no recursion limits on long chains).
"""

from itertools import combinations

__all__ = ["triangulate", "find_triangulation", "construct_junction_tree", "nest_tree"]


def _first_seen_order(factors, sizes):
    order = {}
    for f in factors:
        for v in f:
            order.setdefault(v, len(order))
    for v in sizes:
        # variables that appear in no factor are ignored, like the reference does
        pass
    return order


def triangulate(factors, sizes, order=None):
    """Return (maxcliques, factor_to_maxclique) for the factor graph.

    `order`: an elimination order to follow instead of greedy min-fill - a sequence of variables, eliminated first in
    that order (variables it does not list follow by min-fill).  On a lattice the column-by-column order gives the
    chain of width-(h+1) cliques of SURVEY.md 8d, where min-fill trades a shallower tree for some wider cliques."""
    rank = _first_seen_order(factors, sizes)
    adj = {v: set() for v in rank}
    for f in factors:
        for a, b in combinations(f, 2):
            if a != b:
                adj[a].add(b)
                adj[b].add(a)

    def fill_of(v):
        nb = adj[v]
        missing = 0
        for a in nb:
            missing += len(nb) - 1 - len(adj[a] & nb)
        return missing // 2

    def weight_of(v):
        w = sizes[v]
        for a in adj[v]:
            w *= sizes[a]
        return w

    forced = [v for v in (order or []) if v in adj]
    if len(set(forced)) != len(forced):
        raise ValueError("the elimination order lists a variable twice")
    forced.reverse()                                  # (popped from the end)
    # (costs are only looked at once the given order is used up: built then, for the variables that are left, and kept up to date
    #  for the touched ones from there on - a partial order of k variables used to recompute every remaining variable's cost on
    #  each of its k steps)
    cost = {v: (fill_of(v), weight_of(v), rank[v]) for v in adj} if not forced else None
    remaining = set(adj)
    cliques, member_of = [], {v: [] for v in adj}
    while remaining:
        v = forced.pop() if forced else min(remaining, key=cost.__getitem__)
        nb = adj[v]
        cand = [v] + sorted(nb, key=rank.__getitem__)
        cset = set(cand)
        if not any(cset <= cliques[i][1] for i in member_of[v]):
            idx = len(cliques)
            cliques.append((sorted(cand, key=rank.__getitem__), cset))
            for u in cand:
                member_of[u].append(idx)
        touched = set(nb)
        for a, b in combinations(nb, 2):
            if b not in adj[a]:
                adj[a].add(b)
                adj[b].add(a)
        for a in nb:
            adj[a].discard(v)
            touched |= adj[a]
        remaining.discard(v)
        del adj[v]
        if forced or not remaining:
            continue
        if cost is None:                              # the given order has just run out
            cost = {u: (fill_of(u), weight_of(u), rank[u]) for u in remaining}
        else:
            for u in touched:
                if u in remaining:
                    cost[u] = (fill_of(u), weight_of(u), rank[u])

    maxcliques = [c[0] for c in cliques]
    factor_to_maxclique = []
    for f in factors:
        fset = set(f)
        hosts = member_of[f[0]] if f else range(len(cliques))
        factor_to_maxclique.append(next(i for i in hosts if fset <= cliques[i][1]))
    return maxcliques, factor_to_maxclique


def find_triangulation(factors, sizes):
    """Reference-shaped signature (`construction.py:176`): (fill-in edges, maxcliques,
    factor_to_maxclique).  The fill-in edge list is not used by the hot path; it is derived
    here from the cliques for completeness."""
    maxcliques, f2m = triangulate(factors, sizes)
    original = set()
    for f in factors:
        for a, b in combinations(f, 2):
            original.add(frozenset((a, b)))
    tri = []
    seen = set()
    for c in maxcliques:
        for a, b in combinations(c, 2):
            e = frozenset((a, b))
            if e not in original and e not in seen:
                seen.add(e)
                tri.append((a, b))
    return tri, maxcliques, f2m


def nest_tree(parent, children, sep_of):
    """Flat (parent, ordered child lists, separator index per child) -> nested list."""
    n = len(parent)
    # post-order without recursion
    root = parent.index(-1)
    order, stack = [], [root]
    while stack:
        c = stack.pop()
        order.append(c)
        stack.extend(children[c])
    sub = {}
    for c in reversed(order):
        sub[c] = [c] + [(sep_of[k], sub[k]) for k in children[c]]
    return sub[root]


def construct_junction_tree(cliques, sizes):
    """Return (tree, separators) like the reference's `construct_junction_tree`."""
    n = len(cliques)
    sets = [set(c) for c in cliques]
    where = {}
    for i, c in enumerate(cliques):
        for v in c:
            where.setdefault(v, []).append(i)
    pairs = set()
    for idxs in where.values():
        for a, b in combinations(idxs, 2):
            pairs.add((a, b))

    def table(vs):
        t = 1
        for v in vs:
            t *= sizes[v]
        return t

    scored = []
    for a, b in pairs:
        shared = sets[a] & sets[b]
        scored.append((-len(shared), table(shared), a, b))
    scored.sort()

    root_of = list(range(n))

    def find(x):
        while root_of[x] != x:
            root_of[x] = root_of[root_of[x]]
            x = root_of[x]
        return x

    nbrs = [[] for _ in range(n)]
    for _, _, a, b in scored:
        ra, rb = find(a), find(b)
        if ra != rb:
            root_of[ra] = rb
            nbrs[a].append(b)
            nbrs[b].append(a)
    # join disconnected components with empty separators
    comps = sorted(set(find(i) for i in range(n)))
    reps = {}
    for i in range(n):
        reps.setdefault(find(i), i)
    first = reps[find(0)] if n else None
    for r in comps:
        if n and find(r) != find(0):
            a, b = first, reps[r]
            root_of[find(b)] = find(a)
            nbrs[a].append(b)
            nbrs[b].append(a)

    parent = [-2] * n
    children = [[] for _ in range(n)]
    sep_of, separators = {}, []
    if n:
        parent[0] = -1
        queue = [0]
        for c in queue:
            for k in nbrs[c]:
                if parent[k] == -2:
                    parent[k] = c
                    children[c].append(k)
                    sep_of[k] = n + len(separators)
                    separators.append([v for v in cliques[k] if v in sets[c]])
                    queue.append(k)
    tree = nest_tree(parent, children, sep_of) if n else []
    return tree, separators
