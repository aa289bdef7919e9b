"""Cut a junction tree at separator edges into per-GPU parts.

The only interaction between two sides of a cut is one upward and one downward separator
message (`get_message` returns it / `send_message` receives it, `computation.py:63, 212`),
so whole subtrees can live on different GPUs.  Dependencies follow the tree, so the parts
should hang off a *small* top part: `subtree_owners` expands the heaviest frontier subtree
(its root joins the top part, its children join the frontier) until no frontier subtree is
heavier than an even share, then packs frontier subtrees onto ranks largest-first and
gives the top part to the least loaded rank.
"""

__all__ = ["subtree_owners", "part_weights", "weighted_centroid", "reroot", "partition_tree"]


def weighted_centroid(parent, weights):
    """The clique whose removal leaves no component heavier than half the tree (ties: the lowest index): rooted there,
    no subtree outweighs the rest, so the frontier of `subtree_owners` can be balanced and the quotient tree is as
    shallow as the weights allow ("re-root at the weighted centroid first")."""
    n = len(parent)
    children = [[] for _ in range(n)]
    root = 0
    for c, p in enumerate(parent):
        if p < 0:
            root = c
        else:
            children[p].append(c)
    order, stack = [], [root]
    while stack:
        c = stack.pop()
        order.append(c)
        stack.extend(children[c])
    sub = list(weights)
    for c in reversed(order):
        if parent[c] >= 0:
            sub[parent[c]] += sub[c]
    total = sub[root]
    best, best_key = root, None
    for c in range(n):
        heaviest = max([total - sub[c]] + [sub[k] for k in children[c]])
        if best_key is None or heaviest < best_key:
            best, best_key = c, heaviest
    return best


def reroot(parent, root):
    """The same tree with `root` as its root: parent list with the edges on the path root .. old root turned round."""
    new = list(parent)
    prev, c = -1, root
    while c >= 0:
        nxt = parent[c]
        new[c] = prev
        prev, c = c, nxt
    return new


def partition_tree(parent, weights, n_parts, slack=1.05, replicate_top=False):
    """`subtree_owners` on the tree re-rooted at its weighted centroid.  Returns (root, parent, owner): the plan must be
    made with that root (`engine.Plan(..., root=root)`; the replicated top part has to contain the plan's root)."""
    root = weighted_centroid(parent, weights) if n_parts > 1 else [c for c, p in enumerate(parent) if p < 0][0]
    new_parent = reroot(parent, root)
    return root, new_parent, subtree_owners(new_parent, weights, n_parts, slack=slack, replicate_top=replicate_top)


def subtree_owners(parent, weights, n_parts, slack=1.05, replicate_top=False):
    """parent[c] = parent clique (-1 for the root), weights[c] = cost of clique c (table
    bytes).  Returns owner[c] in [0, n_parts).

    `replicate_top`: the cliques of the top part get owner `n_parts` = "every rank" instead of
    being given to the least loaded rank: each rank then holds the top's tables, receives the
    upward message of every cut edge (the only exchange of a propagate) and forms the top's
    messages itself, so the downward messages need no exchange and no rank carries the top on
    top of its own subtree."""
    n = len(parent)
    if n_parts <= 1:
        return [0] * n
    children = [[] for _ in range(n)]
    root = 0
    for c, p in enumerate(parent):
        if p < 0:
            root = c
        else:
            children[p].append(c)
    order, stack = [], [root]
    while stack:
        c = stack.pop()
        order.append(c)
        stack.extend(children[c])
    sub = list(weights)
    for c in reversed(order):
        if parent[c] >= 0:
            sub[parent[c]] += sub[c]
    share = sub[root] / float(n_parts)
    frontier, top = [root], []
    while True:
        heavy = max(frontier, key=lambda c: sub[c])
        if (sub[heavy] <= share * slack and len(frontier) >= n_parts) or not children[heavy]:
            if not children[heavy] and sub[heavy] > share * slack and len(frontier) < n_parts:
                # a heavy leaf cannot be split further; try the next heaviest expandable one
                cands = [c for c in frontier if children[c]]
                if not cands:
                    break
                heavy = max(cands, key=lambda c: sub[c])
            else:
                break
        frontier.remove(heavy)
        top.append(heavy)
        frontier.extend(children[heavy])
        if not frontier:
            break
    load = [0.0] * n_parts
    owner = [0] * n
    for c in sorted(frontier, key=lambda c: -sub[c]):
        r = min(range(n_parts), key=lambda i: load[i])
        load[r] += sub[c]
        stack = [c]
        while stack:
            x = stack.pop()
            owner[x] = r
            stack.extend(children[x])
    r = n_parts if replicate_top and top else min(range(n_parts), key=lambda i: load[i])
    for c in top:
        owner[c] = r
        if r < n_parts:
            load[r] += weights[c]
    return owner


def part_weights(owner, weights, n_parts):
    """Load of every rank; a replicated clique (owner == n_parts) counts for every rank."""
    out = [0.0] * n_parts
    for o, w in zip(owner, weights):
        if o == n_parts:
            out = [x + w for x in out]
        else:
            out[o] += w
    return out
