"""CPU oracle for the junction-tree message-passing hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the algorithm that jluttine/junction-tree runs in
`JunctionTree.propagate` -> `computation.compute_beliefs`.  It exists so that the HIP
engine in `junction-tree_amd/` can be checked against an independent CPU computation.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
import it; the product package never does (it fails loudly when the HIP library is
missing instead of falling back to anything here).

Parity status: PINNED.  `oracle/gen_golden.py` imports the unmodified reference from
`/root/reference` (in the build container only) and stores its outputs as fixtures under
`tests/golden/`; `tests/test_oracle_golden.py` checks every function below against those
fixtures and against the hard-coded known answers of the reference's own tests.

Three computations are provided, each citing the reference lines it follows:

* `beliefs_exact`      - the *intended* mathematics of the two-pass algorithm (SURVEY.md
                         Appendix A): collect = `computation.py:47-96`, distribute =
                         `computation.py:140-224`, with the all-but-one product computed
                         directly instead of by `remove_message`'s divide-out
                         (`computation.py:99-136`, which is positionally mis-aligned on
                         general inputs - SURVEY.md Appendix B).  This is the checker for
                         the GPU path.
* `beliefs_refshaped`  - the same 5N-1 `numpy.einsum` call sequence the reference issues,
                         including the full-scope message product and the zero-guarded
                         divide-out, but with axes aligned by name.  Used as the CPU
                         baseline (`cpu_baseline.kind == "port"`): same operation count
                         and temporaries as the reference.
* `beliefs_bruteforce` - one einsum over the whole joint, the reference's own test oracle
                         (`tests/test_computation.py:19-32`).

plus `evaluate` / `marginalize` / `propagate` restating `junctiontree.py:203-331`.

Data structures are the reference's (`README.md:43-77`): `tree` is the nested list
`[clique_ix, (sep_ix, subtree), ...]`, `node_vars[i]` the label list of node i, and
`potentials[i]` a numpy array with one axis per label (an axis may have length 1 while
other nodes carry the same label at full length: numpy broadcasting, Appendix A.5).
"""

import numpy as np

__all__ = [
    "flatten_tree", "beliefs_exact", "beliefs_refshaped", "beliefs_bruteforce",
    "evaluate", "marginalize", "propagate", "labelled_einsum",
]


# --------------------------------------------------------------------------- helpers

def labelled_einsum(*args):
    """einsum in interleaved form with arbitrary hashable labels.

    Restates `sum_product.py:22-43`: labels are renumbered to small ints per call and the
    call is forwarded to `numpy.einsum` (explicit output form).  Unlike the reference the
    numbering is by first appearance, so it does not depend on `set` iteration order.
    """
    ops = list(args[:-1])
    out = list(args[-1])
    number = {}
    for labels in ops[1::2] + [out]:
        for lab in labels:
            number.setdefault(lab, len(number))
    if len(number) > 52:
        raise ValueError("numpy.einsum supports at most 52 distinct labels per call")
    call = []
    for arr, labels in zip(ops[0::2], ops[1::2]):
        call += [np.asarray(arr), [number[lab] for lab in labels]]
    call.append([number[lab] for lab in out])
    return np.einsum(*call)


def flatten_tree(tree):
    """Nested-list junction tree -> flat arrays, without recursion.

    Returns (order, parent, parent_sep, children) where `order` lists clique indices in
    pre-order (root first), `parent[c]` is the parent clique (-1 for the root),
    `parent_sep[c]` the separator node between c and its parent, and `children[c]` a list
    of (sep_ix, child_clique) in the order given in the tree.  The tree format is that of
    `README.md:50-66`; tuples and lists are interchangeable as in the reference's tests.
    """
    order, parent, parent_sep, children = [], {}, {}, {}
    stack = [(tree, -1, -1)]
    while stack:
        sub, par, sep = stack.pop()
        c = sub[0]
        order.append(c)
        parent[c], parent_sep[c] = par, sep
        kids = []
        for entry in sub[1:]:
            sep_ix, child = entry[0], entry[1]
            kids.append((sep_ix, child[0]))
        children[c] = kids
        for entry in reversed(sub[1:]):
            stack.append((entry[1], c, entry[0]))
    return order, parent, parent_sep, children


def _ordered_union(label_lists):
    seen, out = set(), []
    for labels in label_lists:
        for lab in labels:
            if lab not in seen:
                seen.add(lab)
                out.append(lab)
    return out


# --------------------------------------------------------------------------- exact

def beliefs_exact(tree, potentials, node_vars, return_z=False):
    """Two-pass sum-product with direct all-but-one products (the checker).

    Collect (`computation.py:47-96`): post-order,
        up_c[S_p] = sum_{C \\ S_p} psi_c * prod_k up_k.
    Distribute (`computation.py:140-224`): pre-order, for each child k
        down_k[S_k] = sum_{C \\ S_k} psi_c * down_p * prod_{j != k} up_j,
        belief[S_k] = up_k * down_k                       (`computation.py:210`)
    and belief[C] = psi_c * down_p * prod_k up_k          (`computation.py:216-224`).
    Results are unnormalised: every returned table sums to Z (Appendix A.3).
    """
    order, parent, parent_sep, children = flatten_tree(tree)
    psi = [np.asarray(p, dtype=np.float64) for p in potentials]
    beliefs = list(psi)
    up, down = {}, {}

    for c in reversed(order):                       # leaves first
        ops = [psi[c], node_vars[c]]
        for sep, child in children[c]:
            ops += [up[child], node_vars[sep]]
        out = node_vars[parent_sep[c]] if parent[c] >= 0 else []
        up[c] = labelled_einsum(*ops, out)
    z = float(up[order[0]])

    for c in order:                                 # root first
        base = [psi[c], node_vars[c]]
        if parent[c] >= 0:
            base += [down[c], node_vars[parent_sep[c]]]
        kids = children[c]
        for i, (sep, child) in enumerate(kids):
            ops = list(base)
            for j, (sep_j, child_j) in enumerate(kids):
                if j != i:
                    ops += [up[child_j], node_vars[sep_j]]
            down[child] = labelled_einsum(*ops, node_vars[sep])
            beliefs[sep] = up[child] * down[child]
        ops = list(base)
        for sep, child in kids:
            ops += [up[child], node_vars[sep]]
        beliefs[c] = labelled_einsum(*ops, node_vars[c])
    return (beliefs, z) if return_z else beliefs


# --------------------------------------------------------------------------- reference-shaped

def _divide_out(msg_prod, prod_vars, msg, msg_vars, keep_vars):
    """Divide `msg` out of `msg_prod` and drop the axes that only `msg` carried.

    Restates `remove_message` (`computation.py:99-136`): zero-guarded division into a fresh
    zeros array (`:131-136`), then index 0 on every axis of `prod_vars` that is not in
    `keep_vars` (`:116-120`).  Here the message is aligned to the product's axes *by
    label* (the reference aligns by position, correct only when the separator's listed
    order matches its order inside `prod_vars`).
    """
    msg = np.asarray(msg)
    src = [v for v in prod_vars if v in msg_vars]
    aligned = labelled_einsum(msg, msg_vars, src)
    shape = [aligned.shape[src.index(v)] if v in src else 1 for v in prod_vars]
    aligned = aligned.reshape(shape)
    quotient = np.divide(msg_prod, aligned, out=np.zeros_like(msg_prod), where=aligned != 0)
    index = tuple(slice(None) if v in keep_vars else 0 for v in prod_vars)
    return quotient[index]


def beliefs_refshaped(tree, potentials, node_vars, counters=None):
    """Same einsum sequence as the reference's `compute_beliefs` (5N-1 calls).

    Per clique in collect: K1 product of child messages at the union scope
    (`computation.py:79-82`; a leaf multiplies the int 1, `:77`) and K2 marginalisation
    with the clique potential (`:84-88`).  Per clique in distribute: K3 product of all
    incoming messages (`:169-172`), per child K4 divide-out (`:197-203`) + K5
    marginalisation (`:205-207`) + K6 separator update (`:210`), then K7 clique belief
    (`:216-224`).  Potentials are copied first (`:245`).  Iterative, so deep chains do not
    hit the recursion limit the reference hits.  `counters`, if given, is
    a dict receiving the number of einsum calls made.
    """
    order, parent, parent_sep, children = flatten_tree(tree)
    beliefs = [np.copy(p) for p in potentials]
    n_einsum = 0

    for c in reversed(order):
        kids = children[c]
        scope = _ordered_union([node_vars[sep] for sep, _ in kids])
        ops = []
        for sep, _ in kids:
            ops += [beliefs[sep], node_vars[sep]]
        if not ops:
            ops = [1, []]
        msg_prod = labelled_einsum(*ops, scope)
        out = node_vars[parent_sep[c]] if parent[c] >= 0 else []
        message = labelled_einsum(msg_prod, scope, beliefs[c], node_vars[c], out)
        n_einsum += 2
        if parent[c] >= 0:
            beliefs[parent_sep[c]] = message

    incoming = {order[0]: (np.array(1), [])}
    for c in order:
        kids = children[c]
        msg_in, msg_in_vars = incoming.pop(c)
        scopes = [node_vars[sep] for sep, _ in kids] + [msg_in_vars]
        scope = _ordered_union(scopes)
        ops = []
        for sep, _ in kids:
            ops += [beliefs[sep], node_vars[sep]]
        ops += [msg_in, msg_in_vars]
        msg_prod = labelled_einsum(*ops, scope)
        n_einsum += 1
        for i, (sep, child) in enumerate(kids):
            others = set()
            for j, labels in enumerate(scopes):
                if j != i:
                    others.update(labels)
            keep = [v for v in scope if v in others]
            mod = _divide_out(msg_prod, scope, beliefs[sep], node_vars[sep], keep)
            message = labelled_einsum(mod, keep, beliefs[c], node_vars[c], node_vars[sep])
            n_einsum += 1
            beliefs[sep] = beliefs[sep] * message
            incoming[child] = (message, node_vars[sep])
        beliefs[c] = labelled_einsum(beliefs[c], node_vars[c], msg_prod, scope, node_vars[c])
        n_einsum += 1
    if counters is not None:
        counters["einsum_calls"] = n_einsum
    return beliefs


# --------------------------------------------------------------------------- brute force

def beliefs_bruteforce(tree, potentials, node_vars):
    """Marginals of the full joint, one einsum per node (`tests/test_computation.py:19-32`).

    Every node array (cliques and separators, in the tree's depth-first order,
    `tests/test_computation.py:6-16`) is an operand, and each node's own scope is the
    output.  Only feasible while the joint has <= 52 labels and fits in time.
    Returns a list indexed like `potentials`.
    """
    order, parent, parent_sep, children = flatten_tree(tree)
    ops = []
    for c in order:
        ops += [potentials[c], node_vars[c]]
        for sep, _ in children[c]:
            ops += [potentials[sep], node_vars[sep]]
    out = [None] * len(potentials)
    for c in order:
        out[c] = labelled_einsum(*ops, node_vars[c])
        for sep, _ in children[c]:
            out[sep] = labelled_einsum(*ops, node_vars[sep])
    return out


# --------------------------------------------------------------------------- propagate

def _einsum_new_axes(arrays, array_vars, out_vars):
    """Product of `arrays` onto `out_vars`, creating length-1 axes for labels that no
    input carries (`junctiontree.py:34-80`, in particular `:52-61`)."""
    have = set(v for labels in array_vars for v in labels)
    arrays = [np.asarray(a) for a in arrays]
    array_vars = [list(v) for v in array_vars]
    missing = [v for v in out_vars if v not in have]
    if missing:
        arrays[0] = arrays[0].reshape((1,) * len(missing) + arrays[0].shape)
        array_vars[0] = missing + array_vars[0]
    ops = []
    for a, labels in zip(arrays, array_vars):
        ops += [a, labels]
    return labelled_einsum(*ops, list(out_vars))


def evaluate(factors, factor_to_maxclique, maxcliques, values):
    """Clique potentials from factor values (`junctiontree.py:203-226`): for each maximal
    clique the product of the factors assigned to it, in the clique's axis order;
    variables not covered by any assigned factor stay length-1 axes."""
    out = []
    for ci, clique in enumerate(maxcliques):
        members = [fi for fi, mc in enumerate(factor_to_maxclique) if mc == ci]
        if not members:
            out.append(np.ones((1,) * len(clique)))
            continue
        out.append(_einsum_new_axes([values[fi] for fi in members],
                                    [factors[fi] for fi in members], clique))
    return out


def marginalize(factors, factor_to_maxclique, maxcliques, clique_beliefs):
    """Factor marginals from clique beliefs (`junctiontree.py:229-274`): sum the clique
    axes that are not in the factor, output in the factor's axis order."""
    return [labelled_einsum(clique_beliefs[mc], maxcliques[mc], list(fvars))
            for fvars, mc in zip(factors, factor_to_maxclique)]


def propagate(tree, separators, maxcliques, factor_to_maxclique, factors, sizes, values,
              engine=beliefs_exact):
    """`JunctionTree.propagate` (`junctiontree.py:297-331`): evaluate -> separators of
    ones (`:312-315`, sized from `sizes` at call time) -> beliefs -> factor marginals.
    Clique potentials with length-1 auxiliary axes are broadcast to full shape first so
    that the distribute phase is well defined."""
    psi = evaluate(factors, factor_to_maxclique, maxcliques, values)
    psi = [np.broadcast_to(p, tuple(sizes[v] for v in clique)).copy()
           for p, clique in zip(psi, maxcliques)]
    seps = [np.ones(tuple(sizes[v] for v in sep)) for sep in separators]
    node_vars = [list(c) for c in maxcliques] + [list(s) for s in separators]
    beliefs = engine(tree, psi + seps, node_vars)
    return marginalize(factors, factor_to_maxclique, maxcliques, beliefs[:len(maxcliques)])
