"""User interface of the junction-tree library, MI355X build.

Same public names and data model as the reference's `junctiontree/junctiontree.py`:
`create_junction_tree(factors, sizes)` (:12-16) returns a `JunctionTree` whose
`propagate(values)` (:297-331) turns factor values into consistent, unnormalised factor
marginals.  Structure classes are plain Python; all numeric work of `propagate` after the
factor product runs on the GPU through `engine.Plan`:

    values --H2D (factor tables only)--> clique potentials formed on the device (evaluate,
    junctiontree.py:203-226) --> collect + distribute (computation.py:37-246 in one plan) -->
    per-factor marginals on the device (junctiontree.py:264-274) --D2H--> list shaped like `values`.
"""

from dataclasses import dataclass, field
from typing import Any

import numpy as np

from . import construction as cons

__all__ = ["create_junction_tree", "argfind1", "take", "is_subset", "einsum",
           "FactorGraph", "CliqueGraph", "JunctionTree"]


def create_junction_tree(factors, sizes, order=None):
    """Create a junction tree for a factor graph (reference: `junctiontree.py:12-16`).  `order` (not in the reference):
    an elimination order for the triangulation instead of greedy min-fill (`construction.triangulate`)."""
    assert all(type(f) == list for f in factors), "Provided factor is not a list"
    return FactorGraph(factors=factors, sizes=sizes).triangulate(order=order).create_junction_tree()


def argfind1(xs, cond):
    """Index of the first element of xs satisfying cond (`junctiontree.py:19-21`)."""
    for i, x in enumerate(xs):
        if cond(x):
            return i
    raise StopIteration


def take(xs, inds):
    """Pick several list elements (`junctiontree.py:24-26`)."""
    return [xs[i] for i in inds]


def is_subset(a, b):
    """Whether every element of a is in b (`junctiontree.py:29-31`)."""
    return set(a) <= set(b)


def einsum(xs, xs_keys, y_keys):
    """Product of arrays onto `y_keys` with arbitrary labels; labels that appear only in
    the output become length-1 axes (reference helper `junctiontree.py:34-80`).  Host-side
    (numpy): it builds clique potentials from factor tables, outside the message-passing
    path."""
    xs = [np.asarray(x) for x in xs]
    xs_keys = [list(k) for k in xs_keys]
    have = set(k for keys in xs_keys for k in keys)
    fresh = [k for k in y_keys if k not in have]
    if fresh:
        xs[0] = xs[0].reshape((1,) * len(fresh) + xs[0].shape)
        xs_keys[0] = fresh + xs_keys[0]
    number = {}
    for keys in xs_keys + [list(y_keys)]:
        for k in keys:
            number.setdefault(k, len(number))
    call = []
    for x, keys in zip(xs, xs_keys):
        call += [x, [number[k] for k in keys]]
    call.append([number[k] for k in y_keys])
    return np.einsum(*call)


def _stage_changed_cliques(plan, ct, xs, changed=None):
    """`evaluate` on the device for the cliques whose member factors differ from what `plan` holds: one call into the
    library for all of them (`engine.Plan.stage_factors`).  Returns the number of cliques formed (`plan.staged_cliques`
    keeps the count of the last call for tests and tools).  What is staged = which factors, over which variables in which
    axis order, with which values: two JunctionTree objects whose junction trees coincide share one cached plan
    (`engine.plan_for` keys on the tree, not on the factors), and the same bytes under a transposed label list are
    another table - `stage_factors` compares all of that, unless the caller names the changed factors (`changed`)."""
    return plan.stage_factors(ct.factor_graph.factors, ct.factor_to_maxclique, xs, changed=changed)


@dataclass(frozen=True)
class FactorGraph:
    """Factors (lists of variables) and the size of every variable (`junctiontree.py:83-117`)."""

    factors: Any
    sizes: Any

    def triangulate(self, order=None):
        """Triangulate and collect the maximal cliques (`junctiontree.py:102-117`)."""
        maxcliques, factor_to_maxclique = cons.triangulate(self.factors, self.sizes, order=order)
        return CliqueGraph(maxcliques=maxcliques, factor_to_maxclique=factor_to_maxclique,
                           factor_graph=self)


@dataclass
class CliqueGraph:
    """Maximal cliques of a triangulated factor graph (`junctiontree.py:120-274`)."""

    maxcliques: Any
    factor_to_maxclique: Any
    factor_graph: Any

    def create_junction_tree(self):
        """`junctiontree.py:138-200`: node list = maxcliques ++ separators, tree of indices."""
        tree, separators = cons.construct_junction_tree(self.maxcliques, self.factor_graph.sizes)
        return JunctionTree(tree=tree, separators=separators, clique_tree=self)

    def _members(self):
        members = [[] for _ in self.maxcliques]
        for fi, mc in enumerate(self.factor_to_maxclique):
            members[mc].append(fi)
        return members

    def evaluate(self, xs):
        """Clique values from factor values (`junctiontree.py:203-226`): the product of the
        factors assigned to each clique in the clique's axis order; variables no assigned
        factor covers stay length-1 axes."""
        out = []
        for clique, members in zip(self.maxcliques, self._members()):
            if not members:
                out.append(np.ones((1,) * len(clique)))
                continue
            out.append(einsum(take(xs, members), take(self.factor_graph.factors, members), clique))
        return out

    def marginalize(self, ys):
        """Factor results from clique results (`junctiontree.py:229-274`) for arrays already
        on the host: sum the clique axes that are not in the factor."""
        return [einsum([ys[mc]], [self.maxcliques[mc]], list(fvars))
                for fvars, mc in zip(self.factor_graph.factors, self.factor_to_maxclique)]


@dataclass(frozen=True)
class JunctionTree:
    """Junction tree of a factor graph (`junctiontree.py:277-331`).

    `tree` = [clique, (separator, subtree), ...] over the node list
    `clique_tree.maxcliques + separators`."""

    tree: Any
    separators: Any
    clique_tree: Any
    _opts: dict = field(default_factory=dict, compare=False, repr=False)
    _memo: dict = field(default_factory=dict, compare=False, repr=False)

    # what a tree remembers between calls (a weak reference to its device plan, derived tables) is not part of its value:
    # a tree pickles and copies like the reference's, before and after it has been used
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_memo"] = {}
        return state

    def __setstate__(self, state):
        for k, v in state.items():
            object.__setattr__(self, k, v)

    def cover(self, trusted=False):
        """Per clique, the variables its potential depends on: the union of the variables of the factors assigned to it
        (`junctiontree.py:203-226` - evaluate leaves every other variable of the clique a length-1 axis, `:52-61`).  The device
        plan keeps no full-size table for a clique that is mostly such axes (`engine.Plan(cover=...)`).  `trusted`: the caller
        vouches that the factor lists are what they were at the last call (`propagate(xs, changed=...)`): the remembered
        cover is returned without looking at them."""
        from .engine import _same_lists as engine_same_lists
        ct = self.clique_tree
        hit = self._memo.get("cover")
        if trusted and hit is not None:
            return hit[1]
        # (compared by value against list copies - list.__eq__ runs in C; hit[0] is a version number `plan` keys on)
        if hit is not None and hit[3] == list(ct.factor_to_maxclique) and engine_same_lists(hit[2], ct.factor_graph.factors):
            return hit[1]
        cover = [[] for _ in ct.maxcliques]
        scalar = [False] * len(ct.maxcliques)
        for fvars, mc in zip(ct.factor_graph.factors, ct.factor_to_maxclique):
            scalar[mc] = scalar[mc] or len(fvars) == 0
            for v in fvars:
                if v not in cover[mc]:
                    cover[mc].append(v)
        for mc, clique in enumerate(ct.maxcliques):
            if scalar[mc]:                                   # (a factor without variables is a value no axis carries: the clique keeps its table)
                cover[mc] = list(clique)
        self._memo["cover"] = ((hit[0] + 1) if hit is not None else 0, cover, [list(f) for f in ct.factor_graph.factors], list(ct.factor_to_maxclique))
        return cover

    def plan(self, dtype="f64", trusted=False, fold=True):
        """The device plan for the current variable sizes (sizes are read at call time, as
        `junctiontree.py:311` does: the reference's tests condition on evidence by setting
        a size to 1, `tests/test_junctiontree.py:393-411`)."""
        import weakref
        from . import engine

        # the plan cache's key names the whole structure (1-2 ms to build for a thousand cliques): a tree remembers the key
        # and a weak reference to the plan it was last given for (dtype, the sizes as they are NOW, its options, and the
        # factor structure its cover was computed from - by value: a recomputed cover may reuse the old list's id)
        sizes = self.clique_tree.factor_graph.sizes
        memo = "plan" if fold else "plan_nofold"
        hit = self._memo.get(memo)
        if trusted and hit is not None and hit[0][0] == dtype:      # (`propagate(xs, changed=...)`: sizes, options and factors are vouched for)
            plan = engine.cached_plan(hit[1], hit[2]())
            if plan is not None:
                return plan
        cover = self.cover(trusted=trusted)
        mark = (dtype, tuple(sizes.items()), tuple(sorted(self._opts.items())), self._memo["cover"][0])
        if hit is not None and hit[0] == mark:
            plan = engine.cached_plan(hit[1], hit[2]())
            if plan is not None:
                return plan
        node_vars = [list(c) for c in self.clique_tree.maxcliques] + [list(s) for s in self.separators]
        # (`fold`: propagate returns factor marginals only, junctiontree.py:327-331 - the plan is told which, so that those of cliques
        #  without a table are formed inside the propagate's launch; fold=False: the plan `compute_beliefs` would make of this tree)
        ct = self.clique_tree
        extra = {"fold": (tuple(ct.factor_to_maxclique), tuple(map(tuple, ct.factor_graph.factors)))} if fold else {}
        plan, key = engine.plan_for(self.tree, node_vars, sizes, dtype, return_key=True, cover=cover, **extra, **self._opts)
        self._memo[memo] = (mark, key, weakref.ref(plan))
        return plan

    def propagate(self, xs, changed=None):
        """Belief propagation: factor values in, unnormalised factor marginals out (same
        list length and array shapes as `xs`; float64).

        `changed` (not in the reference; it answers the FIXME at `junctiontree.py:206-214`): the indices of the factors whose
        tables differ from the previous call on this tree, or "all".  By default every table is compared with what the device
        holds (one vectorised pass over all of them, so that arrays updated in place are seen); a caller that knows what it
        changed skips that - and vouches that the factor structure and every other table are what they were."""
        ct = self.clique_tree
        trusted = changed is not None and "plan" in self._memo
        if trusted and "all_f32" in self._memo:          # (the caller vouches for the structure - shapes and dtypes with it)
            all_f32 = self._memo["all_f32"]
        else:
            all_f32 = self._memo["all_f32"] = all(type(x) is np.ndarray and x.dtype == np.float32 for x in xs)
        plan = self.plan("f32" if all_f32 else "f64", trusted=trusted)
        # evaluate (junctiontree.py:203-226) on the device: only factor tables cross PCIe, and only those of
        # cliques whose factors changed since this plan last saw them (the reference recomputes every clique on
        # every call and says so in a FIXME, junctiontree.py:206-214)
        _stage_changed_cliques(plan, ct, xs, changed=changed)
        # (no wait here: the marginal kernels are enqueued behind the propagate while it runs, and `jtp_get_marginals` waits once for all)
        plan.propagate(sync=False)
        # marginalize (junctiontree.py:229-274) on the device: one launch for all factors, the factors of one clique
        # sharing the passes over its belief table
        return plan.factor_marginals(ct.factor_graph.factors, ct.factor_to_maxclique, trusted=trusted)

    def propagate_evidence_sets(self, xs, evidence_sets):
        """`propagate` for several hard-evidence sets over the same factor values (no counterpart in the
        reference, whose users loop over `propagate` after slicing the factors, `README.md:155-165`):
        `evidence_sets` is a list of {variable: observed state}; returns one list of factor marginals
        per set, each factor with its full shape (entries contradicting the evidence are zero) and
        every table of set e summing to P(evidence e) * Z.  The clique tables are formed once and
        shared by all sets; a pass over a table serves eight sets at a time (JTP_MULTISET)."""
        from . import engine

        ct = self.clique_tree
        if not evidence_sets:
            return []
        all_f32 = all(isinstance(x, np.ndarray) and x.dtype == np.float32 for x in xs)
        node_vars = [list(c) for c in ct.maxcliques] + [list(s) for s in self.separators]
        # one copy of the tables; eight evidence sets per pass over a table (JTP_MULTISET), marginals formed
        # on demand from the tables and each set's final messages
        from ._capi import UnsupportedStructure
        try:
            plan = engine.plan_for(self.tree, node_vars, ct.factor_graph.sizes, "f32" if all_f32 else "f64",
                                   n_batch=len(evidence_sets), multiset=True, **self._opts)
            plan.evidence_mode = "multiset: eight evidence sets per pass over a table"
        except UnsupportedStructure as exc:
            # separators too large for the per-set LDS regions of a multi-set pass (e.g. 64 x 64 doubles): the sets
            # still share one copy of the tables but run one pass each, one HIP stream each - up to eight times the table traffic
            # of a multi-set plan, so it is said, not done silently (`plan.evidence_mode`, and a warning once per tree)
            import warnings
            plan = engine.plan_for(self.tree, node_vars, ct.factor_graph.sizes, "f32" if all_f32 else "f64",
                                   n_batch=len(evidence_sets), share_potentials=True, cover=self.cover(), **self._opts)
            plan.evidence_mode = "one pass per evidence set over shared tables (the multi-set plan was refused: %s)" % exc
            if not self._memo.get("warned_multiset"):
                self._memo["warned_multiset"] = True
                warnings.warn("junctiontree_amd: the evidence sets of this tree run one pass each instead of eight per pass (%s)" % exc,
                              RuntimeWarning, stacklevel=2)
        self._memo["evidence_plan"] = plan
        _stage_changed_cliques(plan, ct, xs)
        for b, observed in enumerate(evidence_sets):
            plan.set_evidence(observed, batch=b)
        plan.propagate(0, len(evidence_sets))
        return [plan.factor_marginals(ct.factor_graph.factors, ct.factor_to_maxclique, batch=b) for b in range(len(evidence_sets))]
