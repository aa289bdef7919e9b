"""Belief propagation on a junction tree (reference: `junctiontree/computation.py`).

`compute_beliefs(tree, potentials, clique_vars)` has the reference's signature and return
value (`computation.py:37-246`): a list indexed like `potentials` holding, for every clique,
psi * (all incoming messages) and, for every separator, up * down - unnormalised, each
summing to Z.  The whole two-pass traversal runs on the GPU (one plan, ~2 x depth kernel
launches); this module only prepares the inputs and fetches the results.

Differences from the reference, all deliberate:
  * downward messages are true all-but-one products - no divide-out, so separators whose
    variable order differs from the clique's, exact zeros and broadcast axes are handled
    correctly (the reference is silently wrong or raises there);
  * no recursion, so chains of any depth work;
  * potentials with length-1 axes are broadcast to the variable's full cardinality (the
    largest length any node gives that variable).
There is no CPU fallback: without libjtprop.so and a GPU this raises.
"""

import numpy as np

from .sum_product import HipSumProduct, SumProduct

__all__ = ["compute_beliefs", "apply_evidence", "sum_product"]

# module singleton, as `computation.py:9`
sum_product = HipSumProduct()


def apply_evidence(potentials, variables, evidence):
    """Shrink potentials to the observed states (reference: `computation.py:11-34`).

    Every observed axis is sliced to length 1 at the observed state; unobserved axes are
    kept.  Like the reference, each result is wrapped in a one-element list and scalars
    pass through unchanged."""
    out = []
    for pot, labels in zip(potentials, variables):
        if np.isscalar(pot):
            out.append([pot])
            continue
        index = tuple(slice(evidence[lab], evidence[lab] + 1) if lab in evidence else slice(None)
                      for lab in labels)
        out.append([pot[index]])
    return out


def _infer_sizes(nodes, potentials, clique_vars):
    sizes = {}
    for n in nodes:
        arr = potentials[n]
        shape = np.shape(arr)
        labels = clique_vars[n]
        if len(shape) != len(labels):
            raise ValueError("potential %d has shape %r but %d variables" % (n, shape, len(labels)))
        for length, lab in zip(shape, labels):
            have = sizes.get(lab, 1)
            if length != 1 and have != 1 and length != have:
                raise ValueError("operands could not be broadcast together: variable %r has "
                                 "lengths %d and %d" % (lab, have, length))
            sizes[lab] = max(have, length)
    return sizes


def compute_beliefs(tree, potentials, clique_vars, dl=sum_product):
    """Consistent beliefs for every node of the junction tree (cliques and separators)."""
    from . import engine

    # The reference's own call form - `compute_beliefs(tree, potentials, clique_vars, SumProduct(numpy.einsum))`,
    # `tests/test_computation.py:46-48` - names exactly the law the device implements, so it is accepted and runs on
    # the GPU like the default; any OTHER callable could be a different semiring and is refused (no host path here).
    # (`optimize=...`, the one switch the reference's source mentions - `SumProduct(np.einsum, optimize=True)`, commented out at
    #  `computation.py:4-9` - chooses numpy's contraction order, not the law: accepted and ignored)
    plain_numpy = (type(dl).__name__ == "SumProduct" and getattr(dl, "func", None) is np.einsum
                   and not getattr(dl, "args", ()) and set(getattr(dl, "kwargs", {})) <= {"optimize"})
    if not isinstance(dl, HipSumProduct) and not plain_numpy:
        raise TypeError(
            "unsupported reference feature: an injected distributive law (`compute_beliefs(..., dl=SumProduct(<einsum-compatible "
            "callable>, ...))`, junctiontree/computation.py:37 with sum_product.py:14-19).  This build runs ONE law - real sum-product, "
            "the law numpy.einsum implements - on the GPU and has no host path: pass junctiontree_amd.computation.sum_product or "
            "SumProduct(numpy.einsum[, optimize=...]) (got %r).  A custom einsum callable can still be wrapped in SumProduct for "
            "its own use." % (dl,))
    order, parent, parent_sep, _ = engine.flatten_tree(tree)
    seps = [parent_sep[c] for c in order if parent[c] != -1]
    sizes = _infer_sizes(order, potentials, clique_vars)
    # separators never contribute values (they are overwritten before use, Appendix A.1) but
    # their labels must be known
    for s in seps:
        for lab in clique_vars[s]:
            if lab not in sizes:
                raise ValueError("separator %d: variable %r is in no clique" % (s, lab))
    all_f32 = all(isinstance(potentials[c], np.ndarray) and potentials[c].dtype == np.float32
                  for c in order)
    plan = engine.plan_for(tree, clique_vars, sizes, "f32" if all_f32 else "f64")
    for c in order:
        plan.set_potential(c, potentials[c])
    plan.propagate()
    beliefs = list(potentials)
    for n in list(order) + seps:
        beliefs[n] = plan.belief(n)
    return beliefs
