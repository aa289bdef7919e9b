"""Distributive-law objects (reference: `junctiontree/sum_product.py`).

`SumProduct(einsum_fn)` keeps the reference's seam (`sum_product.py:10-43`): arbitrary
hashable labels are renumbered and the call is forwarded to the injected einsum-compatible
callable.  `HipSumProduct` is the law this package actually ships: its `einsum` runs one
multiply-and-marginalise on the MI355X (a star-shaped plan whose root carries the union
scope), and `compute_beliefs` recognises it and sends the *whole tree* to the device in one
go instead of one host round trip per einsum.
"""

import numpy as np

__all__ = ["SumProduct", "HipSumProduct"]


class SumProduct:
    """Sum-product distributive law over an injected einsum callable."""

    def __init__(self, einsum, *args, **kwargs):
        self.func = einsum
        self.args = args
        self.kwargs = kwargs

    def einsum(self, *args, **kwargs):
        """`einsum(op0, labels0, op1, labels1, ..., out_labels)` with arbitrary labels.

        Labels are numbered by first appearance (the reference numbers them through a
        `set`, `sum_product.py:34`, which makes the summation order hash dependent).
        The leaf form `einsum(scalar, [])` is passed through as in the reference
        (`computation.py:77`)."""
        items = list(args)
        if len(items) % 2 == 0:
            if any(len(labels) for labels in items[1::2]):
                raise KeyError("explicit output labels are required")    # as the reference
            return self.func(*items, *self.args, **kwargs, **self.kwargs)
        number = {}
        for labels in items[1::2] + [items[-1]]:
            for lab in labels:
                number.setdefault(lab, len(number))
        call = []
        for op, labels in zip(items[0:-1:2], items[1:-1:2]):
            call += [op, [number[lab] for lab in labels]]
        call.append([number[lab] for lab in items[-1]])
        return self.func(*call, *self.args, **kwargs, **self.kwargs)


def hip_einsum(*args):
    """numpy.einsum-compatible (interleaved, explicit output) multiply-and-marginalise on
    the GPU.  Operands may broadcast along length-1 axes like numpy's."""
    from . import engine

    items = list(args)
    if len(items) % 2 == 0:
        items.append([])
    ops = [np.asarray(a, dtype=np.float64) if np.asarray(a).dtype != np.float32 else np.asarray(a)
           for a in items[0:-1:2]]
    subs = [list(s) for s in items[1:-1:2]]
    out = list(items[-1])
    sizes = {}
    for arr, labels in zip(ops, subs):
        if arr.ndim != len(labels):
            raise ValueError("operand has %d axes but %d labels" % (arr.ndim, len(labels)))
        for n, lab in zip(arr.shape, labels):
            if sizes.get(lab, 1) not in (1, n) and n != 1:
                raise ValueError("operands could not be broadcast together: label %r" % (lab,))
            sizes[lab] = max(sizes.get(lab, 1), n)
    for lab in out:
        if lab not in sizes:
            raise ValueError("output label %r is not in any operand" % (lab,))
    union = []
    for labels in subs:
        for lab in labels:
            if lab not in union:
                union.append(lab)
    k = len(ops)
    # node list: 0 = root over the union scope (all-ones), 1..k = operands, k+1..2k = separators
    node_vars = [union] + subs + subs
    tree = [0] + [(k + 1 + i, [1 + i]) for i in range(k)]
    dtype = "f32" if all(a.dtype == np.float32 for a in ops) else "f64"
    plan = engine.plan_for(tree, node_vars, sizes, dtype)
    plan.set_potential(0, np.ones((1,) * len(union)))
    for i, arr in enumerate(ops):
        plan.set_potential(1 + i, arr)
    plan.propagate()
    return plan.marginal(0, out)


class HipSumProduct(SumProduct):
    """The sum-product law executed by libjtprop on the MI355X."""

    def __init__(self):
        super().__init__(hip_einsum)
