"""Host-side functions of the drop-in API that need neither GPU nor libjtprop compute: checked against
outputs captured from the unmodified reference (tests/golden/apply_evidence.npz, oracle/gen_golden.py)."""
import numpy as np
import pytest

from junctiontree_amd import computation as comp


def test_apply_evidence_matches_reference_outputs(golden):
    """`apply_evidence` (reference: junctiontree/computation.py:11-34).  The reference's own test
    (tests/test_computation.py:377-408) forgets to assert; the captured outputs pin shape (observed axes
    kept with length 1), values, the one-element-list wrapping and the scalar pass-through."""
    g = golden("apply_evidence.npz")
    for case in g.meta["cases"]:
        pots = g.arrs(case["potentials"])
        for i in case.get("scalar_inputs", []):
            pots[i] = float(pots[i])                    # (a Python float was handed to the reference)
        evidence = {k: v for k, v in case["evidence"]}
        keep = [np.copy(p) for p in pots]
        out = comp.apply_evidence(pots, case["variables"], evidence)
        assert len(out) == len(pots)
        for o, r, kind, p, k in zip(out, g.arrs(case["ref"]), case["ref_types"], pots, keep):
            assert isinstance(o, list) and len(o) == 1, case["name"]
            assert np.shape(o[0]) == r.shape, case["name"]
            np.testing.assert_array_equal(np.asarray(o[0]), r, err_msg=case["name"])
            assert (kind == "ndarray") == isinstance(o[0], np.ndarray), case["name"]
            np.testing.assert_array_equal(np.asarray(p), k)          # inputs untouched


def test_apply_evidence_case_of_the_reference_test(golden):
    """The equalities the reference's test states (without asserting them), with the axes kept."""
    g = golden("apply_evidence.npz")
    case = g.meta["cases"][0]
    pots = g.arrs(case["potentials"])
    out = comp.apply_evidence(pots, case["variables"], {3: 0, 9: 2})
    np.testing.assert_array_equal(out[0][0][0], pots[0][0, :, :])
    np.testing.assert_array_equal(out[1][0][:, 0], pots[1][:, 2])
    np.testing.assert_array_equal(out[2][0][0], pots[2][0, :])
    np.testing.assert_array_equal(out[3][0], pots[3])
    np.testing.assert_array_equal(out[4][0], pots[4][0:1])
    np.testing.assert_array_equal(out[6][0], pots[6])


def test_elimination_order_gives_valid_junction_trees():
    """`construction.triangulate(..., order=...)` (round 4): every factor lies in its clique, the tree has the running
    intersection property, and the column-by-column order on a lattice gives the chain of width-(h+1) cliques of SURVEY.md 8d."""
    from junctiontree_amd import construction as cons, synthetic
    from junctiontree_amd.engine import flatten_tree
    factors, sizes, _ = synthetic.lattice_mrf(4, 9, 2)
    for order in (None, synthetic.lattice_column_order(4, 9), list(range(36))[::-1], [7, 3]):
        cliques, f2c = cons.triangulate(factors, sizes, order=order)
        for f, c in zip(factors, f2c):
            assert set(f) <= set(cliques[c])
        tree, seps = cons.construct_junction_tree(cliques, sizes)
        order_, parent, parent_sep, _ = flatten_tree(tree)
        assert sorted(order_) == list(range(len(cliques)))
        for v in sizes:                                   # the cliques holding v form a connected subtree
            holders = {c for c in range(len(cliques)) if v in cliques[c]}
            tops = [c for c in holders if parent[c] not in holders]
            assert len(tops) == 1, (v, tops)
        for c in order_[1:]:
            assert set(seps[parent_sep[c] - len(cliques)]) == set(cliques[c]) & set(cliques[parent[c]])
    cliques, _ = cons.triangulate(factors, sizes, order=synthetic.lattice_column_order(4, 9))
    assert len(cliques) == 36 - 4 and max(len(c) for c in cliques) == 5


def test_partial_elimination_orders_continue_with_min_fill():
    """`triangulate(order=...)` with an order that names only some variables: those are eliminated first, greedy min-fill takes over
    from there (the costs are built once when the given order runs out, not on every forced step)."""
    from junctiontree_amd import construction as cons
    rng = np.random.default_rng(3)
    names = list(range(14))
    factors = [[int(v) for v in rng.choice(14, size=int(rng.integers(2, 4)), replace=False)] for _ in range(20)]
    sizes = {v: 2 for v in names}
    used = sorted({v for f in factors for v in f})
    full, _ = cons.triangulate(factors, sizes)
    for k in (0, 1, 5, len(used)):
        partial = [int(v) for v in rng.permutation(used)[:k]]
        cliques, f2c = cons.triangulate(factors, sizes, order=partial)
        for f, c in zip(factors, f2c):
            assert set(f) <= set(cliques[c])
        # the given prefix is honoured: eliminating it by hand and handing the whole order over gives the same cliques
        if k == 0:
            assert cliques == full
    with pytest.raises(ValueError):
        cons.triangulate(factors, sizes, order=[used[0], used[0]])
