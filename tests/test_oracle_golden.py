"""Pin the CPU oracle (oracle/jt_oracle.py) against outputs of the unmodified reference
(tests/golden/*.npz, written by oracle/gen_golden.py) and against the known answers the
reference's own tests hard-code.  CPU only."""
import numpy as np
import pytest

import jt_oracle as oracle
from junctiontree_amd import synthetic
from conftest import as_tree

RTOL = 1e-10


def close(a, b, rtol=RTOL, atol=1e-13):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                               rtol=rtol, atol=atol)


def test_tree_cases_all_engines(golden):
    g = golden("tree_cases.npz")
    assert len(g.meta["cases"]) == 11
    for case in g.meta["cases"]:
        tree = as_tree(case["tree"])
        pots = g.arrs(case["potentials"])
        ref = g.arrs(case["ref_beliefs"])
        bf = g.arrs(case["bruteforce"])
        for engine in (oracle.beliefs_exact, oracle.beliefs_refshaped):
            out = engine(tree, pots, case["variables"])
            assert len(out) == len(ref)
            for o, r, t in zip(out, ref, bf):
                assert np.shape(o) == np.shape(r), case["name"]
                close(o, r)
                close(o, t)
        for o, t in zip(oracle.beliefs_bruteforce(tree, pots, case["variables"]), bf):
            close(o, t)


def test_refshaped_issues_5n_minus_1_einsums(golden):
    g = golden("tree_cases.npz")
    for case in g.meta["cases"]:
        tree = as_tree(case["tree"])
        n = len(oracle.flatten_tree(tree)[0])
        counters = {}
        oracle.beliefs_refshaped(tree, g.arrs(case["potentials"]), case["variables"], counters)
        assert counters["einsum_calls"] == 5 * n - 1


def test_networks_propagate_and_known_answers(golden):
    g = golden("networks.npz")
    import junctiontree_amd.construction as cons
    for name, net in g.meta["networks"].items():
        values = g.arrs(net["values"])
        assert net["ref_agrees_with_bruteforce"]
        # oracle propagate over a junction tree built by the build's own constructor
        maxcliques, f2m = cons.triangulate(net["factors"], net["sizes"])
        tree, seps = cons.construct_junction_tree(maxcliques, net["sizes"])
        out = oracle.propagate(tree, seps, maxcliques, f2m, net["factors"], net["sizes"], values)
        for o, r, t in zip(out, g.arrs(net["ref_propagate"]), g.arrs(net["bruteforce"])):
            close(o, r)
            close(o, t)
        for engine in (oracle.beliefs_refshaped,):
            out2 = oracle.propagate(tree, seps, maxcliques, f2m, net["factors"], net["sizes"],
                                    values, engine=engine)
            for o, r in zip(out2, g.arrs(net["ref_propagate"])):
                close(o, r)

    # hard-coded marginals of the reference's tests (tests/test_junctiontree.py:245-342,:483-525)
    net = g.meta["networks"]["abcdefgh"]
    out = g.arrs(net["ref_propagate"])
    k = net["known"]
    close(out[0], k["P_A"], rtol=1e-7)
    close(out[1].sum(axis=0), k["P_B"], rtol=1e-7)
    close(out[2].sum(axis=0), k["P_C"], rtol=1e-7)
    close(out[3].sum(axis=0), k["P_D"], rtol=1e-7)
    close(out[4].sum(axis=0), k["P_E"], rtol=1e-7)
    close(out[5].sum(axis=0), k["P_G"], rtol=1e-7)
    np.testing.assert_allclose(out[6].sum(axis=(0, 1)), k["P_F_atol0.01"], atol=0.01)
    np.testing.assert_allclose(out[7].sum(axis=(0, 1)), k["P_H_atol0.01"], atol=0.01)
    net = g.meta["networks"]["abcdef"]
    out = g.arrs(net["ref_propagate"])
    k = net["known"]
    close(out[2].sum(axis=1), k["P_C"], rtol=1e-7)
    close(out[1].sum(axis=0), k["P_A"], rtol=1e-7)
    close(out[1].sum(axis=1), k["P_B"], rtol=1e-7)
    close(out[3].sum(axis=0), k["P_D"], rtol=1e-7)
    close(out[4].sum(axis=0), k["P_E"], rtol=1e-7)
    np.testing.assert_allclose(out[5].sum(axis=(0, 1)), k["P_F_atol0.001"], atol=0.001)


def test_hand_built_tree_and_evaluate(golden):
    g = golden("networks.npz")
    net = g.meta["networks"]["abcdefgh"]
    values = g.arrs(net["values"])
    nodes = net["hand_nodes"]
    psi = oracle.evaluate(net["factors"], net["hand_factor_to_maxclique"], nodes[:6], values)
    for o, r in zip(psi, g.arrs(net["hand_evaluate"])):
        assert o.shape == r.shape
        close(o, r)
    close(psi[3], net["known"]["phi_ACE"], rtol=1e-7)
    out = oracle.propagate(as_tree(net["hand_tree"]), nodes[6:], nodes[:6],
                           net["hand_factor_to_maxclique"], net["factors"], net["sizes"], values)
    # the reference's own propagate() on this hand-built tree is right or silently wrong
    # depending on PYTHONHASHSEED (clique ADE keeps length-1 axes); the
    # fixture records which it was when generated.  Truth is brute force.
    for o, t in zip(out, g.arrs(net["bruteforce"])):
        close(o, t)
    if net["hand_propagate_agrees_with_bruteforce"]:
        for o, r in zip(out, g.arrs(net["hand_propagate"])):
            close(o, r)


def test_sprinkler_conditioned(golden):
    g = golden("networks.npz")
    net = g.meta["networks"]["sprinkler"]
    import junctiontree_amd.construction as cons
    for key, known in (("cond_wet", "P_sprinkler_given_wet"),
                       ("cond_wet_rain", "P_sprinkler_given_wet_rain")):
        cond = net[key]
        values = g.arrs(cond["values"])
        maxcliques, f2m = cons.triangulate(net["factors"], net["sizes"])
        tree, seps = cons.construct_junction_tree(maxcliques, net["sizes"])
        out = oracle.propagate(tree, seps, maxcliques, f2m, net["factors"], cond["sizes"], values)
        # the factor table holds exact zeros, which trips the reference's divide-out
        #  for some outputs; `ref_agrees` records where it was right
        for o, r, t, ok in zip(out, g.arrs(cond["ref_propagate"]), g.arrs(cond["bruteforce"]),
                               cond["ref_agrees"]):
            assert o.shape == r.shape
            close(o, t)
            if ok:
                close(o, r)
        marg = out[1].sum(axis=0)
        np.testing.assert_allclose(marg / marg.sum(), net["known"][known], atol=0.01)


def test_evaluate_cases(golden):
    g = golden("evaluate.npz")
    for case in g.meta["cases"]:
        ys = oracle.evaluate(case["factors"], case["f2m"], case["maxcliques"],
                             g.arrs(case["values"]))
        for y, r in zip(ys, g.arrs(case["ref_evaluate"])):
            assert y.shape == r.shape
            close(y, r)


def test_refsafe_synthetic_trees(golden):
    g = golden("refsafe.npz")
    recipes = {"chain_tree": synthetic.chain_tree, "wide_binary_tree": synthetic.wide_binary_tree,
               "random_tree": synthetic.random_tree}
    for case in g.meta["cases"]:
        spec = recipes[case["recipe"]](**case["kwargs"])
        pots = synthetic.potentials_for(spec, seed=case["seed"])
        ref = g.arrs(case["ref_beliefs"])
        for engine in (oracle.beliefs_exact, oracle.beliefs_refshaped):
            out = engine(spec["tree"], pots, spec["node_vars"])
            for o, r in zip(out, ref):
                close(o, r)
        beliefs, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
        for b in beliefs[:spec["n_cliques"]]:
            assert abs(b.sum() - z) <= 1e-12 * abs(z)


def test_divergent_cases_follow_bruteforce_not_reference(golden):
    g = golden("divergent.npz")
    import junctiontree_amd.construction as cons
    for case in g.meta["cases"]:
        if "tree" in case:
            out = oracle.beliefs_exact(as_tree(case["tree"]), g.arrs(case["potentials"]),
                                       case["variables"])
            for o, t in zip(out, g.arrs(case["truth"])):
                close(o, t)
        else:
            maxcliques, f2m = cons.triangulate(case["factors"], case["sizes"])
            tree, seps = cons.construct_junction_tree(maxcliques, case["sizes"])
            out = oracle.propagate(tree, seps, maxcliques, f2m, case["factors"], case["sizes"],
                                   g.arrs(case["values"]))
            for o, t in zip(out, g.arrs(case["truth"])):
                close(o, t, rtol=1e-9)


def test_synthetic_generator_is_stable():
    v = synthetic.synth_values(1, 0, (4,), 1.0)
    assert v.dtype == np.float64 and np.all(v >= 0.5) and np.all(v < 1.5)
    np.testing.assert_array_equal(v, synthetic.synth_values(1, 0, (2, 2), 1.0).ravel())
    assert not np.array_equal(v, synthetic.synth_values(1, 1, (4,), 1.0))
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    ab = synthetic.algorithmic_bytes(spec, 4)
    assert ab["messages"] == 510
    assert abs(ab["total"] / 1e9 - 3.222) < 0.005         # 3.217 GB cliques + separators
    spec = synthetic.chain_tree(n_cliques=1000, card=64, width=3)
    ab = synthetic.algorithmic_bytes(spec, 8)
    assert ab["messages"] == 1998 and abs(ab["total"] / 1e9 - 6.453) < 0.005  # 6.29 GB cliques + separators
