"""Parity of the HIP path with the oracle and the reference's golden vectors, through the
C ABI (libjtprop.so), on a real MI355X.  Tolerances: float64 storage 1e-11 relative to the
oracle (summation-order noise only); float32 storage 1e-6 relative (north-star tolerance),
measured per array against its max magnitude and elementwise on entries above 1e-30*max."""
import numpy as np
import pytest

import jt_oracle as oracle
import junctiontree_amd as jt
from conftest import as_tree
from junctiontree_amd import computation as comp
from junctiontree_amd import engine, synthetic
from junctiontree_amd.sum_product import SumProduct

pytestmark = pytest.mark.gpu

RTOL64, RTOL32 = 1e-11, 1e-6


def close(got, want, rtol=RTOL64, what=""):
    """Tables of non-negative entries (every synthetic and network table; sums of products of
    non-negative terms lose no precision to cancellation): ELEMENTWISE relative error on every entry above
    1e-30 * max, absolute 1e-30 * max below.  Signed data (the reference's randn tree
    cases): relative to the array's max magnitude - an entry there is a difference of large terms."""
    got = np.asarray(got, dtype=np.float64)
    want = np.broadcast_to(np.asarray(want, dtype=np.float64), got.shape)
    scale = np.max(np.abs(want)) if want.size else 0.0
    if want.size and np.all(want >= 0):
        np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-30 * scale + 1e-300, err_msg=what)
    else:
        np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * scale + 1e-300, err_msg=what)


@pytest.fixture(autouse=True)
def _no_cached_plans():
    """Explicit plans of a test must run in the launch mode they ask for (default: dataflow launches in blockIdx
    order, the mode the benchmark times), whatever `compute_beliefs` / `propagate` of an earlier test left in the
    plan cache.  (Since round 3 an idle cached plan no longer forces ticket order - only a propagate IN FLIGHT does,
    `test_ticket_order_only_while_another_plan_is_in_flight` - the empty cache keeps the tests independent.)"""
    engine.clear_plan_cache()
    yield
    engine.clear_plan_cache()


def expected_mode(opts):
    if opts.get("level_launches") or opts.get("split_variants"):
        return "level"
    return "flow_tickets" if opts.get("flow_tickets") or opts.get("n_batch", 1) > 1 else "flow"


def run_plan(spec, pots, dtype, **opts):
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, **opts)
    for c in range(spec["n_cliques"]):
        plan.set_potential(c, pots[c])
    plan.propagate()
    st = plan.stats()
    # the launch mode is observable (jtp_stats): the default is dataflow launches in blockIdx order
    assert st["launch_mode"] == expected_mode(opts), (st["launch_mode"], opts)
    assert st["tickets_used"] == (1 if expected_mode(opts) == "flow_tickets" else 0) and st["flow_fallbacks"] == 0
    out = [plan.belief(n) for n in range(len(spec["node_vars"]))]
    z = plan.z()
    plan.close()
    return out, z


# ------------------------------------------------------------------ reference fixtures

def test_reference_tree_cases(golden):
    g = golden("tree_cases.npz")
    for case in g.meta["cases"]:
        out = comp.compute_beliefs(as_tree(case["tree"]), g.arrs(case["potentials"]), case["variables"])
        for o, r, t in zip(out, g.arrs(case["ref_beliefs"]), g.arrs(case["bruteforce"])):
            assert np.shape(o) == np.shape(r), case["name"]
            close(o, r, what=case["name"])
            close(o, t, what=case["name"])


def test_reference_networks_propagate(golden):
    g = golden("networks.npz")
    for name, net in g.meta["networks"].items():
        tree = jt.create_junction_tree(net["factors"], dict(net["sizes"]))
        values = g.arrs(net["values"])
        out = tree.propagate(values)
        assert len(out) == len(values)
        for o, v, r in zip(out, values, g.arrs(net["ref_propagate"])):
            assert o.shape == v.shape and o.dtype == np.float64
            close(o, r, what=name)
    net = g.meta["networks"]["abcdefgh"]
    out = jt.create_junction_tree(net["factors"], dict(net["sizes"])).propagate(g.arrs(net["values"]))
    k = net["known"]                      # tests/test_junctiontree.py:245-292
    np.testing.assert_allclose(out[0], k["P_A"])
    np.testing.assert_allclose(out[1].sum(axis=0), k["P_B"])
    np.testing.assert_allclose(out[3].sum(axis=0), k["P_D"])
    np.testing.assert_allclose(out[5].sum(axis=0), k["P_G"])
    np.testing.assert_allclose(out[6].sum(axis=(0, 1)), k["P_F_atol0.01"], atol=0.01)
    np.testing.assert_allclose(out[7].sum(axis=(0, 1)), k["P_H_atol0.01"], atol=0.01)


def test_evidence_sets_through_the_public_api(golden):
    """JunctionTree.propagate_evidence_sets: each set must equal a plain propagate() of the same network
    with a one-hot indicator multiplied into one factor that contains the observed variable."""
    g = golden("networks.npz")
    rng = np.random.default_rng(5)
    for name, net in g.meta["networks"].items():
        sizes = dict(net["sizes"])
        tree = jt.create_junction_tree(net["factors"], sizes)
        values = g.arrs(net["values"])
        variables = sorted(sizes)
        evidence_sets = [{}]
        for k in (1, 2, 3):
            picked = rng.choice(len(variables), size=min(k, len(variables)), replace=False)
            evidence_sets.append({variables[i]: int(rng.integers(0, sizes[variables[i]])) for i in picked})
        got = tree.propagate_evidence_sets(values, evidence_sets)
        assert len(got) == len(evidence_sets)
        for observed, out in zip(evidence_sets, got):
            xs = [np.array(v, dtype=np.float64) for v in values]
            for var, state in observed.items():
                f = next(i for i, fv in enumerate(net["factors"]) if var in fv)
                ind = np.zeros(sizes[var])
                ind[state] = 1.0
                shape = [1] * xs[f].ndim
                shape[net["factors"][f].index(var)] = sizes[var]
                xs[f] = xs[f] * ind.reshape(shape)
            want = tree.propagate(xs)
            for o, w, v in zip(out, want, values):
                assert o.shape == np.shape(v)
                close(o, w, what="%s %r" % (name, observed))


def test_evidence_sets_over_a_chain_of_64_by_64_separators_run_as_a_multi_set_plan():
    """A chain whose separators are 64 x 64 doubles (32 KiB each, beyond the 16 KiB a multi-set pass gives one evidence set's
    sub-boxes IF a workgroup had to hold a whole separator): the planner cuts the cliques so that a workgroup touches a slice of
    each - the plan is a multi-set plan (`plan.evidence_mode`), not the one-pass-per-set fall-back, and every set agrees with the
    oracle's `propagate` on indicator-multiplied factors.  (What IS refused: cliques of a row or two whose separators are nearly
    the clique itself - DESIGN.md section 7.)"""
    import warnings
    rng = np.random.default_rng(11)
    n, k = 6, 64
    names = ["x%d" % i for i in range(n + 2)]
    sizes = {v: k for v in names}
    factors = [[names[i], names[i + 1], names[i + 2]] for i in range(n)]
    values = [rng.uniform(0.5, 1.5, (k, k, k)) / k for _ in factors]
    tree = jt.create_junction_tree(factors, sizes)
    assert max(int(np.prod([sizes[v] for v in sp])) for sp in tree.separators) == k * k
    evidence_sets = [{}, {names[0]: 3}, {names[3]: 60, names[7]: 1}, {names[2]: 0, names[4]: 63, names[5]: 17}, {names[1]: 5}]
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # (the fall-back announces itself with a RuntimeWarning)
        got = tree.propagate_evidence_sets(values, evidence_sets)
    plan = tree._memo["evidence_plan"]
    assert plan.evidence_mode.startswith("multiset"), plan.evidence_mode
    assert plan.stats()["flow_fallbacks"] == 0
    for observed, out in zip(evidence_sets, got):
        xs = [v.copy() for v in values]
        for var, state in observed.items():
            f = next(i for i, fv in enumerate(factors) if var in fv)
            ind = np.zeros(k)
            ind[state] = 1.0
            shape = [1, 1, 1]
            shape[factors[f].index(var)] = k
            xs[f] = xs[f] * ind.reshape(shape)
        want = _joint_marginals_chain(factors, sizes, xs)
        for o, w in zip(out, want):
            close(o, w, what="%r" % (observed,))
    engine.clear_plan_cache()


def _joint_marginals_chain(factors, sizes, xs):
    """Factor marginals of a chain of overlapping triples by exact forward / backward sums (numpy, float64): the oracle of
    `test_evidence_sets_over_a_chain_of_64_by_64_separators...` (the brute-force joint of eight 64-state variables is out of reach)."""
    n = len(factors)
    fwd = [None] * n          # fwd[i][b, c]: everything left of factor i summed out, over its first two variables
    fwd[0] = np.ones(xs[0].shape[:2])
    for i in range(1, n):
        fwd[i] = np.einsum("ab,abc->bc", fwd[i - 1], xs[i - 1])
    bwd = [None] * n          # bwd[i][b, c]: everything right of factor i, over its last two variables
    bwd[n - 1] = np.ones(xs[n - 1].shape[1:])
    for i in range(n - 2, -1, -1):
        bwd[i] = np.einsum("bcd,cd->bc", xs[i + 1], bwd[i + 1])
    return [fwd[i][:, :, None] * xs[i] * bwd[i][None, :, :] for i in range(n)]


def test_hand_built_tree_like_reference_test(golden):
    g = golden("networks.npz")
    net = g.meta["networks"]["abcdefgh"]
    nodes = net["hand_nodes"]
    hand = jt.JunctionTree(as_tree(net["hand_tree"]), nodes[6:],
                           jt.CliqueGraph(maxcliques=nodes[:6],
                                          factor_to_maxclique=net["hand_factor_to_maxclique"],
                                          factor_graph=jt.FactorGraph(factors=net["factors"],
                                                                      sizes=net["sizes"])))
    values = g.arrs(net["values"])
    for y, r in zip(hand.clique_tree.evaluate(values), g.arrs(net["hand_evaluate"])):
        assert y.shape == r.shape
        close(y, r)
    for o, t in zip(hand.propagate(values), g.arrs(net["bruteforce"])):
        close(o, t)


def test_sprinkler_conditioned_by_mutating_sizes(golden):
    g = golden("networks.npz")
    net = g.meta["networks"]["sprinkler"]
    tree = jt.create_junction_tree(net["factors"], dict(net["sizes"]))
    for key, known in (("cond_wet", "P_sprinkler_given_wet"),
                       ("cond_wet_rain", "P_sprinkler_given_wet_rain")):
        cond = net[key]
        for var, size in cond["sizes"].items():               # tests/test_junctiontree.py:394, 408
            tree.clique_tree.factor_graph.sizes[var] = size
        out = tree.propagate(g.arrs(cond["values"]))
        for o, r, t, ok in zip(out, g.arrs(cond["ref_propagate"]), g.arrs(cond["bruteforce"]),
                               cond["ref_agrees"]):
            assert o.shape == r.shape
            close(o, t, what=key)
            if ok:
                close(o, r, what=key)
        marg = out[1].sum(axis=0)
        np.testing.assert_allclose(marg / marg.sum(), net["known"][known], atol=0.01)


def test_divergent_cases_follow_bruteforce(golden):
    g = golden("divergent.npz")
    for case in g.meta["cases"]:
        if "tree" in case:
            out = comp.compute_beliefs(as_tree(case["tree"]), g.arrs(case["potentials"]),
                                       case["variables"])
        else:
            out = jt.create_junction_tree(case["factors"], case["sizes"]).propagate(
                g.arrs(case["values"]))
        for o, t in zip(out, g.arrs(case["truth"])):
            close(o, t, rtol=1e-10, what=case["name"])


def test_refsafe_synthetic_trees_f64_and_f32(golden):
    g = golden("refsafe.npz")
    recipes = {"chain_tree": synthetic.chain_tree, "wide_binary_tree": synthetic.wide_binary_tree,
               "random_tree": synthetic.random_tree}
    for case in g.meta["cases"]:
        spec = recipes[case["recipe"]](**case["kwargs"])
        ref = g.arrs(case["ref_beliefs"])
        pots = synthetic.potentials_for(spec, seed=case["seed"])
        out = comp.compute_beliefs(spec["tree"], pots, spec["node_vars"])
        for o, r in zip(out, ref):
            close(o, r, what=case["name"])
        pots32 = [p.astype(np.float32) for p in pots]
        want = oracle.beliefs_exact(spec["tree"], pots32, spec["node_vars"])
        out32 = comp.compute_beliefs(spec["tree"], pots32, spec["node_vars"])
        for o, w in zip(out32, want):
            close(o, w, rtol=RTOL32, what=case["name"] + " f32")


# ------------------------------------------------------------------ wider synthetic coverage

@pytest.mark.parametrize("opts", [
    {}, {"block_log2": 10}, {"block_log2": 11, "lds_budget": 256}, {"layout_policy": 1},
    {"layout_policy": 1, "block_log2": 10, "lds_budget": 128}, {"block_log2": 12, "lds_budget": 2048},
    {"split_variants": True}, {"split_variants": True, "block_log2": 10}, {"keep_root": True},
    {"level_launches": True}, {"level_launches": True, "block_log2": 10, "lds_budget": 256},
    {"layout_policy": 2}, {"layout_policy": 3}, {"layout_policy": 2, "block_log2": 10, "level_launches": True},
    {"layout_policy": 4, "block_log2": 10}, {"layout_policy": 4, "lds_budget": 2048}, {"layout_policy": 4, "lds_budget": 128, "block_log2": 11, "level_launches": True},
])
def test_planner_options_do_not_change_results(opts):
    specs = [
        synthetic.chain_tree(n_cliques=5, card=4, width=3),
        synthetic.chain_tree(n_cliques=4, card=16, width=3),
        synthetic.chain_tree(n_cliques=4, card=3, width=3),
        synthetic.wide_binary_tree(n_cliques=7, width=12, sep=6, card=2, seed=1),
        synthetic.wide_binary_tree(n_cliques=6, width=14, sep=7, card=2, seed=2),
        synthetic.wide_binary_tree(n_cliques=7, width=5, sep=2, card=3, seed=4),
        synthetic.random_tree(n_cliques=9, width=11, sep=5, card=2, seed=3),
        synthetic.random_tree(n_cliques=8, width=4, sep=2, card=5, seed=6),
    ]
    for i, spec in enumerate(specs):
        pots = synthetic.potentials_for(spec, seed=21)
        want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
        out, zz = run_plan(spec, pots, "f64", **opts)
        for o, w in zip(out, want):
            close(o, w, what="spec %d %r" % (i, opts))
        assert abs(zz - z) <= 1e-11 * abs(z)
        out32, _ = run_plan(spec, [p.astype(np.float32) for p in pots[:spec["n_cliques"]]], "f32", **opts)
        want32 = oracle.beliefs_exact(spec["tree"], [p.astype(np.float32) for p in pots], spec["node_vars"])
        for o, w in zip(out32, want32):
            close(o, w, rtol=RTOL32, what="spec %d f32 %r" % (i, opts))


@pytest.mark.parametrize("level_launches", [False, True])
def test_reduce_tasks_on_device(monkeypatch, level_launches):
    """Reduce tasks (sum of a message's partial copies, formed once behind the producer) in dataflow
    and in per-level launches; a low threshold makes small trees use them."""
    monkeypatch.setenv("JTP_REDUCE_MIN", "2")
    for spec in (synthetic.wide_binary_tree(n_cliques=15, width=14, sep=7, card=2, seed=2),
                 synthetic.random_tree(n_cliques=9, width=13, sep=5, card=2, seed=3),
                 synthetic.chain_tree(n_cliques=6, card=16, width=3)):
        pots = synthetic.potentials_for(spec, seed=5)
        want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", block_log2=10,
                           level_launches=level_launches, layout_policy=3)     # (the layout with the most partial copies)
        assert sum(t["kind"] for t in plan.describe()["tasks"]) > 0
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, pots[c])
        for _ in range(3):
            plan.propagate()
        for node in range(len(spec["node_vars"])):
            close(plan.belief(node), want[node], what="node %d" % node)
        assert abs(plan.z() - z) <= 1e-11 * abs(z)
        plan.close()


def test_mid_size_wide_tree_vs_oracle():
    spec = synthetic.wide_binary_tree(n_cliques=31, width=16, sep=8, card=2, seed=0)
    pots = synthetic.potentials_for(spec, seed=2, dtype=np.float32)
    want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
    out, _ = run_plan(spec, pots, "f32")
    for o, w in zip(out, want):
        close(o, w, rtol=RTOL32)
    pots64 = synthetic.potentials_for(spec, seed=2)
    want = oracle.beliefs_refshaped(spec["tree"], pots64, spec["node_vars"])
    out, _ = run_plan(spec, pots64, "f64")
    for o, w in zip(out, want):
        close(o, w)


def test_chain_card64_fp64_vs_oracle():
    """C2 at reduced length (same clique shape: 64^3 doubles, separators 64^2)."""
    spec = synthetic.chain_tree(n_cliques=12, card=64, width=3)
    pots = synthetic.potentials_for(spec, seed=3)
    want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
    out, zz = run_plan(spec, pots, "f64")
    for o, w in zip(out, want):
        close(o, w)
    assert abs(zz - z) <= 1e-11 * abs(z)


def test_many_children_star():
    from test_planner_emulated import star
    for n_children, card in ((4, 2), (7, 3), (13, 2)):
        tree, pots, node_vars, sizes = star(n_children, card=card, seed=n_children)
        want = oracle.beliefs_exact(tree, pots, node_vars)
        out = comp.compute_beliefs(tree, pots, node_vars)
        for o, w in zip(out, want):
            close(o, w)


def test_device_synthetic_fill_matches_numpy_generator():
    spec = synthetic.wide_binary_tree(n_cliques=7, width=12, sep=6, card=2, seed=1)
    for dtype, npdt in (("f32", np.float32), ("f64", np.float64)):
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype)
        plan.fill_synthetic(5, spec["scales"])
        plan.propagate()
        pots = synthetic.potentials_for(spec, seed=5, dtype=npdt)
        want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
        for n in range(len(spec["node_vars"])):
            close(plan.belief(n), want[n], rtol=RTOL32 if dtype == "f32" else RTOL64)
        plan.close()


# ------------------------------------------------------------------ API behaviour

def test_sum_product_einsum_on_device_and_evidence_shrinking():
    """tests/test_computation.py:411-459 with the device law."""
    rng = np.random.default_rng(0)
    A = rng.random((3, 4, 2))
    a = [0, 0, 1]
    A_upd = comp.sum_product.einsum(A, [0, 1, 2], a, [0], [0, 1, 2])
    close(A_upd, A * np.array(a)[:, None, None])
    A_es = A_upd[2, :, :]
    B = rng.random((4, 2))
    close(comp.sum_product.einsum(A_upd, [0, 1, 2], B, [1, 2], [1, 2]),
          comp.sum_product.einsum(A_es, [1, 2], B, [1, 2], [1, 2]))
    Cv = rng.random(3)
    C_upd = comp.sum_product.einsum(Cv, [0], a, [0], [0])
    close(comp.sum_product.einsum(A_upd, [0, 1, 2], C_upd, [0], []),
          comp.sum_product.einsum(A_es, [1, 2], C_upd[2], [], []))
    close(comp.sum_product.einsum(A, ["x", "y", "z"], ["y", "x"]), A.sum(axis=2).T)


def test_the_reference_call_form_with_numpy_einsum_runs_on_the_device():
    """`compute_beliefs(tree, potentials, clique_vars, SumProduct(numpy.einsum))` - how the reference's own tests call it
    (tests/test_computation.py:46-48) - names the law the device implements: same result as the default."""
    rng = np.random.default_rng(5)
    tree = [0, (3, [1]), (4, [2])]
    node_vars = [["a", "b"], ["b", "c"], ["a", "d"], ["b"], ["a"]]
    pots = [rng.random((2, 3)), rng.random((3, 4)), rng.random((2, 5)), np.ones(3), np.ones(2)]
    got = comp.compute_beliefs(tree, pots, node_vars, SumProduct(np.einsum))
    want = oracle.beliefs_exact(tree, pots, node_vars)
    for g, w in zip(got, want):
        close(g, w, rtol=RTOL64)
    # the one switch the reference's source mentions (`SumProduct(np.einsum, optimize=True)`, computation.py:4-9): numpy's
    # contraction order, not another law - accepted; a dtype or an out argument would change what is computed - refused
    for g, w in zip(comp.compute_beliefs(tree, pots, node_vars, SumProduct(np.einsum, optimize=True)), want):
        close(g, w, rtol=RTOL64)
    with pytest.raises(TypeError):
        comp.compute_beliefs(tree, pots, node_vars, SumProduct(np.einsum, dtype=np.float32))


@pytest.mark.parametrize("nv,card,dt", [(7, 3, np.float64), (5, 5, np.float64), (6, 5, np.float32), (8, 3, np.float64), (6, 6, np.float32),
                                        (8, 3, np.float32), (9, 3, np.float32), (6, 7, np.float32), (7, 4, np.float32)])
def test_factor_products_formed_on_the_device_in_mixed_radix_tables(nv, card, dt):
    """`tree.propagate` forms a clique's potential on the device from its factor tables (jtp_set_potential_product,
    `CliqueGraph.evaluate`): a clique of several ROWS whose thread part is stored at true cardinalities, through the public API
    against the brute-force joint.  (Round 3: the kernel took a thread-part variable's digit of the element index instead of
    the place inside the row - wrong from the second row on, 12x off; the explicit-plan tests upload finished tables and
    never saw it.)"""
    rng = np.random.default_rng(0)
    names = list("abcdefghi")[:nv]
    sizes = {v: card for v in names}
    factors = [names, names[:2], names[-3:]]
    values = [rng.uniform(0.2, 1.0, [sizes[v] for v in f]).astype(dt) for f in factors]
    tree = jt.create_junction_tree(factors, sizes)
    got = tree.propagate(values)           # (the widest factor is as wide as its clique: a marginal onto all its variables -
                                           #  four-row workgroups where the sub-box would not fit LDS otherwise, plan_loops)
    d = tree.plan("f32" if dt == np.float32 else "f64").describe()
    assert max(p["phys_elems"] // p["trow"] for p in d["pnodes"]) > 1 and d["tmix"] == (0 if card in (4, 7) else 1)
    ax = {v: i for i, v in enumerate(names)}
    ops = []
    for f, val in zip(factors, values):
        ops += [np.asarray(val, dtype=np.float64), [ax[v] for v in f]]
    joint = np.einsum(*ops, list(range(nv)))
    for f, g in zip(factors, got):
        close(g, np.einsum(joint, list(range(nv)), [ax[v] for v in f]), rtol=RTOL32 if dt == np.float32 else RTOL64, what=str(f))


def test_plans_created_and_run_from_two_host_threads():
    """a plan is not thread safe, distinct plans are independent - created, run and destroyed from two host
    threads at once (ctypes releases the GIL): each thread's results equal the oracle's every time."""
    import threading
    specs = [synthetic.wide_binary_tree(n_cliques=15, width=14, sep=7, card=2, seed=21), synthetic.random_tree(n_cliques=12, width=6, sep=3, card=3, seed=22)]
    errors = []

    def worker(spec, dtype):
        try:
            pots = synthetic.potentials_for(spec, seed=5, dtype=np.float32 if dtype == "f32" else np.float64)
            want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
            for rep in range(6):
                plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, lds_budget=(96 * 1024 if rep % 2 else 0))
                for c in range(spec["n_cliques"]):
                    plan.set_potential(c, pots[c])
                for _ in range(20):
                    plan.propagate(sync=False)
                plan.sync()
                for node in (0, spec["n_cliques"] - 1, len(spec["node_vars"]) - 1):
                    close(plan.belief(node), want[node], rtol=RTOL32 if dtype == "f32" else RTOL64)
                assert abs(plan.z() - z) <= (1e-6 if dtype == "f32" else 1e-11) * abs(z) and plan.stats()["flow_fallbacks"] == 0
                plan.close()
        except Exception as exc:            # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(sp, dt)) for sp, dt in zip(specs, ("f32", "f64"))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_float32_trees_the_planner_refuses_run_in_float64_storage():
    """A clique of few rows with four or more neighbours whose separators are nearly the whole clique cannot be planned with
    1024-element (float32) rows - every message needs the whole thread part in LDS (found by tools/gpu_fuzz.py, FUZZ_BIG, seed
    92488).  Round 4: `jtp_plan_create` itself makes such a plan with float64 tables (round 3: only the Python layer behind
    `compute_beliefs` did, keyed on the message text; a C caller got JTP_EUNSUPPORTED) and says so - `jtp_stats.storage_dtype`,
    a RuntimeWarning from `engine.Plan`, `plan_cache_info()["widened"]`."""
    from test_planner_emulated import random_junction_tree
    rng = np.random.default_rng(92488)
    while True:
        spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 12)), max_width=9, cards=(2, 3, 3, 4, 5, 6, 7))
        if 1 << 14 <= max(p.size for p in pots) <= 1 << 22 and sum(p.size for p in pots) <= 1 << 24:
            break
    cast = [p.astype(np.float32) for p in pots]
    from junctiontree_amd import _capi
    with pytest.warns(RuntimeWarning, match="float64 tables made"):
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
    assert plan.requested_dtype == _capi.JTP_F32 and plan.dtype == _capi.JTP_F64
    plan.close()
    want = oracle.beliefs_exact(spec["tree"], cast, spec["node_vars"])
    engine.clear_plan_cache()
    with pytest.warns(RuntimeWarning):
        got = comp.compute_beliefs(spec["tree"], cast, spec["node_vars"])
    assert engine.plan_cache_info()["widened"] == 1
    for g, w in zip(got, want):
        close(g, w, rtol=RTOL64)


def test_more_than_32_variables_on_a_node_when_the_rest_have_one_state():
    """tests/test_planner_emulated.py, same case, through the device: beliefs, marginals over labels that include one-state
    variables, evidence on a one-state variable."""
    from test_planner_emulated import wide_node_case
    tree, node_vars, sizes, pots = wide_node_case()
    want = oracle.beliefs_exact(tree, pots, node_vars)
    got = comp.compute_beliefs(tree, pots, node_vars)
    for n in range(5):
        assert got[n].shape == want[n].shape
        close(got[n], want[n])
    plan = engine.Plan(tree, node_vars, sizes)
    for c in range(3):
        plan.set_potential(c, pots[c])
    plan.set_evidence({"u3": 0})
    plan.propagate()
    m = plan.marginal(0, ["u1", "b", "u25", "a"])
    assert m.shape == (1, 3, 1, 2)
    close(m, want[0].sum(axis=node_vars[0].index("c")).reshape(-1).reshape(2, 3).T.reshape(1, 3, 1, 2))
    ms = plan.marginals([(1, ["d", "u0"]), (0, ["u29"])])
    assert ms[0].shape == (4, 1) and ms[1].shape == (1,)
    close(ms[1], np.array([want[0].sum()]))
    with pytest.raises(ValueError):
        plan.set_evidence({"u3": 1})
    plan.close()


def test_errors():
    with pytest.raises(ValueError):
        comp.compute_beliefs([0, (2, [1])], [np.ones((2, 3)), np.ones((4, 2)), np.ones(3)],
                             [[1, 2], [2, 3], [2]])                 # 3 vs 4 along variable 2
    with pytest.raises(TypeError):
        comp.compute_beliefs([0], [np.ones(2)], [[1]], dl=object())
    with pytest.raises(TypeError):              # another callable could be another semiring: refused, there is no host path
        comp.compute_beliefs([0], [np.ones(2)], [[1]], dl=SumProduct(lambda *a, **k: np.einsum(*a, **k)))
    plan = engine.Plan([0], [[1, 2]], {1: 2, 2: 3})
    with pytest.raises(ValueError):
        plan.set_potential(0, np.ones((2, 2)))
    plan.close()


def test_inputs_are_not_mutated_and_results_are_fresh():
    pots = [np.arange(6.0).reshape(2, 3) + 1, np.arange(12.0).reshape(3, 4) + 1, np.ones(3)]
    keep = [p.copy() for p in pots]
    out = comp.compute_beliefs([0, (2, [1])], pots, [[3, 5], [5, 9], [5]])
    for p, k in zip(pots, keep):
        np.testing.assert_array_equal(p, k)
    assert all(o.dtype == np.float64 for o in out)


# ------------------------------------------------------------------ full-size properties

def test_full_size_c4_properties():
    """BASELINE.json config 4 (256 cliques of 2^20 float32): size-independent properties.
    Every belief sums to Z; adjacent cliques agree on their separator marginal and it equals
    up*down (calibration); scaling one potential by 2 doubles everything (linearity)."""
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
    plan.fill_synthetic(1, spec["scales"])
    plan.propagate()
    z = plan.z()
    assert np.isfinite(z) and z > 0
    n = spec["n_cliques"]
    rng = np.random.default_rng(0)
    for c in [0, 1, 2] + [int(i) for i in rng.choice(np.arange(3, n), size=20, replace=False)]:
        assert abs(plan.marginal(c, []) - z) <= 2e-6 * z, c
        if c == 0:
            continue
        par = spec["parent"][c]
        sep_node = n + c - 1
        labels = spec["node_vars"][sep_node]
        sb = plan.belief(sep_node)
        close(plan.marginal(c, labels), sb, rtol=2e-6)
        close(plan.marginal(par, labels), sb, rtol=2e-6)
    # sampled cliques and the separators next to them, elementwise against the oracle on the same values (the device
    # fill and synthetic.potentials_for generate the same numbers): root, an inner clique of every level, leaves
    pots = synthetic.potentials_for(spec, seed=1, dtype=np.float32)
    want, z_want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
    del pots
    assert abs(z - z_want) <= RTOL32 * z_want
    for c in [0, 1, 2, 5, 12, 27, 60, 100, 127, 128, 200, n - 1] + [int(i) for i in rng.choice(np.arange(3, n), size=6, replace=False)]:
        close(plan.belief(c), want[c], rtol=RTOL32, what="clique %d" % c)
        if c > 0:
            close(plan.belief(n + c - 1), want[n + c - 1], rtol=RTOL32, what="separator of clique %d" % c)
    del want
    assert plan.stats()["launch_mode"] == "flow" and plan.stats()["flow_fallbacks"] == 0
    # linearity
    leaf = n - 1
    before = plan.belief(leaf)
    plan2 = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
    scales = list(spec["scales"])
    scales[5] *= 2.0
    plan2.fill_synthetic(1, scales)
    plan2.propagate()
    close(plan2.belief(leaf), 2.0 * before, rtol=1e-6)
    assert abs(plan2.z() - 2 * z) <= 2e-6 * z
    plan.close()
    plan2.close()


def test_rccl_binding_single_rank_selftest():
    """One-rank RCCL communicator: unique id, init, grouped send/recv to self, destroy."""
    import ctypes as C
    from junctiontree_amd import _capi
    lib = _capi.lib()
    buf = C.create_string_buffer(128)
    _capi.check(lib.jtp_comm_unique_id(buf))
    _capi.check(lib.jtp_comm_init(0, 1, buf, 0))
    try:
        _capi.check(lib.jtp_comm_selftest(1024))
        spec = synthetic.wide_binary_tree(n_cliques=7, width=12, sep=6, card=2, seed=1)
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", n_ranks=1, rank=0)
        plan.fill_synthetic(3, spec["scales"])
        plan.propagate()
        assert np.isfinite(plan.z())
        plan.close()
    finally:
        _capi.check(lib.jtp_comm_destroy())


def test_device_evaluate_matches_reference_cases(golden):
    """CliqueGraph.evaluate on the device (jtp_set_potential_product): the reference's evaluate
    cases (tests/test_junctiontree.py:38-109), one single-clique plan per maximal clique (its
    belief equals its potential), incl. variables no factor covers (constant axes)."""
    g = golden("evaluate.npz")
    for case in g.meta["cases"]:
        values = g.arrs(case["values"])
        for ci, (clique, want) in enumerate(zip(case["maxcliques"], g.arrs(case["ref_evaluate"]))):
            members = [fi for fi, mc in enumerate(case["f2m"]) if mc == ci]
            plan = engine.Plan([0], [clique], case["sizes"], dtype="f64")
            plan.set_potential_product(0, [values[fi] for fi in members], [case["factors"][fi] for fi in members])
            plan.propagate()
            got = plan.belief(0)
            close(got, np.broadcast_to(want, got.shape))
            plan.close()


def test_device_evaluate_many_factors_and_mixed_dtypes():
    rng = np.random.default_rng(5)
    clique = list("abcdef")
    sizes = {"a": 2, "b": 3, "c": 4, "d": 2, "e": 5, "f": 2}
    var_lists = [["a"], ["b", "a"], ["c"], ["d", "c"], ["e"], ["f", "e"], ["a", "f"], ["b"], ["c", "e"],
                 ["d"], ["e", "a", "b"], ["f"]]                       # 12 factors: two kernel passes
    arrays = [rng.uniform(0.5, 1.5, [sizes[v] for v in vs]) for vs in var_lists]
    arrays[3] = arrays[3].astype(np.float32)
    arrays[6] = arrays[6][:1, :]                                      # broadcast along "a"
    ops = []
    for a, vs in zip(arrays, var_lists):
        ops += [a.astype(np.float64), vs]
    want = oracle.labelled_einsum(*ops, clique)
    plan = engine.Plan([0], [clique], sizes, dtype="f64")
    plan.set_potential_product(0, arrays, var_lists)
    plan.propagate()
    close(plan.belief(0), want, rtol=1e-7)                            # one factor is float32
    plan.set_potential_product(0, [], [])
    plan.propagate()
    close(plan.belief(0), np.ones([sizes[v] for v in clique]))
    with pytest.raises(ValueError):
        plan.set_potential_product(0, [np.ones(3)], [["z"]] if False else [["a"]])   # wrong length along a
    plan.close()


def test_batched_evidence_sets_on_streams():
    """BASELINE config 5 in miniature: several evidence sets share one plan (structure, task
    tables), each with its own potentials and HIP stream; evidence enters as a one-hot indicator
    multiplied into one clique (tests/test_computation.py:411-459 shows the equivalence)."""
    spec = synthetic.wide_binary_tree(n_cliques=15, width=12, sep=6, card=2, seed=2)
    n, nb = spec["n_cliques"], 6
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", n_batch=nb)
    base = synthetic.potentials_for(spec, seed=4)
    all_pots = []
    for b in range(nb):
        rng = np.random.default_rng(1000 + b)
        pots = [p.copy() for p in base]
        for var in rng.choice(len(spec["sizes"]), size=3, replace=False):
            state = int(rng.integers(0, 2))
            host = next(c for c in range(n) if var in spec["node_vars"][c])
            axis = spec["node_vars"][host].index(var)
            ind = np.zeros(2)
            ind[state] = 1.0
            shape = [1] * pots[host].ndim
            shape[axis] = 2
            pots[host] = pots[host] * ind.reshape(shape)
        all_pots.append(pots)
        for c in range(n):
            plan.set_potential(c, pots[c], batch=b)
    plan.propagate(0, nb)
    for b in range(nb):
        want, z = oracle.beliefs_exact(spec["tree"], all_pots[b], spec["node_vars"], return_z=True)
        for node in range(len(spec["node_vars"])):
            close(plan.belief(node, batch=b), want[node], what="batch %d node %d" % (b, node))
        assert abs(plan.z(batch=b) - z) <= 1e-11 * abs(z)
    plan.close()


def test_repeated_propagates_with_new_potentials_and_mixed_launch_modes():
    """The dataflow launches (one per phase) find their inputs through "unwritten" markers in the
    message arena, whose two halves alternate between propagates: run many propagates on one plan,
    changing every potential in between and switching between dataflow and per-level launches
    (per-launch profiling uses the latter), and check every result - a stale or early read of any
    message entry would show up as a wrong belief."""
    spec = synthetic.wide_binary_tree(n_cliques=31, width=15, sep=7, card=2, seed=5)
    n = spec["n_cliques"]
    for dtype in ("f64", "f32"):
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, block_log2=11)
        for it in range(8):
            pots = synthetic.potentials_for(spec, seed=100 + it, dtype=np.float32 if dtype == "f32" else np.float64)
            for c in range(n):
                plan.set_potential(c, pots[c])
            plan.set_profiling(1 if it in (2, 3, 6) else 0, per_launch=it in (2, 3, 6))
            plan.propagate()
            if it % 3 == 1:
                plan.propagate()               # twice on the same inputs: same answer
            want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
            for node in range(len(spec["node_vars"])):
                close(plan.belief(node), want[node], rtol=RTOL32 if dtype == "f32" else 1e-11, what="%s it %d node %d" % (dtype, it, node))
            assert abs(plan.z() - z) <= (1e-5 if dtype == "f32" else 1e-11) * abs(z)
        plan.close()


def test_event_timing_of_every_nth_propagate():
    """jtp_set_profiling_stride: with a stride only every n-th propagate carries events; the reported device time is the
    mean over those and stays a plausible time of ONE propagate, the results are unaffected."""
    spec = synthetic.wide_binary_tree(n_cliques=15, width=14, sep=7, card=2, seed=2)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
    plan.fill_synthetic(1, spec["scales"])
    plan.propagate()
    z0 = plan.z()
    times = {}
    for stride in (1, 4):
        plan.set_profiling(12, stride=stride)
        for _ in range(12):
            plan.propagate(sync=False)
        plan.sync()
        st = plan.stats()
        times[stride] = sum(k["ms"] for k in st["kernels"].values())
        assert times[stride] > 0 and plan.z() == z0
    assert 0.4 < times[4] / times[1] < 2.5, times
    with pytest.raises(ValueError):
        plan.set_profiling(4, stride=0)
    plan.set_profiling(0)
    plan.close()


def test_pinned_host_arrays_in_and_out():
    """Potentials handed over from page-locked arrays (jtp_host_alloc) and beliefs read into them."""
    spec = synthetic.wide_binary_tree(n_cliques=7, width=12, sep=6, card=2, seed=3)
    pots = synthetic.potentials_for(spec, seed=8)
    want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64")
    bufs = []
    for c in range(spec["n_cliques"]):
        a = engine.pinned_empty(pots[c].shape, np.float64)
        a[...] = pots[c]
        plan.set_potential(c, a)
        bufs.append(a)
    plan.propagate()
    for c in range(spec["n_cliques"]):
        got = plan.belief(c, out=bufs[c])
        assert got is bufs[c]
        close(got, want[c])
    with pytest.raises(ValueError):
        plan.belief(0, out=np.empty((3,)))
    del bufs, got, a
    plan.close()


def test_batched_marginals_match_single_requests():
    """jtp_get_marginals (all of CliqueGraph.marginalize in one launch) against jtp_get_marginal and
    the oracle: several requests per clique, permuted axis orders, the empty request (= Z), a repeated
    call served from the plan's cache, and a second, different request list."""
    spec = synthetic.random_tree(n_cliques=9, width=6, sep=3, card=3, seed=12)
    pots = synthetic.potentials_for(spec, seed=3)
    want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64")
    for c in range(spec["n_cliques"]):
        plan.set_potential(c, pots[c])
    plan.propagate()
    rng = np.random.default_rng(0)
    requests = []
    for c in range(spec["n_cliques"]):
        labels = list(spec["node_vars"][c])
        for k in (0, 1, 2, len(labels)):
            requests.append((c, [labels[i] for i in rng.permutation(len(labels))[:k]]))
    for reqs in (requests, requests, requests[::3]):
        got = plan.marginals(reqs)
        assert len(got) == len(reqs)
        for (c, labels), g in zip(reqs, got):
            single = plan.marginal(c, labels)
            # (a request that shares the pass over its clique's belief table with two others - round 4 - is summed in another
            #  order than the same request on its own: equal to rounding, not bit for bit)
            assert g.shape == single.shape and np.allclose(g, single, rtol=1e-13, atol=0.0)
            axes = [spec["node_vars"][c].index(lab) for lab in labels]
            drop = tuple(a for a in range(len(spec["node_vars"][c])) if a not in axes)
            ref = np.transpose(want[c].sum(axis=drop), np.argsort(np.argsort(axes))) if labels else want[c].sum()
            close(g, ref, what="clique %d labels %r" % (c, labels))
    assert abs(plan.marginals([(0, [])])[0] - z) <= 1e-11 * abs(z)
    plan.close()


def test_launch_modes_are_bit_identical():
    """One launch per phase (dataflow), one per level, and ticket-ordered dataflow run the same
    workgroups on the same tables: every belief must agree to the last bit."""
    spec = synthetic.wide_binary_tree(n_cliques=31, width=15, sep=7, card=2, seed=9)
    pots = synthetic.potentials_for(spec, seed=77, dtype=np.float32)
    results, modes = [], []
    for opts in ({}, {"level_launches": True}, {"flow_tickets": True}):
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", block_log2=11, **opts)
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, pots[c])
        plan.propagate()
        plan.propagate()
        st = plan.stats()
        modes.append((st["launch_mode"], st["tickets_used"], st["flow_propagates"], st["n_launches"] > 2))
        results.append([plan.belief(node) for node in range(len(spec["node_vars"]))])
        plan.close()
    # three genuinely different modes (round 2 compared tickets with tickets: cached plans forced them)
    assert modes == [("flow", 0, 2, False), ("level", 0, 0, True), ("flow_tickets", 2, 2, False)], modes
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert np.array_equal(a, b)


def test_ticket_order_only_while_another_plan_is_in_flight():
    """Dataflow launches in blockIdx order are unsafe only while ANOTHER dataflow kernel may be resident (jtp_engine.hip:
    enter_flight).  An idle second plan costs nothing; one with an unsynchronised propagate makes the newcomer draw
    tickets; once it has been waited for, blockIdx order is back.  Results are identical either way."""
    spec = synthetic.wide_binary_tree(n_cliques=15, width=14, sep=7, card=2, seed=4)
    pots = synthetic.potentials_for(spec, seed=8)
    want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
    a = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64")
    b = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64")
    for plan in (a, b):
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, pots[c])
    a.propagate()                                            # waited for: a is idle again
    b.propagate()
    assert a.stats()["launch_mode"] == "flow" and b.stats()["launch_mode"] == "flow"
    a.propagate(sync=False)                                  # a in flight ...
    b.propagate(sync=False)                                  # ... so b draws tickets
    assert a.stats()["launch_mode"] == "flow" and b.stats()["launch_mode"] == "flow_tickets"
    a.propagate(sync=False)                                  # and now b is in flight when a launches again
    assert a.stats()["launch_mode"] == "flow_tickets"
    for plan in (a, b):
        for node in range(len(spec["node_vars"])):           # (the read-out waits for the plan's stream)
            close(plan.belief(node), want[node], what="node %d" % node)
    a.propagate()
    assert a.stats()["launch_mode"] == "flow" and a.stats()["tickets_used"] == 1 and b.stats()["tickets_used"] == 1
    a.close()
    b.close()


def test_two_trees_sharing_one_cached_plan_with_transposed_factor():
    """Two JunctionTree objects with the same junction tree share one cached plan (plan_for keys on the tree, not on
    the factors).  The second model labels one factor's axes the other way round and gets byte-identical arrays: its
    cliques must be staged again, not skipped (round-2 advisor finding: it silently returned the first model's
    marginals)."""
    sizes = {"a": 2, "b": 2, "c": 2, "d": 2}
    f1 = [["a", "b", "c"], ["a", "b"], ["c", "d"]]
    f2 = [["a", "b", "c"], ["b", "a"], ["c", "d"]]
    rng = np.random.default_rng(5)
    values = [rng.random((2, 2, 2)), rng.random((2, 2)), rng.random((2, 2))]
    t1, t2 = jt.create_junction_tree(f1, dict(sizes)), jt.create_junction_tree(f2, dict(sizes))
    assert t1.tree == t2.tree and t1.plan("f64") is t2.plan("f64")          # the premise: one plan
    for tree, factors in ((t1, f1), (t2, f2), (t1, f1)):
        out = tree.propagate(values)
        joint = np.einsum(values[0], [0, 1, 2], values[1], [{"a": 0, "b": 1}[v] for v in factors[1]], values[2], [2, 3], [0, 1, 2, 3])
        for o, labs in zip(out, factors):
            close(o, np.einsum(joint, [0, 1, 2, 3], [{"a": 0, "b": 1, "c": 2, "d": 3}[v] for v in labs]), what=repr(labs))


def test_dataflow_timeout_falls_back_to_level_launches():
    """Fault injection: every dataflow wait times out (flow_debug 8).  The host must notice at the
    next synchronisation - or at the next READ-OUT when the caller did not synchronise - run the propagate
    again with one launch per level, and keep doing so."""
    spec = synthetic.wide_binary_tree(n_cliques=15, width=13, sep=6, card=2, seed=8)
    pots = synthetic.potentials_for(spec, seed=31)
    want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
    for tickets, sync in ((False, True), (True, True), (False, False)):
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", flow_tickets=tickets)
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, pots[c])
        plan.propagate()
        assert plan.stats()["flow_fallbacks"] == 0 and plan.stats()["n_launches"] <= 2      # (both phases in one launch where the planner merges them)
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, 2.0 * pots[c])           # the aborted propagate must not leave ITS OLD results
        plan.debug_set("flow_debug", 8)
        plan.propagate(sync=sync)
        plan.debug_set("flow_debug", 0)
        scale = 2.0 ** spec["n_cliques"]
        for node in range(len(spec["node_vars"])):          # (sync=False: the read-out itself has to notice)
            close(plan.belief(node), scale * want[node], what="node %d" % node)
        assert plan.stats()["flow_fallbacks"] == 1 and plan.stats()["n_launches"] > 2
        plan.propagate()
        assert abs(plan.z() - scale * z) <= 1e-11 * abs(scale * z) and plan.stats()["flow_fallbacks"] == 1
        plan.close()


def test_fallback_keeps_the_messages_of_sets_already_checked():
    """A time-out in one evidence set must not wipe the separator messages of a set whose propagate had
    already been checked (round-1 defect: its separator beliefs then read as marker NaNs)."""
    spec = synthetic.wide_binary_tree(n_cliques=7, width=12, sep=6, card=2, seed=3)
    pots = synthetic.potentials_for(spec, seed=2)
    want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", n_batch=2)
    for b in range(2):
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, pots[c], batch=b)
    plan.propagate(0, 1)                                     # set 0: finished and checked
    plan.debug_set("flow_debug", 8)
    plan.propagate(1, 2)                                     # set 1: times out, is run again per level
    plan.debug_set("flow_debug", 0)
    assert plan.stats()["flow_fallbacks"] == 1
    for b in (1, 0):
        for node in range(len(spec["node_vars"])):
            close(plan.belief(node, batch=b), want[node], what="batch %d node %d" % (b, node))
    plan.close()


@pytest.mark.parametrize("share", [True, False, "multiset"])
def test_hard_evidence_sets(share):
    """jtp_set_evidence: evidence sets that differ only by what is observed, with ONE copy of the clique
    tables (JTP_SHARE_POTENTIALS, BASELINE config 5 in miniature) or with a copy each.  Expected values:
    the oracle on potentials with a one-hot indicator multiplied into one clique containing the variable
    (what apply_evidence amounts to, tests/test_computation.py:411-459 of the reference)."""
    for spec, dtype in ((synthetic.wide_binary_tree(n_cliques=15, width=12, sep=6, card=2, seed=2), "f64"),
                        (synthetic.random_tree(n_cliques=9, width=5, sep=2, card=3, seed=4), "f64"),
                        (synthetic.wide_binary_tree(n_cliques=7, width=13, sep=6, card=2, seed=6), "f32")):
        n, nb = spec["n_cliques"], 5
        np_dt = np.float32 if dtype == "f32" else np.float64
        base = synthetic.potentials_for(spec, seed=4, dtype=np_dt)
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, n_batch=nb, share_potentials=bool(share),
                           multiset=share == "multiset")
        for b in range(nb if not share else 1):
            for c in range(n):
                plan.set_potential(c, base[c], batch=b)
        if share:
            with pytest.raises(ValueError):
                plan.set_potential(0, base[0], batch=1)
        labels = sorted(spec["sizes"])
        observed = []
        for b in range(nb):
            rng = np.random.default_rng(1000 + b)
            obs = {} if b == 0 else {labels[i]: int(rng.integers(0, spec["sizes"][labels[i]]))
                                     for i in rng.choice(len(labels), size=min(4, b + 1), replace=False)}
            observed.append(obs)
            plan.set_evidence(obs, batch=b)
        for rep in range(2):
            plan.propagate(0, nb)
            for b in range(nb):
                pots = [np.asarray(p, dtype=np.float64).copy() for p in base]
                for var, state in observed[b].items():
                    host = next(c for c in range(n) if var in spec["node_vars"][c])
                    axis = spec["node_vars"][host].index(var)
                    ind = np.zeros(spec["sizes"][var])
                    ind[state] = 1.0
                    shape = [1] * pots[host].ndim
                    shape[axis] = spec["sizes"][var]
                    pots[host] = pots[host] * ind.reshape(shape)
                want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
                for node in range(len(spec["node_vars"])):
                    close(plan.belief(node, batch=b), want[node], rtol=RTOL32 if dtype == "f32" else RTOL64,
                          what="share %r batch %d node %d" % (share, b, node))
                assert abs(plan.z(batch=b) - z) <= (1e-6 if dtype == "f32" else 1e-11) * abs(z)
            observed[1], observed[2] = observed[2], observed[1]          # evidence can be replaced
            plan.set_evidence(observed[1], batch=1)
            plan.set_evidence(observed[2], batch=2)
        with pytest.raises(ValueError):
            plan.set_evidence({labels[0]: spec["sizes"][labels[0]]})      # state out of range
        plan.close()


def _indicator_potentials(spec, base, observed):
    pots = [np.asarray(p, dtype=np.float64).copy() for p in base]
    for var, state in observed.items():
        host = next(c for c in range(spec["n_cliques"]) if var in spec["node_vars"][c])
        ind = np.zeros(spec["sizes"][var])
        ind[state] = 1.0
        shape = [1] * pots[host].ndim
        shape[spec["node_vars"][host].index(var)] = spec["sizes"][var]
        pots[host] = pots[host] * ind.reshape(shape)
    return pots


@pytest.mark.parametrize("nb", [16, 64, 11])
def test_multiset_plans_many_evidence_sets(nb):
    """JTP_MULTISET: evidence sets share the clique tables and are served eight at a
    time by one pass over a table (jt_multi_flow); B = 16 / 64 sets (and 11: a padded last group) on trees of
    cardinality 2, 3 and mixed, float64 and float32, dataflow and per-level launches, two propagates (both
    halves of the message arenas).  Expected: the oracle on indicator-multiplied potentials - every separator
    belief, Z, and the clique beliefs / marginals formed on demand."""
    from test_planner_emulated import random_junction_tree
    cases = [(synthetic.wide_binary_tree(n_cliques=15, width=13, sep=6, card=2, seed=2), "f64", {}),
             (synthetic.random_tree(n_cliques=9, width=6, sep=3, card=3, seed=4), "f64", {"level_launches": True}),
             (synthetic.wide_binary_tree(n_cliques=7, width=14, sep=7, card=2, seed=6), "f32", {"block_log2": 11}),
             (random_junction_tree(np.random.default_rng(77), n_cliques=12, max_width=6)[0], "f64", {"flow_tickets": True})]
    for ci, (spec, dtype, opts) in enumerate(cases):
        n = spec["n_cliques"]
        np_dt = np.float32 if dtype == "f32" else np.float64
        rng0 = np.random.default_rng(ci)
        base = [(rng0.uniform(0.5, 1.5, [spec["sizes"][v] for v in spec["node_vars"][c]]) * spec["scales"][c]).astype(np_dt)
                for c in range(n)]
        base += [np.ones([spec["sizes"][v] for v in labs]) for labs in spec["node_vars"][n:]]      # separators (unused values)
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, n_batch=nb, multiset=True, **opts)
        assert plan.describe()["multiset"] == 1
        for c in range(n):
            plan.set_potential(c, base[c])
        labels = sorted(spec["sizes"])
        observed = []
        for b in range(nb):
            rng = np.random.default_rng(500 + b)
            k = min(len(labels), b % 5)
            observed.append({labels[i]: int(rng.integers(0, spec["sizes"][labels[i]])) for i in rng.choice(len(labels), size=k, replace=False)})
            plan.set_evidence(observed[b], batch=b)
        with pytest.raises(ValueError):
            plan.propagate(0, 1)                              # all sets of a multi-set plan run together
        rtol = RTOL32 if dtype == "f32" else RTOL64
        for rep in range(2):
            plan.propagate()
            assert plan.stats()["flow_fallbacks"] == 0
            check = range(nb) if rep == 0 and nb <= 16 else [0, 1, nb // 2, nb - 1]
            for b in check:
                want, z = oracle.beliefs_exact(spec["tree"], _indicator_potentials(spec, base, observed[b]), spec["node_vars"], return_z=True)
                assert abs(plan.z(batch=b) - z) <= rtol * abs(z) + 1e-300, (ci, b)
                for node in range(len(spec["node_vars"])):
                    close(plan.belief(node, batch=b), want[node], rtol=rtol, what="case %d set %d node %d" % (ci, b, node))
                c = int(np.random.default_rng(b).integers(0, n))
                lab = list(spec["node_vars"][c])[:2]
                axes = tuple(range(len(lab), len(spec["node_vars"][c])))
                close(plan.marginal(c, lab, batch=b), want[c].sum(axis=axes), rtol=rtol, what="marginal")
            observed[0], observed[nb - 1] = observed[nb - 1], observed[0]       # evidence can be replaced
            plan.set_evidence(observed[0], batch=0)
            plan.set_evidence(observed[nb - 1], batch=nb - 1)
        plan.close()


def test_propagate_restages_only_cliques_whose_factors_changed(golden):
    """JunctionTree.propagate keeps a content digest of the factor tables behind every clique of its plan and
    forms a clique potential again only when one of them changed - also when it was changed IN PLACE (the
    reference recomputes every clique on every call, FIXME at junctiontree.py:206-214)."""
    g = golden("networks.npz")
    net = g.meta["networks"]["abcdefgh"]
    tree = jt.create_junction_tree(net["factors"], dict(net["sizes"]))
    values = [np.array(v, dtype=np.float64) for v in g.arrs(net["values"])]
    ct = tree.clique_tree

    def want(vals):
        return oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, net["factors"],
                                net["sizes"], vals)

    engine.clear_plan_cache()
    out = tree.propagate(values)
    plan = tree.plan("f64")
    assert plan.staged_cliques == len(ct.maxcliques)
    for o, w in zip(out, want(values)):
        close(o, w)
    out = tree.propagate([v.copy() for v in values])          # equal content in new arrays: nothing to stage
    assert plan.staged_cliques == 0
    for o, w in zip(out, want(values)):
        close(o, w)
    values[3] *= 1.5                                          # one factor, modified in place
    out = tree.propagate(values)
    assert plan.staged_cliques == 1
    for o, w in zip(out, want(values)):
        close(o, w)
    comp.compute_beliefs(tree.tree, [np.ones([net["sizes"][v] for v in labs]) for labs in list(ct.maxcliques) + list(tree.separators)],
                         [list(c) for c in ct.maxcliques] + [list(x) for x in tree.separators])
    out = tree.propagate(values)                              # the plan's tables were overwritten behind propagate's back
    for o, w in zip(out, want(values)):
        close(o, w)


@pytest.mark.parametrize("card,width,sep", [(3, 8, 4), (5, 6, 3), (6, 5, 2), (7, 5, 3), (3, 9, 5)])
def test_wide_cliques_of_odd_cardinalities(card, width, sep):
    """Tables stored at their true cardinalities - above the thread part (mixed-radix rows, rows that do not exist read
    the zero row: round 2) and inside it (round 3: the thread part's variables are mixed-radix digits of a row, arena
    <= 1.25 x the host tables for cardinality 3 width 8/9 and cardinality 5 width 6): beliefs, Z and marginals vs the oracle in float64 and float32, dataflow and
    per-level launches, the padded round-1 layout for comparison (bit-identical results are not expected: the
    summation order differs), hard evidence (its masks are over the LOGICAL index) and multi-set plans."""
    spec = synthetic.wide_binary_tree(n_cliques=7, width=width, sep=sep, card=card, seed=card)
    pots = synthetic.potentials_for(spec, seed=11)
    want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
    arenas = {}
    host_elems = sum(card ** width for _ in range(spec["n_cliques"]))
    for opts in ({}, {"level_launches": True}, {"no_compact": True}, {"block_log2": 10}, {"split_variants": True}, {"flow_tickets": True}):
        for dtype in ("f64", "f32"):
            cast = [p.astype(np.float32) for p in pots] if dtype == "f32" else pots
            ref = oracle.beliefs_exact(spec["tree"], cast, spec["node_vars"]) if dtype == "f32" else want
            plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, **opts)
            d = plan.describe()
            arenas[bool(opts.get("no_compact"))] = d["arena_elems"]
            assert d["compact"] == (0 if opts.get("no_compact") else 1)
            # round 3: the thread part too is stored at true cardinalities (mixed-radix rows reached through a per-clique
            # thread map, kernels *_mix): the arena is the host tables plus alignment, not (4/3)^5 = 4.2 x them
            # (cliques whose bit-field thread part would be at least 0.6 full keep it: cardinality 7, (7/8)^3)
            assert d["tmix"] == (0 if opts.get("no_compact") or card == 7 else 1)
            if not opts.get("no_compact") and (card, width) in ((3, 8), (3, 9), (5, 6)):
                assert d["arena_elems"] <= 1.25 * host_elems, (d["arena_elems"], host_elems)
                assert all(p["trow"] < 2 ** d["TB"] for p in d["pnodes"] if p["tmix"])
                # (cardinality 5 in float32: one bit of a fourth variable lies below bit TB - the rows of a bit-field thread
                #  part at 6/5 of the true size; cardinality 3 would cost 4/3 that way and moves the variable up whole)
                assert {p["tsplit"] >= 0 for p in d["pnodes"] if p["tmix"] and p["nbits"] > d["TB"] + 2} <= {card == 5 and dtype == "f32"}
            for c in range(spec["n_cliques"]):
                plan.set_potential(c, cast[c])
            plan.propagate()
            for node in range(len(spec["node_vars"])):
                close(plan.belief(node), ref[node], rtol=RTOL32 if dtype == "f32" else RTOL64, what="%r %s node %d" % (opts, dtype, node))
            assert abs(plan.z() - z) <= (1e-6 if dtype == "f32" else 1e-11) * z
            lab = list(spec["node_vars"][3])[1:3]
            close(plan.marginal(3, lab), ref[3].sum(axis=tuple(i for i in range(width) if i not in (1, 2))), rtol=RTOL32 if dtype == "f32" else RTOL64)
            plan.close()
    assert arenas[False] < arenas[True]
    # hard evidence on every position of a clique's variable list in turn (thread part, loop and chunk digits)
    labels = sorted(spec["sizes"])
    nb = 6
    for multiset in (False, True):
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", n_batch=nb, share_potentials=True, multiset=multiset)
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, pots[c])
        observed = []
        for b in range(nb):
            rng = np.random.default_rng(900 + b)
            obs = {labels[i]: int(rng.integers(0, card)) for i in rng.choice(len(labels), size=min(len(labels), 2 + b), replace=False)}
            observed.append(obs)
            plan.set_evidence(obs, batch=b)
        plan.propagate(0, nb)
        for b in range(nb):
            w, zb = oracle.beliefs_exact(spec["tree"], _indicator_potentials(spec, pots, observed[b]), spec["node_vars"], return_z=True)
            assert abs(plan.z(batch=b) - zb) <= 1e-11 * zb + 1e-300
            for node in range(len(spec["node_vars"])):
                close(plan.belief(node, batch=b), w[node], what="multiset %r set %d node %d" % (multiset, b, node))
        plan.close()


def test_grid_mrf_through_public_api_vs_bruteforce():
    """Loopy pairwise models (BASELINE config 3 family at brute-forceable size): 3x3, 4x4 and
    3x3x2 binary lattices through create_junction_tree / propagate (the reference is wrong or
    raises on these)."""
    rng = np.random.default_rng(3)
    for dims in ((3, 3), (4, 4), (3, 3, 2)):
        names = {pos: "v" + "_".join(map(str, pos)) for pos in np.ndindex(*dims)}
        factors = []
        for pos in np.ndindex(*dims):
            for ax in range(len(dims)):
                nb = list(pos)
                nb[ax] += 1
                if nb[ax] < dims[ax]:
                    factors.append([names[pos], names[tuple(nb)]])
        sizes = {v: 2 for v in names.values()}
        values = [rng.uniform(0.5, 1.5, (2, 2)) * 0.7 for _ in factors]
        ops = []
        for v, f in zip(values, factors):
            ops += [v, f]
        out = jt.create_junction_tree(factors, sizes).propagate(values)
        for o, f in zip(out, factors):
            close(o, oracle.labelled_einsum(*ops, f), rtol=1e-10, what=str(dims))


@pytest.mark.parametrize("seed", range(10))
def test_random_trees_mixed_cardinalities_on_device(seed):
    """Random junction trees: cardinalities 1..8 (non powers of two are zero padded on the device),
    empty separators, up to 5 children per clique, 0..5 variables shared per edge."""
    from test_planner_emulated import random_junction_tree
    rng = np.random.default_rng(200 + seed)
    spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 40)), max_width=6)
    want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
    opts = [{}, {"block_log2": 10}, {"layout_policy": 1}, {"keep_root": True}, {"split_variants": True}][seed % 5]
    for dtype in ("f64", "f32"):
        cast = [p.astype(np.float32) for p in pots] if dtype == "f32" else pots
        ref = oracle.beliefs_exact(spec["tree"], cast, spec["node_vars"]) if dtype == "f32" else want
        out, zz = run_plan(spec, cast, dtype, **opts)
        for o, w in zip(out, ref):
            close(o, w, rtol=RTOL32 if dtype == "f32" else RTOL64, what="seed %d %s" % (seed, dtype))
    assert abs(zz - z) <= 1e-5 * abs(z)


def test_plan_cache_is_bounded_by_device_bytes():
    """The plan cache behind `compute_beliefs` / `propagate` keeps device plans up to a byte budget, least recently
    used first out (round 2: up to 16 plans of any size).  Twenty distinct mid-size trees under a budget of four of
    them: the cached bytes stay within the budget, a plan used again is kept, and the device memory of forgotten plans
    is given back."""
    import ctypes as C
    from junctiontree_amd import _capi

    def free_bytes():
        free, total = C.c_uint64(0), C.c_uint64(0)
        _capi.check(_capi.lib().jtp_device_memory(0, C.byref(free), C.byref(total)))
        return free.value

    specs = [synthetic.wide_binary_tree(n_cliques=7, width=18, sep=9, card=2, seed=100 + i) for i in range(20)]
    first = engine.plan_for(specs[0]["tree"], specs[0]["node_vars"], specs[0]["sizes"], "f32")
    one = engine.plan_cache_info()["device_bytes"]
    assert one >= 7 * 2 * 4 * 2 ** 18                          # seven tables of 2^18 floats, potentials and beliefs
    start_free = free_bytes() + one
    try:
        engine.set_plan_cache_budget(4 * one + one // 2)
        del first
        for i, spec in enumerate(specs):
            plan = engine.plan_for(spec["tree"], spec["node_vars"], spec["sizes"], "f32")
            info = engine.plan_cache_info()
            assert info["device_bytes"] <= info["budget_bytes"] and info["plans"] <= 4, info
            if i % 3 == 0:                                       # keep plan 0 recently used: it must survive
                assert engine.plan_for(specs[0]["tree"], specs[0]["node_vars"], specs[0]["sizes"], "f32").stats()["device_bytes"] == one
            pots = synthetic.potentials_for(spec, seed=1, dtype=np.float32)
            for c in range(spec["n_cliques"]):
                plan.set_potential(c, pots[c])
            plan.propagate()
            assert np.isfinite(plan.z())
            del plan
        info = engine.plan_cache_info()
        assert info["evictions"] >= 15 and info["hits"] >= 7
        import gc
        gc.collect()
        assert start_free - free_bytes() <= 5 * one + (64 << 20)        # forgotten plans gave their memory back
    finally:
        engine.set_plan_cache_budget(None)


def _joint_marginals(factors, sizes, values):
    names = sorted(sizes)
    ax = {v: i for i, v in enumerate(names)}
    ops = []
    for f, val in zip(factors, values):
        ops += [np.asarray(val, dtype=np.float64), [ax[v] for v in f]]
    joint = np.einsum(*ops, list(range(len(names))))
    return [np.einsum(joint, list(range(len(names))), [ax[v] for v in f]) for f in factors]


def test_factor_tables_are_staged_in_one_call_and_only_where_they_changed():
    """`JunctionTree.propagate` forms the clique potentials with ONE `jtp_set_potential_products` call (round 3: a copy and
    a launch per clique) and only for cliques whose factor tables differ from what the plan holds - compared by VALUE, so
    an array updated in place is seen, the same values in new arrays are not staged again."""
    rng = np.random.default_rng(3)
    names = list("abcdefgh")
    sizes = dict(zip(names, (2, 3, 2, 4, 3, 2, 5, 2)))
    factors = [["a", "b"], ["b", "c"], ["c", "d", "e"], ["e", "f"], ["f", "g"], ["g", "h"], ["a"], ["d"], ["h", "g"]]
    values = [rng.uniform(0.2, 1.0, [sizes[v] for v in f]) for f in factors]
    tree = jt.create_junction_tree(factors, sizes)
    n_cl = len(tree.clique_tree.maxcliques)
    got = tree.propagate(values)
    plan = tree.plan("f64")
    assert plan.staged_cliques == n_cl
    for g, w in zip(got, _joint_marginals(factors, sizes, values)):
        close(g, w)
    tree.propagate([v.copy() for v in values])               # same values, other arrays
    assert plan.staged_cliques == 0
    values[3][0, 1] *= 3.0                                    # one table, in place
    got = tree.propagate(values)
    assert plan.staged_cliques == 1
    for g, w in zip(got, _joint_marginals(factors, sizes, values)):
        close(g, w)
    mixed = [v.astype(np.float32) if i % 2 else v for i, v in enumerate(values)]      # float32 and float64 tables mixed
    mixed[6] = np.array([1, 2])                                                       # ... and an integer one
    values[6] = np.array([1.0, 2.0])
    got = tree.propagate(mixed)
    for g, w in zip(got, _joint_marginals(factors, sizes, [np.asarray(m, dtype=np.float64) for m in mixed])):
        close(g, w, rtol=RTOL64)
    plan.set_potential(0, np.ones([sizes[v] for v in tree.clique_tree.maxcliques[0]]))    # behind the staging's back:
    got = tree.propagate(mixed)                                                            # everything is formed again
    assert plan.staged_cliques == n_cl
    for g, w in zip(got, _joint_marginals(factors, sizes, [np.asarray(m, dtype=np.float64) for m in mixed])):
        close(g, w, rtol=RTOL64)
    with pytest.raises(ValueError):
        tree.propagate(values[:-1] + [np.ones((3, 3))])      # a table of the wrong shape
    engine.clear_plan_cache()


def test_a_caller_may_name_the_factors_it_changed():
    """`tree.propagate(values, changed=[...])` (round 6; the reference's FIXME at `junctiontree.py:206-214`): nothing is compared,
    the cliques of the named factors are formed again, the others stand; "all" forms every clique; an unnamed change is - by
    contract - not seen until a call without `changed` compares everything again."""
    rng = np.random.default_rng(5)
    names = list("abcdefgh")
    sizes = dict(zip(names, (2, 3, 2, 4, 3, 2, 5, 2)))
    factors = [["a", "b"], ["b", "c"], ["c", "d", "e"], ["e", "f"], ["f", "g"], ["g", "h"], ["a"], ["d"], ["h", "g"]]
    values = [rng.uniform(0.2, 1.0, [sizes[v] for v in f]) for f in factors]
    tree = jt.create_junction_tree(factors, sizes)
    n_cl = len(tree.clique_tree.maxcliques)
    got = tree.propagate(values, changed="all")               # (first call: there is nothing to trust yet - everything is staged)
    plan = tree.plan("f64")
    assert plan.staged_cliques == n_cl
    for g, w in zip(got, _joint_marginals(factors, sizes, values)):
        close(g, w)
    values[3] = values[3] * 2.0
    values[7] = values[7] + 0.5
    got = tree.propagate(values, changed=[3, 7, 3])
    assert 1 <= plan.staged_cliques <= 2
    for g, w in zip(got, _joint_marginals(factors, sizes, values)):
        close(g, w)
    stale = [v.copy() for v in values]
    values[0] = values[0] * 5.0                                # not named: not looked at
    got = tree.propagate(values, changed=[])
    assert plan.staged_cliques == 0
    for g, w in zip(got, _joint_marginals(factors, sizes, stale)):
        close(g, w)
    got = tree.propagate(values)                               # a call without `changed` compares everything
    assert plan.staged_cliques >= 1
    for g, w in zip(got, _joint_marginals(factors, sizes, values)):
        close(g, w)
    values = [v * 1.5 for v in values]
    got = tree.propagate(values, changed="all")
    assert plan.staged_cliques == n_cl
    for g, w in zip(got, _joint_marginals(factors, sizes, values)):
        close(g, w)
    with pytest.raises(IndexError):
        tree.propagate(values, changed=[len(factors)])
    with pytest.raises(ValueError):
        tree.propagate(values, changed="some")
    engine.clear_plan_cache()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_cliques_of_many_factors_and_of_large_factor_tables(dt):
    """`jtp_set_potential_products`: a clique that collects more factors than one pass multiplies (JT_EVAL_MAX_F = 8: the rest
    multiply INTO the table in further launches), factor tables too large for LDS (gathered from the staging buffer) and
    beyond the compare-by-value limit (handed over on every call), a broadcast axis."""
    rng = np.random.default_rng(5)
    names = list("abcdefghijklmnopq")[:17]
    sizes = {v: 2 for v in names}
    big = names[:15]                                          # 2^15 entries: 256 KiB as float64 - beyond the 48 KiB of LDS tables
    factors = [big] + [[a, b] for a, b in zip(big[:-1], big[1:])][:10] + [[names[15], names[0]], [names[16], names[15]]] + [[v] for v in big[:3]]
    values = [rng.uniform(0.5, 1.5, [sizes[v] for v in f]).astype(dt) for f in factors]
    tree = jt.create_junction_tree(factors, sizes)
    members = tree.clique_tree._members()
    assert max(len(m) for m in members) > 8
    got = tree.propagate(values)
    want = _joint_marginals(factors, sizes, values)
    for g, w in zip(got, want):
        close(g, w, rtol=RTOL32 if dt == np.float32 else RTOL64)
    # the same through the plan's own entry, with a length-1 (broadcast) axis and tables above the compare-by-value limit
    plan = tree.plan("f32" if dt == np.float32 else "f64")
    c0 = next(c for c, m in enumerate(members) if 0 in m)
    labels = [factors[i] for i in members[c0]]
    arrays = [values[i] for i in members[c0]]
    arrays[1] = arrays[1][:1, :]                               # factor [a, b] given for a = 0 only: broadcast along a
    plan.set_potential_product(c0, arrays, labels)
    plan.propagate()
    vals2 = list(values)
    vals2[members[c0][1]] = np.broadcast_to(arrays[1], values[members[c0][1]].shape)
    for f, w in zip(factors, _joint_marginals(factors, sizes, vals2)):
        mc = tree.clique_tree.factor_to_maxclique[factors.index(f)]
        close(plan.marginal(mc, f), w, rtol=RTOL32 if dt == np.float32 else RTOL64)
    from junctiontree_amd import engine as eng
    old = eng._DIGEST_LIMIT
    eng._DIGEST_LIMIT = 1 << 12                               # the 2^15-entry table now counts as large: always handed over
    try:
        engine.clear_plan_cache()
        got = tree.propagate(values)
        assert tree.plan("f32" if dt == np.float32 else "f64").staged_cliques == len(members)
        got = tree.propagate(values)
        assert tree.plan("f32" if dt == np.float32 else "f64").staged_cliques == 1          # only the large table's clique
        for g, w in zip(got, want):
            close(g, w, rtol=RTOL32 if dt == np.float32 else RTOL64)
    finally:
        eng._DIGEST_LIMIT = old
        engine.clear_plan_cache()


def test_junction_trees_of_a_given_elimination_order():
    """`create_junction_tree(factors, sizes, order=...)`: whatever elimination order builds the tree, the factor marginals are
    the brute-force ones (a 3 x 4 lattice: min-fill, column by column, row by row, a random order)."""
    factors, sizes, values = synthetic.lattice_mrf(3, 4, 3, seed=2, dtype=np.float64)
    want = _joint_marginals(factors, sizes, values)
    rng = np.random.default_rng(0)
    orders = [None, synthetic.lattice_column_order(3, 4), list(range(12)), [int(v) for v in rng.permutation(12)], [5, 6]]
    shapes = set()
    for order in orders:
        tree = jt.create_junction_tree(factors, sizes, order=order)
        shapes.add(tuple(sorted(len(c) for c in tree.clique_tree.maxcliques)))
        for g, w in zip(tree.propagate(values), want):
            close(g, w)
    assert len(shapes) > 1                                    # (the orders really give different trees)
    with pytest.raises(ValueError):
        jt.create_junction_tree(factors, sizes, order=[1, 1])
    engine.clear_plan_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("card,width,sep,dtype", [(3, 11, 5, "f32"), (5, 7, 3, "f32"), (3, 10, 5, "f64"), (6, 6, 3, "f32"), (7, 6, 3, "f32"), (7, 5, 2, "f64")])
def test_chunks_that_do_not_exist_are_run_once_per_arena_not_once_per_propagate(card, width, sep, dtype):
    """Tables at true cardinalities (round 5): the workgroups of chunks whose own digits do not exist write nothing but the zeros of
    their partial copies.  They are no longer in the block lists of a single-set plan - `jtp_plan_create` zeroes those copies once
    per arena half (`init_blocks` in the description; cardinality 7: bit-field rows, the others mixed-radix rows) - so every later
    propagate must still find those zeros: three propagates with changing potentials (both arena
    halves in use), each against the oracle, in dataflow launches and in per-level launches; the same plan with JTP_KEEP_INVALID=1
    (rounds 2-4: every chunk in the lists) gives the same bits."""
    import os
    spec = synthetic.wide_binary_tree(n_cliques=7, width=width, sep=sep, card=card, seed=card + width)
    results = {}
    for mode in ("flow", "levels", "keep"):
        if mode == "keep":
            os.environ["JTP_KEEP_INVALID"] = "1"
        try:
            plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, level_launches=(mode == "levels"))
        finally:
            os.environ.pop("JTP_KEEP_INVALID", None)
        d = plan.describe()
        assert d["tmix"] == (0 if card == 7 else 1)
        if mode == "keep":
            assert not d["init_blocks"] and any(b[23] & 1 for b in d["blocks"])
        else:
            assert d["init_blocks"] and not any(b[23] & 1 for b in d["blocks"])
        got = []
        for rep in range(3):
            pots = synthetic.potentials_for(spec, seed=20 + rep)
            for c in plan.cliques:
                plan.set_potential(c, pots[c])
            plan.propagate()
            want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
            tol = 2e-5 if dtype == "f32" else 1e-11
            for c in plan.cliques:
                b = plan.belief(c)
                np.testing.assert_allclose(b, want[c], rtol=tol, atol=tol * np.max(want[c]))
            got.append([plan.belief(c).copy() for c in plan.cliques] + [plan.belief(s).copy() for s in plan.seps])
        assert plan.stats()["flow_fallbacks"] == 0
        results[mode] = got
        plan.close()
    for rep in range(3):
        for a, b in zip(results["flow"][rep], results["keep"][rep]):
            np.testing.assert_array_equal(a, b)
        for a, b in zip(results["flow"][rep], results["levels"][rep]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("card,width,sep", [(3, 8, 4), (5, 6, 3), (6, 5, 2)])
def test_multiset_plans_with_odd_cardinalities_share_evidence_free_subtrees_while_evidence_changes(monkeypatch, card, width, sep):
    """Multi-set plans whose tables are stored at true cardinalities (round 5): the partial copies of chunks that do not exist are
    zeroed once per arena; the copy pass of the evidence-free subtrees (`jt_multi_fanout`) marks whole messages "unwritten" in the
    other arena half - those copies included - and the engine sets them back.  Evidence that MOVES between propagates turns
    skipped tasks into running ones and back (both arena halves): every set's Z and beliefs against the oracle each time, no
    dataflow time-out."""
    monkeypatch.setenv("JTP_EF_SHARE", "1")
    spec = synthetic.wide_binary_tree(n_cliques=15, width=width, sep=sep, card=card, seed=card)
    pots = synthetic.potentials_for(spec, seed=5)
    labels = sorted(spec["sizes"])
    nb = 12
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", n_batch=nb, share_potentials=True, multiset=True)
    d = plan.describe()
    assert d["init_blocks"] and not any(b[23] & 1 for b in d["blocks"])
    for c in range(spec["n_cliques"]):
        plan.set_potential(c, pots[c])
    for rnd in range(4):
        rng = np.random.default_rng(40 + rnd)
        observed = []
        for b in range(nb):
            k = 0 if (b + rnd) % 3 == 0 else 1 + (b + rnd) % 4          # some sets without evidence, the others with evidence that moves
            obs = {labels[i]: int(rng.integers(0, card)) for i in rng.choice(len(labels), size=k, replace=False)}
            observed.append(obs)
            plan.set_evidence(obs, batch=b)
        plan.propagate(0, nb)
        for b in range(nb):
            w, zb = oracle.beliefs_exact(spec["tree"], _indicator_potentials(spec, pots, observed[b]), spec["node_vars"], return_z=True)
            assert abs(plan.z(batch=b) - zb) <= 1e-11 * zb + 1e-300, (rnd, b)
            for node in (0, spec["n_cliques"] - 1, spec["n_cliques"], len(spec["node_vars"]) - 1):
                close(plan.belief(node, batch=b), w[node], what="round %d set %d node %d" % (rnd, b, node))
    st = plan.stats()
    assert st["flow_fallbacks"] == 0 and st["launch_mode"] == "flow"
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,mode", [("f64", "flow"), ("f32", "flow"), ("f64", "tickets"), ("f32", "levels")])
def test_skipped_producers_re_arm_the_partial_copies_their_reduce_tasks_sum(monkeypatch, dtype, mode):
    """Round 6 (advisor, round 5): a multi-set plan that shares evidence-free subtrees skips a collect task for a group whose sets observe
    nothing below it.  Where a REDUCE task sums that producer's partial copies, the copies are entries of their own - a skipped
    producer used to leave them alone, so when evidence moved into the subtree, out of it and back in, the reduce task of the third
    propagate found the first propagate's values in its arena half (no marker = written) and summed them.  The copy pass now re-arms
    them.  Evidence of ONE subtree goes in, out and in again with another observed state, six propagates over both arena halves,
    every set's Z and beliefs against the oracle."""
    monkeypatch.setenv("JTP_EF_SHARE", "1")
    monkeypatch.setenv("JTP_REDUCE_MIN", "2")
    spec = synthetic.wide_binary_tree(n_cliques=15, width=13, sep=6, card=2, seed=2)
    pots = synthetic.potentials_for(spec, seed=8)
    cast = [p.astype(np.float32) for p in pots] if dtype == "f32" else pots
    nb = 8
    # (round 6: with active lists a workgroup of one run of eight sets waits for entries of another run's workgroup - also in ticket order,
    #  also with one launch per level)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, n_batch=nb, share_potentials=True, multiset=True,
                       flow_tickets=(mode == "tickets"), level_launches=(mode == "levels"))
    d = plan.describe()
    assert all(L["blk_off"] % 8 == 0 and L["nblocks"] % 8 == 0 for L in d["launches"])      # (every level starts on a multiple of eight records)
    ups = [s for s in d["pseps"] if s["up_red_task"] >= 0]
    assert ups, "this tree must have upward reduce tasks"
    # a variable private to the subtree below a separator with an upward reduce task (it is in the child clique, not in the separator)
    sp = ups[-1]
    child_vars = [v for v in d["pnodes"][sp["child"]]["vars"] if v not in sp["vars"]]
    inside = plan.var_labels[child_vars[0]]
    for c in range(spec["n_cliques"]):
        plan.set_potential(c, cast[c])
    tol = RTOL32 if dtype == "f32" else RTOL64
    # (a group's task is skipped only when NO set of the group observes anything below it: all eight sets move together)
    for rnd, state in enumerate([0, None, 1, None, None, 0]):
        observed = []
        for b in range(nb):
            obs = {}
            if state is not None:
                obs[inside] = (state + b) % 2
            observed.append(obs)
            plan.set_evidence(obs, batch=b)
        plan.propagate(0, nb)
        for b in range(nb):
            w, zb = oracle.beliefs_exact(spec["tree"], _indicator_potentials(spec, cast, observed[b]), spec["node_vars"], return_z=True)
            assert abs(plan.z(batch=b) - zb) <= (1e-6 if dtype == "f32" else 1e-11) * zb + 1e-300, (rnd, b)
            for node in (0, sp["child"], sp["node"], spec["n_cliques"] - 1):
                close(plan.belief(node, batch=b), w[node], rtol=tol, what="round %d set %d node %d" % (rnd, b, node))
    st = plan.stats()
    assert st["flow_fallbacks"] == 0 and st["launch_mode"] == {"flow": "flow", "tickets": "flow_tickets", "levels": "level"}[mode]
    plan.close()


@pytest.mark.gpu
def test_factor_marginals_folded_into_the_propagate(monkeypatch):
    """Round 6 (`jtp_tree_desc.fold_*`): `JunctionTree.propagate` returns factor marginals only (`junctiontree.py:327-331`), so its plan is
    told the list; the requests on cliques that keep no table become tasks of the propagate's own launch and `factor_marginals` with
    that list only unpacks them.  Same marginals as the read-out forms (JTP_NO_FOLD=1) and as the oracle's `propagate`; dataflow and
    per-level launches bit for bit; an evidence set that observes something and any OTHER request list are served by the read-out."""
    factors, sizes, values = synthetic.lattice_mrf(5, 16, 4, dtype=np.float64)
    outs, want = {}, None
    # (a lattice this small plans as a chain of latency-bound levels, whose distribute kernel is built without folded tasks: planned
    #  here as the large ones are)
    monkeypatch.setenv("JTP_TINY_LEVEL_ELEMS", "0")
    # (... and the planner folds by itself only where the distribute levels leave the chip's slots idle - the column-sweep tree below;
    #  JTP_FOLD=1: wherever the plan's form allows)
    monkeypatch.setenv("JTP_FOLD", "1")
    for mode in ("fold", "nofold", "fold_levels"):
        if mode == "nofold":
            monkeypatch.setenv("JTP_NO_FOLD", "1")
        else:
            monkeypatch.delenv("JTP_NO_FOLD", raising=False)
        engine.clear_plan_cache()
        tree = jt.create_junction_tree(factors, sizes)
        ct = tree.clique_tree
        if want is None:
            want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, values)
        if mode == "fold_levels":
            tree._opts["level_launches"] = True
        out = tree.propagate(values)
        plan = tree.plan("f64")
        n_fold = sum(1 for t in plan.describe()["tasks"] if t["fold"])
        assert (n_fold > 0) == (mode != "nofold"), (mode, n_fold)
        assert all(t["lean_off"] > 0 and t["unit"] and t["mode"] == 0 and 1 <= t["n_out"] <= 3 for t in plan.describe()["tasks"] if t["fold"])
        outs[mode] = [o.copy() for o in out]
        if mode == "fold":
            # the plain hot-path plan of the same tree has none; a second call gives the same bits
            assert not any(t["fold"] for t in tree.plan("f64", fold=False).describe()["tasks"])
            for a, b in zip(tree.propagate(values, changed=[]), out):
                np.testing.assert_array_equal(a, b)
            # another request list: the read-out
            sub = [(ct.factor_to_maxclique[i], factors[i]) for i in (0, 5, 11)]
            for (c, labs), m in zip(sub, plan.marginals(sub)):
                close(m, want[[0, 5, 11][sub.index((c, labs))]], what="subset request")
            # an evidence set that observes something: the folded tasks of cliques that host the variable do not run - the read-out does
            var = factors[7][0]
            plan.set_evidence({var: 1})
            plan.propagate()
            ev_vals = [v.copy() for v in values]
            ind = np.zeros(sizes[var])
            ind[1] = 1.0
            ev_vals[7] = ev_vals[7] * ind.reshape([-1 if lab == var else 1 for lab in factors[7]])
            ev_want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, ev_vals)
            for o, w in zip(plan.factor_marginals(factors, ct.factor_to_maxclique), ev_want):
                close(o, w, what="with evidence")
            plan.set_evidence({})
    engine.clear_plan_cache()
    for mode in outs:
        for o, w in zip(outs[mode], want):
            close(o, w, what=mode)
    for a, b in zip(outs["fold"], outs["nofold"]):
        close(a, b, what="folded against read-out")
    for a, b in zip(outs["fold"], outs["fold_levels"]):
        np.testing.assert_array_equal(a, b)
    # propagates queued back to back on a lattice whose folded tasks read messages that arrive as several partial copies, in the launch
    # that makes them: every copy is waited for (the first build waited for the first one only; tools/soak.py found NaN marginals)
    monkeypatch.delenv("JTP_TINY_LEVEL_ELEMS")
    factors, sizes, values = synthetic.lattice_mrf(6, 40, 8)
    tree = jt.create_junction_tree(factors, sizes)
    first = tree.propagate(values)
    plan = tree.plan("f32")
    d = plan.describe()
    assert any(t["fold"] and any(m["npart"] > 1 and m["same_launch"] for m in t["in"]) for t in d["tasks"])
    for i in range(600):
        plan.propagate(sync=False)
        if i % 10 == 9:
            for a, b in zip(plan.factor_marginals(factors, tree.clique_tree.factor_to_maxclique), first):
                np.testing.assert_array_equal(a, b)
    assert plan.stats()["flow_fallbacks"] == 0
    engine.clear_plan_cache()
    # the planner's own choice (no JTP_FOLD): the min-fill tree of that lattice fills the chip level after level - its marginals stay with
    # the read-out; the column-sweep tree's levels leave slots idle - folded; both give the oracle's marginals
    monkeypatch.delenv("JTP_FOLD")
    want = None
    for order, folds in ((None, False), (synthetic.lattice_column_order(6, 40), True)):
        tree = jt.create_junction_tree(factors, sizes, order=order)
        out = tree.propagate(values)
        assert any(t["fold"] for t in tree.plan("f32").describe()["tasks"]) == folds
        if want is None:
            ct = tree.clique_tree
            want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, [np.asarray(v, dtype=np.float64) for v in values])
        for o, w in zip(out, want):
            close(o, w, rtol=RTOL32, what="planner's choice, %s tree" % ("column-sweep" if order else "min-fill"))
        engine.clear_plan_cache()


@pytest.mark.gpu
def test_lean_and_generic_unit_passes_agree(monkeypatch):
    """Round 6: the lean unit pass (`jt_unit_lean`, the default) and the generic one (`JTP_NO_LEAN=1`) on a lattice whose junction tree is
    mostly unit cliques: every factor marginal of both against the oracle's `propagate`, the two against each other to rounding (the
    lean record orders a task's incoming tables its own way: another product order), dataflow and per-level launches of the lean pass
    bit for bit, the same propagate ten times the same bits."""
    factors, sizes, values = synthetic.lattice_mrf(5, 16, 4, dtype=np.float64)
    want = None
    outs = {}
    for mode in ("lean", "generic", "lean_levels"):
        if mode == "generic":
            monkeypatch.setenv("JTP_NO_LEAN", "1")
        else:
            monkeypatch.delenv("JTP_NO_LEAN", raising=False)
        engine.clear_plan_cache()
        tree = jt.create_junction_tree(factors, sizes)
        if want is None:
            ct = tree.clique_tree
            want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, values)
        if mode == "lean_levels":
            tree._opts["level_launches"] = True
        out = tree.propagate(values)
        plan = tree.plan("f64")
        d = plan.describe()
        n_lean = sum(1 for t in d["tasks"] if t["lean_off"] > 0)
        assert (n_lean > 0) == (mode != "generic") and plan.stats()["n_unit_cliques"] > 0
        assert plan.stats()["launch_mode"] == ("level" if mode == "lean_levels" else "flow")
        if mode == "lean":
            for rep in range(10):
                again = tree.propagate(values, changed=[])
                for a, b in zip(out, again):
                    np.testing.assert_array_equal(a, b)
        outs[mode] = [o.copy() for o in out]
    engine.clear_plan_cache()
    for mode in outs:
        for o, w in zip(outs[mode], want):
            close(o, w, what=mode)
    for a, b in zip(outs["lean"], outs["generic"]):
        close(a, b, what="lean against generic")
    for a, b in zip(outs["lean"], outs["lean_levels"]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("card,width,sep,dtype", [(3, 11, 5, "f32"), (3, 10, 5, "f64"), (5, 7, 3, "f32"), (3, 9, 4, "f32")])
def test_mixed_radix_rows_two_per_step(card, width, sep, dtype):
    """Compact mixed-radix rows (round 5): where at most 128 logical threads of the thread part own an entry that exists, two waves
    serve a row and a workgroup works on two rows per step (`vgroups` = 2 in the task records; sums over thread-part variables go
    through LDS adds in wave order).  Beliefs, Z and a marginal against the oracle, dataflow and per-level launches, hard evidence,
    an odd number of rows per workgroup among the tasks; the same propagate twenty times gives the same bits; JTP_NO_VGROUPS=1 (one
    row per step, round 3) agrees to rounding."""
    import os
    spec = synthetic.wide_binary_tree(n_cliques=15, width=width, sep=sep, card=card, seed=3 * card + width)
    pots = synthetic.potentials_for(spec, seed=31)
    cast = [p.astype(np.float32) for p in pots] if dtype == "f32" else pots
    ref, z = oracle.beliefs_exact(spec["tree"], cast, spec["node_vars"], return_z=True)
    tol = RTOL32 if dtype == "f32" else RTOL64
    got = {}
    for mode in ("flow", "levels", "one_row"):
        if mode == "one_row":
            os.environ["JTP_NO_VGROUPS"] = "1"
        try:
            plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, level_launches=(mode == "levels"))
        finally:
            os.environ.pop("JTP_NO_VGROUPS", None)
        d = plan.describe()
        assert d["tmix"] == 1
        groups = {t["vgroups"] for t in d["tasks"] if t["kind"] == 0}
        assert groups == ({0} if mode == "one_row" else {2}), groups
        if mode != "one_row":
            assert any(t["total"] % 2 == 1 for t in d["tasks"] if t["kind"] == 0)      # (27 = 3^3 rows, 25, 9 ...: the second group's last step is empty)
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, cast[c])
        plan.propagate()
        for node in range(len(spec["node_vars"])):
            close(plan.belief(node), ref[node], rtol=tol, what="%s node %d" % (mode, node))
        assert abs(plan.z() - z) <= (1e-6 if dtype == "f32" else 1e-11) * z
        lab = list(spec["node_vars"][3])[1:3]
        close(plan.marginal(3, lab), ref[3].sum(axis=tuple(i for i in range(width) if i not in (1, 2))), rtol=tol)
        first = [plan.belief(node).copy() for node in range(len(spec["node_vars"]))]
        if mode == "flow":
            for rep in range(20):
                plan.propagate()
                for node in (0, 5, spec["n_cliques"], len(spec["node_vars"]) - 1):
                    np.testing.assert_array_equal(plan.belief(node), first[node], err_msg="propagate %d node %d" % (rep, node))
        got[mode] = first
        # hard evidence (its masks are over the LOGICAL index of the thread part)
        labels = sorted(spec["sizes"])
        rng = np.random.default_rng(77)
        obs = {labels[i]: int(rng.integers(0, card)) for i in rng.choice(len(labels), size=4, replace=False)}
        plan.set_evidence(obs)
        plan.propagate()
        w, zb = oracle.beliefs_exact(spec["tree"], _indicator_potentials(spec, cast, obs), spec["node_vars"], return_z=True)
        assert abs(plan.z() - zb) <= (1e-6 if dtype == "f32" else 1e-11) * zb + 1e-300
        for node in (0, 3, spec["n_cliques"] + 1):
            close(plan.belief(node), w[node], rtol=tol, what="%s evidence node %d" % (mode, node))
        assert plan.stats()["flow_fallbacks"] == 0
        plan.close()
    for a, b in zip(got["flow"], got["levels"]):
        np.testing.assert_array_equal(a, b)
