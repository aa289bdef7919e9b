"""Cliques that keep no table (round 5) on the MI355X: the HIP path of plans made with `cover` against the oracle, through the C ABI.

The reference never materialises the variables of a clique that none of its assigned factors covers (`junctiontree/junctiontree.py:52-61`,
evaluate `:203-226`), and `propagate` returns factor marginals only (`:264-274, 327-331`).  The engine now does the same: such a clique is
a unit clique (kernels `jt_pass<..., UNIT>`: no table rows loaded, the product of its factors a static table staged like a message, no
belief table), beliefs and marginals of such cliques are formed on demand.  tests/test_lean_emulated.py checks the same plans on the CPU."""
import numpy as np
import pytest

import jt_oracle as oracle
import junctiontree_amd as jt
from junctiontree_amd import engine, synthetic
from test_gpu_parity import close, RTOL32, RTOL64
from test_lean_emulated import _with_cover, brute_force_marginals
from test_planner_emulated import random_junction_tree, star

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_cached_plans():
    engine.clear_plan_cache()
    yield
    engine.clear_plan_cache()


@pytest.mark.parametrize("h,w,card,dt,sweep", [(4, 8, 8, np.float32, False), (3, 9, 5, np.float64, False), (4, 9, 4, np.float32, True),
                                                (5, 7, 2, np.float64, False), (4, 12, 8, np.float32, True)])
def test_lattice_models_through_the_api_vs_oracle(h, w, card, dt, sweep):
    """config-3 shaped models through `tree.propagate`: EVERY factor marginal against the oracle's propagate, on the min-fill tree
    (most cliques hold few factors or none) and on the column-sweep tree of SURVEY.md 8d (2-3 of 7 variables of a clique covered)"""
    factors, sizes, values = synthetic.lattice_mrf(h, w, card, dtype=dt)
    tree = jt.create_junction_tree(factors, sizes, order=synthetic.lattice_column_order(h, w) if sweep else None)
    ct = tree.clique_tree
    got = tree.propagate(values)
    plan = tree.plan("f32" if dt == np.float32 else "f64")
    st = plan.stats()
    assert st["n_unit_cliques"] > 0 and st["launch_mode"] == "flow" and st["flow_fallbacks"] == 0
    assert st["algorithmic_bytes"] < st["algorithmic_bytes_full"]
    want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes,
                            [np.asarray(v, dtype=np.float64) for v in values])
    for f, (g, w_) in enumerate(zip(got, want)):
        close(g, w_, RTOL32 if dt == np.float32 else RTOL64, "factor %d" % f)
    # again with new values: only the static tables (and the few stored tables) are formed again
    values2 = [v * dt(1.25) for v in values]
    got2 = tree.propagate(values2)
    scale = 1.25 ** len(factors)
    for f, (g, w_) in enumerate(zip(got2, want)):
        close(np.asarray(g) / scale, w_, 2 * RTOL32 if dt == np.float32 else 1e-10, "factor %d, second call" % f)
    # beliefs of cliques that keep no table are formed on demand and agree with the factor marginals they imply
    d = plan.describe()
    unit = [p["real"] for p in d["pnodes"] if p["unit"] and p["real"] >= 0 and p["stat"] >= 0][:3]
    z = plan.z()
    for c in unit:
        b = plan.belief(c)
        assert b.shape == tuple(sizes[v] for v in ct.maxcliques[c])
        assert abs(b.sum() - z) <= (1e-5 if dt == np.float32 else 1e-10) * z


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("level", [False, True])
def test_random_trees_with_random_covers_on_device(seed, level, monkeypatch):
    """explicit plans: cardinalities 1-8 (padded thread parts, rows at true cardinalities, mixed-radix rows beside unit cliques),
    every clique covered at random; every clique belief (unit cliques: on demand) and separator belief against the oracle"""
    rng = np.random.default_rng(900 + seed)
    if seed % 2 == 0:
        monkeypatch.setenv("JTP_UNIT_RATIO", "1")
    spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 14)))
    cover, pots = _with_cover(spec, pots, rng)
    full = [np.broadcast_to(np.asarray(p, dtype=np.float64), [spec["sizes"][v] for v in vs]).copy() for p, vs in zip(pots, spec["node_vars"])]
    want = oracle.beliefs_exact(spec["tree"], full, spec["node_vars"])
    for dtype, rtol in (("f64", RTOL64), ("f32", RTOL32)):
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, cover=cover, level_launches=level)
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, pots[c].astype(np.float32) if dtype == "f32" else pots[c])
        plan.propagate()
        ref = want
        if dtype == "f32":      # the oracle on the values the device holds
            ref = oracle.beliefs_exact(spec["tree"], [f.astype(np.float32).astype(np.float64) for f in full], spec["node_vars"])
        for n in range(len(spec["node_vars"])):
            close(plan.belief(n), ref[n], rtol, "seed %d node %d (%s)" % (seed, n, dtype))
        # marginals of unit cliques and of cliques that keep tables in one request list
        reqs = [(c, spec["node_vars"][c][:2]) for c in range(spec["n_cliques"])] + [(c, spec["node_vars"][c][-1:]) for c in range(spec["n_cliques"])]
        for (c, labs), m in zip(reqs, plan.marginals(reqs)):
            axes = tuple(i for i, v in enumerate(spec["node_vars"][c]) if v not in labs)
            w_ = np.asarray(ref[c]).sum(axis=axes)
            order = [v for v in spec["node_vars"][c] if v in labs]
            w_ = np.transpose(w_, [order.index(v) for v in labs])
            close(m, w_, rtol, "seed %d marginal of clique %d onto %r" % (seed, c, labs))
        plan.close()


@pytest.mark.parametrize("card,width,sep", [(3, 8, 4), (5, 6, 3), (6, 5, 2), (2, 14, 6), (4, 7, 3)])
def test_wide_unit_cliques_of_odd_cardinalities(card, width, sep, monkeypatch):
    monkeypatch.setenv("JTP_UNIT_RATIO", "1")
    spec = synthetic.wide_binary_tree(n_cliques=7, width=width, sep=sep, card=card, seed=card)
    base = synthetic.potentials_for(spec, seed=11)
    cover, pots = _with_cover(spec, base, np.random.default_rng(card), p_none=0.2)
    cover[3], pots[3] = list(spec["node_vars"][3]), base[3]
    full = [np.broadcast_to(np.asarray(p, dtype=np.float64), [spec["sizes"][v] for v in vs]).copy() for p, vs in zip(pots, spec["node_vars"])]
    want = oracle.beliefs_exact(spec["tree"], full, spec["node_vars"])
    for opts in ({}, {"level_launches": True}, {"flow_tickets": True}):
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", cover=cover, **opts)
        for c in range(7):
            plan.set_potential(c, pots[c])
        plan.propagate()
        for n in range(len(spec["node_vars"])):
            close(plan.belief(n), want[n], RTOL64, "node %d %r" % (n, opts))
        assert 0 < plan.stats()["n_unit_cliques"] < 7
        plan.close()


@pytest.mark.parametrize("n_children", [4, 7, 13])
def test_hub_of_many_children_with_a_static_table_on_device(n_children):
    """a tree with more than three children per node whose hub keeps no table (its factors cover two of its six variables)"""
    tree, pots, node_vars, sizes = star(n_children, card=2, seed=n_children)
    n = n_children + 1
    hub = node_vars[0]
    pots = list(pots)
    pots[0] = np.asarray(pots[0])[(slice(None), slice(None)) + (slice(0, 1),) * (len(hub) - 2)]
    cover = {c: list(node_vars[c]) for c in range(n)}
    cover[0] = hub[:2]
    full = [np.broadcast_to(np.asarray(p, dtype=np.float64), [sizes[v] for v in vs]).copy() for p, vs in zip(pots, node_vars)]
    want = oracle.beliefs_exact(tree, full, node_vars)
    for dtype, rtol in (("f64", RTOL64), ("f32", RTOL32)):
        plan = engine.Plan(tree, node_vars, sizes, dtype=dtype, cover=cover)
        for c in range(n):
            plan.set_potential(c, pots[c])
        plan.propagate()
        d = plan.describe()
        assert d["pnodes"][0]["unit"] and any(p["real"] < 0 and p["unit"] for p in d["pnodes"])
        for node in range(len(node_vars)):
            close(plan.belief(node), want[node], rtol, "node %d" % node)
        plan.close()


def test_evidence_and_unit_cliques(monkeypatch):
    """hard evidence applies to the entries of a unit clique as to a stored table's (jtp_set_evidence on plans with `cover`)"""
    factors, sizes, values = synthetic.lattice_mrf(3, 6, 4, dtype=np.float64)
    tree = jt.create_junction_tree(factors, sizes)
    ct = tree.clique_tree
    node_vars = [list(c) for c in ct.maxcliques] + [list(s) for s in tree.separators]
    names = sorted(sizes)
    sets = [{}, {names[0]: 1}, {names[3]: 2, names[7]: 0, names[11]: 3}, {v: 0 for v in names[:6]}]
    plan = engine.Plan(tree.tree, node_vars, sizes, dtype="f64", cover=tree.cover(), n_batch=len(sets), share_potentials=True)
    assert plan.stats()["n_unit_cliques"] > 0
    plan.stage_factors(factors, ct.factor_to_maxclique, values)
    for b, obs in enumerate(sets):
        plan.set_evidence(obs, batch=b)
    plan.propagate(0, len(sets))
    order = sorted({v for f in factors for v in f}, key=str)
    for b, obs in enumerate(sets):
        vals = [np.array(v, dtype=np.float64) for v in values]
        done = set()
        for i, f in enumerate(factors):           # the indicator of every observed variable goes into one factor that has it
            for ax, v in enumerate(f):
                if v in obs and v not in done:
                    done.add(v)
                    ind = np.zeros(sizes[v])
                    ind[obs[v]] = 1.0
                    vals[i] = vals[i] * ind.reshape([-1 if a == ax else 1 for a in range(len(f))])
        want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, vals)
        got = plan.factor_marginals(factors, ct.factor_to_maxclique, batch=b)
        for f, (g, w_) in enumerate(zip(got, want)):
            close(g, w_, 1e-10, "set %d factor %d" % (b, f))
    plan.close()


def test_errors_of_plans_with_cover():
    tree, node_vars, sizes = [0, (2, [1])], [[1, 2, 3], [2, 3, 4], [2, 3]], {1: 2, 2: 3, 3: 2, 4: 2}
    plan = engine.Plan(tree, node_vars, sizes, cover={0: [1], 1: []})
    with pytest.raises(ValueError, match="not depending on that variable"):
        plan.set_potential(0, np.ones((2, 3, 2)))
    with pytest.raises(ValueError, match="depending on none"):
        plan.set_potential(1, np.full((1, 1, 1), 2.0))
    plan.set_potential(1, np.ones((1, 1, 1)))
    plan.set_potential(0, np.array([0.25, 0.75]).reshape(2, 1, 1))
    with pytest.raises(ValueError, match="not depending on variable"):
        plan.set_potential_product(0, [np.ones((2, 3))], [[1, 2]])
    plan.set_potential_product(0, [np.array([0.25, 0.75]), np.array([[2.0], [4.0]])], [[1], [1, 2]])       # (a length-1 axis of an uncovered variable)
    plan.propagate()
    want = np.array([0.5, 3.0])[:, None, None] * np.ones((2, 3, 2)) * 2          # x the states of variable 4 summed out below
    close(plan.belief(0), want, 1e-12)
    close(plan.belief(1), np.full((3, 2, 2), 3.5), 1e-12)
    assert abs(plan.z() - 3.5 * 12) < 1e-9
    plan.close()


def test_evidence_sets_whose_multiset_plan_is_refused_say_so():
    """separators too large for the per-set LDS regions of a multi-set pass (a hub whose neighbours share nearly all of it): `propagate_evidence_sets` still
    answers - one pass per set over shared tables - and says so (a warning, `plan.evidence_mode`) instead of silently running
    up to eight times the table traffic (robustness)."""
    rng = np.random.default_rng(5)
    # a hub of twelve binary variables and four neighbours that each share eleven of them: a set's sub-boxes at the hub are
    # three or four tables of 2^11 doubles, beyond the 16 KiB region of a multi-set pass however the hub is cut
    hub = list(range(12))
    factors = [hub] + [[v for v in hub if v != i] + [12 + i] for i in range(4)]
    sizes = {v: 2 for v in range(16)}
    values = [rng.uniform(0.5, 1.5, [2] * len(f)) for f in factors]
    tree = jt.create_junction_tree(factors, sizes)
    sets = [{}, {0: 1, 14: 0}, {5: 1}]
    with pytest.warns(RuntimeWarning, match="one pass each"):
        res = tree.propagate_evidence_sets(values, sets)
    assert tree._memo["evidence_plan"].evidence_mode.startswith("one pass per evidence set")
    ct = tree.clique_tree
    for obs, got in zip(sets, res):
        vals = [v.copy() for v in values]
        done = set()
        for i, f in enumerate(factors):
            for ax, v in enumerate(f):
                if v in obs and v not in done:
                    done.add(v)
                    ind = np.zeros(2)
                    ind[obs[v]] = 1.0
                    vals[i] = vals[i] * ind.reshape([-1 if a == ax else 1 for a in range(len(f))])
        want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, vals)
        for g, w_ in zip(got, want):
            close(g, w_, 1e-10)
    # a tree whose multi-set plan is made says that too
    small = jt.create_junction_tree([["a", "b"], ["b", "c"]], {"a": 2, "b": 3, "c": 2})
    small.propagate_evidence_sets([np.ones((2, 3)), np.ones((3, 2))], [{}, {"a": 1}])
    assert small._memo["evidence_plan"].evidence_mode.startswith("multiset")


def test_evidence_free_subtrees_are_copied_not_recomputed(monkeypatch):
    """Multi-set plans (round 5): where no set of a group observes anything below a clique, the group copies the evidence-free
    upward message (group 0 of the launch) instead of streaming the table again.  Two sets that differ only by evidence in one
    leaf, a group without any evidence and a group with evidence everywhere: every separator belief and Z must be BIT-identical
    to the plan that computes everything (JTP_NO_EF_SHARE=1), and agree with the oracle."""
    spec = synthetic.wide_binary_tree(n_cliques=31, width=13, sep=6, card=2, seed=3)
    n, nb = spec["n_cliques"], 20
    base = synthetic.potentials_for(spec, seed=4, dtype=np.float32)
    leaf_var = [v for v in spec["node_vars"][n - 1] if v not in spec["node_vars"][spec["parent"][n - 1]]][0]
    labels = sorted(spec["sizes"])
    rng = np.random.default_rng(8)
    observed = [{}, {leaf_var: 1}] + [{} for _ in range(6)]                       # group 1: evidence in ONE leaf (set 1 only)
    observed += [{} for _ in range(8)]                                             # group 2: none at all
    observed += [{labels[i]: int(rng.integers(0, 2)) for i in rng.choice(len(labels), size=12, replace=False)} for _ in range(4)]      # group 3

    def run(share):
        if share:            # (by itself from eight groups of sets on: forced here for the three groups of the test)
            monkeypatch.delenv("JTP_NO_EF_SHARE", raising=False)
            monkeypatch.setenv("JTP_EF_SHARE", "1")
        else:
            monkeypatch.delenv("JTP_EF_SHARE", raising=False)
            monkeypatch.setenv("JTP_NO_EF_SHARE", "1")
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", n_batch=nb, multiset=True)
        for c in range(n):
            plan.set_potential(c, base[c])
        for b, obs in enumerate(observed):
            plan.set_evidence(obs, batch=b)
        out = []
        for rep in range(2):                     # (twice: both halves of the message arenas)
            plan.propagate(0, nb)
            out = [[plan.belief(s, batch=b) for s in range(n, 2 * n - 1)] + [np.array(plan.z(batch=b))] for b in range(nb)]
        st = plan.stats()
        some = [plan.belief(c, batch=b) for b in (0, 1, 9, 17) for c in (0, n // 2, n - 1)]
        plan.close()
        return out, st, some

    got, st, bel = run(True)
    want, st0, bel0 = run(False)
    for b in range(nb):
        for g, w_ in zip(got[b], want[b]):
            np.testing.assert_array_equal(g, w_, err_msg="evidence set %d" % b)
    for g, w_ in zip(bel, bel0):
        np.testing.assert_array_equal(g, w_)
    # (the engine's own table bytes: four groups of table passes - the evidence-free one included - less what was skipped, against three)
    assert st["algorithmic_bytes"] < st0["algorithmic_bytes"] * 1.1
    # sets 0 and 2..15 are evidence-free: identical to each other; set 1 differs from set 0 only along the leaf's path to the root
    for b in list(range(2, 16)):
        for g, w_ in zip(got[b], got[0]):
            np.testing.assert_array_equal(g, w_)
    from test_gpu_configs import _with_evidence
    for b in (1, 17):
        ref, z = oracle.beliefs_exact(spec["tree"], _with_evidence(spec, base, observed[b]), spec["node_vars"], return_z=True)
        assert abs(float(got[b][-1]) - z) <= RTOL32 * z
        for i, s in enumerate(range(n, 2 * n - 1)):
            close(got[b][i], ref[s], RTOL32, "set %d separator %d" % (b, s))
