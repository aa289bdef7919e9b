"""Planner checks on the CPU: the task tables `jtp_plan_create` emits (JTP_PLAN_ONLY, no GPU
needed) are executed by tests/emulator.py and compared with the oracle.  This pins layouts,
the F/A/R loop split, message buffers with partial copies, virtual-clique binarisation and
the level schedule; the HIP kernels consuming the same tables are checked on the GPU in
tests/test_gpu_parity.py."""
import numpy as np
import pytest

import jt_oracle as oracle
from conftest import as_tree
from emulator import Emulator
from junctiontree_amd import engine, synthetic


def emulate(tree, potentials, node_vars, sizes, dtype="f64", **opts):
    plan = engine.Plan(tree, node_vars, sizes, dtype=dtype, plan_only=True, **opts)
    desc = plan.describe()
    emu = Emulator(desc)
    for c in plan.cliques:
        ids = [plan.var_id[lab] for lab in node_vars[c]]
        emu.set_potential(plan.abi_of[c], ids, [sizes[lab] for lab in node_vars[c]], potentials[c])
    emu.propagate()
    level_bel, level_msg = emu.bel.copy(), emu.msg.copy()
    emu.propagate_flow()                       # the one-launch-per-phase order computes the same thing
    np.testing.assert_array_equal(emu.bel, level_bel)
    np.testing.assert_array_equal(emu.msg, level_msg)
    psep_of = {s["node"]: i for i, s in enumerate(desc["pseps"]) if s["node"] >= 0}
    out = {}
    for c in plan.cliques:
        ids = [plan.var_id[lab] for lab in node_vars[c]]
        out[c] = emu.belief(plan.abi_of[c], ids, [sizes[lab] for lab in node_vars[c]])
    for s in plan.seps:
        ids = [plan.var_id[lab] for lab in node_vars[s]]
        out[s] = emu.sep_belief(psep_of[plan.abi_of[s]], ids, [sizes[lab] for lab in node_vars[s]])
    plan.close()
    return out, desc


def check(tree, potentials, node_vars, sizes, **opts):
    want = oracle.beliefs_exact(tree, potentials, node_vars)
    stats = None
    for dtype in ("f64", "f32"):
        got, desc = emulate(tree, potentials, node_vars, sizes, dtype=dtype, **opts)
        for n, arr in got.items():
            full = np.broadcast_to(want[n], arr.shape)
            np.testing.assert_allclose(arr, full, rtol=1e-11, atol=1e-13, err_msg="node %d" % n)
        stats = desc
    return stats


def infer_sizes(potentials, node_vars):
    sizes = {}
    for p, labels in zip(potentials, node_vars):
        for n, lab in zip(np.shape(p), labels):
            sizes[lab] = max(sizes.get(lab, 1), n)
    return sizes


def test_reference_tree_cases(golden):
    g = golden("tree_cases.npz")
    for case in g.meta["cases"]:
        pots = g.arrs(case["potentials"])
        check(as_tree(case["tree"]), pots, case["variables"], infer_sizes(pots, case["variables"]))


def test_divergent_cases(golden):
    g = golden("divergent.npz")
    for case in g.meta["cases"]:
        if "tree" not in case:
            continue
        pots = g.arrs(case["potentials"])
        check(as_tree(case["tree"]), pots, case["variables"], infer_sizes(pots, case["variables"]))


@pytest.mark.parametrize("opts", [
    {}, {"block_log2": 10}, {"block_log2": 11, "lds_budget": 256}, {"layout_policy": 1},
    {"layout_policy": 1, "block_log2": 10, "lds_budget": 128}, {"keep_root": True},
    {"layout_policy": 2}, {"layout_policy": 3, "block_log2": 11}, {"layout_policy": 2, "block_log2": 10, "lds_budget": 256},
    {"layout_policy": 4, "block_log2": 10}, {"layout_policy": 4, "lds_budget": 2048}, {"layout_policy": 4, "lds_budget": 128, "block_log2": 11},
])
def test_synthetic_trees(opts):
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
    for spec in specs:
        pots = synthetic.potentials_for(spec, seed=21)
        desc = check(spec["tree"], pots, spec["node_vars"], spec["sizes"], **opts)
        assert desc["n_messages"] == 2 * (spec["n_cliques"] - 1)


@pytest.mark.parametrize("red_min", ["2", "4", "0"])
def test_reduce_tasks_sum_partial_copies(monkeypatch, red_min):
    """Messages written as >= JTP_REDUCE_MIN partial copies get a reduce task behind their producer
    and consumers read the sum (the default threshold of 8 only triggers on large cliques)."""
    monkeypatch.setenv("JTP_REDUCE_MIN", red_min)
    specs = [
        synthetic.wide_binary_tree(n_cliques=7, width=14, sep=7, card=2, seed=2),
        synthetic.random_tree(n_cliques=9, width=13, sep=5, card=2, seed=3),
        synthetic.chain_tree(n_cliques=4, card=16, width=3),
    ]
    n_red = 0
    for spec in specs:
        pots = synthetic.potentials_for(spec, seed=5)
        desc = check(spec["tree"], pots, spec["node_vars"], spec["sizes"], block_log2=10, layout_policy=3)
        n_red += sum(t["kind"] for t in desc["tasks"])
        for s in desc["pseps"]:
            for d in ("up", "dn"):
                reduced = s[d + "_red_task"] >= 0
                assert reduced == (red_min != "0" and s[d + "_npart"] >= int(red_min))
                assert s[d + "_rnpart"] == (1 if reduced else s[d + "_npart"])
    assert (n_red > 0) == (red_min != "0")


def star(n_children, card=2, seed=0):
    rng = np.random.default_rng(seed)
    hub = list(range(6))
    node_vars = [hub]
    nxt = 6
    for _ in range(n_children):
        shared = [hub[i] for i in rng.choice(6, size=int(rng.integers(1, 4)), replace=False)]
        node_vars.append(shared + [nxt, nxt + 1])
        nxt += 2
    n = n_children + 1
    seps = [[v for v in node_vars[c] if v in hub] for c in range(1, n)]
    tree = [0] + [(n + i, [1 + i]) for i in range(n_children)]
    sizes = {v: card for v in range(nxt)}
    node_vars = node_vars + seps
    pots = [rng.uniform(0.5, 1.5, [card] * len(vs)) for vs in node_vars[:n]]
    pots += [np.ones([card] * len(vs)) for vs in seps]
    return tree, pots, node_vars, sizes


@pytest.mark.parametrize("n_children", [4, 5, 7, 10, 13])
def test_many_children_use_virtual_cliques(n_children):
    tree, pots, node_vars, sizes = star(n_children, card=2, seed=n_children)
    desc = check(tree, pots, node_vars, sizes)
    assert any(p["real"] < 0 for p in desc["pnodes"])
    assert all(len(p["children"]) <= 3 for p in desc["pnodes"])
    tree, pots, node_vars, sizes = star(n_children, card=3, seed=n_children)
    check(tree, pots, node_vars, sizes, block_log2=10)


def test_broadcast_axes_and_card_one():
    # clique 1's potential is constant along variable 9 (length-1 axis), variable 4 has cardinality 1
    tree = [0, (2, [1])]
    node_vars = [[3, 5, 4], [5, 9], [5]]
    sizes = {3: 2, 5: 3, 4: 1, 9: 4}
    rng = np.random.default_rng(0)
    pots = [rng.standard_normal((2, 3, 1)), rng.standard_normal((3, 1)), np.ones(3)]
    got, _ = emulate(tree, pots, node_vars, sizes)
    full = [pots[0], np.broadcast_to(pots[1], (3, 4)).copy(), pots[2]]
    want = oracle.beliefs_exact(tree, full, node_vars)
    for n in range(3):
        np.testing.assert_allclose(got[n], want[n], rtol=1e-11, atol=1e-13)


def test_invalid_structures_are_rejected():
    with pytest.raises(ValueError):      # separator variable missing from the parent clique
        engine.Plan([0, (2, [1])], [[1, 2], [2, 3], [3]], {1: 2, 2: 2, 3: 2}, plan_only=True)
    with pytest.raises(ValueError):      # clique index used twice
        engine.Plan([0, (2, [0])], [[1, 2], [2, 3], [2]], {1: 2, 2: 2, 3: 2}, plan_only=True)
    with pytest.raises(KeyError):        # unknown variable size, like the reference (junctiontree.py:313)
        engine.Plan([0], [[1, 2]], {1: 2}, plan_only=True)
    with pytest.raises(ValueError):      # table too large for one plan
        engine.Plan([0], [list(range(40))], {v: 2 for v in range(40)}, plan_only=True)


def test_full_scale_configs_plan():
    """C4 and C2 at BASELINE.json's sizes: plan only (no tables are allocated)."""
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", plan_only=True)
    d = plan.describe()
    assert d["n_messages"] == 510 and d["arena_elems"] == (256 << 20) + 2048      # + the two shared rows (zero row, scratch row)
    assert abs(d["alg_bytes"] / 1e9 - 3.22) < 0.03
    assert d["max_lds"] <= 64 * 1024
    assert max(max(s["up_npart"], s["dn_npart"]) for s in d["pseps"]) <= 64
    plan.close()
    spec = synthetic.chain_tree(n_cliques=1000, card=64, width=3)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", plan_only=True)
    d = plan.describe()
    # re-rooted at the centre: 500 + 501 levels (+ launches of reduce tasks when launching per level)
    assert d["n_messages"] == 1998 and sum(1 for L in d["launches"] if L["variant"] != 16) == 1001
    assert len(d["segments"]) == 2
    plan.close()


def random_junction_tree(rng, n_cliques, max_width=5, cards=(1, 2, 2, 3, 4, 5, 8), max_children=5):
    """Random junction tree with mixed cardinalities, empty separators and wide fan-out.  Grown
    clique by clique (a child shares a random subset of its parent's variables and adds fresh
    ones), so the running-intersection property holds by construction."""
    sizes, clique_vars, parent, n_children = {}, [], [-1], [0]
    def fresh():
        v = len(sizes)
        sizes[v] = int(rng.choice(cards))
        return v
    clique_vars.append([fresh() for _ in range(int(rng.integers(1, max_width + 1)))])
    for c in range(1, n_cliques):
        cands = [p for p in range(c) if n_children[p] < max_children]
        p = int(rng.choice(cands))
        pv = clique_vars[p]
        k = int(rng.integers(0, min(len(pv), max_width - 1) + 1))
        shared = [pv[i] for i in rng.choice(len(pv), size=k, replace=False)] if k else []
        n_new = int(rng.integers(0 if shared else 1, max_width - len(shared) + 1))
        vars_c = shared + [fresh() for _ in range(n_new)]
        if not vars_c:
            vars_c = [fresh()]
        rng.shuffle(vars_c)
        clique_vars.append([int(v) for v in vars_c])
        parent.append(p)
        n_children[p] += 1
        n_children.append(0)
    spec = synthetic._assemble(parent, clique_vars, sizes)
    # scaled like the synthetic recipes so that Z stays O(1) (float32 storage cannot hold 1e38)
    pots = [rng.uniform(0.5, 1.5, [sizes[v] for v in vs]) * spec["scales"][c] for c, vs in enumerate(clique_vars)]
    pots += [np.ones([sizes[v] for v in vs]) for vs in spec["node_vars"][n_cliques:]]
    return spec, pots


@pytest.mark.parametrize("seed", range(6))
def test_random_trees_mixed_cardinalities(seed):
    rng = np.random.default_rng(100 + seed)
    spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 14)))
    opts = [{}, {"block_log2": 10}, {"layout_policy": 1}, {"keep_root": True}][seed % 4]
    check(spec["tree"], pots, spec["node_vars"], spec["sizes"], **opts)


def _comm_sequences(tree, node_vars, sizes, owner, world, **opts):
    """Per (src, dst): the separators in the order src issues its sends to dst, and the order dst
    issues its receives from src (an upward and a downward message of one separator are distinct)."""
    sends, recvs = {}, {}
    for rank in range(world):
        plan = engine.Plan(tree, node_vars, sizes, dtype="f64", plan_only=True, n_ranks=world, rank=rank,
                           owner=owner, **opts)
        for op in plan.describe()["comm"]:
            key = (rank, op["peer"]) if op["send"] else (op["peer"], rank)
            (sends if op["send"] else recvs).setdefault(key, []).append((op["psep"], op["up"], op["count"]))
        plan.close()
    return sends, recvs


def test_exchange_order_matches_on_both_sides_of_every_cut():
    """ncclSend/ncclRecv pair operations between two ranks in issue order: for every ordered pair of
    ranks the sender's sequence must equal the receiver's, whatever the clique numbering (round-1
    defect: receivers enumerated by parent, senders by child)."""
    # the advisor's reproducer: two same-level cut children whose parents are numbered the other way round
    tree = [0, (5, [1, (7, [4])]), (6, [2, (8, [3])])]
    node_vars = [[0, 1], [1, 2], [0, 3], [3, 4], [2, 5], [1], [0], [2], [3]]
    sizes = {v: 2 + (v % 3) for v in range(6)}          # unequal separator sizes
    sends, recvs = _comm_sequences(tree, node_vars, sizes, {0: 0, 1: 0, 2: 0, 3: 1, 4: 1}, 2)
    assert sends and sends == recvs
    rng = np.random.default_rng(11)
    from junctiontree_amd import partition
    for trial in range(12):
        base = (synthetic.random_tree(n_cliques=int(rng.integers(8, 40)), width=6, sep=3, card=2, seed=trial)
                if trial % 2 else synthetic.wide_binary_tree(n_cliques=int(rng.integers(8, 40)), width=6, sep=3, card=2, seed=trial))
        spec = synthetic.renumber(base, rng.permutation(base["n_cliques"]))
        world = int(rng.integers(2, 6))
        if trial % 3 == 0:       # arbitrary owners (parts need not be connected): many cuts per pair
            owner = [int(o) for o in rng.integers(0, world, spec["n_cliques"])]
        else:
            owner = partition.subtree_owners(spec["parent"], [1.0] * spec["n_cliques"], world)
        for opts in ({}, {"keep_root": True}):
            sends, recvs = _comm_sequences(spec["tree"], spec["node_vars"], spec["sizes"], owner, world, **opts)
            assert sends == recvs, "trial %d" % trial
        if trial % 3:            # the top of the partition replicated on every rank: upward messages go to all ranks
            owner = partition.subtree_owners(spec["parent"], [1.0] * spec["n_cliques"], world, replicate_top=True)
            sends, recvs = _comm_sequences(spec["tree"], spec["node_vars"], spec["sizes"], owner, world)
            assert sends == recvs and all(up for seq in sends.values() for _, up, _ in seq), "trial %d (replicated top)" % trial


def test_renumbered_tree_gives_the_same_beliefs():
    base = synthetic.random_tree(n_cliques=9, width=7, sep=3, card=2, seed=3)
    spec = synthetic.renumber(base, np.random.default_rng(0).permutation(9))
    pots = synthetic.potentials_for(spec, seed=4)
    check(spec["tree"], pots, spec["node_vars"], spec["sizes"])


@pytest.mark.parametrize("opts", [{}, {"block_log2": 10}, {"layout_policy": 3}, {"keep_root": True, "block_log2": 11}])
def test_multiset_plans_on_the_emulator(opts):
    """JTP_MULTISET plans (evidence sets share the tables, eight per pass): every downward message is its own
    marginalisation task (parent's message and the siblings' upward messages in, one message out), no belief
    table is written.  The emulator executes one set's tables; separator beliefs (up x down) against the oracle."""
    specs = [
        synthetic.chain_tree(n_cliques=5, card=4, width=3),
        synthetic.wide_binary_tree(n_cliques=7, width=12, sep=6, card=2, seed=1),
        synthetic.wide_binary_tree(n_cliques=7, width=5, sep=2, card=3, seed=4),
        synthetic.random_tree(n_cliques=9, width=11, sep=5, card=2, seed=3),
    ]
    tree, pots, node_vars, sizes = star(7, card=2, seed=2)          # virtual cliques
    specs.append({"tree": tree, "node_vars": node_vars, "sizes": sizes, "n_cliques": 8, "pots": pots})
    for spec in specs:
        pots = spec.get("pots") or synthetic.potentials_for(spec, seed=21)
        want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
        for dtype in ("f64", "f32"):
            plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, plan_only=True, multiset=True, n_batch=3, **opts)
            desc = plan.describe()
            assert desc["multiset"] == 1
            assert all(t["setb"] in (4096, 16384) and t["lds_bytes"] == 16384 + 8 * t["setb"] for t in desc["tasks"] if t["kind"] == 0)
            for p in desc["pnodes"]:
                assert len(p["down_tasks"]) == len(p["children"]) and p["distribute_task"] == -1
            emu = Emulator(desc)
            for c in plan.cliques:
                ids = [plan.var_id[lab] for lab in spec["node_vars"][c]]
                emu.set_potential(plan.abi_of[c], ids, [spec["sizes"][lab] for lab in spec["node_vars"][c]], pots[c])
            emu.propagate()
            level_msg = emu.msg.copy()
            emu.propagate_flow()
            np.testing.assert_array_equal(emu.msg, level_msg)
            psep_of = {s["node"]: i for i, s in enumerate(desc["pseps"]) if s["node"] >= 0}
            for sn in plan.seps:
                ids = [plan.var_id[lab] for lab in spec["node_vars"][sn]]
                got = emu.sep_belief(psep_of[plan.abi_of[sn]], ids, [spec["sizes"][lab] for lab in spec["node_vars"][sn]])
                np.testing.assert_allclose(got, want[sn], rtol=1e-11, atol=1e-13, err_msg="separator %d" % sn)
            plan.close()


@pytest.mark.parametrize("card,width,sep", [(3, 8, 4), (5, 6, 3), (6, 5, 2), (7, 5, 3), (3, 9, 5)])
def test_rows_above_the_thread_part_are_stored_at_true_cardinalities(card, width, sep):
    """Mixed-radix rows (round 2): a variable wholly above the thread part of a table counts its true cardinality,
    padding index bits store nothing; rows that do not exist read the arena's zero row.  Checks results against
    the oracle and the arena against the padded (round-1) layout."""
    spec = synthetic.wide_binary_tree(n_cliques=7, width=width, sep=sep, card=card, seed=card)
    pots = synthetic.potentials_for(spec, seed=11)
    for opts in ({}, {"block_log2": 10}, {"multiset": True, "n_batch": 2}):
        want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
        sizes = {}
        for compact in (True, False):
            import os
            os.environ["JTP_NO_COMPACT"] = "0" if compact else "1"
            try:
                plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", plan_only=True, **opts)
            finally:
                del os.environ["JTP_NO_COMPACT"]
            desc = plan.describe()
            assert desc["compact"] == (1 if compact else 0)
            sizes[compact] = desc["arena_elems"]
            # round 5: the partial copies of chunks whose digits do not exist are zeroed once per arena (init_blocks), not once per propagate
            assert all((b[23] & 1) == 0 for b in desc["blocks"])
            assert all((b[23] & 1) == 1 for b in desc["init_blocks"])
            if opts.get("block_log2") == 10 and compact:
                assert desc["init_blocks"]
            # round 5, compact mixed-radix rows: all of a plan's table-keeping cliques or none; a clique's list names exactly the logical
            # threads that own an entry (ascending), padded to 128 with one that owns none
            VEC = desc["VEC"]
            for p_ in desc["pnodes"]:
                if desc["tmix_compact"] and not p_["unit"]:
                    owners = [t for t in range(256) if any(x >= 0 for x in p_["tmap"][t * VEC:(t + 1) * VEC])]
                    assert len(p_["vmap"]) == 128 and p_["vmap"][:len(owners)] == owners and len(owners) <= 128
                    assert all(v not in owners for v in p_["vmap"][len(owners):])
                else:
                    assert not p_.get("vmap")
            assert {t["vgroups"] for t in desc["tasks"] if t["kind"] == 0 and not t["unit"]} <= ({2} if desc["tmix_compact"] else {0})
            if compact and not opts.get("multiset") and card in (3, 5):
                assert desc["tmix_compact"] == 1
            emu = Emulator(desc)
            for c in plan.cliques:
                ids = [plan.var_id[lab] for lab in spec["node_vars"][c]]
                emu.set_potential(plan.abi_of[c], ids, [spec["sizes"][lab] for lab in spec["node_vars"][c]], pots[c])
            emu.propagate()
            keep = emu.msg.copy()
            emu.propagate_flow()
            np.testing.assert_array_equal(emu.msg, keep)
            psep_of = {s["node"]: i for i, s in enumerate(desc["pseps"]) if s["node"] >= 0}
            for sn in plan.seps:
                ids = [plan.var_id[lab] for lab in spec["node_vars"][sn]]
                got = emu.sep_belief(psep_of[plan.abi_of[sn]], ids, [spec["sizes"][lab] for lab in spec["node_vars"][sn]])
                np.testing.assert_allclose(got, want[sn], rtol=1e-11, atol=1e-13)
            if not opts.get("multiset"):
                for c in plan.cliques:
                    ids = [plan.var_id[lab] for lab in spec["node_vars"][c]]
                    got = emu.belief(plan.abi_of[c], ids, [spec["sizes"][lab] for lab in spec["node_vars"][c]])
                    np.testing.assert_allclose(got, want[c], rtol=1e-11, atol=1e-13)
            if compact:
                assert any(p["group_mask"] for p in desc["pnodes"])           # some variable is stored at its true cardinality
            plan.close()
        host = 7 * card ** width
        assert sizes[True] < sizes[False]
        print("card %d width %d: host %d, compact arena %d (%.2fx), padded arena %d (%.2fx)" % (
            card, width, host, sizes[True], sizes[True] / host, sizes[False], sizes[False] / host))


def wide_node_case():
    """A 3-clique tree whose root clique lists 33 variables, 30 of them with one state."""
    rng = np.random.default_rng(3)
    ones = ["u%d" % i for i in range(30)]
    sizes = dict({lab: 1 for lab in ones}, a=2, b=3, c=2, d=4, e=2)
    node_vars = [ones[:20] + ["a", "b", "c"] + ones[20:], ["b", "c", "d"] + ones[:5], ["c", "e"], ["b", "c"] + ones[3:5], ["c"]]
    tree = [0, (3, [1]), (4, [2])]
    pots = [rng.random([sizes[lab] for lab in node_vars[n]]) for n in range(3)] + [np.ones([sizes[lab] for lab in node_vars[n]]) for n in (3, 4)]
    return tree, node_vars, sizes, pots


def test_more_than_32_variables_on_a_node_when_the_rest_have_one_state():
    """The reference takes up to 52 labels per einsum (sum_product.py:34-41); a dense table is at most 2^31 entries,
    so beyond 31 variables the rest have cardinality 1.  engine.Plan keeps such variables on the host: the device
    plan is that of the tree without them (the GPU side: test_gpu_parity.py, same case)."""
    tree, node_vars, sizes, pots = wide_node_case()
    assert len(node_vars[0]) == 33
    plan = engine.Plan(tree, node_vars, sizes, plan_only=True)
    d = plan.describe()
    assert sorted(plan.var_id) == ["a", "b", "c", "d", "e"] and len(plan._trivial) == 30
    assert plan.node_shape[0] == tuple(sizes[lab] for lab in node_vars[0])
    assert plan._drop_trivial(pots[0], node_vars[0]).shape == (2, 3, 2)
    with pytest.raises(ValueError):
        plan._drop_trivial(np.ones((2,) + pots[0].shape[1:]), node_vars[0])      # a one-state variable's axis must have length 1
    plan.close()
    # the same tree with the one-state variables given two states is beyond the C ABI's limit, and says so
    with pytest.raises(Exception, match="variables"):
        engine.Plan(tree, node_vars, {lab: max(2, k) for lab, k in sizes.items()}, plan_only=True)


def test_float32_layouts_that_do_not_fit_lds_are_planned_with_float64_tables():
    """A clique of few rows with four or more neighbours whose separators are nearly the whole clique cannot be planned with
    1024-element (float32) rows (tools/gpu_fuzz.py, FUZZ_BIG, seed 92488).  `jtp_plan_create` makes the plan with float64
    tables then (`jtp_stats.storage_dtype`; `engine.Plan` warns) - for a C caller as for the Python layer - and the task
    tables it emits compute the oracle's beliefs."""
    from junctiontree_amd import _capi
    rng = np.random.default_rng(92488)
    while True:
        spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 12)), max_width=9, cards=(2, 3, 3, 4, 5, 6, 7))
        if 1 << 14 <= max(p.size for p in pots) <= 1 << 22 and sum(p.size for p in pots) <= 1 << 24:
            break
    with pytest.warns(RuntimeWarning, match="float64 tables made"):
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", plan_only=True)
    assert plan.requested_dtype == _capi.JTP_F32 and plan.dtype == _capi.JTP_F64 and plan.describe()["dtype"] == _capi.JTP_F64
    plan.close()
    want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
    with pytest.warns(RuntimeWarning):
        got, _ = emulate(spec["tree"], pots, spec["node_vars"], spec["sizes"], dtype="f32")
    for n, arr in got.items():
        np.testing.assert_allclose(arr, np.broadcast_to(want[n], arr.shape), rtol=1e-11, atol=1e-13, err_msg="node %d" % n)
    # a multi-set request is NOT widened: its caller chooses between the multi-set pass and one pass per set
    with pytest.raises(ValueError):
        engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", plan_only=True, n_batch=4, multiset=True, lds_budget=64)
