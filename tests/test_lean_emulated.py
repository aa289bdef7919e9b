"""Cliques that keep no table (round 5), checked on the CPU: plans made with `cover` - which variables of each clique its
potential depends on - executed by tests/emulator.py and compared with the oracle and with the brute-force joint.

The reference leaves every variable of a clique that none of its assigned factors covers as a length-1 axis and never
materialises it (`junctiontree/junctiontree.py:52-61`, evaluate `:203-226`); rounds 1-4 of this engine stored every clique at
its full shape.  A clique described as depending on few of its variables is now a UNIT clique: no table, the product of its
factors a static table over the covered variables that every pass stages like one more incoming message, no belief table
(beliefs and marginals on demand).  The HIP kernels consuming the same task tables are checked on the GPU in
tests/test_gpu_parity.py."""
import pickle

import numpy as np
import pytest

import jt_oracle as oracle
from emulator import Emulator
import junctiontree_amd as jt
from junctiontree_amd import engine, synthetic
from test_planner_emulated import random_junction_tree, star


def emulate_factor_graph(factors, sizes, values, dtype="f64", order=None, **opts):
    """create_junction_tree -> plan with the tree's cover -> emulator -> factor marginals (host marginalize of the beliefs)"""
    tree = jt.create_junction_tree(factors, dict(sizes), order=order)
    ct = tree.clique_tree
    node_vars = [list(c) for c in ct.maxcliques] + [list(s) for s in tree.separators]
    plan = engine.Plan(tree.tree, node_vars, sizes, dtype=dtype, plan_only=True, cover=tree.cover(), **opts)
    desc = plan.describe()
    emu = Emulator(desc)
    pots = ct.evaluate(values)
    for c in plan.cliques:
        ids = [plan.var_id[lab] for lab in node_vars[c]]
        emu.set_potential(plan.abi_of[c], ids, [sizes[lab] for lab in node_vars[c]], pots[c])
    emu.propagate()
    level = emu.msg.copy()
    emu.propagate_flow()
    np.testing.assert_array_equal(level, emu.msg)
    bel = []
    for c in plan.cliques:
        ids = [plan.var_id[lab] for lab in node_vars[c]]
        bel.append(emu.belief(plan.abi_of[c], ids, [sizes[lab] for lab in node_vars[c]]))
    plan.close()
    return tree, ct.marginalize(bel), desc


def brute_force_marginals(factors, sizes, values):
    order = sorted({v for f in factors for v in f}, key=str)
    ax = {v: i for i, v in enumerate(order)}
    ops = []
    for f, val in zip(factors, values):
        ops += [np.asarray(val, dtype=np.float64), [ax[v] for v in f]]
    joint = np.einsum(*ops, list(range(len(order))), optimize=True)
    return [np.einsum(joint, list(range(len(order))), [ax[v] for v in f]) for f in factors]


@pytest.mark.parametrize("h,w,card,dtype", [(3, 6, 4, "f64"), (4, 8, 3, "f32"), (4, 7, 8, "f32"), (3, 9, 5, "f64"), (5, 6, 2, "f32")])
def test_lattices_against_the_oracle(h, w, card, dtype):
    """config-3 shaped models: pairwise factors on a lattice - most cliques hold few factors or none"""
    factors, sizes, values = synthetic.lattice_mrf(h, w, card)
    values = [np.asarray(v, dtype=np.float64) for v in values]
    tree, got, desc = emulate_factor_graph(factors, sizes, values, dtype=dtype)
    ct = tree.clique_tree
    want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, values)
    for g, w_ in zip(got, want):
        np.testing.assert_allclose(g, w_, rtol=1e-11, atol=1e-300)
    units = [p for p in desc["pnodes"] if p["unit"] and p["real"] >= 0]
    assert units and desc["lean"] == 1 and desc["has_unit"] == 1
    # a clique without factors has no static table; the others' are over exactly the covered variables
    for p in units:
        assert (p["stat"] >= 0) == bool(p["cover"])
        if p["stat"] >= 0:
            assert sorted(desc["statics"][p["stat"]]["vars"]) == sorted(p["cover"])
    # nothing of a unit clique is in the arenas: what is stored is the cliques that keep tables (plus the two shared rows)
    stored = sum(-(-p["phys_elems"] // 256) * 256 for p in desc["pnodes"] if not p["unit"])
    assert desc["arena_elems"] == (2 << desc["TB"]) + stored


def test_column_sweep_tree_of_a_lattice():
    """SURVEY.md 8d's tree of config 3 (one clique per eliminated variable, 2-3 of 7 variables covered): every clique a unit one"""
    factors, sizes, values = synthetic.lattice_mrf(4, 9, 4)
    values = [np.asarray(v, dtype=np.float64) for v in values]
    tree, got, desc = emulate_factor_graph(factors, sizes, values, dtype="f32", order=synthetic.lattice_column_order(4, 9))
    ct = tree.clique_tree
    want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, values)
    for g, w_ in zip(got, want):
        np.testing.assert_allclose(g, w_, rtol=1e-11, atol=1e-300)
    real = [p for p in desc["pnodes"] if p["real"] >= 0]
    assert sum(p["unit"] for p in real) >= len(real) - 4


@pytest.mark.parametrize("seed", range(24))
def test_random_factor_graphs_against_the_joint(seed, monkeypatch):
    rng = np.random.default_rng(500 + seed)
    if seed % 3 == 0:
        monkeypatch.setenv("JTP_UNIT_RATIO", "1")          # every partly covered clique becomes a unit clique
    nv = int(rng.integers(3, 11))
    names = ["v%d" % i for i in range(nv)] if seed % 2 else list(range(nv))
    while True:
        sizes = {v: int(rng.integers(1, 6)) for v in names}
        if np.prod([float(k) for k in sizes.values()]) <= 2e5:
            break
    factors = []
    for _ in range(int(rng.integers(1, 2 * nv))):
        k = int(rng.integers(1, min(3, nv) + 1))
        factors.append([names[i] for i in rng.choice(nv, size=k, replace=False)])
    if seed % 4 == 1:
        factors.append([names[i] for i in rng.choice(nv, size=min(nv, 6), replace=False)])         # one wide factor: wide cliques
    values = [rng.uniform(0.2, 1.0, [sizes[v] for v in f]) for f in factors]
    opts = [{}, {"block_log2": 10}, {"layout_policy": 2}, {"keep_root": True}][seed % 4]
    tree, got, desc = emulate_factor_graph(factors, sizes, values, dtype="f32" if seed % 2 else "f64", **opts)
    for g, w_ in zip(got, brute_force_marginals(factors, sizes, values)):
        np.testing.assert_allclose(g, w_, rtol=1e-11, atol=1e-300, err_msg="seed %d" % seed)


def _with_cover(spec, pots, rng, p_none=0.3):
    """a random cover per clique and potentials that depend on nothing else (the other axes: length 1)"""
    cover, out = {}, list(pots)
    for c in range(spec["n_cliques"]):
        vs = spec["node_vars"][c]
        mode = rng.random()
        keep = [] if mode < p_none else [v for v in vs if rng.random() < 0.4]
        cover[c] = keep
        index = tuple(slice(None) if v in keep else slice(0, 1) for v in vs)
        out[c] = np.ones([1] * len(vs)) if not keep else np.asarray(pots[c])[index]
    return cover, out


def check_tree(tree, pots, node_vars, sizes, cover, n_cliques, **opts):
    # (the oracle gets every table at its full shape: an axis that has length 1 in EVERY array would be a variable of one state to it)
    full = [np.broadcast_to(np.asarray(p, dtype=np.float64), [sizes[v] for v in vs]).copy() for p, vs in zip(pots, node_vars)]
    want = oracle.beliefs_exact(tree, full, node_vars)
    descs = []
    for dtype in ("f64", "f32"):
        plan = engine.Plan(tree, node_vars, sizes, dtype=dtype, plan_only=True, cover=cover, **opts)
        desc = plan.describe()
        emu = Emulator(desc)
        for c in plan.cliques:
            ids = [plan.var_id[lab] for lab in node_vars[c]]
            emu.set_potential(plan.abi_of[c], ids, [sizes[lab] for lab in node_vars[c]], pots[c])
        emu.propagate()
        level_bel, level_msg = emu.bel.copy(), emu.msg.copy()
        emu.propagate_flow()
        np.testing.assert_array_equal(emu.bel, level_bel)
        np.testing.assert_array_equal(emu.msg, level_msg)
        psep_of = {s["node"]: i for i, s in enumerate(desc["pseps"]) if s["node"] >= 0}
        for c in plan.cliques:
            ids = [plan.var_id[lab] for lab in node_vars[c]]
            got = emu.belief(plan.abi_of[c], ids, [sizes[lab] for lab in node_vars[c]])
            np.testing.assert_allclose(got, np.broadcast_to(want[c], got.shape), rtol=1e-11, atol=1e-13, err_msg="clique %d" % c)
        for s in plan.seps:
            ids = [plan.var_id[lab] for lab in node_vars[s]]
            got = emu.sep_belief(psep_of[plan.abi_of[s]], ids, [sizes[lab] for lab in node_vars[s]])
            np.testing.assert_allclose(got, np.broadcast_to(want[s], got.shape), rtol=1e-11, atol=1e-13, err_msg="separator %d" % s)
        plan.close()
        descs.append(desc)
    return descs


@pytest.mark.parametrize("seed", range(12))
def test_random_trees_with_random_covers(seed, monkeypatch):
    """junction trees of mixed cardinalities (1-8: padded thread parts, rows at true cardinalities, mixed-radix rows in the
    same plan), every clique covered at random - none, some or all of its variables"""
    rng = np.random.default_rng(900 + seed)
    if seed % 2 == 0:
        monkeypatch.setenv("JTP_UNIT_RATIO", "1")
    spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 14)))
    cover, pots = _with_cover(spec, pots, rng)
    opts = [{}, {"block_log2": 10}, {"layout_policy": 1}, {"keep_root": True}, {"layout_policy": 3, "block_log2": 11}, {"lds_budget": 2048}][seed % 6]
    check_tree(spec["tree"], pots, spec["node_vars"], spec["sizes"], cover, spec["n_cliques"], **opts)


@pytest.mark.parametrize("card,width,sep", [(3, 8, 4), (5, 6, 3), (6, 5, 2), (7, 5, 3), (2, 13, 6), (4, 7, 3)])
def test_wide_cliques_of_odd_cardinalities_partly_covered(card, width, sep, monkeypatch):
    """cliques of several rows: a unit clique's rows exist at the true cardinalities (JT_NO_ROW), its thread part is a padded
    bit field whose entries that exist the thread map marks - beside cliques that keep mixed-radix tables"""
    monkeypatch.setenv("JTP_UNIT_RATIO", "1")
    spec = synthetic.wide_binary_tree(n_cliques=7, width=width, sep=sep, card=card, seed=card)
    pots = synthetic.potentials_for(spec, seed=11)
    rng = np.random.default_rng(card)
    cover, pots = _with_cover(spec, pots, rng, p_none=0.2)
    cover[3] = list(spec["node_vars"][3])                     # one clique covered whole keeps its table
    pots[3] = synthetic.potentials_for(spec, seed=11)[3]
    descs = check_tree(spec["tree"], pots, spec["node_vars"], spec["sizes"], cover, 7)
    for desc in descs:
        kinds = [p["unit"] for p in desc["pnodes"] if p["real"] >= 0]
        assert 0 < sum(kinds) < len(kinds)
        for p in desc["pnodes"]:
            if p["unit"]:
                assert not p["tmix"] and p["tsplit"] < 0


@pytest.mark.parametrize("n_children", [4, 5, 7, 10])
@pytest.mark.parametrize("below", [False, True])
def test_hub_with_many_children_and_a_static_table(n_children, below):
    """more than three children per node: a unit hub with a static table takes its parent's message, the table and TWO children
    per pass (the root: the table and three) - the binarisation gives it virtual cliques (unit cliques themselves) for the rest"""
    tree, pots, node_vars, sizes = star(n_children, card=2, seed=n_children)
    n = n_children + 1
    hub = node_vars[0]
    pots = list(pots)
    pots[0] = np.asarray(pots[0])[(slice(None), slice(None)) + (slice(0, 1),) * (len(hub) - 2)]
    opts = {}
    if below:
        # the hub hangs below another clique (and stays there: JTP_KEEP_ROOT): node numbers 0..n-1 cliques, then separators
        fresh = max(sizes) + 1
        sizes[fresh] = 2
        cl, sp = node_vars[:n], node_vars[n:]
        node_vars = cl + [[hub[0], hub[3], fresh]] + sp + [[hub[0], hub[3]]]
        rng = np.random.default_rng(n_children)
        pots = pots[:n] + [rng.uniform(0.5, 1.5, (2, 2, 2))] + pots[n:] + [np.ones((2, 2))]
        # cliques 0..n-1 as before, the new top = n; the star's separators n+1..n+n_children, the top's = n+n_children+1
        tree = [n, (n + n_children + 1, [0] + [(n + 1 + i, [1 + i]) for i in range(n_children)])]
        opts = {"keep_root": True}
        n += 1
    cover = {c: list(node_vars[c]) for c in range(n)}
    cover[0] = hub[:2]
    descs = check_tree(tree, pots, node_vars, sizes, cover, n, **opts)
    for desc in descs:
        hubp = desc["pnodes"][0]
        assert hubp["unit"] and hubp["stat"] >= 0 and (hubp["parent"] >= 0) == below
        assert len(hubp["children"]) <= (2 if below else 3) and any(p["real"] < 0 and p["unit"] for p in desc["pnodes"])
        for t in desc["tasks"]:
            assert t["n_in"] <= 4


def test_descriptions_that_contradict_themselves_are_refused():
    tree, node_vars, sizes = [0, (2, [1])], [[1, 2, 3], [2, 3, 4], [2, 3]], {1: 2, 2: 3, 3: 2, 4: 2}
    with pytest.raises(ValueError, match="not one of its variables"):
        engine.Plan(tree, node_vars, sizes, plan_only=True, cover={0: [4], 1: []})
    plan = engine.Plan(tree, node_vars, sizes, plan_only=True, cover={0: [1], 1: []})
    d = plan.describe()
    assert [p["unit"] for p in d["pnodes"]] == [1, 1] and [p["stat"] for p in d["pnodes"]] == [0, -1]
    plan.close()
    # a clique covered (nearly) whole keeps its table
    plan = engine.Plan(tree, node_vars, sizes, plan_only=True, cover={0: [1, 2], 1: [2, 3, 4]})
    assert [p["unit"] for p in plan.describe()["pnodes"]] == [0, 0]
    plan.close()


def test_a_tree_pickles_after_use():
    """a JunctionTree that had been asked for its plan could not be pickled (a weak reference in its state)"""
    factors, sizes = [["a", "b"], ["b", "c"], ["c", "d"]], {"a": 2, "b": 3, "c": 2, "d": 2}
    tree = jt.create_junction_tree(factors, sizes)
    tree.cover()
    tree._memo["plan"] = ("mark", "key", lambda: None)       # what plan() leaves behind (a weak reference)
    again = pickle.loads(pickle.dumps(tree))
    assert again == tree and again._memo == {} and again.cover() == tree.cover()
    import copy
    assert copy.deepcopy(tree) == tree


def test_lean_records_say_what_the_task_records_say():
    """Round 6: a unit task of one outgoing message whose incoming tables have one copy each carries a LEAN record (`JtLean`,
    `jtp_internal.h`) that `jt_unit_lean` runs from instead of interpreting the task record.  The record is derived data: every field
    is checked here against the task record the emulator executes - the weights are the bit deposits of `free_pos`, the tables that
    depend on the element bits come first, the fold mask is the outgoing message's run, the workgroup records point at it."""
    import struct
    factors, sizes, _ = synthetic.lattice_mrf(6, 14, 8)
    tree = jt.create_junction_tree(factors, sizes)
    node_vars = [list(c) for c in tree.clique_tree.maxcliques] + [list(s) for s in tree.separators]
    seen = 0
    for dtype in ("f32", "f64"):
        plan = engine.Plan(tree.tree, node_vars, sizes, dtype=dtype, plan_only=True, cover=tree.cover())
        d = plan.describe()
        for t, tk in enumerate(d["tasks"]):
            eligible = (tk["kind"] == 0 and tk["unit"] and tk["mode"] == 0 and tk["n_out"] == 1 and tk["n_in"] <= 3 and tk["bel_off"] < 0
                        and all(m["npart"] == 1 for m in tk["in"]))
            assert (tk["lean_off"] > 0) == eligible, t
            if not eligible:
                assert "lean" not in tk
                continue
            seen += 1
            w = tk["lean"]
            assert len(w) == 176

            def msg(i):
                r = w[32 * i:32 * (i + 1)]
                off = struct.unpack("<q", struct.pack("<2i", r[0], r[1]))[0]
                return dict(off=off, nfree=r[2], lds_off=r[3], flags=r[4], src=r[5], e_w=r[6:8], w_lo=r[8:16], w_hi=r[16:24], t_w=r[24:32])
            tail = w[160:]
            n_in, n_e, total, rmask, red_e, red_lane, red_wave, settle, out_pstride, some_invalid = tail[:10]
            assert n_in == tk["n_in"] and total == tk["total"] and settle == tk["settle"]
            assert n_e == sum(1 for m in tk["in"] if m["e_dep"])
            order = [msg(i)["src"] for i in range(n_in)]
            assert sorted(order) == list(range(n_in))
            assert all(tk["in"][s]["e_dep"] for s in order[:n_e]) and not any(tk["in"][s]["e_dep"] for s in order[n_e:])
            for i in range(n_in + 1):
                lm = msg(i) if i < n_in else msg(4)
                m = tk["in"][lm["src"]] if i < n_in else tk["out"][0]
                assert (lm["off"], lm["nfree"], lm["lds_off"]) == (m["off"], m["nfree"], m["lds_off"])
                assert lm["flags"] == (1 if m["same_launch"] else 0) | (2 if m["fixed"] else 0)
                fp = list(m["free_pos"]) + [None] * 16
                assert lm["w_lo"] == [1 << fp[b] if b < m["nfree"] else 0 for b in range(8)]
                assert lm["w_hi"][:5] == [1 << fp[8 + b] if 8 + b < m["nfree"] else 0 for b in range(5)]
                assert lm["w_hi"][5:] == [m["npart"], m["pstride"], 0]
                assert lm["t_w"] == list(m["t_w"]) and lm["e_w"] == list(m["e_w"])
            out = tk["out"][0]
            assert (red_e, red_lane, red_wave, out_pstride) == (out["red_e"], out["red_lane"], out["red_wave"], out["pstride"])
            assert rmask == (1 << (tk["out_run"] & 0xff)) - 1 and msg(4)["src"] == 4
            assert struct.unpack("<q", struct.pack("<2i", tail[10], tail[11]))[0] == tk["tmap_off"]
            assert tail[14] == (1 if any(row[0] == 0xFFFFFFFF or row[0] == -1 for row in tk["itab"]) else 0)
        for b in d["blocks"]:
            assert bool(b[23] & 4) == (d["tasks"][b[0]]["lean_off"] > 0)
    assert seen > 50


def test_marginals_named_at_plan_creation_become_tasks_of_the_propagate(monkeypatch):
    """Round 6 (`jtp_tree_desc.fold_*`, `PlanBuilder::fold_marginals`): the factor marginals a plan is told about at creation - requests on
    cliques that keep no table - become lean tasks of the distribute phase, on the level of their clique, reading the final messages of
    the clique's neighbours and writing into a region of the message arena behind the separators'.  Structure only (the GPU tests
    compare the values with the oracle and with the read-out)."""
    # the planner's own choice first (`fold_marginals`: where at least half of the distribute levels leave the chip's resident slots
    # idle): the column-sweep tree of a lattice folds, its min-fill tree - level after level fuller than the chip - does not
    big = synthetic.lattice_mrf(6, 40, 8)
    for order, folds in ((None, False), (synthetic.lattice_column_order(6, 40), True)):
        bt = jt.create_junction_tree(big[0], big[1], order=order)
        bnv = [list(c) for c in bt.clique_tree.maxcliques] + [list(s) for s in bt.separators]
        bf = (tuple(bt.clique_tree.factor_to_maxclique), tuple(map(tuple, big[0])))
        bd = engine.Plan(bt.tree, bnv, big[1], dtype="f32", plan_only=True, cover=bt.cover(), fold=bf).describe()
        assert any(t["fold"] for t in bd["tasks"]) == folds
    monkeypatch.setenv("JTP_FOLD", "1")                      # from here on: wherever the plan's form allows
    factors, sizes, _ = synthetic.lattice_mrf(6, 14, 8)
    tree = jt.create_junction_tree(factors, sizes)
    ct = tree.clique_tree
    node_vars = [list(c) for c in ct.maxcliques] + [list(s) for s in tree.separators]
    fold = (tuple(ct.factor_to_maxclique), tuple(map(tuple, factors)))
    plain = engine.Plan(tree.tree, node_vars, sizes, dtype="f32", plan_only=True, cover=tree.cover()).describe()
    d = engine.Plan(tree.tree, node_vars, sizes, dtype="f32", plan_only=True, cover=tree.cover(), fold=fold).describe()
    folded = [(i, t) for i, t in enumerate(d["tasks"]) if t["fold"]]
    assert folded and not any(t["fold"] for t in plain["tasks"])
    assert d["msg_doubles"] > plain["msg_doubles"] and d["n_tasks"] == plain["n_tasks"] + len(folded)
    sep_end = max(max(s["up_off"] + (s["up_npart"] << s["nbits"]), s["dn_off"] + (s["dn_npart"] << s["nbits"]), s["up_roff"] + (s["up_rnpart"] << s["nbits"]),
                      s["dn_roff"] + (s["dn_rnpart"] << s["nbits"])) for s in d["pseps"])
    level_of = {}
    for L in d["launches"]:
        for t in L["tasks"]:
            level_of[t] = (L["phase"], L["level"])
    n_requests = 0
    for i, t in folded:
        p = d["pnodes"][t["pnode"]]
        assert p["unit"] and p["stat"] >= 0 and t["unit"] and t["kind"] == 0 and t["mode"] == 0 and t["lean_off"] > 0
        assert 1 <= t["n_out"] <= 3 and t["n_in"] <= 4 and t["bel_off"] < 0
        assert level_of[i][0] == 1 and level_of[i][1] >= p["depth"]       # on the clique's own level, or a later one that has room (JTP_FOLD_SLOTS)
        # inputs: the parent's final downward message, the static table, every child's final upward message
        expect = []
        if p["psep"] >= 0:
            expect.append(d["pseps"][p["psep"]]["dn_roff"])
        expect.append(d["statics"][p["stat"]]["off"])
        expect += [d["pseps"][d["pnodes"][k]["psep"]]["up_roff"] for k in p["children"]]
        assert [m["off"] for m in t["in"]] == expect and [m["fixed"] for m in t["in"]].count(1) == 1
        for m in t["out"]:
            assert m["off"] >= sep_end and m["off"] + m["npart"] * m["pstride"] <= d["msg_doubles"]
        n_requests += t["n_out"]
    # every request on a clique without a table is folded, three to a task at most
    assert n_requests == sum(1 for c in ct.factor_to_maxclique if d["pnodes"][c]["unit"])
    for b in d["blocks"]:
        assert bool(b[23] & 16) == bool(d["tasks"][b[0]]["fold"]) and (not (b[23] & 16) or (b[23] & 4))
    # a chain of latency-bound levels, a multi-set plan: no folded tasks (their launches are built without them)
    small = synthetic.lattice_mrf(5, 10, 4)
    st = jt.create_junction_tree(small[0], small[1])
    snv = [list(c) for c in st.clique_tree.maxcliques] + [list(s) for s in st.separators]
    sf = (tuple(st.clique_tree.factor_to_maxclique), tuple(map(tuple, small[0])))
    assert not any(t["fold"] for t in engine.Plan(st.tree, snv, small[1], dtype="f64", plan_only=True, cover=st.cover(), fold=sf).describe()["tasks"])


@pytest.mark.parametrize("h,w,card,dtype,sweep", [(3, 7, 4, "f64", False), (4, 8, 2, "f32", False), (3, 6, 4, "f32", True), (5, 6, 2, "f64", False)])
def test_folded_marginal_tasks_emulated(monkeypatch, h, w, card, dtype, sweep):
    """Round 6: the task tables of a plan with folded marginal tasks, executed on the CPU: per-level order, then the dataflow order in
    which every entry a workgroup reads - every partial copy of it - must have been written by an earlier workgroup; the partial copies
    of each folded output summed as `jt_marg_unpack` sums them, against the oracle's `propagate`; the messages the other tasks form are
    the plain plan's, bit for bit."""
    monkeypatch.setenv("JTP_TINY_LEVEL_ELEMS", "0")              # (small lattices plan as chains of latency-bound levels, which carry no folded tasks)
    monkeypatch.setenv("JTP_FOLD", "1")                          # (... and the planner folds by itself only where the levels leave slots idle)
    factors, sizes, values = synthetic.lattice_mrf(h, w, card)
    values = [np.asarray(v, dtype=np.float64) for v in values]
    tree = jt.create_junction_tree(factors, dict(sizes), order=synthetic.lattice_column_order(h, w) if sweep else None)
    ct = tree.clique_tree
    f2c = ct.factor_to_maxclique
    node_vars = [list(c) for c in ct.maxcliques] + [list(s) for s in tree.separators]
    plan = engine.Plan(tree.tree, node_vars, sizes, dtype=dtype, plan_only=True, cover=tree.cover(), fold=(tuple(f2c), tuple(map(tuple, factors))))
    plain = engine.Plan(tree.tree, node_vars, sizes, dtype=dtype, plan_only=True, cover=tree.cover())
    d, d0 = plan.describe(), plain.describe()
    fold_tasks = [i for i, t in enumerate(d["tasks"]) if t["fold"]]
    assert fold_tasks and d["tmix"] == 0
    # which request went to which task: requests on cliques without a table, grouped by clique in request order, three to a task
    groups, open_ = [], {}
    for i, c in enumerate(f2c):
        if not d["pnodes"][plan.abi_of[c]]["unit"]:
            continue
        if c not in open_ or len(groups[open_[c]]) >= 3:
            open_[c] = len(groups)
            groups.append([])
        groups[open_[c]].append(i)
    assert len(groups) == len(fold_tasks) and [len(g) for g in groups] == [d["tasks"][t]["n_out"] for t in fold_tasks]
    emu, emu0 = Emulator(d), Emulator(d0)
    pots = ct.evaluate(values)
    for c in plan.cliques:
        ids = [plan.var_id[lab] for lab in node_vars[c]]
        for e, pl in ((emu, plan), (emu0, plain)):
            e.set_potential(pl.abi_of[c], ids, [sizes[lab] for lab in node_vars[c]], pots[c])
    emu.propagate()
    level = emu.msg.copy()
    emu.propagate_flow()
    np.testing.assert_array_equal(level, emu.msg)
    emu0.propagate()
    np.testing.assert_array_equal(emu.msg[:d0["msg_doubles"]], emu0.msg[:d0["msg_doubles"]])      # the folded outputs lie behind everything else
    want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, f2c, factors, sizes, values)
    for t, grp in zip(fold_tasks, groups):
        assert d["tasks"][t]["pnode"] == plan.abi_of[f2c[grp[0]]]
        for j, i in enumerate(grp):
            got = emu.folded_marginal(t, j, [plan.var_id[lab] for lab in factors[i]], [sizes[lab] for lab in factors[i]])
            np.testing.assert_allclose(got, want[i], rtol=1e-11, atol=1e-300)
    plan.close()
    plain.close()
