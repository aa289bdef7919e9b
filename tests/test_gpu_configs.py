"""BASELINE.json configs at their STATED shapes on a real MI355X, against the oracle (tolerances as in
test_gpu_parity.close: float64 storage 1e-11, float32 storage 1e-6, elementwise):

  configs[1]  chain of 1000 cliques, width 3, cardinality 64, float64 - every belief vs oracle.beliefs_exact
  configs[2]  grid MRF, cardinality 8, float32: a 6 x 12 lattice (cliques up to 8^8 entries) vs the oracle, and
              the full 6 x 167 lattice of SURVEY.md 8d (1002 variables, 9 GiB of tables) through properties
  configs[4]  evidence sets on the full width-20 tree (256 x 2^20 float32, tables shared by the sets): Z and
              sampled beliefs of four sets vs the oracle with indicator-multiplied potentials
(configs[3], the width-20 tree itself, is test_gpu_parity.test_full_size_c4_properties and bench.py.)"""
import numpy as np
import pytest

import jt_oracle as oracle
import junctiontree_amd as jt
from junctiontree_amd import engine, synthetic
from test_gpu_parity import RTOL32, RTOL64, close

pytestmark = pytest.mark.gpu


def lattice(h, w, card, seed=0, dtype=np.float32):
    return synthetic.lattice_mrf(h, w, card, seed, dtype)


def test_config2_chain_full_length_vs_oracle():
    """configs[1] at full length: 1000 x 64^3 doubles (2 GiB of tables; re-rooted: 500 dependent levels per
    phase inside ONE dataflow launch each, poll back-off and all), every clique and separator belief."""
    spec = synthetic.chain_tree(n_cliques=1000, card=64, width=3)
    n = spec["n_cliques"]
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64")
    plan.fill_synthetic(3, spec["scales"])                   # same numbers as synthetic.potentials_for(seed=3)
    for _ in range(2):                                       # twice: both halves of the message arena
        plan.propagate()
    st = plan.stats()
    assert st["n_launches"] == 2 and st["flow_fallbacks"] == 0
    pots = synthetic.potentials_for(spec, seed=3)
    want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
    del pots
    assert abs(plan.z() - z) <= RTOL64 * abs(z)
    for node in range(len(spec["node_vars"])):
        close(plan.belief(node), want[node], rtol=RTOL64, what="node %d" % node)
        want[node] = None
    plan.close()


def test_config3_lattice_card8_f32_vs_oracle():
    """configs[2] at its stated clique shape, reduced length: 6 x 12 lattice, cardinality 8, float32,
    pairwise factors, own junction-tree builder (cliques up to width 8 = 8^8 entries, separators up to 8^7),
    through the public API; every factor marginal against the oracle's propagate."""
    factors, sizes, values = lattice(6, 12, 8)
    tree = jt.create_junction_tree(factors, sizes)
    ct = tree.clique_tree
    assert max(len(c) for c in ct.maxcliques) >= 7
    out = tree.propagate(values)
    plan = tree.plan("f32")
    d = plan.describe()
    st = plan.stats()
    # sub-boxes beyond 64 KiB make the engine launch per level (jtp_plan_create); otherwise one launch per phase, or - plans made
    # mostly of cliques that keep no table, round 5 - one launch for both phases
    assert (st["n_launches"] <= 2) == (d["max_lds"] <= 64 * 1024), (st["n_launches"], d["max_lds"])
    want = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, values)
    assert len(out) == len(values)
    for i, (o, w, v) in enumerate(zip(out, want, values)):
        assert o.shape == v.shape and o.dtype == np.float64
        close(o, w, rtol=RTOL32, what="factor %d %r" % (i, factors[i]))
    z = plan.z()
    assert abs(z - float(np.sum(want[0]))) <= RTOL32 * z
    # the same tables under the "traffic first" bit order (what the cost-model search replaced on such cliques):
    # larger sub-boxes, more partial copies - a different schedule of the same sums
    node_vars = [list(c) for c in ct.maxcliques] + [list(s) for s in tree.separators]
    other = engine.Plan(tree.tree, node_vars, sizes, dtype="f32", layout_policy=2)
    assert {p["layout"] for p in other.describe()["pnodes"]} == {2} and 4 in {p["layout"] for p in d["pnodes"]}
    for c, psi in enumerate(ct.evaluate(values)):
        other.set_potential(c, psi)
    other.propagate()
    assert abs(other.z() - z) <= RTOL32 * z
    for c in (0, len(ct.maxcliques) // 2, len(ct.maxcliques) - 1):
        close(other.belief(c), plan.belief(c), rtol=RTOL32, what="clique %d, layout policy 2 vs 4" % c)
    other.close()


def test_config3_full_lattice_properties():
    """configs[2] as restated in SURVEY.md 8d: 6 x 167 lattice (1002 variables, 1831 pairwise factors of
    cardinality 8, float32): every factor marginal sums to Z, and the single-variable marginals implied by different factors
    agree (calibration across the whole tree).  Round 5: the 9 GiB of full-shape clique tables are no longer on the device - a
    clique stores what its factors cover (365 of the 878 cliques hold no factor at all), as the reference's evaluate leaves it
    (junctiontree.py:52-61)."""
    factors, sizes, values = lattice(6, 167, 8)
    tree = jt.create_junction_tree(factors, sizes)
    assert len(sizes) == 1002 and len(factors) == 1831
    assert max(len(c) for c in tree.clique_tree.maxcliques) <= 9
    out = tree.propagate(values)
    plan = tree.plan("f32")
    d, st = plan.describe(), plan.stats()
    full = sum(int(np.prod([sizes[v] for v in c])) for c in tree.clique_tree.maxcliques) * 4
    assert full > 9 * 2 ** 30 and d["arena_elems"] * 4 + st["fixed_bytes"] <= 0.3 * 2 ** 30          # what the tables WOULD take; what they take
    assert st["n_unit_cliques"] >= 365 and st["algorithmic_bytes"] < 0.2 * st["algorithmic_bytes_full"]
    # (round 2: the searched layouts keep every sub-box set below 64 KiB, so this config runs as two dataflow
    #  launches; with larger sub-boxes the engine launches per level)
    assert (plan.stats()["n_launches"] <= 2) == (plan.describe()["max_lds"] <= 64 * 1024)
    z = plan.z()
    assert np.isfinite(z) and z > 0
    sums = np.array([o.sum() for o in out])
    assert np.max(np.abs(sums - z)) <= 5e-6 * z
    marg, worst = {}, 0.0
    for f, o in zip(factors, out):
        for ax, v in enumerate(f):
            m = o.sum(axis=1 - ax)
            if v in marg:
                worst = max(worst, float(np.max(np.abs(m - marg[v]) / marg[v])))
            else:
                marg[v] = m
    assert worst < 5e-6, worst
    engine.clear_plan_cache()


def _with_evidence(spec, base, observed):
    pots = [np.asarray(p, dtype=np.float64) for p in base]
    n = spec["n_cliques"]
    for var, state in observed.items():
        host = next(c for c in range(n) if var in spec["node_vars"][c])
        ind = np.zeros(spec["sizes"][var])
        ind[state] = 1.0
        shape = [1] * pots[host].ndim
        shape[spec["node_vars"][host].index(var)] = spec["sizes"][var]
        pots[host] = pots[host] * ind.reshape(shape)
    return pots


def test_config5_evidence_sets_on_the_width20_tree():
    """configs[4] on one GPU: 32 evidence sets (16 observed variables each) on the full
    width-20 tree with ONE copy of the 256 x 2^20 float32 tables.  Four of the sets against the oracle run on
    indicator-multiplied potentials (Z, sampled clique beliefs and the separators next to them); every set
    through its own consistency (sampled clique beliefs sum to the set's Z)."""
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    n, nb = spec["n_cliques"], 32
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", n_batch=nb, share_potentials=True)
    plan.fill_synthetic(1, spec["scales"])
    labels = sorted(spec["sizes"])
    observed = []
    for b in range(nb):
        rng = np.random.default_rng(1000 + b)
        observed.append({labels[i]: int(rng.integers(0, 2)) for i in rng.choice(len(labels), size=16, replace=False)})
        plan.set_evidence(observed[b], batch=b)
    plan.propagate(0, nb)
    zs = [plan.z(batch=b) for b in range(nb)]
    assert all(np.isfinite(z) and z > 0 for z in zs)
    rng = np.random.default_rng(0)
    for b in range(nb):
        for c in rng.choice(n, size=3, replace=False):
            assert abs(plan.marginal(int(c), [], batch=b) - zs[b]) <= 2e-6 * zs[b], (b, c)
    base = synthetic.potentials_for(spec, seed=1, dtype=np.float32)
    for b in (0, 9, 18, 31):
        want, z = oracle.beliefs_exact(spec["tree"], _with_evidence(spec, base, observed[b]), spec["node_vars"], return_z=True)
        assert abs(zs[b] - z) <= RTOL32 * z, b
        for c in [0, 1, n - 1] + [int(x) for x in rng.choice(np.arange(2, n - 1), size=5, replace=False)]:
            close(plan.belief(c, batch=b), want[c], rtol=RTOL32, what="set %d clique %d" % (b, c))
            if c > 0:
                close(plan.belief(n + c - 1, batch=b), want[n + c - 1], rtol=RTOL32, what="set %d separator of %d" % (b, c))
        del want
    plan.close()


def test_config5_multiset_64_sets_on_the_width20_tree():
    """configs[4] through the multi-set kernel (jt_multi_flow: one pass over a table serves a group of evidence sets) at
    CONFIG SCALE: 64 evidence sets of 16 observed variables on the full width-20 tree, one copy of the 256 x 2^20
    float32 tables.  Five sets against the oracle on indicator-multiplied potentials - Z, sampled clique beliefs (formed
    on demand: a multi-set plan keeps no belief tables) and the separator beliefs next to them; EVERY set through its
    own consistency: sampled clique marginals sum to the set's Z, and the set's Z is P(evidence) * Z of the evidence-free
    tree, i.e. at most that."""
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    n, nb = spec["n_cliques"], 64
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", n_batch=nb, multiset=True)
    assert plan.describe()["multiset"] == 1
    plan.fill_synthetic(1, spec["scales"])
    labels = sorted(spec["sizes"])
    observed = []
    for b in range(nb):
        rng = np.random.default_rng(1000 + b)
        observed.append({labels[i]: int(rng.integers(0, 2)) for i in rng.choice(len(labels), size=16, replace=False)})
        plan.set_evidence(observed[b], batch=b)
    for _ in range(2):                                       # both halves of the message arenas
        plan.propagate()
    st = plan.stats()
    assert st["flow_fallbacks"] == 0 and st["launch_mode"] == "flow" and st["n_launches"] == 2
    zs = [plan.z(batch=b) for b in range(nb)]
    z_free = 1.0058528272803358                              # bench.py: Z of the evidence-free tree (oracle)
    assert all(np.isfinite(z) and 0 < z < z_free for z in zs)
    rng = np.random.default_rng(0)
    for b in range(nb):
        for c in rng.choice(n, size=2, replace=False):
            assert abs(plan.marginal(int(c), [], batch=b) - zs[b]) <= 2e-6 * zs[b], (b, c)
    base = synthetic.potentials_for(spec, seed=1, dtype=np.float32)
    for b in (0, 7, 21, 40, 63):                              # (sets of different groups of the launch)
        want, z = oracle.beliefs_exact(spec["tree"], _with_evidence(spec, base, observed[b]), spec["node_vars"], return_z=True)
        assert abs(zs[b] - z) <= RTOL32 * z, b
        for c in [0, 1, n - 1] + [int(x) for x in rng.choice(np.arange(2, n - 1), size=4, replace=False)]:
            close(plan.belief(c, batch=b), want[c], rtol=RTOL32, what="set %d clique %d" % (b, c))
            if c > 0:
                close(plan.belief(n + c - 1, batch=b), want[n + c - 1], rtol=RTOL32, what="set %d separator of %d" % (b, c))
        del want
    plan.close()


def test_config3_lattice_6x60_every_factor_marginal_vs_oracle():
    """configs[2] at its stated clique shapes over 60 of the 167 columns (360 variables, 649 pairwise factors, float32,
    cliques up to 8^8 entries): EVERY factor marginal element by element against the oracle's `propagate` (VERDICT r3, weak
    item 1: the full lattice is checked through properties, the oracle comparison ran on 6 x 12 only), on the min-fill tree
    and on the column-sweep tree of SURVEY.md 8d (`create_junction_tree(..., order=...)`), which must agree with each other
    too."""
    factors, sizes, values = lattice(6, 60, 8)
    trees = {"min-fill": jt.create_junction_tree(factors, sizes),
             "column sweep": jt.create_junction_tree(factors, sizes, order=synthetic.lattice_column_order(6, 60))}
    widths = {k: sorted({len(c) for c in t.clique_tree.maxcliques}) for k, t in trees.items()}
    assert max(widths["column sweep"]) == 7 and len(trees["column sweep"].clique_tree.maxcliques) == 360 - 6
    ct = trees["min-fill"].clique_tree
    want = oracle.propagate(trees["min-fill"].tree, trees["min-fill"].separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, values)
    outs = {}
    for name, tree in trees.items():
        out = tree.propagate(values)
        assert len(out) == len(values)
        for i, (o, w) in enumerate(zip(out, want)):
            assert o.shape == w.shape and o.dtype == np.float64
            close(o, w, rtol=RTOL32, what="%s tree, factor %d %r" % (name, i, factors[i]))
        assert tree.plan("f32").stats()["flow_fallbacks"] == 0
        outs[name] = out
        engine.clear_plan_cache()
    for a, b in zip(outs["min-fill"], outs["column sweep"]):
        close(a, b, rtol=2 * RTOL32)


def test_config5_all_512_sets_on_one_device():
    """configs[4] at its stated COUNT on one device: 512 evidence sets (16 observed variables each) over the shared tables of
    the full width-20 tree, one multi-set plan (64 groups of eight sets per pass over a table).  Eight sets - one per
    64 - against the oracle on indicator-multiplied potentials (Z and two clique beliefs each); every set through its own
    consistency (a clique marginal sums to the set's Z, which is below the evidence-free Z)."""
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    n, nb = spec["n_cliques"], 512
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", n_batch=nb, multiset=True)
    plan.fill_synthetic(1, spec["scales"])
    labels = sorted(spec["sizes"])
    observed = []
    for b in range(nb):
        rng = np.random.default_rng(1000 + b)
        observed.append({labels[i]: int(rng.integers(0, 2)) for i in rng.choice(len(labels), size=16, replace=False)})
        plan.set_evidence(observed[b], batch=b)
    for _ in range(2):
        plan.propagate()
    st = plan.stats()
    assert st["flow_fallbacks"] == 0 and st["n_launches"] == 2
    zs = [plan.z(batch=b) for b in range(nb)]
    assert all(np.isfinite(z) and 0 < z < 1.0058528272803358 for z in zs)
    rng = np.random.default_rng(1)
    for b in range(nb):
        c = int(rng.integers(0, n))
        assert abs(plan.marginal(c, [], batch=b) - zs[b]) <= 2e-6 * zs[b], (b, c)
    base = synthetic.potentials_for(spec, seed=1, dtype=np.float32)
    for b in range(5, nb, 64):                                # eight sets, one in every 64
        want, z = oracle.beliefs_exact(spec["tree"], _with_evidence(spec, base, observed[b]), spec["node_vars"], return_z=True)
        assert abs(zs[b] - z) <= RTOL32 * z, b
        for c in (0, int(rng.integers(1, n))):
            close(plan.belief(c, batch=b), want[c], rtol=RTOL32, what="set %d clique %d" % (b, c))
        del want
    plan.close()
