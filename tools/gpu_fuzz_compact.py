"""One-off: trees whose cliques all store their thread part at true cardinalities (one cardinality per tree: 3, 5, 9, 10, 11 - and 6, 7, 12 for the
one-row and bit-field forms), random shapes and planner options: beliefs, Z, marginals and a second propagate under hard evidence against the oracle;
how many of the plans ran the compact kernels (two rows per step) is reported.      python tools/gpu_fuzz_compact.py [N] [first seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "junction-tree_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import jt_oracle as oracle
from test_gpu_parity import close, RTOL32, RTOL64, _indicator_potentials
from junctiontree_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
opts_all = [{}, {}, {"level_launches": True}, {"flow_tickets": True}, {"block_log2": 12}, {"layout_policy": 2}, {"keep_root": True}, {"block_log2": 10}]
kinds = {}
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    card = int(rng.choice([3, 3, 5, 5, 9, 10, 11, 6, 7, 12]))
    lo, hi = int(np.ceil(13 / np.log2(card))), int(np.floor(20 / np.log2(card)))
    width = int(rng.integers(lo, hi + 1))
    sep = int(rng.integers(1, width))
    ncl = int(rng.integers(3, 12))
    recipe = synthetic.wide_binary_tree if seed % 3 else synthetic.random_tree
    spec = recipe(n_cliques=ncl, width=width, sep=sep, card=card, seed=seed)
    pots = synthetic.potentials_for(spec, seed=seed + 1)
    opts = opts_all[seed % len(opts_all)]
    labels = sorted(spec["sizes"])
    obs = {labels[i]: int(rng.integers(0, card)) for i in rng.choice(len(labels), size=min(len(labels), int(rng.integers(1, 5))), replace=False)}
    for dtype in ("f64", "f32"):
        cast = [p.astype(np.float32) for p in pots] if dtype == "f32" else pots
        ref, z = oracle.beliefs_exact(spec["tree"], cast, spec["node_vars"], return_z=True)
        tol = RTOL32 if dtype == "f32" else RTOL64
        engine._cache.clear()
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, **opts)
        d = plan.describe()
        key = (card, "compact" if d.get("tmix_compact") else ("mixed" if d["tmix"] else "bit-field"))
        kinds[key] = kinds.get(key, 0) + 1
        for c in range(spec["n_cliques"]):
            plan.set_potential(c, cast[c])
        for rep in range(2):                       # (both halves of the message arena)
            plan.propagate()
            for node in range(len(spec["node_vars"])):
                close(plan.belief(node), ref[node], rtol=tol, what="seed %d %s %r node %d rep %d" % (seed, dtype, opts, node, rep))
            assert abs(plan.z() - z) <= (1e-6 if dtype == "f32" else 1e-11) * z
        c = int(rng.integers(0, spec["n_cliques"]))
        vs = list(spec["node_vars"][c])
        labs = [vs[i] for i in rng.permutation(len(vs))[:int(rng.integers(0, min(len(vs), 3) + 1))]]
        want = np.einsum(ref[c], list(range(len(vs))), [vs.index(v) for v in labs])
        close(plan.marginal(c, labs), want, rtol=tol, what="seed %d marginal" % seed)
        plan.set_evidence(obs)
        plan.propagate()
        w, zb = oracle.beliefs_exact(spec["tree"], _indicator_potentials(spec, cast, obs), spec["node_vars"], return_z=True)
        assert abs(plan.z() - zb) <= (1e-6 if dtype == "f32" else 1e-11) * zb + 1e-300, (seed, dtype)
        for node in rng.permutation(len(spec["node_vars"]))[:4]:
            close(plan.belief(int(node)), w[int(node)], rtol=tol, what="seed %d %s evidence node %d" % (seed, dtype, node))
        assert plan.stats()["flow_fallbacks"] == 0
        plan.close()
    if (seed - first) % 10 == 9:
        print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("%d trees x 2 storage types ok in %.0f s; (cardinality, form of the rows) of the plans: %r" % (n, time.time() - t0, dict(sorted(kinds.items()))))
