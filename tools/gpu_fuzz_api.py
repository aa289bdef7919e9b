"""One-off: N random factor graphs through the public API (jt.create_junction_tree + tree.propagate) against the brute-force joint.
    python tools/gpu_fuzz_api.py [N] [first seed]"""
import os, sys, time, string
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "junction-tree_amd"))
import numpy as np
import junctiontree_amd as jt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0, n_sets, n_again, n_cond, failed = time.time(), 0, 0, 0, []
n_fold_plans = n_fold_tasks = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    wide = os.environ.get("FUZZ_WIDE")        # wider factors (up to 6 variables of up to 5 states) over up to 12 variables
    nv = int(rng.integers(2, 13 if wide else 11))
    names = ["v%d" % i for i in range(nv)] if seed % 2 else list(range(nv))
    while True:
        sizes = {v: int(rng.integers(1, 6 if wide else 5)) for v in names}
        if np.prod([float(k) for k in sizes.values()]) <= 4e6:
            break
    nf = int(rng.integers(1, 2 * nv))
    factors = []
    for _ in range(nf):
        k = int(rng.integers(1, min(6 if wide else 4, nv) + 1))
        factors.append([names[i] for i in rng.choice(nv, size=k, replace=False)])
    used = {v for f in factors for v in f}
    f32 = seed % 3 == 0
    values = [rng.uniform(0.2, 1.0, [sizes[v] for v in f]).astype(np.float32 if f32 else np.float64) for f in factors]
    tree = jt.create_junction_tree(factors, dict(sizes))          # (the tree keeps the dict it is given: conditioning below changes it)
    try:
        got = tree.propagate(values)
    except Exception as exc:
        print("seed %d: %s: %s" % (seed, type(exc).__name__, exc), flush=True)
        failed.append(seed)
        continue
    nft = sum(1 for t in tree.plan("f32" if f32 else "f64").describe()["tasks"] if t["fold"])     # (round 6) marginals formed inside the propagate
    n_fold_plans += nft > 0
    n_fold_tasks += nft
    # brute force: the joint over the variables that occur, then every factor's marginal
    order = sorted(used, key=str)
    ax = {v: i for i, v in enumerate(order)}
    ops = []
    for f, val in zip(factors, values):
        ops += [np.asarray(val, dtype=np.float64), [ax[v] for v in f]]
    joint = np.einsum(*ops, list(range(len(order))), optimize=True)
    for f, g in zip(factors, got):
        want = np.einsum(joint, list(range(len(order))), [ax[v] for v in f])
        assert g.shape == want.shape, (seed, f)
        np.testing.assert_allclose(g, want, rtol=2e-6 if f32 else 1e-11, atol=1e-30, err_msg="seed %d factor %r" % (seed, f))
    if seed % 3 == 1:          # some factors get new values: only their cliques are staged again (content digests), same answer as from scratch
        for rnd in range(2):
            for i in rng.choice(len(factors), size=int(rng.integers(1, len(factors) + 1)), replace=False):
                values[i] = rng.uniform(0.2, 1.0, values[i].shape).astype(values[i].dtype)
            got = tree.propagate(values)
            ops = []
            for f, val in zip(factors, values):
                ops += [np.asarray(val, dtype=np.float64), [ax[v] for v in f]]
            joint = np.einsum(*ops, list(range(len(order))), optimize=True)
            for f, g in zip(factors, got):
                want = np.einsum(joint, list(range(len(order))), [ax[v] for v in f])
                np.testing.assert_allclose(g, want, rtol=2e-6 if f32 else 1e-11, atol=1e-30, err_msg="seed %d round %d factor %r" % (seed, rnd, f))
        n_again += 2
    if seed % 5 == 2:          # the reference's way of conditioning (tests/test_junctiontree.py:393-411): set a variable's size to 1 and slice the factors
        cond = [np.array(v, copy=True) for v in values]
        csizes = dict(sizes)
        for v in [order[i] for i in rng.permutation(len(order))[:int(rng.integers(1, min(3, len(order)) + 1))]]:
            st = int(rng.integers(0, sizes[v]))
            tree.clique_tree.factor_graph.sizes[v] = 1
            csizes[v] = 1
            for i, f in enumerate(factors):
                if v in f:
                    cond[i] = np.take(cond[i], [st], axis=f.index(v))
            try:
                got_c = tree.propagate(cond)
            except Exception as exc:
                print("seed %d conditioning on %r (state %d): %s; sizes %r; factors %r; shapes %r" % (seed, v, st, exc, dict(tree.clique_tree.factor_graph.sizes), factors, [c.shape for c in cond]), flush=True)
                raise
            ops = []
            for f, val in zip(factors, cond):
                ops += [np.asarray(val, dtype=np.float64), [ax[u] for u in f]]
            jc = np.einsum(*ops, list(range(len(order))), optimize=True)
            for f, g in zip(factors, got_c):
                want = np.einsum(jc, list(range(len(order))), [ax[u] for u in f])
                assert g.shape == want.shape, (seed, f, g.shape, want.shape)
                np.testing.assert_allclose(g, want, rtol=2e-6 if f32 else 1e-11, atol=1e-30, err_msg="seed %d conditioned on %r factor %r" % (seed, v, f))
        for v in order:
            tree.clique_tree.factor_graph.sizes[v] = sizes[v]
        n_cond += 1
    if seed % 4 == 0:          # hard-evidence sets over the same values: every factor of set e against joint x indicators
        sets = [{}] + [{order[i]: int(rng.integers(0, sizes[order[i]])) for i in rng.choice(len(order), size=int(rng.integers(1, min(3, len(order)) + 1)), replace=False)}
                       for _ in range(int(rng.integers(1, 10)))]
        res = tree.propagate_evidence_sets(values, sets)
        for obs, got_e in zip(sets, res):
            je = joint.copy()
            for v, st in obs.items():
                ind = np.zeros(sizes[v]); ind[st] = 1.0
                shape = [1] * je.ndim; shape[ax[v]] = sizes[v]
                je = je * ind.reshape(shape)
            for f, g in zip(factors, got_e):
                want = np.einsum(je, list(range(len(order))), [ax[v] for v in f])
                np.testing.assert_allclose(g, want, rtol=2e-6 if f32 else 1e-11, atol=1e-30, err_msg="seed %d evidence %r factor %r" % (seed, obs, f))
        n_sets += len(sets)
    if (seed - first) % 50 == 49:
        print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("%d random factor graphs through create_junction_tree + propagate agree with the brute-force joint; %d evidence sets through propagate_evidence_sets and %d propagates with some factors changed too; %d graphs conditioned the reference's way (sizes set to 1, factors sliced) (%.0f s); plans with folded marginal tasks: %d (%d tasks); raised: %r" % (n, n_sets, n_again, n_cond, time.time() - t0, n_fold_plans, n_fold_tasks, failed))
