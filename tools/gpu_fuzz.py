"""One-off wider run of tests/test_gpu_parity.py::test_random_trees_mixed_cardinalities_on_device: N random junction trees
(cardinalities 1..8, up to 40 cliques of up to `width` variables, planner options cycled) against the oracle, f64 and f32.
    python tools/gpu_fuzz.py [N] [first seed] [width]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "junction-tree_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import jt_oracle as oracle
from test_planner_emulated import random_junction_tree
from test_gpu_parity import close, RTOL32, RTOL64
from junctiontree_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
width = int(sys.argv[3]) if len(sys.argv) > 3 else 6
opts_all = [{}, {"block_log2": 10}, {"layout_policy": 1}, {"keep_root": True}, {"split_variants": True}, {"level_launches": True}, {"flow_tickets": True}, {"no_compact": True}]
modes = {}
refused = []


def run_plan(spec, pots, dtype, **opts):
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, **opts)
    for c in range(spec["n_cliques"]):
        plan.set_potential(c, pots[c])
    plan.propagate()
    st, d = plan.stats(), plan.describe()
    assert st["flow_fallbacks"] == 0
    key = (st["launch_mode"], d["tmix"], d["compact"])
    modes[key] = modes.get(key, 0) + 1
    out = [plan.belief(n) for n in range(len(spec["node_vars"]))]
    z = plan.z()
    # marginals onto random subsets of random cliques' variables (their own planner tasks and kernels): against the belief summed on the host
    mrng = np.random.default_rng(len(pots) + int(pots[0].size))
    reqs = []
    for _ in range(4):
        c = int(mrng.integers(0, spec["n_cliques"]))
        vs = list(spec["node_vars"][c])
        k = int(mrng.integers(0, len(vs) + 1))
        reqs.append((c, [vs[i] for i in mrng.permutation(len(vs))[:k]]))
    got = plan.marginals(reqs) + [plan.marginal(*reqs[0])]
    for (c, labs), g in zip(reqs + [reqs[0]], got):
        vs = list(spec["node_vars"][c])
        want = np.einsum(out[c], list(range(len(vs))), [vs.index(v) for v in labs])
        tol = 1e-5 if dtype == "f32" else 1e-10
        assert g.shape == want.shape and np.all(np.abs(g - want) <= tol * max(np.max(np.abs(want)), 1e-300)), ("marginal", c, labs, float(np.max(np.abs(g - want))), float(np.max(np.abs(want))))
    # the same plan again with new tables on some cliques (both halves of the message arena, markers re-armed by the producers)
    cur = [np.array(p0, copy=True) for p0 in pots]
    for rnd in range(3):
        for c in mrng.permutation(spec["n_cliques"])[:int(mrng.integers(1, spec["n_cliques"] + 1))]:
            cur[c] = (cur[c] * mrng.uniform(0.5, 1.5, cur[c].shape)).astype(cur[c].dtype)
            plan.set_potential(int(c), cur[c])
        plan.propagate(sync=bool(rnd % 2))
        want2, z2 = oracle.beliefs_exact(spec["tree"], cur, spec["node_vars"], return_z=True)
        for node in mrng.permutation(len(spec["node_vars"]))[:3]:
            close(plan.belief(int(node)), want2[int(node)], rtol=RTOL32 if dtype == "f32" else RTOL64, what="round %d node %d" % (rnd, node))
        assert abs(plan.z() - z2) <= 1e-5 * abs(z2)
    assert plan.stats()["flow_fallbacks"] == 0
    plan.close()
    return out, z


t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    if os.environ.get("FUZZ_BIG"):          # fewer, much larger cliques of cardinalities 2..7 (tables of up to 2^22 entries, many rows)
        while True:
            spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 12)), max_width=width, cards=(2, 3, 3, 4, 5, 6, 7))
            big = max(p.size for p in pots)
            if 1 << 14 <= big <= 1 << 22 and sum(p.size for p in pots) <= 1 << 24:
                break
    else:
        spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 40)), max_width=width)
    want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
    opts = opts_all[int(os.environ["FUZZ_OPTS"]) if "FUZZ_OPTS" in os.environ else seed % len(opts_all)]
    for dtype in ("f64", "f32"):
        cast = [p.astype(np.float32) for p in pots] if dtype == "f32" else pots
        ref = oracle.beliefs_exact(spec["tree"], cast, spec["node_vars"]) if dtype == "f32" else want
        engine._cache.clear()
        try:
            out, zz = run_plan(spec, cast, dtype, **opts)
        except Exception as exc:                     # a structure the planner refuses (JTP_EUNSUPPORTED) or any other error: reported, not a mismatch
            print("seed %d %s %r: %s: %s" % (seed, dtype, opts, type(exc).__name__, exc), flush=True)
            refused.append(seed)
            zz = z
            continue
        for o, w in zip(out, ref):
            close(o, w, rtol=RTOL32 if dtype == "f32" else RTOL64, what="seed %d %s %r" % (seed, dtype, opts))
    assert abs(zz - z) <= 1e-5 * abs(z), (seed, zz, z)
    if (seed - first) % 20 == 19:
        print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("%d random trees x 2 storage types ok in %.0f s; (launch mode, mixed-radix rows, compact) of the plans: %r; refused: %r" % (n, time.time() - t0, modes, sorted(set(refused))))
