"""One-off wider run of the hard-evidence tests: N random junction trees (cardinalities 1..8), 3..11 evidence sets each with random
observations, as separate tables, shared tables and multi-set plans; every belief and Z of every set against the oracle on
indicator-multiplied potentials.      python tools/gpu_fuzz_evidence.py [N] [first seed] [width]"""
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
width = int(sys.argv[3]) if len(sys.argv) > 3 else 5
t0, refused, done = time.time(), [], {}
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    if os.environ.get("FUZZ_BIG"):          # fewer, larger cliques of cardinalities 2..7 (tables of up to 2^20 entries, many rows)
        while True:
            spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 10)), max_width=width, cards=(2, 3, 3, 4, 5, 6, 7))
            if 1 << 13 <= max(p.size for p in pots) <= 1 << 20 and sum(p.size for p in pots) <= 1 << 22:
                break
    else:
        spec, pots = random_junction_tree(rng, n_cliques=int(rng.integers(2, 25)), max_width=width)
    nc, nb = spec["n_cliques"], int(rng.integers(3, 12))
    share = [False, True, "multiset"][seed % 3]
    dtype = ("f64", "f32")[(seed // 3) % 2]
    base = [p.astype(np.float32) for p in pots] if dtype == "f32" else pots
    labels = sorted(spec["sizes"], key=str)
    observed = [{}]
    for b in range(1, nb):
        k = int(rng.integers(1, min(5, len(labels)) + 1))
        observed.append({labels[i]: int(rng.integers(0, spec["sizes"][labels[i]])) for i in rng.choice(len(labels), size=k, replace=False)})
    try:
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, n_batch=nb, share_potentials=bool(share), multiset=share == "multiset")
    except ValueError as exc:
        print("seed %d %s share=%r: refused: %s" % (seed, dtype, share, exc), flush=True)
        refused.append(seed)
        continue
    for b in range(nb if not share else 1):
        for c in range(nc):
            plan.set_potential(c, base[c], batch=b)
    for b in range(nb):
        plan.set_evidence(observed[b], batch=b)
    plan.propagate(0, nb)
    for b in range(nb):
        ps = [np.asarray(p, dtype=np.float64).copy() for p in base]
        for var, state in observed[b].items():
            host = next(c for c in range(nc) if var in spec["node_vars"][c])
            axis = spec["node_vars"][host].index(var)
            ind = np.zeros(spec["sizes"][var]); ind[state] = 1.0
            shape = [1] * ps[host].ndim; shape[axis] = spec["sizes"][var]
            ps[host] = ps[host] * ind.reshape(shape)
        want, z = oracle.beliefs_exact(spec["tree"], ps, spec["node_vars"], return_z=True)
        for node in range(len(spec["node_vars"])):
            close(plan.belief(node, batch=b), want[node], rtol=RTOL32 if dtype == "f32" else RTOL64, what="seed %d share %r set %d node %d" % (seed, share, b, node))
        assert abs(plan.z(batch=b) - z) <= (1e-6 if dtype == "f32" else 1e-11) * abs(z) + 1e-300, (seed, b, plan.z(batch=b), z)
    assert plan.stats()["flow_fallbacks"] == 0
    done[(share, dtype)] = done.get((share, dtype), 0) + nb
    plan.close()
    if (seed - first) % 20 == 19:
        print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("%d random trees ok in %.0f s; evidence sets checked per (sharing, storage): %r; refused: %r" % (n, time.time() - t0, done, refused))
