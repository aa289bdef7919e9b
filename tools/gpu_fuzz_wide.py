"""One-off: trees of WIDE power-of-two cliques (the regime of the benchmark configs: 2^14..2^21-entry tables, 64-row workgroups, partial copies,
reduce tasks) with random shape parameters, every belief and Z against the oracle.      python tools/gpu_fuzz_wide.py [N] [first seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "junction-tree_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import jt_oracle as oracle
from test_gpu_parity import close, RTOL32, RTOL64
from junctiontree_amd import engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
opts_all = [{}, {}, {"level_launches": True}, {"flow_tickets": True}, {"block_log2": 12}, {"layout_policy": 2}, {"layout_policy": 3}, {"keep_root": True}]
t0, modes, refused = time.time(), {}, []
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    card = int(rng.choice([2, 2, 2, 4, 8]))
    bits = int(rng.integers(14, 22))
    width = max(2, bits // {2: 1, 4: 2, 8: 3}[card])
    sep = int(rng.integers(1, width))
    nc = int(rng.integers(3, 128 if bits <= 16 else (40 if bits <= 18 else 12)))
    recipe = [synthetic.wide_binary_tree, synthetic.random_tree, synthetic.chain_tree][seed % 3]
    if recipe is synthetic.chain_tree:
        c3 = int(rng.choice([16, 32, 64]))
        spec = synthetic.chain_tree(n_cliques=nc, card=c3, width=3)
    else:
        spec = recipe(n_cliques=nc, width=width, sep=sep, card=card, seed=seed)
    dtype = ("f32", "f64")[(seed // 3) % 2]
    pots = synthetic.potentials_for(spec, seed=seed, dtype=np.float32 if dtype == "f32" else np.float64)
    want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
    opts = opts_all[seed % len(opts_all)]
    try:
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=dtype, **opts)
    except ValueError as exc:
        print("seed %d %s %r (%s, %d cliques, width %d, sep %d, card %d): refused: %s" % (seed, dtype, opts, recipe.__name__, nc, width, sep, card, exc), flush=True)
        refused.append(seed)
        continue
    for c in range(spec["n_cliques"]):
        plan.set_potential(c, pots[c])
    for rep in range(2):
        plan.propagate()
    st = plan.stats()
    assert st["flow_fallbacks"] == 0
    for node in range(len(spec["node_vars"])):
        close(plan.belief(node), want[node], rtol=RTOL32 if dtype == "f32" else RTOL64, what="seed %d %s %r node %d" % (seed, dtype, opts, node))
    assert abs(plan.z() - z) <= (1e-6 if dtype == "f32" else 1e-11) * abs(z)
    d = plan.describe()
    key = (st["launch_mode"], max(t["total"] for t in d["tasks"]), any(t["kind"] == 1 for t in d["tasks"]))
    modes[key] = modes.get(key, 0) + 1
    plan.close()
    if (seed - first) % 10 == 9:
        print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("%d wide trees ok in %.0f s; (launch mode, most rows per workgroup, reduce tasks) of the plans: %r; refused: %r" % (n, time.time() - t0, modes, refused))
