"""Development aid: time every rank's share of the multi-GPU C4 plan on ONE GPU.

    JTP_FAKE_COMM=1 python tools/rank_time.py [world] [steps]

With JTP_FAKE_COMM the engine replaces each exchange group by a fill of the receive buffers, so a
rank's kernels run as they would between exchanges (minus the RCCL calls).  The 8-GPU step time is
bounded below by the slowest chain  rank-subtree collect -> top of the tree -> rank-subtree distribute.
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
os.environ.setdefault("JTP_FAKE_COMM", "1")
from junctiontree_amd import engine, partition, synthetic

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
n = spec["n_cliques"]
replicate = not (len(sys.argv) > 3 and sys.argv[3] == "noreplicate")
owner = partition.subtree_owners(spec["parent"], [1.0] * n, world, replicate_top=replicate)
print("world %d, top part %s" % (world, "replicated on every rank" if replicate else "on one rank"))
alg = synthetic.algorithmic_bytes(spec, 4)
for rank in range(world):
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", n_ranks=world, rank=rank, owner=owner)
    plan.fill_synthetic(1, spec["scales"])
    for _ in range(3):
        plan.propagate(sync=False)
    plan.sync()
    # (a) the steps between ONE event pair (what bench.py's `c4_rank_share_of_8` reports); (b) with the three phase events per
    #     propagate that split collect from distribute (2-3 us of idle GPU each: the per-propagate figure of rounds 1-3)
    plan.region_begin()
    for _ in range(steps):
        plan.propagate(sync=False)
    region_us = plan.region_end() / steps * 1e3
    plan.set_profiling(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.propagate(sync=False)
    plan.sync()
    dt = (time.perf_counter() - t0) / steps
    st = plan.stats()
    d = plan.describe()
    print("rank %d: %3d cliques  %2d launches  %d exchange groups  %.1f us/propagate (collect %.1f, distribute %.1f)  share %.0f MB  | between one event pair: %.1f us/propagate" % (
        rank, sum(1 for o in owner if o in (rank, world)), st["n_launches"], sum(1 for k, _, _ in d["flow_steps"] if k == 1),
        dt * 1e6, st["collect_ms"] * 1e3, st["distribute_ms"] * 1e3, st["algorithmic_bytes"] / 1e6, region_us))
    plan.close()
