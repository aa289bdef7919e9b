"""Benchmark of the message-passing hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W            (single GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one propagate (collect + distribute) over BASELINE.json config 4: the synthetic
wide-clique junction tree (256 cliques of width 20, cardinality 2, 2^20-entry float32
potentials, 10 variables shared per edge, balanced binary tree), potentials resident in HBM
(generated on the device by jtp_fill_synthetic).  With N > 1 the tree is cut into subtrees
(junctiontree_amd/partition.py), one process per GPU, separator messages exchanged by RCCL
send/recv at the cuts; total work is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0.  `value` = algorithmic clique-potential GB/s of the whole
job (SURVEY.md 8d definition of algorithmic bytes), `messages_per_sec` beside it.
`roofline` is for the dominant kernel (jt_distribute_flow: the whole distribute phase in one
launch), timed with hipEvents on the plan's own stream during the timed steps: one event before
and one after the launch in every propagate.  `cpu_baseline` times the numpy restatement of the
reference's einsum sequence (oracle/jt_oracle.py: beliefs_refshaped) on one host core over a
bounded sample of the same workload; it is a checker/baseline, never the measured path.

No torch in this process: the launcher (torch.distributed.run) only provides RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the rendezvous (RCCL unique id, barrier, max of the
timings) runs over plain sockets (junctiontree_amd/rendezvous.py), because importing torch next
to libjtprop.so brings a second ROCm runtime into the process and RCCL's communicator init fails.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "junction-tree_amd"))

Z_DEFAULT_C4 = 1.0058528272803358     # partition function of the default workload (one GPU; numpy oracle agrees to 1e-9)
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md "Chip-level parameters": 8.0 TB/s spec
TRAFFIC_FILE = "r02_hbm_traffic.json"


# beliefs_refshaped issues the reference's 5N-1 einsum sequence but with label-aligned (first-appearance) axis
# orders; the unmodified reference (colour-sorted labels, list(set()) scopes) ran 1.83x (this round; 2.7x measured by
# the round-1 judge) longer on the full config-4 input in the build container.  The port therefore FLATTERS the CPU.
REFERENCE_OVER_PORT_TIME = 1.83


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _one_sample_propagate(args):
    width, sep, card, n_sample, seed = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "junction-tree_amd"))
    import numpy as np
    import jt_oracle as oracle
    from junctiontree_amd import synthetic
    spec = synthetic.wide_binary_tree(n_cliques=n_sample, width=width, sep=sep, card=card, seed=0)
    pots = synthetic.potentials_for(spec, seed=1 + seed, dtype=np.float32)
    t0 = time.perf_counter()
    oracle.beliefs_refshaped(spec["tree"], pots, spec["node_vars"])
    return time.perf_counter() - t0, synthetic.algorithmic_bytes(spec, 4)


def cpu_baseline(width, sep, card, n_sample, seed):
    """Reference-shaped numpy path on one core over a smaller tree of the same clique shape."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import jt_oracle as oracle
    from junctiontree_amd import synthetic
    spec = synthetic.wide_binary_tree(n_cliques=n_sample, width=width, sep=sep, card=card, seed=seed)
    pots = synthetic.potentials_for(spec, seed=1, dtype=np.float32)
    t0 = time.perf_counter()
    c0 = time.process_time()
    beliefs = oracle.beliefs_refshaped(spec["tree"], pots, spec["node_vars"])
    wall = time.perf_counter() - t0
    cpu = time.process_time() - c0
    ab = synthetic.algorithmic_bytes(spec, 4)
    return {
        "value": ab["total"] / wall / 1e9, "unit": "GB/s",
        "messages_per_sec": ab["messages"] / wall,
        "cores": 1, "kind": "port", "Z": float(np.sum(beliefs[0], dtype=np.float64)),
        "cpu_model": _cpu_model(), "host_cores": os.cpu_count(),
        "reference_over_port_time": REFERENCE_OVER_PORT_TIME,
        "reference_equivalent_value": ab["total"] / wall / 1e9 / REFERENCE_OVER_PORT_TIME,
        "sample": "%d-clique balanced binary tree, same clique shape (width %d, card %d, %d shared); "
                  "one propagate, %.1f s wall, cpu/wall %.2f" % (n_sample, width, card, sep, wall, cpu / max(wall, 1e-9)),
    }


def cpu_baseline_all_cores(width, sep, card, n_sample=16):
    """BASELINE configs[4] on the host: evidence sets are independent, so the CPU runs P of them at
    once, one process per core, each one propagate of a bounded sample tree of the same clique shape."""
    import multiprocessing as mp
    procs = max(1, min(os.cpu_count() or 1, 16))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(procs) as pool:
        res = pool.map(_one_sample_propagate, [(width, sep, card, n_sample, i) for i in range(procs)])
    wall = time.perf_counter() - t0
    inner = max(r[0] for r in res)
    ab = res[0][1]
    return {
        "value": ab["total"] * procs / inner / 1e9, "unit": "GB/s",
        "messages_per_sec": ab["messages"] * procs / inner, "cores": procs, "kind": "port",
        "sample": "%d processes, each one propagate of a %d-clique tree of the same clique shape with its own values "
                  "(independent evidence sets); slowest process %.2f s, pool wall %.1f s" % (procs, n_sample, inner, wall),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # 200 x 0.65 ms: the timed region is not inside box-to-box noise
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c4", choices=["c4", "c2"],
                    help="c4 = BASELINE configs[3] (default, the metric's workload); c2 = configs[1]: chain of "
                         "1000 cliques, width 3, cardinality 64, float64 (latency-bound, reported in DESIGN.md)")
    ap.add_argument("--batch", type=int, default=1,
                    help="independent evidence sets per step, one HIP stream each (BASELINE configs[4] in "
                         "miniature; default 1 = the metric's workload)")
    ap.add_argument("--share", action="store_true",
                    help="with --batch: the evidence sets share one set of clique tables (JTP_SHARE_POTENTIALS) and "
                         "differ by hard evidence on 16 variables each (SURVEY 8d, config 5)")
    ap.add_argument("--multiset", action="store_true",
                    help="with --batch: JTP_MULTISET plan - the evidence sets share the tables and every pass over a table "
                         "serves eight sets (no belief tables; beliefs and marginals are formed on demand)")
    ap.add_argument("--no-replicate-top", action="store_true",
                    help="N > 1: give the top part of the partition to one rank instead of replicating it on all")
    ap.add_argument("--cliques", type=int, default=256)
    ap.add_argument("--width", type=int, default=20)
    ap.add_argument("--sep", type=int, default=10)
    ap.add_argument("--card", type=int, default=2)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-sample", type=int, default=256,
                    help="cliques in the CPU baseline tree (default: the full workload, ~10-25 s on one core; 0 = skip)")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the CPU port on every host core at once (independent evidence sets, SURVEY.md 8d)")
    ap.add_argument("--no-profile", action="store_true", help="no hipEvent pairs in the timed region")
    ap.add_argument("--block-log2", type=int, default=0)
    ap.add_argument("--lds-budget", type=int, default=0)
    ap.add_argument("--layout-policy", type=int, default=0)
    ap.add_argument("--per-launch", action="store_true", help="print per-launch device times to stderr")
    ap.add_argument("--level-launches", action="store_true",
                    help="one launch per tree level instead of one dataflow launch per phase")
    ap.add_argument("--split-variants", action="store_true",
                    help="one launch per (level, clique shape): per-shape timings (profiling aid)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import ctypes as C
    from junctiontree_amd import _capi, engine, partition, synthetic
    lib = _capi.lib()                               # loads libjtprop.so (and its HIP runtime) first
    ndev = _capi.device_count()
    if ndev <= 0:
        raise SystemExit("no HIP device visible: bench.py measures the GPU path only")
    device = local_rank % ndev                      # one process per GPU (ranks > GPUs only in smoke runs)
    if world > ndev:                                # ranks sharing a GPU: dataflow launches need ticket order
        os.environ["JTP_FLOW_TICKETS"] = "1"

    from junctiontree_amd.rendezvous import Rendezvous
    master_port = int(os.environ.get("MASTER_PORT", "29500"))
    run_id = "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "none") if ch.isalnum())[:32]
    rdzv = Rendezvous(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), master_port + 1,
                      port_file="/tmp/jtp_rdzv_%d_%s_%d" % (master_port, run_id, os.getppid()))
    if world > 1:
        uid = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            _capi.check(lib.jtp_comm_unique_id(buf))
            uid = buf.raw
        uid = rdzv.broadcast(uid)
        _capi.check(lib.jtp_comm_init(rank, world, C.c_char_p(uid), device))

    def barrier():
        rdzv.barrier()

    if args.config == "c2":
        args.dtype, args.cpu_sample = "f64", 0
        spec = synthetic.chain_tree(n_cliques=1000 if args.cliques == 256 else args.cliques,
                                    card=64 if args.card == 2 else args.card, width=3 if args.width == 20 else args.width)
    else:
        spec = synthetic.wide_binary_tree(n_cliques=args.cliques, width=args.width, sep=args.sep,
                                          card=args.card, seed=0)
    itemsize = 4 if args.dtype == "f32" else 8
    alg = synthetic.algorithmic_bytes(spec, itemsize)
    n = spec["n_cliques"]
    # the small top part of the partition is replicated on every rank (one exchange per propagate instead of two)
    owner = partition.subtree_owners(spec["parent"], [1.0] * n, world, replicate_top=not args.no_replicate_top)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=args.dtype,
                       device=device, n_ranks=world, rank=rank, owner=owner, n_batch=args.batch,
                       block_log2=args.block_log2, lds_budget=args.lds_budget,
                       layout_policy=args.layout_policy, split_variants=args.split_variants,
                       level_launches=args.level_launches, share_potentials=args.share, multiset=args.multiset)
    if args.share or args.multiset:
        import numpy as np
        plan.fill_synthetic(1, spec["scales"])
        labels = sorted(spec["sizes"])
        for b in range(args.batch):
            rng = np.random.default_rng(1000 + b + 64 * rank)
            plan.set_evidence({labels[i]: int(rng.integers(0, spec["sizes"][labels[i]]))
                               for i in rng.choice(len(labels), size=16, replace=False)}, batch=b)
    else:
        for b in range(args.batch):
            plan.fill_synthetic(1 + b, spec["scales"], batch=b)

    for _ in range(args.warmup):
        plan.propagate(sync=False)
    plan.sync()
    if args.share or args.multiset:
        # Evidence sets that share their tables: a table need only be read ONCE per batch, whatever the engine does
        # (a --share plan streams it once per set, mostly out of the Infinity Cache; a --multiset plan once per group
        # of eight sets).  Algorithmic bytes: tables once per batch; per set its messages and - unless beliefs are
        # formed on demand (--multiset) - its belief tables.
        sz = [1] * len(spec["node_vars"])
        for i, labs in enumerate(spec["node_vars"]):
            for v in labs:
                sz[i] *= spec["sizes"][v]
        tables, seps = sum(sz[:spec["n_cliques"]]) * itemsize, sum(sz[spec["n_cliques"]:]) * 8
        per_set = 5 * seps + (0 if args.multiset else tables)
        alg = dict(alg, total=(2 * tables + per_set * args.batch) / args.batch, read=2 * tables / args.batch)
    if not args.no_profile:
        plan.set_profiling(args.steps, per_launch=args.per_launch or args.split_variants)
    barrier()
    plan.sync()                                      # hipStreamSynchronize on the plan's stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.propagate(sync=False)
    plan.sync()
    barrier()
    elapsed = rdzv.allreduce_max(time.perf_counter() - t0)      # max over ranks

    stats = plan.stats()
    z = plan.z() if plan.owns(plan.root) else None
    if world > 1:                                    # the rank holding the root clique knows Z: share it
        z = rdzv.allreduce_max(z if z is not None else -1.0e300)
    if args.per_launch and not args.no_profile and rank == 0:
        for L in plan.launch_ms():
            print("# %s level %d: %4d tasks %5d blocks %8.4f ms %7.0f GB/s" % (
                "collect   " if L["phase"] == 0 else "distribute", L["level"], L["ntasks"], L["nblocks"],
                L["ms"], L["alg_bytes"] / max(L["ms"], 1e-9) / 1e6), file=sys.stderr)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gbps = alg["total"] * args.batch * args.steps / elapsed / 1e9
        out = {
            "metric": "clique-potential GB/s (algorithmic bytes per propagate / time; messages/sec alongside), "
                      "synthetic width-%d tree" % args.width,
            "value": gbps, "unit": "GB/s",
            "messages_per_sec": alg["messages"] * args.batch * args.steps / elapsed,
            "read_GBps": alg["read"] * args.batch * args.steps / elapsed / 1e9,
            "frac_of_hbm_roofline": gbps / (HBM_PEAK_GBPS * world),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": ("BASELINE.json configs[1]: chain of %d cliques, width 3, cardinality 64, float64" % n)
                            if args.config == "c2" else
                            "BASELINE.json configs[3]: %d cliques, width %d, cardinality %d (2^%d-entry %s "
                            "potentials), %d shared variables per edge, balanced binary tree"
                            % (n, args.width, args.card, args.width, args.dtype, args.sep),
                "algorithmic_bytes_per_step": alg["total"] * args.batch, "messages_per_step": alg["messages"] * args.batch,
                "evidence_sets_per_step": args.batch, "shared_potentials": bool(args.share or args.multiset),
                "multiset": bool(args.multiset),
                "engine_table_bytes_per_step": stats["algorithmic_bytes"] if args.multiset else None,
                "parallelism": "1 GPU" if world == 1 else "subtree-sharded x%d, RCCL send/recv at cuts%s" % (world, "" if args.no_replicate_top else ", top part replicated"),
                "launches_per_step": stats["n_launches"], "Z": z,
            },
        }
        if not args.no_profile and stats["kernels"]:
            name, k = max(stats["kernels"].items(), key=lambda kv: kv[1]["ms"])
            per_launch_bytes = k["bytes"] / k["launches"]          # (profiled: evidence set 0 only)
            per_launch_ms = k["ms"] / k["launches"]
            achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
            traffic = None                 # HBM bytes per launch: NOT measured in this run - read from the committed
            traffic_source = None          # rocprofv3 PMC passes of the same command (tools/collect_profiles.sh)
            try:
                traffic_file = os.path.join("profiles", TRAFFIC_FILE)
                with open(os.path.join(ROOT, traffic_file)) as fh:
                    prof = json.load(fh)["kernels"]
                if args.config == "c4" and args.cliques == 256 and args.width == 20 and world == 1 and args.dtype == "f32":
                    traffic = prof["void " + name]["hbm_bytes_per_launch"]
                    traffic_source = traffic_file + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2)"
            except (OSError, KeyError, ValueError):
                pass
            out["roofline"] = {
                "bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                "launches_per_step": k["launches"], "avg_launch_ms": per_launch_ms,
                "algorithmic_bytes_per_launch": per_launch_bytes,
                "rank0_kernels": {kn: {"ms_per_step": kv["ms"], "launches": kv["launches"],
                                       "GBps": kv["bytes"] / max(kv["ms"], 1e-12) / 1e6}
                                  for kn, kv in stats["kernels"].items()},
                "collect_ms": stats["collect_ms"], "distribute_ms": stats["distribute_ms"],
            }
        if args.cpu_sample > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.width, args.sep, args.card, args.cpu_sample, 0)
            if args.cpu_all_cores:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.width, args.sep, args.card)
            if args.cpu_sample == n and args.config == "c4" and z is not None:      # same tree, same values: same Z
                zc = out["cpu_baseline"]["Z"]
                out["parity"] = {"Z_gpu": z, "Z_cpu_oracle": zc, "rel_err": abs(z - zc) / abs(zc)}
        default_c4 = (args.config == "c4" and args.cliques == 256 and args.width == 20 and args.sep == 10
                      and args.card == 2 and args.dtype == "f32")
        if default_c4 and z is not None:
            # Z of the default workload as one GPU and the numpy oracle compute it: a sharded run must agree
            out["config"]["Z_expected"] = Z_DEFAULT_C4
            out["config"]["Z_rel_err"] = abs(z - Z_DEFAULT_C4) / Z_DEFAULT_C4
        print(json.dumps(out), flush=True)

    plan.close()
    if world > 1:
        barrier()
        lib.jtp_comm_destroy()
    rdzv.close()


if __name__ == "__main__":
    main()
