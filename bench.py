"""Benchmark of the message-passing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]      (N > 1: this process starts N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      (the launcher starts them)

One "step" = one propagate (collect + distribute) over BASELINE.json config 4: the synthetic
wide-clique junction tree (256 cliques of width 20, cardinality 2, 2^20-entry float32
potentials, 10 variables shared per edge, balanced binary tree), potentials resident in HBM
(generated on the device by jtp_fill_synthetic).  With N > 1 the tree is cut into subtrees
(junctiontree_amd/partition.py), one process per GPU, separator messages exchanged by RCCL
send/recv at the cuts; total work is fixed, so scaling is "strong".  `--config c2` / `--config c3`
run BASELINE configs[1] (chain of 1000 cliques of 64^3 doubles) and configs[2] (6 x 167 lattice
MRF of cardinality 8, junction tree by this repo's own builder) in the same format.

Prints ONE JSON line on rank 0.  `value` = algorithmic clique-potential GB/s of the whole
job (SURVEY.md 8d definition of algorithmic bytes), `messages_per_sec` beside it.
`roofline` is for the dominant kernel (config 4: jt_propagate_flow, both phases of the propagate
in one dataflow launch; configs 2 and 3: jt_distribute_flow[_chain], the whole distribute phase in
one launch), timed with hipEvents on the plan's own stream: ONE event before the first timed step and
ONE after the last (no event between the steps: an event costs 2-3 us of idle GPU), divided by the
steps; where a step is two launches, the kernel's share of it comes from a few propagates timed after
the timed region.  With the default workload on one GPU the line also carries `configs`: one measured
sub-result per other BASELINE config (c2, c3, c3_api_end_to_end, c5_multiset64), each with
`ms_per_step`, algorithmic bytes, roofline fraction and a parity flag.  `cpu_baseline` times the numpy restatement of the
reference's einsum sequence (oracle/jt_oracle.py: beliefs_refshaped) on one host core over a
bounded sample of the same workload; it is a checker/baseline, never the measured path.

No torch in this process: a launcher only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT; the rendezvous (RCCL unique id, barrier, max of the timings) runs over plain sockets
(junctiontree_amd/rendezvous.py), because importing torch next to libjtprop.so brings a second ROCm
runtime into the process and RCCL's communicator init fails.
"""

import argparse
import faulthandler
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "junction-tree_amd"))

Z_DEFAULT_C4 = 1.0058528272803358     # partition function of the default workload (one GPU; numpy oracle agrees to 1e-9)
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md "Chip-level parameters": 8.0 TB/s spec
# float64 vector pipe: v_fma_f64 measured on this pool with every CU busy (tools/f64_rate.hip, profiles/r02_f64_rate.txt:
# 5.1 cycles per wave instruction and SIMD = 61.1 TFLOP/s of fused multiply-adds, 30.5 T lane instructions/s; a
# v_mul_f64 takes the same 5.1 cycles); 78.6 TFLOP/s is the data-sheet figure (4 cycles)
F64_PEAK_TFLOPS, F64_PEAK_TINSTS, F64_SPEC_TFLOPS = 61.1, 30.5, 78.6


def multiset_roofline(stats, ms, alg_bytes):
    """jt_multi_flow (evidence sets over shared tables): bound by the float64 vector pipe, not by HBM - one pass over a table
    row serves eight sets, so the arithmetic per byte is eight times a single set's."""
    tf = stats["f64_flops"] / (ms * 1e-3) / 1e12
    ti = stats["f64_insts"] / (ms * 1e-3) / 1e12
    return {"bound": "valu_f64", "kernel": "jt_multi_flow<float>", "achieved": tf, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_PEAK_TFLOPS,
            "peak_source": "v_fma_f64 measured on this pool (profiles/r02_f64_rate.txt); data sheet %.1f" % F64_SPEC_TFLOPS,
            "f64_flops_per_step": stats["f64_flops"], "f64_lane_insts_per_step": stats["f64_insts"],
            "pipe_occupancy": ti / F64_PEAK_TINSTS,
            "pipe_occupancy_note": "float64 vector instructions (a multiplication holds the pipe as long as a fused multiply-add) / %.1f T lane instructions/s measured" % F64_PEAK_TINSTS,
            "hbm": {"algorithmic_bytes_per_step": alg_bytes, "GBps": alg_bytes / (ms * 1e-3) / 1e9, "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "traffic": None, "traffic_source": "profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --batch 64 --multiset)"}}
TRAFFIC_FILE = os.path.join("profiles", "r06_hbm_traffic.json")
TRAFFIC_CASES_FILE = os.path.join("profiles", "r06_hbm_traffic_cases.json")


def case_traffic(case, source_id):
    """HBM bytes per step of one of the other configs, from the committed rocprofv3 PMC passes (tools/collect_profiles.sh) - quoted
    only when that file was measured on THIS build of the library.  (bytes or None, where from)"""
    try:
        with open(os.path.join(ROOT, TRAFFIC_CASES_FILE)) as fh:
            prof = json.load(fh)
        if prof.get("source_id") != source_id:
            return None, "%s is of library build %s, the running one is %s: not quoted" % (TRAFFIC_CASES_FILE, prof.get("source_id"), source_id)
        rec = prof["cases"][case]
        return rec["hbm_bytes_per_step"], "%s, measured on library build %s (= the running one; separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2)" % (TRAFFIC_CASES_FILE, source_id)
    except (OSError, KeyError, ValueError, TypeError):
        return None, None
REF_OVER_PORT_FILE = os.path.join("tests", "golden", "ref_over_port.json")


def reference_over_port():
    """Wall time of the unmodified reference / wall time of the port (oracle.beliefs_refshaped) on the full config-4
    input, as measured in the build container by oracle/time_reference.py (the reference cannot travel to the GPU
    box).  The port is the faster one: the CPU baseline FLATTERS the CPU."""
    try:
        with open(os.path.join(ROOT, REF_OVER_PORT_FILE)) as fh:
            rec = json.load(fh)
        return float(rec["reference_over_port_time"]), REF_OVER_PORT_FILE + " (oracle/time_reference.py, build container)"
    except (OSError, KeyError, ValueError):
        return None, None


class _stdout_to_stderr:
    """RCCL prints a version banner on file descriptor 1 when a communicator is made; this process's standard output is ONE JSON
    line - whatever a library writes to fd 1 inside this block goes to standard error instead."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


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


def cpu_baseline(spec, itemsize, what):
    """Reference-shaped numpy path (the reference's 5N-1 einsum sequence) on one core over `spec`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import jt_oracle as oracle
    from junctiontree_amd import synthetic
    pots = synthetic.potentials_for(spec, seed=1, dtype=np.float32 if itemsize == 4 else np.float64)
    t0 = time.perf_counter()
    c0 = time.process_time()
    beliefs = oracle.beliefs_refshaped(spec["tree"], pots, spec["node_vars"])
    wall = time.perf_counter() - t0
    cpu = time.process_time() - c0
    ab = synthetic.algorithmic_bytes(spec, itemsize)
    ratio, ratio_src = reference_over_port()
    return {
        "value": ab["total"] / wall / 1e9, "unit": "GB/s",
        "messages_per_sec": ab["messages"] / wall,
        "cores": 1, "kind": "port", "Z": float(np.sum(beliefs[0], dtype=np.float64)),
        "cpu_model": _cpu_model(), "host_cores": os.cpu_count(),
        "reference_over_port_time": ratio, "reference_over_port_source": ratio_src,
        "reference_equivalent_value": ab["total"] / wall / 1e9 / ratio if ratio else None,
        "sample": "%s; one propagate, %.1f s wall, cpu/wall %.2f" % (what, wall, cpu / max(wall, 1e-9)),
    }


def cpu_baseline_lattice(h, w, card):
    """configs[2]: the oracle's `propagate` (evaluate + the reference's einsum sequence + marginalize) on a bounded
    lattice of the same height and cardinality, junction tree by this repo's builder."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import jt_oracle as oracle
    import junctiontree_amd as jt
    from junctiontree_amd import synthetic
    factors, sizes, values = synthetic.lattice_mrf(h, w, card)
    tree = jt.create_junction_tree(factors, sizes)
    ct = tree.clique_tree
    t0 = time.perf_counter()
    out = oracle.propagate(tree.tree, tree.separators, ct.maxcliques, ct.factor_to_maxclique, factors, sizes, values)
    wall = time.perf_counter() - t0
    tables = sum(int(np.prod([sizes[v] for v in c])) for c in ct.maxcliques) * 4
    seps = sum(int(np.prod([sizes[v] for v in s])) if len(s) else 1 for s in tree.separators) * 8
    total = 3 * tables + 5 * seps
    return {
        "value": total / wall / 1e9, "unit": "GB/s", "messages_per_sec": 2 * (len(ct.maxcliques) - 1) / wall,
        "cores": 1, "kind": "port", "Z": float(np.sum(out[0])), "cpu_model": _cpu_model(), "host_cores": os.cpu_count(),
        "sample": "%d x %d lattice of cardinality %d (%d cliques, %.0f MB of tables), oracle.propagate, one core, %.1f s wall"
                  % (h, w, card, len(ct.maxcliques), tables / 1e6, wall),
    }


def cpu_baseline_all_cores(width, sep, card, n_sample=16):
    """BASELINE configs[4] on the host: evidence sets are independent, so the CPU runs P of them at
    once, one process per core, each one propagate of a bounded sample tree of the same clique shape.  Called BEFORE
    this process makes its first HIP call (a process that has touched the GPU must not start others)."""
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


# ---------------------------------------------------------------------------------------------------------------
# The other BASELINE configs, measured in the same process after the headline workload (`configs` in the JSON line).

def _timed(plan, steps, warmup=3):
    """(wall ms per step, device ms per step) of `steps` propagates: one hipEvent pair around the region."""
    for _ in range(warmup):
        plan.propagate(sync=False)
    plan.sync()
    t0 = time.perf_counter()
    plan.region_begin()
    for _ in range(steps):
        plan.propagate(sync=False)
    dev_ms = plan.region_end()
    plan.sync()
    return (time.perf_counter() - t0) / steps * 1e3, dev_ms / steps


def _sub_result(workload, alg_bytes, messages, wall_ms, dev_ms, steps, parity, **more):
    out = {"workload": workload, "ms_per_step": wall_ms, "device_ms_per_step": dev_ms, "steps": steps,
           "algorithmic_bytes_per_step": alg_bytes, "value": alg_bytes / (wall_ms * 1e-3) / 1e9, "unit": "GB/s",
           "messages_per_sec": messages / (wall_ms * 1e-3), "frac": alg_bytes / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "bound": "hbm", "parity": parity}
    out.update(more)
    if "roofline" in out:
        out["bound"] = out["roofline"]["bound"]
        out["frac"] = out["roofline"]["frac"]
        out["frac_of_hbm"] = out["roofline"]["hbm"]["frac"]
    return out


def sub_c2(device, source_id, steps=20):
    """configs[1]: chain of 1000 cliques of 64^3 doubles.  Parity: calibration - the separator marginal of two adjacent
    cliques agrees wherever it is taken, every belief sums to Z (five places along the chain).  A single chain is a sequence of
    dependent hand-overs (latency-bound): `four_chains_in_flight` is the same workload with the chip given
    something to overlap - four evidence sets, one stream each, ticket order."""
    import numpy as np
    from junctiontree_amd import engine, synthetic
    spec = synthetic.chain_tree(n_cliques=1000, card=64, width=3)
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", device=device)
    try:
        plan.fill_synthetic(1, spec["scales"])
        wall, dev = _timed(plan, steps)
        alg = synthetic.algorithmic_bytes(spec, 8)
        z = plan.z()
        worst = 0.0
        for c in (0, 250, 499, 750, 998):
            a, b = plan.marginals([(c, [c + 1, c + 2]), (c + 1, [c + 1, c + 2])])
            worst = max(worst, float(np.max(np.abs(a - b)) / np.max(np.abs(a))), abs(float(a.sum()) - z) / abs(z))
        launches = plan.stats()["n_launches"]
    finally:
        plan.close()
    four = None
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", device=device, n_batch=4)
    try:
        for b in range(4):
            plan.fill_synthetic(1 + b, spec["scales"], batch=b)
        w4, d4 = _timed(plan, max(steps // 2, 5))
        four = {"workload": "four such chains (evidence sets with their own tables) in flight, one HIP stream each", "ms_per_step": w4,
                "ms_per_chain": w4 / 4, "value": 4 * alg["total"] / (w4 * 1e-3) / 1e9, "unit": "GB/s",
                "frac": 4 * alg["total"] / (w4 * 1e-3) / 1e9 / HBM_PEAK_GBPS, "messages_per_sec": 4 * alg["messages"] / (w4 * 1e-3)}
    finally:
        plan.close()
    sample = synthetic.chain_tree(n_cliques=100, card=64, width=3)
    traffic, src = case_traffic("c2", source_id)
    return _sub_result("BASELINE.json configs[1]: chain of 1000 cliques, width 3, cardinality 64, float64", alg["total"], alg["messages"],
                       wall, dev, steps, {"kind": "calibration", "rel_err": worst, "tolerance": 1e-9, "ok": bool(worst <= 1e-9)},
                       dtype="f64", Z=z, launches_per_step=launches, four_chains_in_flight=four,
                       cpu_baseline=cpu_baseline(sample, 8, "chain of 100 of the 1000 cliques, same clique shape"),
                       traffic=traffic, traffic_source=src)


def sub_c3(device, source_id, steps=10, api_calls=5, lattice_w=167):
    """configs[2] as restated in SURVEY.md 8d (6 x 167 lattice MRF, cardinality 8, float32), twice: the hot path
    (collect + distribute with the clique potentials resident) and the whole API call tree.propagate(values) with every
    factor table new (H2D of the factor tables, evaluate, collect + distribute, factor marginals, D2H)."""
    import numpy as np
    import junctiontree_amd as jt
    from junctiontree_amd import engine, synthetic
    factors, sizes, values = synthetic.lattice_mrf(6, lattice_w, 8)
    t0 = time.perf_counter()
    tree = jt.create_junction_tree(factors, sizes)
    t_build = time.perf_counter() - t0
    ct = tree.clique_tree
    engine.clear_plan_cache()
    t0 = time.perf_counter()
    out = tree.propagate(values)
    t_first = time.perf_counter() - t0
    # two plans of one tree: the API's (round 6: it is told the factor marginals `propagate` returns, and forms those of cliques without a
    # table inside its launch) and the one `compute_beliefs` makes of the same tree - collect + distribute and nothing else: the hot path
    plan_api = tree.plan("f32")
    plan = tree.plan("f32", fold=False)
    plan.stage_factors(ct.factor_graph.factors, ct.factor_to_maxclique, values)
    try:
        n = len(ct.maxcliques)
        tables = sum(int(np.prod([sizes[v] for v in c])) for c in ct.maxcliques) * 4
        seps = sum(int(np.prod([sizes[v] for v in sp])) if len(sp) else 1 for sp in tree.separators) * 8
        root_table = int(np.prod([sizes[v] for v in ct.maxcliques[plan.root]])) * 4
        # Two byte counts.  SURVEY.md 8d to the letter: every clique at its FULL shape, read in both passes and its belief written
        # (what rounds 1-4 streamed).  And what the problem needs: a clique's potential at the shape its factors COVER (the reference
        # never materialises the other axes, junctiontree.py:52-61; a clique without factors has none), read once per pass, no
        # belief tables (propagate returns factor marginals, :264-274), plus the messages - the engine's own accounting
        # (jtp_stats.algorithmic_bytes), which since round 5 is what it moves.
        alg_full = 3 * tables - root_table + 5 * seps
        st0 = plan.stats()
        alg = st0["algorithmic_bytes"]
        wall, dev = _timed(plan, steps)
        z = plan.z()
        # every factor table new on every call; the caller SAYS so (`changed="all"`, round 6: nothing is compared, the factor lists are
        # not looked at again) - and, beside it, the call that leaves the comparison to the library (one vectorised pass over all tables)
        api, api_cmp = [], []
        import gc
        gc.collect()
        gc.disable()          # (as timeit does: a collection of this process's whole heap - four configs' worth of objects - is not the call's)
        for r in range(api_calls):
            vals = [v * np.float32(1.0 + 1e-3 * (r + 1)) for v in values]
            plan.sync()
            t0 = time.perf_counter()
            out = tree.propagate(vals, changed="all")
            api.append((time.perf_counter() - t0) * 1e3)
        staged = plan_api.staged_cliques
        for r in range(api_calls):
            vals = [v * np.float32(1.0 + 2e-3 * (r + 1)) for v in values]
            plan.sync()
            t0 = time.perf_counter()
            out = tree.propagate(vals)
            api_cmp.append((time.perf_counter() - t0) * 1e3)
        gc.enable()
        out = tree.propagate(values)
        z = plan_api.z()
        folded_wall, _ = _timed(plan_api, steps)
        sums = np.array([m.sum() for m in out])
        seen, worst = {}, 0.0
        for f, o in zip(factors, out):
            for ax, v in enumerate(f):
                m = o.sum(axis=1 - ax)
                if v in seen:
                    worst = max(worst, float(np.max(np.abs(m - seen[v]) / seen[v])))
                else:
                    seen[v] = m
        sum_err = float(np.max(np.abs(sums - z)) / z)
        parity = {"kind": "calibration", "factor_marginals_sum_to_Z_rel_err": sum_err, "calibration_rel_err": worst,
                  "tolerance": 5e-6, "ok": bool(sum_err <= 5e-6 and worst < 5e-6)}
        wl = ("BASELINE.json configs[2] as restated in SURVEY.md 8d: 6 x %d lattice MRF, %d pairwise factors, cardinality 8, float32; "
              "junction tree by this repo's builder: %d cliques, max width %d" % (lattice_w, len(factors), n, max(len(c) for c in ct.maxcliques)))
        traffic, src = case_traffic("c3", source_id)
        d = plan.describe()
        hot = _sub_result(wl, alg, 2 * (n - 1), wall, dev, steps, parity, dtype="f32", Z=z, junction_tree_build_s=t_build,
                          first_call_s=t_first, launches_per_step=plan.stats()["n_launches"],
                          bytes_definition="covered-shape potentials (static tables of the cliques that hold factors, read once per pass) + messages; no belief tables",
                          full_shape={"algorithmic_bytes_per_step": alg_full, "value": alg_full / (wall * 1e-3) / 1e9, "unit": "GB/s",
                                      "frac": alg_full / (wall * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                      "definition": "SURVEY.md 8d to the letter: every clique at its full shape, read in both passes, belief written"},
                          cliques_without_table=st0["n_unit_cliques"], static_tables=st0["n_static_tables"],
                          device_table_bytes=int(d["arena_elems"]) * 4 + int(st0["fixed_bytes"]), full_shape_table_bytes=tables,
                          traffic=traffic, traffic_source=src, cpu_baseline=cpu_baseline_lattice(6, 20, 8))
        # the column-sweep tree of SURVEY.md 8d beside it (one clique per eliminated variable, 2-3 of 7 variables covered)
        try:
            t0 = time.perf_counter()
            sweep = jt.create_junction_tree(factors, sizes, order=synthetic.lattice_column_order(6, lattice_w))
            sweep.propagate(values)
            sp_plan = sweep.plan("f32", fold=False)
            sp_plan.stage_factors(sweep.clique_tree.factor_graph.factors, sweep.clique_tree.factor_to_maxclique, values)
            sw_wall, sw_dev = _timed(sp_plan, max(steps // 2, 3))
            sst = sp_plan.stats()
            hot["column_sweep_tree"] = {"cliques": len(sweep.clique_tree.maxcliques), "ms_per_step": sw_wall, "device_ms_per_step": sw_dev,
                                        "algorithmic_bytes_per_step": sst["algorithmic_bytes"], "cliques_without_table": sst["n_unit_cliques"],
                                        "full_shape_algorithmic_bytes_per_step": sst["algorithmic_bytes_full"], "Z": sp_plan.z(),
                                        "setup_s": time.perf_counter() - t0}
        except Exception as exc:          # noqa: BLE001 - recorded, not raised
            hot["column_sweep_tree"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        api_ms = min(api)
        # the API call moves, beyond the hot path's bytes: the tables of the cliques that hold factors written once more
        # (evaluate; the others stay all ones) and their beliefs read once more (marginalize: the requests on one clique
        # share the pass); the factor tables in and the factor marginals out are under 2 MB
        # beyond the hot path's bytes: the potentials written once (evaluate) and read once more by the marginal passes
        api_alg = alg + 2 * (int(d["arena_elems"]) * 4 + int(st0["fixed_bytes"]))
        e2e = {"workload": wl + "; tree.propagate(values, changed=\"all\") with all %d factor tables new: H2D, evaluate (%d cliques formed), collect + "
                                 "distribute, %d factor marginals, D2H" % (len(factors), staged, len(factors)),
               "ms_per_step": api_ms, "ms_per_step_median": sorted(api)[len(api) // 2], "steps": api_calls,
               "ms_per_step_tables_compared_by_the_library": min(api_cmp),
               "propagate_of_the_api_plan_ms": folded_wall,
               "folded_marginal_tasks": sum(1 for t in plan_api.describe()["tasks"] if t.get("fold")),
               "algorithmic_bytes_per_step": api_alg, "value": api_alg / (api_ms * 1e-3) / 1e9, "unit": "GB/s",
               "frac": api_alg / (api_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "bound": "hbm", "hot_path_share": wall / api_ms,
               "d2h_bytes": int(sum(o.nbytes for o in out)), "parity": parity}
        return hot, e2e
    finally:
        engine.clear_plan_cache()


def sub_c5(device, spec, source_id, n_sets=64, steps=10, oracle_sets=1):
    """configs[4] on one device = one rank's share of the 512 evidence sets: 64 sets over the shared width-20 tables
    (JTP_MULTISET).  Parity: Z of `oracle_sets` sets against the numpy oracle on indicator-multiplied potentials, and a
    marginal of every set sums to that set's Z."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import jt_oracle as oracle
    from junctiontree_amd import engine, synthetic
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", device=device, n_batch=n_sets, multiset=True)
    try:
        plan.fill_synthetic(1, spec["scales"])
        labels = sorted(spec["sizes"])
        evidence = []
        for b in range(n_sets):
            rng = np.random.default_rng(1000 + b)
            ev = {labels[i]: int(rng.integers(0, spec["sizes"][labels[i]])) for i in rng.choice(len(labels), size=16, replace=False)}
            evidence.append(ev)
            plan.set_evidence(ev, batch=b)
        wall, dev = _timed(plan, steps)
        n = spec["n_cliques"]
        sz = [1] * len(spec["node_vars"])
        for i, labs in enumerate(spec["node_vars"]):
            for v in labs:
                sz[i] *= spec["sizes"][v]
        tables, seps = sum(sz[:n]) * 4, sum(sz[n:]) * 8
        alg = 2 * tables + 5 * seps * n_sets             # tables once per batch (collect + distribute), messages per set
        zs = [plan.z(b) for b in range(n_sets)]
        worst = 0.0
        for b in range(n_sets):
            m = plan.marginals([(n - 1 - (b % 8), [spec["node_vars"][n - 1 - (b % 8)][0]])], batch=b)[0]
            worst = max(worst, abs(float(m.sum()) - zs[b]) / abs(zs[b]))
        zerr = 0.0
        t0 = time.perf_counter()
        for b in range(oracle_sets):
            pots = synthetic.potentials_for(spec, seed=1, dtype=np.float32)
            done = set()
            for c in range(n):                            # the indicator goes into the first clique that holds the variable
                for ax, v in enumerate(spec["node_vars"][c]):
                    if v in evidence[b] and v not in done:
                        done.add(v)
                        ind = np.zeros(spec["sizes"][v])
                        ind[evidence[b][v]] = 1.0
                        shape = [1] * pots[c].ndim
                        shape[ax] = -1
                        pots[c] = pots[c] * ind.reshape(shape)
            _, zo = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
            zerr = max(zerr, abs(zs[b] - zo) / abs(zo))
        t_oracle = time.perf_counter() - t0
        st = plan.stats()
        parity = {"kind": "Z of %d evidence set(s) vs oracle (indicator-multiplied potentials, %.1f s of numpy); every set: a marginal sums to its Z" % (oracle_sets, t_oracle),
                  "Z_rel_err": zerr, "marginal_sums_rel_err": worst, "tolerance": 1e-6, "ok": bool(zerr <= 1e-6 and worst <= 1e-6)}
        roof = multiset_roofline(st, dev, alg)
        roof["hbm"]["traffic"], roof["hbm"]["traffic_source"] = case_traffic("multiset64", source_id)
        # the CPU beside it: the oracle run above IS one evidence set of this workload on one host core
        per_set = (2 * tables + 5 * seps)
        cpu = {"value": per_set * oracle_sets / t_oracle / 1e9, "unit": "GB/s", "messages_per_sec": 2 * (n - 1) * oracle_sets / t_oracle,
               "cores": 1, "kind": "port", "cpu_model": _cpu_model(), "host_cores": os.cpu_count(),
               "sample": "%d evidence set(s) of this workload (oracle.beliefs_exact on indicator-multiplied potentials), %.1f s wall" % (oracle_sets, t_oracle)}
        return _sub_result("BASELINE.json configs[4], one rank's share: %d evidence sets (16 observed variables each) over the shared tables of the "
                           "width-20 tree, JTP_MULTISET" % n_sets, alg, 2 * (n - 1) * n_sets, wall, dev, steps, parity, dtype="f32",
                           evidence_sets_per_step=n_sets, ms_per_evidence_set=wall / n_sets, roofline=roof,
                           engine_table_bytes_per_step=st["algorithmic_bytes"], launches_per_step=st["n_launches"], cpu_baseline=cpu)
    finally:
        plan.close()


def sub_rank_share(device, spec, world=8, steps=30, single_ms=None):
    """What one GPU can say about the 8-GPU run of configs[3] (no 8-GPU node has been available to the driver): every rank's share
    of the partitioned tree timed ALONE on this GPU, the exchange at the cuts replaced by fills of the receive buffers
    (JTP_FAKE_COMM).  A projection aid - the slowest share bounds the sharded step from below, the RCCL exchange comes on top -
    not a multi-GPU measurement; `value` of a real `--gpus 8` run is what counts."""
    from junctiontree_amd import engine, partition
    n = spec["n_cliques"]
    root, _, owner = partition.partition_tree(spec["parent"], [1.0] * n, world, replicate_top=True)
    old = os.environ.get("JTP_FAKE_COMM")
    per_rank, per_rank_rccl, n_ops = [], [], []

    def shares(mode, out):
        os.environ["JTP_FAKE_COMM"] = mode
        for rank in range(world):
            plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", device=device, n_ranks=world, rank=rank,
                               owner=owner, root=root)
            try:
                plan.fill_synthetic(1, spec["scales"])
                # (the smaller of two regions of `steps` propagates: one share of eight at 463 us beside seven at 177-181 us was seen
                #  once in a driver-form run - a share runs 5 ms in all, anything else on the box shows)
                dev = min(_timed(plan, steps)[1] for _ in range(2))
                out.append(dev * 1e3)
                if mode == "2":
                    n_ops.append((sum(1 for op in plan.describe()["comm"] if op["send"]), sum(1 for op in plan.describe()["comm"] if not op["send"])))
            finally:
                plan.close()

    loop_err = None
    try:
        shares("1", per_rank)
        # ... and the same shares with the exchange step as what a sharded run issues there: ONE RCCL group of the step's
        # ncclSend / ncclRecv calls on the plan's stream, between the collect launch and the merged launch - in loop-back (a
        # communicator of one rank, every call addressed to this rank itself): the group's own cost on this GPU, no wire
        import ctypes as C
        from junctiontree_amd import _capi
        lib = _capi.lib()
        try:
            buf = C.create_string_buffer(128)
            with _stdout_to_stderr():
                _capi.check(lib.jtp_comm_unique_id(buf))
                _capi.check(lib.jtp_comm_init(0, 1, C.c_char_p(buf.raw), device))
            try:
                shares("2", per_rank_rccl)
            finally:
                with _stdout_to_stderr():
                    lib.jtp_comm_destroy()
        except Exception as exc:          # noqa: BLE001 - recorded, not raised
            loop_err = "%s: %s" % (type(exc).__name__, exc)
    finally:
        if old is None:
            os.environ.pop("JTP_FAKE_COMM", None)
        else:
            os.environ["JTP_FAKE_COMM"] = old
    out = {"workload": "BASELINE.json configs[3] cut for %d ranks (top part replicated): each rank's share run alone on ONE GPU, exchanges "
                       "replaced by fills (JTP_FAKE_COMM=1) and by the real RCCL group in loop-back (JTP_FAKE_COMM=2) - a projection aid, not a "
                       "multi-GPU measurement" % world,
           "device_us_per_rank_share": per_rank, "slowest_share_us": max(per_rank), "steps": steps, "regions_per_share": 2,
           "cliques_per_rank": [sum(1 for o in owner if o in (r, world)) for r in range(world)]}
    if per_rank_rccl:
        diffs = sorted(b - a for a, b in zip(per_rank, per_rank_rccl))
        out.update({"device_us_per_rank_share_rccl_loopback": per_rank_rccl, "slowest_share_rccl_loopback_us": max(per_rank_rccl),
                    "rccl_group_us_loopback": diffs[len(diffs) // 2],
                    "rccl_group_note": "median over the ranks of (share with the exchange as ONE grouped %d x ncclSend + %d x ncclRecv of 8 KiB to self on the "
                                       "plan's stream) - (share with fills): what the group costs on this GPU; the xGMI hop of a real run comes on top"
                                       % (n_ops[0][0], n_ops[0][1])})
        if single_ms:
            out["projected_speedup_at_%d_ranks" % world] = {"with_fills": single_ms * 1e3 / max(per_rank), "with_rccl_group_loopback": single_ms * 1e3 / max(per_rank_rccl),
                                                            "single_gpu_ms_per_step": single_ms, "target": 3.5}
    if loop_err:
        out["rccl_loopback_error"] = loop_err
    return out


def sub_configs(device, spec_c4, source_id, single_ms):
    """Run the sub-configs one after the other; a failure is recorded in its place, never raised (the headline line stands)."""
    out = {}
    t_all = time.perf_counter()

    def guard(name, fn):
        t0 = time.perf_counter()
        try:
            res = fn()
        except Exception as exc:          # noqa: BLE001 - recorded, not raised
            res = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if isinstance(res, tuple):
            for k, r in zip(name, res):
                out[k] = r
        elif isinstance(name, tuple):
            for k in name:
                out[k] = res
        else:
            out[name] = res
        return time.perf_counter() - t0

    secs = {"c2": guard("c2", lambda: sub_c2(device, source_id)),
            "c3": guard(("c3", "c3_api_end_to_end"), lambda: sub_c3(device, source_id)),
            "c5_multiset64": guard("c5_multiset64", lambda: sub_c5(device, spec_c4, source_id)),
            "c4_rank_share_of_8": guard("c4_rank_share_of_8", lambda: sub_rank_share(device, spec_c4, single_ms=single_ms))}
    out["wall_s"] = dict(secs, total=time.perf_counter() - t_all)
    return out



def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N rank processes from HERE - this process never loads
    libjtprop.so and never touches a GPU - wait for them, pass rank 0's JSON line on, and fail if any rank fails or
    the run exceeds --spawn-timeout.  Children are killed by their exact PIDs."""
    import socket
    import subprocess
    import tempfile
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               JTP_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out0 = tempfile.TemporaryFile(mode="w+")
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + args.spawn_timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = 1
            print("bench.py: rank %d exited with code %d" % bad[0], file=sys.stderr)
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            rc = 1
            print("bench.py: ranks still running after %d s (--spawn-timeout)" % args.spawn_timeout, file=sys.stderr)
            break
        time.sleep(0.1)
    for p in procs:
        if p.poll() is None:
            p.kill()
        p.wait()
    out0.seek(0)
    text = out0.read()
    sys.stdout.write(text)
    sys.stdout.flush()
    if rc == 0 and not any(line.startswith("{") for line in text.splitlines()):
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    return rc


def main():
    faulthandler.enable()              # (a fault inside the library says where, instead of ending the run without a word)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # 200 x 0.65 ms: the timed region is not inside box-to-box noise
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c4", choices=["c4", "c2", "c3"],
                    help="c4 = BASELINE configs[3] (default, the metric's workload); c2 = configs[1]: chain of 1000 cliques, "
                         "width 3, cardinality 64, float64 (latency-bound); c3 = configs[2] as restated in SURVEY.md 8d: "
                         "6 x 167 lattice MRF, cardinality 8, float32, junction tree by this repo's builder")
    ap.add_argument("--batch", type=int, default=1,
                    help="independent evidence sets per step, one HIP stream each (BASELINE configs[4] in "
                         "miniature; default 1 = the metric's workload)")
    ap.add_argument("--share", action="store_true",
                    help="with --batch: the evidence sets share one set of clique tables (JTP_SHARE_POTENTIALS) and "
                         "differ by hard evidence on 16 variables each (SURVEY 8d, config 5)")
    ap.add_argument("--multiset", action="store_true",
                    help="with --batch: JTP_MULTISET plan - the evidence sets share the tables and every pass over a table "
                         "serves a group of sets (no belief tables; beliefs and marginals are formed on demand)")
    ap.add_argument("--no-replicate-top", action="store_true",
                    help="N > 1: give the top part of the partition to one rank instead of replicating it on all")
    ap.add_argument("--cliques", type=int, default=256)
    ap.add_argument("--width", type=int, default=20)
    ap.add_argument("--sep", type=int, default=10)
    ap.add_argument("--card", type=int, default=2)
    ap.add_argument("--lattice-w", type=int, default=167, help="--config c3: lattice columns (6 rows)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="size of the CPU baseline's sample: cliques of the tree (c4: default 256 = the full workload, ~4-25 s "
                         "on one core; c2: default 100 of the 1000 cliques) or lattice columns (c3: default 60); 0 = skip (and skip the sub-results "
                         "for the other configs: measurement tools run the headline workload alone)")
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
    ap.add_argument("--idle-plans", type=int, default=0,
                    help="create this many other (small, idle) device plans first: A/B for the launch-order rule - an idle "
                         "plan must not cost the benchmark plan its blockIdx-order launches")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the sub-results for the other BASELINE configs (c2, c3, c3 API call, c5 multi-set) that the default "
                         "one-GPU run adds to its line")
    ap.add_argument("--spawn-timeout", type=int, default=900, help="seconds the self-started rank processes may take")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.config == "c3" and world != 1:
        raise SystemExit("--config c3 runs on one GPU")
    if args.config == "c3" and (args.batch != 1 or args.share or args.multiset or args.split_variants or args.dtype != "f32"):
        raise SystemExit("--config c3 is one float32 evidence set: --batch / --share / --multiset / --split-variants / --dtype do not apply")
    # Evidence sets are independent (BASELINE configs[4]: "512 sets, 8 MI355X"): with N ranks every rank takes ITS OWN --batch
    # sets over its own copy of the tables - replicas only, no exchange on the data path; the line's value is the sum over the
    # ranks and "scaling" is weak (per-GPU work fixed).
    replicas = (args.share or args.multiset) and world > 1

    # host-only work that starts processes: before this process makes its first HIP call
    all_cores = None
    if args.cpu_all_cores and rank == 0 and world == 1:
        all_cores = cpu_baseline_all_cores(args.width, args.sep, args.card)

    import ctypes as C
    import numpy as np
    from junctiontree_amd import _capi, engine, partition, synthetic
    lib = _capi.lib()                               # loads libjtprop.so (and its HIP runtime) first
    ndev = _capi.device_count()
    if ndev <= 0:
        raise SystemExit("no HIP device visible: bench.py measures the GPU path only")
    if world > ndev and not os.environ.get("JTP_RCCL_LIB"):
        # one process per GPU: RCCL refuses two ranks on one device ("invalid usage", duplicate GPU) - say what is wrong
        # instead.  (JTP_RCCL_LIB = a stand-in transport, tests/mock_rccl: ranks may then share a GPU.)
        if rank == 0:
            print("bench.py: %d ranks need %d devices, found %d (one process per GPU over RCCL; HIP_VISIBLE_DEVICES / "
                  "ROCR_VISIBLE_DEVICES limit what a process sees)" % (world, world, ndev), file=sys.stderr, flush=True)
        raise SystemExit(3)
    device = local_rank % ndev                      # one process per GPU (ranks > GPUs only over the mock transport)
    if world > ndev:                                # ranks sharing a GPU: dataflow launches need ticket order
        os.environ["JTP_FLOW_TICKETS"] = "1"
    version = lib.jtp_version().decode()
    source_id = version.rsplit("src:", 1)[-1] if "src:" in version else "unknown"

    from junctiontree_amd.rendezvous import Rendezvous
    master_port = int(os.environ.get("MASTER_PORT", "29500"))
    run_id = "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "none") if ch.isalnum())[:32]
    rdzv = Rendezvous(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), master_port + 1,
                      port_file="/tmp/jtp_rdzv_%d_%s_%d" % (master_port, run_id, os.getppid()))
    if world > 1:
        uid = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            with _stdout_to_stderr():
                _capi.check(lib.jtp_comm_unique_id(buf))
            uid = buf.raw
        uid = rdzv.broadcast(uid)
        with _stdout_to_stderr():
            _capi.check(lib.jtp_comm_init(rank, world, C.c_char_p(uid), device))

    def barrier():
        rdzv.barrier()

    idle = []
    for i in range(args.idle_plans):
        sp = synthetic.wide_binary_tree(n_cliques=7, width=12, sep=6, card=2, seed=10 + i)
        p = engine.Plan(sp["tree"], sp["node_vars"], sp["sizes"], dtype="f32", device=device)
        p.fill_synthetic(1, sp["scales"])
        p.propagate()                               # has run, has been waited for: idle
        idle.append(p)

    lattice = None
    if args.config == "c3":
        import junctiontree_amd as jt
        args.dtype = "f32"
        t0 = time.perf_counter()
        factors, sizes, values = synthetic.lattice_mrf(6, args.lattice_w, 8)
        # (JTP_BENCH_C3_SWEEP=1: the column-sweep tree of SURVEY.md 8d instead of this repo's min-fill tree)
        tree = jt.create_junction_tree(factors, sizes, order=synthetic.lattice_column_order(6, args.lattice_w) if os.environ.get("JTP_BENCH_C3_SWEEP") else None)
        t_build = time.perf_counter() - t0
        ct = tree.clique_tree
        node_vars = [list(c) for c in ct.maxcliques] + [list(s) for s in tree.separators]
        # (cover: which variables of each clique its factors cover - the others are never materialised, junctiontree.py:52-61)
        plan = engine.Plan(tree.tree, node_vars, sizes, dtype="f32", device=device, block_log2=args.block_log2,
                           lds_budget=args.lds_budget, layout_policy=args.layout_policy, level_launches=args.level_launches,
                           cover=None if os.environ.get("JTP_BENCH_NO_COVER") else tree.cover())
        plan.stage_factors(factors, ct.factor_to_maxclique, values)      # evaluate: every clique in one call, one launch
        plan.sync()
        n = len(ct.maxcliques)
        tables = sum(int(np.prod([sizes[v] for v in c])) for c in ct.maxcliques) * 4
        seps = sum(int(np.prod([sizes[v] for v in s])) if len(s) else 1 for s in tree.separators) * 8
        root_table = int(np.prod([sizes[v] for v in ct.maxcliques[plan.root]])) * 4
        alg_full = 3 * tables - root_table + 5 * seps        # SURVEY.md 8d to the letter: every clique at its full shape, belief written
        st0 = plan.stats()
        # what the problem needs (and the engine moves): potentials at the shape their factors cover, no belief tables, the messages
        alg = {"total": st0["algorithmic_bytes"], "write": 3 * seps, "messages": 2 * (n - 1)} if plan.cover is not None else \
              {"total": alg_full, "write": tables + 3 * seps, "messages": 2 * (n - 1)}
        alg["read"] = alg["total"] - alg["write"]
        lattice = {"tree": tree, "factors": factors, "values": values, "t_build": t_build, "alg_full": alg_full, "stats": st0,
                   "max_width": max(len(c) for c in ct.maxcliques)}
        spec = None
    else:
        if args.config == "c2":
            args.dtype = "f64"
            spec = synthetic.chain_tree(n_cliques=1000 if args.cliques == 256 else args.cliques,
                                        card=64 if args.card == 2 else args.card, width=3 if args.width == 20 else args.width)
        else:
            spec = synthetic.wide_binary_tree(n_cliques=args.cliques, width=args.width, sep=args.sep,
                                              card=args.card, seed=0)
        itemsize = 4 if args.dtype == "f32" else 8
        alg = synthetic.algorithmic_bytes(spec, itemsize)
        n = spec["n_cliques"]
        # the small top part of the partition is replicated on every rank (one exchange per propagate instead of two)
        # (re-rooted at the weighted centroid first, SURVEY.md 8e: the balanced tree of config 4 is rooted there already,
        #  a chain handed over with its end as the root is hung from its middle)
        part_root, _, owner = partition.partition_tree(spec["parent"], [1.0] * n, world, replicate_top=not args.no_replicate_top)
        if replicas:
            part_root, owner = None, None
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype=args.dtype,
                           device=device, n_ranks=1 if replicas else world, rank=0 if replicas else rank, owner=owner, n_batch=args.batch,
                           root=part_root if world > 1 and not replicas else None,
                           block_log2=args.block_log2, lds_budget=args.lds_budget,
                           layout_policy=args.layout_policy, split_variants=args.split_variants,
                           level_launches=args.level_launches, share_potentials=args.share, multiset=args.multiset)
        if args.share or args.multiset:
            plan.fill_synthetic(1, spec["scales"])
            labels = sorted(spec["sizes"])
            for b in range(args.batch):
                rng = np.random.default_rng(1000 + b + args.batch * rank)          # (rank r of N holds sets r * batch .. of the N * batch)
                plan.set_evidence({labels[i]: int(rng.integers(0, spec["sizes"][labels[i]]))
                                   for i in rng.choice(len(labels), size=16, replace=False)}, batch=b)
        else:
            for b in range(args.batch):
                plan.fill_synthetic(1 + b, spec["scales"], batch=b)
    itemsize = 4 if args.dtype == "f32" else 8

    for _ in range(args.warmup):
        plan.propagate(sync=False)
    plan.sync()
    if args.share or args.multiset:
        # Evidence sets that share their tables: a table need only be read ONCE per batch, whatever the engine does
        # (a --share plan streams it once per set, mostly out of the Infinity Cache; a --multiset plan once per group
        # of sets).  Algorithmic bytes: tables once per batch; per set its messages and - unless beliefs are
        # formed on demand (--multiset) - its belief tables.
        sz = [1] * len(spec["node_vars"])
        for i, labs in enumerate(spec["node_vars"]):
            for v in labs:
                sz[i] *= spec["sizes"][v]
        tables, seps = sum(sz[:spec["n_cliques"]]) * itemsize, sum(sz[spec["n_cliques"]:]) * 8
        per_set = 5 * seps + (0 if args.multiset else tables)
        alg = dict(alg, total=(2 * tables + per_set * args.batch) / args.batch, read=2 * tables / args.batch)
    # The timed region holds NO per-propagate events: one event before its first launch and one after its last
    # (jtp_region_begin / jtp_region_end) give the device time of the K steps, which cannot exceed the wall clock around them.
    plan.set_profiling(0)
    barrier()
    plan.sync()                                      # hipStreamSynchronize on the plan's stream
    t0 = time.perf_counter()
    plan.region_begin()
    for _ in range(args.steps):
        plan.propagate(sync=False)
    region_ms = plan.region_end()
    plan.sync()
    barrier()
    elapsed = rdzv.allreduce_max(time.perf_counter() - t0)      # max over ranks
    # How a step divides over its launches (plans of two launches: collect, distribute; per level with --per-launch): from a
    # few propagates timed with events AFTER the timed region - the events cost 2-3 us of idle GPU each, which is why they
    # are not in it.
    if not args.no_profile:
        n_prof = 8
        plan.set_profiling(n_prof, per_launch=args.per_launch or args.split_variants, stride=1)
        for _ in range(n_prof):
            plan.propagate(sync=False)
        plan.sync()

    stats = plan.stats()
    z = plan.z() if plan.owns(plan.root) else None
    belief_dev = None
    rccl_info = per_rank_ms = None
    if world > 1:
        # what RCCL itself says about the communicator, and every rank's own step time, gathered on rank 0
        import ctypes as C2
        nr, ur, cd = C2.c_int32(-1), C2.c_int32(-1), C2.c_int32(-1)
        lib.jtp_comm_info(C2.byref(nr), C2.byref(ur), C2.byref(cd))
        mine_rec = json.dumps({"rank": rank, "device": device, "ncclCommCount": nr.value, "ncclCommUserRank": ur.value, "ncclCommCuDevice": cd.value,
                               "ms_per_step": region_ms / args.steps}).encode()
        recs = [json.loads(x.decode()) for x in rdzv.allgather(mine_rec)]
        rccl_info = {"ncclCommCount": sorted({r["ncclCommCount"] for r in recs}), "ranks": [{k: r[k] for k in ("rank", "device", "ncclCommUserRank", "ncclCommCuDevice")} for r in recs]}
        per_rank_ms = [r["ms_per_step"] for r in recs]
    if replicas:
        # replicas: every rank checks its own first evidence set (a marginal sums to that set's Z), the worst goes into the line
        m = plan.marginals([(n - 1, [spec["node_vars"][n - 1][0]])], batch=0)[0]
        belief_dev = rdzv.allreduce_max(abs(float(m.sum()) - z) / abs(z))
    elif world > 1:                                  # the rank holding the root clique knows Z: share it
        z = rdzv.allreduce_max(z if z is not None else -1.0e300)
        # ... and every rank checks the DISTRIBUTE side of the sharded run on cliques of its own (Z only proves the
        # collect side): a clique belief sums to Z whatever the clique.  First, middle and last clique a rank owns, one
        # single-variable marginal each (formed on the device); the worst relative deviation over all ranks goes
        # into the line.
        mine = [c for c in range(n) if plan.owns(c)]
        picks = sorted({mine[0], mine[len(mine) // 2], mine[-1]}) if mine else []
        dev = 0.0
        if picks:
            for m in plan.marginals([(c, [spec["node_vars"][c][0]]) for c in picks]):
                dev = max(dev, abs(float(m.sum()) - z) / abs(z))
        belief_dev = rdzv.allreduce_max(dev)
    if args.per_launch and not args.no_profile and rank == 0:
        for L in plan.launch_ms():
            print("# %s level %d: %4d tasks %5d blocks %8.4f ms %7.0f GB/s" % (
                "collect   " if L["phase"] == 0 else "distribute", L["level"], L["ntasks"], L["nblocks"],
                L["ms"], L["alg_bytes"] / max(L["ms"], 1e-9) / 1e6), file=sys.stderr)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        nrep = world if replicas else 1                    # replicas: every rank ran the whole batch of ITS sets
        gbps = alg["total"] * args.batch * nrep * args.steps / elapsed / 1e9
        if args.config == "c2":
            workload = "BASELINE.json configs[1]: chain of %d cliques, width 3, cardinality 64, float64" % n
        elif args.config == "c3":
            workload = ("BASELINE.json configs[2] as restated in SURVEY.md 8d: 6 x %d lattice MRF, %d pairwise factors, cardinality 8, "
                        "float32; junction tree by this repo's builder: %d cliques, max width %d"
                        % (args.lattice_w, len(lattice["factors"]), n, lattice["max_width"]))
        else:
            workload = ("BASELINE.json configs[3]: %d cliques, width %d, cardinality %d (2^%d-entry %s potentials), %d shared "
                        "variables per edge, balanced binary tree" % (n, args.width, args.card, args.width, args.dtype, args.sep))
        out = {
            "metric": "clique-potential GB/s (algorithmic bytes per propagate / time; messages/sec alongside), "
                      + ("synthetic width-%d tree" % args.width if args.config == "c4" else "config %s" % args.config),
            "value": gbps, "unit": "GB/s",
            "messages_per_sec": alg["messages"] * args.batch * nrep * args.steps / elapsed,
            "read_GBps": alg["read"] * args.batch * nrep * args.steps / elapsed / 1e9,
            "frac_of_hbm_roofline": gbps / (HBM_PEAK_GBPS * world),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if replicas else "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": workload,
                "algorithmic_bytes_per_step": alg["total"] * args.batch * nrep, "messages_per_step": alg["messages"] * args.batch * nrep,
                "evidence_sets_per_step": args.batch * nrep, "evidence_sets_per_rank": args.batch, "shared_potentials": bool(args.share or args.multiset),
                "multiset": bool(args.multiset),
                "engine_table_bytes_per_step": stats["algorithmic_bytes"] if args.multiset else None,
                "parallelism": "1 GPU" if world == 1 else ("replicas x%d: every rank its own %d evidence sets over its own copy of the tables, no exchange" % (world, args.batch) if replicas else
                                                          "subtree-sharded x%d, RCCL send/recv at cuts%s" % (world, "" if args.no_replicate_top else ", top part replicated")),
                "launches_per_step": stats["n_launches"], "launch_mode": stats["launch_mode"],
                "idle_plans": args.idle_plans, "Z": z, "library": version,
            },
        }
        if not args.no_profile and stats["kernels"]:
            name, k = max(stats["kernels"].items(), key=lambda kv: kv[1]["ms"])
            per_launch_bytes = k["bytes"] / k["launches"]          # (profiled: evidence set 0 only)
            # the kernel's share of a step, from the evented propagates after the timed region; its time = that share of
            # the region's device time per step (one launch per step: share 1)
            all_ms = sum(kv["ms"] for kv in stats["kernels"].values())
            share = k["ms"] / all_ms if all_ms > 0 else 1.0
            step_dev_ms = region_ms / args.steps
            per_launch_ms = step_dev_ms * share / k["launches"]
            achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
            # HBM bytes per launch: NOT measured in this run - read from the committed rocprofv3 PMC passes of the same
            # command (tools/collect_profiles.sh), and only when that file was measured on THIS build of the library
            traffic = traffic_source = None
            try:
                with open(os.path.join(ROOT, TRAFFIC_FILE)) as fh:
                    prof = json.load(fh)
                default_run = (args.config == "c4" and args.cliques == 256 and args.width == 20 and world == 1
                               and args.dtype == "f32" and args.batch == 1)
                if default_run and prof.get("source_id") == source_id:
                    traffic = prof["kernels"]["void " + name]["hbm_bytes_per_launch"]
                    traffic_source = "%s, measured on library build %s (= the running one; separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2)" % (TRAFFIC_FILE, source_id)
                elif default_run:
                    traffic_source = "%s is of library build %s, the running one is %s: not quoted" % (TRAFFIC_FILE, prof.get("source_id"), source_id)
            except (OSError, KeyError, ValueError):
                pass
            out["roofline"] = {
                "bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                "launches_per_step": k["launches"], "avg_launch_ms": per_launch_ms,
                "device_ms_per_step": step_dev_ms, "kernel_share_of_step": share,
                "avg_launch_ms_note": "device time of the %d timed steps between ONE hipEvent pair on the plan's stream (no event between the steps) / steps%s"
                                      % (args.steps, "" if len(stats["kernels"]) == 1 and k["launches"] == 1 else
                                         " x this kernel's share of a step, the share from 8 propagates timed with events after the timed region"),
                "algorithmic_bytes_per_launch": per_launch_bytes,
                "rank0_kernels": {kn: {"ms_per_step": step_dev_ms * (kv["ms"] / all_ms if all_ms > 0 else 1.0), "launches": kv["launches"],
                                       "GBps": kv["bytes"] / max(step_dev_ms * (kv["ms"] / all_ms if all_ms > 0 else 1.0), 1e-12) / 1e6}
                                  for kn, kv in stats["kernels"].items()},
                "collect_ms": stats["collect_ms"], "distribute_ms": stats["distribute_ms"],
            }
            if args.multiset:
                hbm_line = {k: out["roofline"][k] for k in ("achieved", "frac", "traffic", "traffic_source")}
                out["roofline"] = dict(multiset_roofline(stats, step_dev_ms, alg["total"] * args.batch), hbm_engine_bytes=hbm_line,
                                       collect_ms=stats["collect_ms"], distribute_ms=stats["distribute_ms"])
            if world > 1:
                out["roofline"]["note_multi_rank"] = ("rank 0's figures; the phase spans of a sharded plan contain its exchange steps "
                                                      "(ncclSend/ncclRecv between the launches)")
            if any(kn.startswith("jt_propagate_flow") for kn in stats["kernels"]):
                # both phases ran inside ONE launch (plans whose messages are small beside their tables): there is no
                # boundary between the phases to put an event on
                out["roofline"]["collect_ms"] = out["roofline"]["distribute_ms"] = None
                out["roofline"]["phases_in_one_launch"] = True
        cpu_n = args.cpu_sample if args.cpu_sample >= 0 else {"c4": n, "c2": 100, "c3": 60}[args.config]
        if cpu_n > 0 and world == 1:
            if args.config == "c4":
                sample = synthetic.wide_binary_tree(n_cliques=cpu_n, width=args.width, sep=args.sep, card=args.card, seed=0)
                out["cpu_baseline"] = cpu_baseline(sample, 4, "%d-clique balanced binary tree, same clique shape (width %d, card %d, %d shared)%s"
                                                   % (cpu_n, args.width, args.card, args.sep, " = the full workload" if cpu_n == n else ""))
                if cpu_n == n and z is not None and args.batch == 1 and not (args.share or args.multiset):      # same tree, same values: same Z
                    zc = out["cpu_baseline"]["Z"]
                    out["parity"] = {"Z_gpu": z, "Z_cpu_oracle": zc, "rel_err": abs(z - zc) / abs(zc)}
            elif args.config == "c2":
                sample = synthetic.chain_tree(n_cliques=min(cpu_n, n), card=spec["sizes"][spec["node_vars"][0][0]], width=len(spec["node_vars"][0]))
                out["cpu_baseline"] = cpu_baseline(sample, 8, "chain of %d of the %d cliques, same clique shape" % (min(cpu_n, n), n))
            else:
                out["cpu_baseline"] = cpu_baseline_lattice(6, cpu_n, 8)
            if all_cores is not None:
                out["cpu_baseline_all_cores"] = all_cores
        default_c4 = (args.config == "c4" and args.cliques == 256 and args.width == 20 and args.sep == 10
                      and args.card == 2 and args.dtype == "f32" and not (args.share or args.multiset))
        if default_c4 and z is not None:
            # Z of the default workload as one GPU and the numpy oracle compute it: a sharded run must agree
            out["config"]["Z_expected"] = Z_DEFAULT_C4
            out["config"]["Z_rel_err"] = abs(z - Z_DEFAULT_C4) / Z_DEFAULT_C4
        if world > 1:
            out["config"]["rccl"] = rccl_info
            out["config"]["ms_per_step_per_rank"] = per_rank_ms
        if replicas:
            out["config"]["replica_marginal_sums_rel_err"] = belief_dev       # (every rank: a marginal of its first evidence set vs that set's Z)
            out["config"]["multi_gpu_note"] = "replicas only: RCCL is initialised (the communicator's size is in config.rccl) and carries no data"
        elif world > 1:
            out["config"]["sharded_belief_sums_rel_err"] = belief_dev      # (three cliques per rank: sum of the belief vs Z)
            out["config"]["multi_gpu_note"] = ("transport: %s" % os.environ["JTP_RCCL_LIB"] if os.environ.get("JTP_RCCL_LIB")
                                               else "transport: RCCL (librccl.so.1), ncclSend/ncclRecv grouped per cut level")
        if args.share or args.multiset:
            out["config"]["Z_note"] = "Z is evidence set 0's P(evidence) * Z; parity of evidence runs: tests/test_gpu_configs.py"
        if lattice is not None:
            # size-independent checks in the line: every factor marginal sums to Z; the single-variable marginals that
            # different factors imply agree (calibration across the whole tree)
            tree = lattice["tree"]
            ct = tree.clique_tree
            marg = plan.marginals([(mc, list(fv)) for fv, mc in zip(ct.factor_graph.factors, ct.factor_to_maxclique)])
            sums = np.array([m.sum() for m in marg])
            seen, worst = {}, 0.0
            for f, o in zip(lattice["factors"], marg):
                for ax, v in enumerate(f):
                    m = o.sum(axis=1 - ax)
                    if v in seen:
                        worst = max(worst, float(np.max(np.abs(m - seen[v]) / seen[v])))
                    else:
                        seen[v] = m
            out["parity"] = {"factor_marginals_sum_to_Z_rel_err": float(np.max(np.abs(sums - z)) / z),
                             "calibration_rel_err": worst, "tolerance": 5e-6,
                             "ok": bool(np.max(np.abs(sums - z)) <= 5e-6 * z and worst < 5e-6)}
            out["config"]["junction_tree_build_s"] = lattice["t_build"]
            out["config"]["bytes_definition"] = ("covered-shape potentials + messages, no belief tables (what the reference's evaluate / propagate leave, "
                                                 "junctiontree.py:52-61, 264-274)" if plan.cover is not None else "SURVEY.md 8d: every clique at its full shape")
            out["config"]["full_shape"] = {"algorithmic_bytes_per_step": lattice["alg_full"], "value": lattice["alg_full"] * args.steps / elapsed / 1e9, "unit": "GB/s",
                                           "frac": lattice["alg_full"] * args.steps / elapsed / 1e9 / HBM_PEAK_GBPS,
                                           "definition": "SURVEY.md 8d to the letter: every clique at its full shape, read in both passes, belief written"}
            out["config"]["cliques_without_table"] = lattice["stats"]["n_unit_cliques"]
        run_subs = (default_c4 and world == 1 and args.batch == 1 and not args.no_configs and args.cpu_sample != 0 and not args.level_launches
                    and not args.split_variants and not args.per_launch and args.idle_plans == 0
                    and args.block_log2 == 0 and args.lds_budget == 0 and args.layout_policy == 0)
        if run_subs:
            plan.close()                 # (its 2 GiB of arenas are not needed any more; config 3 wants 18)
            out["configs"] = sub_configs(device, spec, source_id, ms_per_step)
        print(json.dumps(out), flush=True)

    for p in idle:
        p.close()
    plan.close()
    if world > 1:
        barrier()
        lib.jtp_comm_destroy()
    rdzv.close()


if __name__ == "__main__":
    main()
