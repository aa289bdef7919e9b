"""The multi-rank path end to end on ONE GPU: W processes, each a rank with its share of the tree, real
kernels and the real exchange schedule; only the transport is replaced - tests/mock_rccl moves the
ncclSend/ncclRecv payloads through mailboxes in /dev/shm (JTP_RCCL_LIB), stream ordered and asynchronous like
the real transport (wait kernel -> copy -> signal kernel on the caller's stream; nothing synchronises a stream),
because RCCL refuses several ranks on one device and the test box has a single GPU.  Checks, over three consecutive propagates (the message arena
alternates between its halves), every belief of every rank against the oracle, with the default
dataflow launches and with one launch per level, with and without reduce tasks at the cuts."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MOCK_SRC = os.path.join(HERE, "mock_rccl", "mock_rccl.cpp")
MOCK_LIB = os.path.join(HERE, "mock_rccl", "libmockrccl.so")


def _build_mock():
    if not os.path.exists(MOCK_LIB) or os.path.getmtime(MOCK_LIB) < os.path.getmtime(MOCK_SRC):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "--offload-arch=gfx950", "-fPIC", "-shared",
                               MOCK_SRC, "-o", MOCK_LIB])
    return MOCK_LIB


def _spec(synthetic, recipe, kwargs):
    kwargs = dict(kwargs)
    perm_seed = kwargs.pop("renumber", None)
    kwargs.pop("replicate_top", None)
    spec = getattr(synthetic, recipe)(**kwargs)
    if perm_seed is not None:           # arbitrary clique numbering (the recipes number breadth first)
        spec = synthetic.renumber(spec, np.random.default_rng(perm_seed).permutation(spec["n_cliques"]))
    return spec


def _worker(rank, world, port_file, recipe, kwargs, opts, env, queue):
    for p in (os.path.join(ROOT, "junction-tree_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(env)
    import ctypes as C
    from junctiontree_amd import _capi, engine, partition, synthetic
    from junctiontree_amd.rendezvous import Rendezvous
    try:
        lib = _capi.lib()
        rdzv = Rendezvous(rank, world, "127.0.0.1", 0, timeout=120.0, port_file=port_file)
        uid = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            _capi.check(lib.jtp_comm_unique_id(buf))
            uid = buf.raw
        uid = rdzv.broadcast(uid)
        _capi.check(lib.jtp_comm_init(rank, world, C.c_char_p(uid), 0))
        spec = _spec(synthetic, recipe, kwargs)
        n = spec["n_cliques"]
        weights = [float(np.prod([spec["sizes"][v] for v in spec["node_vars"][c]])) for c in range(n)]
        owner = partition.subtree_owners(spec["parent"], weights, world, replicate_top=kwargs.get("replicate_top", False))
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", n_ranks=world, rank=rank,
                           owner=owner, **opts)
        results = []
        for rep in range(3):
            pots = synthetic.potentials_for(spec, seed=40 + rep)
            for c in range(n):
                if owner[c] in (rank, world):
                    plan.set_potential(c, pots[c])
            rdzv.barrier()
            plan.propagate()
            mine = {c: plan.belief(c) for c in range(n) if owner[c] in (rank, world)}
            z = plan.z() if plan.owns(plan.root) else None
            results.append((mine, z))
        n_comm = len(plan.describe()["comm"])
        fallbacks = plan.stats()["flow_fallbacks"]
        plan.close()
        rdzv.barrier()
        lib.jtp_comm_destroy()
        rdzv.close()
        queue.put((rank, "ok", results, owner, n_comm, fallbacks))
    except Exception as exc:                        # noqa: BLE001
        queue.put((rank, "error", repr(exc), None, 0, 0))
        raise


@pytest.mark.gpu
@pytest.mark.parametrize("world,recipe,kwargs,opts,env", [
    (2, "wide_binary_tree", {"n_cliques": 15, "width": 13, "sep": 6, "card": 2, "seed": 1}, {}, {}),
    (4, "wide_binary_tree", {"n_cliques": 31, "width": 14, "sep": 7, "card": 2, "seed": 2}, {"block_log2": 11}, {"JTP_REDUCE_MIN": "2"}),
    (3, "random_tree", {"n_cliques": 14, "width": 11, "sep": 5, "card": 2, "seed": 3}, {"level_launches": True}, {}),
    (2, "chain_tree", {"n_cliques": 9, "card": 8, "width": 3}, {}, {}),
    (3, "random_tree", {"n_cliques": 26, "width": 11, "sep": 5, "card": 2, "seed": 8, "renumber": 2}, {}, {}),
    (4, "wide_binary_tree", {"n_cliques": 31, "width": 13, "sep": 6, "card": 2, "seed": 5, "replicate_top": True}, {}, {}),
    (3, "random_tree", {"n_cliques": 28, "width": 11, "sep": 5, "card": 2, "seed": 6, "renumber": 4, "replicate_top": True}, {"level_launches": True}, {}),
    # (round 3) cardinalities that are not powers of two: tables with mixed-radix rows (kernels *_mix) and rows stored at true
    # cardinalities on every rank, messages across the cuts as padded bit fields
    (2, "wide_binary_tree", {"n_cliques": 15, "width": 8, "sep": 4, "card": 3, "seed": 7}, {}, {}),
    (3, "random_tree", {"n_cliques": 20, "width": 6, "sep": 3, "card": 5, "seed": 9, "renumber": 3, "replicate_top": True}, {}, {}),
    (4, "random_tree", {"n_cliques": 24, "width": 7, "sep": 3, "card": 3, "seed": 11}, {"level_launches": True}, {}),
])
def test_ranks_sharing_one_gpu_through_mock_transport(world, recipe, kwargs, opts, env, tmp_path):
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import jt_oracle as oracle
    from junctiontree_amd import synthetic

    # (ranks share the GPU here: ticket order, see DESIGN.md 5 - on a GPU of its own a rank runs blockIdx order)
    env = dict(env, JTP_RCCL_LIB=_build_mock(), JTP_FLOW_TICKETS="1")
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port_file = str(tmp_path / "rdzv_port")
    procs = [ctx.Process(target=_worker, args=(r, world, port_file, recipe, kwargs, opts, env, queue)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            rank, status, results, owner, n_comm, fallbacks = queue.get(timeout=300)
            assert status == "ok", results
            got[rank] = (results, owner, n_comm, fallbacks)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    spec = _spec(synthetic, recipe, kwargs)
    n = spec["n_cliques"]
    owner = got[0][1]
    assert len(set(owner) - {world}) == world
    for rep in range(3):
        pots = synthetic.potentials_for(spec, seed=40 + rep)
        want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
        seen = {}
        for rank in range(world):
            mine, zr = got[rank][0][rep]
            seen.update(mine)
            if zr is not None:
                assert abs(zr - z) <= 1e-11 * abs(z)
        assert sorted(seen) == list(range(n))
        for c in range(n):
            np.testing.assert_allclose(seen[c], want[c], rtol=1e-11, atol=1e-13 * np.max(np.abs(want[c])))
    cut = [c for c in range(n) if spec["parent"][c] >= 0 and owner[c] != owner[spec["parent"][c]]]
    assert len(cut) >= world - 1
    assert sum(got[r][2] for r in range(world)) == sum(2 * (world - 1) if owner[spec["parent"][c]] == world else 4 for c in cut)
    assert all(got[r][3] == 0 for r in range(world))



@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 4])
def test_bench_starts_its_own_ranks(gpus):
    """`python bench.py --gpus N` with no launcher and WORLD_SIZE unset (how the driver may call it): the parent starts
    the N rank processes itself before it touches a GPU, relays rank 0's JSON line and exits 0.  Here the N ranks share
    the box's one GPU over the mock transport, on the FULL config-4 tree (256 x 2^20 float32, top part replicated):
    Z of the sharded run must agree with the one-GPU / numpy-oracle value."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["JTP_RCCL_LIB"] = _build_mock()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "5", "--warmup", "2",
                          "--cpu-sample", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout.decode()
    out = json.loads(lines[0])
    assert out["n_gpus"] == gpus and out["steps"] == 5 and out["scaling"] == "strong"
    assert out["config"]["Z_rel_err"] <= 1e-6, out["config"]
    # (the distribute side of the sharded run: three clique beliefs per rank sum to Z)
    assert out["config"]["sharded_belief_sums_rel_err"] <= 1e-6, out["config"]
    assert out["config"]["launch_mode"] == "flow_tickets"          # ranks share the GPU here
    assert out["value"] > 0 and out["ms_per_step"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("flag", ["--multiset", "--share"])
def test_bench_replicas_of_evidence_sets_over_two_ranks(flag):
    """BASELINE configs[4] at N > 1 ("512 observation sets, 8 MI355X") is replicas only: `bench.py --gpus 2 --batch 8 --multiset`
    gives every rank its own eight evidence sets over its own copy of the tables, no exchange; rank 0's line sums the ranks
    (scaling "weak") and says what the communicator reports.  Here the two ranks share the box's one GPU (mock transport)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["JTP_RCCL_LIB"] = _build_mock()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-sample", "0",
                          "--batch", "8", flag, "--cliques", "63", "--width", "16", "--sep", "8"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout.decode()
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and cfg["evidence_sets_per_step"] == 16 and cfg["evidence_sets_per_rank"] == 8
    assert cfg["rccl"]["ncclCommCount"] == [2] and sorted(r["ncclCommUserRank"] for r in cfg["rccl"]["ranks"]) == [0, 1]
    assert len(cfg["ms_per_step_per_rank"]) == 2 and cfg["replica_marginal_sums_rel_err"] <= 1e-6
    assert "replicas" in cfg["parallelism"] and out["value"] > 0


def test_bench_rank_failure_is_an_error():
    """Ranks that fail (here: a configuration the rank processes refuse) must fail the whole run with a non-zero
    exit code and no result line, not hang the parent.  Needs no GPU: the ranks exit before they load the library."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c3", "--steps", "2",
                          "--spawn-timeout", "60"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert res.returncode != 0
    assert not any(ln.startswith("{") for ln in res.stdout.decode().splitlines())
    assert b"exited with code" in res.stderr


@pytest.mark.gpu
def test_bench_with_more_ranks_than_devices_says_so():
    """`python bench.py --gpus 2` on a box with ONE device and real RCCL (no stand-in transport): one clear line - how many
    devices the run needs and how many were found - and a non-zero exit, before RCCL is asked to put two ranks on one GPU
    (round 3: ncclCommInitRank's "invalid usage")."""
    from junctiontree_amd import _capi
    if _capi.device_count() != 1:
        pytest.skip("needs a box with exactly one visible device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "JTP_RCCL_LIB")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
                          "--spawn-timeout", "120"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = res.stderr.decode()
    assert res.returncode != 0, err[-1000:]
    assert "2 ranks need 2 devices, found 1" in err, err[-1000:]
    assert not any(ln.startswith("{") for ln in res.stdout.decode().splitlines())
