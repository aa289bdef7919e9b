"""The N > 1 path on CPU: two processes (gloo), each holding one part of the tree.

The partition (junctiontree_amd/partition.py), the per-rank plan and its exchange schedule
(send/recv groups between level launches, emitted by the C planner) are the product's host
logic; here each rank executes ITS plan with the CPU emulator of the task tables and moves the
separator messages with torch.distributed (gloo) exactly where the GPU build calls
ncclSend/ncclRecv.  Rank 0 then checks every belief against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, recipe, kwargs, queue):
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(root, "junction-tree_amd"), os.path.join(root, "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from emulator import Emulator
    from junctiontree_amd import engine, partition, synthetic

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kwargs = dict(kwargs)
        perm_seed = kwargs.pop("renumber", None)
        replicate = kwargs.pop("replicate_top", False)
        spec = getattr(synthetic, recipe)(**{k: v for k, v in kwargs.items() if k != "centroid"})
        if perm_seed is not None:       # arbitrary clique numbering, as construct_junction_tree produces
            spec = synthetic.renumber(spec, np.random.default_rng(perm_seed).permutation(spec["n_cliques"]))
        centroid = kwargs.pop("centroid", False)
        n = spec["n_cliques"]
        weights = [float(np.prod([spec["sizes"][v] for v in spec["node_vars"][c]])) for c in range(n)]
        root, parent_of = None, spec["parent"]
        if centroid:        # re-rooted at the weighted centroid: the plan hangs the tree from there
            root, parent_of, owner = partition.partition_tree(spec["parent"], weights, world, replicate_top=replicate)
            assert root != 0 and parent_of[root] == -1
        else:
            owner = partition.subtree_owners(spec["parent"], weights, world, replicate_top=replicate)
        assert len(set(owner) - {world}) == world and (world in owner) == bool(replicate)
        plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f64", plan_only=True,
                           n_ranks=world, rank=rank, owner=owner, block_log2=12, root=root)
        desc = plan.describe()
        emu = Emulator(desc)
        pots = synthetic.potentials_for(spec, seed=9)
        for c in range(n):
            if owner[c] in (rank, world):
                ids = [plan.var_id[v] for v in spec["node_vars"][c]]
                emu.set_potential(c, ids, [spec["sizes"][v] for v in spec["node_vars"][c]], pots[c])

        def comm(ops, msg):
            reqs, recvs = [], []
            for op in ops:
                view = msg[op["off"]:op["off"] + op["count"]]
                if op["send"]:
                    assert not np.any(np.isnan(view)), "sending a message that was never computed"
                    reqs.append(dist.isend(torch.from_numpy(view.copy()), dst=op["peer"]))
                else:
                    buf = torch.empty(op["count"], dtype=torch.float64)
                    reqs.append(dist.irecv(buf, src=op["peer"]))
                    recvs.append((view, buf))
            for r in reqs:
                r.wait()
            for view, buf in recvs:
                view[:] = buf.numpy()

        emu.propagate(comm)
        level_bel = emu.bel.copy()
        emu.propagate_flow(comm)               # dataflow segments between the same exchange groups
        np.testing.assert_array_equal(emu.bel, level_bel)
        mine = {}
        for c in range(n):
            if owner[c] in (rank, world):
                ids = [plan.var_id[v] for v in spec["node_vars"][c]]
                mine[c] = emu.belief(c, ids, [spec["sizes"][v] for v in spec["node_vars"][c]])
        n_comm = len(desc["comm"])
        gathered = [None] * world
        dist.gather_object((mine, n_comm, owner, parent_of), gathered if rank == 0 else None, dst=0)
        if rank == 0:
            import jt_oracle as oracle
            want = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"])
            seen = {}
            for part, _, _, _ in gathered:
                seen.update(part)
            assert sorted(seen) == list(range(n))
            for c in range(n):
                np.testing.assert_allclose(seen[c], want[c], rtol=1e-11, atol=1e-14)
            par = gathered[0][3]
            cut = [c for c in range(n) if par[c] >= 0 and owner[c] != owner[par[c]]]
            cuts = len(cut)
            # every cut edge carries one message up and one down, seen once by each side; below a replicated
            # parent the upward message goes to every other rank and the downward one is formed locally
            assert sum(g[1] for g in gathered) == sum(2 * (world - 1) if owner[par[c]] == world else 4 for c in cut)
            queue.put(("ok", cuts))
    except Exception as exc:                        # noqa: BLE001
        queue.put(("error rank %d" % rank, repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("recipe,kwargs", [
    ("wide_binary_tree", {"n_cliques": 15, "width": 12, "sep": 6, "card": 2, "seed": 1}),
    ("random_tree", {"n_cliques": 14, "width": 11, "sep": 5, "card": 2, "seed": 3}),
    ("wide_binary_tree", {"n_cliques": 15, "width": 13, "sep": 6, "card": 2, "seed": 4, "reduce_min": "2"}),
    ("random_tree", {"n_cliques": 24, "width": 10, "sep": 4, "card": 2, "seed": 7, "renumber": 5}),
    ("wide_binary_tree", {"n_cliques": 31, "width": 11, "sep": 5, "card": 2, "seed": 2, "replicate_top": True}),
    ("random_tree", {"n_cliques": 30, "width": 10, "sep": 4, "card": 2, "seed": 9, "renumber": 3, "replicate_top": True}),
    # (round 3) cardinalities that are not powers of two: mixed-radix rows on both ranks, padded bit-field messages across the cut
    ("wide_binary_tree", {"n_cliques": 15, "width": 7, "sep": 3, "card": 3, "seed": 5}),
    ("random_tree", {"n_cliques": 18, "width": 5, "sep": 2, "card": 5, "seed": 6, "renumber": 2, "replicate_top": True}),
    # (round 4) a chain given with its END as the root: partitioned after re-rooting at the weighted centroid, the plan hung from there
    ("chain_tree", {"n_cliques": 21, "card": 6, "width": 3, "centroid": True}),
    ("chain_tree", {"n_cliques": 24, "card": 5, "width": 3, "centroid": True, "replicate_top": True}),
    ("random_tree", {"n_cliques": 26, "width": 9, "sep": 4, "card": 2, "seed": 11, "renumber": 4, "centroid": True, "replicate_top": True}),
])
def test_two_rank_exchange_schedule(recipe, kwargs, monkeypatch):
    import multiprocessing as mp
    kwargs = dict(kwargs)
    if "reduce_min" in kwargs:          # reduce tasks at the cuts: the summed message is what travels
        monkeypatch.setenv("JTP_REDUCE_MIN", kwargs.pop("reduce_min"))
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, recipe, kwargs, queue)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("multi-rank worker hung")
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    status, cuts = queue.get(timeout=5)
    assert status == "ok" and cuts >= 1


def test_partition_is_balanced_for_the_benchmark_tree():
    from junctiontree_amd import partition, synthetic
    spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
    for world in (2, 4, 8):
        owner = partition.subtree_owners(spec["parent"], [1.0] * 256, world)
        loads = partition.part_weights(owner, [1.0] * 256, world)
        assert max(loads) <= 256 / world * 1.25
        cuts = sum(1 for c in range(1, 256) if owner[c] != owner[spec["parent"][c]])
        assert cuts <= world          # shallow quotient tree: at most one cut per part


def test_partition_after_rerooting_at_the_weighted_centroid():
    """"general trees: re-root at the weighted centroid first".  A chain handed over with its end as the root:
    from there the first cut leaves the top part holding half the chain; hung from its middle, two ranks get a half each."""
    from junctiontree_amd import partition, synthetic
    spec = synthetic.chain_tree(n_cliques=41, card=4, width=3)
    w = [1.0] * 41
    assert partition.weighted_centroid(spec["parent"], w) == 20
    root, par, owner = partition.partition_tree(spec["parent"], w, 2, replicate_top=True)
    assert root == 20 and par[20] == -1 and par[19] == 20 and par[21] == 20 and par[0] == 1
    loads = partition.part_weights(owner, w, 2)
    assert max(loads) <= 22 and owner[20] == 2 and owner[0] != owner[40]
    # weights count: a heavy end pulls the centroid towards it
    heavy = [1.0] * 40 + [100.0]
    assert partition.weighted_centroid(spec["parent"], heavy) == 40
    # the balanced binary tree of the benchmark is rooted at its centroid already
    c4 = synthetic.wide_binary_tree(n_cliques=255, width=12, sep=6, card=2, seed=0)
    assert partition.weighted_centroid(c4["parent"], [1.0] * 255) == 0


def _rdzv_worker(rank, world, port, queue):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "junction-tree_amd"))
    from junctiontree_amd.rendezvous import Rendezvous
    z = Rendezvous(rank, world, "127.0.0.1", port, port_file="/tmp/jtp_rdzv_test_%d" % port)
    payload = z.broadcast(b"x" * 128 if rank == 0 else None)
    z.barrier()
    m = z.allreduce_max(1.0 + rank)
    z.close()
    queue.put((rank, payload, m))


def test_socket_rendezvous_used_by_bench():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, 3, port, queue)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=60)
    got = sorted(queue.get(timeout=5) for _ in range(3))
    assert [g[0] for g in got] == [0, 1, 2]
    assert all(g[1] == b"x" * 128 and g[2] == 3.0 for g in got)
