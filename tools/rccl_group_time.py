"""Diagnostic: what the exchange step of a sharded propagate - ONE RCCL group of 7 ncclSend + 7 ncclRecv of 8 KiB -
costs on this GPU in loop-back, and what of it is the RCCL kernel itself (run under `rocprofv3 --kernel-trace --stats`: the kernel's
average duration against the step's share of the propagate).   python tools/rccl_group_time.py [rank] [steps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import _capi, engine, partition, synthetic
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
root, _, owner = partition.partition_tree(spec["parent"], [1.0] * spec["n_cliques"], 8, replicate_top=True)
lib = _capi.lib()
buf = C.create_string_buffer(128)
_capi.check(lib.jtp_comm_unique_id(buf))
_capi.check(lib.jtp_comm_init(0, 1, C.c_char_p(buf.raw), 0))
out = {}
for mode in ("1", "2", "1", "2"):
    os.environ["JTP_FAKE_COMM"] = mode
    plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32", n_ranks=8, rank=rank, owner=owner, root=root)
    plan.fill_synthetic(1, spec["scales"])
    for _ in range(5):
        plan.propagate(sync=False)
    plan.sync()
    plan.region_begin()
    for _ in range(steps):
        plan.propagate(sync=False)
    ms = plan.region_end() / steps
    out.setdefault(mode, []).append(ms * 1e3)
    plan.close()
lib.jtp_comm_destroy()
print("rank %d share: exchange as fills %s us, as the RCCL group in loop-back %s us -> the group costs %.1f us of stream time"
      % (rank, ["%.1f" % v for v in out["1"]], ["%.1f" % v for v in out["2"]], min(out["2"]) - min(out["1"])))
