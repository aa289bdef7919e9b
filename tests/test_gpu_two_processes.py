"""Two PROCESSES on one GPU, both with dataflow propagates in flight, no JTP_FLOW_TICKETS in the environment: each finds the
other through the shared-memory board (/dev/shm/jtprop_flight_<PCI bus id>, jtp_engine.hip `board`) and launches in ticket
order - no dataflow wait times out, no fall-back to level launches (round 3: an environment variable, else a 2 s stall),
and the results are the oracle's."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.path.join(%(root)r, "junction-tree_amd")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import numpy as np
import jt_oracle as oracle
from junctiontree_amd import engine, synthetic
seed, start = int(sys.argv[1]), float(sys.argv[2])
spec = synthetic.wide_binary_tree(n_cliques=63, width=16, sep=8, card=2, seed=seed)
plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
plan.fill_synthetic(3, spec["scales"])
plan.propagate()
while time.time() < start:
    time.sleep(0.001)
t_end = time.time() + 2.5
n = 0
while time.time() < t_end:
    for _ in range(20):
        plan.propagate(sync=False)
    plan.sync()
    n += 20
st = plan.stats()
pots = synthetic.potentials_for(spec, seed=3, dtype=np.float32)
want, z = oracle.beliefs_exact(spec["tree"], pots, spec["node_vars"], return_z=True)
worst = abs(plan.z() - z) / z
for c in (0, 31, 62):
    worst = max(worst, float(np.max(np.abs(plan.belief(c) - want[c]) / want[c])))
print(json.dumps({"propagates": n, "fallbacks": st["flow_fallbacks"], "foreign_seen": st["foreign_seen"],
                  "tickets_used": st["tickets_used"], "launch_mode": st["launch_mode"], "worst": worst}))
'''


def test_two_processes_find_each_other_and_use_ticket_order():
    env = {k: v for k, v in os.environ.items() if k != "JTP_FLOW_TICKETS"}
    start = time.time() + 8.0                       # (library load + plan creation of both)
    procs = [subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}, str(11 + i), repr(start)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(2)]
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            pytest.fail("worker hung")
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    for o in outs:
        assert o["fallbacks"] == 0 and o["worst"] < 1e-6 and o["propagates"] >= 100, outs
    # they overlapped for 2.5 s: at least one of them met the other in flight, and ran those propagates in ticket order
    assert sum(o["foreign_seen"] for o in outs) > 0, outs
    assert all(o["tickets_used"] >= o["foreign_seen"] for o in outs), outs
