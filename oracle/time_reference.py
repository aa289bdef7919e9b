"""Build-container script (imports the unmodified reference from /root/reference; never runs on the GPU box):
wall time of the reference's `compute_beliefs` against the CPU restatement `jt_oracle.beliefs_refshaped` on the FULL
BASELINE configs[3] input (256 cliques x 2^20 float32, 10 shared variables per edge), one core each, and the agreement
of their outputs.  Writes tests/golden/ref_over_port.json, which bench.py quotes in `cpu_baseline` so that the port's
rate can be turned into a reference-equivalent one (the port is the faster of the two: it FLATTERS the CPU).

    PYTHONHASHSEED=0 python oracle/time_reference.py [n_cliques]

Test infrastructure only: nothing in the product path imports this file or anything under oracle/.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "junction-tree_amd"))
sys.path.insert(0, HERE)
sys.setrecursionlimit(20000)

import gen_golden                                     # noqa: E402  (puts /root/reference on the path, colours labels)
import jt_oracle as oracle                            # noqa: E402
from junctiontree_amd import synthetic                # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    spec = synthetic.wide_binary_tree(n_cliques=n, width=20, sep=10, card=2, seed=0)
    pots = synthetic.potentials_for(spec, seed=1, dtype=np.float32)
    t0 = time.perf_counter()
    c0 = time.process_time()
    ref = gen_golden.run_reference_safe(spec, pots)
    t_ref, c_ref = time.perf_counter() - t0, time.process_time() - c0
    t0 = time.perf_counter()
    c0 = time.process_time()
    port = oracle.beliefs_refshaped(spec["tree"], pots, spec["node_vars"])
    t_port, c_port = time.perf_counter() - t0, time.process_time() - c0
    err = max(float(np.max(np.abs(r - p)) / np.max(np.abs(p))) for r, p in zip(ref, port))
    z = float(np.sum(port[0], dtype=np.float64))
    cpu = "unknown"
    with open("/proc/cpuinfo") as fh:
        for line in fh:
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    out = {
        "workload": "BASELINE configs[3]: %d cliques, width 20, cardinality 2, 10 shared variables per edge, float32 "
                    "(synthetic.wide_binary_tree seed 0, potentials seed 1)" % n,
        "reference_s": t_ref, "port_s": t_port, "reference_over_port_time": t_ref / t_port,
        "reference_cpu_over_wall": c_ref / t_ref, "port_cpu_over_wall": c_port / t_port,
        "max_rel_difference_of_outputs": err, "Z": z,
        "host": cpu, "python": sys.version.split()[0], "numpy": np.__version__,
        "script": "oracle/time_reference.py (build container; reference = /root/reference, unmodified, labels coloured "
                  "per SURVEY.md Appendix C)",
    }
    path = os.path.join(ROOT, "tests", "golden", "ref_over_port.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
