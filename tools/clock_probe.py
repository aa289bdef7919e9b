"""Diagnostic (round 6): is the 3 % run-to-run spread of config 4 the device's clocks?  Times blocks of 100 propagates back to back for a few
seconds while a thread samples what sysfs shows an unprivileged process (hwmon freq*_input, power1_average / power1_input, pp_dpm_sclk / mclk)."""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd"))
from junctiontree_amd import engine, synthetic

def read(path):
    try:
        return open(path).read().strip()
    except OSError as e:
        return "<%s>" % e.strerror

cards = sorted(glob.glob("/sys/class/drm/card*/device"))
files = []
for c in cards:
    for pat in ("hwmon/hwmon*/freq*_input", "hwmon/hwmon*/power1_average", "hwmon/hwmon*/power1_input", "hwmon/hwmon*/temp*_input", "pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent"):
        files += sorted(glob.glob(os.path.join(c, pat)))
print("cards", cards, "files", len(files))
samples, stop = [], False
def sampler():
    while not stop:
        row = {}
        for f in files:
            v = read(f)
            if "\n" in v:
                v = [l for l in v.splitlines() if l.endswith("*")][:1]
            row[f] = v
        samples.append((time.perf_counter(), row))
        time.sleep(0.02)
spec = synthetic.wide_binary_tree(n_cliques=256, width=20, sep=10, card=2, seed=0)
plan = engine.Plan(spec["tree"], spec["node_vars"], spec["sizes"], dtype="f32")
plan.fill_synthetic(1, spec["scales"])
th = threading.Thread(target=sampler); th.start()
blocks = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    t0 = time.perf_counter()
    for _ in range(100): plan.propagate(sync=False)
    plan.sync()
    t1 = time.perf_counter()
    blocks.append((t0, t1, (t1 - t0) / 100 * 1e3))
    if rep % 20 == 19: time.sleep(1.0)       # an idle spell
stop = True; th.join()
for t0, t1, ms in blocks:
    rows = [r for t, r in samples if t0 <= t <= t1]
    brief = {}
    for r in rows[-1:]:
        for f, v in r.items():
            brief[os.path.basename(os.path.dirname(f))[:6] + "/" + os.path.basename(f)] = v
    print("%.4f ms  %s" % (ms, brief))
