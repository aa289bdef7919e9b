"""Reduce rocprofv3 output (tools/collect_profiles.sh) to the summaries committed under profiles/.

    python tools/summarize_profiles.py gpurun_out/prof        # on the GPU box: writes <dir>/summary/*
"""
import csv, glob, json, os, sys
from collections import defaultdict

src = sys.argv[1]
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)

# the library build everything here was measured on: bench.py quotes `roofline.traffic` from hbm_traffic.json only when
# this id is the running library's (junction-tree_amd/build.py writes lib/BUILD_ID: "<source id> <git head at build time>")
try:
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "junction-tree_amd", "junctiontree_amd", "lib", "BUILD_ID")) as fh:
        build = fh.read().split()
except OSError:
    build = []
source_id = build[0] if build else "unknown"
git_head = build[-1] if len(build) > 1 else "unknown"


def rows(pattern):
    for path in glob.glob(os.path.join(src, pattern), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


# kernel stats as rocprofv3 wrote them (jt_* kernels only) + our own reduction of the trace
stats = [r for r in rows("kt/**/*kernel_stats.csv") if "jt_" in r.get("Name", "")]
if stats:
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as fh:
        fh.write("# library build: %s git %s (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0)\n" % (source_id, git_head))
        w = csv.DictWriter(fh, fieldnames=list(stats[0].keys()))
        w.writeheader()
        w.writerows(stats)
dur = defaultdict(list)
for r in rows("kt/**/*kernel_trace.csv"):
    name = r.get("Kernel_Name", "")
    if "jt_" in name:
        dur[name.split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
trace = {k: {"launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)} for k, v in dur.items()}

# PMC passes: FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; one pair of passes per case (collect_profiles.sh)
cases = sorted(os.path.basename(p)[len("fetch_"):] for p in glob.glob(os.path.join(src, "fetch_*")) if os.path.isdir(p))
all_traffic = {}
for case in cases:
    pmc = defaultdict(lambda: defaultdict(list))
    for which in ("fetch", "write"):
        for r in rows("%s_%s/**/*counter_collection.csv" % (which, case)):
            name = r.get("Kernel_Name", "")
            if "jt_" in name:
                pmc[name.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    traffic = {}
    for name, c in pmc.items():
        f, wv = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
        if not f or not wv:
            continue
        fetch = sum(f) / len(f) * 1024 * 2          # gfx950: 128-B requests tallied at 64 B (MI355X_MICROARCH.md, HBM)
        write = sum(wv) / len(wv) * 1024
        traffic[name] = {"launches_profiled": len(f), "FETCH_SIZE_KB_per_launch_raw": sum(f) / len(f),
                         "fetch_bytes_per_launch_corrected_x2": fetch, "WRITE_SIZE_KB_per_launch": sum(wv) / len(wv),
                         "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
    # per STEP: every launch of the message-passing kernels in the profiled run (bench.py --steps 5 --warmup 1: six propagates)
    hot = ("jt_propagate_flow", "jt_propagate_flow_marg", "jt_collect_flow", "jt_distribute_flow", "jt_distribute_flow_chain", "jt_multi_flow", "jt_multi_fanout", "jt_collect_level",
           "jt_distribute_level", "jt_reduce_level", "jt_collect_flow_mix", "jt_distribute_flow_mix", "jt_collect_level_mix", "jt_distribute_level_mix")
    per_step = 0.0
    for name, c in pmc.items():
        bare = name.replace("void ", "").split("<")[0]
        if bare in hot and c.get("FETCH_SIZE") and c.get("WRITE_SIZE"):
            per_step += (sum(c["FETCH_SIZE"]) * 1024 * 2 + sum(c["WRITE_SIZE"]) * 1024) / 6.0
    all_traffic[case] = {"kernels": traffic, "hbm_bytes_per_step": per_step,
                         "hbm_bytes_per_step_note": "all launches of the message-passing kernels in the profiled run / its six propagates (5 steps + 1 warm-up)"}
note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 5 --warmup 1 --no-profile "
        "[case arguments]`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); averages "
        "over all launches of the kernel (one launch per phase)")
traffic = all_traffic.get("single", {}).get("kernels", {})
json.dump({"source_id": source_id, "git_head_at_build": git_head, "note": note, "kernels": traffic}, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
json.dump({"source_id": source_id, "git_head_at_build": git_head, "note": note + "; cases: single = the default bench (one evidence set); multiset64 = 64 sets, one pass per group of eight sets; "
           "c2 / c3 = bench.py --config c2 / c3 (c3: potentials at the shape their factors cover, round 5); c3_sweep = c3 on the column-sweep tree of "
           "SURVEY.md 8d; c3_full_tables = c3 with every clique materialised, as in rounds 1-4 (JTP_BENCH_NO_COVER=1)", "cases": all_traffic},
          open(os.path.join(dst, "hbm_traffic_cases.json"), "w"), indent=1)
valu = defaultdict(list)
for r in rows("valu_*/**/*counter_collection.csv"):
    if "jt_" in r.get("Kernel_Name", ""):
        valu[r["Kernel_Name"].split("(")[0] + " " + r["Counter_Name"]].append(float(r["Counter_Value"]))
json.dump(dict({k: sum(v) / len(v) for k, v in valu.items()}, source_id=source_id,
               note="rocprofv3 --pmc passes (a few counters each) over bench.py --steps 3 --warmup 1 --batch 64 --multiset; means over the kernel's launches"),
          open(os.path.join(dst, "counters_multiset64.json"), "w"), indent=1)
# kernel stats of the whole API call on config 3 (tools/run_c3.py under rocprofv3 --kernel-trace --stats)
stats3 = [r for r in rows("kt_c3api/**/*kernel_stats.csv") if "jt_" in r.get("Name", "")]
if stats3:
    with open(os.path.join(dst, "c3_api_kernel_stats.csv"), "w", newline="") as fh:
        fh.write("# library build: %s git %s (rocprofv3 --kernel-trace --stats -- python3 tools/run_c3.py)\n" % (source_id, git_head))
        w = csv.DictWriter(fh, fieldnames=list(stats3[0].keys()))
        w.writeheader()
        w.writerows(stats3)
json.dump(dict(trace, source_id=source_id), open(os.path.join(dst, "kernel_trace_summary.json"), "w"), indent=1)
print(json.dumps(trace, indent=1))
print(json.dumps(all_traffic, indent=1))
