"""Copy what tools/collect_profiles.sh brought back (gpurun_out/prof) into profiles/ under this round's names.
    python tools/publish_profiles.py [rNN]"""
import json, os, shutil, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
P, D = os.path.join("gpurun_out", "prof"), "profiles"
for f in ("bench", "bench_driver_form", "bench_batch4", "bench_c2", "bench_c3", "bench_c5_multiset8", "bench_c5_multiset16", "bench_c5_multiset64", "bench_c5_multiset512", "bench_c5_share16"):
    shutil.copy(os.path.join(P, f + ".json"), os.path.join(D, "%s_%s.json" % (rnd, f)))
shutil.copy(os.path.join(P, "kt_bench.json"), os.path.join(D, rnd + "_bench_under_rocprof.json"))
shutil.copy(os.path.join(P, "bench_level.json"), os.path.join(D, rnd + "_bench_level_launches.json"))
for f in ("per_launch", "rank_time_8", "odd_cardinalities", "stage_times", "stage_times_multiset8", "stage_times_rank0_of_8", "c3_api", "c3_api_column_sweep", "counters_c3"):
    shutil.copy(os.path.join(P, f + ".txt"), os.path.join(D, "%s_%s.txt" % (rnd, f)))
for c in ("c2",):          # (config 3 runs the lean unit pass, which carries no time stamps: round 6)          # STAMPS_SUMMARY=1 still prints a line per level: keep every tenth, and the medians
    L = open(os.path.join(P, "stage_times_%s.txt" % c)).read().splitlines()
    lv = [l for l in L[1:] if " level " in l]
    keep = [L[0], "# STAMPS_SUMMARY=1 python tools/stamps.py %s ...: one line per level (every 10th kept here), then the medians over the levels" % c]
    keep += lv[::10] + [l for l in L[1:] if " level " not in l]
    open(os.path.join(D, "%s_stage_times_%s.txt" % (rnd, c)), "w").write("\n".join(keep) + "\n")
for src, dst in (("kernel_stats.csv", "bench_kernel_stats.csv"), ("kernel_trace_summary.json", "kernel_trace_summary.json"), ("hbm_traffic.json", "hbm_traffic.json"),
                 ("hbm_traffic_cases.json", "hbm_traffic_cases.json"), ("counters_multiset64.json", "counters_multiset64.json"), ("c3_api_kernel_stats.csv", "c3_api_kernel_stats.csv")):
    shutil.copy(os.path.join(P, "summary", src), os.path.join(D, "%s_%s" % (rnd, dst)))
rows = [json.loads(l) for l in open(os.path.join(P, "idle_plans_ab.jsonl")) if l.startswith("{")]
with open(os.path.join(D, rnd + "_idle_plans_ab.txt"), "w") as fh:
    fh.write("# library build: " + rows[0]["config"]["library"] + "\n# python3 bench.py --steps 40 --warmup 10 --cpu-sample 0 --idle-plans K, alternating on one box "
             "(round 2: a second, idle plan switched every plan to ticket order and cost 10 percent)\n")
    for r in rows:
        fh.write("idle plans %d: %.4f ms/step  launch mode %s\n" % (r["config"]["idle_plans"], r["ms_per_step"], r["config"]["launch_mode"]))
print(open(os.path.join(D, rnd + "_hbm_traffic.json")).read()[:400])
