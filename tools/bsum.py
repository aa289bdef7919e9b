"""Print the essentials of bench JSON files: python tools/bsum.py file [file ...]"""
import json, sys
for path in sys.argv[1:]:
    try:
        lines = [l for l in open(path) if l.startswith("{")]
        d = json.loads(lines[-1])
    except Exception as exc:
        print("%-44s unreadable (%r)" % (path, exc))
        continue
    c = d["config"]
    print("%-44s %8.4f ms/step %8.0f GB/s  %9.0f msg/s  sets %d" % (path.split("/")[-1], d["ms_per_step"], d["value"], d["messages_per_sec"], c.get("evidence_sets_per_step", 1)))
    for k, v in (d.get("roofline") or {}).get("rank0_kernels", {}).items():
        print("      %-30s %2d launches %8.4f ms %7.0f GB/s" % (k, v["launches"], v["ms_per_step"], v["GBps"]))
