import sys,json
tag=sys.argv[1]
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): 
        if line.startswith("#"): print(line)
        continue
    d=json.loads(line)
    print("%-40s ms/step %.4f  GB/s %.0f" % (tag, d["ms_per_step"], d["value"]))
    r=d.get("roofline")
    if r:
        for k,v in r["rank0_kernels"].items(): print("      %-28s %2d launches %.4f ms %.0f GB/s" % (k, v["launches"], v["ms_per_step"], v["GBps"]))
