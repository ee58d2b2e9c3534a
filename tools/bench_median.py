import json, sys, statistics
d = json.load(sys.stdin)
c = d["stage_ms"]["eref_count_each_step"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "step", round(d["ms_per_step"], 2), "count median", round(statistics.median(c), 2),
      "min", min(c), "max", max(c), "scan", round(d["stage_ms"]["eref_scan_refs"], 2))
