import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["ms_per_step"], 2), round(d["value"] / 1e6, 2), {k: round(v, 2) for k, v in d["stage_ms"].items() if isinstance(v, float)})
