# round 6, last session: smoke() and the default bench run of the final tree (cpu_baseline + e2e; roofline.traffic from the committed profile of this build)
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/r06y_bench_line.json 2> gpurun_out/r06y_bench.err || { echo "default run exit status $?"; tail -5 gpurun_out/r06y_bench.err; }
python tools/bench_brief.py default < gpurun_out/r06y_bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06y_bench_line.json"))
print("failed_checks", d.get("failed_checks")); r = d["roofline"]
print("roofline", {k: r[k] for k in ("frac", "avg_launch_ms", "frac_phase_a_bytes_only", "traffic", "traffic_source")})
print("e2e", round(d["e2e"]["seconds"], 3), d["e2e"]["one_process_stage04"].get("seconds"), d["e2e"].get("vs_cpu_baseline"))
print("cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"].get("parity"))
PY
