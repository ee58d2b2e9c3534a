# round 6: the whole GPU suite on the consolidated tree (eref.hip split, options removed, the 1M-size reference golden), then the default
# bench run (cpu_baseline + e2e) for the record
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r06e_gpu_tests.log 2>&1; rc=$?
tail -22 gpurun_out/r06e_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r06e_bench_line.json 2> gpurun_out/r06e_bench.err || { echo "default run failed ($?)"; tail -5 gpurun_out/r06e_bench.err; }
python tools/bench_brief.py default < gpurun_out/r06e_bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06e_bench_line.json"))
print("failed_checks", d.get("failed_checks")); print("roofline", {k: d["roofline"][k] for k in ("frac", "avg_launch_ms", "frac_phase_a_bytes_only", "traffic")})
print("e2e", d["e2e"].get("seconds"), d["e2e"].get("vs_cpu_baseline")); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("parity"))
print(d["cpu_baseline"].get("reference_eref", {}).get("stdout_vs_eref_cli"))
PY
