# round-4 batch s: stream B two deep (--graph-lag 1) with and without holding the matching rounds back
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0.5 "$@" 2> gpurun_out/r04s.err | python tools/bench_brief.py $tag || { echo "$tag failed"; tail -3 gpurun_out/r04s.err; }; }
for rep in 1 2; do
  run default.$rep
  run lag.$rep --graph-lag 1
  run lag_l2.$rep --graph-lag 1 --stage04-hold l2
  run lag_1.$rep --graph-lag 1 --stage04-hold 1
  run lag_l2_fused.$rep --graph-lag 1 --stage04-hold l2 --fused-probe 1
done
python - <<'PY'
import json,subprocess,sys
a=json.loads(subprocess.run([sys.executable,"bench.py","--steps","3","--warmup","2","--no-cpu-baseline","--no-e2e","--soak-seconds","0.3"],capture_output=True).stdout.decode().strip().splitlines()[-1])
b=json.loads(subprocess.run([sys.executable,"bench.py","--steps","3","--warmup","2","--no-cpu-baseline","--no-e2e","--soak-seconds","0.3","--graph-lag","1","--stage04-hold","l2"],capture_output=True).stdout.decode().strip().splitlines()[-1])
print("digests equal:", a["config"]["result_digest"]["eref_rows"]==b["config"]["result_digest"]["eref_rows"], a["config"]["result_digest"]["graph_and_components"]==b["config"]["result_digest"]["graph_and_components"], b["config"]["result_digest"], b.get("failed_checks"))
PY
