# round-4 GPU batch g: classify with FASTG offsets, e2e traces after the host changes, full GPU test suite
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r04g_tests.log 2>&1; echo "gpu tests rc=$?"; tail -18 gpurun_out/r04g_tests.log
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04g.err | python tools/bench_brief.py default.$rep
done
bash tools/e2e_trace.sh r04g
