: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_stage04.py tests/test_gpu_cli.py -x -q > gpurun_out/r04y_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04y_tests.log
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04y.err | python tools/bench_brief.py default.$rep
done
cd /tmp; bash "$GRAFT_REPO_ROOT"/tools/prof_stats.sh > "$GRAFT_REPO_ROOT"/gpurun_out/r04y_stats.log 2>&1
cd "$GRAFT_REPO_ROOT"; t=$(find gpurun_out/prof_cur -name '*kernel_trace.csv' | head -1); python tools/stage04_timeline.py $t 60
