# matching by vertex-side search (no atomics in an iteration): the suites that pin the decomposition, then the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_stage04.py tests/test_gpu_cli.py tests/test_gpu_graph_abi.py tests/test_gpu_pipeline.py -x -q > gpurun_out/r04za_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r04za_tests.log
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04za.err | python tools/bench_brief.py default.$rep
done
timeout -k 10 400 python bench.py --contigs 5000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04za.err | python tools/bench_brief.py 5m
(cd /tmp && timeout -k 10 300 bash "$GRAFT_REPO_ROOT"/tools/prof_stats.sh > "$GRAFT_REPO_ROOT"/gpurun_out/r04za_stats.log 2>&1); t=$(find gpurun_out/prof_cur -name "*kernel_trace.csv" | head -1); python tools/stage04_timeline.py $t 60 2>&1 | tail -32
