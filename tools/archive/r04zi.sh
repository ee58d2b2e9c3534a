# column decode pipelined behind the record walk: the suites that load BAMs, then the chain's traces
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_cli.py tests/test_gpu_stage04.py tests/test_gpu_configs.py tests/test_gpu_graph_fuzz.py tests/test_gpu_bench_workloads.py -x -q > gpurun_out/r04zi_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04zi_tests.log
timeout -k 10 500 bash tools/e2e_trace.sh r04zi
