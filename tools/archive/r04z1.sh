# matching's threaded readers: CLI / stage-04 tests (with the parsers forced into parts as well), then the chain's traces
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_stage04.py -x -q > gpurun_out/r04z1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04z1_tests.log
PALACE_HOST_THREADS=5 timeout -k 10 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_stage04.py -x -q > gpurun_out/r04z1_tests5.log 2>&1; echo "tests(5 parts) rc=$?"; tail -3 gpurun_out/r04z1_tests5.log
timeout -k 10 500 bash tools/e2e_trace.sh r04z1
