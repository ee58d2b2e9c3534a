# round-4 GPU batch h: CLI tests after the fork-first change, stream B alone, e2e traces, rehearsals
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_cli.py tests/test_gpu_stage04.py tests/test_gpu_pipeline.py tests/test_gpu_graph_fuzz.py tests/test_gpu_graph_abi.py -x -q --durations=6 > gpurun_out/r04h_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r04h_tests.log
PALACE_BENCH_SKIP_EREF=1 timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04h.err | python tools/bench_brief.py streamB_alone
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04h.err | python tools/bench_brief.py default
bash tools/e2e_trace.sh r04h
