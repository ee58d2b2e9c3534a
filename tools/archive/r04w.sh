# round-4 batch w: closed arcs flagged in the decomposition (tests of everything that decomposes, then the step)
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_stage04.py tests/test_gpu_graph_abi.py tests/test_gpu_cli.py tests/test_gpu_pipeline.py -x -q --durations=5 > gpurun_out/r04w_tests.log 2>&1; echo "tests rc=$?"; tail -9 gpurun_out/r04w_tests.log
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04w.err | python tools/bench_brief.py default.$rep
done
PALACE_BENCH_SKIP_EREF=1 timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04w.err | python tools/bench_brief.py streamB_alone
timeout -k 10 300 python bench.py --contigs 5000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04w.err | python tools/bench_brief.py 5m
