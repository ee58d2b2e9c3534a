# round-4 GPU batch i: classify with batched loads (tests + kernel time by rocprofv3), full profile of the step
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_graph_abi.py tests/test_gpu_graph_fuzz.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_gpu_cli.py -x -q --durations=5 > gpurun_out/r04i_tests.log 2>&1; echo "tests rc=$?"; tail -10 gpurun_out/r04i_tests.log
cd /tmp && bash "$GRAFT_REPO_ROOT"/tools/prof_full.sh r04i
cd "$GRAFT_REPO_ROOT" && cat gpurun_out/r04i.md | head -80
