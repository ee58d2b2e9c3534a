# round-4 GPU batch k: two-pass classify (tests + kernel times)
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_graph_abi.py tests/test_gpu_graph_fuzz.py tests/test_gpu_configs.py tests/test_gpu_cli.py tests/test_gpu_stage04.py -x -q --durations=5 > gpurun_out/r04k_tests.log 2>&1; echo "tests rc=$?"; tail -10 gpurun_out/r04k_tests.log
cd /tmp && bash "$GRAFT_REPO_ROOT"/tools/prof_stats.sh > "$GRAFT_REPO_ROOT"/gpurun_out/r04k_stats.log 2>&1; cd "$GRAFT_REPO_ROOT"; grep -E "classify|depth_select|resolve|compact|bin1|bin2|lds_count" gpurun_out/prof_cur.md | head -12
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04k.err | python tools/bench_brief.py default
