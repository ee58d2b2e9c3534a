# round-4 GPU batch m: classify with block-level appends (tests + kernel times), the device inflate against zlib
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_inflate.py -x -q > gpurun_out/r04m_inflate.log 2>&1; echo "inflate tests rc=$?"; tail -12 gpurun_out/r04m_inflate.log
python -m pytest tests/test_gpu_graph_abi.py tests/test_gpu_graph_fuzz.py tests/test_gpu_configs.py::test_config4_long_contigs_full_size_and_oracle_sample tests/test_gpu_cli.py -x -q > gpurun_out/r04m_tests.log 2>&1; echo "graph tests rc=$?"; tail -4 gpurun_out/r04m_tests.log
cd /tmp && bash "$GRAFT_REPO_ROOT"/tools/prof_stats.sh > "$GRAFT_REPO_ROOT"/gpurun_out/r04m_stats.log 2>&1; cd "$GRAFT_REPO_ROOT"; grep -E "classify|depth_select|resolve|compact|bin1|bin2|lds_count" gpurun_out/prof_cur.md | head -12
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04m.err | python tools/bench_brief.py default
