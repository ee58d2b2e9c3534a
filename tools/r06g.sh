# round 6: the gather kernel of the indexed scan takes its tiles from a list of the active refs' tiles (eref_need_kernel) instead of a binary
# search per workgroup over every tile of the DB: the suites that pin the scan, then the step (scan time: stage_ms.eref_scan_refs) and rocprof
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests/test_gpu_eref.py tests/test_gpu_configs.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r06g_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r06g_tests.log; [ $rc -eq 0 ] || { tail -30 gpurun_out/r06g_tests.log; exit $rc; }
AB_STEPS=30 bash tools/ab.sh r06g 3 default | tee gpurun_out/r06g_variants.log
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/r06g_st --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > /dev/null 2> gpurun_out/r06g_st.err
grep -h "eref_gather\|eref_need\|eref_window\|eref_sentinel" $(find gpurun_out/r06g_st -name '*kernel_stats.csv' | head -1) | cut -c1-120
timeout -k 10 300 python bench.py --contigs 5000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r06g_5m.err | python tools/bench_brief.py 5m
