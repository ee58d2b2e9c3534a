# round 6, first GPU session: the suites that pin what changed (eref: partial counts are one fused count per reset or a failure;
# decomposition: the slot word is the slot's state), the exchange selftest, and an A/B on one box: HEAD~'s library (tools/ab/lib_base.so)
# against the tree's, with and without the matching rounds beside the count launch
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_eref.py tests/test_gpu_stage04.py tests/test_gpu_graph_abi.py tests/test_match_second_opinion.py -x -q -m gpu --durations=8 > gpurun_out/r06a_tests.log 2>&1; rc=$?
tail -14 gpurun_out/r06a_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 palace_amd/bin/exchange_selftest > gpurun_out/r06a_selftest.log 2>&1 || { echo "selftest failed"; tail -5 gpurun_out/r06a_selftest.log; exit 1; }
tail -1 gpurun_out/r06a_selftest.log
AB_STEPS=30 bash tools/ab.sh r06a 3 new "base,PALACE_HIP_SO=$PWD/tools/ab/lib_base.so" "norounds,PALACE_BENCH_DIAG_SKIP=match" | tee gpurun_out/r06a_variants.log
