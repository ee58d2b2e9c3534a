# round 6, first GPU session: the eref suite on the ADVICE fixes (partial counts: one fused count per reset or a failure), the exchange
# selftest, and the baseline of the day for the A/B runs that follow (default step; stage 04's share of the count launch)
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_eref.py -x -q -m gpu > gpurun_out/r06a_eref_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r06a_eref_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 palace_amd/bin/exchange_selftest > gpurun_out/r06a_selftest.log 2>&1 || { echo "selftest failed"; tail -5 gpurun_out/r06a_selftest.log; exit 1; }
tail -1 gpurun_out/r06a_selftest.log
AB_STEPS=30 bash tools/ab.sh r06a 2 default "norounds,PALACE_BENCH_DIAG_SKIP=match" "nostage04,PALACE_BENCH_DIAG_SKIP=stage04" | tee gpurun_out/r06a_variants.log
