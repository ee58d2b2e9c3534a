# round 6, second GPU session: the whole GPU suite on the tree (new: the timed path's rows against the oracle at 500k / 1M contigs, the
# all-schemes mode of bench.py --gpus N rehearsed on one GPU, the scan's ref-range filter), then one rank's stream A under shard_counts
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=25 --deselect tests/test_gpu_configs.py::test_headline_size_eref_cli_and_timed_path_equal_reference > gpurun_out/r06b_gpu_tests.log 2>&1; rc=$?
tail -40 gpurun_out/r06b_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/counts_share_diag.sh 1000000 | tee gpurun_out/r06b_counts_share.log
