: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
bash "$GRAFT_REPO_ROOT"/tools/prof_stats.sh > "$GRAFT_REPO_ROOT"/gpurun_out/r04x_stats.log 2>&1
cd "$GRAFT_REPO_ROOT"; t=$(find gpurun_out/prof_cur -name '*kernel_trace.csv' | head -1); python tools/stage04_timeline.py $t 40
