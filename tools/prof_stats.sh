
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_cur
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cur --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/prof_cur.json 2> gpurun_out/prof_cur.err || exit 1
f=$(find gpurun_out/prof_cur -name '*kernel_stats.csv' | head -1)
python3 tools/rocprof_summary.py gpurun_out/prof_cur.md --stats $f
head -22 gpurun_out/prof_cur.md
