# usage (GPU box): bash tools/prof_median.sh  -> per-kernel median durations of a 5-step bench run
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_med
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_med --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/prof_med.json 2> gpurun_out/prof_med.err || exit 1
s=$(find gpurun_out/prof_med -name '*kernel_stats.csv' | head -1)
t=$(find gpurun_out/prof_med -name '*kernel_trace.csv' | head -1)
python3 tools/rocprof_summary.py gpurun_out/prof_med.md --stats $s --trace $t
sed -n '/per-kernel duration/,$p' gpurun_out/prof_med.md | head -${1:-9}
