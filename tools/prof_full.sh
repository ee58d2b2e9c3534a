# usage (on the GPU box): [BENCH_EXTRA='--contigs 500000'] bash tools/prof_full.sh <tag>   -> gpurun_out/<tag>.md (+ bench line gpurun_out/<tag>_bench.json)
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-prof}
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 $BENCH_EXTRA > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_stats.err || exit 1
echo "stats pass done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --soak-seconds 0 $BENCH_EXTRA > /dev/null 2> gpurun_out/${tag}_fetch.err || exit 1
echo "fetch pass done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_write --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --soak-seconds 0 $BENCH_EXTRA > /dev/null 2> gpurun_out/${tag}_write.err || exit 1
echo "write pass done"
s=$(find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1)
t=$(find gpurun_out/${tag}_stats -name '*kernel_trace.csv' | head -1)
f=$(find gpurun_out/${tag}_fetch -name '*counter_collection.csv' | head -1)
w=$(find gpurun_out/${tag}_write -name '*counter_collection.csv' | head -1)
python3 tools/rocprof_summary.py gpurun_out/${tag}.md --stats $s --trace $t --pmc FETCH_SIZE=$f --pmc WRITE_SIZE=$w
python3 tools/traffic_json.py $f $w gpurun_out/${tag}_traffic.json "profiles/${tag}.md (tools/prof_full.sh)" ${TRAFFIC_CONTIGS:-1000000}
