# end state once more (the library's build id changed with the mark_before_level2 option): profiles of the 1M and the 500k workload,
# the full line
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp && bash "$R"/tools/prof_full.sh r04z > "$R"/gpurun_out/r04z_prof.log 2>&1; tail -2 "$R"/gpurun_out/r04z_prof.log
cd /tmp && BENCH_EXTRA='--contigs 500000' TRAFFIC_CONTIGS=500000 bash "$R"/tools/prof_full.sh r04z_500k > "$R"/gpurun_out/r04z_500k_prof.log 2>&1; tail -1 "$R"/gpurun_out/r04z_500k_prof.log
cd "$R"
cp gpurun_out/r04z_traffic.json profiles/phase_a_traffic.json
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04z_bench_line.json 2> gpurun_out/r04z_bench_line.err; echo "full line rc=$?"; python tools/bench_brief.py full < gpurun_out/r04z_bench_line.json
timeout -k 10 300 python bench.py --contigs 500000 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/r04z_bench_line_500k.json 2> gpurun_out/r04z_500k.err; python tools/bench_brief.py 500k < gpurun_out/r04z_bench_line_500k.json
