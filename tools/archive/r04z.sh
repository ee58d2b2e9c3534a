# round-4 end state: profiles (1M and 500k), the full bench line, the other workloads' lines, the fused-probe and lag/hold variants,
# stream B alone, the chain's traces
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp && bash "$R"/tools/prof_full.sh r04z > "$R"/gpurun_out/r04z_prof.log 2>&1; tail -2 "$R"/gpurun_out/r04z_prof.log
cd /tmp && BENCH_EXTRA='--contigs 500000' TRAFFIC_CONTIGS=500000 bash "$R"/tools/prof_full.sh r04z_500k > "$R"/gpurun_out/r04z_500k_prof.log 2>&1; tail -1 "$R"/gpurun_out/r04z_500k_prof.log
cd "$R"
cp gpurun_out/r04z_traffic.json profiles/phase_a_traffic.json
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04z_bench_line.json 2> gpurun_out/r04z_bench_line.err; echo "full line rc=$?"; python tools/bench_brief.py full < gpurun_out/r04z_bench_line.json
timeout -k 10 300 python bench.py --contigs 500000 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/r04z_bench_line_500k.json 2> gpurun_out/r04z_500k.err; python tools/bench_brief.py 500k < gpurun_out/r04z_bench_line_500k.json
timeout -k 10 400 python bench.py --contigs 5000000 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/r04z_bench_line_5m.json 2> gpurun_out/r04z_5m.err; python tools/bench_brief.py 5m < gpurun_out/r04z_bench_line_5m.json
timeout -k 10 300 python bench.py --workload long --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/r04z_bench_line_long.json 2> gpurun_out/r04z_long.err; python tools/bench_brief.py long < gpurun_out/r04z_bench_line_long.json
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04z.err | python tools/bench_brief.py default.$rep
  timeout -k 10 300 python bench.py --fused-probe 1 --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04z.err | python tools/bench_brief.py fused_probe.$rep
  timeout -k 10 300 python bench.py --graph-lag 1 --stage04-hold l2 --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04z.err | python tools/bench_brief.py lag_l2.$rep
  PALACE_BENCH_DIAG_SKIP=stage04 timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04z.err | python tools/bench_brief.py no_stage04.$rep
done
