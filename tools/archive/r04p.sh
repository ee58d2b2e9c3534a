# round-4 final measurements: profiles (1M and 500k), bench lines (1M full, 5M, long), stream B alone, fused-probe A/B, e2e traces
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && bash "$GRAFT_REPO_ROOT"/tools/prof_full.sh r04p > "$GRAFT_REPO_ROOT"/gpurun_out/r04p_prof.log 2>&1; tail -3 "$GRAFT_REPO_ROOT"/gpurun_out/r04p_prof.log
cd /tmp && BENCH_EXTRA='--contigs 500000' TRAFFIC_CONTIGS=500000 bash "$GRAFT_REPO_ROOT"/tools/prof_full.sh r04p_500k > "$GRAFT_REPO_ROOT"/gpurun_out/r04p_500k_prof.log 2>&1; tail -2 "$GRAFT_REPO_ROOT"/gpurun_out/r04p_500k_prof.log
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/r04p_traffic.json profiles/phase_a_traffic.json          # (on the box only: so that the full line below quotes this build's traffic)
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04p_bench_line.json 2> gpurun_out/r04p_bench_line.err; echo "full line rc=$?"; python tools/bench_brief.py full < gpurun_out/r04p_bench_line.json
timeout -k 10 300 python bench.py --contigs 5000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/r04p_bench_line_5m.json 2> gpurun_out/r04p_5m.err; python tools/bench_brief.py 5m < gpurun_out/r04p_bench_line_5m.json
timeout -k 10 300 python bench.py --workload long --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/r04p_bench_line_long.json 2> gpurun_out/r04p_long.err; python tools/bench_brief.py long < gpurun_out/r04p_bench_line_long.json
for rep in 1 2; do
  PALACE_BENCH_FUSED_PROBE=0 timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04p.err | python tools/bench_brief.py probe_kernel.$rep
  PALACE_BENCH_FUSED_PROBE=1 timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04p.err | python tools/bench_brief.py fused.$rep
done
cd /tmp && PALACE_BENCH_SKIP_EREF=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/gpurun_out/r04p_streamB --output-format csv -- python3 "$GRAFT_REPO_ROOT"/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > "$GRAFT_REPO_ROOT"/gpurun_out/r04p_streamB.json 2> "$GRAFT_REPO_ROOT"/gpurun_out/r04p_streamB.err
cd "$GRAFT_REPO_ROOT"; f=$(find gpurun_out/r04p_streamB -name '*kernel_stats.csv' | head -1); python3 tools/rocprof_summary.py gpurun_out/r04p_streamB.md --stats $f --note "stream B alone (PALACE_BENCH_SKIP_EREF=1): generateGraph + stage 04 kernels without the counting kernels beside them"; grep -E "depth_select|graph_classify|compact|resolve" gpurun_out/r04p_streamB.md | head -8
bash tools/e2e_trace.sh r04p > gpurun_out/r04p_e2e.log 2>&1; head -14 gpurun_out/r04p_e2e.log
