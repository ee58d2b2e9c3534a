# round-4 final measurements, default mode (count launch = Phase A alone): profile, full bench line, 500k profile, stage-04 diagnosis
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && bash "$GRAFT_REPO_ROOT"/tools/prof_full.sh r04q > "$GRAFT_REPO_ROOT"/gpurun_out/r04q_prof.log 2>&1; tail -2 "$GRAFT_REPO_ROOT"/gpurun_out/r04q_prof.log
cd /tmp && BENCH_EXTRA='--contigs 500000' TRAFFIC_CONTIGS=500000 bash "$GRAFT_REPO_ROOT"/tools/prof_full.sh r04q_500k > "$GRAFT_REPO_ROOT"/gpurun_out/r04q_500k_prof.log 2>&1; tail -1 "$GRAFT_REPO_ROOT"/gpurun_out/r04q_500k_prof.log
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/r04q_traffic.json profiles/phase_a_traffic.json
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04q_bench_line.json 2> gpurun_out/r04q_bench_line.err; echo "full line rc=$?"; python tools/bench_brief.py full < gpurun_out/r04q_bench_line.json
timeout -k 10 300 python bench.py --contigs 500000 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/r04q_bench_line_500k.json 2> gpurun_out/r04q_500k.err; python tools/bench_brief.py 500k < gpurun_out/r04q_bench_line_500k.json
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04q.err | python tools/bench_brief.py all.$rep
  PALACE_BENCH_DIAG_SKIP=match timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04q.err | python tools/bench_brief.py no_match.$rep
  PALACE_BENCH_DIAG_SKIP=stage04 timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04q.err | python tools/bench_brief.py no_stage04.$rep
done
