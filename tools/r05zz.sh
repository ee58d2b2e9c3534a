# round-5 end state, second half of the round (Phase B inside the count launch, the forked start back in force): everything DESIGN.md /
# profiles/ quote for it, in two GPU-box sessions.
#   bash tools/r05zz.sh a   -> prof_full r05zz (stats + FETCH / WRITE passes), the --fused-probe 2 / 1 / 0 A/B, the 500k / long / 5M lines
#   bash tools/r05zz.sh b   -> the default bench run (cpu_baseline + e2e), the e2e trace
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
case "${1:-a}" in
a)
  bash tools/prof_full.sh r05zz > gpurun_out/r05zz_prof.log 2>&1 || { tail -5 gpurun_out/r05zz_prof.log; exit 1; }
  echo "prof_full done"; head -30 gpurun_out/r05zz.md
  AB_STEPS=30 bash tools/ab.sh r05zz 3 default "fuse1@--fused-probe 1" "nofuse@--fused-probe 0" | tee gpurun_out/r05zz_variants.log
  for spec in "500k:--contigs 500000" "long:--workload long" "5m:--contigs 5000000 --steps 10 --warmup 2"; do
    tag=${spec%%:*}; args=${spec#*:}
    # shellcheck disable=SC2086
    timeout -k 10 600 python bench.py --no-cpu-baseline --no-e2e --soak-seconds 0 $args > gpurun_out/r05zz_bench_line_$tag.json 2> gpurun_out/r05zz_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/r05zz_$tag.err; }
    python tools/bench_brief.py $tag < gpurun_out/r05zz_bench_line_$tag.json
  done ;;
b)
  timeout -k 10 900 python bench.py > gpurun_out/r05zz_bench_line.json 2> gpurun_out/r05zz_bench.err || { echo "default run failed"; tail -5 gpurun_out/r05zz_bench.err; }
  python tools/bench_brief.py default < gpurun_out/r05zz_bench_line.json
  bash tools/e2e_trace.sh r05zz > gpurun_out/r05zz_e2e_trace.log 2>&1; head -20 gpurun_out/r05zz_e2e_trace.log ;;
esac
