# round-6 end state: everything DESIGN.md / profiles/ quote for the final build, one GPU-box session:
#   prof_full r06z (kernel stats + FETCH / WRITE passes -> r06z.md, r06z_traffic.json), the SQ-counter passes, the 500k / long / 5M lines
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/prof_full.sh r06z > gpurun_out/r06z_prof.log 2>&1 || { tail -5 gpurun_out/r06z_prof.log; exit 1; }
echo "prof_full done"; head -12 gpurun_out/r06z.md
bash tools/pmc_sq_passes.sh > gpurun_out/r06z_pmcsq.log 2>&1 && cp gpurun_out/pmcsq_table.txt gpurun_out/r06z_pmc_sq.txt && echo "SQ passes done" || { echo "SQ passes failed"; tail -3 gpurun_out/r06z_pmcsq.log; }
for spec in "500k:--contigs 500000" "long:--workload long" "5m:--contigs 5000000 --steps 10 --warmup 2"; do
  tag=${spec%%:*}; args=${spec#*:}
  # shellcheck disable=SC2086
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-e2e --soak-seconds 0 $args > gpurun_out/r06z_bench_line_$tag.json 2> gpurun_out/r06z_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/r06z_$tag.err; }
  python tools/bench_brief.py $tag < gpurun_out/r06z_bench_line_$tag.json
done
