# round-5 end state: everything DESIGN.md / profiles/ quote for this round, in one GPU-box session.
#   bash tools/r05z.sh            -> gpurun_out/r05z*.{md,json,log}
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/prof_full.sh r05z > gpurun_out/r05z_prof.log 2>&1 || { tail -5 gpurun_out/r05z_prof.log; exit 1; }
echo "prof_full done"; head -30 gpurun_out/r05z.md
# A/B on this box: the default step; Phase B probing for itself; what stage 04 / its matching rounds cost the count launch
# (the lines of the DIAG_SKIP runs fail their own checks on purpose); stream B alone with the decomposition on 256 / 2048 workgroups
AB_STEPS=30 bash tools/ab.sh r05z 3 default "nofuse@--fused-probe 0" skip_stage04,PALACE_BENCH_DIAG_SKIP=stage04 skip_match,PALACE_BENCH_DIAG_SKIP=match \
    wide,PALACE_OPT_DECOMP_GRID=2048 streamB_256,PALACE_BENCH_SKIP_EREF=1 streamB_2048,PALACE_BENCH_SKIP_EREF=1,PALACE_OPT_DECOMP_GRID=2048 | tee gpurun_out/r05z_variants.log
for spec in "500k:--contigs 500000" "long:--workload long" "5m:--contigs 5000000 --steps 10 --warmup 2"; do
  tag=${spec%%:*}; args=${spec#*:}
  # shellcheck disable=SC2086
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-e2e --soak-seconds 0 $args > gpurun_out/r05z_bench_line_$tag.json 2> gpurun_out/r05z_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/r05z_$tag.err; }
  python tools/bench_brief.py $tag < gpurun_out/r05z_bench_line_$tag.json
done
