# round-4 batch r: where to hold stage 04's rounds back now that the classify kernel no longer disturbs the counting kernels
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04r.err | python tools/bench_brief.py $tag || { echo "$tag failed"; tail -3 gpurun_out/r04r.err; }; }
for rep in 1 2; do
  run default.$rep X=1
  run late_l2.$rep PALACE_BENCH_STAGE04_LATE=l2
  run late_1.$rep PALACE_BENCH_STAGE04_LATE=1
  run late_l2_fused.$rep PALACE_BENCH_STAGE04_LATE=l2 PALACE_BENCH_FUSED_PROBE=1
  run late_1_fused.$rep PALACE_BENCH_STAGE04_LATE=1 PALACE_BENCH_FUSED_PROBE=1
done
