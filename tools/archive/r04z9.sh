# stage 04's rounds held back until level 1 of the count launch is done (they then run beside level 2, the count kernel and
# Phase B): against the default and the l2 hold; alternated, 30 steps each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2 3; do
  for hold in 0 l1 l2; do
    timeout -k 10 300 python bench.py --stage04-hold $hold --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04z9.err | python tools/bench_brief.py hold_$hold.$rep
  done
done
