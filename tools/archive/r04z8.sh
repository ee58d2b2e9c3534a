# where stage 04 starts, once more at the end of the round (its rounds are 1 ms shorter than when this was measured): default /
# behind the partition kernels / behind the whole count launch; alternated, 30 steps each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
  for hold in 0 l2 1; do
    timeout -k 10 300 python bench.py --stage04-hold $hold --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04z8.err | python tools/bench_brief.py hold_$hold.$rep
  done
done
