# constants of palace_amd/multigpu.py's cost model at the end of round 4: the count launch of rank 0's key share for W = 1, 2, 4, 8
# (PALACE_OPT_KEY_SHARE; results are then partial, the refs check is skipped by the bench for such runs), with stage 04 beside it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for w in 1 2 4 8; do
  PALACE_OPT_KEY_SHARE=$w timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04z7.err | python tools/bench_brief.py share_1_of_$w
done
PALACE_BENCH_FINAL=0 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04z7.err | python tools/bench_brief.py three_planes
