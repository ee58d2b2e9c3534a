# usage (GPU box): bash tools/ab_bench.sh [reps]  -- alternates every tools/ab/lib_*.so on the SAME box (box-to-box spread is
# larger than the few-percent effects being compared) and prints the per-step count medians
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in $(seq 1 ${1:-3}); do
  for f in $(ls "$GRAFT_REPO_ROOT"/tools/ab/lib_*.so | sort); do
    v=$(basename $f .so)
    PALACE_HIP_SO=$f timeout -k 10 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/ab.err | python tools/bench_median.py $v.$rep
  done
done
