# round-4 batch t: grid size of the decomposition kernels (workgroups per launch) against what they cost the counting kernels
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for f in $(ls tools/ab/lib_g*.so | sort); do
    v=$(basename $f .so)
    PALACE_HIP_SO=$GRAFT_REPO_ROOT/$f timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04t.err | python tools/bench_brief.py $v.$rep || { echo "$v failed"; tail -3 gpurun_out/r04t.err; }
  done
done
