# round-4 batch u: what a stream of random gathers / atomics / stores beside the counting kernels costs them (stage 04 left out of the timed steps)
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp PALACE_BENCH_DIAG_SKIP=stage04
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04u.err | python tools/bench_brief.py $tag || { echo "$tag failed"; tail -3 gpurun_out/r04u.err; }; }
run none X=1
# 40 M operations per step in 70 launches over 8 M slots of 8 bytes (64 MB: the size of the matching state), 256 workgroups
run gather8 PALACE_BENCH_DISTURB=0:40000000:70:8000000:256
run atomic64 PALACE_BENCH_DISTURB=1:40000000:70:8000000:256
run store8 PALACE_BENCH_DISTURB=2:40000000:70:8000000:256
run gather1 PALACE_BENCH_DISTURB=3:40000000:70:8000000:256
run atomic64_10M PALACE_BENCH_DISTURB=1:10000000:70:8000000:256
run gather8_160M PALACE_BENCH_DISTURB=0:160000000:70:8000000:256
run empty_launches PALACE_BENCH_DISTURB=0:70:70:8000000:256
run none2 X=1
