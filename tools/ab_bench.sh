# usage (GPU box): bash tools/ab_bench.sh  -- alternates tools/ab/lib_A.so and lib_B.so on the SAME box (box-to-box spread is
# larger than the few-percent effects being compared), prints the per-step count medians
for rep in 1 2 3; do
  for v in A B; do
    PALACE_HIP_SO=$GRAFT_REPO_ROOT/tools/ab/lib_$v.so timeout -k 10 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline 2> gpurun_out/ab.err | python tools/bench_median.py $v$rep
  done
done
