# round-4 GPU batch l: which part of graph_depth_select_kernel takes its 0.69 ms (variants), stream B alone
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp PALACE_BENCH_SKIP_EREF=1
for f in $(ls "$GRAFT_REPO_ROOT"/tools/ab/lib_*.so | sort); do
  v=$(basename $f .so)
  rm -rf "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v
  PALACE_HIP_SO=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v --output-format csv -- python3 "$GRAFT_REPO_ROOT"/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v.json 2> "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v.err
  s=$(find "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; grep -E "depth_select|graph_classify" $s | cut -d, -f1-4 | cut -c1-120
done
