# usage (GPU box): bash tools/prof_variants.sh [rows]  -> per-kernel medians of a 3-step bench run for every tools/ab/lib_*.so
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for f in $(ls "$GRAFT_REPO_ROOT"/tools/ab/lib_*.so | sort); do
  v=$(basename $f .so)
  rm -rf gpurun_out/pv_$v
  PALACE_HIP_SO=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pv_$v --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/pv_$v.json 2> gpurun_out/pv_$v.err || { echo "$v failed"; tail -3 gpurun_out/pv_$v.err; continue; }
  s=$(find gpurun_out/pv_$v -name '*kernel_stats.csv' | head -1)
  t=$(find gpurun_out/pv_$v -name '*kernel_trace.csv' | head -1)
  python3 tools/rocprof_summary.py gpurun_out/pv_$v.md --stats $s --trace $t
  echo "== $v"
  sed -n '/per-kernel duration/,$p' gpurun_out/pv_$v.md | grep -E "bin1|bin2|streams|usable|lds_count|mark_" | head -${1:-8}
done
