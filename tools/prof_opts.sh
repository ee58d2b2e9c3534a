# usage (GPU box): bash tools/prof_opts.sh OPT v1 v2 ...  -> per-kernel medians with PALACE_OPT_<OPT>=v for each v
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
opt=$1; shift
for v in "$@"; do
  rm -rf gpurun_out/po_$v
  export PALACE_OPT_$opt=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/po_$v --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/po_$v.json 2> gpurun_out/po_$v.err || { echo "$v failed"; tail -3 gpurun_out/po_$v.err; continue; }
  s=$(find gpurun_out/po_$v -name '*kernel_stats.csv' | head -1)
  t=$(find gpurun_out/po_$v -name '*kernel_trace.csv' | head -1)
  python3 tools/rocprof_summary.py gpurun_out/po_$v.md --stats $s --trace $t
  echo "== $opt=$v"
  sed -n '/per-kernel duration/,$p' gpurun_out/po_$v.md | grep -E "bin1|bin2|streams|usable|lds_count|mark_" | head -8
done
