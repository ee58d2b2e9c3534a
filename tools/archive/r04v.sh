# round-4 batch v: what the partition kernels' own reservations (returning global atomics) cost them -- variants with fake offsets (results wrong)
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for f in $(ls "$GRAFT_REPO_ROOT"/tools/ab/lib_*.so | sort); do
  v=$(basename $f .so)
  rm -rf "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v
  PALACE_BENCH_DIAG_SKIP=stage04 PALACE_HIP_SO=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v --output-format csv -- python3 "$GRAFT_REPO_ROOT"/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v.json 2> "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v.err
  s=$(find "$GRAFT_REPO_ROOT"/gpurun_out/pv_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python3 - "$s" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("bin1","bin2","lds_count")): print("  ", r["Name"].split("(")[0][:60], r["Calls"], round(float(r["AverageNs"])/1e6,3))
PY
done
