# usage (GPU box): bash tools/ab_reads.sh  -> step / count time of the resident step for the forms the reads can have, same box
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
run() {   # tag, env..., -- bench args
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-e2e --no-cpu-baseline --soak-seconds 0 $BENCH_ARGS > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "$tag failed"; return 1; }
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:24s} step {d['ms_per_step']:.3f} ms  count {d['stage_ms']['eref_count_both_sides']:.3f} ms  scan {d['stage_ms']['eref_scan_refs']:.3f}  frac {d['roofline']['frac']:.4f}  refs {d['config']['refs_reported']} digest {d['config']['result_digest']['eref_rows']}")
PY
}
BENCH_ARGS="--reads ascii" run ascii_all_planes PALACE_BENCH_FINAL=0 &&
BENCH_ARGS="--reads ascii" run ascii_final PALACE_BENCH_FINAL=1 &&
BENCH_ARGS="--reads packed" run packed_all_planes PALACE_BENCH_FINAL=0 &&
BENCH_ARGS="--reads packed" run packed_final PALACE_BENCH_FINAL=1
[ -n "$AB_PPL" ] && for p in $AB_PPL; do BENCH_ARGS="--reads packed" run packed_final_ppl$p PALACE_BENCH_FINAL=1 PALACE_OPT_BIN1_PPL=$p; done
true
[ -n "$AB_DEPTH" ] && { BENCH_ARGS="--reads packed --batches-in-flight 1" run packed_final_depth1 PALACE_BENCH_FINAL=1; BENCH_ARGS="--reads packed --batches-in-flight 2" run packed_final_depth2 PALACE_BENCH_FINAL=1; }
true
[ -n "$AB_PARTS" ] && for p in $AB_PARTS; do BENCH_ARGS="--reads packed" run packed_final_parts$p PALACE_BENCH_FINAL=1 PALACE_OPT_LEVEL1_PARTS=$p; done
true
