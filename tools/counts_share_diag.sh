# usage (GPU box): bash tools/counts_share_diag.sh [contigs]  -> stream A of ONE rank of W under shard_counts on one GPU: the count launch of a
# 1/W share of the reads (partial entry counts of the whole DB's index) and the indexed scan of 1/W of the refs (the model's constants,
# palace_amd/multigpu.py MODEL / STEP; results are partial by design: the line fails its own checks)
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
C=${1:-1000000}
for w in 2 4 7 8; do
  PALACE_OPT_COUNTS_SHARE=$w timeout -k 10 300 python bench.py --contigs $C --steps 20 --warmup 2 --no-e2e --no-cpu-baseline --soak-seconds 0 > gpurun_out/cs_${C}_$w.json 2> gpurun_out/cs_${C}_$w.err
  python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']; print('contigs', sys.argv[3], 'share 1/' + sys.argv[2], 'step', round(d['ms_per_step'],3), 'count', round(s['eref_count_both_sides'],3), 'merge', round(s['eref_table_merge'],3), 'scan', round(s['eref_scan_refs'],3), 'stage04', round(s['graph_filter_and_matching_on_device'],3))" gpurun_out/cs_${C}_$w.json $w $C || tail -3 gpurun_out/cs_${C}_$w.err
done
