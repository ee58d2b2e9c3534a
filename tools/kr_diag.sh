# usage (GPU box): bash tools/kr_diag.sh -> count-launch time of ONE rank's share when the key space is split W ways (all reads, one GPU)
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for w in 1 2 4 8 16; do
  PALACE_OPT_KEY_SHARE=$w timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-e2e --no-cpu-baseline --soak-seconds 0 > gpurun_out/kr_$w.json 2> gpurun_out/kr_$w.err
  python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']; print('share 1/' + sys.argv[2], 'step', round(d['ms_per_step'],3), 'count', round(s['eref_count_both_sides'],3))" gpurun_out/kr_$w.json $w
done
