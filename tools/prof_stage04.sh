# usage (GPU box): bash tools/prof_stage04.sh  -> per-kernel stats of the generateGraph + stage-04 stream alone (eref skipped)
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PALACE_BENCH_SKIP_EREF=1
rm -rf gpurun_out/prof_s4
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s4 --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/prof_s4.json 2> gpurun_out/prof_s4.err || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_s4/*/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "palace" in r["Name"] or "rocclr" in r["Name"]]
tot=0
for r in rows[:40]:
    print(r["Name"].replace("palace::(anonymous namespace)::","")[:60], r["Calls"], round(float(r["AverageNs"])/1e3,1),"us avg", round(float(r["TotalDurationNs"])/1e6/6,3),"ms/step")
PY
