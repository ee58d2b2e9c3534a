# usage (GPU box): bash tools/eref_cli_repeat.sh  -> eref CLI on the bench's 1M-contig files, several runs back to back per slab size
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out /tmp/e2e_keep
d=$(mktemp -d /tmp/e2e_keep/palace_e2e_repeat.XXXXXX) || exit 1
[ -n "$d" ] && [ -d "$d" ] || { echo "no work dir"; exit 1; }
PALACE_BENCH_WORK_DIR="$d" PALACE_BENCH_KEEP=1 timeout -k 10 600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --soak-seconds 0 > gpurun_out/rep_bench.json 2> gpurun_out/rep_bench.err
echo "inputs in $d"; ls $d | head -20
for slab in 268435456 1073741824 268435456 1073741824; do
  for i in 1 2 3; do
    t0=$(date +%s.%N)
    PALACE_TRACE=1 PALACE_EREF_SLAB=$slab palace_amd/bin/eref $d/reads_1.fq $d/reads_2.fq $d/phagedb.fa $d/s_tmp.txt 0.9 0.85 16 > /tmp/e2e_keep/out_${slab}_$i.txt 2> gpurun_out/rep_${slab}_$i.err
    t1=$(date +%s.%N)
    echo "slab=$slab wall=$(python3 -c "print(round($t1-$t0,3))") $(grep -E 'hip runtime up' gpurun_out/rep_${slab}_$i.err | tr -s ' ') $(md5sum < /tmp/e2e_keep/out_${slab}_$i.txt | cut -c1-8)"
  done
done
