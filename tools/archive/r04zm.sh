# why is eref's second run 0.5 s on some boxes and 0.25 s on others: first run (builds the 2.4 GB index file), second run traced, three times
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
W=$(mktemp -d /tmp/palace_r04zm.XXXXXX) || exit 1
export PALACE_BENCH_WORK_DIR="$W" PALACE_BENCH_KEEP=1
timeout -k 10 400 python bench.py --steps 1 --warmup 1 --soak-seconds 0 --no-cpu-baseline > gpurun_out/r04zm_keep.json 2> gpurun_out/r04zm_keep.err || { tail -5 gpurun_out/r04zm_keep.err; exit 1; }
B=palace_amd/bin
t() { s=$(date +%s%N); "$@"; e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms"; }
grep -a Dirty /proc/meminfo; cat /proc/sys/vm/dirty_ratio /proc/sys/vm/dirty_background_ratio; df -h "$W" | tail -1
for i in 1 2 3; do
  rm -f $W/phagedb.fa.k32.index.dat $W/phagedb.fa.genome.len.txt
  echo -n "first "; t $B/eref $W/reads_1.fq $W/reads_2.fq $W/phagedb.fa $W/s_tmp.txt 0.9 0.85 16 > $W/refs.txt
  grep -a Dirty /proc/meminfo
  echo -n "second "; PALACE_TRACE=1 t $B/eref $W/reads_1.fq $W/reads_2.fq $W/phagedb.fa $W/s_tmp.txt 0.9 0.85 16 > $W/refs.txt 2> gpurun_out/r04zm_second_$i.err
  grep -a "^\[" gpurun_out/r04zm_second_$i.err | cut -c1-110
  echo -n "third "; t $B/eref $W/reads_1.fq $W/reads_2.fq $W/phagedb.fa $W/s_tmp.txt 0.9 0.85 16 > $W/refs.txt
done
rm -rf -- "$W"
