# usage (GPU box): bash tools/e2e_repeat.sh  -> writes the e2e input files once (bench.py, kept), then traces of the executables on them
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
W=$(mktemp -d /tmp/palace_e2e_repeat.XXXXXX) || exit 1          # this run's own directory: bench.py writes the files there and keeps them
[ -n "$W" ] && [ -d "$W" ] || { echo "no work dir"; exit 1; }
export PALACE_BENCH_WORK_DIR="$W" PALACE_BENCH_KEEP=1
python bench.py --steps 1 --warmup 1 --soak-seconds 0 --no-cpu-baseline > gpurun_out/e2e_keep.json 2> gpurun_out/e2e_keep.err || exit 1
echo "work dir $W"
for i in 1 2; do
  for m in packed ascii; do
    s=$(date +%s%N); PALACE_EREF_INPUT=$m PALACE_TRACE=1 palace_amd/bin/eref $W/reads_1.fq $W/reads_2.fq $W/phagedb.fa $W/s_tmp.txt 0.9 0.85 16 > $W/refs_$m.txt 2> gpurun_out/e2e_eref_${m}_$i.err; e=$(date +%s%N); echo "eref ($m reads) wall $(( (e - s) / 1000000 )) ms"
  done
  cmp $W/refs_packed.txt $W/refs_ascii.txt && echo "eref stdout identical, $(wc -l < $W/refs_packed.txt) lines"
  s=$(date +%s%N); PALACE_TRACE=1 palace_amd/bin/generateGraph --hit-seqs $W/hit_seqs.out --node-scores $W/node_scores.out --blast $W/assembly_graph.fasta.blast --fasta-fai $W/assembly_graph.fasta.fai --paths $W/contigs.paths --filtered-pre $W/x_pre --filtered $W/x_filt --all-hit-segs $W/x_hits --linear $W/x_lin --cycle $W/x_cyc --cycle-nodup $W/x_nodup --all-result $W/x_all -s -i 10 $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/x_graph 5.0 2> gpurun_out/e2e_gg_$i.err; e=$(date +%s%N); echo "generateGraph(fused) wall $(( (e - s) / 1000000 )) ms"
done
cat gpurun_out/e2e_eref_packed_2.err | grep -a "^\[" ; cat gpurun_out/e2e_eref_ascii_2.err | grep -a "^\[" ; cat gpurun_out/e2e_gg_2.err | grep -a "^\["
rm -rf -- "$W"
