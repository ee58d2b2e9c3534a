# usage (GPU box): bash tools/e2e_trace.sh <tag>  -- the files of the 1M-contig sample once (bench.py, kept), then every executable of the
# chain and the fused generateGraph on them, twice each, with PALACE_TRACE=1; walls to stdout, traces to gpurun_out/<tag>_*.err
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-e2e}
mkdir -p gpurun_out
W=$(mktemp -d /tmp/palace_e2e_trace.XXXXXX) || exit 1
[ -n "$W" ] && [ -d "$W" ] || { echo "no work dir"; exit 1; }
export PALACE_BENCH_WORK_DIR="$W" PALACE_BENCH_KEEP=1
python bench.py --steps 1 --warmup 1 --soak-seconds 0 --no-cpu-baseline > gpurun_out/${tag}_keep.json 2> gpurun_out/${tag}_keep.err || { tail -5 gpurun_out/${tag}_keep.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_keep.json"))
e = d["e2e"]
print("bench e2e:", round(e["seconds"], 3), e["stage_s"], "fused", e["one_process_stage04"].get("seconds"), e["one_process_stage04"].get("stage_s"))
PY
B=palace_amd/bin; S=palace_amd/scripts
exec 3>&1          # (walls go to the script's stdout whatever a command's own stdout is redirected to)
t() { s=$(date +%s%N); PALACE_TRACE_T0=$s "$@"; e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms" >&3; }
for i in 1 2; do
  echo "== run $i"
  echo -n "eref "; PALACE_TRACE=1 t $B/eref $W/reads_1.fq $W/reads_2.fq $W/phagedb.fa $W/s_tmp.txt 0.9 0.85 16 > $W/refs.txt 2> gpurun_out/${tag}_eref_$i.err; tail -1 $W/refs.txt > /dev/null
  echo -n "generateGraph "; PALACE_TRACE=1 t $B/generateGraph $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/t_graph.txt 5.0 2> gpurun_out/${tag}_gg_$i.err
  echo -n "filter_graph.py "; PALACE_TRACE=1 t python $S/filter_graph.py $W/assembly_graph.fastg.fai $W/t_graph.txt $W/t_pre.txt 5.0 0 $W/hit_seqs.out $W/node_scores.out $W/assembly_graph.fasta.blast 0.7 $W/assembly_graph.fasta.fai $W/t_allhit.txt $W/contigs.paths 0.7 2> gpurun_out/${tag}_fg_$i.err
  echo -n "uniq "; t uniq $W/t_pre.txt > $W/t_filt.txt
  echo -n "matching "; PALACE_TRACE=1 t $B/matching -g $W/t_filt.txt -r $W/t_lin.txt -c $W/t_cyc.txt -s -i 10 -l $W/contigs.paths 2> gpurun_out/${tag}_m_$i.err
  echo -n "remove_cycle_dup.py "; t python $S/remove_cycle_dup.py $W/t_cyc.txt $W/t_nodup.txt > /dev/null
  echo -n "generateGraph(fused) "; PALACE_TRACE=1 t $B/generateGraph --hit-seqs $W/hit_seqs.out --node-scores $W/node_scores.out --blast $W/assembly_graph.fasta.blast --fasta-fai $W/assembly_graph.fasta.fai --paths $W/contigs.paths --filtered-pre $W/x_pre --filtered $W/x_filt --all-hit-segs $W/x_hits --linear $W/x_lin --cycle $W/x_cyc --cycle-nodup $W/x_nodup --all-result $W/x_all -s -i 10 $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/x_graph 5.0 2> gpurun_out/${tag}_ggf_$i.err
done
for k in eref gg fg m ggf; do echo "---- $k"; grep -a "^\[" gpurun_out/${tag}_${k}_2.err; done
rm -rf -- "$W"
