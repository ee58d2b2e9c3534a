# device inflate helpers, per-phase laps: generateGraph plain and fused on the 1M-contig sample's files; 0 / 1 / 2 helpers, alternated
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
W=$(mktemp -d /tmp/palace_r04z4.XXXXXX) || exit 1
export PALACE_BENCH_WORK_DIR="$W" PALACE_BENCH_KEEP=1
timeout -k 10 400 python bench.py --steps 1 --warmup 1 --soak-seconds 0 --no-cpu-baseline > gpurun_out/r04z4_keep.json 2> gpurun_out/r04z4_keep.err || { tail -5 gpurun_out/r04z4_keep.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r04z4_keep.json"))
e = d["e2e"]
print("bench e2e:", round(e["seconds"], 3), e["stage_s"], "fused", e["one_process_stage04"].get("seconds"), e["one_process_stage04"].get("stage_s"), "ok", e.get("agrees_with_resident_step"), e["one_process_stage04"].get("files_identical_to_the_chain"))
PY
B=palace_amd/bin
t() { s=$(date +%s%N); "$@"; e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms"; }
for rep in 1 2 3; do
for dev in 0 2 1; do
  export PALACE_BAM_DEVICE=$dev
  tag="dev${dev}_$rep"
  echo -n "$tag generateGraph "; PALACE_TRACE=1 t $B/generateGraph $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/t_graph_$tag.txt 5.0 2> gpurun_out/r04z4_gg_$tag.err
  echo -n "$tag fused "; PALACE_TRACE=1 t $B/generateGraph --hit-seqs $W/hit_seqs.out --node-scores $W/node_scores.out --blast $W/assembly_graph.fasta.blast --fasta-fai $W/assembly_graph.fasta.fai --paths $W/contigs.paths --filtered-pre $W/x_pre --filtered $W/x_filt --all-hit-segs $W/x_hits --linear $W/x_lin --cycle $W/x_cyc --cycle-nodup $W/x_nodup --all-result $W/x_all_$tag -s -i 10 $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/x_graph 5.0 2> gpurun_out/r04z4_ggf_$tag.err
  grep -a "bam/device\|record boundaries\|bam header\|bam records" gpurun_out/r04z4_gg_$tag.err gpurun_out/r04z4_ggf_$tag.err | cut -c20-190
done; done
echo -n "filter_graph.py "; PALACE_TRACE=1 t python palace_amd/scripts/filter_graph.py $W/assembly_graph.fastg.fai $W/t_graph_dev0_1.txt $W/t_pre.txt 5.0 0 $W/hit_seqs.out $W/node_scores.out $W/assembly_graph.fasta.blast 0.7 $W/assembly_graph.fasta.fai $W/t_allhit.txt $W/contigs.paths 0.7 2> gpurun_out/r04z4_fg.err; grep -a "^\[" gpurun_out/r04z4_fg.err
md5sum $W/t_graph_*.txt $W/x_all_* | awk '{print $1}' | sort | uniq -c
rm -rf -- "$W"
