# device inflate helpers + huge pages for the inflated stream: tests, then generateGraph (plain and fused) on the 1M-contig sample's
# files under the four settings, twice each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cat /sys/kernel/mm/transparent_hugepage/enabled
timeout -k 10 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_stage04.py -x -q > gpurun_out/r04z3_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04z3_tests.log
W=$(mktemp -d /tmp/palace_r04z3.XXXXXX) || exit 1
export PALACE_BENCH_WORK_DIR="$W" PALACE_BENCH_KEEP=1
timeout -k 10 400 python bench.py --steps 1 --warmup 1 --soak-seconds 0 --no-cpu-baseline > gpurun_out/r04z3_keep.json 2> gpurun_out/r04z3_keep.err || { tail -5 gpurun_out/r04z3_keep.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r04z3_keep.json"))
e = d["e2e"]
print("bench e2e:", round(e["seconds"], 3), e["stage_s"], "fused", e["one_process_stage04"].get("seconds"), e["one_process_stage04"].get("stage_s"), "ok", e.get("agrees_with_resident_step"), e["one_process_stage04"].get("files_identical_to_the_chain"))
PY
B=palace_amd/bin
t() { s=$(date +%s%N); "$@"; e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms"; }
for rep in 1 2; do
for dev in 0 2 1; do for small in "" 1; do
  export PALACE_BAM_DEVICE=$dev
  if [ -n "$small" ]; then export PALACE_BAM_SMALL_PAGES=1; else unset PALACE_BAM_SMALL_PAGES; fi
  tag="dev${dev}_small${small:-0}_$rep"
  echo -n "$tag generateGraph "; PALACE_TRACE=1 t $B/generateGraph $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/t_graph_$tag.txt 5.0 2> gpurun_out/r04z3_gg_$tag.err
  echo -n "$tag fused "; PALACE_TRACE=1 t $B/generateGraph --hit-seqs $W/hit_seqs.out --node-scores $W/node_scores.out --blast $W/assembly_graph.fasta.blast --fasta-fai $W/assembly_graph.fasta.fai --paths $W/contigs.paths --filtered-pre $W/x_pre --filtered $W/x_filt --all-hit-segs $W/x_hits --linear $W/x_lin --cycle $W/x_cyc --cycle-nodup $W/x_nodup --all-result $W/x_all_$tag -s -i 10 $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/x_graph 5.0 2> gpurun_out/r04z3_ggf_$tag.err
  grep -a "members were inflated\|record boundaries\|bam records" gpurun_out/r04z3_gg_$tag.err gpurun_out/r04z3_ggf_$tag.err | cut -c1-160
done; done; done
md5sum $W/t_graph_*.txt $W/x_all_* | awk '{print $1}' | sort | uniq -c
rm -rf -- "$W"
