# round 6: generateGraph on the 1M-contig sample's files, the record walk with the segments walked ahead by the inflate threads (default)
# against the walker alone (PALACE_BAM_SERIAL_WALK=1), alternated on one box; walls to stdout, the last traces to gpurun_out/r06c_*.err
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
W=$(mktemp -d /tmp/palace_r06c.XXXXXX) || exit 1
[ -n "$W" ] && [ -d "$W" ] || { echo "no work dir"; exit 1; }
export PALACE_BENCH_WORK_DIR="$W" PALACE_BENCH_KEEP=1
python bench.py --steps 1 --warmup 1 --soak-seconds 0 --no-cpu-baseline > gpurun_out/r06c_keep.json 2> gpurun_out/r06c_keep.err || { tail -5 gpurun_out/r06c_keep.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r06c_keep.json"))
e = d["e2e"]
print("bench e2e:", round(e["seconds"], 3), e["stage_s"], "fused", e["one_process_stage04"].get("seconds"), e["one_process_stage04"].get("stage_s"))
PY
B=palace_amd/bin
t() { s=$(date +%s%N); "$@"; e=$(date +%s%N); echo -n " $(( (e - s) / 1000000 ))"; }
gg() { PALACE_TRACE=1 $B/generateGraph $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/g_$1.txt 5.0 2> gpurun_out/r06c_gg_$1.err; }
ggf() { PALACE_TRACE=1 $B/generateGraph --hit-seqs $W/hit_seqs.out --node-scores $W/node_scores.out --blast $W/assembly_graph.fasta.blast --fasta-fai $W/assembly_graph.fasta.fai --paths $W/contigs.paths --filtered-pre $W/x_pre_$1 --filtered $W/x_filt_$1 --all-hit-segs $W/x_hits_$1 --linear $W/x_lin_$1 --cycle $W/x_cyc_$1 --cycle-nodup $W/x_nodup_$1 --all-result $W/x_all_$1 -s -i 10 $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/x_graph_$1 5.0 2> gpurun_out/r06c_ggf_$1.err; }
for rep in 1 2 3 4 5; do
  echo -n "rep $rep generateGraph ms: ahead"; t gg ahead; echo -n "  serial"; PALACE_BAM_SERIAL_WALK=1 t gg serial
  echo -n "   fused: ahead"; t ggf ahead; echo -n "  serial"; PALACE_BAM_SERIAL_WALK=1 t ggf serial; echo
done
cmp $W/g_ahead.txt $W/g_serial.txt && cmp $W/x_all_ahead $W/x_all_serial && cmp $W/x_filt_ahead $W/x_filt_serial && echo "outputs identical ($(wc -l < $W/g_ahead.txt) graph lines)"
for k in gg_ahead gg_serial; do echo "---- $k"; grep -a "^\[bam" gpurun_out/r06c_$k.err | head -12; done
rm -rf -- "$W"
