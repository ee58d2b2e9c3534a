# device helpers with batches of 8192 members and two fifths of what is left per claim: generateGraph (plain and fused) with 0 / 2 / 1
# helpers, alternated, three times; the device-inflate test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_cli.py -x -q -k "inflate or members" > gpurun_out/r04ze_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04ze_tests.log
W=$(mktemp -d /tmp/palace_r04ze.XXXXXX) || exit 1
export PALACE_BENCH_WORK_DIR="$W" PALACE_BENCH_KEEP=1
timeout -k 10 400 python bench.py --steps 1 --warmup 1 --soak-seconds 0 --no-cpu-baseline > gpurun_out/r04ze_keep.json 2> gpurun_out/r04ze_keep.err || { tail -5 gpurun_out/r04ze_keep.err; exit 1; }
B=palace_amd/bin
t() { s=$(date +%s%N); "$@"; e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms"; }
for rep in 1 2; do
for dev in 0 2 3 b4 b2; do
  unset PALACE_BAM_DEVICE_BATCH; case $dev in b4) export PALACE_BAM_DEVICE=2 PALACE_BAM_DEVICE_BATCH=4096;; b2) export PALACE_BAM_DEVICE=3 PALACE_BAM_DEVICE_BATCH=2560;; *) export PALACE_BAM_DEVICE=$dev;; esac
  tag="dev${dev}_$rep"
  echo -n "$tag generateGraph "; PALACE_TRACE=1 t $B/generateGraph $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/t_graph_$tag.txt 5.0 2> gpurun_out/r04ze_gg_$tag.err
  echo -n "$tag fused "; PALACE_TRACE=1 t $B/generateGraph --hit-seqs $W/hit_seqs.out --node-scores $W/node_scores.out --blast $W/assembly_graph.fasta.blast --fasta-fai $W/assembly_graph.fasta.fai --paths $W/contigs.paths --filtered-pre $W/x_pre --filtered $W/x_filt --all-hit-segs $W/x_hits --linear $W/x_lin --cycle $W/x_cyc --cycle-nodup $W/x_nodup --all-result $W/x_all_$tag -s -i 10 $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/x_graph 5.0 2> gpurun_out/r04ze_ggf_$tag.err
  grep -a "bam/device\|record boundaries\|bam header\|bam records\|members were\|side\]" gpurun_out/r04ze_gg_$tag.err gpurun_out/r04ze_ggf_$tag.err | cut -c20-200
done; done
md5sum $W/t_graph_*.txt $W/x_all_* | awk '{print $1}' | sort | uniq -c
rm -rf -- "$W"
