# the inflate kernel with its window in the output buffer (28 waves per CU instead of 4): parity tests, then the whole 1M-contig BAM
# on the device against the host (bin/gpuinflate), then generateGraph with 0 / 1 / 2 device helpers
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_inflate.py -x -q > gpurun_out/r04zd_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04zd_tests.log
W=$(mktemp -d /tmp/palace_r04zd.XXXXXX) || exit 1
export PALACE_BENCH_WORK_DIR="$W" PALACE_BENCH_KEEP=1
timeout -k 10 400 python bench.py --steps 1 --warmup 1 --soak-seconds 0 --no-cpu-baseline > gpurun_out/r04zd_keep.json 2> gpurun_out/r04zd_keep.err || { tail -5 gpurun_out/r04zd_keep.err; exit 1; }
B=palace_amd/bin
timeout -k 10 120 $B/gpuinflate $W/reads_pe_primary.sort.bam 16; echo "gpuinflate rc=$?"
timeout -k 10 120 $B/gpuinflate $W/reads_pe_primary.sort.bam 16
t() { s=$(date +%s%N); "$@"; e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms"; }
for rep in 1 2; do
for dev in 0 2 1; do
  export PALACE_BAM_DEVICE=$dev
  tag="dev${dev}_$rep"
  echo -n "$tag generateGraph "; PALACE_TRACE=1 t $B/generateGraph $W/reads_pe_primary.sort.bam $W/assembly_graph.fastg.fai $W/t_graph_$tag.txt 5.0 2> gpurun_out/r04zd_gg_$tag.err
  grep -a "bam/device\|record boundaries\|bam header\|bam records\|members were" gpurun_out/r04zd_gg_$tag.err | cut -c1-200
done; done
md5sum $W/t_graph_*.txt | awk '{print $1}' | sort | uniq -c
rm -rf -- "$W"
