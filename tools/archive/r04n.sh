# round-4 GPU batch n: device inflate against zlib, resident select kernel (tests + kernel times + A/B of the step on one box)
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_inflate.py -x -q > gpurun_out/r04n_inflate.log 2>&1; echo "inflate tests rc=$?"; tail -12 gpurun_out/r04n_inflate.log | cut -c1-400
python -m pytest tests/test_gpu_graph_abi.py tests/test_gpu_graph_fuzz.py tests/test_gpu_configs.py::test_config4_long_contigs_full_size_and_oracle_sample -x -q > gpurun_out/r04n_tests.log 2>&1; echo "graph tests rc=$?"; tail -4 gpurun_out/r04n_tests.log
cd /tmp && bash "$GRAFT_REPO_ROOT"/tools/prof_stats.sh > "$GRAFT_REPO_ROOT"/gpurun_out/r04n_stats.log 2>&1; cd "$GRAFT_REPO_ROOT"; grep -E "classify|depth_select|compact|bin1|bin2|lds_count" gpurun_out/prof_cur.md | head -12
for rep in 1 2; do
  for f in $(ls tools/ab/lib_a_base.so) default; do
    if [ "$f" = default ]; then unset PALACE_HIP_SO; tag=new; else export PALACE_HIP_SO=$GRAFT_REPO_ROOT/$f; tag=old_classify; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04n.err | python tools/bench_brief.py $tag.$rep
  done
done
# the device inflate on the 1M-contig sample's BAM (0.9 GB, ~36 000 members) against the host's
W=$(mktemp -d /tmp/palace_r04n.XXXXXX) && PALACE_BENCH_WORK_DIR="$W" PALACE_BENCH_KEEP=1 timeout -k 10 400 python bench.py --steps 1 --warmup 0 --soak-seconds 0 --no-cpu-baseline > gpurun_out/r04n_keep.json 2> gpurun_out/r04n_keep.err
ls -la "$W"/reads_pe_primary.sort.bam && timeout -k 10 200 palace_amd/bin/gpuinflate "$W"/reads_pe_primary.sort.bam 16
[ -n "$W" ] && rm -rf -- "$W"
