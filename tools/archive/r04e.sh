# round-4 GPU batch e: fused probe v3 (prefetch at start, hit list in LDS), classify with per-wave appends, stage-04 hold points
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_eref.py tests/test_gpu_graph_abi.py tests/test_gpu_graph_fuzz.py tests/test_gpu_configs.py::test_config4_long_contigs_full_size_and_oracle_sample -x -q -k "probe or scan or stdout or final or bench_shaped or fuzz or adversarial or config4" > gpurun_out/r04e_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04e_tests.log
run() { # tag env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04e.err | python tools/bench_brief.py $tag || { echo "$tag failed"; tail -5 gpurun_out/r04e.err; }
}
for rep in 1 2; do
  run old.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=0
  run fused.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=1
  run late_l2.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=0 PALACE_BENCH_STAGE04_LATE=l2
  run late_1.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=0 PALACE_BENCH_STAGE04_LATE=1
  run late_l2_fused.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=1 PALACE_BENCH_STAGE04_LATE=l2
  run late_1_fused.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=1 PALACE_BENCH_STAGE04_LATE=1
  run s32x_late_l2.$rep PALACE_BENCH_STAGE04_CUS=32 PALACE_BENCH_FUSED_PROBE=0 PALACE_BENCH_STAGE04_LATE=l2
done
