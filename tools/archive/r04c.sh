# round-4 GPU batch: fused probe + stage-04 stream A/B, the new tests
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04c_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04c_smoke.log
python -m pytest tests/test_gpu_eref.py -x -q -k "probe or scan or stdout or final" > gpurun_out/r04c_eref.log 2>&1; echo "eref tests rc=$?"; tail -5 gpurun_out/r04c_eref.log
run() { # tag env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04c.err | python tools/bench_brief.py $tag || { echo "$tag failed"; tail -5 gpurun_out/r04c.err; }
}
for rep in 1 2; do
  run old.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=0
  run fused.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=1
  run s32.$rep PALACE_BENCH_STAGE04_CUS=32 PALACE_BENCH_FUSED_PROBE=0
  run s16.$rep PALACE_BENCH_STAGE04_CUS=16 PALACE_BENCH_FUSED_PROBE=0
  run s64.$rep PALACE_BENCH_STAGE04_CUS=64 PALACE_BENCH_FUSED_PROBE=0
  run both32.$rep PALACE_BENCH_STAGE04_CUS=32 PALACE_BENCH_FUSED_PROBE=1
  run both16.$rep PALACE_BENCH_STAGE04_CUS=16 PALACE_BENCH_FUSED_PROBE=1
done
python -m pytest tests/test_gpu_pipeline.py -x -q --durations=8 > gpurun_out/r04c_pipeline.log 2>&1; echo "pipeline rc=$?"; tail -15 gpurun_out/r04c_pipeline.log
