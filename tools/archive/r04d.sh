# round-4 GPU batch: fused probe (reordered) and stage 04 on CUs of its own, A/B on one box
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_eref.py -x -q -k "probe or scan or stdout or final" > gpurun_out/r04d_eref.log 2>&1; echo "eref tests rc=$?"; tail -3 gpurun_out/r04d_eref.log
run() { # tag env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/r04d.err | python tools/bench_brief.py $tag || { echo "$tag failed"; tail -5 gpurun_out/r04d.err; }
}
for rep in 1 2; do
  run old.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=0
  run fused.$rep PALACE_BENCH_STAGE04_CUS=0 PALACE_BENCH_FUSED_PROBE=1
  run s32x.$rep PALACE_BENCH_STAGE04_CUS=32 PALACE_BENCH_FUSED_PROBE=0
  run s16x.$rep PALACE_BENCH_STAGE04_CUS=16 PALACE_BENCH_FUSED_PROBE=0
  run s8x.$rep PALACE_BENCH_STAGE04_CUS=8 PALACE_BENCH_FUSED_PROBE=0
  run s64x.$rep PALACE_BENCH_STAGE04_CUS=64 PALACE_BENCH_FUSED_PROBE=0
  run s32x_fused.$rep PALACE_BENCH_STAGE04_CUS=32 PALACE_BENCH_FUSED_PROBE=1
  run s16x_fused.$rep PALACE_BENCH_STAGE04_CUS=16 PALACE_BENCH_FUSED_PROBE=1
  run s32x_noprio.$rep PALACE_BENCH_STAGE04_CUS=32 PALACE_BENCH_FUSED_PROBE=0 PALACE_BENCH_PRIO=0
done
