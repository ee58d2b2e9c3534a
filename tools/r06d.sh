# round 6: the two count-launch experiments as library variants (tools/build_variant.sh: gate = membership-gated counting in the fused count
# kernel, -DPALACE_GATE=1; bigrows = level-2 staging rows of 144 slots, one workgroup per CU, -DPALACE_BIN2_BIG=1): the eref suite on each
# (parity), then the step alternated on one box; then the record-walk A/B of generateGraph (tools/r06c.sh)
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in gate bigrows; do
  PALACE_HIP_SO=$PWD/tools/ab/lib_$v.so timeout -k 10 400 python -m pytest tests/test_gpu_eref.py -x -q -m gpu > gpurun_out/r06d_tests_$v.log 2>&1; rc=$?
  echo "$v: $(tail -1 gpurun_out/r06d_tests_$v.log)"
  [ $rc -eq 0 ] || { tail -30 gpurun_out/r06d_tests_$v.log; exit $rc; }
done
AB_STEPS=30 bash tools/ab.sh r06d 3 default "gate,PALACE_HIP_SO=$PWD/tools/ab/lib_gate.so" "bigrows,PALACE_HIP_SO=$PWD/tools/ab/lib_bigrows.so" | tee gpurun_out/r06d_variants.log
bash tools/r06c.sh | tee gpurun_out/r06c_walk_ab.log
