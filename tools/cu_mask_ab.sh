# usage (GPU box): bash tools/cu_mask_ab.sh [reps] -- the step with the generateGraph + matching stream (B) and / or the eref stream (A)
# confined to subsets of the compute units (hipExtStreamCreateWithCUMask), alternated on the SAME box
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
ALL=ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff
RR=0101010101010101010101010101010101010101010101010101010101010101      # every 8th bit: one XCD if the mask is dealt round-robin over the XCDs
run() { # tag maskA maskB
  PALACE_BENCH_CU_MASK_A=$2 PALACE_BENCH_CU_MASK_B=$3 timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 2> gpurun_out/cu_mask.err | python tools/bench_brief.py $1 || { echo "$1 failed"; tail -5 gpurun_out/cu_mask.err; }
}
for rep in $(seq 1 ${1:-2}); do
  run base.$rep "" ""
  run B8.$rep "" ff
  run B32.$rep "" ffffffff
  run B64.$rep "" ffffffffffffffff
  run Brr32.$rep "" $RR
  run A-32_B32.$rep ffffffffffffffffffffffffffffffffffffffffffffffffffffffff00000000 ffffffff
  run A-rr_Brr.$rep fefefefefefefefefefefefefefefefefefefefefefefefefefefefefefefefe $RR
  run A-8_B8.$rep ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff00 ff
done
