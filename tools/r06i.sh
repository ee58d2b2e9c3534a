# round 6: the decomposition's grid once more, after the slot-word change (PALACE_OPT_DECOMP_GRID; the one-GPU bench's default is 256): first session 128 .. 2048,
# this one 32 / 64 / 96
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
AB_STEPS=30 bash tools/ab.sh r06i2 3 default g32,PALACE_OPT_DECOMP_GRID=32 g64,PALACE_OPT_DECOMP_GRID=64 g96,PALACE_OPT_DECOMP_GRID=96 | cut -c1-400 | tee gpurun_out/r06i2_variants.log
