# A/B runs of bench.py on ONE GPU box (box-to-box spread, +-0.15 ms, is larger than most effects compared), variants alternated.
#   usage (GPU box):  bash tools/ab.sh <tag> <reps> <variant> [<variant> ...]
#   variant = label[,ENV=VALUE ...][@extra bench.py arguments]        e.g.
#     bash tools/ab.sh r06x 3 default grid2048,PALACE_OPT_DECOMP_GRID=2048 "long@--workload long"
# One line per run on stdout (tools/bench_brief.py: ms per step, M contigs/s, stage times); full JSON lines in gpurun_out/<tag>_ab.jsonl.
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; reps=$2; shift 2
STEPS=${AB_STEPS:-30}
for rep in $(seq 1 "$reps"); do
  for variant in "$@"; do
    spec=${variant%%@*}; extra=""; [ "$spec" != "$variant" ] && extra=${variant#*@}
    label=${spec%%,*}; envs=""; [ "$label" != "$spec" ] && envs=$(echo "${spec#*,}" | tr ',' ' ')
    # shellcheck disable=SC2086
    env $envs timeout -k 10 300 python bench.py --steps "$STEPS" --warmup 3 --no-cpu-baseline --no-e2e --soak-seconds 0 $extra 2> gpurun_out/${tag}_ab.err \
      | tee -a gpurun_out/${tag}_ab.jsonl | python tools/bench_brief.py "$label.$rep" || { echo "$label.$rep FAILED"; tail -3 gpurun_out/${tag}_ab.err; exit 1; }
  done
done
