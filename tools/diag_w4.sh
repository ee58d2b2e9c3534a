# usage (GPU box): bash tools/diag_w4.sh <reads form> <runs>  -> refs reported by repeated 4-rank rehearsals on one GPU (gloo)
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
A="--contigs 20000 --refs 200 --steps 3 --warmup 0 --no-cpu-baseline --no-e2e --soak-seconds 0"
for i in $(seq 1 ${2:-6}); do
  PALACE_BENCH_SCHEME=shard_reads PALACE_BENCH_ONE_DEVICE=1 PALACE_BENCH_BACKEND=gloo timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port $((29540 + i)) bench.py --gpus 4 $A --reads $1 > gpurun_out/w4_$1_$i.json 2> gpurun_out/w4_$1_$i.err || { echo "run $i failed"; exit 1; }
  python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1]); print(sys.argv[1], d['config']['refs_reported'], d['config']['result_digest']['eref_rows'])" gpurun_out/w4_$1_$i.json
done
