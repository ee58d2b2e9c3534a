# usage (GPU box): bash tools/diag_small.sh -> refs reported / row digest of the 20000-contig step for every form of the reads, 1 and 4 ranks
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
A="--contigs 20000 --refs 200 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --soak-seconds 0"
show() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1]); print(sys.argv[1], d['n_gpus'], d['config']['refs_reported'], d['config']['result_digest'])" $1; }
for r in ascii packed; do for f in 0 1; do
  PALACE_BENCH_FINAL=$f timeout -k 10 120 python bench.py $A --reads $r > gpurun_out/diag_w1_${r}_$f.json 2> gpurun_out/diag_w1_${r}_$f.err && show gpurun_out/diag_w1_${r}_$f.json
done; done
for r in ascii packed; do
  PALACE_BENCH_ONE_DEVICE=1 PALACE_BENCH_BACKEND=gloo timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 $A --reads $r > gpurun_out/diag_w4_$r.json 2> gpurun_out/diag_w4_$r.err && show gpurun_out/diag_w4_$r.json
done
