# round 6: the all-schemes mode of bench.py --gpus 4 rehearsed in full on one GPU (gloo, every rank on device 0): seven configurations, each a child process per rank
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
s=$(date +%s)
PALACE_BENCH_ONE_DEVICE=1 PALACE_BENCH_BACKEND=gloo timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 4 --contigs 200000 --steps 10 --warmup 2 > gpurun_out/r06j_line.json 2> gpurun_out/r06j.err; echo "exit $? after $(( $(date +%s) - s )) s"
grep "bench rank 0" gpurun_out/r06j.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06j_line.json") if l.startswith("{")][-1])
print(d["config"]["parallelism"]); print(json.dumps(d["parallelism_measured"], indent=1)); print("failed", d.get("failed_checks"), "value", d["value"], "weak", d["weak"]["value"])
PY
