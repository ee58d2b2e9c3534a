# the 5M-contig step: 10 steps behind 2 warm-up steps (as r04p / r04w), twice; then 5 behind 1
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
timeout -k 10 400 python bench.py --contigs 5000000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/r04z6_5m_$rep.json 2> gpurun_out/r04z6_5m.err; python tools/bench_brief.py 5m.$rep < gpurun_out/r04z6_5m_$rep.json
done
timeout -k 10 400 python bench.py --contigs 5000000 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > gpurun_out/r04z6_5m_3.json 2> gpurun_out/r04z6_5m.err; python tools/bench_brief.py 5m.short < gpurun_out/r04z6_5m_3.json
