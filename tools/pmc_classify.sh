# usage (GPU box): bash tools/pmc_classify.sh -- SQ counters of the generateGraph kernels (stream B alone: PALACE_BENCH_SKIP_EREF=1)
: "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export PALACE_BENCH_SKIP_EREF=1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_INSTS SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp -d "$GRAFT_REPO_ROOT"/gpurun_out/pmccl/g$i --output-format csv -- python3 "$GRAFT_REPO_ROOT"/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --soak-seconds 0 > "$GRAFT_REPO_ROOT"/gpurun_out/pmccl_g$i.log 2>&1 || { tail -5 "$GRAFT_REPO_ROOT"/gpurun_out/pmccl_g$i.log; exit 1; }
  echo "group $i done"
done
cd "$GRAFT_REPO_ROOT" && python3 tools/pmc_table.py gpurun_out/pmccl --filter graph_classify
