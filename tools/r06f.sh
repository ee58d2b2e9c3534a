# round 6, two level-2 experiments as library variants against the tree: flushlds = the row flush with count / destination from LDS broadcast
# reads and the address in vector registers (review item 1b; profiles/r06f_bin2_flush_lds_experiment.patch), bin2nt = the level-1 records read
# with non-temporal loads (profiles/r06f_bin2_nt_loads_experiment.patch).  Parity (eref suite), the step alternated on one box, then SQ
# instruction counters, WRITE_SIZE and the kernel's duration for each
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in flushlds bin2nt; do
  PALACE_HIP_SO=$PWD/tools/ab/lib_$v.so timeout -k 10 400 python -m pytest tests/test_gpu_eref.py -x -q -m gpu > gpurun_out/r06f_tests_$v.log 2>&1; rc=$?
  echo "$v: $(tail -1 gpurun_out/r06f_tests_$v.log)"; [ $rc -eq 0 ] || { tail -30 gpurun_out/r06f_tests_$v.log; exit $rc; }
done
AB_STEPS=30 bash tools/ab.sh r06f 3 default "flushlds,PALACE_HIP_SO=$PWD/tools/ab/lib_flushlds.so" "bin2nt,PALACE_HIP_SO=$PWD/tools/ab/lib_bin2nt.so" | tee gpurun_out/r06f_variants.log
for v in default flushlds bin2nt; do
  unset PALACE_HIP_SO; [ $v = default ] || export PALACE_HIP_SO=$PWD/tools/ab/lib_$v.so
  rm -rf gpurun_out/r06f_pmc_$v gpurun_out/r06f_wr_$v gpurun_out/r06f_st_$v
  timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -d gpurun_out/r06f_pmc_$v --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --soak-seconds 0 > /dev/null 2> gpurun_out/r06f_pmc_$v.err || { echo "pmc $v failed"; tail -3 gpurun_out/r06f_pmc_$v.err; }
  timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r06f_wr_$v --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --soak-seconds 0 > /dev/null 2> gpurun_out/r06f_wr_$v.err || echo "write pass $v failed"
  echo "== $v"; python3 tools/pmc_table.py gpurun_out/r06f_pmc_$v gpurun_out/r06f_wr_$v --filter eref_bin2
  python3 tools/pmc_table.py gpurun_out/r06f_wr_$v --filter eref_lds_count
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/r06f_st_$v --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --soak-seconds 0 > /dev/null 2> gpurun_out/r06f_st_$v.err
  grep -h "eref_bin2_kernel\|eref_lds_count\|eref_bin1" $(find gpurun_out/r06f_st_$v -name '*kernel_stats.csv' | head -1) | cut -c1-140
done 2>&1 | tee gpurun_out/r06f_pmc.log
