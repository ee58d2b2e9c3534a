# round 6: s_setprio in the late phases of the partition kernels (profiles/r06h_setprio_experiment.patch).  Sessions 1-2: level 1 from its placement phase
# on (priority 2 or 3: -0.05 ms on the count launch, 6 of 6 repetitions), level 2's flush (nothing).  This session: WHERE level 1 raises it (behind the key /
# histogram phase = 2, the placement phase = 4, the sweep = 5) and the count kernel's probe epilogue on top
: "${GRAFT_REPO_ROOT:?}"; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/tools/ab
AB_STEPS=30 bash tools/ab.sh r06h3 5 default "p1at2,PALACE_HIP_SO=$L/lib_p1at2.so" "p1at4,PALACE_HIP_SO=$L/lib_p1at4.so" "p1at5,PALACE_HIP_SO=$L/lib_p1at5.so" "p1at4p3,PALACE_HIP_SO=$L/lib_p1at4p3.so" | cut -c1-120 | tee gpurun_out/r06h3_variants.log
