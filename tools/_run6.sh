cd $GRAFT_REPO_ROOT
bash tools/ab_run.sh 4000000 product early > gpurun_out/r6_early_ab_kernels.txt 2>&1
cat gpurun_out/r6_early_ab_kernels.txt
