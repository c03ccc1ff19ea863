cd $GRAFT_REPO_ROOT
bash tools/ab_run.sh 4000000 product noload nov nolds noloadnov dmaonly nomfma 2>&1 | grep -v "Mpts\|forward-mode" > gpurun_out/r6_ablate_kernels.txt
cat gpurun_out/r6_ablate_kernels.txt
