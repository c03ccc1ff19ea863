cd $GRAFT_REPO_ROOT
AB_WHICH=all bash tools/ab_run.sh 4000000 product dma1lane noload 2>&1 | grep -v "forward-mode\|k_mlp_shade \|Mpts" > gpurun_out/r6_dma1lane.txt
cat gpurun_out/r6_dma1lane.txt
