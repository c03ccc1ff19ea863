cd $GRAFT_REPO_ROOT
./tools/bin/pair_model_agpr > gpurun_out/r6_pair_model_agpr.txt 2>&1
cat gpurun_out/r6_pair_model_agpr.txt
