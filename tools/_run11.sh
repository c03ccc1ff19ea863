cd $GRAFT_REPO_ROOT
AB_WHICH=all bash tools/ab_run.sh 4000000 product late nowait nowaitlate noload 2>&1 | grep -v "Mpts\|forward-mode" > gpurun_out/r6_ablate_wait.txt
cat gpurun_out/r6_ablate_wait.txt
