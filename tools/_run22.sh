cd $GRAFT_REPO_ROOT
AB_WHICH=shade bash tools/ab_run.sh 4000000 product nofeat product nofeat 2>&1 | grep -E "===|k_mlp_color"
