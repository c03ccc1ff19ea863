cd $GRAFT_REPO_ROOT
AB_WHICH=all bash tools/ab_run.sh 4000000 product split product split 2>&1 | grep -v "forward-mode\|k_mlp_shade " > gpurun_out/r6_split_ab.txt
cat gpurun_out/r6_split_ab.txt
MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_split.so python -m pytest tests/test_mlp_gpu.py -x -q -m gpu 2>&1 | tail -2
