cd $GRAFT_REPO_ROOT
python -m pytest tests/test_mlp_gpu.py tests/test_render_gpu.py -x -q -m gpu 2>&1 | tail -3
cp multiply_amd/libmultiply_hip.so multiply_amd/ab_libs/libmultiply_hip_new.so
AB_WHICH=all bash tools/ab_run.sh 4000000 product new product new 2>&1 | grep -v "forward-mode\|k_mlp_shade " > gpurun_out/r6_prologue_ab.txt
cat gpurun_out/r6_prologue_ab.txt
