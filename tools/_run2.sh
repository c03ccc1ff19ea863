cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
E=$PWD/multiply_amd/ab_libs/libmultiply_hip_early.so
for rep in 1 2; do
echo "== default"; python tools/mlp_microbench.py 4000000 all 2>&1 | grep Mpts
echo "== early";   MP_LIB_PATH=$E python tools/mlp_microbench.py 4000000 all 2>&1 | grep Mpts
done > gpurun_out/r6_early_ab.txt 2>&1
MP_LIB_PATH=$E python -m pytest tests/test_mlp_gpu.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/r6_early_ab.txt
cat gpurun_out/r6_early_ab.txt
