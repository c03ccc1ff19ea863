cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_mlp_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_mlp_tests.txt
python tools/mlp_microbench.py 4000000 sdf > gpurun_out/r6_sdf_microbench.txt 2>&1
python tools/mlp_microbench.py 4000000 sdf >> gpurun_out/r6_sdf_microbench.txt 2>&1
python tools/sampler_precision.py 8 > gpurun_out/r6_sampler_precision.log 2>&1
cat gpurun_out/r6_mlp_tests.txt gpurun_out/r6_sdf_microbench.txt; tail -8 gpurun_out/r6_sampler_precision.log
