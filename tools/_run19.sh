cd $GRAFT_REPO_ROOT
for rep in 1 2; do for n in stash32 v16 u16; do
  echo "== $n"; MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so python tools/train_bench.py 60 4 2>&1 | grep ms_per_iter | cut -c1-200
done; done > gpurun_out/r6_stash_ab.txt 2>&1
for n in v16 u16; do echo "== $n tests"; MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so python -m pytest tests/test_train_gpu.py -q -m gpu -k fused_sdf 2>&1 | grep -E "passed|failed|rel-to-max" | tail -4; MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so python -m pytest tests/test_train_step_gpu.py -q -m gpu -rA -k "all_parameter or benchmarked" 2>&1 | grep -E "passed|failed|worst relative|bench workload" | cut -c1-260; done >> gpurun_out/r6_stash_ab.txt 2>&1
cat gpurun_out/r6_stash_ab.txt
