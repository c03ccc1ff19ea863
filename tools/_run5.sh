cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in dmaonly nomfma; do
  echo "== $n"; MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so python tools/mlp_microbench.py 4000000 all 2>&1 | grep Mpts
done > gpurun_out/r6_ablate2.txt 2>&1
cat gpurun_out/r6_ablate2.txt
