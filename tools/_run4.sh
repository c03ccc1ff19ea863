cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in "" noload nov nolds noloadnov; do
  if [ -z "$n" ]; then echo "== product"; python tools/mlp_microbench.py 4000000 all 2>&1 | grep Mpts
  else echo "== $n"; MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so python tools/mlp_microbench.py 4000000 all 2>&1 | grep Mpts; fi
done > gpurun_out/r6_ablate.txt 2>&1
cat gpurun_out/r6_ablate.txt
