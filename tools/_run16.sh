cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in product nolds noload nov noloadnov; do
  echo "=== $n"
  if [ $n = product ]; then python bench.py --steps 5 --warmup 2 --train-steps 0 --no-cpu-baseline --breakdown 2>&1 >/dev/null | grep -E "ms/frame|shaded"
  else MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so python bench.py --steps 5 --warmup 2 --train-steps 0 --no-cpu-baseline --breakdown 2>&1 >/dev/null | grep -E "ms/frame"; fi
done > gpurun_out/r6_frame_ablations.txt 2>&1
cat gpurun_out/r6_frame_ablations.txt
