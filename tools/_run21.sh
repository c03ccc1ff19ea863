cd $GRAFT_REPO_ROOT
MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_geomprof.so python tools/geom_prof.py 512 2>&1 | grep -v amdgpu > gpurun_out/r6_geom_prof.txt
cat gpurun_out/r6_geom_prof.txt
