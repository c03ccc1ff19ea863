cd $GRAFT_REPO_ROOT
( echo "== colour (ReLU)"; MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_stamp1.so python tools/chunk_timeline.py 10 3 ) > gpurun_out/r6_chunk_timeline2.txt 2>&1
cat gpurun_out/r6_chunk_timeline2.txt
