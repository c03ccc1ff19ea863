cd $GRAFT_REPO_ROOT
( echo "== colour"; MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_stamp1.so python tools/tile_timeline.py color ) > gpurun_out/r6_tile_timeline_color.txt 2>&1
grep -v "chunk " gpurun_out/r6_tile_timeline_color.txt
