#!/bin/bash
# round 5, GPU call 1: targeted tests of the round's host-side changes + the cheap kernel experiments (one box, one call)
cd "$(dirname "$0")/.."
O=gpurun_out/r05b1; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest -x -q -m gpu \
    tests/test_train_step_gpu.py::test_two_live_training_graphs_keep_their_gradients_apart \
    "tests/test_train_step_gpu.py::test_training_forward_loss_and_all_parameter_gradients" \
    tests/test_geom_gpu.py::test_device_hull_give_up_path_falls_back_to_the_host_hull \
    tests/test_parallel_gpu.py::test_data_parallel_sampler_vote_reproduces_the_single_process_depths \
    tests/test_parallel_gpu.py::test_ray_sharded_render_and_dp_training_match_the_oracle \
    "tests/test_render_gpu.py::test_full_size_frame_properties[configs3_4p_256]" \
    tests/test_bench_gpu.py -s 2>&1 | tail -60 ) > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt
( timeout 120 tools/bin/mphase_model ) > $O/mphase_model.txt 2>&1
for n in base sig6 sig4; do
  lib="$PWD/multiply_amd/libmultiply_hip.so"; [ $n != base ] && lib="$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so"
  ( MP_LIB_PATH="$lib" timeout 300 python tools/sig_bits.py 20000 ) 2>&1 | tail -1 >> $O/sig_bits.txt
done
( MP_LIB_PATH="$PWD/multiply_amd/ab_libs/libmultiply_hip_onem0.so" timeout 600 python -m pytest -x -q -m gpu tests/test_mlp_gpu.py 2>&1 | tail -3 ) > $O/onem0_tests.txt 2>&1
for rep in 1 2; do for n in base onem0; do
  lib="$PWD/multiply_amd/libmultiply_hip.so"; [ $n != base ] && lib="$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so"
  echo "== $n (rep $rep)" >> $O/onem0_ab.txt
  ( MP_LIB_PATH="$lib" timeout 300 python tools/mlp_microbench.py 4000000 ) 2>&1 | grep -E "Mpts" >> $O/onem0_ab.txt
done; done
( timeout 900 python tools/sampler_precision.py 8 ) > $O/sampler_precision.log 2>&1
cp gpurun_out/sampler_precision.txt $O/ 2>/dev/null
( timeout 300 python tools/train_bench.py 20 3 ) > $O/train_bench.txt 2>&1
ls -la $O
