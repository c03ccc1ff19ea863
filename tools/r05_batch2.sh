#!/bin/bash
# round 5, GPU call 2: precise sampler kernel, sigma-from-h forward sweep, launch diet of the training iteration
cd "$(dirname "$0")/.."
O=gpurun_out/r05b2; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( timeout 600 python -m pytest -x -q -m gpu tests/test_mlp_gpu.py -s 2>&1 | grep -E "parity|passed|failed|Error|error" | tail -40 ) > $O/mlp_tests.txt 2>&1
( MP_LIB_PATH="$R/multiply_amd/ab_libs/libmultiply_hip_sigh.so" timeout 600 python -m pytest -x -q -m gpu tests/test_mlp_gpu.py -s 2>&1 | grep -E "parity|passed|failed|Error|error" | tail -40 ) > $O/mlp_tests_sigh.txt 2>&1
for n in base sigh; do
  lib="$R/multiply_amd/libmultiply_hip.so"; [ $n != base ] && lib="$R/multiply_amd/ab_libs/libmultiply_hip_$n.so"
  ( MP_LIB_PATH="$lib" timeout 300 python tools/sig_bits.py 20000 ) 2>&1 | tail -1 >> $O/sig_bits.txt
done
for rep in 1 2; do for n in base sigh; do
  lib="$R/multiply_amd/libmultiply_hip.so"; [ $n != base ] && lib="$R/multiply_amd/ab_libs/libmultiply_hip_$n.so"
  echo "== $n (rep $rep)" >> $O/sigh_ab.txt
  ( MP_LIB_PATH="$lib" timeout 300 python tools/mlp_microbench.py 4000000 shade ) 2>&1 | grep -E "Mpts" >> $O/sigh_ab.txt
done; done
( timeout 900 python -m pytest -x -q -m gpu tests/test_train_step_gpu.py tests/test_train_gpu.py tests/test_state_dict_gpu.py 2>&1 | tail -15 ) > $O/train_tests.txt 2>&1
( timeout 300 python tools/train_bench.py 20 4 ) > $O/train_bench.txt 2>&1
( cd /tmp; rm -rf /tmp/tk; timeout 600 rocprofv3 --kernel-trace -d /tmp/tk -o tk -- python $R/tools/train_bench.py 20 4 > /tmp/tk.log 2>&1 || tail -5 /tmp/tk.log
  DB=$(find /tmp/tk -name '*.db' | head -1); python $R/tools/rocpd_summary.py $DB $R/$O/train_kernels.txt | tail -3 ) > $O/train_trace.log 2>&1
( timeout 900 python tools/sampler_precision.py 8 ) > $O/sampler_precision.log 2>&1
cp gpurun_out/sampler_precision.txt $O/ 2>/dev/null
( timeout 300 python bench.py --steps 5 --warmup 2 --train-steps 0 --no-cpu-baseline --breakdown ) > $O/bench_f16.json 2> $O/bench_f16.err
( MP_SAMPLER_SDF=bf16x3 timeout 300 python bench.py --steps 5 --warmup 2 --train-steps 0 --no-cpu-baseline --breakdown ) > $O/bench_precise.json 2> $O/bench_precise.err
( MP_LIB_PATH="$R/multiply_amd/ab_libs/libmultiply_hip_sigh.so" timeout 300 python bench.py --steps 5 --warmup 2 --train-steps 0 --no-cpu-baseline --breakdown ) > $O/bench_sigh.json 2> $O/bench_sigh.err
ls -la $O
