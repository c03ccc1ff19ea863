#!/bin/bash
# round 5, GPU call 3: fused loss, fused background net, launch count of the training iteration
cd "$(dirname "$0")/.."
O=gpurun_out/r05b3; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( timeout 600 python -m pytest -x -q -m gpu tests/test_loss_gpu.py -s 2>&1 | grep -E "parity|passed|failed|Error|error|assert" | tail -40 ) > $O/loss_tests.txt 2>&1
( timeout 900 python -m pytest -x -q -m gpu tests/test_train_gpu.py -k "fused_background or background_branch" -s 2>&1 | grep -E "grad parity|passed|failed|Error|error|assert" | tail -60 ) > $O/bg_tests.txt 2>&1
( timeout 1200 python -m pytest -x -q -m gpu tests/test_train_step_gpu.py tests/test_state_dict_gpu.py tests/test_pipeline_gpu.py 2>&1 | tail -25 ) > $O/train_tests.txt 2>&1
( timeout 300 python tools/train_bench.py 20 4 ) > $O/train_bench.txt 2>&1
( cd /tmp; rm -rf /tmp/tk; timeout 600 rocprofv3 --kernel-trace -d /tmp/tk -o tk -- python $R/tools/train_bench.py 20 4 > /tmp/tk.log 2>&1 || tail -5 /tmp/tk.log
  DB=$(find /tmp/tk -name '*.db' | head -1); python $R/tools/rocpd_summary.py $DB $R/$O/train_kernels.txt | tail -3 ) > $O/train_trace.log 2>&1
ls -la $O
