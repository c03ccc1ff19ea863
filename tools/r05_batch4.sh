#!/bin/bash
# round 5, GPU call 4: regularisers, launch diet after the loss / background fusion, shard latency on the round's kernels
cd "$(dirname "$0")/.."
O=gpurun_out/r05b4; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( timeout 1200 python -m pytest -x -q -m gpu tests/test_train_step_gpu.py tests/test_loss_gpu.py tests/test_bench_gpu.py::test_single_gpu_line_has_roofline_and_cpu_baseline -s 2>&1 | grep -E "parity\]|passed|failed|Error|error|assert" | tail -40 ) > $O/tests.txt 2>&1
( timeout 300 python tools/train_bench.py 20 4 ) > $O/train_bench.txt 2>&1
( cd /tmp; rm -rf /tmp/tk; timeout 600 rocprofv3 --kernel-trace -d /tmp/tk -o tk -- python $R/tools/train_bench.py 20 4 > /tmp/tk.log 2>&1 || tail -5 /tmp/tk.log
  DB=$(find /tmp/tk -name '*.db' | head -1); python $R/tools/rocpd_summary.py $DB $R/$O/train_kernels.txt | tail -3 ) > $O/train_trace.log 2>&1
( timeout 600 python tools/shard_latency.py 5 ) > $O/shard_latency.txt 2>&1
ls -la $O
