#!/bin/bash
# GPU box: HBM-side traffic of the TRAINING iteration's kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only)
cd "$(dirname "$0")/.."
O=gpurun_out/train_pmc; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp; rm -rf /tmp/tp_*
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tp_fetch -o run -- python $R/tools/train_bench.py 6 2 > /dev/null 2> $R/$O/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tp_write -o run -- python $R/tools/train_bench.py 6 2 > /dev/null 2> $R/$O/write.log
python $R/tools/pmc_traffic.py /tmp/tp_fetch /tmp/tp_write 2>&1 | grep -v "^{" | head -40 > $R/$O/traffic.txt
cat $R/$O/traffic.txt | head -12
