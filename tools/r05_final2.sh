#!/bin/bash
# round 5, final GPU session, second half (the first stopped at an oracle NameError): the -m gpu suite, the bench lines with CPU legs
cd "$(dirname "$0")/.."
O=gpurun_out/r05final; mkdir -p $O
export TMPDIR=/tmp
( timeout 2400 python -m pytest -q -m gpu tests 2>&1 | tail -25 ) > $O/gpu_suite.txt 2>&1
cp gpurun_out/parity_4k.txt $O/ 2>/dev/null
( timeout 1200 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
( timeout 900 python bench.py --steps 5 --warmup 2 --persons 4 --samples 256 --train-steps 5 --train-warmup 2 --cpu-rays 1024 --cpu-train-iters 1 --breakdown ) > $O/bench_4p256.json 2> $O/bench_4p256.err
( timeout 900 python bench.py --steps 10 --warmup 3 --persons 1 --samples 64 --train-steps 10 --cpu-rays 4096 --cpu-train-iters 1 --breakdown ) > $O/bench_1p64.json 2> $O/bench_1p64.err
( timeout 600 python tools/sampler_precision.py 8 0.01 ) > $O/sampler_precision_beta.log 2>&1
cp gpurun_out/sampler_precision_beta0.01.txt $O/ 2>/dev/null
ls -la $O
