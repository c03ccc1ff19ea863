#!/bin/bash
# round 6, final GPU session: the round's rocprofv3 set, the bench lines of BASELINE.md's table, the training trace, shard latencies
cd "$(dirname "$0")/.."
O=gpurun_out/r06final; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( timeout 1200 bash tools/profile_round.sh r06 ) > $O/profile_round.log 2>&1
( cd /tmp; rm -rf /tmp/tk; timeout 600 rocprofv3 --kernel-trace -d /tmp/tk -o tk -- python $R/tools/train_bench.py 60 4 > $R/$O/train_bench_traced.txt 2>&1
  DB=$(find /tmp/tk -name '*.db' | head -1); python $R/tools/rocpd_summary.py $DB $R/$O/train_kernels.txt | tail -3 ) > $O/train_trace.log 2>&1
( timeout 300 python tools/train_bench.py 60 4 ) > $O/train_bench.txt 2>&1
( timeout 1500 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
( timeout 300 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline --sampler-sdf f16 --breakdown ) > $O/bench_f16sampler.json 2> $O/bench_f16sampler.err
( timeout 300 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline --breakdown ) > $O/bench_breakdown.json 2> $O/bench_breakdown.err
( timeout 700 python bench.py --steps 5 --warmup 2 --persons 4 --samples 256 --train-steps 5 --train-warmup 2 --cpu-rays 1024 --cpu-train-iters 1 --breakdown ) > $O/bench_4p256.json 2> $O/bench_4p256.err
( timeout 600 python bench.py --steps 10 --warmup 3 --persons 1 --samples 64 --train-steps 10 --cpu-rays 4096 --cpu-train-iters 1 --breakdown ) > $O/bench_1p64.json 2> $O/bench_1p64.err
( timeout 600 python tools/shard_latency.py 5 ) > $O/shard_latency.txt 2>&1
ls -la $O gpurun_out/prof_r06
