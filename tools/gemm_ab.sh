#!/bin/bash
# GPU box: tools/gemm_ab.sh name...  -> tools/gemm_microbench.py per ablation side library (tools/ab_build.sh)
cd "$(dirname "$0")/.."
for n in "$@"; do echo "== $n"; MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so python tools/gemm_microbench.py 49500 2>&1 | grep gemm_; done
