#!/bin/bash
# GPU box: for every side library (or the names given) run the MLP micro-benchmark under a rocprofv3 kernel trace and print
# the per-kernel table.   tools/ab_run.sh [n_points] [name ...]   -> gpurun_out/ab_<name>.txt
cd "$(dirname "$0")/.."
N="${1:-4000000}"; shift
names="$@"
[ -z "$names" ] && names=$(ls multiply_amd/ab_libs/ | sed 's/libmultiply_hip_\(.*\)\.so/\1/')
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in $names; do
  lib="$PWD/multiply_amd/ab_libs/libmultiply_hip_$n.so"
  rm -rf /tmp/ab_$n
  ( cd /tmp && MP_LIB_PATH="$lib" rocprofv3 --kernel-trace -d /tmp/ab_$n -o run -- python "$OLDPWD/tools/mlp_microbench.py" "$N" ${AB_WHICH:-all} ) > gpurun_out/ab_$n.log 2>&1
  db=$(find /tmp/ab_$n -name "*.db" | head -1)
  echo "=== $n"; grep -E "Mpts/s" gpurun_out/ab_$n.log
  python tools/rocpd_summary.py "$db" gpurun_out/ab_$n.txt | grep -E "k_mlp|k_background" 
done
