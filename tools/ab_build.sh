#!/bin/bash
# Builds ablation side libraries (CPU container, hipcc cross-compiles):  tools/ab_build.sh name1:-DFLAG1 name2:"-DA -DB" ...
# -> multiply_amd/ab_libs/libmultiply_hip_<name>.so ; run them on the GPU box with tools/ab_run.sh
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"; [ "$flags" = "$spec" ] && flags=""
  echo "== $name  [$flags]"
  MP_BUILD_TAG="$name" MP_EXTRA_FLAGS="$flags" python -m multiply_amd.build >/dev/null || exit 1
done
ls -la multiply_amd/ab_libs/
