#!/bin/bash
# GPU box: the round's profile set for profiles/ (run through gpurun; everything lands in gpurun_out/prof_<tag>/):
#   tools/profile_round.sh r02
#  1. rocprofv3 --kernel-trace of the default bench (render frames + training iterations) -> per-kernel table
#  2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of ONE frame            -> HBM-side traffic per kernel
#  3. rocprofv3 --pmc <SQ counters> (two passes) of ONE frame                                   -> MFMA / VALU / LDS activity per kernel
# PMC passes use --kernel-trace only (never combined with the sys / hip / hsa trace domains).
TAG="${1:-r02}"
cd "$(dirname "$0")/.."
OUT="$PWD/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO="$PWD"
cd /tmp
rm -rf /tmp/prof_*
rocprofv3 --kernel-trace -d /tmp/prof_kt -o run -- python "$REPO/bench.py" --no-cpu-baseline > "$OUT/bench_under_trace.json" 2> "$OUT/kt.log"
python "$REPO/tools/rocpd_summary.py" "$(find /tmp/prof_kt -name '*.db' | head -1)" "$OUT/kernel_trace.txt" > /dev/null
ONE="python $REPO/bench.py --steps 1 --warmup 0 --train-steps 0 --no-cpu-baseline"
$ONE > "$OUT/bench_one_frame.json" 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -o run -- $ONE > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o run -- $ONE > /dev/null 2> "$OUT/pmc_write.log"
python "$REPO/tools/pmc_traffic.py" /tmp/prof_fetch /tmp/prof_write > "$OUT/pmc_traffic_raw.txt" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/prof_sq1 -o run -- $ONE > /dev/null 2> "$OUT/pmc_sq1.log"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_sq2 -o run -- $ONE > /dev/null 2> "$OUT/pmc_sq2.log"
{
  for k in k_mlp_fwdsave k_mlp_grad k_mlp_color k_background k_tf_sdf_val k_mlp_sdf k_warp_inverse k_warp_jacobian k_sampler_bound k_sampler_resample k_composite; do
    echo "== $k"
    python "$REPO/tools/pmc_kernel.py" /tmp/prof_sq1 "$k"
    python "$REPO/tools/pmc_kernel.py" /tmp/prof_sq2 "$k"
  done
} > "$OUT/pmc_sq.txt" 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > "$OUT/mfma_counter_names.txt"
ls -la "$OUT"
