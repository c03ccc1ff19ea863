#!/usr/bin/env python
"""gpurun_out/prof_<tag>/ (written on the GPU box by tools/profile_round.sh) -> the tracked summaries profiles/<tag>_*:
    python tools/write_profiles.py r02"""
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles")

# ---- kernel trace
kt = open(os.path.join(src, "kernel_trace.txt")).read()
bench = json.loads([l for l in open(os.path.join(src, "bench_under_trace.json")) if l.startswith("{")][0])
with open(os.path.join(dst, tag + "_kernel_trace.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace -- python bench.py --no-cpu-baseline   (MI355X, round {tag[1:]}; tools/profile_round.sh + "
            f"tools/rocpd_summary.py)\n# 2 warm-up + 5 timed frames of the 512x512 2-person scene (N_samples 128), then 2 + 10 training "
            f"iterations of 512 rays.\n# the same run printed: {bench['value'] / 1e6:.3f} M rays/s, {bench['ms_per_step']:.1f} ms/frame, "
            f"roofline.frac {bench['roofline']['frac']:.3f} ({bench['roofline']['kernel']}, avg launch "
            f"{bench['roofline']['avg_launch_ms']:.2f} ms by HIP events), {bench['train_iter']['ms_per_iter']:.1f} ms/train-iter\n"
            f"# mp_mlp_shade_rev = k_mlp_fwdsave + k_mlp_grad, dispatched once per 2M-point segment of the worklist (segments past the "
            f"device-side count return at once):\n#   per call = (total fwdsave + total grad) / 14 calls (7 frames x 2 persons)\n")
    f.write(kt)

# ---- traffic
raw = open(os.path.join(src, "pmc_traffic_raw.txt")).read().strip().split("\n")
pmc = json.loads(raw[-1])
# what the measurement is a measurement OF (bench.py attaches the file only to a run of the same kernels on the same workload)
sys.path.insert(0, root)
import bench as _bench   # noqa: E402
one = [l for l in open(os.path.join(src, "bench_one_frame.json")) if l.startswith("{")]
one = json.loads(one[0]) if one else bench
pmc["_meta"] = {"kernel_source_sha16": _bench.kernel_source_hash(), "dominant": one["roofline"]["kernel"].split(" = ")[0],
                "algorithmic_flop_per_launch": one["roofline"]["algorithmic_flop_per_launch"],
                "launches_per_step": one["roofline"]["launches_per_step"], "command": "bench.py --steps 1 --warmup 0 --train-steps 0 --no-cpu-baseline"}
json.dump(pmc, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"))
with open(os.path.join(dst, tag + "_pmc_traffic.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, each with --kernel-trace only) of\n"
            "#   python bench.py --steps 1 --warmup 0 --train-steps 0 --no-cpu-baseline      (MI355X: ONE frame)\n"
            "# read = 2 x 1024 x FETCH_SIZE (gfx950 half-count correction, MI355X_MICROARCH.md HBM section); write = 1024 x WRITE_SIZE "
            "(uncalibrated)\n# per-dispatch averages; k_mlp_fwdsave / k_mlp_grad run once per 2M-point worklist segment (n includes the "
            "empty ones).\n# Infinity-Cache hits are counted by these counters: every figure is an UPPER bound on HBM bytes.\n")
    f.write("\n".join(raw[:-1]) + "\n")

# ---- SQ counters with derived figures
txt = open(os.path.join(src, "pmc_sq.txt")).read()
blocks = re.split(r"^== ", txt, flags=re.M)[1:]
with open(os.path.join(dst, tag + "_pmc_sq.txt"), "w") as f:
    f.write("# rocprofv3 --pmc <SQ counters> (two passes of 8 counters, --kernel-trace only) of ONE frame:\n"
            "#   python bench.py --steps 1 --warmup 0 --train-steps 0 --no-cpu-baseline        (MI355X; tools/profile_round.sh)\n"
            "# per-dispatch averages (fwdsave / grad: over all segment dispatches incl. the empty ones -- ratios are unaffected).\n"
            "# derived:  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   [matrix-pipe utilisation; GRBM_GUI_ACTIVE is\n"
            "#           summed over the 8 XCDs: /8 x clock = the dispatch time of the kernel trace; BUSY = 16 cycles x MFMA count]\n"
            "#           clock = GRBM_GUI_ACTIVE / 8 / kernel-trace duration: 1.98 GHz for k_mlp_fwdsave (peak figures assume 2.4)\n"
            "#           VALU issue share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES,  parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES,\n"
            "#           issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES,  VALU instr per MFMA instr, LDS conflict share\n"
            "#           (SQ_INSTS_VALU_MFMA_MOPS_F16 counts 512-FLOP units: 32 per v_mfma_f32_16x16x32_f16)\n\n")
    f.write(f"{'kernel':22s} {'MFMA busy':>9s} {'VALU issue':>10s} {'parked':>7s} {'stalled':>8s} {'VALU/MFMA':>9s} {'LDS confl':>9s}\n")
    rows = []
    for b in blocks:
        name, *lines = b.strip().split("\n")
        c = {}
        for l in lines:
            m = re.match(r"(\w+)\s+(\d+)", l.strip())
            if m:
                c[m.group(1)] = float(m.group(2))
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        mfma = c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0) / 32.0
        rows.append((name, c))
        f.write(f"{name:22s} {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):8.1f}% "
                f"{100 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:9.1f}% {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:6.1f}% "
                f"{100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:7.1f}% "
                f"{(c['SQ_INSTS_VALU'] - mfma) / mfma if mfma else float('nan'):9.2f} "
                f"{100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):8.1f}%\n")
    f.write("\n# raw per-dispatch averages\n")
    f.write(txt)
print(open(os.path.join(dst, tag + "_pmc_sq.txt")).read()[:2400])
