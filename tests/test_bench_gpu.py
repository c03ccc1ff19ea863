"""bench.py's contract (the driver depends on it): one JSON line with the headline metric, `roofline` and `cpu_baseline`
at N = 1; `--gpus N` starts N ranks itself (or asserts the launcher's WORLD_SIZE) and reports n_gpus = N with the
strong-scaling leg of one frame.  Small frames so that the whole file runs in about a minute."""
import json
import os
import subprocess
import sys

import pytest

from tests import tolerances as TOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_roofline_and_cpu_baseline():
    # --cpu-budget-s 0: the wall-time guard of the CPU oracle's render leg at its tightest (a slower host must not turn the headline run
    # into a driver timeout): the sample stops after the minimum number of rays -- here the whole small sample is below the 4 096-ray
    # floor, so it completes and says how many rays it did (round 6: checked in THIS run instead of a second 70 s bench run)
    d = _run(["--res", "64", "--steps", "2", "--warmup", "1", "--train-steps", "2", "--train-warmup", "1", "--cpu-rays", "128",
              "--cpu-train-iters", "1", "--cpu-train-rays", "128", "--cpu-budget-s", "0"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "rays/s" and d["value"] > 0 and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c and c["parity_rgb_max_abs"] < 5e-4
    assert c["rays"] == 128 and c["seconds"] > 0 and "128 rays" in c["sample"]
    assert d["config"]["sampler_sdf"] == "bf16x3" and len(d["lib_source_sha16"]) == 16
    t = d["train_iter"]
    assert t["ms_per_iter"] > 0 and t["rays_per_iter"] == 512 and t["cpu_baseline"]["value"] > 0 and 0 < t["roofline"]["frac"] < 1
    assert t["dtype"] == "bf16x3" and "dtype_note" in d and t["cpu_baseline"]["parity_grad_rel_worst"] < 2e-2
    assert t["cpu_baseline"]["parity_loss_abs"] < 1e-4 and t["cpu_baseline"]["rays"] == 128
    # round 6: the sampler of that iteration against the oracle's own sampler (same draws), and the bce-guard flags
    tc = t["cpu_baseline"]
    assert tc["parity_sampler_depth_mean_abs"] < 1.3e-4 and tc["parity_sampler_depth_rays_above_3e-3"] <= max(TOL.TRAIN_DEPTH_RAYS["floor"], TOL.TRAIN_DEPTH_RAYS["frac"] * tc["parity_sampler_depth_rays"])
    assert tc["bce_guard"]["mismatch"] in (False, True) and d["train_iter"]["stash_bytes_per_point"] > 46 * 1024


def test_gpus_flag_starts_the_ranks_and_reports_strong_scaling():
    """`python bench.py --gpus 2` with no launcher: two ranks (sharing this box's GPU over gloo), n_gpus = 2, the frame's
    convergence groups split between them, the weak-scaling leg beside it"""
    d = _run(["--gpus", "2", "--res", "64", "--steps", "2", "--warmup", "1", "--train-steps", "2", "--train-warmup", "1"],
             env={"MP_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and "cpu_baseline" not in d
    assert d["collective_backend"] == "gloo" and d["rccl_ranks_seen"] == 2      # an all_reduce of ones over the data-path backend
    assert d["train_iter"]["host_ms_per_iter"] > 0
    assert d["config"]["rays_per_step"] == 64 * 64 and "ray-sharded dp2" in d["config"]["parallelism"]
    assert d["weak"]["scaling"] == "weak" and d["weak"]["value"] > 0
    assert d["train_iter"]["rays_per_iter_per_gpu"] == 256 and d["train_iter"]["rays_per_iter"] == 512
    pr = d["per_rank"]       # the diagnosis of a scaling run: per-rank frame time, ray share, time inside the image all_gather
    assert len(pr["ms_per_step"]) == 2 and sum(pr["rays"]) == 64 * 64 and all(t >= 0 for t in pr["all_gather_ms_per_step"])


@pytest.mark.parametrize("mode,ranks,persons,slots", [("person", 2, 2, 0), ("hybrid", 4, 2, 2)])
def test_person_and_hybrid_modes_render_the_single_process_frame(mode, ranks, persons, slots):
    """BASELINE.json configs[3]'s measurement path: `--mode person` (persons sharded, one all_to_all per person slot and chunk) and
    `--mode hybrid` (person teams x ray shards) render ONE frame per step over the ranks (sharing this box's GPU over gloo); the
    line carries the per-rank frame time, the exchanges' own time and the image all_gather's"""
    d = _run(["--gpus", str(ranks), "--mode", mode, "--persons", str(persons), "--res", "64", "--steps", "2", "--warmup", "1",
              "--train-steps", "0", "--chunk-rays", "2048"] + (["--person-slots", str(slots)] if slots else []),
             env={"MP_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == ranks and d["value"] > 0 and d["rccl_ranks_seen"] == ranks and d["config"]["persons"] == persons
    assert mode in d["config"]["parallelism"] and "weak" not in d
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == ranks and all(x > 0 for x in pr["exchanges_per_step"]) and all(t >= 0 for t in pr["exchange_ms_per_step"])
    assert 0 < d["roofline"]["frac"] < 1


def test_single_gpu_line_of_the_four_person_256_sample_workload():
    """BASELINE.json configs[3] on one GPU: `--persons 4 --samples 256` (small frame here), same fields as the headline line"""
    d = _run(["--persons", "4", "--samples", "256", "--res", "48", "--steps", "1", "--warmup", "1", "--train-steps", "0", "--cpu-rays", "96"])
    assert d["config"]["persons"] == 4 and "4-person" in d["config"]["workload"] and "N_samples=256" in d["config"]["workload"]
    assert 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["parity_rgb_max_abs"] < 5e-4


def test_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """the SAME command on the backend the driver's multi-GPU runs use ("nccl" = RCCL over xGMI): one rank per GPU, the image
    all_gather, the sampler's convergence vote and the bucketed gradient all-reduce as real device collectives.  A one-GPU box cannot run it -- said loudly, not
    silently replaced by the gloo variant above."""
    import torch
    if torch.cuda.device_count() < 2:
        msg = (f"RCCL PATH NOT EXECUTED: this box exposes {torch.cuda.device_count()} GPU(s); bench.py --gpus 2 on backend 'nccl' "
               f"needs one GPU per rank (the gloo variant of the same code path ran in the test above)")
        print("\n[WARNING] " + msg, file=sys.stderr)
        pytest.skip(msg)
    d = _run(["--gpus", "2", "--res", "64", "--steps", "2", "--warmup", "1", "--train-steps", "2", "--train-warmup", "1"])
    assert d["n_gpus"] == 2 and d["collective_backend"] == "nccl" and d["value"] > 0
    # the gradient buckets are reduced while the backward runs: the "allreduce" phase is only what is left after it returned
    assert d["train_iter"]["rays_per_iter_per_gpu"] == 256 and d["train_iter"]["gpu_ms"]["allreduce"] >= 0


def test_world_size_mismatch_is_refused():
    e = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--res", "64"], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)
