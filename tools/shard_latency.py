#!/usr/bin/env python
"""One GPU: how long does ONE rank of an N-rank strong-scaling run need for its share of the frame (round-robin convergence
groups, bench.py's sharding), without the all_gather?  t(1) / (N t(N)) bounds the scaling efficiency from the compute side:
it exposes per-call fixed costs and small-launch inefficiency before an 8-GPU node is available.
    python tools/shard_latency.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import bench                                                   # noqa: E402
from multiply_amd.parallel import shard_input_interleaved      # noqa: E402

steps = int(args[0]) if args else 5
model, inp, tables, sc = bench.build_model(128, seed=0, H=512, W=512, tile=8)
model.convergence_group = 512
model.async_setup = os.environ.get('MP_ASYNC_SETUP', '1') == '1'
base = None
for world in (1, 2, 4, 8):
    worst = 0.0
    per_rank = []
    for rank in range(world):
        share, ids = shard_input_interleaved(inp, rank, world, 512, 8) if world > 1 else (inp, None)
        gin = bench.to_dev(share)
        with torch.no_grad():
            model(gin); model(gin)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                model(gin)
            torch.cuda.synchronize()
        per_rank.append((time.perf_counter() - t0) / steps)
        worst = max(worst, per_rank[-1])
    base = base or worst
    print(f"world {world}: {gin['uv'].shape[1]:7d} rays/rank  {1e3 * worst:7.2f} ms/frame/rank  "
          f"compute-side efficiency {base / (world * worst):.3f}  -> {262144 / worst / 1e6:.2f} M rays/s   per rank ms: "
          + " ".join(f"{1e3 * t:.1f}" for t in per_rank))

# where does a 1/8 share spend its time?  (phase events of Multiply, one mid-image rank)
share, ids = shard_input_interleaved(inp, 3, 8, 512, 8)
gin = bench.to_dev(share)
model.profile = True
with torch.no_grad():
    model(gin)
    torch.cuda.synchronize()
    model.phase_events = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        model(gin)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
ph = model.phase_times_ms()
tot = sum(v[1] for v in ph.values()) / steps
print(f"1/8 share: wall {1e3 * wall:.2f} ms/frame, phases sum {tot:.2f} ms: " + ", ".join(f"{k} {v[1] / steps:.2f}" for k, v in ph.items()))
