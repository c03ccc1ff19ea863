#!/usr/bin/env python
"""(Lives under tests/ because it times the CPU oracle as the baseline.)  Input producer: items/s of the HBM-resident dataset (multiply_amd/datasets.py, one mp_sample_pixels launch per item)
vs the reference's per-item path restated on the CPU (oracle/dataset_oracle.py: decode the frame's PNGs, gather with
numpy -- what one DataLoader worker of the reference does, code/lib/datasets/Hi4D.py:229-306).
    python tests/bench_producer.py [frames=12] [H=940] [W=1280]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd.config import to_config            # noqa: E402
from multiply_amd.datasets import Hi4DDataset         # noqa: E402
from multiply_amd.synthetic import write_sequence     # noqa: E402
from oracle.dataset_oracle import Hi4DDatasetOracle   # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 12
H = int(sys.argv[2]) if len(sys.argv) > 2 else 940
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
root = os.path.join(tempfile.mkdtemp(), "seq")
t0 = time.perf_counter()
write_sequence(root, n_frames=F, H=H, W=W)
print(f"wrote {F} frames {H}x{W}, 2 persons ({time.perf_counter() - t0:.1f} s)")
opt = to_config(dict(data_root=os.path.dirname(root), data_dir="seq", start_frame=0, end_frame=F, num_sample=512,
                     using_SAM=False))
t0 = time.perf_counter()
ds = Hi4DDataset(opt, rng=np.random.RandomState(0))
torch.cuda.synchronize()
t_load = time.perf_counter() - t0
ora = Hi4DDatasetOracle(root, 0, F, 512)
rs = np.random.RandomState(0)
n_cpu = min(F, 8)
t0 = time.perf_counter()
for i in range(n_cpu):
    want = ora.__getitem__(i % F, rng=rs)
t_cpu = (time.perf_counter() - t0) / n_cpu
for i in range(20):
    ds[i % F]
torch.cuda.synchronize()
n_gpu = 2000
t0 = time.perf_counter()
for i in range(n_gpu):
    got = ds[i % F]
torch.cuda.synchronize()
t_gpu = (time.perf_counter() - t0) / n_gpu
print(f"one-time decode + upload: {t_load:.2f} s ({F} frames, {ds.store.images.numel() / 1e6:.0f} MB of pixels in HBM)")
print(f"per item (512 samples): CPU path {t_cpu * 1e3:.1f} ms = {1 / t_cpu:.1f} items/s per worker process; "
      f"resident path {t_gpu * 1e6:.0f} us = {1 / t_gpu:.0f} items/s  (x{t_cpu / t_gpu:.0f})")
