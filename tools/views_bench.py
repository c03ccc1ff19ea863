#!/usr/bin/env python
"""All views of one frame (all persons + each person alone), the caller pattern of the reference's validation / test
steps (multiply_model.py:982-989): P + 1 separate forward() calls vs ONE Multiply.render_views() call.
    python tools/views_bench.py [n_samples=128] [res=512]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, to_dev     # noqa: E402

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 128
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
model, inp, _, _ = build_model(n_samples, H=res, W=res)
model.convergence_group = 512
gin = to_dev(inp)
P = model.num_person


def separate():
    return {i: (model(gin) if i == -1 else model(gin, i)) for i in [-1] + list(range(P))}


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


t_sep, a = timeit(separate)
t_one, b = timeit(lambda: model.render_views(gin))
same = all(torch.equal(torch.nan_to_num(a[i][k]), torch.nan_to_num(b[i][k])) for i in a for k in a[i])
R = res * res
print(f"{P + 1} views of a {res}x{res} frame: separate forward() calls {t_sep * 1e3:.1f} ms ({(P + 1) * R / t_sep / 1e6:.2f} M rays/s), "
      f"render_views {t_one * 1e3:.1f} ms ({(P + 1) * R / t_one / 1e6:.2f} M rays/s), x{t_sep / t_one:.2f}; identical: {same}")
