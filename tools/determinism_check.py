#!/usr/bin/env python
"""GPU box: is the rendered frame a function of the library alone?  `save` renders bench.py's headline frame twice (compares the two
renders bit for bit) and stores the outputs; `cmp` renders it again -- in another process, possibly with another library through
MP_LIB_PATH (another cluster layout, another point-to-wave mapping: tools/ab_build.sh) -- and counts the elements that differ.
    python tools/determinism_check.py save;  MP_LIB_PATH=... python tools/determinism_check.py cmp        (profiles/r06_determinism.txt)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
sys.argv = sys.argv[:1]
import bench
model, inp, tables, sc = bench.build_model(128)
model.convergence_group = 512
with torch.no_grad():
    out = model(bench.to_dev(inp))
    torch.cuda.synchronize()
    out2 = model(bench.to_dev(inp))
    torch.cuda.synchronize()
keys = ("rgb_values", "acc_map", "acc_person_list", "normal_values", "fg_rgb_values")
cur = {k: out[k].detach().cpu() for k in keys}
print("same process, second render:", {k: float((cur[k] - out2[k].detach().cpu()).abs().nan_to_num().max()) for k in keys})
if mode == "save":
    torch.save(cur, "/tmp/det_ref.pt")
else:
    ref = torch.load("/tmp/det_ref.pt")
    for k in keys:
        d = (cur[k].double() - ref[k].double()).abs().nan_to_num()
        print(f"{k:18s} max {float(d.max()):.3e} mean {float(d.mean()):.3e} differing elements {int((d > 0).sum())} of {d.numel()}")
