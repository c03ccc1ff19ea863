#!/usr/bin/env python
"""Generates tests/golden/mise_golden.npz with the REFERENCE's own MISE extractor (code/lib/libmise/mise.pyx), built from
its source by oracle/Makefile into oracle/_ref/ (Cython -> C++ -> g++, the toolchain the reference's setup.py uses).
Analytic fields stand in for the network: the extractor only sees (points, values).
Run in the build container only:  make -C oracle && python tests/golden/make_mise_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle", "_ref"))
import mise      # noqa: E402  (the reference's extension module)


def fields():
    def sphere(p):
        return np.linalg.norm(p - np.array([0.03, -0.02, 0.05]), axis=1) - 0.31
    def blobs(p):       # union of two spheres and a thin bar: components, thin features, tangencies near voxel corners
        a = np.linalg.norm(p - np.array([-0.2, 0.0, 0.0]), axis=1) - 0.17
        b = np.linalg.norm(p - np.array([0.22, 0.05, -0.03]), axis=1) - 0.12
        q = np.abs(p - np.array([0.0, 0.0, 0.0])) - np.array([0.3, 0.015, 0.02])
        bar = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)
        return np.minimum(np.minimum(a, b), bar)
    def torus(p):
        q = np.stack([np.linalg.norm(p[:, [0, 2]], axis=1) - 0.25, p[:, 1]], 1)
        return np.linalg.norm(q, axis=1) - 0.07
    def plane(p):       # exact zeros on lattice points: the >= / <= threshold rule matters
        return p[:, 0]
    return {"sphere": sphere, "blobs": blobs, "torus": torus, "plane": plane}


out = {}
for name, f in fields().items():
    for res0, depth in ((8, 2), (4, 3), (6, 1)):
        ex = mise.MISE(res0, depth, 0.0)
        pts, nq = ex.query(), []
        while pts.shape[0]:
            world = ((pts.astype(np.float32) / ex.resolution - 0.5) * 1.1).astype(np.float64)
            ex.update(pts, f(world).astype(np.float32).astype(np.float64))
            nq.append(pts.shape[0])
            pts = ex.query()
        key = f"{name}_{res0}_{depth}"
        out[key + "_dense"] = ex.to_dense()
        out[key + "_nq"] = np.asarray(nq)
        print(key, "queries per pass", nq)
np.savez_compressed(os.path.join(HERE, "mise_golden.npz"), **out)
