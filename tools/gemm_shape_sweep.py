#!/usr/bin/env python
"""How the split-bfloat16 `nt` product of the training path scales with the contraction length at the row count of one person's
samples (63 k x 256 outputs): a time that does not grow with K is a streaming (prologue / epilogue / HBM) bound, one that grows
with K an MFMA / LDS bound.  Beside it: hipBLASLt's fp32 and bf16 products of the same shapes and a plain copy of the bytes.
  python tools/gemm_shape_sweep.py [rows]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import train as T

M = int(sys.argv[1]) if len(sys.argv) > 1 else 63000
N = 256


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print(f"rows {M}, {N} output columns; microseconds per product")
print(f"{'K':>6} {'nt bf16x3':>10} {'nt fp32':>9} {'mm fp32':>9} {'mm bf16':>9} {'copy in+out':>12} {'MB moved':>9}")
for K in (64, 128, 256, 512, 1024):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    Ab, Bb, Cb = A.bfloat16(), B.bfloat16(), C.bfloat16()
    T.TRAIN_PRECISION = "bf16x3"
    t3 = timeit(lambda: T.gemm_nt(T._p(A), K, T._p(B), K, T._p(C), N, M, N, K))
    T.TRAIN_PRECISION = "f32"
    t1 = timeit(lambda: T.gemm_nt(T._p(A), K, T._p(B), K, T._p(C), N, M, N, K))
    tm = timeit(lambda: torch.mm(A, B.t(), out=C))
    tb = timeit(lambda: torch.mm(Ab, Bb.t(), out=Cb))
    src = torch.empty(M * (K + N), device="cuda"); dst = torch.empty_like(src)
    tc = timeit(lambda: dst[:M * K].copy_(src[:M * K])) * 0 + timeit(lambda: dst.copy_(src)) / 2      # half: read K + write N columns ~ (K + N) floats moved once
    print(f"{K:6d} {t3:10.1f} {t1:9.1f} {tm:9.1f} {tb:9.1f} {tc:12.1f} {M * (K + N) * 4 / 1e6:9.1f}")
