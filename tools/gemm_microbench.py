#!/usr/bin/env python
"""Times the fp32 training GEMMs at the shapes of one training iteration:  python tools/gemm_microbench.py [rows]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import train as T

M = int(sys.argv[1]) if len(sys.argv) > 1 else 306000
N = K = 256
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
dW = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timeit(lambda: T.gemm_nt(T._p(A), K, T._p(B), K, T._p(C), N, M, N, K))
print(f"gemm_nt {M}x{N}x{K}: {t * 1e3:.3f} ms  {2.0 * M * N * K / t / 1e12:.1f} TFLOP/s")
t = timeit(lambda: T.gemm_tn(T._p(C), N, T._p(A), K, T._p(dW), K, N, K, M, T._p(db), M))
print(f"gemm_tn {N}x{K}x{M}: {t * 1e3:.3f} ms  {2.0 * M * N * K / t / 1e12:.1f} TFLOP/s")
t = timeit(lambda: torch.mm(A, B.t(), out=C))
print(f"torch.mm (hipBLASLt fp32) same shape: {t * 1e3:.3f} ms  {2.0 * M * N * K / t / 1e12:.1f} TFLOP/s")
