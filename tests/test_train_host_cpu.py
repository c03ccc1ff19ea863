"""Host-side pieces of the training path that need no GPU: the batched permutation prefixes (what the reference draws with
torch.randperm(n)[:k], ray_sampler.py:202 / sampler.py:100), the integer tables' CPU route, the coarse stash sizes."""
import numpy as np
import torch

from multiply_amd import hip
from multiply_amd import train as T


def test_random_prefixes_are_prefixes_of_uniform_permutations():
    g = torch.Generator().manual_seed(3)
    sizes, k = [128, 256, 384, 512, 640] * 2, 32
    idx = T._random_prefixes(sizes, k, torch.device("cpu"), g)
    assert idx.shape == (len(sizes), k) and idx.dtype == torch.int64
    for row, n in zip(idx, sizes):
        assert int(row.min()) >= 0 and int(row.max()) < n and len(set(row.tolist())) == k          # distinct, in range
    # uniformity: over many draws every index of a row is selected with probability k / n, and the FIRST entry is uniform over n
    n, k, reps = 40, 8, 4000
    draws = torch.stack([T._random_prefixes([n, 64], k, torch.device("cpu"), g)[0] for _ in range(reps)])
    counts = torch.bincount(draws.reshape(-1), minlength=n).double()
    expect = reps * k / n
    assert float(((counts - expect) ** 2 / expect).sum()) < 85.0           # chi-square, 39 degrees of freedom: p(> 85) ~ 3e-5
    first = torch.bincount(draws[:, 0], minlength=n).double()
    assert float(((first - reps / n) ** 2 / (reps / n)).sum()) < 85.0
    # equal sizes take the unmasked route
    same = T._random_prefixes([100, 100, 100], 100, torch.device("cpu"), g)
    assert all(sorted(r.tolist()) == list(range(100)) for r in same)


def test_device_ints_on_the_cpu_is_a_plain_tensor():
    t = hip.device_ints([3, 1 << 40, 7], "cpu")
    assert t.dtype == torch.int64 and t.tolist() == [3, 1 << 40, 7]


def test_stash_sizes_are_coarse_and_capped():
    dev = torch.device("cpu")
    a = T._big_empty(1000, dev, grain=4096)
    assert a.numel() == 4096
    assert T._big_empty(4097, dev, grain=4096).numel() == 8192
    # an affordable upper bound is what gets allocated -- the same size whatever the iteration's point count
    assert T._big_empty(1000, dev, grain=4096, cap=10000, cap_bytes=1 << 20).numel() == 12288
    assert T._big_empty(9000, dev, grain=4096, cap=10000, cap_bytes=1 << 20).numel() == 12288
    # a bound that is too large (or smaller than the request) is ignored
    assert T._big_empty(1000, dev, grain=4096, cap=10000, cap_bytes=1 << 10).numel() == 4096
    assert T._big_empty(20000, dev, grain=4096, cap=10000, cap_bytes=1 << 20).numel() == 20480


def test_packed_cache_generation_changes_on_mode_switch():
    """Multiply.train() / eval() drop the packed-weight caches (optimizers that do not bump tensor versions): the generation that
    is part of every cache key moves on every switch, and only then"""
    from multiply_amd.multiply import Multiply
    m = Multiply.__new__(Multiply)      # the mode switch needs nothing of the scene model
    torch.nn.Module.__init__(m)
    g0 = hip._GENERATION[0]
    m.train(True)                       # already training: nothing to drop
    assert hip._GENERATION[0] == g0
    m.eval()
    assert hip._GENERATION[0] == g0 + 1 and not m.training
    m.eval()
    assert hip._GENERATION[0] == g0 + 1
    m.train()
    assert hip._GENERATION[0] == g0 + 2 and m.training
