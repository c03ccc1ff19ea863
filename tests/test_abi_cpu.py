"""CPU-side checks: the C-ABI library exports every declared symbol, host-side packing logic, config surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from multiply_amd.build import build
    return build(verbose=False)        # hipcc cross-compiles gfx950 without a GPU


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    hdr = open(os.path.join(REPO, "include", "multiply_hip.h")).read()
    names = sorted(set(re.findall(r"\b(mp_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.mp_arch.restype = ctypes.c_char_p
    assert lib.mp_arch() == b"gfx950"


def test_struct_layouts_match_header():
    from multiply_amd import hip
    assert ctypes.sizeof(hip.MpLayer) == 24 and ctypes.sizeof(hip.MpNet) == 8 + 24 * hip.MAX_LAYERS
    assert ctypes.sizeof(hip.MpSamplerCfg) == 32
    assert ctypes.sizeof(hip.MpSamplerState) == 11 * ctypes.sizeof(ctypes.c_void_p)


def test_k_slot_permutation_is_a_bijection():
    from multiply_amd import hip
    f = [hip.reg_slot_feature(s) for s in range(256)]
    assert sorted(f) == list(range(256))
    # lane group g of K step ks owns output rows 4g..4g+3 of blocks 2ks and 2ks+1 (mlp_core.hpp)
    for s in range(256):
        ks, sl = divmod(s, 32); g, e = divmod(sl, 8)
        row = f[s] - 32 * ks
        assert row // 16 == (0 if e < 4 else 1) and (row % 16) // 4 == g


def test_layer_plans_cover_every_weight_exactly_once():
    from multiply_amd import hip
    from tests.util import seeded_networks
    m, _ = seeded_networks(1, 0)
    for net, role, ks_in in [(m.foreground_implicit_network_list[0], "full", 2), (m.bg_implicit_network, "full", 3),
                             (m.foreground_rendering_network_list[0], "color", 2), (m.bg_rendering_network, "color", 3)]:
        from multiply_amd.networks import ImplicitNet
        plans = hip.implicit_plans(net, role) if isinstance(net, ImplicitNet) else hip.rendering_plans(net)
        for p in plans:
            w = p.lin.weight_v if hasattr(p.lin, "weight_v") else p.lin.weight
            out_dim, in_dim = w.shape
            cm = p.colmap(ks_in)
            used = sorted(c for c in cm if c >= 0)
            hoisted = list(range(p.hoist[0], p.hoist[0] + p.hoist[1])) if p.hoist else []
            assert sorted(used + hoisted) == list(range(in_dim)), "every input column is packed or hoisted exactly once"
            rows = sorted(r for r in p.rowmap if r >= 0)
            assert rows == list(range(out_dim))


def test_reverse_sweep_plans_transpose_every_hidden_layer():
    """implicit_grad_plans: layers 7..1 transposed (rows = the layer's inputs, K slots = its outputs, each exactly once),
    then the Fourier-feature part of layer 0; sigmoid / capture wiring in MpLayer.aux."""
    from multiply_amd import hip
    from tests.util import seeded_networks
    m, _ = seeded_networks(1, 0)
    net = m.foreground_implicit_network_list[0]
    plans = hip.implicit_grad_plans(net)
    lins = list(net.layers())
    assert [p.lin for p in plans] == [lins[l] for l in (7, 6, 5, 4, 3, 2, 1, 0)]
    for p, l in zip(plans, (7, 6, 5, 4, 3, 2, 1, 0)):
        w = p.lin.weight_v if hasattr(p.lin, "weight_v") else p.lin.weight
        out_dim, in_dim = w.shape
        assert p.transpose
        cm = p.colmap(0)
        assert sorted(c for c in cm if c >= 0) == list(range(out_dim)), "K slots = the layer's outputs, each once"
        rows = [r for r in p.rowmap if r >= 0]
        if l > 0:
            assert rows == list(range(in_dim)) and p.act == hip.ACT_SIGMUL and (p.aux & 0xff) == l      # sigma'_{l-1}
            assert (p.aux >> 8) == (2 if l == 4 else 0)
        else:
            assert rows == list(range(net.embed_dim)) and p.act == hip.ACT_NONE and (p.aux >> 8) == 1
    # the skip layer's transpose scales its rows like the forward plan scales the matching input columns
    rs = plans[3].row_scale
    assert len(rs) == 256 and np.allclose(rs[:217], 2 ** -0.5) and np.allclose(rs[217:], 2 ** -0.5 * hip.SOFTPLUS_K)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from multiply_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        hip.lib()


def test_no_device_fails_loudly():
    from multiply_amd import hip
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="ROCm device"):
        hip.require_device()


def test_cluster_perm_partitions_all_vertices(smpl_tables):
    """the static cluster order of the exact nearest-vertex search (smpl.knn_cluster_perm): every vertex once; fine clusters of at most
    MP_KNN_CLUSTER vertices, none empty; the PAIRS (2c, 2c + 1) -- the coarse clusters of the training searches (csrc/geom.hip
    k_knn_build) -- are the kd leaves, full except the last; both granularities spatially compact; the constants equal the header's"""
    from multiply_amd import hip
    from multiply_amd.smpl import knn_cluster_perm
    hdr = open(os.path.join(REPO, "include", "multiply_hip.h")).read()
    CL, NC = (int(re.search(r"#define %s (\d+)" % k, hdr).group(1)) for k in ("MP_KNN_CLUSTER", "MP_KNN_NC"))
    assert (CL, NC) == (hip.KNN_CLUSTER, hip.KNN_NC) and hip.KNN_CB_ROWS == NC + NC // 2 and NC % 2 == 0 and 2 * CL <= 64
    v = np.asarray(smpl_tables["v_template"], dtype=np.float32)
    perm = knn_cluster_perm(v)
    assert perm.shape == (NC * CL,) and NC * CL >= 6890
    assert sorted(perm[perm >= 0].tolist()) == list(range(6890))

    def radii(size):
        out, counts = [], []
        for c in range(NC * CL // size):
            ids = perm[c * size:(c + 1) * size]
            ids = ids[ids >= 0]
            counts.append(len(ids))
            out.append(np.linalg.norm(v[ids] - v[ids].mean(0), axis=1).max() if len(ids) else 0.0)
        return np.asarray(out), np.asarray(counts)
    r_fine, n_fine = radii(CL)
    r_pair, n_pair = radii(2 * CL)
    assert n_fine.min() >= 1 and n_fine.max() == CL
    assert (n_pair[:-1] == 2 * CL).all() and 1 <= n_pair[-1] <= 2 * CL          # full kd leaves, the last one partial
    # clusters are spatially compact: mean cluster radius far below the body size, the halves tighter than the leaves
    assert np.mean(r_pair) < 0.15 and np.mean(r_fine) < 0.8 * np.mean(r_pair)


def test_docs_quote_the_current_number_of_entry_points():
    hdr = open(os.path.join(REPO, "include", "multiply_hip.h")).read()
    n = len(set(re.findall(r"\b(mp_[a-z_0-9]+)\s*\(", hdr)))
    for doc in ("README.md", "DESIGN.md"):
        quoted = re.findall(r"(\d+) `extern \"C\"` entry points", open(os.path.join(REPO, doc)).read())
        assert quoted and all(int(q) == n for q in quoted), (doc, quoted, n)
