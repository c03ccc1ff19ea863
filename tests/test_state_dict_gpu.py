"""Checkpoint contract: the module tree of multiply_amd.Multiply has EXACTLY the reference's state-dict keys, shapes and
dtypes -- including the SMPL tables under smpl_server_list.N.smpl.* / deformer_list.N.smpl.smpl.* -- and the same parameter
list (tests/golden/reference_state_keys.npz, written by tests/golden/make_train_golden.py from the reference's own module
tree), so a reference checkpoint loads with strict=True and a checkpoint saved here loads in the reference."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_state_dict_keys_shapes_dtypes_and_strict_round_trip():
    from tests.test_render_gpu import build
    ref = np.load(os.path.join(HERE, "golden", "reference_state_keys.npz"), allow_pickle=False)
    model, _, inp = build(H=8, W=8)
    sd = model.state_dict()
    keys = [str(k) for k in ref["keys"]]
    assert sorted(sd.keys()) == keys
    for k, shp, dt in zip(keys, ref["shapes"], ref["dtypes"]):
        want = tuple(int(x) for x in str(shp).split(",") if x)
        assert tuple(sd[k].shape) == want, (k, tuple(sd[k].shape), want)
        assert str(sd[k].dtype) == str(dt), (k, sd[k].dtype, dt)
    assert sorted(k for k, _ in model.named_parameters()) == [str(k) for k in ref["param_keys"]]
    # a "reference checkpoint": same keys, other values for the networks
    g = torch.Generator().manual_seed(3)
    ckpt = {}
    for k, v in sd.items():
        if v.is_floating_point() and "smpl" not in k:
            ckpt[k] = (v.cpu() + 0.01 * torch.randn(v.shape, generator=g)).to(v.dtype)
        else:
            ckpt[k] = v.cpu().clone()
    before = model({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()})["rgb_values"].clone()
    missing, unexpected = model.load_state_dict(ckpt, strict=True)
    assert not missing and not unexpected
    after = model({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()})["rgb_values"]
    torch.cuda.synchronize()
    assert not torch.equal(torch.nan_to_num(before), torch.nan_to_num(after))      # the loaded weights are the ones used
    back = model.state_dict()
    for k in keys:
        assert torch.equal(back[k].cpu(), ckpt[k]), k
    # SMPLServer.forward's output dict (smpl.py:50-94)
    sp = inp["smpl_params"].cuda()
    out = model.smpl_server_list[0](sp[:, 0, 0], sp[:, 0, 1:4], sp[:, 0, 4:76], sp[:, 0, 76:])
    assert sorted(out) == ["smpl_all_jnts", "smpl_jnts", "smpl_tfs", "smpl_verts", "smpl_weights"]
    assert out["smpl_all_jnts"].shape == (1, 29, 3) and out["smpl_jnts"].shape == (1, 24, 3)
    assert torch.equal(out["smpl_all_jnts"][0, :24], out["smpl_jnts"][0])
    assert torch.equal(out["smpl_all_jnts"][0, 24:], out["smpl_verts"][0, [332, 6260, 2800, 4071, 583]])
