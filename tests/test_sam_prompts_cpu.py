"""The prompt construction around Segment-Anything (multiply_amd/sam_prompts.py) against its line-by-line restatement
(oracle/sam_oracle.py) under the same random stream, and the refresh loop end to end on a written sequence with a stand-in
predictor (SAM itself is a third-party model that is not part of this repository)."""
import os

import numpy as np
import pytest

from multiply_amd import sam_prompts as SP
from oracle import sam_oracle as SO


def scene(rs, H=60, W=88, P=3):
    yy, xx = np.mgrid[:H, :W]
    masks = np.zeros((P, H, W), dtype=bool)
    joints = np.zeros((P, 29, 2), dtype=np.int32)
    for p in range(P):
        cx, cy = W * (0.25 + 0.25 * p), H * 0.5
        masks[p] = (((xx - cx) / (0.14 * W)) ** 2 + ((yy - cy) / (0.4 * H)) ** 2) < 1
        joints[p, :, 0] = np.clip(rs.normal(cx, 0.12 * W, 29), -5, W + 4).astype(np.int32)     # some off the mask, some off the image
        joints[p, :, 1] = np.clip(rs.normal(cy, 0.3 * H, 29), -5, H + 4).astype(np.int32)
    return masks, joints


def test_prompts_match_the_restatement_under_the_same_random_stream():
    rs = np.random.RandomState(0)
    for trial in range(4):
        masks, joints = scene(rs, P=2 + trial % 2)
        if trial == 3:
            joints[0, :27] = [[0, 0]] * 27                      # no key point on the mask: random fallback pixel
        a, b = np.random.RandomState(42), np.random.RandomState(42)
        for person in range(masks.shape[0]):
            coords, labels = SP.point_prompts(masks, joints, person, a)
            want = SO.person_prompts(masks, joints, person, b)
            assert coords.shape == want[0].shape and np.array_equal(coords, want[0]), (trial, person)
            assert np.array_equal(labels, want[1])
            assert np.array_equal(SP.box_from_mask(masks[person]), want[2])
            got_mask = SP.mask_prompt(masks[person])
            assert got_mask.shape == (1, 256, 256) and np.array_equal(got_mask, want[3])
            assert labels[:int(labels.sum())].all() and (labels[int(labels.sum()):] == 0).all() and (labels == 0).sum() >= 10
        assert a.randint(0, 1 << 30) == b.randint(0, 1 << 30)   # both consumed the stream identically


def test_mask_prompt_keeps_the_shape_of_the_mask():
    m = np.zeros((40, 100), dtype=bool)
    m[10:30, 20:60] = True
    lg = SP.mask_prompt(m)[0]
    on = lg > 0
    # the 40 x 100 image sits at the top of a 100 x 100 canvas: rows 10..30 -> 25.6..76.8, columns 20..60 -> 51.2..153.6
    assert on[:25].sum() == 0 and on[78:].sum() == 0 and on[30:70, 60:140].all()
    assert abs(float(on.mean()) - (20 * 40) / (100 * 100)) < 0.004
    assert np.allclose(np.unique(np.round(lg, 3)), [-13.816, 13.802], atol=2e-3)   # fp32 logit of eps and of 1 - eps (eps = 1e-6)
    tall = np.zeros((100, 40), dtype=bool)
    tall[:, 30:] = True
    assert (SP.mask_prompt(tall)[0][:, :70] < 0).all()          # taller than wide: columns stay left-aligned


class FakePredictor:
    """stands in for segment_anything.SamPredictor: a smooth function of its prompts, so that every input matters"""

    def __init__(self):
        self.calls, self.image = [], None

    def set_image(self, image):
        assert image.dtype == np.uint8 and image.shape[2] == 3
        self.image = image

    def predict(self, point_coords, point_labels, mask_input, box, multimask_output, return_logits):
        assert not multimask_output and return_logits and mask_input.shape == (1, 256, 256) and box.shape == (1, 4)
        self.calls.append((point_coords.copy(), point_labels.copy(), float(mask_input.mean()), box.copy()))
        H, W = self.image.shape[:2]
        yy, xx = np.mgrid[:H, :W]
        x0, y0, x1, y1 = box[0]
        logits = np.where((xx >= x0) & (xx <= x1) & (yy >= y0) & (yy <= y1), 4.0, -4.0) + 0.01 * point_labels.sum()
        return logits[None] > 0, np.ones(1), np.clip(mask_input * 0.5 + 0.1, -20, 20)


def test_refresh_loop_on_a_written_sequence(tmp_path):
    from multiply_amd.config import to_config
    from multiply_amd.synthetic import write_sequence
    root = str(tmp_path / "seq")
    w = write_sequence(root, n_frames=3, H=48, W=64)
    rs = np.random.RandomState(1)
    masks = np.stack([w["masks"][f] for f in range(1, 3)])                        # frames 1, 2 (start_frame = 1)
    joints = np.stack([np.stack([np.stack([rs.randint(0, 64, 29), rs.randint(0, 48, 29)], 1) for _ in range(2)]) for _ in range(2)])
    stage = tmp_path / "run"
    (stage / "stage_instance_mask" / "00050").mkdir(parents=True)
    np.save(str(stage / "stage_instance_mask" / "00050" / "all_person_smpl_mask.npy"), masks)
    np.save(str(stage / "stage_instance_mask" / "00050" / "2d_keypoint.npy"), joints.astype(np.int32))
    pred = FakePredictor()
    server = SP.SAMServer(to_config(dict(data_root=os.path.dirname(root), data_dir=os.path.basename(root), start_frame=1,
                                         end_frame=3)), predictor=pred)
    out = server.get_sam_mask(50, stage_dir=str(stage))
    assert out.shape == (2, 2, 48, 64) and len(pred.calls) == 2 * 2 * 3           # frames x persons x three rounds
    saved = np.load(str(stage / "stage_sam_mask" / "00050" / "sam_opt_mask.npy"))
    assert np.array_equal(saved, out)
    # the three rounds of one person share the prompts and feed the logits back (mean of the mask input changes)
    c0, c1, c2 = pred.calls[:3]
    assert np.array_equal(c0[0], c1[0]) and np.array_equal(c1[0], c2[0]) and c0[2] != c1[2] != c2[2]
    # the dataset reads what was written: (F, P, H, W) logits -> per-frame (H, W, P)
    with pytest.raises(RuntimeError):
        SP.SAMServer(server.opt)                                                  # no segment_anything here, no predictor given
