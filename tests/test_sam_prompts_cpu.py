"""The prompt construction around Segment-Anything (multiply_amd/sam_prompts.py) against the REFERENCE's own
SAMServer.get_sam_mask run on the same written sequences (tests/golden/sam_golden.npz, made by make_sam_golden.py with
segment_anything / hydra / cv2 stubbed: the fixture holds every prompt the reference handed to the predictor, in order, and
the mask file it wrote), and the refresh loop end to end on a written sequence with a stand-in predictor (SAM itself is a
third-party model that is not part of this repository)."""
import os

import numpy as np
import pytest

from multiply_amd import sam_prompts as SP
from tests.sam_standin import StandInPredictor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sam_golden.npz")


@pytest.mark.parametrize("case", ["wide", "tall", "square"])
def test_prompts_and_mask_file_match_the_reference_run(case, tmp_path):
    """the whole refresh (random stream seeded once, then per frame and person: fallback positives, the permutation, the ten
    random negatives, the neighbours' key points; box; padded 256 x 256 mask logits; three rounds) reproduces what the
    reference's sam_model.py:58-239 did on these inputs"""
    from PIL import Image
    from multiply_amd.config import to_config
    g = np.load(GOLDEN)
    images, masks, joints = g[f"{case}_images"], g[f"{case}_masks"], g[f"{case}_joints"]
    start, end = (int(v) for v in g[f"{case}_range"])
    root = tmp_path / "data" / "seq"
    (root / "image").mkdir(parents=True)
    for f in range(images.shape[0]):
        Image.fromarray(images[f]).save(str(root / "image" / ("%04d.png" % f)))
    stage = tmp_path / "run"
    (stage / "stage_instance_mask" / "00050").mkdir(parents=True)
    np.save(str(stage / "stage_instance_mask" / "00050" / "all_person_smpl_mask.npy"), masks[start:end])
    np.save(str(stage / "stage_instance_mask" / "00050" / "2d_keypoint.npy"), joints[start:end])
    pred = StandInPredictor()
    server = SP.SAMServer(to_config(dict(data_root=str(tmp_path / "data"), data_dir="seq", start_frame=start, end_frame=end)),
                          predictor=pred)
    out = server.get_sam_mask(50, stage_dir=str(stage))
    P = masks.shape[1]
    assert len(pred.calls) == 3 * (end - start) * P
    for k in range((end - start) * P):
        first = pred.calls[3 * k]
        want = g[f"{case}_coords_{k}"]
        assert first["coords"].shape == want.shape and np.array_equal(first["coords"], want), (case, k)
        assert np.array_equal(first["labels"], g[f"{case}_labels_{k}"]) and np.array_equal(first["box"], g[f"{case}_box_{k}"])
        on = np.unpackbits(g[f"{case}_mask_on_{k}"])[:256 * 256].reshape(256, 256).astype(bool)
        assert np.array_equal(first["mask_input"][0] > 0, on), (case, k)
        assert np.array_equal(np.unique(first["mask_input"]), g[f"{case}_mask_vals_{k}"])           # fp32 logit(eps), logit(1 - eps)
        for r in range(3):                                                                        # the logits fed back round to round
            assert np.array_equal(pred.calls[3 * k + r]["coords"], want)
            assert abs(float(pred.calls[3 * k + r]["mask_input"].astype(np.float64).mean()) - g[f"{case}_mask_mean_rounds_{k}"][r]) < 1e-6
    want = g[f"{case}_written"]
    assert out.shape == want.shape and out.dtype == want.dtype and np.allclose(out, want, atol=1e-5, rtol=0)
    assert np.array_equal(np.load(str(stage / "stage_sam_mask" / "00050" / "sam_opt_mask.npy")), out)


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
