"""A stand-in for segment_anything.SamPredictor (a third-party ViT and checkpoint, absent here): a deterministic function of
EVERY prompt it is given, which records each call.  Used twice: tests/golden/make_sam_golden.py runs the reference's
SAMServer.get_sam_mask around it (recording the prompts the reference builds), and tests/test_sam_prompts_cpu.py runs
multiply_amd.sam_prompts around it and compares call by call."""
import numpy as np


class StandInPredictor:
    def __init__(self, *_, **__):
        self.calls, self.image = [], None

    def set_image(self, image):
        assert image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3
        self.image = image

    def predict(self, point_coords, point_labels, mask_input, box, multimask_output, return_logits):
        assert not multimask_output and return_logits and mask_input.shape == (1, 256, 256) and box.shape == (1, 4)
        self.calls.append(dict(coords=np.array(point_coords), labels=np.array(point_labels), box=np.array(box),
                               mask_input=np.array(mask_input, dtype=np.float32)))
        H, W = self.image.shape[:2]
        yy, xx = np.mgrid[:H, :W].astype(np.float32)
        x0, y0, x1, y1 = (float(v) for v in box[0])
        logits = np.where((xx >= x0) & (xx <= x1) & (yy >= y0) & (yy <= y1), 4.0, -4.0).astype(np.float32)
        sign = 2.0 * np.asarray(point_labels, dtype=np.float32) - 1.0
        for (px, py), s in zip(np.asarray(point_coords, dtype=np.float32), sign):      # a bump per point prompt
            logits += s * np.exp(-((xx - px) ** 2 + (yy - py) ** 2) / 18.0).astype(np.float32)
        logits += np.float32(0.25) * np.float32(np.tanh(np.asarray(mask_input, dtype=np.float32).mean()))
        logits += (self.image.astype(np.float32).mean(axis=2) / 255.0 - 0.5) * np.float32(0.125)
        low = np.clip(np.asarray(mask_input, dtype=np.float32) * np.float32(0.5) + np.float32(0.0625 * len(point_labels)),
                      -20, 20).astype(np.float32)
        return logits[None], np.ones(1, dtype=np.float32), low
