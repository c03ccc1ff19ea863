"""The caller-side chunk helpers of the reference (code/lib/utils/idr_utils.py:3-30), for callers that keep the
reference's loop `for s in split_input(...): res.append(model(s))`; `merge_output` concatenates the chunk results.

MI355X-first callers do not need them: `Multiply.forward` takes the whole frame in one call (set
`model.convergence_group = pixel_per_batch` for the chunked loop's exact sampler semantics) and
`Multiply.render_views` renders the all-person view and every single-person view from one sampling pass."""
import torch


def split_input(model_input, total_pixels, n_pixels=10000):
    """list of shallow copies of `model_input`, one per run of n_pixels consecutive pixels of 'uv' (1, R, 2)"""
    total = int(total_pixels)
    chunks = []
    for start in range(0, total, int(n_pixels)):
        piece = dict(model_input)
        piece["uv"] = model_input["uv"][:, start:min(start + int(n_pixels), total)]
        chunks.append(piece)
    return chunks


def merge_output(res, total_pixels, batch_size):
    """concatenate per-chunk output dicts along the pixel axis: 1-D entries -> (B*total,), others -> (B*total, C);
    entries that are None in the first chunk are dropped (idr_utils.py:22-23)"""
    merged = {}
    n = int(batch_size) * int(total_pixels)
    for key, first in res[0].items():
        if first is None:
            continue
        if first.dim() == 1:
            merged[key] = torch.cat([r[key].reshape(batch_size, -1, 1) for r in res], 1).reshape(n)
        else:
            merged[key] = torch.cat([r[key].reshape(batch_size, -1, r[key].shape[-1]) for r in res], 1).reshape(n, -1)
    return merged
