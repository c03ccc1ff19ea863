"""CPU restatement of the prompt construction around Segment-Anything -- TEST INFRASTRUCTURE ONLY (tests/; the product path
is multiply_amd/sam_prompts.py).  Follows code/lib/model/sam_model.py:58-231 statement by statement, with the global numpy
random stream replaced by an explicit RandomState (np.random.seed(42) in the reference = RandomState(42) here).
PARITY UNPINNED for the two third-party pieces it touches: SAM itself (absent: the tests use a stand-in predictor) and
cv2.resize (absent: `resize_nearest_half` below is a separate, loop-free formulation of the 8-bit bilinear downscale)."""
import numpy as np


def resize_to_256(canvas):
    """bilinear resize of an (n, n) uint8 image to 256 x 256 written as an explicit gather of the four neighbours"""
    n = canvas.shape[0]
    out = np.zeros((256, 256), dtype=np.uint8)
    scale = n / 256.0
    for i in range(256):
        fy = (i + 0.5) * scale - 0.5
        y0 = int(np.floor(fy)); wy = fy - y0
        ya, yb = min(max(y0, 0), n - 1), min(max(y0 + 1, 0), n - 1)
        fx = (np.arange(256) + 0.5) * scale - 0.5
        x0 = np.floor(fx).astype(int); wx = fx - x0
        xa, xb = np.clip(x0, 0, n - 1), np.clip(x0 + 1, 0, n - 1)
        top = canvas[ya, xa] * (1 - wx) + canvas[ya, xb] * wx
        bot = canvas[yb, xa] * (1 - wx) + canvas[yb, xb] * wx
        out[i] = np.floor(top * (1 - wy) + bot * wy + 0.5).astype(np.uint8)
    return out


def person_prompts(image_mask_all, smpl_joint_i, person_id, rng):
    """-> input_point, input_label, bounding_box, resized_mask_logit[None]  for one person of one frame"""
    image_mask = image_mask_all[person_id]
    negative_image_mask_list = []
    for neg_person_i in range(image_mask_all.shape[0]):
        if neg_person_i != person_id:
            negative_image_mask_list.append(image_mask_all[neg_person_i])
    negative_image_mask = np.max(np.stack(negative_image_mask_list, axis=0), axis=0)
    indices = np.argwhere(image_mask)
    x_min, y_min = np.min(indices[:, 1]), np.min(indices[:, 0])
    x_max, y_max = np.max(indices[:, 1]), np.max(indices[:, 0])
    x_min = max(0, x_min - int(0.03 * (x_max - x_min)))
    y_min = max(0, y_min - int(0.03 * (y_max - y_min)))
    x_max = min(image_mask.shape[1], x_max + int(0.03 * (x_max - x_min)))
    y_max = min(image_mask.shape[0], y_max + int(0.03 * (y_max - y_min)))
    bounding_box = np.array([x_min, y_min, x_max, y_max])
    height, width = image_mask.shape
    max_dim = max(height, width)
    canvas = np.zeros((max_dim, max_dim), dtype=np.uint8)
    if height > width:
        canvas[0:height, 0:width] = image_mask
    else:
        canvas[0:height, max_dim - width:max_dim] = image_mask
    resized_mask = resize_to_256(canvas)
    positive_point_candidate = smpl_joint_i[person_id, :27]
    negative_point_candidate_list = []
    for neg_person_i in range(image_mask_all.shape[0]):
        if neg_person_i != person_id:
            negative_point_candidate_list.append(smpl_joint_i[neg_person_i, :27])
    negative_point_candidate = np.concatenate(negative_point_candidate_list, axis=0)
    point_list = []
    for j in range(positive_point_candidate.shape[0]):
        p = positive_point_candidate[j]
        try:
            if image_mask[p[1], p[0]] > 0.7:
                point_list.append(p)
        except Exception:
            pass
    positive_points = np.array(point_list)
    if len(positive_points) == 0:
        positive_point_list = []
        try_time = 0
        while len(positive_points) < 1 and try_time < 10000000:
            x = rng.randint(0, image_mask.shape[1])
            y = rng.randint(0, image_mask.shape[0])
            try_time += 1
            if image_mask[y, x] > 0.7:
                positive_point_list.append([x, y])
                positive_points = np.array(positive_point_list)
                break
        if len(positive_points) == 0:
            positive_point_list.append(positive_point_candidate[-1])
            positive_points = np.array(positive_point_list)
    num_positive_points = len(positive_points)
    positive_labels = np.ones(num_positive_points)
    sampled = rng.choice(num_positive_points, num_positive_points, replace=False)
    sampled_positive_points, sampled_positive_labels = positive_points[sampled], positive_labels[sampled]
    negative_points = []
    while len(negative_points) < 10:
        x = rng.randint(0, image_mask.shape[1])
        y = rng.randint(0, image_mask.shape[0])
        if image_mask[y, x] == 0:
            negative_points.append([x, y])
    for j in range(negative_point_candidate.shape[0]):
        p = negative_point_candidate[j]
        try:
            if image_mask[p[1], p[0]] < 0.7 and negative_image_mask[p[1], p[0]] > 0.7:
                negative_points.append([p[0], p[1]])
        except Exception:
            pass
    negative_labels = np.zeros(len(negative_points))
    negative_points = np.array(negative_points)
    input_point = np.concatenate((sampled_positive_points, negative_points), axis=0)
    input_label = np.concatenate((sampled_positive_labels, negative_labels))
    p01 = np.clip(resized_mask.astype(np.float32), 1e-6, 1 - 1e-6)
    return input_point, input_label, bounding_box, np.log(p01 / (1 - p01))[None]
