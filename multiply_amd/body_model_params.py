"""Per-frame SMPL parameter tables the reference's trainer optimises (code/lib/model/body_model_params.py:5-49): one
embedding row per training frame for global_orient (3), body_pose (69) and transl (3), ONE shared row for betas (10).
Interface kept verbatim (constructor, init_parameters, set_requires_grad, forward(frame_ids) -> dict); the rows feed
Multiply.forward's smpl_pose / smpl_trans / smpl_shape inputs and receive gradients from the hand-written backward
(multiply_amd/train.py: mp_tr_warp_bwd -> mp_smpl_pose_bwd)."""
import torch
import torch.nn as nn


class BodyModelParams(nn.Module):
    PER_FRAME = {"global_orient": 3, "transl": 3, "body_pose": 69}

    def __init__(self, num_frames, model_type="smpl"):
        super().__init__()
        if model_type != "smpl":
            raise ValueError(f"Unknown model type {model_type}")
        self.num_frames, self.model_type = num_frames, model_type
        self.params_dim = {"betas": 10, "global_orient": 3, "transl": 3, "body_pose": 69}
        self.param_names = self.params_dim.keys()
        for name, dim in self.params_dim.items():           # registration order = the reference's (state-dict order)
            table = nn.Embedding(1 if name == "betas" else num_frames, dim)
            table.weight.data.zero_()
            table.weight.requires_grad = False
            setattr(self, name, table)

    def init_parameters(self, param_name, data, requires_grad=False):
        table = getattr(self, param_name)
        table.weight.data = data[..., :self.params_dim[param_name]]
        table.weight.requires_grad = requires_grad

    def set_requires_grad(self, param_name, requires_grad=True):
        getattr(self, param_name).weight.requires_grad = requires_grad

    def forward(self, frame_ids):
        shared = torch.zeros_like(frame_ids)                 # betas: row 0 for every frame
        return {name: getattr(self, name)(shared if name == "betas" else frame_ids) for name in self.param_names}
