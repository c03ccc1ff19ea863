"""Sampler configuration objects with the reference's names and constructor arguments
(code/lib/model/ray_sampler.py:14-64).  The algorithm itself (VolSDF Algorithm 1) runs in csrc/sampler.hip, driven by
Multiply.forward; these classes carry the hyper-parameters, and ErrorBoundSampler.get_z_vals keeps the reference's public
entry point for callers outside forward()."""
import torch


class RaySampler:
    def __init__(self, near, far):
        self.near = near
        self.far = far


class UniformSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, take_sphere_intersection=False, far=-1):
        super().__init__(near, 2.0 * scene_bounding_sphere if far == -1 else far)
        self.N_samples = N_samples
        self.scene_bounding_sphere = scene_bounding_sphere
        self.take_sphere_intersection = take_sphere_intersection


class ErrorBoundSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, N_samples_eval, N_samples_extra, eps, beta_iters,
                 max_total_iters, inverse_sphere_bg=False, N_samples_inverse_sphere=0, add_tiny=0.0):
        super().__init__(near, 2.0 * scene_bounding_sphere)
        self.N_samples = int(N_samples)
        self.N_samples_eval = int(N_samples_eval)
        self.N_samples_extra = int(N_samples_extra)
        self.eps = float(eps)
        self.beta_iters = int(beta_iters)
        self.max_total_iters = int(max_total_iters)
        self.scene_bounding_sphere = float(scene_bounding_sphere)
        self.add_tiny = float(add_tiny)
        self.inverse_sphere_bg = inverse_sphere_bg
        self.uniform_sampler = UniformSampler(scene_bounding_sphere, near, N_samples_eval,
                                              take_sphere_intersection=inverse_sphere_bg)
        # the reference overrides the configured count with 32 (ray_sampler.py:62-64)
        self.N_samples_inverse_sphere = 32
        if inverse_sphere_bg:
            self.inverse_sphere_sampler = UniformSampler(1.0, 0.0, 32, False, far=1.0)
        if not inverse_sphere_bg:
            raise NotImplementedError("the shipped configs always render with the inverted-sphere background")

    def get_z_vals(self, ray_dirs, cam_loc, model, cond, smpl_tfs, eval_mode, smpl_verts, person_id):
        """ray_sampler.py:66-220 for explicit rays: -> ((z_vals (R, N + N_extra + 2), z_vals_inverse_sphere (R, 32)),
        z_samples_eik (R, 1)) like the reference.  The depths come from the same device kernels Multiply.forward drives
        (Multiply.sample_rays).  `model.training` selects the reference's random branches (ray_sampler.py:32-40 stratified
        jitter, :171 random u of the final inverse-CDF draw, :202 randperm of the extra samples, and the jittered
        inverted-sphere depths); the draws are taken here with torch's generator on the rays' device, in the reference's order
        (multiply_amd.train.make_draws draws the same quantities for a whole training forward)."""
        dev = ray_dirs.device
        R = ray_dirs.reshape(-1, 3).shape[0]
        draws = None
        if model.training:
            NE, NS, NX = self.N_samples_eval, self.N_samples, self.N_samples_extra
            draws = dict(t_rand=torch.rand(R, NE, device=dev), u_final=torch.rand(R, NS, device=dev),
                         extra_idx=torch.stack([torch.randperm(NE * k, device=dev)[:NX] for k in range(1, self.max_total_iters + 1)]
                                               ).to(torch.int32).contiguous())
        z_vals = model.sample_rays(ray_dirs, cam_loc, cond, smpl_tfs, smpl_verts, person_id, draws=draws)
        idx = torch.randint(z_vals.shape[-1], (z_vals.shape[0],), device=z_vals.device)          # ray_sampler.py:212-213
        z_eik = torch.gather(z_vals, 1, idx.unsqueeze(-1))
        n_bg = self.inverse_sphere_sampler.N_samples
        t = torch.linspace(0.0, 1.0, steps=n_bg, device=z_vals.device)                            # near 0, far 1
        z_bg = t[None].expand(z_vals.shape[0], -1)
        if model.training:                                                                         # ray_sampler.py:32-40
            mids = 0.5 * (z_bg[..., 1:] + z_bg[..., :-1])
            upper, lower = torch.cat([mids, z_bg[..., -1:]], -1), torch.cat([z_bg[..., :1], mids], -1)
            z_bg = lower + (upper - lower) * torch.rand(z_bg.shape, device=z_vals.device)
        return (z_vals, z_bg * (1.0 / self.scene_bounding_sphere)), z_eik
