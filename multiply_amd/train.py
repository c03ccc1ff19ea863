"""Training path: layer-wise fp32 forward with stash + hand-written backward (HIP), behind one torch.autograd.Function.

Replaces what torch autograd does for the reference's training iteration (multiply_model.py:192-217 around
Multiply.forward in training mode, multiply.py:254-545): the differentiable part of the forward (SDF net in forward
mode = value + spatial tangents, colour net, compositing, background) is evaluated layer by layer with the exact-fp32
MFMA GEMMs of csrc/gemm.hip, every pre-activation is kept, and the adjoint sweep -- including the mixed second
derivatives through the normals and the eikonal term -- is the reverse pass over that forward-mode graph.
Gradients are produced for every network parameter (weight-norm g/v, biases, lin_pose), density.beta and the frame
latent code.  Gradients w.r.t. the SMPL pose / translation (BodyModelParams) are NOT produced yet.

The non-differentiable sampler (VolSDF Algorithm 1, ray_sampler.py:81-191, `torch.no_grad()` in the reference) runs on
the fused bf16 kernels exactly like in eval mode, with the training-mode randomness drawn by torch.rand on the device.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import hip

F32 = torch.float32


def _p(t):
    return hip.ptr(t)


def _chk(code, what):
    hip.check(code, what)


def gemm_nt(A, lda, B, ldb, Cm, ldc, M, N, K, bias=None, bias_rows=0, accumulate=False, relu=False):
    _chk(hip.lib().mp_gemm_nt(A, lda, B, ldb, Cm, ldc, M, N, K, bias, bias_rows, int(accumulate), int(relu),
                              hip.stream()), "mp_gemm_nt")


def gemm_tn(A, lda, B, ldb, Cm, ldc, M, N, K):
    _chk(hip.lib().mp_gemm_tn(A, lda, B, ldb, Cm, ldc, M, N, K, hip.stream()), "mp_gemm_tn")


def off(t, n_floats):
    """device pointer `n_floats` floats into tensor t"""
    return C.c_void_p(t.data_ptr() + 4 * n_floats)


class LinW:
    """effective fp32 weights of one nn.Linear (weight-norm resolved), its transpose, and gradient buffers"""

    def __init__(self, lin):
        self.lin = lin
        self.wn = hasattr(lin, "weight_g")
        self.v = (lin.weight_v if self.wn else lin.weight).detach().contiguous()
        self.g = lin.weight_g.detach().reshape(-1).contiguous() if self.wn else None
        self.b = lin.bias.detach().contiguous()
        self.out_dim, self.in_dim = self.v.shape
        dev = self.v.device
        self.W = torch.empty(self.out_dim, self.in_dim, dtype=F32, device=dev)
        self.WT = torch.empty(self.in_dim, self.out_dim, dtype=F32, device=dev)
        _chk(hip.lib().mp_tr_wn_fwd(_p(self.v), _p(self.g), self.out_dim, self.in_dim, _p(self.W), _p(self.WT),
                                    hip.stream()), "mp_tr_wn_fwd")
        self.dW = torch.zeros(self.out_dim, self.in_dim, dtype=F32, device=dev)
        self.db = torch.zeros(self.out_dim, dtype=F32, device=dev)

    def param_grads(self):
        """gradients in the order of `params()`"""
        dv = torch.empty_like(self.v)
        dg = torch.empty(self.out_dim, 1, dtype=F32, device=self.v.device) if self.wn else None
        _chk(hip.lib().mp_tr_wn_bwd(_p(self.v), _p(self.g), self.out_dim, self.in_dim, _p(self.dW), _p(dv),
                                    _p(dg) if self.wn else None, hip.stream()), "mp_tr_wn_bwd")
        return [dg, dv, self.db] if self.wn else [dv, self.db]

    def params(self):
        lin = self.lin
        return [lin.weight_g, lin.weight_v, lin.bias] if self.wn else [lin.weight, lin.bias]


class ImplicitTrain:
    """ImplicitNet (networks.py:126-208) evaluated layer by layer for P points, optionally in forward mode."""

    def __init__(self, net, x, cond_vec, fwd):
        L = hip.lib()
        self.net, self.fwd = net, fwd
        dev = x.device
        self.P = P = x.shape[0]
        self.rows = rows = 4 * P if fwd else P
        self.E = E = net.embed_dim
        self.cond = cond_vec
        self.lins = [LinW(l) for l in net.layers()]
        nl = len(self.lins)
        self.IN = torch.empty(rows, E, dtype=F32, device=dev)
        _chk(L.mp_tr_pe(_p(x), net.d_in, P, net.multires, int(fwd), C.c_float(1.0), _p(self.IN), E, 0, hip.stream()),
             "mp_tr_pe")
        self.Z, self.X = [], []          # pre-activations and layer inputs
        r2 = 1.0 / math.sqrt(2.0)
        Pm = P if fwd else 0
        for l, lw in enumerate(self.lins):
            out = lw.out_dim
            Z = torch.empty(rows, out, dtype=F32, device=dev)
            if l == 0:
                self.b0 = torch.empty(out, dtype=F32, device=dev)
                _chk(L.mp_tr_hoist_fwd(_p(lw.W), out, lw.in_dim, _p(lw.b), E, net.cond_dim, _p(cond_vec), _p(self.b0),
                                       hip.stream()), "mp_tr_hoist_fwd")
                Xl = self.IN
                gemm_nt(_p(Xl), E, _p(lw.W), lw.in_dim, _p(Z), out, rows, out, E, _p(self.b0), P)
            else:
                Zp, prev_out = self.Z[l - 1], self.lins[l - 1].out_dim
                if l in net.skip_in:
                    Xl = torch.empty(rows, prev_out + E, dtype=F32, device=dev)
                    _chk(L.mp_tr_softplus_fwd(_p(Zp), prev_out, rows, prev_out, Pm, C.c_float(r2), _p(Xl), prev_out + E, 0,
                                              hip.stream()), "mp_tr_softplus_fwd")
                    _chk(L.mp_tr_copy_cols(_p(self.IN), E, 0, _p(Xl), prev_out + E, prev_out, rows, E, C.c_float(r2), 0,
                                           hip.stream()), "mp_tr_copy_cols")
                else:
                    Xl = torch.empty(rows, prev_out, dtype=F32, device=dev)
                    _chk(L.mp_tr_softplus_fwd(_p(Zp), prev_out, rows, prev_out, Pm, C.c_float(1.0), _p(Xl), prev_out, 0,
                                              hip.stream()), "mp_tr_softplus_fwd")
                gemm_nt(_p(Xl), Xl.shape[1], _p(lw.W), lw.in_dim, _p(Z), out, rows, out, lw.in_dim, _p(lw.b), P)
            self.Z.append(Z)
            self.X.append(Xl)
        self.out = self.Z[-1]            # [rows][257]

    def backward(self, dZ_last):
        """dZ_last [rows][257] -> accumulates dW/db of every layer; returns d cond (hoisted conditioning adjoint)"""
        L = hip.lib()
        net, rows, P, E = self.net, self.rows, self.P, self.E
        Pm = P if self.fwd else 0
        r2 = 1.0 / math.sqrt(2.0)
        dZ = dZ_last
        dcond = None
        for l in range(len(self.lins) - 1, -1, -1):
            lw, Xl = self.lins[l], self.X[l]
            out = lw.out_dim
            kin = E if l == 0 else lw.in_dim
            gemm_tn(_p(dZ), out, _p(Xl), Xl.shape[1], _p(lw.dW), lw.in_dim, out, kin, rows)
            _chk(L.mp_tr_colsum(_p(dZ), out, P, out, _p(lw.db), hip.stream()), "mp_tr_colsum")
            if l == 0:
                # hoisted conditioning: dW0[:, E:] += db (x) cond ; d cond = W0[:, E:]^T db
                _chk(L.mp_tr_hoist_bwd(_p(lw.db), out, lw.in_dim, E, net.cond_dim, _p(self.cond), _p(lw.dW), hip.stream()),
                     "mp_tr_hoist_bwd")
                dcond = torch.zeros(net.cond_dim, dtype=F32, device=dZ.device)
                gemm_tn(_p(lw.db), 1, off(lw.W, E), lw.in_dim, _p(dcond), net.cond_dim, 1, net.cond_dim, out)
                break
            prev_out = self.lins[l - 1].out_dim
            dX = torch.empty(rows, lw.in_dim, dtype=F32, device=dZ.device)
            gemm_nt(_p(dZ), out, _p(lw.WT), out, _p(dX), lw.in_dim, rows, lw.in_dim, out)
            dZp = torch.empty(rows, prev_out, dtype=F32, device=dZ.device)
            scale = r2 if l in net.skip_in else 1.0
            _chk(L.mp_tr_softplus_bwd(_p(self.Z[l - 1]), prev_out, rows, prev_out, Pm, C.c_float(scale), _p(dX), lw.in_dim,
                                      0, _p(dZp), prev_out, hip.stream()), "mp_tr_softplus_bwd")
            dZ = dZp
        return dcond

    def params(self):
        return [p for lw in self.lins for p in lw.params()]

    def param_grads(self):
        return [g for lw in self.lins for g in lw.param_grads()]


class RenderTrain:
    """RenderingNet (networks.py:263-312): mode 'pose_no_view' (inputs XA = [x_c, n] (6), feat) or 'nerf_frame_encoding'
    (XA = PE_4(view) (27), feat).  feat is read in place from the SDF net's last layer (ld 257, column 1..)."""

    def __init__(self, net, XA, feat_ptr, feat_ld, n, cond_vec):
        L = hip.lib()
        self.net, self.n = net, n
        dev = XA.device
        self.lins = [LinW(l) for l in net.layers()]
        self.pose_mode = net.mode == "pose_no_view"
        self.na = XA.shape[1]                                # 6 or 27
        self.c_h0, self.n_h = (6, 8) if self.pose_mode else (27, 32)   # hoisted columns
        self.c_feat = self.c_h0 + self.n_h
        lw0 = self.lins[0]
        self.cond = cond_vec
        if self.pose_mode:
            lp = net.lin_pose
            self.lp_w, self.lp_b = lp.weight.detach().contiguous(), lp.bias.detach().contiguous()
            self.pose8 = torch.empty(8, dtype=F32, device=dev)
            _chk(L.mp_tr_hoist_fwd(_p(self.lp_w), 8, 69, _p(self.lp_b), 0, 69, _p(cond_vec), _p(self.pose8), hip.stream()),
                 "mp_tr_hoist_fwd")
            self.hvec = self.pose8
        else:
            self.hvec = cond_vec
        self.b0 = torch.empty(lw0.out_dim, dtype=F32, device=dev)
        _chk(L.mp_tr_hoist_fwd(_p(lw0.W), lw0.out_dim, lw0.in_dim, _p(lw0.b), self.c_h0, self.n_h, _p(self.hvec),
                               _p(self.b0), hip.stream()), "mp_tr_hoist_fwd")
        self.XA, self.feat_ptr, self.feat_ld = XA, feat_ptr, feat_ld
        self.H = []
        nl = len(self.lins)
        H0 = torch.empty(n, lw0.out_dim, dtype=F32, device=dev)
        last0 = nl == 1
        gemm_nt(_p(XA), self.na, _p(lw0.W), lw0.in_dim, _p(H0), lw0.out_dim, n, lw0.out_dim, self.na, _p(self.b0), n)
        gemm_nt(feat_ptr, feat_ld, off(lw0.W, self.c_feat), lw0.in_dim, _p(H0), lw0.out_dim, n, lw0.out_dim, 256, None, 0,
                accumulate=True, relu=not last0)
        self.H.append(H0)
        for l in range(1, nl):
            lw = self.lins[l]
            Hl = torch.empty(n, lw.out_dim, dtype=F32, device=dev)
            gemm_nt(_p(self.H[l - 1]), self.lins[l - 1].out_dim, _p(lw.W), lw.in_dim, _p(Hl), lw.out_dim, n, lw.out_dim,
                    lw.in_dim, _p(lw.b), n, relu=l < nl - 1)
            self.H.append(Hl)
        self.rgb = torch.empty(n, 3, dtype=F32, device=dev)
        _chk(L.mp_tr_sigmoid_fwd(_p(self.H[-1]), n * 3, _p(self.rgb), hip.stream()), "mp_tr_sigmoid_fwd")

    def backward(self, drgb, dXA, dfeat_ptr, dfeat_ld):
        """drgb [n][3] -> dW/db, dXA [n][na] (written), d feat (+= into dfeat_ptr); returns d hoisted-vector"""
        L = hip.lib()
        n, dev = self.n, drgb.device
        nl = len(self.lins)
        dZ = torch.empty(n, 3, dtype=F32, device=dev)
        _chk(L.mp_tr_sigmoid_bwd(_p(self.rgb), _p(drgb), n * 3, _p(dZ), hip.stream()), "mp_tr_sigmoid_bwd")
        for l in range(nl - 1, 0, -1):
            lw, Hp = self.lins[l], self.H[l - 1]
            pout = self.lins[l - 1].out_dim
            gemm_tn(_p(dZ), lw.out_dim, _p(Hp), pout, _p(lw.dW), lw.in_dim, lw.out_dim, lw.in_dim, n)
            _chk(L.mp_tr_colsum(_p(dZ), lw.out_dim, n, lw.out_dim, _p(lw.db), hip.stream()), "mp_tr_colsum")
            dH = torch.empty(n, pout, dtype=F32, device=dev)
            gemm_nt(_p(dZ), lw.out_dim, _p(lw.WT), lw.out_dim, _p(dH), pout, n, pout, lw.out_dim)
            dZp = torch.empty(n, pout, dtype=F32, device=dev)
            _chk(L.mp_tr_relu_bwd(_p(Hp), pout, n, pout, _p(dH), pout, _p(dZp), pout, hip.stream()), "mp_tr_relu_bwd")
            dZ = dZp
        lw0 = self.lins[0]
        o0 = lw0.out_dim
        gemm_tn(_p(dZ), o0, _p(self.XA), self.na, _p(lw0.dW), lw0.in_dim, o0, self.na, n)
        gemm_tn(_p(dZ), o0, self.feat_ptr, self.feat_ld, off(lw0.dW, self.c_feat), lw0.in_dim, o0, 256, n)
        _chk(L.mp_tr_colsum(_p(dZ), o0, n, o0, _p(lw0.db), hip.stream()), "mp_tr_colsum")
        _chk(L.mp_tr_hoist_bwd(_p(lw0.db), o0, lw0.in_dim, self.c_h0, self.n_h, _p(self.hvec), _p(lw0.dW), hip.stream()),
             "mp_tr_hoist_bwd")
        dh = torch.zeros(self.n_h, dtype=F32, device=dev)
        gemm_tn(_p(lw0.db), 1, off(lw0.W, self.c_h0), lw0.in_dim, _p(dh), self.n_h, 1, self.n_h, o0)
        # data gradients
        gemm_nt(_p(dZ), o0, _p(lw0.WT), o0, _p(dXA), self.na, n, self.na, o0)
        gemm_nt(_p(dZ), o0, off(lw0.WT, self.c_feat * o0), o0, dfeat_ptr, dfeat_ld, n, 256, o0, None, 0, accumulate=True)
        self.extra_grads = []
        if self.pose_mode:
            dlp_w = torch.zeros(8, 69, dtype=F32, device=dev)
            _chk(L.mp_tr_hoist_bwd(_p(dh), 8, 69, 0, 69, _p(self.cond), _p(dlp_w), hip.stream()), "mp_tr_hoist_bwd")
            self.extra_grads = [dlp_w, dh]
        return dh

    def params(self):
        ps = [self.net.lin_pose.weight, self.net.lin_pose.bias] if self.pose_mode else []
        return ps + [p for lw in self.lins for p in lw.params()]

    def param_grads(self):
        return list(self.extra_grads) + [g for lw in self.lins for g in lw.param_grads()]
