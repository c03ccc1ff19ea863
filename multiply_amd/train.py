"""Training path: layer-wise fp32 forward with stash + hand-written backward (HIP), behind one torch.autograd.Function.

Replaces what torch autograd does for the reference's training iteration (multiply_model.py:192-217 around
Multiply.forward in training mode, multiply.py:254-545): the differentiable part of the forward (SDF net in forward
mode = value + spatial tangents, colour net, compositing, background) is evaluated layer by layer with the exact-fp32
MFMA GEMMs of csrc/gemm.hip, every pre-activation is kept, and the adjoint sweep -- including the mixed second
derivatives through the normals and the eikonal term -- is the reverse pass over that forward-mode graph.
Gradients are produced for every network parameter (weight-norm g/v, biases, lin_pose), density.beta and the frame
latent code, and -- when the caller's smpl_pose / smpl_trans / smpl_shape require grad (BodyModelParams in the reference's
trainer) -- for those too, through the canonical warp, the normals' Jacobian, the pose conditioning and SMPL's bone transforms.

The non-differentiable sampler (VolSDF Algorithm 1, ray_sampler.py:81-191, `torch.no_grad()` in the reference) runs on
the fused half-precision kernels exactly like in eval mode, with the training-mode randomness drawn by torch.rand on the device.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.distributed as dist

from . import hip

F32 = torch.float32


def _p(t):
    return hip.ptr(t)


_DEBUG_SYNC = bool(int(__import__("os").environ.get("MP_DEBUG_SYNC", "0")))


def _chk(code, what):
    hip.check(code, what)
    if _DEBUG_SYNC:                      # debugging aid: attribute asynchronous faults to the launch that caused them
        torch.cuda.synchronize()
        print("[mp sync ok]", what, flush=True)


# Arithmetic of the training GEMMs (csrc/gemm.hip):
#   "bf16x3" (default): every fp32 operand split into two bfloat16 halves on its way into LDS, three 16-bit MFMAs per product
#            (hi.hi + hi.lo + lo.hi), fp32 accumulation: ~2^-16 relative per product, the range of fp32 (no loss scaling),
#            5x less matrix-pipe time than the exact-fp32 instruction -- the GEMMs run at the rate HBM delivers their operands;
#   "f32":   v_mfma_f32_16x16x4_f32, bitwise an fmaf chain -- the cross-check (tests/test_train_step_gpu.py runs both).
TRAIN_PRECISION = __import__("os").environ.get("MP_TRAIN_PRECISION", "bf16x3")


def gemm_nt(A, lda, B, ldb, Cm, ldc, M, N, K, bias=None, bias_rows=0, accumulate=False, relu=False):
    if TRAIN_PRECISION == "f32":
        fn, name = hip.lib().mp_gemm_nt, "mp_gemm_nt"
    elif TRAIN_PRECISION == "bf16x3":
        fn, name = hip.lib().mp_gemm_nt_bf16x3, "mp_gemm_nt_bf16x3"
    else:
        raise ValueError(f"MP_TRAIN_PRECISION {TRAIN_PRECISION!r}: expected 'bf16x3' or 'f32'")
    _chk(fn(A, lda, B, ldb, Cm, ldc, M, N, K, bias, bias_rows, int(accumulate), int(relu), hip.stream()), name)


def gemm_tn(A, lda, B, ldb, Cm, ldc, M, N, K, colsum=None, colsum_rows=0):
    """Cm[M,N] += A[K,M]^T B[K,N];  colsum[M] += column sums of A's first colsum_rows rows (the bias gradient)"""
    fn = hip.lib().mp_gemm_tn if TRAIN_PRECISION == "f32" else hip.lib().mp_gemm_tn_bf16x3
    _chk(fn(A, lda, B, ldb, Cm, ldc, M, N, K, colsum, colsum_rows, hip.stream()), "mp_gemm_tn")


class MpTnGroup(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("colsum", C.c_void_p), ("lda", C.c_int),
                ("ldb", C.c_int), ("ldc", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("colsum_rows", C.c_int),
                ("pad_", C.c_int)]


def tn_group(A, lda, B, ldb, Cm, ldc, M, N, K, colsum=None, colsum_rows=0):
    """one contraction of a grouped launch (mp_gemm_tn_bf16x3_grouped); pointers as returned by _p() / off()"""
    return MpTnGroup(A.value, B.value, Cm.value, colsum.value if colsum is not None else None, lda, ldb, ldc, M, N, K,
                     colsum_rows, 0)


def gemm_tn_grouped(groups):
    """all weight-gradient contractions of `groups` in ONE launch (aligned 128-multiples only); chunks of 24"""
    for i in range(0, len(groups), 24):
        chunk = groups[i:i + 24]
        arr = (MpTnGroup * len(chunk))(*chunk)
        _chk(hip.lib().mp_gemm_tn_bf16x3_grouped(arr, len(chunk), hip.stream()), "mp_gemm_tn_bf16x3_grouped")


def _big_empty(n_floats, dev, grain=1 << 26, cap=None, cap_bytes=6 << 30):
    """fp32 scratch of at least n_floats, allocated in multiples of `grain` floats (256 MiB).  The per-iteration stashes are
    gigabytes whose exact size follows the number of rays that hit each body, i.e. changes every iteration: an exact-size
    request misses torch's caching allocator whenever it exceeds every cached block, and a fresh hipMalloc of 3 GB stalls the
    host for milliseconds (measured: 520 torch.empty calls = 12 ms of host time per iteration, nearly all of it in the two
    arena allocations).  A few coarse sizes are cached after the first iterations and always hit.  cap (floats): an upper bound of
    every request of this call site; when it is affordable (<= cap_bytes) it is what is allocated -- one size for good, so that no
    later iteration with a few more hit rays pays a fresh hipMalloc inside a timed region."""
    if cap is not None and n_floats <= cap and 4 * cap <= cap_bytes:
        n_floats = cap
    return torch.empty((n_floats + grain - 1) // grain * grain, dtype=F32, device=dev)


_ZP = [None]          # the running adjoint sweep's hip.ZeroPool (TrainGraph.backward); None outside a sweep


def _zeros(*shape, device):
    """zero-initialised fp32 tensor: a view of the sweep's zero-filled block when a sweep is running, else a fill of its own"""
    zp = _ZP[0]
    if zp is not None and zp.device == torch.device(device):
        return zp.take(*shape)
    return torch.zeros(*shape, dtype=F32, device=device)


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

    def __init__(self, net, x, cond_vec, fwd, lins=None):
        L = hip.lib()
        self.net, self.fwd, self.x = net, fwd, x
        dev = x.device
        self.P = P = x.shape[0]
        self.rows = rows = 4 * P if fwd else P
        self.E = E = net.embed_dim
        self.cond = cond_vec
        self.lins = lins if lins is not None else [LinW(l) for l in net.layers()]
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

    def backward(self, dZ_last, want_dx=False):
        """dZ_last [rows][257] -> accumulates dW/db of every layer; returns d cond (hoisted conditioning adjoint).
        want_dx: also the adjoint of the input points, self.dx [P][d_in] (pose optimisation)."""
        L = hip.lib()
        net, rows, P, E = self.net, self.rows, self.P, self.E
        Pm = P if self.fwd else 0
        r2 = 1.0 / math.sqrt(2.0)
        dZ = dZ_last
        dcond = None
        dIN = torch.zeros(rows, E, dtype=F32, device=dZ.device) if want_dx else None
        # the 256 x 256 weight gradients wait for ONE grouped launch at the end (six small contractions launched one by one
        # cost 64 us each; their operands stay alive in `held`)
        groups, held = [], []
        for l in range(len(self.lins) - 1, -1, -1):
            lw, Xl = self.lins[l], self.X[l]
            out = lw.out_dim
            kin = E if l == 0 else lw.in_dim
            if TRAIN_PRECISION == "bf16x3" and l > 0 and out == 256 and kin == 256 and Xl.shape[1] == 256:
                groups.append(tn_group(_p(dZ), out, _p(Xl), 256, _p(lw.dW), lw.in_dim, out, kin, rows, _p(lw.db), P))
                held.append(dZ)
            elif l == 0:
                # layer 0's bias gradient of THIS evaluation on its own (db0), then added to the accumulator: the hoisted
                # conditioning's adjoint below must not see what other evaluations of the same network left in lw.db
                # (the zero-pose regulariser evaluates a network under two conditionings in one sweep)
                db0 = _zeros(out, device=dZ.device)
                gemm_tn(_p(dZ), out, _p(Xl), Xl.shape[1], _p(lw.dW), lw.in_dim, out, kin, rows, _p(db0), P)
                lw.db.add_(db0)
            else:
                gemm_tn(_p(dZ), out, _p(Xl), Xl.shape[1], _p(lw.dW), lw.in_dim, out, kin, rows, _p(lw.db), P)
            if l == 0:
                # hoisted conditioning: dW0[:, E:] += db (x) cond ; d cond = W0[:, E:]^T db
                _chk(L.mp_tr_hoist_bwd(_p(db0), out, lw.in_dim, E, net.cond_dim, _p(self.cond), _p(lw.dW), hip.stream()),
                     "mp_tr_hoist_bwd")
                dcond = _zeros(net.cond_dim, device=dZ.device)
                gemm_tn(_p(db0), 1, off(lw.W, E), lw.in_dim, _p(dcond), net.cond_dim, 1, net.cond_dim, out)
                if want_dx:
                    gemm_nt(_p(dZ), out, _p(lw.WT), out, _p(dIN), E, rows, E, out, accumulate=True)
                    self.dx = torch.zeros(P, net.d_in, dtype=F32, device=dZ.device)
                    _chk(L.mp_tr_pe_bwd(_p(self.x), net.d_in, P, net.multires, int(self.fwd), _p(dIN), E, _p(self.dx),
                                        hip.stream()), "mp_tr_pe_bwd")
                break
            prev_out = self.lins[l - 1].out_dim
            dX = torch.empty(rows, lw.in_dim, dtype=F32, device=dZ.device)
            gemm_nt(_p(dZ), out, _p(lw.WT), out, _p(dX), lw.in_dim, rows, lw.in_dim, out)
            if want_dx and l in net.skip_in:      # the skip connection's copy of the encoded input
                _chk(L.mp_tr_copy_cols(_p(dX), lw.in_dim, prev_out, _p(dIN), E, 0, rows, E, C.c_float(r2), 1, hip.stream()),
                     "mp_tr_copy_cols")
            dZp = torch.empty(rows, prev_out, dtype=F32, device=dZ.device)
            scale = r2 if l in net.skip_in else 1.0
            _chk(L.mp_tr_softplus_bwd(_p(self.Z[l - 1]), prev_out, rows, prev_out, Pm, C.c_float(scale), _p(dX), lw.in_dim,
                                      0, _p(dZp), prev_out, hip.stream()), "mp_tr_softplus_bwd")
            dZ = dZp
        if groups:
            gemm_tn_grouped(groups)
        return dcond

    def params(self):
        return [p for lw in self.lins for p in lw.params()]

    def param_grads(self):
        return [g for lw in self.lins for g in lw.param_grads()]


class ImplicitTrainRev:
    """Foreground ImplicitNet for P points with d sdf / d x by REVERSE-over-reverse differentiation:

        forward   Z_l = X_l W_l^T + b_l,  X_{l+1} = softplus(Z_l)                       (value only, P rows)
        sweep     V_7 = s_7 (.) W_8[sdf row];  U_{l-1} = V_l W_l;  V_{l-1} = s_{l-1} (.) U_{l-1};  grad = J_PE^T (V_0 W_0in + ...)
        backward  the adjoint of BOTH sweeps (sigma'' enters through d s_l = U_l (.) dV_l)

    = what torch autograd does for the reference (multiply.py:643-659 with create_graph=True): 6 GEMMs per layer over P
    rows, where the forward-mode class above spends 3 GEMMs over 4P rows.  Same results, same parameter gradients.
    self.out [P][257] = last layer, self.grad [P][3] = d sdf / d x."""

    def __init__(self, net, x, cond_vec, lins=None):
        L = hip.lib()
        st = hip.stream()
        self.net, self.x, self.cond = net, x, cond_vec
        dev = x.device
        self.P = P = x.shape[0]
        self.E = E = net.embed_dim
        assert net.d_in == 3 and len(net.skip_in) == 1
        self.lins = lins = lins if lins is not None else [LinW(l) for l in net.layers()]
        self.nl = nl = len(lins)
        r2 = 1.0 / math.sqrt(2.0)
        f32 = dict(dtype=F32, device=dev)
        self.IN = torch.empty(P, E, **f32)
        _chk(L.mp_tr_pe(_p(x), 3, P, net.multires, 0, C.c_float(1.0), _p(self.IN), E, 0, st), "mp_tr_pe")
        # ---- value sweep
        self.Z, self.X = [], []
        for l, lw in enumerate(lins):
            out = lw.out_dim
            Z = torch.empty(P, out, **f32)
            if l == 0:
                self.b0 = torch.empty(out, **f32)
                _chk(L.mp_tr_hoist_fwd(_p(lw.W), out, lw.in_dim, _p(lw.b), E, net.cond_dim, _p(cond_vec), _p(self.b0), st),
                     "mp_tr_hoist_fwd")
                Xl = self.IN
                gemm_nt(_p(Xl), E, _p(lw.W), lw.in_dim, _p(Z), out, P, out, E, _p(self.b0), P)
            else:
                po = lins[l - 1].out_dim
                if l in net.skip_in:
                    Xl = torch.empty(P, po + E, **f32)
                    _chk(L.mp_tr_softplus_fwd(_p(self.Z[l - 1]), po, P, po, 0, C.c_float(r2), _p(Xl), po + E, 0, st), "softplus")
                    _chk(L.mp_tr_copy_cols(_p(self.IN), E, 0, _p(Xl), po + E, po, P, E, C.c_float(r2), 0, st), "copy_cols")
                else:
                    Xl = torch.empty(P, po, **f32)
                    _chk(L.mp_tr_softplus_fwd(_p(self.Z[l - 1]), po, P, po, 0, C.c_float(1.0), _p(Xl), po, 0, st), "softplus")
                gemm_nt(_p(Xl), Xl.shape[1], _p(lw.W), lw.in_dim, _p(Z), out, P, out, lw.in_dim, _p(lw.b), P)
            self.Z.append(Z)
            self.X.append(Xl)
        self.out = self.Z[-1]
        # ---- reverse sweep for d sdf / d x
        nh = nl - 1                                   # hidden layers 0..nh-1
        self.w8 = lins[nh].W[0].contiguous()          # sdf row of the last layer
        self.V = [None] * nh
        self.T = [None] * nh                          # T[l] = V_l W_l  (U_{l-1} = scale_l * T[l][:, :out_{l-1}])
        self.V[nh - 1] = torch.empty(P, lins[nh - 1].out_dim, **f32)
        _chk(L.mp_tr_sigmul(_p(self.Z[nh - 1]), lins[nh - 1].out_dim, P, lins[nh - 1].out_dim, None, 0, _p(self.w8),
                            C.c_float(1.0), _p(self.V[nh - 1]), lins[nh - 1].out_dim, st), "mp_tr_sigmul")
        self.Gpe = torch.zeros(P, E, **f32)
        for l in range(nh - 1, 0, -1):
            lw, po = lins[l], lins[l - 1].out_dim
            T = torch.empty(P, lw.in_dim, **f32)
            gemm_nt(_p(self.V[l]), lw.out_dim, _p(lw.WT), lw.out_dim, _p(T), lw.in_dim, P, lw.in_dim, lw.out_dim)
            sc = r2 if l in net.skip_in else 1.0
            if l in net.skip_in:
                _chk(L.mp_tr_copy_cols(_p(T), lw.in_dim, po, _p(self.Gpe), E, 0, P, E, C.c_float(r2), 1, st), "copy_cols")
            self.T[l] = T
            self.V[l - 1] = torch.empty(P, po, **f32)
            _chk(L.mp_tr_sigmul(_p(self.Z[l - 1]), po, P, po, _p(T), lw.in_dim, None, C.c_float(sc), _p(self.V[l - 1]), po, st),
                 "mp_tr_sigmul")
        lw0 = lins[0]
        gemm_nt(_p(self.V[0]), lw0.out_dim, _p(lw0.WT), lw0.out_dim, _p(self.Gpe), E, P, E, lw0.out_dim, accumulate=True)
        self.grad = torch.empty(P, 3, **f32)
        _chk(L.mp_tr_pe_grad_fwd(_p(x), P, net.multires, _p(self.Gpe), E, _p(self.grad), st), "mp_tr_pe_grad_fwd")

    def backward(self, dZ_last, dgrad, want_dx=False):
        """dZ_last [P][257], dgrad [P][3] -> dW/db of every layer; returns d cond; want_dx: self.dx [P][3]"""
        L = hip.lib()
        st = hip.stream()
        net, P, E, lins = self.net, self.P, self.E, self.lins
        nh = self.nl - 1
        dev = dZ_last.device
        f32 = dict(dtype=F32, device=dev)
        r2 = 1.0 / math.sqrt(2.0)
        # ---- adjoint of the reverse sweep (ascending l)
        dGpe = torch.empty(P, E, **f32)
        self.dx = torch.zeros(P, 3, **f32) if want_dx else None
        _chk(L.mp_tr_pe_grad_bwd(_p(self.x), P, net.multires, _p(dgrad), _p(self.Gpe), E, _p(dGpe), E, _p(self.dx), st),
             "mp_tr_pe_grad_bwd")
        lw0 = lins[0]
        dV = torch.empty(P, lw0.out_dim, **f32)
        gemm_nt(_p(dGpe), E, _p(lw0.W), lw0.in_dim, _p(dV), lw0.out_dim, P, lw0.out_dim, E)
        gemm_tn(_p(self.V[0]), lw0.out_dim, _p(dGpe), E, _p(lw0.dW), lw0.in_dim, lw0.out_dim, E, P)
        dS = [None] * nh
        for l in range(0, nh - 1):
            lw1 = lins[l + 1]
            out_l = lins[l].out_dim
            sc = r2 if (l + 1) in net.skip_in else 1.0
            dU = torch.empty(P, out_l, **f32)
            dS[l] = torch.empty(P, out_l, **f32)
            _chk(L.mp_tr_rev_adj(_p(self.Z[l]), out_l, P, out_l, _p(self.T[l + 1]), lw1.in_dim, None, C.c_float(sc), _p(dV), out_l,
                                 _p(dU), out_l, _p(dS[l]), out_l, st), "mp_tr_rev_adj")
            if (l + 1) in net.skip_in:
                dT = torch.empty(P, lw1.in_dim, **f32)
                _chk(L.mp_tr_copy_cols(_p(dU), out_l, 0, _p(dT), lw1.in_dim, 0, P, out_l, C.c_float(r2), 0, st), "copy_cols")
                _chk(L.mp_tr_copy_cols(_p(dGpe), E, 0, _p(dT), lw1.in_dim, out_l, P, E, C.c_float(r2), 0, st), "copy_cols")
            else:
                dT = dU
            dV = torch.empty(P, lw1.out_dim, **f32)
            gemm_nt(_p(dT), lw1.in_dim, _p(lw1.W), lw1.in_dim, _p(dV), lw1.out_dim, P, lw1.out_dim, lw1.in_dim)
            gemm_tn(_p(self.V[l + 1]), lw1.out_dim, _p(dT), lw1.in_dim, _p(lw1.dW), lw1.in_dim, lw1.out_dim, lw1.in_dim, P)
        # top of the sweep: V_7 = s_7 (.) W_8[sdf row]
        top = lins[nh - 1].out_dim
        dU = torch.empty(P, top, **f32)
        dS[nh - 1] = torch.empty(P, top, **f32)
        _chk(L.mp_tr_rev_adj(_p(self.Z[nh - 1]), top, P, top, None, 0, _p(self.w8), C.c_float(1.0), _p(dV), top, _p(dU), top,
                             _p(dS[nh - 1]), top, st), "mp_tr_rev_adj")
        dw8 = torch.empty(top, **f32)
        _chk(L.mp_tr_colsum(_p(dU), top, P, top, _p(dw8), st), "mp_tr_colsum")
        lins[nh].dW[0] += dw8
        # ---- adjoint of the value sweep (descending l)
        dZ = dZ_last
        dcond = None
        dIN = torch.zeros(P, E, **f32) if want_dx else None
        for l in range(nh, -1, -1):
            lw, Xl = lins[l], self.X[l]
            out = lw.out_dim
            kin = E if l == 0 else lw.in_dim
            if l == 0:           # (this evaluation's own bias gradient: see ImplicitTrain.backward)
                db0 = _zeros(out, device=dZ.device)
                gemm_tn(_p(dZ), out, _p(Xl), Xl.shape[1], _p(lw.dW), lw.in_dim, out, kin, P, _p(db0), P)
                lw.db.add_(db0)
            else:
                gemm_tn(_p(dZ), out, _p(Xl), Xl.shape[1], _p(lw.dW), lw.in_dim, out, kin, P, _p(lw.db), P)
            if l == 0:
                _chk(L.mp_tr_hoist_bwd(_p(db0), out, lw.in_dim, E, net.cond_dim, _p(self.cond), _p(lw.dW), st), "mp_tr_hoist_bwd")
                dcond = torch.zeros(net.cond_dim, **f32)
                gemm_tn(_p(db0), 1, off(lw.W, E), lw.in_dim, _p(dcond), net.cond_dim, 1, net.cond_dim, out)
                if want_dx:
                    gemm_nt(_p(dZ), out, _p(lw.WT), out, _p(dIN), E, P, E, out, accumulate=True)
                    _chk(L.mp_tr_pe_bwd(_p(self.x), 3, P, net.multires, 0, _p(dIN), E, _p(self.dx), st), "mp_tr_pe_bwd")
                break
            po = lins[l - 1].out_dim
            dX = torch.empty(P, lw.in_dim, **f32)
            gemm_nt(_p(dZ), out, _p(lw.WT), out, _p(dX), lw.in_dim, P, lw.in_dim, out)
            sc = r2 if l in net.skip_in else 1.0
            if want_dx and l in net.skip_in:
                _chk(L.mp_tr_copy_cols(_p(dX), lw.in_dim, po, _p(dIN), E, 0, P, E, C.c_float(r2), 1, st), "copy_cols")
            dZp = torch.empty(P, po, **f32)
            _chk(L.mp_tr_dz(_p(self.Z[l - 1]), po, P, po, _p(dX), lw.in_dim, C.c_float(sc), _p(dS[l - 1]), po, _p(dZp), po, st),
                 "mp_tr_dz")
            dZ = dZp
        return dcond

    def params(self):
        return [p for lw in self.lins for p in lw.params()]

    def param_grads(self):
        return [g for lw in self.lins for g in lw.param_grads()]


class MpWnDesc(C.Structure):
    _fields_ = [("v", C.c_void_p), ("g", C.c_void_p), ("W", C.c_void_p), ("WT", C.c_void_p), ("dW_off", C.c_longlong),
                ("dv_off", C.c_longlong), ("dg_off", C.c_longlong), ("out_dim", C.c_int), ("in_dim", C.c_int), ("row0", C.c_int),
                ("pad_", C.c_int)]


class LinP(LinW):
    """LinW with PERSISTENT effective weights (W, optionally the transpose WT) and, under a TrainState, gradient accumulators /
    parameter gradients that are views of the iteration's two flat buffers (one fill, one batched weight-norm launch per group
    instead of two launches per layer).  Standalone (no TrainState): refresh() resolves the weight norm and zeroes its own
    accumulators, param_grads() runs the per-layer adjoint like LinW."""

    def __init__(self, lin, pad_rows=0, need_wt=False, standalone=True):
        """pad_rows: gradient accumulators of max(out_dim, pad_rows) rows (a 217-row layer contracted as 256 rows takes the
        aligned path of mp_gemm_tn; the extra rows receive exact zeros and are never read)"""
        self.lin = lin
        self.wn = hasattr(lin, "weight_g")
        v = lin.weight_v if self.wn else lin.weight
        self.out_dim, self.in_dim = v.shape
        dev = v.device
        self.W = torch.empty(self.out_dim, self.in_dim, dtype=F32, device=dev)
        self.WT = torch.empty(self.in_dim, self.out_dim, dtype=F32, device=dev) if need_wt else None
        self.rows = max(self.out_dim, pad_rows)
        self.standalone = standalone
        self._g = None
        self.read_params()
        if standalone:
            self.dW_full = torch.zeros(self.rows, self.in_dim, dtype=F32, device=dev)
            self.db_full = torch.zeros(self.rows, dtype=F32, device=dev)
            self.dW, self.db = self.dW_full[:self.out_dim], self.db_full[:self.out_dim]
            self.refresh()

    def read_params(self):
        lin = self.lin
        self.v = (lin.weight_v if self.wn else lin.weight).detach().contiguous()
        self.g = lin.weight_g.detach().reshape(-1).contiguous() if self.wn else None
        self.b = lin.bias.detach().contiguous()

    def refresh(self):
        self.read_params()
        _chk(hip.lib().mp_tr_wn_fwd(_p(self.v), _p(self.g), self.out_dim, self.in_dim, _p(self.W), _p(self.WT), hip.stream()),
             "mp_tr_wn_fwd")
        self.dW_full.zero_()
        self.db_full.zero_()

    def bind(self, acc, o_dW, o_db, gbuf, o_dv, o_dg):
        """this iteration's accumulators / gradients: views of the flat buffers (TrainState.begin)"""
        n = self.rows * self.in_dim
        self.dW_full = acc[o_dW:o_dW + n].view(self.rows, self.in_dim)
        self.db_full = acc[o_db:o_db + self.rows]
        self.dW, self.db = self.dW_full[:self.out_dim], self.db_full[:self.out_dim]
        self._g = (gbuf, o_dv, o_dg)

    def param_grads(self):
        if self.standalone:
            gs = super().param_grads()
            gs[-1] = self.db.clone()          # the accumulator itself lives on: never hand it to autograd
            return gs
        # TrainState.finish_group ran the batched adjoint.  FRESH view objects on purpose: autograd's AccumulateGrad takes a
        # gradient over without a copy only if nothing else references the tensor object (a view kept here would make it
        # clone all ~110 gradients of an iteration)
        gbuf, o_dv, o_dg = self._g
        db = self.db_full[:self.out_dim]
        if not self.wn:
            return [self.dW_full[:self.out_dim], db]
        dv = gbuf[o_dv:o_dv + self.out_dim * self.in_dim].view(self.out_dim, self.in_dim)
        return [gbuf[o_dg:o_dg + self.out_dim].view(self.out_dim, 1), dv, db]


class TrainState:
    """Per-model state of the training path, shared by every network of an iteration: persistent effective weights, ONE batched
    weight-norm launch per group (a person's two networks; the background's two) in the forward and one in the backward, and two
    flat per-iteration buffers -- gradient accumulators (one fill) and parameter gradients -- of which every layer's tensors are
    views.  Replaces ~80 weight-norm launches, ~80 fills and ~40 allocations per iteration."""

    def __init__(self, model):
        m = self.model = model
        self.groups = []                      # (key, [nets])
        for p in range(len(m.foreground_implicit_network_list)):
            self.groups.append((p, [m.foreground_implicit_network_list[p], m.foreground_rendering_network_list[p]]))
        self.groups.append(("bg", [m.bg_implicit_network, m.bg_rendering_network]))
        self.lins = {}
        self.acc_size = self.grad_size = 0
        self.layout = {}                      # id(LinP) -> offsets
        self.tables = {}
        for key, nets in self.groups:
            for net in nets:
                fused = isinstance(net, _implicit_net_type()) and (fused_sdf_supported(net) or fused_bg_supported(net))
                ls = []
                for i, lin in enumerate(net.layers()):
                    lp = LinP(lin, pad_rows=256 if (fused and i == 3) else 0, need_wt=True, standalone=False)
                    o_dW = self.acc_size; self.acc_size += (lp.rows * lp.in_dim + 3) // 4 * 4
                    o_db = self.acc_size; self.acc_size += (lp.rows + 3) // 4 * 4
                    o_dv = self.grad_size; self.grad_size += (lp.out_dim * lp.in_dim + 3) // 4 * 4
                    o_dg = self.grad_size; self.grad_size += (lp.out_dim + 3) // 4 * 4 if lp.wn else 0
                    self.layout[id(lp)] = (o_dW, o_db, o_dv, o_dg)
                    ls.append(lp)
                self.lins[id(net)] = ls
        self.dev = next(iter(self.lins.values()))[0].W.device
        self._key = None

    def _build_tables(self):
        for key, nets in self.groups:
            lps = [lp for net in nets for lp in self.lins[id(net)]]
            arr = (MpWnDesc * len(lps))()
            row = 0
            for i, lp in enumerate(lps):
                o_dW, o_db, o_dv, o_dg = self.layout[id(lp)]
                arr[i] = MpWnDesc(lp.v.data_ptr(), lp.g.data_ptr() if lp.wn else None, lp.W.data_ptr(),
                                  lp.WT.data_ptr() if lp.WT is not None else None, o_dW, o_dv, o_dg, lp.out_dim, lp.in_dim, row, 0)
                row += lp.out_dim
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.tables[key] = (host.to(self.dev), len(lps), row, lps)

    def begin(self, keys=None):
        """start of a training forward: effective weights of the groups in `keys` (default: all).  The gradient accumulators are
        NOT part of this: they belong to a backward pass (start_backward), so that several forwards may be alive before their
        backwards run (loss = f(A) + f(B), a forward between forward and backward, a retained graph swept twice) without one
        graph's sweep accumulating into another's buffers."""
        L, st = hip.lib(), hip.stream()
        ptrs = []
        for ls in self.lins.values():
            for lp in ls:
                lp.read_params()
                ptrs.append(lp.v.data_ptr())
                ptrs.append(lp.g.data_ptr() if lp.wn else 0)
        key = tuple(ptrs)
        if key != self._key:
            self._key = key
            self._build_tables()
        for k, _ in self.groups:
            if keys is None or k in keys:
                tab, n, rows, _ = self.tables[k]
                _chk(L.mp_tr_wn_fwd_multi(_p(tab), n, rows, st), "mp_tr_wn_fwd_multi")
        self.acc = self.gbuf = None
        return self

    def start_backward(self):
        """start of ONE adjoint sweep: fresh flat buffers -- accumulators (one fill) and parameter gradients -- and every shared
        layer's dW / db / gradient views re-bound to them.  Fresh per sweep on purpose: autograd takes the gradient views over as
        param.grad without a copy, so a buffer must never be written by a later sweep."""
        self.acc = torch.zeros(self.acc_size, dtype=F32, device=self.dev)
        self.gbuf = torch.empty(self.grad_size, dtype=F32, device=self.dev)
        for ls in self.lins.values():
            for lp in ls:
                lp.bind(self.acc, *self.layout[id(lp)][:2], self.gbuf, *self.layout[id(lp)][2:])
        self._finished = set()
        return self

    def finish_group(self, key):
        """the weight-norm adjoint of one group's layers: accumulators -> parameter gradients (views of the flat buffer)"""
        if key in self._finished:
            raise RuntimeError(f"TrainState.finish_group({key!r}) twice in one backward sweep")
        self._finished.add(key)
        tab, n, rows, lps = self.tables[key]
        _chk(hip.lib().mp_tr_wn_bwd_multi(_p(tab), n, rows, _p(self.acc), _p(self.gbuf), hip.stream()), "mp_tr_wn_bwd_multi")


def _implicit_net_type():
    from .networks import ImplicitNet
    return ImplicitNet


def train_state(model):
    st = model.__dict__.get("_mp_train_state")
    if st is None:
        st = model.__dict__["_mp_train_state"] = TrainState(model)
    return st


def fused_sdf_supported(net):
    """the network shape csrc/tfuse.hip is specialised for: the shipped foreground ImplicitNet (confs/model/*.yaml)"""
    return (net.d_in == 3 and net.multires == 6 and list(net.skip_in) == [4] and net.num_layers - 1 == 9 and net.cond_dim == 69
            and list(net.dims[1:-1]) == [256] * 8 and net.dims[-1] == 257)


class FusedSDFState:
    """Per-network device state of the layer-fused training kernels (csrc/tfuse.hip): persistent effective weights, the split-bf16
    chunk stream, the bias table and the pointer tables mp_tf_sdf_pack reads."""

    def __init__(self, net, lins=None):
        """lins: the network's LinP list of a TrainState (weights resolved and accumulators bound by TrainState.begin); None:
        standalone layers owned by this object (unit tests, tools)"""
        self.net = net
        self.shared = lins is not None
        self.lins = lins if self.shared else [LinP(l, pad_rows=256 if i == 3 else 0) for i, l in enumerate(net.layers())]
        dev = self.lins[0].W.device
        arena, pack = C.c_longlong(0), C.c_longlong(0)
        _chk(hip.lib().mp_tf_sdf_sizes(1, C.byref(arena), C.byref(pack)), "mp_tf_sdf_sizes")
        self.arena_per_point = int(arena.value)
        self.wpack = torch.empty(int(pack.value), dtype=torch.uint8, device=dev)
        self.bias_all = torch.empty(9 * 288, dtype=F32, device=dev)
        self.b0 = torch.empty(256, dtype=F32, device=dev)
        self.wtab = _table([lw.W for lw in self.lins], dev)
        self._btab_key, self.btab = None, None

    def refresh(self, cond_vec):
        L, st = hip.lib(), hip.stream()
        net, lins = self.net, self.lins
        if not self.shared:
            for lw in lins:
                lw.refresh()
        lw0 = lins[0]
        _chk(L.mp_tr_hoist_fwd(_p(lw0.W), 256, lw0.in_dim, _p(lw0.b), net.embed_dim, net.cond_dim, _p(cond_vec), _p(self.b0), st),
             "mp_tr_hoist_fwd")
        bs = [self.b0] + [lw.b for lw in lins[1:]]
        key = tuple(b.data_ptr() for b in bs)
        if key != self._btab_key:
            self._btab_key, self.btab = key, _table(bs, self.b0.device)
        _chk(L.mp_tf_sdf_pack(_p(self.wtab), _p(self.btab), _p(self.wpack), _p(self.bias_all), st), "mp_tf_sdf_pack")
        return self


def fused_sdf_state(net, lins=None):
    key = "_mp_tfuse_shared" if lins is not None else "_mp_tfuse"
    st = net.__dict__.get(key)
    if st is None or (lins is not None and st.lins is not lins):
        st = net.__dict__[key] = FusedSDFState(net, lins)
    return st


class ImplicitTrainFused(ImplicitTrainRev):
    """ImplicitTrainRev's arithmetic (value sweep + gradient sweep, and the adjoint of both) on the LAYER-FUSED kernels of
    csrc/tfuse.hip: two launches instead of ~90 per person -- mp_tf_sdf_fwd (both forward sweeps) and mp_tf_sdf_bwd (the adjoint
    w.r.t. the activations) -- plus one weight-gradient contraction per layer over [dZ_l; V_l]^T [X_l; dT_l] (K = 2 P rows).
    Same interface: self.out [P][257], self.grad [P][3], backward(dZ_last, dgrad) -> d cond.  The adjoint of the input points
    (pose optimisation) is not produced here: TrainGraph takes ImplicitTrainRev when it is needed."""

    def __init__(self, net, x, cond_vec, lins=None, p_cap=None, cap_bytes=6 << 30):
        """p_cap: the largest P this caller can ever pass (all rays hit the body): the stash is then sized for it, i.e. the SAME
        allocation every iteration (_big_empty) -- when that is at most cap_bytes (the caller divides its arena budget by the
        number of persons whose stashes are alive together: a crowded scene must not hold n_persons x the all-rays-hit arena)"""
        L, st = hip.lib(), hip.stream()
        assert fused_sdf_supported(net)
        self.net, self.x, self.cond = net, x, cond_vec
        self.P = P = x.shape[0]
        self.E = E = net.embed_dim
        dev = x.device
        self.fs = fs = fused_sdf_state(net, lins).refresh(cond_vec)
        self.lins, self.nl = fs.lins, len(fs.lins)
        arena = C.c_longlong(0)
        _chk(L.mp_tf_sdf_sizes(P, C.byref(arena), None), "mp_tf_sdf_sizes")
        cap = None
        if p_cap is not None and p_cap >= P:
            capv = C.c_longlong(0)
            _chk(L.mp_tf_sdf_sizes(int(p_cap), C.byref(capv), None), "mp_tf_sdf_sizes")
            cap = int(capv.value)
        self.arena = _big_empty(int(arena.value), dev, cap=cap, cap_bytes=cap_bytes)
        R1 = 256 * (P + 1)                               # every [P][256] stash tensor carries one pad row (csrc/tfuse.hip)
        self.o_dZ = lambda l: l * R1
        self.o_V = lambda l: (8 + l) * R1
        self.o_X = lambda l: (15 + l) * R1
        self.o_dT = lambda l: (23 + l) * R1
        self.o_IN, self.o_dG, self.o_G = 46 * R1, 46 * R1 + E * P, 46 * R1 + 2 * E * P
        r2 = 1.0 / math.sqrt(2.0)
        _chk(L.mp_tr_pe(_p(x), 3, P, net.multires, 0, C.c_float(1.0), off(self.arena, self.o_IN), E, 0, st), "mp_tr_pe")
        self.feat = torch.empty(P + 1, 256, dtype=F32, device=dev)  # columns 1.. of the reference's output (+ the pad row)
        self.sdf = torch.empty(P + 1, dtype=F32, device=dev)        # column 0
        self.w8 = fs.lins[8].W                          # row 0 = the sdf row of the last layer
        _chk(L.mp_tf_sdf_fwd(_p(fs.wpack), _p(fs.bias_all), _p(self.w8), _p(self.arena), P, _p(self.feat), _p(self.sdf), st),
             "mp_tf_sdf_fwd")
        # the skip connection re-injects the Fourier features into layer 4's input (times 1/sqrt 2): columns 217.. of X_4
        _chk(L.mp_tr_copy_cols(off(self.arena, self.o_IN), E, 0, off(self.arena, self.o_X(4)), 256, 256 - E, P, E, C.c_float(r2), 0,
                               st), "mp_tr_copy_cols")
        self.grad = torch.empty(P, 3, dtype=F32, device=dev)
        _chk(L.mp_tr_pe_grad_fwd(_p(x), P, net.multires, off(self.arena, self.o_G), E, _p(self.grad), st), "mp_tr_pe_grad_fwd")

    @staticmethod
    def _launch(groups):
        if TRAIN_PRECISION == "bf16x3":
            gemm_tn_grouped(groups)
        else:                                   # exact-fp32 cross-check: one launch per contraction
            for g in groups:
                gemm_tn(C.c_void_p(g.A), g.lda, C.c_void_p(g.B), g.ldb, C.c_void_p(g.C), g.ldc, g.M, g.N, g.K,
                        C.c_void_p(g.colsum) if g.colsum else None, g.colsum_rows)

    @property
    def out(self):
        """the reference's [P][257] layout (column 0 = sdf), assembled on demand (tests; the trainer reads feat / sdf)"""
        return torch.cat([self.sdf[:self.P, None], self.feat[:self.P]], 1)

    def backward(self, dfeat, dsdf, dgrad, want_dx=False, tn_groups=None):
        """dfeat [P][256], dsdf [P], dgrad [P][3] -> dW / db of every layer; returns d cond.  tn_groups (a list): the aligned
        weight-gradient contractions are appended to it instead of launched (the caller launches them grouped)"""
        assert not want_dx, "the fused SDF kernels do not produce the adjoint of the input points"
        L, st = hip.lib(), hip.stream()
        net, P, E, lins, fs, A = self.net, self.P, self.E, self.lins, self.fs, self.arena
        dev = dfeat.device
        r2 = 1.0 / math.sqrt(2.0)
        dG = off(A, self.o_dG)
        _chk(L.mp_tr_pe_grad_bwd(_p(self.x), P, net.multires, _p(dgrad), off(A, self.o_G), E, dG, E, None, st), "mp_tr_pe_grad_bwd")
        lw8 = lins[8]                                   # the sdf row's gradient goes straight into row 0 of dW_8 / db_8
        _chk(L.mp_tf_sdf_bwd(_p(fs.wpack), _p(self.w8), _p(A), P, _p(dfeat), _p(dsdf), _p(lw8.dW), _p(lw8.db), st), "mp_tf_sdf_bwd")
        _chk(L.mp_tr_copy_cols(dG, E, 0, off(A, self.o_dT(4)), 256, 256 - E, P, E, C.c_float(r2), 0, st), "mp_tr_copy_cols")
        # weight gradients  dW_l += dZ_l^T X_l (value sweep; bias gradient = its column sums) + V_l^T dT_l (gradient sweep)
        lw0 = lins[0]
        db0 = _zeros(256, device=dev)        # layer 0's bias gradient of THIS evaluation (see ImplicitTrain.backward)
        gemm_tn(off(A, self.o_dZ(0)), 256, off(A, self.o_IN), E, _p(lw0.dW), lw0.in_dim, 256, E, P, _p(db0), P)
        lw0.db.add_(db0)
        gemm_tn(off(A, self.o_V(0)), 256, dG, E, _p(lw0.dW), lw0.in_dim, 256, E, P)
        groups = tn_groups if tn_groups is not None else []
        for l in range(1, 8):
            lw = lins[l]                                # (layer 3: 217 rows contracted as 256, see LinP)
            rows = lw.dW_full.shape[0]
            groups.append(tn_group(off(A, self.o_dZ(l)), 256, off(A, self.o_X(l)), 256, _p(lw.dW_full), lw.in_dim, rows, lw.in_dim, P,
                                   _p(lw.db_full), P))
            groups.append(tn_group(off(A, self.o_V(l)), 256, off(A, self.o_dT(l)), 256, _p(lw.dW_full), lw.in_dim, rows, lw.in_dim, P))
        groups.append(tn_group(_p(dfeat), 256, off(A, self.o_X(8)), 256, off(lw8.dW, 256), 256, 256, 256, P, off(lw8.db, 1), P))
        if tn_groups is None:
            self._launch(groups)
        _chk(L.mp_tr_hoist_bwd(_p(db0), 256, lw0.in_dim, E, net.cond_dim, _p(self.cond), _p(lw0.dW), st), "mp_tr_hoist_bwd")
        dcond = _zeros(net.cond_dim, device=dev)
        gemm_tn(_p(db0), 1, off(lw0.W, E), lw0.in_dim, _p(dcond), net.cond_dim, 1, net.cond_dim, 256)
        self.dx = None
        return dcond


def fused_bg_supported(net):
    """the network shape the fused background kernels (csrc/tfuse.hip k_tf_bg_*) are specialised for: the shipped NeRF++ net"""
    return (net.d_in == 4 and net.multires == 10 and list(net.skip_in) == [4] and net.num_layers - 1 == 9 and net.cond_dim == 32
            and list(net.dims[1:-1]) == [256] * 8 and net.dims[-1] == 257 and not hasattr(net.lin0, "weight_g"))


class FusedBGState:
    """chunk stream, bias table and pointer tables of the fused background kernels for one ImplicitNet (LinP layers shared with the
    TrainState, or its own)"""

    def __init__(self, net, lins=None):
        self.net = net
        self.shared = lins is not None
        self.lins = lins if self.shared else [LinP(l, pad_rows=256 if i == 3 else 0) for i, l in enumerate(net.layers())]
        dev = self.lins[0].W.device
        arena, pack = C.c_longlong(0), C.c_longlong(0)
        _chk(hip.lib().mp_tf_bg_sizes(1, C.byref(arena), C.byref(pack)), "mp_tf_bg_sizes")
        self.wpack = torch.empty(int(pack.value), dtype=torch.uint8, device=dev)
        self.bias_all = torch.empty(9 * 288, dtype=F32, device=dev)
        self.b0 = torch.empty(256, dtype=F32, device=dev)
        self.wtab = _table([lw.W for lw in self.lins], dev)
        self._btab_key, self.btab = None, None

    def refresh(self, code):
        L, st = hip.lib(), hip.stream()
        net, lins = self.net, self.lins
        if not self.shared:
            for lw in lins:
                lw.refresh()
        lw0 = lins[0]
        _chk(L.mp_tr_hoist_fwd(_p(lw0.W), 256, lw0.in_dim, _p(lw0.b), net.embed_dim, net.cond_dim, _p(code), _p(self.b0), st),
             "mp_tr_hoist_fwd")
        bs = [self.b0] + [lw.b for lw in lins[1:]]
        key = tuple(b.data_ptr() for b in bs)
        if key != self._btab_key:
            self._btab_key, self.btab = key, _table(bs, self.b0.device)
        _chk(L.mp_tf_bg_pack(_p(self.wtab), _p(self.btab), _p(self.wpack), _p(self.bias_all), st), "mp_tf_bg_pack")
        return self


def fused_bg_state(net, lins=None):
    key = "_mp_tfuse_shared" if lins is not None else "_mp_tfuse"
    st = net.__dict__.get(key)
    if st is None or (lins is not None and st.lins is not lins):
        st = net.__dict__[key] = FusedBGState(net, lins)
    return st


class ImplicitTrainFusedBG:
    """ImplicitTrain(fwd=False)'s arithmetic for the background ImplicitNet on the layer-fused kernels (csrc/tfuse.hip
    mp_tf_bg_fwd / mp_tf_bg_bwd): the nine layers in one launch each way, the weight gradients as one contraction per layer.
    x [P][4] (the inverted-sphere points), code (32,) the frame's latent row.  self.sdf [P+1], self.feat [P+1][256]."""

    def __init__(self, net, x, code, lins=None):
        L, st = hip.lib(), hip.stream()
        assert fused_bg_supported(net)
        self.net, self.x, self.cond = net, x, code
        self.P = P = x.shape[0]
        self.E = E = net.embed_dim                       # 84
        dev = x.device
        self.fs = fs = fused_bg_state(net, lins).refresh(code)
        self.lins = fs.lins
        arena = C.c_longlong(0)
        _chk(L.mp_tf_bg_sizes(P, C.byref(arena), None), "mp_tf_bg_sizes")
        self.arena = A = _big_empty(int(arena.value), dev, grain=1 << 22)
        R1 = 256 * (P + 1)
        self.o_dZ = lambda l: l * R1
        self.o_X = lambda l: (7 + l) * R1
        self.o_IN = 16 * R1
        _chk(L.mp_tr_pe(_p(x), 4, P, net.multires, 0, C.c_float(1.0), off(A, self.o_IN), E, 0, st), "mp_tr_pe")
        self.feat = torch.empty(P + 1, 256, dtype=F32, device=dev)
        self.sdf = torch.empty(P + 1, dtype=F32, device=dev)
        _chk(L.mp_tf_bg_fwd(_p(fs.wpack), _p(fs.bias_all), _p(A), P, _p(self.feat), _p(self.sdf), st), "mp_tf_bg_fwd")
        # the skip connection re-injects the Fourier features into layer 4's input (times 1/sqrt 2): columns 172.. of X_4
        _chk(L.mp_tr_copy_cols(off(A, self.o_IN), E, 0, off(A, self.o_X(4)), 256, 256 - E, P, E, C.c_float(1.0 / math.sqrt(2.0)), 0,
                               st), "mp_tr_copy_cols")

    def backward(self, dfeat, dsdf):
        """dfeat [P][256], dsdf [P] -> dW / db of every layer; returns d code (the hoisted conditioning's adjoint)"""
        L, st = hip.lib(), hip.stream()
        net, P, E, lins, fs, A = self.net, self.P, self.E, self.lins, self.fs, self.arena
        lw8 = lins[8]
        _chk(L.mp_tf_bg_bwd(_p(fs.wpack), _p(lw8.W), _p(A), P, _p(dfeat), _p(dsdf), _p(lw8.dW), _p(lw8.db), st), "mp_tf_bg_bwd")
        lw0 = lins[0]
        db0 = _zeros(256, device=dfeat.device)      # layer 0's bias gradient of THIS evaluation (see ImplicitTrain.backward)
        gemm_tn(off(A, self.o_dZ(0)), 256, off(A, self.o_IN), E, _p(lw0.dW), lw0.in_dim, 256, E, P, _p(db0), P)
        lw0.db.add_(db0)
        groups = []
        for l in range(1, 8):
            lw = lins[l]                                # (layer 3: 172 rows contracted as 256, see LinP)
            rows = lw.dW_full.shape[0]
            groups.append(tn_group(off(A, self.o_dZ(l)), 256, off(A, self.o_X(l)), 256, _p(lw.dW_full), lw.in_dim, rows, lw.in_dim, P,
                                   _p(lw.db_full), P))
        groups.append(tn_group(_p(dfeat), 256, off(A, self.o_X(8)), 256, off(lw8.dW, 256), 256, 256, 256, P, off(lw8.db, 1), P))
        ImplicitTrainFused._launch(groups)
        _chk(L.mp_tr_hoist_bwd(_p(db0), 256, lw0.in_dim, E, net.cond_dim, _p(self.cond), _p(lw0.dW), st), "mp_tr_hoist_bwd")
        dcode = _zeros(net.cond_dim, device=dfeat.device)
        gemm_tn(_p(db0), 1, off(lw0.W, E), lw0.in_dim, _p(dcode), net.cond_dim, 1, net.cond_dim, 256)
        return dcode

    def params(self):
        return [p for lw in self.lins for p in lw.params()]

    def param_grads(self):
        return [g for lw in self.lins for g in lw.param_grads()]


class RenderTrain:
    """RenderingNet (networks.py:263-312): mode 'pose_no_view' (inputs XA = [x_c, n] (6), feat) or 'nerf_frame_encoding'
    (XA = PE_4(view) (27), feat).  feat is read in place from the SDF net's last layer (ld 257, column 1..)."""

    def __init__(self, net, XA, feat_ptr, feat_ld, n, cond_vec, lins=None):
        L = hip.lib()
        self.net, self.n = net, n
        dev = XA.device
        self.lins = lins if lins is not None else [LinW(l) for l in net.layers()]
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

    def backward(self, drgb, dXA, dfeat_ptr, dfeat_ld, feat_accumulate=True):
        """drgb [n][3] -> dW/db, dXA [n][na] (written), d feat (+= into dfeat_ptr, or written with feat_accumulate=False);
        returns d hoisted-vector"""
        L = hip.lib()
        n, dev = self.n, drgb.device
        nl = len(self.lins)
        dZ = torch.empty(n, 3, dtype=F32, device=dev)
        _chk(L.mp_tr_sigmoid_bwd(_p(self.rgb), _p(drgb), n * 3, _p(dZ), hip.stream()), "mp_tr_sigmoid_bwd")
        for l in range(nl - 1, 0, -1):
            lw, Hp = self.lins[l], self.H[l - 1]
            pout = self.lins[l - 1].out_dim
            gemm_tn(_p(dZ), lw.out_dim, _p(Hp), pout, _p(lw.dW), lw.in_dim, lw.out_dim, lw.in_dim, n, _p(lw.db), n)
            dH = torch.empty(n, pout, dtype=F32, device=dev)
            gemm_nt(_p(dZ), lw.out_dim, _p(lw.WT), lw.out_dim, _p(dH), pout, n, pout, lw.out_dim)
            dZp = torch.empty(n, pout, dtype=F32, device=dev)
            _chk(L.mp_tr_relu_bwd(_p(Hp), pout, n, pout, _p(dH), pout, _p(dZp), pout, hip.stream()), "mp_tr_relu_bwd")
            dZ = dZp
        lw0 = self.lins[0]
        o0 = lw0.out_dim
        db0 = _zeros(o0, device=dev)                  # this evaluation's own bias gradient (see ImplicitTrain.backward)
        gemm_tn(_p(dZ), o0, _p(self.XA), self.na, _p(lw0.dW), lw0.in_dim, o0, self.na, n, _p(db0), n)
        lw0.db.add_(db0)
        gemm_tn(_p(dZ), o0, self.feat_ptr, self.feat_ld, off(lw0.dW, self.c_feat), lw0.in_dim, o0, 256, n)
        _chk(L.mp_tr_hoist_bwd(_p(db0), o0, lw0.in_dim, self.c_h0, self.n_h, _p(self.hvec), _p(lw0.dW), hip.stream()),
             "mp_tr_hoist_bwd")
        dh = _zeros(self.n_h, device=dev)
        gemm_tn(_p(db0), 1, off(lw0.W, self.c_h0), lw0.in_dim, _p(dh), self.n_h, 1, self.n_h, o0)
        # data gradients
        gemm_nt(_p(dZ), o0, _p(lw0.WT), o0, _p(dXA), self.na, n, self.na, o0)
        gemm_nt(_p(dZ), o0, off(lw0.WT, self.c_feat * o0), o0, dfeat_ptr, dfeat_ld, n, 256, o0, None, 0, accumulate=feat_accumulate)
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


def fused_col_supported(net):
    """the RenderingNet shape the fused colour kernels of csrc/tfuse.hip are specialised for (the shipped foreground net)"""
    return net.mode == "pose_no_view" and list(net.dims) == [270, 256, 256, 256, 256, 3]


class FusedColState:
    """chunk stream, bias table and pointer tables of the fused colour kernels for one RenderingNet (its LinP layers shared with
    the TrainState, or its own)"""

    def __init__(self, net, lins=None):
        self.net = net
        self.shared = lins is not None
        self.lins = lins if self.shared else [LinP(l) for l in net.layers()]
        dev = self.lins[0].W.device
        stash, pack = C.c_longlong(0), C.c_longlong(0)
        _chk(hip.lib().mp_tf_col_sizes(1, C.byref(stash), C.byref(pack)), "mp_tf_col_sizes")
        self.stash_per_point = int(stash.value)
        self.wpack = torch.empty(int(pack.value), dtype=torch.uint8, device=dev)
        self.bias_all = torch.empty(5 * 288, dtype=F32, device=dev)
        self.b0 = torch.empty(256, dtype=F32, device=dev)
        self.pose8 = torch.empty(8, dtype=F32, device=dev)
        self.wtab = _table([lw.W for lw in self.lins], dev)
        self._btab_key, self.btab = None, None

    def refresh(self, cond_vec):
        L, st = hip.lib(), hip.stream()
        net, lins = self.net, self.lins
        if not self.shared:
            for lw in lins:
                lw.refresh()
        lp = net.lin_pose
        self.lp_w, self.lp_b = lp.weight.detach().contiguous(), lp.bias.detach().contiguous()
        _chk(L.mp_tr_hoist_fwd(_p(self.lp_w), 8, 69, _p(self.lp_b), 0, 69, _p(cond_vec), _p(self.pose8), st), "mp_tr_hoist_fwd")
        lw0 = lins[0]
        _chk(L.mp_tr_hoist_fwd(_p(lw0.W), 256, lw0.in_dim, _p(lw0.b), 6, 8, _p(self.pose8), _p(self.b0), st), "mp_tr_hoist_fwd")
        bs = [self.b0] + [lw.b for lw in lins[1:]]
        key = tuple(b.data_ptr() for b in bs)
        if key != self._btab_key:
            self._btab_key, self.btab = key, _table(bs, self.b0.device)
        _chk(L.mp_tf_col_pack(_p(self.wtab), _p(self.btab), _p(self.wpack), _p(self.bias_all), st), "mp_tf_col_pack")
        return self


def fused_col_state(net, lins=None):
    key = "_mp_tfuse_shared" if lins is not None else "_mp_tfuse"
    st = net.__dict__.get(key)
    if st is None or (lins is not None and st.lins is not lins):
        st = net.__dict__[key] = FusedColState(net, lins)
    return st


class RenderTrainFused:
    """RenderTrain's arithmetic for the foreground colour net on the layer-fused kernels (csrc/tfuse.hip: mp_tf_col_fwd /
    mp_tf_col_bwd): the five layers in one launch each way, the weight gradients as one contraction per layer.  feat [>= n][256]
    (row stride 256: the fused SDF net's feature rows), XA [n][6]."""

    def __init__(self, net, XA, feat, n, cond_vec, lins=None):
        L, st = hip.lib(), hip.stream()
        assert fused_col_supported(net) and feat.shape[1] == 256 and feat.is_contiguous()
        self.net, self.n, self.XA, self.feat, self.cond = net, n, XA, feat, cond_vec
        dev = XA.device
        self.cs = cs = fused_col_state(net, lins).refresh(cond_vec)
        self.lins = cs.lins
        stash = C.c_longlong(0)
        _chk(L.mp_tf_col_sizes(n, C.byref(stash), None), "mp_tf_col_sizes")
        self.stash = _big_empty(int(stash.value), dev)
        self.rgb = torch.empty(n, 3, dtype=F32, device=dev)
        _chk(L.mp_tf_col_fwd(_p(cs.wpack), _p(cs.bias_all), _p(self.stash), _p(feat), _p(XA), n, _p(self.rgb), st), "mp_tf_col_fwd")

    def backward(self, drgb, dXA, dfeat, tn_groups=None):
        """drgb [n][3] -> dW / db, dXA [n][6] and rows [0, n) of dfeat [.][256] (both written); returns d pose-embedding.
        tn_groups: see ImplicitTrainFused.backward"""
        L, st = hip.lib(), hip.stream()
        n, lins, cs, S = self.n, self.lins, self.cs, self.stash
        dev = drgb.device
        NL = 256 * (n + 1)                               # one pad row per stash tensor
        dz4 = torch.empty(n, 3, dtype=F32, device=dev)
        _chk(L.mp_tf_col_bwd(_p(cs.wpack), _p(S), _p(lins[4].W), _p(self.rgb), _p(drgb), n, _p(dfeat), _p(dXA), _p(dz4), st),
             "mp_tf_col_bwd")
        H = lambda l: off(S, l * NL)
        dZ = lambda l: off(S, (4 + l) * NL)
        lw0 = lins[0]
        db0 = _zeros(256, device=dev)                 # this evaluation's own bias gradient (see ImplicitTrain.backward)
        gemm_tn(dZ(0), 256, _p(self.XA), 6, _p(lw0.dW), lw0.in_dim, 256, 6, n, _p(db0), n)
        lw0.db.add_(db0)
        groups = tn_groups if tn_groups is not None else []
        groups.append(tn_group(dZ(0), 256, _p(self.feat), 256, off(lw0.dW, 14), lw0.in_dim, 256, 256, n))
        for l in range(1, 4):
            lw = lins[l]
            groups.append(tn_group(dZ(l), 256, H(l - 1), 256, _p(lw.dW), 256, 256, 256, n, _p(lw.db), n))
        if tn_groups is None:
            ImplicitTrainFused._launch(groups)
        lw4 = lins[4]
        gemm_tn(_p(dz4), 3, H(3), 256, _p(lw4.dW), 256, 3, 256, n, _p(lw4.db), n)
        # the hoisted pose embedding: dW_0[:, 6:14] += db_0 (x) pose8 ; d pose8 = W_0[:, 6:14]^T db_0 ; then lin_pose's own gradients
        _chk(L.mp_tr_hoist_bwd(_p(db0), 256, lw0.in_dim, 6, 8, _p(cs.pose8), _p(lw0.dW), st), "mp_tr_hoist_bwd")
        dh = _zeros(8, device=dev)
        gemm_tn(_p(db0), 1, off(lw0.W, 6), lw0.in_dim, _p(dh), 8, 1, 8, 256)
        dlp_w = _zeros(8, 69, device=dev)
        _chk(L.mp_tr_hoist_bwd(_p(dh), 8, 69, 0, 69, _p(self.cond), _p(dlp_w), st), "mp_tr_hoist_bwd")
        self.extra_grads = [dlp_w, dh]
        self.lp_w = cs.lp_w
        return dh

    def params(self):
        return [self.net.lin_pose.weight, self.net.lin_pose.bias] + [p for lw in self.lins for p in lw.params()]

    def param_grads(self):
        return list(self.extra_grads) + [g for lw in self.lins for g in lw.param_grads()]


# ======================================================================================================================
# Training-mode Multiply.forward (multiply.py:174-588, `self.training` branches) as ONE autograd node
# ======================================================================================================================
N_EIKONAL = 512          # multiply.py:324
ARENA_BUDGET_BYTES = int(__import__("os").environ.get("MP_TRAIN_ARENA_GB", "24")) << 30   # fixed-size SDF stashes of one iteration, all persons
# 'fused' (ImplicitTrainFused: layer-fused kernels, default) | 'reverse' (ImplicitTrainRev, layer by layer) | 'forward'
SDF_TRAIN_MODE = __import__("os").environ.get("MP_SDF_TRAIN_MODE", "fused")
# background ImplicitNet: 'fused' (ImplicitTrainFusedBG, default) | 'layerwise' (ImplicitTrain: the cross-check)
BG_TRAIN_MODE = __import__("os").environ.get("MP_BG_TRAIN_MODE", "fused")


ZERO_POSE_SAMPLES = 2000          # multiply.py:363
SMPL_SURFACE_THRESHOLD = 0.02     # multiply.py:358
SURFACE_EXCLUDED_PARTS = ("head", "rightHand", "leftHand", "rightFoot", "leftFoot", "leftHandIndex1", "rightHandIndex1")   # multiply.py:340-343


def _table(ts, dev):
    return hip.device_ints([t.data_ptr() for t in ts], dev)      # (no host wait: hip._PinnedInts)


def surface_sampling_weights(model, n_verts, dev):
    """multiply.py:339-345: SMPL vertices the surface regulariser samples from -- all but head, hands and feet (cached)"""
    w = model.__dict__.get("_mp_surface_weights")
    if w is None or w.shape[0] != n_verts or w.device != torch.device(dev):
        part = getattr(model, "smpl_vertex_part", None)
        if part is None:
            raise FileNotFoundError("smpl_surface_weight > 0 needs the SMPL vertex segmentation (the reference reads "
                                    "./outputs/smpl_vert_segmentation.json, multiply.py:113, an asset it does not ship): set "
                                    "opt.smpl_vert_segmentation_path or assign model.smpl_vertex_part = {part name: [vertex ids]}")
        w = torch.ones(n_verts)
        w[[i for k in SURFACE_EXCLUDED_PARTS for i in part[k]]] = 0
        w = model.__dict__["_mp_surface_weights"] = w.to(dev)
    return w


def _random_prefixes(sizes, k, dev, gen):
    """row i = the first k entries of a uniform random permutation of range(sizes[i]) (what torch.randperm(n)[:k] draws), all
    rows in three launches: the k smallest of n independent uniform keys, in the order of their keys, ARE such a prefix.
    [One torch.randperm per row is ~7 launches each: 84 of an iteration's ~570.]"""
    n = max(sizes)
    assert k <= min(sizes), "a permutation prefix longer than the permutation"
    keys = torch.rand(len(sizes), n, device=dev, generator=gen)
    if min(sizes) < n:
        lim = hip.device_ints(sizes, dev).unsqueeze(1)
        keys = torch.where(torch.arange(n, device=dev).unsqueeze(0) < lim, keys, keys.new_full((), 2.0))
    return torch.topk(keys, k, dim=1, largest=False, sorted=True).indices


def make_draws(model, cx, gen=None):
    """The randomness one training forward consumes (ray_sampler.py:38,171,202; multiply.py:325; sampler.py:100):
    per person  t_rand [R_p,NE], u_final [R_p,N], extra_idx [max_iters,N_extra], eik_idx [512], eik_noise [512,3];
    shared      bg_rand [R,N_bg].  Drawn on the device with torch's generator; the permutation prefixes (the reference's
    torch.randperm(n)[:k]) come from _random_prefixes, all persons at once."""
    rs, dev = model.ray_sampler, cx["dev"]
    NE, NS, NX = rs.N_samples_eval, rs.N_samples, rs.N_samples_extra
    kw = dict(device=dev, generator=gen)
    draws = {"person": {}}
    persons = list(cx["persons"])
    T = rs.max_total_iters
    extra = _random_prefixes([NE * k for k in range(1, T + 1)] * len(persons), NX, dev, gen).to(torch.int32).reshape(len(persons), T, NX)
    nvs = [model.smpl_server_list[p].verts_c.reshape(-1, 3).shape[0] for p in persons]
    eik = _random_prefixes(nvs, N_EIKONAL, dev, gen)
    for n, p in enumerate(persons):
        Rp = max(int(cx["n_hit"][n]), 1)
        draws["person"][p] = dict(
            t_rand=torch.rand(Rp, NE, **kw), u_final=torch.rand(Rp, NS, **kw), extra_idx=extra[n].contiguous(),
            eik_idx=eik[n], eik_noise=torch.randn(N_EIKONAL, 3, **kw))
    draws["bg_rand"] = torch.rand(cx["R"], rs.N_samples_inverse_sphere, **kw)
    # the two regularisers of multiply.py:336-394 (weight 0 in the shipped configs): their vertex draws
    if model.smpl_surface_weight > 0:
        for n, p in enumerate(persons):                               # idx_weight.multinomial(num_pixels, replacement=True)
            draws["person"][p]["surf_idx"] = torch.multinomial(surface_sampling_weights(model, nvs[n], dev), cx["R"],
                                                               replacement=True, generator=gen)
    if model.zero_pose_weight > 0:
        nv_all = [v.shape[1] for v in model.mesh_v_cano_list]
        zp = _random_prefixes(nv_all * len(persons), min(ZERO_POSE_SAMPLES, min(nv_all)), dev, gen).reshape(len(persons), len(nv_all), -1)
        draws["zp_idx"] = {(q, p): zp[n, p] for n, q in enumerate(persons) for p in range(len(nv_all))}
    return draws


class TrainGraph:
    """Everything one training forward keeps for its backward."""

    def __init__(self, model, cx, input, cond_zero, draws, surface_flags=False, pose_grad=False, shard=None, ts=None):
        """shard = (world, rank): person-sharded training: cx holds only this rank's persons
        {p : p % world == rank}; their per-sample rows are all-gathered once in the forward, every rank composites all rays
        (512 of them: cheaper than a second exchange in the backward), the background branch is ray-sliced."""
        self.model, self.cx, self.input, self.cond_zero, self.draws = model, cx, input, cond_zero, draws
        self.surface_flags, self.pose_grad, self.shard = surface_flags, pose_grad, shard
        self.ts = ts if ts is not None else train_state(model).begin()     # shared layers: weights resolved, accumulators zeroed

    # ---- forward ------------------------------------------------------------------------------------------------
    def run(self):
        m, cx, L = self.model, self.cx, hip.lib()
        dev, R, dirs, pose, beta = cx["dev"], cx["R"], cx["dirs"], cx["pose"], cx["beta"]
        st = hip.stream()
        f32 = dict(dtype=F32, device=dev)
        rs = m.ray_sampler
        NZ = rs.N_samples + rs.N_samples_extra + 2
        S = NZ - 1
        persons = cx["persons"]
        self.fg = {}
        z_l, sdf_l, rgb_l, nrm_l, inv_l = [], [], [], [], []
        if self.cond_zero:                                            # multiply.py:271-273
            for p in persons:
                cx["per"][p]["cond"] = torch.zeros_like(cx["per"][p]["cond"])
        # the samplers of all persons advance together (Multiply._sample_persons): in ray-sharded data-parallel training the
        # convergence vote is then ONE collective per sampler iteration for all persons
        todo = [p for p in persons if self.draws["person"][p].get("z_given") is None]
        m.__dict__["_mp_in_train_graph"] = True       # (the near-fp32 sampler mode may share this iteration's resolved weights)
        try:
            sampled = m._sample_persons(cx, {p: self.draws["person"][p] for p in todo}, persons=todo) if todo else {}
        finally:
            m.__dict__["_mp_in_train_graph"] = False
        for n, p in enumerate(persons):
            pp = cx["per"][p]
            dr = self.draws["person"][p]
            Rp = max(int(cx["n_hit"][n]), 1)
            imp, ren, dfm = m.foreground_implicit_network_list[p], m.foreground_rendering_network_list[p], m.deformer_list[p]
            server = m.smpl_server_list[p]
            skin_w = server.tables.lbs_weights
            if dr.get("z_given") is not None:
                # depths handed in by the caller instead of sampled here (the sampler runs without gradients in the reference,
                # ray_sampler.py:86-87): lets a test drive everything downstream from an INDEPENDENT sampler's depths
                zfinal, iters, wcount = dr["z_given"].to(dev).float().contiguous(), None, None
                assert zfinal.shape == (Rp, NZ), f"z_given of person {p}: {tuple(zfinal.shape)} != {(Rp, NZ)}"
            else:
                zfinal, iters, wcount = sampled[p]
            npts = Rp * S
            E = N_EIKONAL
            Pt = npts + E
            X = torch.empty(Pt, 3, **f32)                             # canonical points: samples, then eikonal points
            # the nearest POSED vertex of every sample: the pose adjoint needs it, and it seeds the canonical nearest-vertex search
            # of the Jacobian (csrc/geom.hip k_warp_jacobian: its canonical distance is a tight, exact search radius -- the
            # unseeded search opens every cluster: 228 us instead of ~30 per person)
            nn_posed = torch.empty(npts, dtype=torch.int32, device=dev)
            nn_cano = torch.empty(npts, dtype=torch.int32, device=dev) if self.pose_grad else None
            # (a training batch's rays are random pixels: the warp first groups the samples by their nearest vertex cluster)
            bin_work = torch.empty(int(L.mp_warp_bin_work_bytes(npts)), dtype=torch.uint8, device=dev)
            _chk(L.mp_warp_inverse_shade(_p(dirs), _p(pose), _p(pp["hit_index"]), _p(pp["count"]), _p(zfinal), NZ, S, Rp,
                                         _p(pp["vsorted"]), _p(pp["cbound"]), _p(pp["btab"]), 0, _p(beta),
                                         _p(X), None, None, None, None, None, _p(nn_posed), _p(bin_work), st), "mp_warp_inverse_shade")
            jinv = torch.empty(npts, 9, **f32)
            _chk(L.mp_warp_jacobian(_p(X), None, None, 0, 0, npts, _p(dfm.vsorted_c), _p(dfm.cbound_c), _p(pp["btab"]),
                                    _p(jinv), _p(nn_cano), _p(nn_posed), _p(dfm.verts_c_flat), st), "mp_warp_jacobian")
            flags = None
            if self.surface_flags:        # multiply.py:311-315: in / off-surface rays w.r.t. the current canonical mesh
                fv = m.mesh_face_vertices_list[p].detach().reshape(-1, 9).to(dev).float().contiguous()
                sd = torch.empty(npts, **f32)
                _chk(L.mp_mesh_signed_distance(_p(X), npts, _p(fv), fv.shape[0], _p(sd), st), "mp_mesh_signed_distance")
                off_p = torch.empty(Rp, dtype=torch.uint8, device=dev); in_p = torch.empty(Rp, dtype=torch.uint8, device=dev)
                _chk(L.mp_mesh_ray_flags(_p(sd), Rp, S, C.c_float(m.threshold), _p(off_p), _p(in_p), st), "mp_mesh_ray_flags")
                flags = (off_p.bool(), in_p.bool(), sd)
            # eikonal points near the canonical surface (multiply.py:322-327, sampler.py:84-108 with global_ratio 0)
            vc = server.verts_c.reshape(-1, 3)
            X[npts:] = vc[dr["eik_idx"]] + dr["eik_noise"] * m.sampler.local_sigma
            mode = SDF_TRAIN_MODE
            if mode == "fused" and (self.pose_grad or TRAIN_PRECISION != "bf16x3" or not fused_sdf_supported(imp)):
                mode = "reverse"      # the fused kernels: split-bf16 arithmetic, the shipped network shape, no d x_c
            rev = mode != "forward"
            li, lr = self.ts.lins[id(imp)], self.ts.lins[id(ren)]
            if mode == "fused":
                # all persons' stashes live from forward to backward: the fixed all-rays-hit size only while the sum stays in budget
                it = ImplicitTrainFused(imp, X, pp["cond"], lins=li, p_cap=R * S + E,
                                        cap_bytes=min(6 << 30, ARENA_BUDGET_BYTES // max(len(persons), 1)))
            else:
                it = ImplicitTrainRev(imp, X, pp["cond"], lins=li) if rev else ImplicitTrain(imp, X, pp["cond"], fwd=True, lins=li)
            gptr = _p(it.grad) if rev else None
            fusedp = mode == "fused"     # the fused kernels hand the last layer over as feat [Pt][256] + sdf [Pt], not [Pt][257]
            XA = torch.empty(npts, 6, **f32); nrm = torch.empty(npts, 3, **f32)
            sdf = it.sdf[:npts] if fusedp else torch.empty(npts, **f32)
            z8 = None if fusedp else _p(it.out)
            _chk(L.mp_tr_shade_in_fwd(z8, Pt, npts, _p(X), _p(jinv), _p(XA), _p(nrm), None if fusedp else _p(sdf), gptr, st),
                 "mp_tr_shade_in_fwd")
            gth = torch.empty(E, 3, **f32)
            _chk(L.mp_tr_eik_fwd(z8, Pt, npts, E, _p(gth), gptr, st), "mp_tr_eik_fwd")
            if fusedp and fused_col_supported(ren):
                rt = RenderTrainFused(ren, XA, it.feat, npts, pp["cond"], lins=lr)
            else:
                rt = RenderTrain(ren, XA, _p(it.feat), 256, npts, pp["cond"], lins=lr) if fusedp else \
                    RenderTrain(ren, XA, off(it.out, 1), 257, npts, pp["cond"], lins=lr)
            self.fg[p] = dict(it=it, rt=rt, X=X, jinv=jinv, XA=XA, sdf=sdf, nrm=nrm, gth=gth, zfinal=zfinal, iters=iters,
                              wcount=wcount, npts=npts, Pt=Pt, Rp=Rp, flags=flags, nn_posed=nn_posed,
                              nn_cano=nn_cano)
            z_l.append(zfinal); sdf_l.append(sdf); rgb_l.append(rt.rgb); nrm_l.append(nrm); inv_l.append(pp["inv_index"])

        self.reg_items, self.reg_losses = [], (None, None)
        if m.smpl_surface_weight > 0 or m.zero_pose_weight > 0:
            self._regularisers_forward()
        all_persons = list(persons)
        self.remote = {}
        s0, s1 = 0, R
        if self.shard is not None:
            world, rank = self.shard
            P_total = int(self.input["smpl_trans"].shape[1])
            all_persons = list(range(P_total))
            n_slot = (P_total + world - 1) // world
            E = N_EIKONAL
            width = NZ + 7 * S + 3                                   # z, sdf, rgb3, nrm3 per sample; hit, off, in flags
            send = torch.zeros(n_slot, R * width + E * 3, **f32)
            for j, p in enumerate(persons):
                f, pp = self.fg[p], cx["per"][p]
                n_hit = int(cx["n_hit"][j])
                rows = pp["hit_index"][:n_hit].long()
                dense = torch.zeros(R, width, **f32)
                dense[rows, :NZ] = f["zfinal"][:n_hit]
                dense[rows, NZ:NZ + S] = f["sdf"][:n_hit * S].reshape(n_hit, S)
                dense[rows, NZ + S:NZ + 4 * S] = f["rt"].rgb[:n_hit * S].reshape(n_hit, 3 * S)
                dense[rows, NZ + 4 * S:NZ + 7 * S] = f["nrm"][:n_hit * S].reshape(n_hit, 3 * S)
                dense[rows, NZ + 7 * S] = 1.0
                if f["flags"] is not None:
                    dense[rows, NZ + 7 * S + 1] = f["flags"][0][:n_hit].float()
                    dense[rows, NZ + 7 * S + 2] = f["flags"][1][:n_hit].float()
                send[j, :R * width] = dense.reshape(-1)
                send[j, R * width:] = f["gth"].reshape(-1)
            bufs = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(bufs, send)                               # THE exchange step of the forward
            ar = torch.arange(R, device=dev, dtype=torch.int32)
            for p in all_persons:
                if p in self.fg:
                    continue
                blk = bufs[p % world][p // world]
                d = blk[:R * width].reshape(R, width)
                hit = d[:, NZ + 7 * S] > 0.5
                self.remote[p] = dict(z=d[:, :NZ].contiguous(), sdf=d[:, NZ:NZ + S].contiguous(),
                                      rgb=d[:, NZ + S:NZ + 4 * S].contiguous(), nrm=d[:, NZ + 4 * S:NZ + 7 * S].contiguous(),
                                      inv=torch.where(hit, ar, torch.full_like(ar, -1)).contiguous(), hit=hit,
                                      off=d[:, NZ + 7 * S + 1] > 0.5, inn=d[:, NZ + 7 * S + 2] > 0.5,
                                      gth=blk[R * width:].reshape(E, 3).contiguous())
            z_l, sdf_l, rgb_l, nrm_l, inv_l = [], [], [], [], []
            for p in all_persons:
                if p in self.fg:
                    f = self.fg[p]
                    z_l.append(f["zfinal"]); sdf_l.append(f["sdf"]); rgb_l.append(f["rt"].rgb); nrm_l.append(f["nrm"])
                    inv_l.append(cx["per"][p]["inv_index"])
                else:
                    r = self.remote[p]
                    z_l.append(r["z"]); sdf_l.append(r["sdf"]); rgb_l.append(r["rgb"]); nrm_l.append(r["nrm"]); inv_l.append(r["inv"])
            n_slice = (R + world - 1) // world
            s0, s1 = min(R, rank * n_slice), min(R, (rank + 1) * n_slice)
            self.n_slice = n_slice
        self.all_persons, self.ray_slice = all_persons, (s0, s1)

        # ---- background (multiply.py:482-484, 514-539); depths jittered per ray in training (ray_sampler.py:32-40)
        self.bg = None
        bg_rgb = None
        if self.input.get("idx", None) is not None:
            key = "image_id" if "image_id" in self.input else "idx"
            # the frame's row of the latent table, looked up ON THE DEVICE: int(<device tensor>) is a device -> host copy that
            # waits for everything enqueued so far (the whole previous iteration), i.e. the host could never run ahead
            w_lat = m.frame_latent_encoder.weight
            self.frame = torch.as_tensor(self.input[key]).reshape(-1)[:1].to(w_lat.device, torch.long, non_blocking=True)
            code = w_lat.detach().index_select(0, self.frame)[0].contiguous()
            NB = rs.N_samples_inverse_sphere
            Rb = s1 - s0                                              # this rank's rays of the background branch
            bdirs = dirs[s0:s1].contiguous()
            # stratified depths (ray_sampler.py:32-40): the bin edges are constants of (NB, bounding sphere) -- built once per model,
            # already flipped and scaled, so that the iteration pays one fused multiply-add instead of ten small launches
            strata = m.__dict__.get("_mp_bg_strata")
            if strata is None or strata[0] != (NB, float(rs.scene_bounding_sphere), str(dev)):
                t = torch.linspace(0.0, 1.0, NB, device=dev)[None]
                mids = 0.5 * (t[:, 1:] + t[:, :-1])
                upper = torch.cat([mids, t[:, -1:]], -1); lower = torch.cat([t[:, :1], mids], -1)
                c = 1.0 / rs.scene_bounding_sphere
                strata = m.__dict__["_mp_bg_strata"] = ((NB, float(rs.scene_bounding_sphere), str(dev)),
                                                        torch.flip(lower * c, dims=[-1]).contiguous(),
                                                        torch.flip((upper - lower) * c, dims=[-1]).contiguous())
            # zbg = flip((lower + (upper - lower) u) / r) = flip(lower / r) + flip((upper - lower) / r) flip(u)
            zbg = torch.addcmul(strata[1], strata[2], torch.flip(self.draws["bg_rand"][s0:s1], dims=[-1])).contiguous()
            bg_rgb = torch.zeros(R, 3, **f32) if Rb != R else None
            if Rb > 0:
                pts = torch.empty(Rb * NB, 4, **f32)
                cam = pose.reshape(4, 4)[:3, 3].contiguous()
                _chk(L.mp_tr_bg_points(_p(bdirs), _p(cam), _p(zbg), Rb, NB, C.c_float(m.sdf_bounding_sphere), _p(pts), st),
                     "mp_tr_bg_points")
                bgnet = m.bg_implicit_network
                fused_bg = BG_TRAIN_MODE == "fused" and TRAIN_PRECISION == "bf16x3" and fused_bg_supported(bgnet)
                if fused_bg:       # the nine layers in one launch (csrc/tfuse.hip k_tf_bg_fwd)
                    bit = ImplicitTrainFusedBG(bgnet, pts, code, lins=self.ts.lins[id(bgnet)])
                else:
                    bit = ImplicitTrain(bgnet, pts, code, fwd=False, lins=self.ts.lins[id(bgnet)])
                drep = bdirs[:, None, :].expand(Rb, NB, 3).reshape(-1, 3).contiguous()
                XAb = torch.empty(Rb * NB, 27, **f32)
                _chk(L.mp_tr_pe(_p(drep), 3, Rb * NB, 4, 0, C.c_float(1.0), _p(XAb), 27, 0, st), "mp_tr_pe")
                if fused_bg:
                    brt = RenderTrain(m.bg_rendering_network, XAb, _p(bit.feat), 256, Rb * NB, code,
                                      lins=self.ts.lins[id(m.bg_rendering_network)])
                    sdfb = bit.sdf[:Rb * NB]
                else:
                    brt = RenderTrain(m.bg_rendering_network, XAb, off(bit.out, 1), 257, Rb * NB, code,
                                      lins=self.ts.lins[id(m.bg_rendering_network)])
                    sdfb = torch.empty(Rb * NB, **f32)
                    _chk(L.mp_tr_copy_cols(_p(bit.out), 257, 0, _p(sdfb), 1, 0, Rb * NB, 1, C.c_float(1.0), 0, st),
                         "mp_tr_copy_cols")
                bg_slice = torch.empty(Rb, 3, **f32)
                _chk(L.mp_tr_bg_comp_fwd(_p(sdfb), _p(brt.rgb), _p(zbg), Rb, NB, _p(bg_slice), st), "mp_tr_bg_comp_fwd")
                if bg_rgb is None:
                    bg_rgb = bg_slice                                 # the whole call's rays: no scatter into a zero image
                else:
                    bg_rgb[s0:s1] = bg_slice
                self.bg = dict(it=bit, rt=brt, zbg=zbg, sdfb=sdfb, XAb=XAb, NB=NB, code=code, pts=pts, Rb=Rb)
            if self.shard is not None:                                # every rank composites all rays
                pad = torch.zeros(self.n_slice, 3, **f32)
                pad[:Rb] = bg_rgb[s0:s1]
                parts = [torch.empty_like(pad) for _ in range(self.shard[0])]
                dist.all_gather(parts, pad)
                bg_rgb = torch.cat(parts, 0)[:R].contiguous()
        self.bg_rgb = bg_rgb

        # ---- compositing (multiply.py:425-480, 544-545)
        self.tabs = tuple(_table(ts, dev) for ts in (inv_l, z_l, sdf_l, rgb_l, nrm_l))
        t_inv, t_z, t_sdf, t_rgb, t_nrm = self.tabs
        P = len(all_persons)
        rgb_values = torch.empty(R, 3, **f32); fg_rgb_values = torch.empty(R, 3, **f32)
        normal_values = torch.empty(R, 3, **f32); acc_map = torch.empty(R, **f32)
        acc_person = torch.empty(R, P, **f32); bg_T = torch.empty(R, **f32)
        _chk(L.mp_composite(R, P, NZ, _p(t_inv), _p(t_z), _p(t_sdf), _p(t_rgb), _p(t_nrm), _p(beta),
                            _p(bg_rgb) if bg_rgb is not None else None, _p(rgb_values), _p(fg_rgb_values),
                            _p(normal_values), _p(acc_map), _p(acc_person), _p(bg_T), st), "mp_composite")
        grad_theta = torch.cat([self.fg[p]["gth"] if p in self.fg else self.remote[p]["gth"] for p in all_persons],
                               0)[None]                                            # multiply.py:565
        self.bg_T = bg_T
        self.NZ = NZ
        if self.reg_items:
            return (rgb_values, normal_values, acc_map, acc_person, grad_theta) + self.reg_losses
        return rgb_values, normal_values, acc_map, acc_person, grad_theta

    # ---- the two optional regularisers (multiply.py:336-394) -----------------------------------------------------------------
    def _regularisers_forward(self):
        """smpl_surface: the SDF at `num_pixels` posed SMPL vertices (head / hands / feet left out), warped to canonical space, must not
        exceed 0.02 -- mean of (sdf - 0.02) over the offenders, per rendered person.  zero_pose: for every rendered person q and every
        network p, the network's outputs at 2 000 vertices of p's canonical mesh under q's pose conditioning vs under a zero
        conditioning -- L1 of the sdf + L1 of the features (the reference pairs q's conditioning with network p exactly like this).
        Value-only passes of the SDF net, layer by layer (ImplicitTrain) on the iteration's shared weights; their adjoints run at the
        head of the backward sweep, before any person's weight-norm adjoint retires its accumulators."""
        m, cx, L, st = self.model, self.cx, hip.lib(), hip.stream()
        dev = cx["dev"]
        if self.pose_grad and m.smpl_surface_weight > 0:
            # (the zero-pose term's pose adjoint -- it reaches the pose through the conditioning only -- is built, round 6; the surface
            # term's needs the adjoint of the posed VERTICES through blend shapes and skinning, which mp_smpl_pose_bwd does not carry)
            raise NotImplementedError("smpl_surface with body-model inputs under optimisation: the adjoint of the posed vertices is not built")
        ssl = torch.zeros(1, dtype=F32, device=dev)
        zpl = torch.zeros(1, dtype=F32, device=dev)
        for q in cx["persons"]:
            pp = cx["per"][q]
            cond = pp["cond"]
            if m.smpl_surface_weight > 0:
                imp = m.foreground_implicit_network_list[q]
                idx = self.draws["person"][q]["surf_idx"].to(dev).long()
                pts = pp["verts"].index_select(0, idx).contiguous()
                n = pts.shape[0]
                xc = torch.empty(n, 3, dtype=F32, device=dev)
                _chk(L.mp_warp_inverse(_p(pts), None, None, None, None, None, 0, 1, n, _p(pp["vsorted"]), _p(pp["cbound"]),
                                       _p(pp["btab"]), 0, None, None, _p(xc), None, None, None, None, None, st), "mp_warp_inverse")
                it = ImplicitTrain(imp, xc, cond, fwd=False, lins=self.ts.lins[id(imp)])
                sdf = it.out[:, 0]
                mask = sdf > SMPL_SURFACE_THRESHOLD
                cnt = mask.sum().clamp(min=1).to(F32)
                ssl = ssl + torch.where(mask, sdf - SMPL_SURFACE_THRESHOLD, torch.zeros_like(sdf)).sum() / cnt      # 0 when none offends
                self.reg_items.append(("surf", it, mask, cnt))
            if m.zero_pose_weight > 0:
                for p, vcano in enumerate(m.mesh_v_cano_list):
                    net = m.foreground_implicit_network_list[p]
                    pts = vcano.reshape(-1, 3).to(dev).float().index_select(0, self.draws["zp_idx"][(q, p)].to(dev).long()).contiguous()
                    lins = self.ts.lins[id(net)]
                    it1 = ImplicitTrain(net, pts, cond, fwd=False, lins=lins)
                    it0 = ImplicitTrain(net, pts, torch.zeros_like(cond), fwd=False, lins=lins)
                    d = it1.out - it0.out
                    zpl = zpl + d[:, 0].abs().mean() + d[:, 1:].abs().mean()
                    self.reg_items.append(("zero", it1, it0, torch.sign(d), q))
        self.reg_losses = (ssl, zpl)

    def _regularisers_backward(self, d_ssl, d_zpl):
        dev = self.cx["dev"]
        self.reg_dcond = {}             # person -> adjoint of its pose conditioning from the zero-pose term (pose optimisation)
        for item in self.reg_items:
            if item[0] == "surf":
                _, it, mask, cnt = item
                if d_ssl is None:
                    continue
                dZ = torch.zeros(it.P, 257, dtype=F32, device=dev)
                dZ[:, 0] = d_ssl.reshape(()) * mask.to(F32) / cnt
                it.backward(dZ)
            else:
                _, it1, it0, sg, q = item
                if d_zpl is None:
                    continue
                n = sg.shape[0]
                dZ = torch.empty(n, 257, dtype=F32, device=dev)
                dZ[:, 0] = sg[:, 0] * (d_zpl.reshape(()) / n)
                dZ[:, 1:] = sg[:, 1:] * (d_zpl.reshape(()) / (n * 256))
                dcond = it1.backward(dZ)
                it0.backward(-dZ)
                if self.pose_grad and not self.cond_zero:
                    # person q's conditioning = smpl_pose[q, 3:] / pi (multiply.py:270): the term's only path to the body-model inputs
                    self.reg_dcond[q] = dcond if q not in self.reg_dcond else self.reg_dcond[q] + dcond

    # ---- backward -----------------------------------------------------------------------------------------------
    def backward(self, d_rgb_values, d_acc_map, d_acc_person, d_grad_theta, d_ssl=None, d_zpl=None):
        """-> {id(parameter): gradient}"""
        m, cx, L = self.model, self.cx, hip.lib()
        dev, R, beta = cx["dev"], cx["R"], cx["beta"]
        st = hip.stream()
        f32 = dict(dtype=F32, device=dev)
        persons = cx["persons"]
        all_persons = self.all_persons
        self.ts.start_backward()                  # this sweep's own accumulators / gradient buffer (TrainState.start_backward)
        if self.reg_items:
            self._regularisers_backward(d_ssl, d_zpl)
        P = len(all_persons)
        S = self.NZ - 1
        t_inv, t_z, t_sdf, t_rgb, _ = self.tabs
        zp = self.zp = _ZP[0] = hip.ZeroPool(dev)          # this sweep's zero-initialised tensors: views of one zero-filled block
        zero = lambda t, shape: zp.take(shape) if t is None else t.contiguous().float()
        d_rgb_values = zero(d_rgb_values, (R, 3)); d_acc_map = zero(d_acc_map, (R,)); d_acc_person = zero(d_acc_person, (R, P))
        # remote persons (person-sharded mode) get scratch rows: their owners compute the same compositing adjoint
        # (a fused person's d sdf vector also covers its eikonal points, which no compositing term reaches: zeros)
        dsdf_l = [zp.take((self.fg[p]["Pt"] if isinstance(self.fg[p]["it"], ImplicitTrainFused) else self.fg[p]["npts"])
                          if p in self.fg else R * S) for p in all_persons]
        drgb_l = [zp.take(self.fg[p]["npts"] if p in self.fg else R * S, 3) for p in all_persons]
        d_bg_rgb = zp.take(R, 3)
        d_beta = zp.take(1)
        t_dsdf, t_drgb = _table(dsdf_l, dev), _table(drgb_l, dev)
        _chk(L.mp_tr_composite_bwd(R, P, self.NZ, _p(t_inv), _p(t_z), _p(t_sdf), _p(t_rgb), _p(beta),
                                   _p(self.bg_rgb) if self.bg_rgb is not None else None, _p(d_rgb_values), _p(d_acc_map),
                                   _p(d_acc_person), _p(t_dsdf), _p(t_drgb), _p(d_bg_rgb), _p(d_beta), st),
             "mp_tr_composite_bwd")
        grads = {}
        self.pose_grads = {}
        # data-parallel training: buckets of retired gradients are all-reduced while the sweep goes on (parallel.BucketedGradientSync)
        sync = getattr(m, "grad_bucket_sync", None)
        # the buckets retire in the order of THIS rank's persons: person-sharded ranks own different persons, so their bucket
        # sizes and counts differ and the collectives would mismatch -- that mode sums its shared gradients with PersonShardedGradSync
        assert sync is None or self.shard is None, "grad_bucket_sync (ray-/frame-sharded data parallelism) cannot be combined with shard="

        def collect(obj):
            for prm, g in zip(obj.params(), obj.param_grads()):
                grads[id(prm)] = g if id(prm) not in grads else grads[id(prm)] + g

        def retire(*objs):
            if sync is not None:
                prms = [prm for o in objs for prm in o.params()]
                sync.retire(prms, [grads[id(prm)] for prm in prms])

        for p in persons:
            n = all_persons.index(p)                                  # position among the composited persons
            f = self.fg[p]
            it, rt, npts, Pt = f["it"], f["rt"], f["npts"], f["Pt"]
            rev = isinstance(it, ImplicitTrainRev)
            fusedp = isinstance(it, ImplicitTrainFused)
            dgrad = zp.take(Pt, 3) if rev else None
            dXA = torch.empty(npts, 6, **f32)
            if fusedp:                 # d features as their own aligned matrix, written (not accumulated) by the colour net
                dZ8 = None
                dfeat = torch.empty(Pt, 256, **f32)
                dfeat[npts:].zero_()   # the eikonal points have no colour path
                tn_groups = []             # the person's aligned weight-gradient contractions: ONE grouped launch below
                if isinstance(rt, RenderTrainFused):
                    rt.backward(drgb_l[n], dXA, dfeat, tn_groups=tn_groups)
                else:
                    rt.backward(drgb_l[n], dXA, _p(dfeat), 256, feat_accumulate=False)
            else:
                dZ8 = torch.zeros(Pt if rev else 4 * Pt, 257, **f32)
                rt.backward(drgb_l[n], dXA, off(dZ8, 1), 257)
            djinv = torch.empty(npts, 9, **f32) if self.pose_grad else None
            _chk(L.mp_tr_shade_in_bwd(None if fusedp else _p(it.out), Pt, npts, _p(f["jinv"]), _p(dXA), _p(dsdf_l[n]), None, _p(dZ8),
                                      _p(djinv), _p(it.grad) if rev else None, _p(dgrad), st), "mp_tr_shade_in_bwd")
            if d_grad_theta is not None:
                dg = d_grad_theta.reshape(-1, 3)[n * N_EIKONAL:(n + 1) * N_EIKONAL].contiguous().float()
                _chk(L.mp_tr_eik_bwd(Pt, npts, N_EIKONAL, _p(dg), _p(dZ8), _p(dgrad), st), "mp_tr_eik_bwd")
            if fusedp:
                dcond = it.backward(dfeat, dsdf_l[n], dgrad, tn_groups=tn_groups)
                ImplicitTrainFused._launch(tn_groups)
            else:
                dcond = it.backward(dZ8, dgrad, want_dx=self.pose_grad) if rev else it.backward(dZ8, want_dx=self.pose_grad)
            self.ts.finish_group(p)                                   # one batched weight-norm adjoint for the person's two nets
            collect(it); collect(rt)
            retire(it, rt)
            if self.pose_grad:
                # x_c enters the SDF net (value + tangent rows) and the colour net (XA[:, :3]); the transforms also shape
                # the normals through Jinv.  -> d tfs -> d (scale, transl, thetas, betas)   (multiply.py:196-206, 270)
                pp = cx["per"][p]
                server = m.smpl_server_list[p]
                dxc = (it.dx[:npts] + dXA[:, :3]).contiguous()
                dtfs = torch.zeros(24, 16, **f32)
                _chk(L.mp_tr_warp_bwd(_p(f["X"]), _p(dxc), _p(f["jinv"]), _p(djinv), _p(f["nn_posed"]), _p(f["nn_cano"]),
                                      npts, _p(server.tables.lbs_weights), _p(pp["tfs"]), _p(dtfs), st), "mp_tr_warp_bwd")
                dprm = torch.empty(86, **f32)
                _chk(L.mp_smpl_pose_bwd(_p(server.tables.parents), _p(pp["prm"]), _p(server.tfs_c_inv),
                                        _p(pp["rest_joints"]), _p(server.tables.j_shapedirs), _p(dtfs), _p(dprm), st),
                     "mp_smpl_pose_bwd")
                if not self.cond_zero:                                   # cond = smpl_pose[3:] / pi  (multiply.py:270)
                    dc = dcond + torch.mv(rt.lp_w.t(), rt.extra_grads[1])
                    if p in getattr(self, "reg_dcond", {}):             # + the zero-pose regulariser's share (round 6)
                        dc = dc + self.reg_dcond[p]
                    dprm[7:76] += dc / math.pi
                self.pose_grads[p] = dprm
        if self.bg is not None:
            b = self.bg
            bit, brt, NB, Rb = b["it"], b["rt"], b["NB"], b["Rb"]
            rows = Rb * NB
            s0, s1 = self.ray_slice
            d_bg_slice = d_bg_rgb[s0:s1].contiguous()
            dsdfb = torch.empty(rows, **f32); drgbb = torch.empty(rows, 3, **f32)
            _chk(L.mp_tr_bg_comp_bwd(_p(b["sdfb"]), _p(brt.rgb), _p(b["zbg"]), Rb, NB, _p(d_bg_slice), _p(dsdfb), _p(drgbb),
                                     st), "mp_tr_bg_comp_bwd")
            dXAb = torch.empty(rows, 27, **f32)
            if isinstance(bit, ImplicitTrainFusedBG):
                dfeatb = torch.empty(rows, 256, **f32)         # written (not accumulated) by the colour net
                dcode = brt.backward(drgbb, dXAb, _p(dfeatb), 256, feat_accumulate=False)
                dcode = dcode + bit.backward(dfeatb, dsdfb)
            else:
                dZ8b = torch.zeros(rows, 257, **f32)
                dcode = brt.backward(drgbb, dXAb, off(dZ8b, 1), 257)
                _chk(L.mp_tr_copy_cols(_p(dsdfb), 1, 0, _p(dZ8b), 257, 0, rows, 1, C.c_float(1.0), 0, st), "mp_tr_copy_cols")
                dcode = dcode + bit.backward(dZ8b)
            self.ts.finish_group("bg")
            collect(bit); collect(brt)
            w = m.frame_latent_encoder.weight
            gw = zp.take(tuple(w.shape), dtype=w.dtype)
            gw.index_copy_(0, self.frame, dcode.reshape(1, -1))
            grads[id(w)] = gw
        bp = m.density.beta
        grads[id(bp)] = (d_beta.reshape(bp.shape) * torch.sign(bp.detach())).to(bp.dtype)     # density.py:31-33
        if sync is not None:
            tail = [bp] + ([m.frame_latent_encoder.weight] + b["it"].params() + b["rt"].params() if self.bg is not None else [])
            sync.retire(tail, [grads[id(prm)] for prm in tail])
            grads = sync.finish(grads)
        _ZP[0] = None
        return grads


class _TrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, smpl_pose, smpl_trans, smpl_shape, *params):
        ctx.graph, ctx.params = graph, params
        ctx.body = (smpl_pose, smpl_trans, smpl_shape)
        outs = graph.run()
        ctx.mark_non_differentiable(outs[1])          # normal_values: no loss term reads it (loss.py:108-177)
        return outs

    @staticmethod
    def backward(ctx, d_rgb, d_nrm, d_acc, d_accp, d_gth, d_ssl=None, d_zpl=None):
        graph = ctx.graph
        g = graph.backward(d_rgb, d_acc, d_accp, d_gth, d_ssl, d_zpl)
        body = [None, None, None]
        if graph.pose_grad:
            sl = ((4, 76), (1, 4), (76, 86))
            for k, t in enumerate(ctx.body):
                if t is not None and ctx.needs_input_grad[1 + k]:
                    gt = torch.zeros_like(t)
                    for p, dprm in graph.pose_grads.items():
                        gt[0, p] = dprm[sl[k][0]:sl[k][1]].to(t.device)
                    body[k] = gt
        return (None, *body) + tuple(g.get(id(p)) for p in ctx.params)


def forward_train(model, input, id=-1, cond_zero_shit=False, canonical_pose=False, draws=None, shard=None):
    """Multiply.forward with self.training == True.  Returns the reference's 19-key dict (multiply.py:566-588); the five
    tensors rgb_values / acc_map / acc_person_list / grad_theta (/ normal_values, not differentiable) hang off ONE autograd
    node whose backward is the hand-written adjoint sweep."""
    epoch = int(input["current_epoch"])
    if (model.smpl_surface_weight > 0 or model.zero_pose_weight > 0) and shard is not None:
        raise NotImplementedError("the smpl_surface / zero_pose regularisers (multiply.py:336-394) are not built for person-sharded training")
    if model.zero_pose_weight > 0 and not (isinstance(id, int) and id == -1):
        # the zero-pose term evaluates EVERY person's network (multiply.py:377-394); the adjoint sweep retires only the rendered
        # persons' networks, so the other networks' weight gradients would be dropped while the loss value contains their terms
        raise NotImplementedError("zero_pose_weight > 0 with a subset of the persons rendered (id != -1): the regulariser's gradients to "
                                  "the networks that are not rendered are not collected")
    if shard is not None:                   # person-sharded: this rank evaluates persons {p : p % world == rank}
        assert id == -1, "person-sharded training renders all persons"
        id = [p for p in range(int(input["smpl_trans"].shape[1])) if p % shard[0] == shard[1]]
    # the setup's one host sync must not wait for the previous iteration's backward pass (Multiply._setup); opt in when the
    # inputs are resident (model.async_setup = True: bench.py, a prefetching data loader)
    cx = model._setup(input, id, canonical_pose, side_stream=bool(getattr(model, "async_setup", False)))
    dev = cx["dev"]
    cond_zero = epoch < 20 or epoch % 20 == 0 or bool(cond_zero_shit)              # multiply.py:271-273
    if draws is None:
        draws = make_draws(model, cx)
    body = [input["smpl_pose"], input["smpl_trans"], input["smpl_shape"]]
    pose_grad = (not canonical_pose) and any(torch.is_tensor(t) and t.requires_grad for t in body)
    graph = TrainGraph(model, cx, input, cond_zero, draws, surface_flags=epoch < 250, pose_grad=pose_grad, shard=shard)
    params = [p for p in model.parameters() if p.requires_grad]
    with torch.enable_grad():                                                       # multiply.py:176
        outs = _TrainFn.apply(graph, *body, *params)
        rgb_values, normal_values, acc_map, acc_person, grad_theta = outs[:5]
        smpl_surface_loss, zero_pose_loss = (outs[5], outs[6]) if len(outs) == 7 else (None, None)
        temporal_loss = torch.zeros(1, device=dev)
        if epoch > 250:                                                             # multiply.py:242-243
            temporal_loss = torch.mean(torch.square(input["smpl_pose_last"].to(dev) - input["smpl_pose"].to(dev)))
    last = graph.all_persons[-1]
    cam = cx["pose"].reshape(4, 4)[:3, 3]
    if last in graph.fg:
        fl = graph.fg[last]
        hit = cx["per"][last]["hit_index"][:fl["Rp"]].long()
        points = cam[None, None, :] + fl["zfinal"][:, :-1, None] * cx["dirs"][hit][:, None, :]
    else:                                   # person-sharded: the last person lives on another rank
        rl = graph.remote[last]
        hit = torch.nonzero(rl["hit"]).flatten()
        if hit.numel() == 0:
            hit = torch.zeros(1, dtype=torch.long, device=dev)      # the reference's empty-hit fallback, multiply.py:262-263
        points = cam[None, None, :] + rl["z"][hit][:, :-1, None] * cx["dirs"][hit][:, None, :]
    _z3 = torch.zeros(3, device=dev)            # the dict's constant zero entries: one fill
    _zi = iter(range(3))
    zeros1 = lambda: _z3[next(_zi):][:1]
    index_off_surface = index_in_surface = None
    if epoch < 250:                                                                 # multiply.py:549-557
        P = len(graph.all_persons)
        off_all = torch.ones(cx["R"], P, dtype=torch.bool, device=dev)
        in_all = torch.zeros(cx["R"], P, dtype=torch.bool, device=dev)
        for n, p in enumerate(graph.all_persons):
            if p in graph.fg:
                f = graph.fg[p]
                rays = cx["per"][p]["hit_index"][:f["Rp"]].long()
                off_all[rays, n] = f["flags"][0]
                in_all[rays, n] = f["flags"][1]
            else:
                r = graph.remote[p]
                off_all[:, n] = torch.where(r["hit"], r["off"], off_all[:, n])
                in_all[:, n] = torch.where(r["hit"], r["inn"], in_all[:, n])
        index_off_surface, index_in_surface = off_all.all(dim=1), in_all.any(dim=1)
    out = {
        "zero_pose_loss": zero_pose_loss if zero_pose_loss is not None else zeros1(), "t_list": [], "fg_rgb_values_each_person_list": [],
        "cam_loc": cam[None].expand(cx["R"], 3), "hitted_mask_idx": [], "mean_hitted_vertex_list": [],
        "points": points, "rgb_values": rgb_values, "normal_values": normal_values,
        "index_outside": input.get("index_outside"), "index_off_surface": index_off_surface,
        "index_in_surface": index_in_surface,
        "acc_map": acc_map, "grad_theta": grad_theta, "interpenetration_loss": zeros1(), "temporal_loss": temporal_loss,
        "acc_person_list": acc_person, "smpl_surface_loss": smpl_surface_loss if smpl_surface_loss is not None else zeros1(),
        "epoch": input["current_epoch"],
    }
    if "sam_mask" in input:
        out["sam_mask"] = input["sam_mask"].squeeze()
    model._last_train = graph
    model.last_stats = {"n_hit": cx["n_hit"], "iters": [graph.fg[p]["iters"] for p in cx["persons"]],
                        "n_sdf_evals": [graph.fg[p]["wcount"] for p in cx["persons"]],
                        "hull_host_fallbacks": getattr(model, "hull_host_fallbacks", 0)}
    return out
