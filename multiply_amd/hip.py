"""ctypes binding of libmultiply_hip.so (include/multiply_hip.h) + weight packing helpers.

This is the only place the Python host side touches the C ABI.  There is NO fallback: if the shared library is
missing or the device is not a gfx950, importing/using this module raises.
PyTorch is used for device memory and streams only (tensor.data_ptr(), torch.cuda.current_stream()).
"""
import ctypes as C
import functools
import math
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MP_LIB_PATH", os.path.join(_HERE, "libmultiply_hip.so"))  # override: experiments only

MAX_LAYERS, MAX_CHUNKS, BIAS_STRIDE = 10, 9, 288
ACT_NONE, ACT_SOFTPLUS, ACT_RELU, ACT_SIGMUL = 0, 1, 2, 3
KNN_CLUSTER, KNN_NC = int(os.environ.get("MP_KNN_CLUSTER", 32)), int(os.environ.get("MP_KNN_NC", 216))   # = include/multiply_hip.h
KNN_CB_ROWS = KNN_NC + KNN_NC // 2      # mp_knn_build's sphere table: the clusters, then the pairs of clusters the training searches use


class MpLayer(C.Structure):
    _fields_ = [("n_chunk", C.c_int), ("use_reg", C.c_int), ("use_in", C.c_int), ("act", C.c_int),
                ("out_chunk", C.c_int), ("aux", C.c_int)]


class MpNet(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("total_chunks", C.c_int), ("layer", MpLayer * MAX_LAYERS)]


class MpPackLayer(C.Structure):
    _fields_ = [("v", C.c_void_p), ("g", C.c_void_p), ("b", C.c_void_p), ("rowmap", C.c_void_p), ("colmap", C.c_void_p),
                ("colscale", C.c_void_p), ("hoist_vec", C.c_void_p), ("wpack_layer", C.c_void_p), ("bias_layer", C.c_void_p),
                ("out_dim", C.c_int), ("in_dim", C.c_int), ("n_rows", C.c_int), ("hoist_col0", C.c_int), ("hoist_n", C.c_int),
                ("bias_scale", C.c_float)]


class MpSamplerCfg(C.Structure):
    _fields_ = [("n_samples", C.c_int), ("n_samples_eval", C.c_int), ("n_samples_extra", C.c_int),
                ("beta_iters", C.c_int), ("max_total_iters", C.c_int), ("eps", C.c_float), ("add_tiny", C.c_float),
                ("near_", C.c_float)]


class MpSamplerState(C.Structure):
    _fields_ = [("zs", C.c_void_p), ("sdfs", C.c_void_p), ("nz", C.c_void_p), ("znew", C.c_void_p),
                ("sdfnew", C.c_void_p), ("beta", C.c_void_p), ("ray_active", C.c_void_p), ("group_flag", C.c_void_p),
                ("zfinal", C.c_void_p), ("iters", C.c_void_p), ("any_active", C.c_void_p)]


_lib = None


def lib():
    """Loads the HIP library; raises loudly when it is absent (no CPU / PyTorch fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                               f"g.build()'` (hipcc --offload-arch=gfx950). The MultiPly hot path has no fallback.")
        if "MP_LIB_PATH" not in os.environ:
            # the library must be what THIS tree's sources produce (multiply_amd/build.py stamps it with a content hash)
            from . import build as B
            have, want = B.library_hash(), B.source_hash()
            if have is not None and have != want:
                raise RuntimeError(f"{LIB_PATH} was built from sources {have}, this tree is {want}: rebuild it with "
                                   f"`python -m multiply_amd.build` (a stale library is never loaded)")
        _lib = C.CDLL(LIB_PATH)
        _declare_prototypes(_lib)
    return _lib


def lib_source_sha16():
    """content hash of the sources the loaded library was built from (multiply_amd/build.py)"""
    from . import build as B
    return B.library_hash()


HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "multiply_hip.h")


def header_prototypes(path=HEADER_PATH):
    """{name: (restype, [argtypes])} parsed from include/multiply_hip.h -- the header is the single source of truth
    for the C ABI; scalars map to their exact ctypes width (a `long long` or `float` passed as a default Python int /
    double would otherwise be widened or truncated by ctypes' default conversions), every pointer to c_void_p."""
    import re
    with open(path) as f:
        txt = re.sub(r"/\*.*?\*/", " ", f.read(), flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    scal = {"int": C.c_int, "float": C.c_float, "long long": C.c_longlong, "unsigned": C.c_uint, "void": None}
    protos = {}
    for ret, name, args in re.findall(r"\b(int|void|const char\s*\*)\s+(mp_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", txt):
        at = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a:
                at.append(C.c_void_p)
            else:
                ty = re.sub(r"\bconst\b", "", a).strip().rsplit(" ", 1)[0].strip()
                at.append(scal[ty])
        protos[name] = (C.c_char_p if "char" in ret else (None if ret == "void" else C.c_int), at)
    return protos


def _declare_prototypes(l):
    for name, (rt, at) in header_prototypes().items():
        fn = getattr(l, name)
        fn.restype, fn.argtypes = rt, at


def require_device():
    if not torch.cuda.is_available():
        raise RuntimeError("multiply_amd needs a ROCm device (MI355X / gfx950); none is visible")
    if not lib().mp_device_ok():
        raise RuntimeError("multiply_amd kernels are built for gfx950 only; the current device is not a gfx950")


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}")


# ------------------------------------------------------------------------------------------------ packing
def reg_slot_feature(s):
    """K slot s (0..255) of a register-fed K step -> index of the previous layer's output row it multiplies
    (the permutation documented in csrc/mlp_core.hpp)."""
    ks, sl = divmod(s, 32)
    g, e = divmod(sl, 8)
    return 32 * ks + (4 * g + e if e < 4 else 16 + 4 * g + e - 4)


_REG_FEATURE = np.array([reg_slot_feature(s) for s in range(256)], dtype=np.int64)


class LayerPlan:
    """How one nn.Linear maps onto the packed layout."""

    def __init__(self, lin, rowmap, reg_cols=None, in_cols=None, scale=1.0, hoist=None, act=ACT_NONE, out_chunk=-1,
                 in_scale=None, bias_scale=1.0, transpose=False, row_scale=None, aux=0):
        """transpose: pack W^T (rows = the layer's INPUT features, K = its outputs; reverse sweep); row_scale: optional
        per-row factors of the transposed matrix (the input-side scales of the forward plan); aux: see MpLayer.aux."""
        self.lin, self.act, self.out_chunk = lin, act, out_chunk
        self.transpose, self.row_scale, self.aux = transpose, row_scale, aux
        self.in_scale = float(scale if in_scale is None else in_scale)   # factor on the input-fed K slots
        self.bias_scale = float(bias_scale)
        rowmap = list(rowmap)
        while len(rowmap) % 32:
            rowmap.append(-1)
        self.rowmap = np.array(rowmap, dtype=np.int32)
        self.reg_cols = None if reg_cols is None else np.asarray(reg_cols, dtype=np.int64)
        self.in_cols = None if in_cols is None else np.asarray(in_cols, dtype=np.int64)
        self.scale = float(scale)
        self.hoist = hoist  # (col0, n) or None

    def colmap(self, ks_in):
        n = (8 + ks_in) * 32
        cm = -np.ones(n, dtype=np.int32)
        if self.reg_cols is not None:
            f = _REG_FEATURE
            ok = f < len(self.reg_cols)
            cm[:256][ok] = self.reg_cols[f[ok]]
        if self.in_cols is not None:
            assert len(self.in_cols) <= ks_in * 32
            cm[256:256 + len(self.in_cols)] = self.in_cols
        return cm


class PackedNet:
    """half (f16) fragment-ordered weights + fp32 bias table of one network role on the device."""

    def __init__(self, plans, ks_in, device):
        assert len(plans) <= MAX_LAYERS
        self.plans, self.ks_in, self.device = plans, ks_in, device
        self.chunk_bytes = 2 * (8 + ks_in) * 1024
        self.net = MpNet()
        self.net.n_layers = len(plans)
        off, self.offsets = 0, []
        for i, p in enumerate(plans):
            nch = len(p.rowmap) // 32
            assert 1 <= nch <= MAX_CHUNKS
            L = self.net.layer[i]
            L.n_chunk, L.use_reg, L.use_in = nch, int(p.reg_cols is not None), int(p.in_cols is not None)
            L.act, L.out_chunk, L.aux = p.act, p.out_chunk, p.aux
            self.offsets.append(off)
            off += nch
        self.net.total_chunks = off
        acts = [p.act != ACT_NONE for p in plans]
        assert acts == sorted(acts, reverse=True), "the core runs hidden layers first, then the linear output layer(s)"
        self.wpack = torch.zeros(off * self.chunk_bytes, dtype=torch.uint8, device=device)
        self.bias = torch.zeros(MAX_LAYERS * BIAS_STRIDE, dtype=torch.float32, device=device)
        self.rowmaps = [torch.from_numpy(p.rowmap).to(device) for p in plans]
        self.colmaps = [torch.from_numpy(p.colmap(ks_in)).to(device) for p in plans]
        self.colscales = []
        for p in plans:
            cs = torch.full(((8 + ks_in) * 32,), p.scale, dtype=torch.float32, device=device)
            cs[256:] = p.in_scale
            self.colscales.append(cs)
        self.version = None

    def _params(self, lin):
        if hasattr(lin, "weight_g"):
            return lin.weight_v, lin.weight_g, lin.bias
        return lin.weight, None, lin.bias

    def _pack_layer(self, i, weights, hoist_vec):
        p = self.plans[i]
        v, g, b = self._params(p.lin)
        if p.transpose:      # reverse sweep: effective weight (weight norm resolved), transposed, input-side scales on the rows
            w = v.detach().float()
            if g is not None:
                w = w * (g.detach().reshape(-1, 1) / w.norm(dim=1, keepdim=True))
            w = w.t()
            if p.row_scale is not None:
                w = w * torch.as_tensor(p.row_scale, dtype=torch.float32, device=w.device).reshape(-1, 1)
            v, g, b = w.contiguous(), None, torch.zeros(w.shape[0], dtype=torch.float32, device=w.device)
        v, b = v.detach().contiguous(), b.detach().contiguous()
        g = None if g is None else g.detach().reshape(-1).contiguous()
        h0, hn = p.hoist if (p.hoist is not None and hoist_vec is not None) else (0, 0)
        wp = C.c_void_p(self.wpack.data_ptr() + self.offsets[i] * self.chunk_bytes) if weights else None
        bp = C.c_void_p(self.bias.data_ptr() + 4 * i * BIAS_STRIDE)
        check(lib().mp_pack_layer(ptr(v), ptr(g), ptr(b), v.shape[0], v.shape[1], ptr(self.rowmaps[i]),
                                  len(p.rowmap), ptr(self.colmaps[i]), ptr(self.colscales[i]), self.ks_in, h0, hn,
                                  ptr(hoist_vec) if hn else None, C.c_float(p.bias_scale), wp, bp, stream()),
              "mp_pack_layer")

    def param_version(self):
        vs = [_GENERATION[0]]
        for p in self.plans:
            for t in self._params(p.lin):
                if t is not None:
                    vs.append((t.data_ptr(), t._version))
        return tuple(vs)

    def _table(self, weights, hoisted):
        """device-resident MpPackLayer records of every layer (mp_pack_layers), cached per (parameter pointers, variant): the
        hoisted vector is read from a persistent buffer of this object, so the table survives from call to call"""
        ptrs = tuple(t.data_ptr() if t is not None else 0 for p in self.plans for t in self._params(p.lin))
        key = (weights, hoisted)
        cache = self.__dict__.setdefault("_tables", {})
        if key not in cache or cache[key][0] != ptrs:
            arr = (MpPackLayer * len(self.plans))()
            for i, p in enumerate(self.plans):
                v, g, b = self._params(p.lin)
                assert v.is_contiguous() and b.is_contiguous() and v.dtype == torch.float32
                h0, hn = p.hoist if (p.hoist is not None and hoisted) else (0, 0)
                skip = not weights and p.hoist is None            # bias-only refresh: only the hoisted layer changes
                arr[i] = MpPackLayer(v.data_ptr(), g.data_ptr() if g is not None else None, b.data_ptr(),
                                     self.rowmaps[i].data_ptr(), self.colmaps[i].data_ptr(), self.colscales[i].data_ptr(),
                                     self._hoist_buf.data_ptr() if hn else None,
                                     (self.wpack.data_ptr() + self.offsets[i] * self.chunk_bytes) if weights else None,
                                     None if skip else self.bias.data_ptr() + 4 * i * BIAS_STRIDE,
                                     v.shape[0], v.shape[1], 0 if skip else len(p.rowmap), h0, hn, p.bias_scale)
            cache[key] = (ptrs, torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device))
        return cache[key][1]

    def refresh(self, hoist_vec=None, force=False):
        """(Re)packs the weights if a parameter changed (or `force`: training mode, see invalidate_packed), and the hoisted
        layer-0 bias for this call's conditioning."""
        ver = self.param_version()
        full = force or ver != self.version
        if not any(p.transpose for p in self.plans):
            # ONE launch for all layers (mp_pack_layers); the call's conditioning goes through a persistent buffer
            hoisted = hoist_vec is not None and any(p.hoist is not None for p in self.plans)
            if hoisted:
                hn = max(p.hoist[1] for p in self.plans if p.hoist is not None)
                if self.__dict__.get("_hoist_buf") is None or self._hoist_buf.numel() < hn:
                    self._hoist_buf = torch.empty(hn, dtype=torch.float32, device=self.device)
                    self.__dict__.pop("_tables", None)
                self._hoist_buf[:hn].copy_(hoist_vec.detach().reshape(-1)[:hn].float(), non_blocking=True)
            if full or hoisted:
                tab = self._table(bool(full), hoisted)
                check(lib().mp_pack_layers(ptr(tab), len(self.plans), self.ks_in, stream()), "mp_pack_layers")
            self.version = ver
            return
        for i, p in enumerate(self.plans):
            if full:
                self._pack_layer(i, True, hoist_vec if p.hoist is not None else None)
            elif p.hoist is not None:
                self._pack_layer(i, False, hoist_vec)
        self.version = ver


SOFTPLUS_K = 100.0 * math.log2(math.e)   # scaled units of the softplus networks (csrc/mlp_core.hpp)


def implicit_plans(net, variant):
    """LayerPlans of an ImplicitNet (fg: d_in 3, L 6, cond 69; bg: d_in 4, L 10, cond 32).
    variant 'sdf': last layer = sdf row only; 'full': 256 feature rows then the sdf row as the extra out chunk.
    Hidden layers work in scaled units z' = K z, h' = K h: biases and input-fed weights x K, register-fed weights
    unchanged, last (linear) layer's weights x 1/K."""
    E, nl, K = net.embed_dim, net.num_layers - 1, SOFTPLUS_K
    plans = []
    for l, lin in enumerate(net.layers()):
        out_dim = (lin.weight_v if hasattr(lin, "weight_v") else lin.weight).shape[0]
        last = l == nl - 1
        act = ACT_NONE if last else ACT_SOFTPLUS
        if last:
            if variant == "sdf":
                plans.append(LayerPlan(lin, [0], reg_cols=np.arange(256), scale=1.0 / K, act=act, out_chunk=0))
            else:
                plans.append(LayerPlan(lin, list(range(1, out_dim)) + [0], reg_cols=np.arange(256), scale=1.0 / K,
                                       act=act, out_chunk=8))
        elif l == 0:
            hoist = (E, net.cond_dim) if net.cond_dim > 0 else None
            plans.append(LayerPlan(lin, range(out_dim), in_cols=np.arange(E), hoist=hoist, act=act, in_scale=K,
                                   bias_scale=K))
        elif l in net.skip_in:
            prev = net.dims[l] - E   # width of the previous layer's output
            r2 = 1.0 / math.sqrt(2.0)
            plans.append(LayerPlan(lin, range(out_dim), reg_cols=np.arange(prev), in_cols=prev + np.arange(E), scale=r2,
                                   in_scale=r2 * K, act=act, bias_scale=K))
        else:
            plans.append(LayerPlan(lin, range(out_dim), reg_cols=np.arange(net.dims[l]), act=act, bias_scale=K))
    return plans


def implicit_grad_plans(net):
    """Reverse sweep of the foreground ImplicitNet for d sdf / d x (csrc/mlp.hip k_mlp_shade_rev, sweep 2): layers 7..1 transposed
    (rows = that layer's inputs, K = its outputs in K-slot order, outputs multiplied by the stored sigmoid of the layer
    below), then the input-fed part of layer 0 transposed.  The skip layer's transpose also carries the 39 rows of the
    re-injected encoding (captured as gradient rows, capture id 2); layer 0's rows are captured with id 1."""
    if not (net.d_in == 3 and net.multires == 6 and list(net.skip_in) == [4] and net.num_layers - 1 == 9):
        # the shipped configs' foreground network (confs/model/*.yaml: dims 8 x 256, skip_in [4], multires 6); other depths /
        # skip positions still render through the forward-mode kernel
        raise NotImplementedError("reverse-mode shading is specialised for the 9-layer, skip-at-4, multires-6 ImplicitNet of the "
                                  "shipped configs; set model.shade_mode = 'forward' (MP_SHADE_MODE=forward) for other shapes")
    E, K = net.embed_dim, SOFTPLUS_K
    lins = list(net.layers())
    r2 = 1.0 / math.sqrt(2.0)
    plans = []
    for l in range(7, 0, -1):
        lin = lins[l]
        wv = lin.weight_v if hasattr(lin, "weight_v") else lin.weight
        out_dim, in_dim = wv.shape
        row_scale, aux = None, (l - 1) + 1           # outputs (= h_{l-1} adjoint) x sigma'_{l-1}
        if l in net.skip_in:
            prev = in_dim - E
            row_scale = np.concatenate([np.full(prev, r2), np.full(E, r2 * K)])
            aux |= 2 << 8
        plans.append(LayerPlan(lin, range(in_dim), reg_cols=np.arange(out_dim), act=ACT_SIGMUL, transpose=True,
                               row_scale=row_scale, aux=aux))
    lin0 = lins[0]
    plans.append(LayerPlan(lin0, range(E), reg_cols=np.arange(lin0.bias.shape[0]), act=ACT_NONE, transpose=True,
                           row_scale=np.concatenate([np.full(E, K), np.zeros(net.cond_dim)]), aux=1 << 8))
    return plans


def sdf_row_slots(net):
    """the sdf row of the last layer (effective weights) as 256 halves in K-slot order: V_8 of the reverse sweep"""
    lin = list(net.layers())[-1]
    w = (lin.weight_v if hasattr(lin, "weight_v") else lin.weight).detach().float()
    if hasattr(lin, "weight_g"):
        w = w * (lin.weight_g.detach().reshape(-1, 1) / w.norm(dim=1, keepdim=True))
    idx = torch.as_tensor(_REG_FEATURE, device=w.device)
    return w[0][idx].to(torch.float16).contiguous()


def rendering_plans(net):
    """LayerPlans of a RenderingNet: 'pose_no_view' = [x_c3, n3 | pose8 hoisted | feat256]; 'nerf_frame_encoding' =
    [PE4(view) 27 | frame32 hoisted | feat256]."""
    plans, nl = [], net.num_layers - 1
    for l, lin in enumerate(net.layers()):
        out_dim = (lin.weight_v if hasattr(lin, "weight_v") else lin.weight).shape[0]
        last = l == nl - 1
        act = ACT_NONE if last else ACT_RELU
        rows = range(out_dim)
        if l == 0:
            if net.mode == "pose_no_view":
                plans.append(LayerPlan(lin, rows, reg_cols=14 + np.arange(256), in_cols=np.arange(6), hoist=(6, 8),
                                       act=act, out_chunk=0 if last else -1))
            else:
                plans.append(LayerPlan(lin, rows, reg_cols=59 + np.arange(256), in_cols=np.arange(27), hoist=(27, 32),
                                       act=act, out_chunk=0 if last else -1))
        else:
            plans.append(LayerPlan(lin, rows, reg_cols=np.arange(net.dims[l]), act=act, out_chunk=0 if last else -1))
    return plans


class _PinnedInts:
    """Small integer tables (device pointer lists, sizes) host -> device WITHOUT a host wait.  torch.tensor(list, device=dev)
    is a blocking copy from pageable memory: torch synchronises the current stream behind it, i.e. the host waits for every
    kernel enqueued so far -- seven such tables per training iteration kept the host from ever running ahead of the GPU.
    Here the values go into a ring of pinned memory and are copied with non_blocking=True (stream-ordered, no wait); a slot is
    reused after 64 k entries, long after its copy has run."""

    def __init__(self, n=1 << 16):
        self.buf = torch.empty(n, dtype=torch.int64).pin_memory()
        self.view = self.buf.numpy()
        self.pos = 0

    def put(self, values, dev):
        k = len(values)
        if self.pos + k > self.buf.numel():
            self.pos = 0
        sl = slice(self.pos, self.pos + k)
        self.view[sl] = values
        self.pos += k
        return self.buf[sl].to(dev, non_blocking=True)


_PINNED = {}


def device_ints(values, dev):
    """int64 device tensor of a short Python list (see _PinnedInts); CPU device: a plain tensor"""
    dev = torch.device(dev)
    if dev.type != "cuda":
        return torch.tensor(list(values), dtype=torch.int64, device=dev)
    ring = _PINNED.get("ring")
    if ring is None:
        ring = _PINNED["ring"] = _PinnedInts()
    return ring.put(list(values), dev)


class ZeroPool:
    """Zero-initialised scratch tensors of one call / one adjoint sweep as views of a few large zero-filled blocks: ONE fill per
    block instead of one per tensor (a training iteration asked for ~70 small zero tensors: gradient seeds, device counters,
    adjoint accumulators -- each a launch of its own).  A block lives as long as any of its views; a new pool per call, so a
    view handed to autograd as a gradient is never written again.  Requests above a quarter of a block get their own fill."""

    def __init__(self, device, block_bytes=8 << 20):
        self.device, self.block_bytes = torch.device(device), block_bytes
        self.buf, self.off = None, 0

    def take(self, *shape, dtype=torch.float32):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        if nbytes > self.block_bytes // 4 or self.device.type != "cuda":
            return torch.zeros(shape, dtype=dtype, device=self.device)
        if self.buf is None or self.off + nbytes > self.block_bytes:
            self.buf, self.off = torch.zeros(self.block_bytes, dtype=torch.uint8, device=self.device), 0
        v = self.buf[self.off:self.off + nbytes].view(dtype).view(shape)
        self.off += (nbytes + 255) // 256 * 256
        return v


# The packed weights are keyed on the parameters' (data_ptr, _version).  That is NOT enough for every optimizer: torch's fused
# Adam (torch._fused_adam_) updates the parameters WITHOUT bumping their version counters, so a cache keyed on them alone would
# keep rendering with the weights of the first step.  Two guards: a forward in training mode always repacks (force=True: the
# weights are expected to change between calls there), and every train() / eval() switch of the model bumps this generation,
# which is part of every cache key -- the first eval render after training packs afresh whatever the optimizer did.
_GENERATION = [0]


def invalidate_packed():
    """Forget every packed-weight cache (call after modifying parameters in a way that does not bump their _version)."""
    _GENERATION[0] += 1


def packed(module, role, ks_in):
    """Per-module cache of PackedNet objects."""
    cache = module.__dict__.setdefault("_mp_packed", {})
    dev = next(module.parameters()).device
    key = (role, ks_in, str(dev))
    if key not in cache:
        from .networks import ImplicitNet
        if isinstance(module, ImplicitNet):
            plans = implicit_grad_plans(module) if role == "grad" else implicit_plans(module, role)
        else:
            plans = rendering_plans(module)
        cache[key] = PackedNet(plans, ks_in, dev)
    return cache[key]


class PoseEmbed:
    """lin_pose(cond) of RenderingNet 'pose_no_view' (networks.py:279-280) computed by the bias path of mp_pack_layer."""

    def __init__(self, net):
        self.net = net
        dev = net.lin_pose.weight.device
        self.rowmap = torch.tensor(list(range(8)) + [-1] * 24, dtype=torch.int32, device=dev)
        self.colmap = -torch.ones(10 * 32, dtype=torch.int32, device=dev)
        self.colscale = torch.ones(10 * 32, dtype=torch.float32, device=dev)
        self.out = torch.zeros(BIAS_STRIDE, dtype=torch.float32, device=dev)

    def __call__(self, cond_vec):
        lp = self.net.lin_pose
        w, b = lp.weight.detach().contiguous(), lp.bias.detach().contiguous()
        check(lib().mp_pack_layer(ptr(w), None, ptr(b), 8, 69, ptr(self.rowmap), 32, ptr(self.colmap),
                                  ptr(self.colscale), 2, 0, 69, ptr(cond_vec), C.c_float(1.0), None, ptr(self.out),
                                  stream()),
              "mp_pack_layer(lin_pose)")
        return self.out  # first 8 floats


# ------------------------------------------------------------------------------------------------ module-level ops
def implicit_forward(net, x, cond_vec):
    """ImplicitNet.forward for external callers: (N, d_in) -> (N, 257) fp32 (features are f16-rounded)."""
    require_device()
    x = x.detach().float().contiguous()
    ks_in = 2 if net.d_in == 3 else 3
    pk = packed(net, "full", ks_in)
    pk.refresh(None if cond_vec is None else cond_vec.detach().float().contiguous())
    out = torch.empty(x.shape[0], 257, dtype=torch.float32, device=x.device)
    check(lib().mp_mlp_full(C.byref(pk.net), ptr(pk.wpack), ptr(pk.bias), ptr(x), net.d_in, x.shape[0], ptr(out),
                            stream()), "mp_mlp_full")
    return out


def implicit_sdf(net, x_c, cond_vec, mode="f16"):
    """sdf column of the foreground ImplicitNet at canonical points (N,3) -> (N,).  mode 'f16' (mp_mlp_sdf) | 'f16x2' (split
    activations on the same packed weights, mp_mlp_sdf_x2)."""
    require_device()
    x_c = x_c.detach().float().contiguous()
    pk = packed(net, "sdf", 2)
    pk.refresh(cond_vec.detach().float().contiguous())
    out = torch.empty(x_c.shape[0], dtype=torch.float32, device=x_c.device)
    fn = {"f16": lib().mp_mlp_sdf, "f16x2": lib().mp_mlp_sdf_x2}[mode]
    check(fn(C.byref(pk.net), ptr(pk.wpack), ptr(pk.bias), ptr(x_c), None, None, x_c.shape[0], ptr(out), stream()),
          "mp_mlp_sdf" + ("_x2" if mode == "f16x2" else ""))
    return out


SHADE_MODE = os.environ.get("MP_SHADE_MODE", "reverse")   # "reverse" (2 sweeps, 2 network columns per point) | "forward"


class GradNet:
    """packed reverse-sweep network + the sdf row in K-slot order of one foreground ImplicitNet (cached on the module)"""

    def __init__(self, imp):
        self.imp = imp
        self.pk = packed(imp, "grad", 0)      # no input-fed K steps in the reverse sweep
        self.w8 = None
        self.version = None

    def refresh(self):
        self.pk.refresh(None)
        if self.version != self.pk.version:
            self.w8 = sdf_row_slots(self.imp)
            self.version = self.pk.version
        return self


def grad_net(imp):
    g = imp.__dict__.get("_mp_gradnet")
    if g is None:
        g = imp.__dict__["_mp_gradnet"] = GradNet(imp)
    return g.refresh()


SEG_POINTS = int(os.environ.get("MP_SEG_POINTS", 1 << 21))   # work items per segment of the reverse-mode shading (x 2 KiB of stored sigmoids)


def sig_scratch(device, n_points):
    """(buffer, segment size) for the stored sigmoids of the reverse-mode shading kernels: mp_sig_bytes_per_point() bytes (2 KiB:
    one byte per hidden unit; 4 KiB in a -DMP_EXP_SIG16 build) per work item of ONE segment, sized to the call (min(n rounded up to
    a tile, SEG_POINTS)) and grown on demand -- never a fixed multi-GiB block.  New parts are zero-initialised: the K steps a
    narrower layer never writes (layer 3 has 217 outputs) must read as finite numbers in the reverse sweep."""
    per = int(lib().mp_sig_bytes_per_point())
    seg = min(SEG_POINTS, max(256, (int(n_points) + 255) // 256 * 256))
    key = str(device)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < seg * per:
        grown = torch.empty(seg * per, dtype=torch.uint8, device=device)
        old = 0 if buf is None else buf.numel()
        if old:
            grown[:old] = buf
        grown[old:].zero_()
        buf = _SCRATCH[key] = grown
    return buf, seg


_SCRATCH = {}


def shade_rev_launch(pki, gn, x_c, jinv, worklist, count, n, sdf, nrm, feat):
    buf, seg = sig_scratch(x_c.device, n)
    check(lib().mp_mlp_shade_rev(C.byref(pki.net), ptr(pki.wpack), ptr(pki.bias), C.byref(gn.pk.net), ptr(gn.pk.wpack),
                                 ptr(gn.w8), ptr(x_c), ptr(jinv), ptr(worklist), ptr(count), n, ptr(sdf), ptr(nrm),
                                 ptr(feat), ptr(buf), seg, stream()), "mp_mlp_shade_rev")


def shade_points(imp, ren, x_c, jinv, cond_vec, mode=None):
    """sdf, normals, rgb at canonical points (ImplicitNet value + input gradient, RenderingNet 'pose_no_view').
    mode 'reverse': mp_mlp_shade_rev (two sweeps); 'forward': the forward-mode kernel mp_mlp_shade."""
    require_device()
    n = x_c.shape[0]
    x_c = x_c.detach().float().contiguous()
    jinv = jinv.detach().float().contiguous()
    cond_vec = cond_vec.detach().float().contiguous()
    pki = packed(imp, "full", 2)
    pki.refresh(cond_vec)
    pkr = packed(ren, "color", 2)
    pe = ren.__dict__.setdefault("_mp_pose_embed", None) or PoseEmbed(ren)
    ren.__dict__["_mp_pose_embed"] = pe
    pkr.refresh(pe(cond_vec))
    dev = x_c.device
    sdf = torch.empty(n, dtype=torch.float32, device=dev)
    nrm = torch.empty(n, 3, dtype=torch.float32, device=dev)
    rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
    tiles = (n + 255) // 256 * 4
    feat = torch.empty(tiles * 8 * 4 * 1024, dtype=torch.uint8, device=dev)
    if (mode or SHADE_MODE) == "reverse":
        shade_rev_launch(pki, grad_net(imp), x_c, jinv, None, None, n, sdf, nrm, feat)
    else:
        check(lib().mp_mlp_shade(C.byref(pki.net), ptr(pki.wpack), ptr(pki.bias), ptr(x_c), ptr(jinv), None, None, n,
                                 ptr(sdf), ptr(nrm), ptr(feat), stream()), "mp_mlp_shade")
    check(lib().mp_mlp_color(C.byref(pkr.net), ptr(pkr.wpack), ptr(pkr.bias), ptr(x_c), ptr(nrm), ptr(feat), None,
                             None, n, ptr(rgb), stream()), "mp_mlp_color")
    return sdf, nrm, rgb


@functools.lru_cache(maxsize=None)
def _feat_frag_index():
    """feature index of [ks][g][e] in the colour kernel's operand-fragment order (csrc/mlp_core.hpp K permutation)"""
    idx = np.empty((8, 4, 8), dtype=np.int64)
    for ks in range(8):
        for g in range(4):
            for e in range(8):
                idx[ks, g, e] = 32 * ks + (4 * g + e if e < 4 else 16 + 4 * g + e - 4)
    return idx


def pack_feature_fragments(feat):
    """(N, 256) fp32 features -> the f16 fragment stream mp_mlp_color reads:
    [tile of 64 items][K step][16-column block][lane = column + 16 g][8 halves] (the layout k_mlp_fwdsave writes)."""
    n, dev = feat.shape[0], feat.device
    tiles = (n + 255) // 256 * 4
    f = torch.zeros(tiles * 64, 256, dtype=torch.float32, device=dev)
    f[:n] = feat
    idx = torch.from_numpy(_feat_frag_index()).to(dev)                     # (8, 4, 8)
    frag = f.reshape(tiles, 4, 16, 256)[:, :, :, idx]                      # (tile, block, j, ks, g, e)
    frag = frag.permute(0, 3, 1, 4, 2, 5).contiguous()                     # (tile, ks, block, g, j, e): lane = j + 16 g
    return frag.to(torch.float16).reshape(-1).view(torch.uint8)


def rendering_forward(net, points, normals, view_dirs, body_pose, feature_vectors, frame_latent_code):
    """RenderingNet.forward for external callers (networks.py:263-312), mode 'pose_no_view': rgb (N, 3) from canonical
    points, normals, the pose conditioning (69,) and (N, 256) feature vectors, through mp_mlp_color (features are rounded to
    f16 operands like everywhere on the inference path).  The background network ('nerf_frame_encoding') exists only fused
    into mp_background (its inputs never leave the kernel): a standalone call raises."""
    require_device()
    if net.mode != "pose_no_view":
        raise NotImplementedError("the background RenderingNet is evaluated inside mp_background only (hip.background)")
    x = points.detach().float().reshape(-1, 3).contiguous()
    nrm = normals.detach().float().reshape(-1, 3).contiguous()
    n = x.shape[0]
    cond_vec = body_pose.detach().float().reshape(-1).contiguous()
    pk = packed(net, "color", 2)
    pe = net.__dict__.get("_mp_pose_embed") or PoseEmbed(net)
    net.__dict__["_mp_pose_embed"] = pe
    pk.refresh(pe(cond_vec))
    frag = pack_feature_fragments(feature_vectors.detach().float().reshape(n, -1))
    rgb = torch.empty(n, 3, dtype=torch.float32, device=x.device)
    check(lib().mp_mlp_color(C.byref(pk.net), ptr(pk.wpack), ptr(pk.bias), ptr(x), ptr(nrm), ptr(frag), None, None, n, ptr(rgb),
                             stream()), "mp_mlp_color")
    return rgb


def background(bg_imp, bg_ren, dirs, cam, z_bg, frame_code, radius=3.0):
    """bg_rgb (R,3) of the NeRF++ background branch. z_bg: (n_bg,) shared or (R,n_bg) per-ray inverse depths, descending."""
    require_device()
    dirs = dirs.detach().float().contiguous()
    cam = cam.detach().float().contiguous()
    z_bg = z_bg.detach().float().contiguous()
    code = frame_code.detach().float().reshape(-1).contiguous()
    pki = packed(bg_imp, "full", 3)
    pki.refresh(code)
    pkr = packed(bg_ren, "color", 3)
    pkr.refresh(code)
    R = dirs.shape[0]
    if z_bg.shape[-1] != 32:
        raise NotImplementedError("the fused background kernel is specialised for the 32 inverse-sphere samples of the shipped "
                                  f"configs (N_samples_inverse_sphere), got {z_bg.shape[-1]}")
    out = torch.empty(R, 3, dtype=torch.float32, device=dirs.device)
    check(lib().mp_background(C.byref(pki.net), ptr(pki.wpack), ptr(pki.bias), C.byref(pkr.net), ptr(pkr.wpack),
                              ptr(pkr.bias), ptr(dirs), ptr(cam), ptr(z_bg), int(z_bg.dim() == 2), R,
                              C.c_float(radius), ptr(out), stream()), "mp_background")
    return out
