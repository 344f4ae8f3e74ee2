"""ctypes binding of the C ABI in include/mlhot.h.

`MlhotLib(path)` wraps ONE shared object.  The product loads
`csrc/libmlhot.so` through `mlhot.lib()`; the test-suite additionally wraps the
host-simulation flavour (tests/hostsim) with the same class to check the index
arithmetic without a GPU.  Every wrapper takes torch tensors, passes raw
`data_ptr()`s and the current HIP stream, and raises `MlhotError` on a non-zero
return code - nothing here computes anything.
"""
import ctypes as C
import os

import torch

HEADS = 8
MAX_HIDDEN = 4

ACT = {"none": 0, "relu": 1, "tanh": 2}
AGG = {"mean": 0, "max": 1, "baco": 2, "attention": 3}
LOSS = {"azimuth": 0, "mse": 1, "quaternion": 2, "degree": 3, "distractor": 4}

_f = C.c_void_p  # every device pointer travels as void*


ABI_VERSION = 7     # include/mlhot.h MLHOT_ABI_VERSION (2: + nt_xent, mt19937_normal, the *_staged entries, trunk / skinny flat gradients; 3: + conv12_fwd / _bwd; 4: + np_vanilla_bwd_loss; 5: + mt19937_advance; 6: + host_f32_to_u8_exact; 7: + loss_plus_fwd / _bwd)


class MlhotError(RuntimeError):
    pass


class EncParams(C.Structure):
    _fields_ = [(n, _f) for n in ("w1", "b1", "w2", "b2", "w3", "b3", "wl", "bl")]


class NpDims(C.Structure):
    _fields_ = [("T", C.c_int), ("Nc", C.c_int), ("Nq", C.c_int), ("label_dim", C.c_int), ("y_dim", C.c_int),
                ("dim_w", C.c_int), ("dim_r", C.c_int), ("dim_z", C.c_int), ("n_hidden", C.c_int),
                ("hidden", C.c_int * MAX_HIDDEN), ("dec_hidden", C.c_int), ("agg_mode", C.c_int),
                ("out_tanh", C.c_int), ("m_feat", C.c_int)]


class NpParams(C.Structure):
    """Mirrors mlhot_np_params AND mlhot_np_grads (same field order, grads lack `proj`)."""
    _fields_ = [("enc", EncParams), ("ty_w", _f), ("ty_b", _f),
                ("er_w", _f * (MAX_HIDDEN + 1)), ("er_b", _f * (MAX_HIDDEN + 1)),
                ("r2z_w", _f), ("r2z_b", _f), ("dec_w", _f * 3), ("dec_b", _f * 3),
                ("mu_w", _f), ("mu_b", _f), ("var_w", _f), ("var_b", _f),
                ("wk_w", _f * HEADS), ("wk_b", _f * HEADS), ("wv_w", _f * HEADS), ("wv_b", _f * HEADS),
                ("wq_w", _f * HEADS), ("wq_b", _f * HEADS), ("wo_w", _f), ("wo_b", _f), ("proj", _f)]


class NpGrads(C.Structure):
    _fields_ = NpParams._fields_[:-1]


class LossDesc(C.Structure):
    """mlhot_loss_desc: the loss whose gradient mlhot_np_vanilla_bwd_loss takes itself."""
    _fields_ = [("kind", C.c_int), ("gt", C.c_void_p), ("gt_dim", C.c_int), ("dloss", C.c_void_p), ("value", C.c_void_p)]


# (struct path, state_dict key) for the vanilla CNP/ANP family
def vanilla_param_map(n_hidden, baco, attention):
    m = [(("enc", "w1"), "encoder_w0.0.weight"), (("enc", "b1"), "encoder_w0.0.bias"),
         (("enc", "w2"), "encoder_w0.2.weight"), (("enc", "b2"), "encoder_w0.2.bias"),
         (("enc", "w3"), "encoder_w0.5.weight"), (("enc", "b3"), "encoder_w0.5.bias"),
         (("enc", "wl"), "encoder_w0.8.weight"), (("enc", "bl"), "encoder_w0.8.bias"),
         (("ty_w",), "transform_y.weight"), (("ty_b",), "transform_y.bias"),
         (("r2z_w",), "r_to_z.weight"), (("r2z_b",), "r_to_z.bias")]
    for i in range(n_hidden + 1):
        m += [(("er_w", i), f"encoder_r.layers.{2 * i}.weight"), (("er_b", i), f"encoder_r.layers.{2 * i}.bias")]
    for i in range(3):
        m += [(("dec_w", i), f"decoder0.{2 * i}.weight"), (("dec_b", i), f"decoder0.{2 * i}.bias")]
    if baco:
        m += [(("mu_w",), "rs_to_mu.weight"), (("mu_b",), "rs_to_mu.bias"),
              (("var_w",), "rs_to_var.weight"), (("var_b",), "rs_to_var.bias")]
    if attention:
        for tag, name in (("wk", "_W_k"), ("wv", "_W_v"), ("wq", "_W_q")):
            for i in range(HEADS):
                m += [((tag + "_w", i), f"{name}.{i}.linear.weight"), ((tag + "_b", i), f"{name}.{i}.linear.bias")]
        m += [(("wo_w",), "_W.linear.weight"), (("wo_b",), "_W.linear.bias")]
    return m


def _set(struct, path, ptr):
    if len(path) == 1:
        setattr(struct, path[0], ptr)
    elif isinstance(path[1], int):
        getattr(struct, path[0])[path[1]] = ptr
    else:
        setattr(getattr(struct, path[0]), path[1], ptr)


def _get(struct, path):
    if len(path) == 1:
        return getattr(struct, path[0])
    if isinstance(path[1], int):
        return getattr(struct, path[0])[path[1]]
    return getattr(getattr(struct, path[0]), path[1])


class BbbItem(C.Structure):          # struct mlhot_bbb_item
    _fields_ = [("mu", C.c_void_p), ("rho", C.c_void_p), ("eps", C.c_void_p), ("w", C.c_void_p), ("dw", C.c_void_p), ("dmu", C.c_void_p),
                ("drho", C.c_void_p), ("n", C.c_size_t), ("eps2", C.c_void_p), ("w2", C.c_void_p), ("dw2", C.c_void_p)]


class TrunkWset(C.Structure):        # struct mlhot_trunk_wset
    _fields_ = [("w", C.c_void_p * 13), ("b", C.c_void_p * 13), ("dw", C.c_void_p * 13), ("db", C.c_void_p * 13), ("skip_k", C.c_int)]


class TrunkPass(C.Structure):        # struct mlhot_trunk_pass
    _fields_ = [("img", C.c_void_p), ("n_img", C.c_int), ("wset", C.c_int), ("act", C.c_void_p * 9), ("dfeat", C.c_void_p)]


class ChainLayer(C.Structure):       # struct mlhot_chain_layer
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("K", C.c_int), ("N", C.c_int), ("act", C.c_int), ("side", C.c_void_p),
                ("side_w", C.c_int), ("side_ld", C.c_int), ("side_first", C.c_int), ("y", C.c_void_p), ("ldy", C.c_int)]


class ChainGrads(C.Structure):       # struct mlhot_chain_grads
    _fields_ = [("dw", C.c_void_p), ("db", C.c_void_p), ("g", C.c_void_p), ("ldg", C.c_int), ("dside", C.c_void_p),
                ("dside_ld", C.c_int), ("dside_accumulate", C.c_int)]


class LinearJob(C.Structure):        # struct mlhot_linear_job
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int), ("w", C.c_void_p), ("b", C.c_void_p), ("y", C.c_void_p), ("ldy", C.c_int),
                ("M", C.c_int), ("K", C.c_int), ("N", C.c_int), ("act", C.c_int), ("dy", C.c_void_p), ("lddy", C.c_int),
                ("dx", C.c_void_p), ("lddx", C.c_int), ("dx_accumulate", C.c_int), ("dw", C.c_void_p), ("db", C.c_void_p),
                ("x2", C.c_void_p), ("ldx2", C.c_int), ("K2", C.c_int), ("dx2", C.c_void_p), ("lddx2", C.c_int)]


_GRAD_ARENA = None


def set_grad_arena(arena):
    """Install (or, with None, remove) the flat gradient buffer the weight / bias gradients of the following calls are written
    into (mlhot/arena.py).  Process-global, like the parameters it mirrors: one model trains at a time."""
    global _GRAD_ARENA
    _GRAD_ARENA = arena


def get_grad_arena():
    return _GRAD_ARENA


def _grad_like(t):
    """Storage for the gradient of `t`: its slot in the installed arena, or a fresh tensor."""
    if _GRAD_ARENA is not None:
        v = _GRAD_ARENA.slot(t)
        if v is not None:
            return v
    return torch.empty_like(t)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _addr(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None


def _chk(*ts):
    for t in ts:
        if t is not None and (t.dtype not in (torch.float32, torch.int32, torch.uint8) or not t.is_contiguous()):
            raise MlhotError(f"mlhot expects contiguous fp32 tensors, got {t.dtype} contiguous={t.is_contiguous()}")


class MlhotLib:
    _last = None      # the most recently loaded library (size queries of the static test helpers)

    def __init__(self, path):
        if not os.path.exists(path):
            raise MlhotError(f"mlhot: shared library not found: {path} (run __graft_entry__.build())")
        self.path = path
        self.c = C.CDLL(path)
        MlhotLib._last = self
        c = self.c
        c.mlhot_version.restype = C.c_int
        if c.mlhot_version() != ABI_VERSION:       # before any other symbol is touched: an older prebuilt library lacks the newer ones
            raise MlhotError(f"mlhot: libmlhot.so has ABI version {c.mlhot_version()}, this binding needs {ABI_VERSION} "
                             f"(include/mlhot.h MLHOT_ABI_VERSION) - rebuild with mlhot.build.build_product(force=True)")
        c.mlhot_last_error.restype = C.c_char_p
        for fn in ("mlhot_enc_vanilla_saved_bytes", "mlhot_enc_vanilla_scratch_bytes", "mlhot_linear_bwd_scratch_bytes",
                   "mlhot_favor_ws_bytes", "mlhot_np_struct_bytes", "mlhot_np_saved_bytes", "mlhot_np_scratch_bytes",
                   "mlhot_np_grads_flat_layout", "mlhot_conv12_scratch_bytes"):
            getattr(c, fn).restype = C.c_size_t
        c.mlhot_conv12_scratch_bytes.argtypes = [C.c_int]
        c.mlhot_enc_vanilla_saved_bytes.argtypes = [C.c_int]
        c.mlhot_enc_vanilla_scratch_bytes.argtypes = [C.c_int, C.c_int]
        c.mlhot_favor_ws_bytes.argtypes = [C.c_int] * 6
        c.mlhot_np_struct_bytes.argtypes = [C.c_int]
        c.mlhot_np_saved_bytes.argtypes = [C.POINTER(NpDims)]
        c.mlhot_np_scratch_bytes.argtypes = [C.POINTER(NpDims)]
        c.mlhot_np_grads_flat_layout.argtypes = [C.POINTER(NpDims), C.POINTER(NpGrads)]
        i, z, P = C.c_int, C.c_size_t, C.c_void_p
        c.mlhot_enc_vanilla_fwd.argtypes = [P, i, P, i, C.POINTER(EncParams), i, P, i, P, i, P, P, z, P]
        c.mlhot_enc_vanilla_bwd.argtypes = [P, i, P, i, C.POINTER(EncParams), i, P, i, P, i, P, C.POINTER(EncParams), P, z, P]
        c.mlhot_conv12_fwd.argtypes = [P, i, P, P, P, P, P, P]
        c.mlhot_conv12_bwd.argtypes = [P, i, P, P, P, P, P, P, P, P, P, P, z, P]
        c.mlhot_linear_fwd.argtypes = [P, i, P, P, P, i, i, i, i, i, P]
        c.mlhot_linear_bwd.argtypes = [P, i, P, P, i, P, i, i, i, i, i, P, i, i, P, P, P, z, P]
        c.mlhot_agg_fwd.argtypes = [i, P, P, i, i, i, P, P, P, P]
        c.mlhot_agg_bwd.argtypes = [i, P, P, P, P, P, P, i, i, i, P, P, P]
        c.mlhot_favor_fwd.argtypes = [P, P, P, P, i, i, i, i, i, i, P, P, z, P]
        c.mlhot_favor_bwd.argtypes = [P, P, P, P, i, i, i, i, i, i, P, P, P, P, P, P, z, P]
        c.mlhot_favor_fwd_staged.argtypes = [P, P, P, P, i, i, i, i, i, i, P, P, z, i, P, P]
        c.mlhot_favor_bwd_staged.argtypes = [P, P, P, P, i, i, i, i, i, i, P, P, P, P, P, P, z, i, P, P]
        c.mlhot_loss_fwd.argtypes = [i, P, P, i, i, i, P, P]
        c.mlhot_loss_bwd.argtypes = [i, P, P, i, i, i, P, P, P]
        c.mlhot_loss_plus_fwd.argtypes = [i, P, P, i, i, i, P, C.c_float, P, P, P]
        c.mlhot_loss_plus_bwd.argtypes = [i, P, P, i, i, i, P, C.c_float, P, P, P]
        c.mlhot_conv2d_bwd_scratch_bytes.restype = C.c_size_t
        c.mlhot_conv2d_bwd_scratch_bytes.argtypes = [i] * 8
        c.mlhot_conv2d_fwd.argtypes = [P, P, P, P] + [i] * 9 + [P]
        c.mlhot_conv2d_bwd.argtypes = [P, P, P, P] + [i] * 9 + [P, P, P, P, z, P]
        c.mlhot_add_relu_fwd.argtypes = [P, P, P, z, P]
        c.mlhot_add_relu_bwd.argtypes = [P, P, P, z, P]
        c.mlhot_pool2_fwd.argtypes = [P, P, P, i, i, i, P]
        c.mlhot_pool2_bwd.argtypes = [P, P, P, i, i, i, P]
        c.mlhot_bbb_sample_fwd.argtypes = [P, P, P, P, P, P, z, P]
        c.mlhot_bbb_sample_bwd.argtypes = [P, P, P, P, P, P, P, z, P]
        f32 = C.c_float
        c.mlhot_bn_relu_fwd.argtypes = [P, P, P, P, P, f32, f32, i, i, i, P, P, P, P]
        c.mlhot_bn_relu_bwd.argtypes = [P, P, P, P, P, P, f32, i, i, i, P, P, P, P]
        c.mlhot_spatial_mean_fwd.argtypes = [P, P, i, i, P]
        c.mlhot_spatial_mean_bwd.argtypes = [P, P, i, i, P]
        c.mlhot_np_vanilla_fwd.argtypes = [C.POINTER(NpDims), C.POINTER(NpParams), P, P, P, P, P, P, z, P]
        c.mlhot_np_vanilla_bwd.argtypes = [C.POINTER(NpDims), C.POINTER(NpParams), P, P, P, P, P, C.POINTER(NpGrads), P, P, z, P]
        c.mlhot_np_vanilla_fwd_staged.argtypes = [C.POINTER(NpDims), C.POINTER(NpParams), P, P, P, P, P, P, z, i, P, P]
        c.mlhot_np_vanilla_bwd_loss.argtypes = [C.POINTER(NpDims), C.POINTER(NpParams), P, P, P, P, P, C.POINTER(LossDesc), C.POINTER(NpGrads), P, P, z, P]
        c.mlhot_np_vanilla_bwd_staged.argtypes = [C.POINTER(NpDims), C.POINTER(NpParams), P, P, P, P, P, C.POINTER(NpGrads), P, P, z, i, P, P]
        c.mlhot_mlp_chain_fwd.argtypes = [P, i, i, C.POINTER(ChainLayer), i, P]
        c.mlhot_mlp_chain_bwd.argtypes = [P, i, i, C.POINTER(ChainLayer), C.POINTER(ChainGrads), i, P, i, P, i, i, P]
        c.mlhot_linear_multi_fwd.argtypes = [C.POINTER(LinearJob), i, P]
        c.mlhot_linear_multi_bwd.argtypes = [C.POINTER(LinearJob), i, P]
        for which, st in ((0, NpDims), (1, NpParams), (2, NpGrads), (3, ChainLayer), (4, ChainGrads), (5, LinearJob)):
            if c.mlhot_np_struct_bytes(which) != C.sizeof(st):
                raise MlhotError(f"mlhot: ABI struct size mismatch for {st.__name__}")

    # ------------------------------------------------------------------------------------------
    def _rc(self, rc, what):
        if rc != 0:
            raise MlhotError(f"{what} failed (code {rc}): {self.c.mlhot_last_error().decode()}")

    @staticmethod
    def _bytes(n, like):
        return torch.empty(max(int(n), 256), dtype=torch.uint8, device=like.device)

    def set_option(self, name, value):
        self.c.mlhot_set_option.argtypes = [C.c_char_p, C.c_int]
        self._rc(self.c.mlhot_set_option(name.encode(), int(value)), "mlhot_set_option")

    # ---- bench-only launch profiler -----------------------------------------------------------
    def prof_begin(self, max_records=4096):
        self._prof_cap = max_records
        self.c.mlhot_prof_begin.argtypes = [C.c_int]
        self._rc(self.c.mlhot_prof_begin(max_records), "mlhot_prof_begin")

    def prof_end(self):
        """-> list of (label, ms) per kernel launch since prof_begin()"""
        cap = self._prof_cap
        labels, ms = (C.c_char_p * cap)(), (C.c_float * cap)()
        self.c.mlhot_prof_end.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
        n = self.c.mlhot_prof_end(labels, ms, cap)
        return [(labels[i].decode(), float(ms[i])) for i in range(n)]

    # ---- E1 ------------------------------------------------------------------------------------
    @staticmethod
    def enc_struct(tensors):
        """tensors: (w1,b1,w2,b2,w3,b3,wl,bl)"""
        s = EncParams()
        for name, t in zip(("w1", "b1", "w2", "b2", "w3", "b3", "wl", "bl"), tensors):
            setattr(s, name, t.data_ptr())
        return s

    def enc_vanilla_fwd(self, img0, img1, params, dim_w):
        """img0 [n0,1,128,128], img1 [n1,1,128,128] or None -> feat0 [n0,dim_w], feat1, saved"""
        n0 = img0.shape[0]
        n1 = 0 if img1 is None else img1.shape[0]
        _chk(img0, img1, *params)
        n = n0 + n1
        feat0 = torch.empty(n0, dim_w, device=img0.device)
        feat1 = torch.empty(n1, dim_w, device=img0.device)
        saved = self._bytes(self.c.mlhot_enc_vanilla_saved_bytes(n), img0)
        sb = self.c.mlhot_enc_vanilla_scratch_bytes(n, dim_w)
        scratch = self._bytes(sb, img0)
        ps = self.enc_struct(params)
        self._rc(self.c.mlhot_enc_vanilla_fwd(_ptr(img0), n0, _ptr(img1), n1, C.byref(ps), dim_w, _ptr(feat0), dim_w,
                                              _ptr(feat1), dim_w, _ptr(saved), _ptr(scratch), sb, _stream(img0)),
                 "mlhot_enc_vanilla_fwd")
        return feat0, feat1, saved

    def conv12_fwd(self, img, w1, b1, w2, b2):
        """The encoder's first block on its own (conv1 + ReLU + conv2 + ReLU + 2x2 max-pool of [n,1,128,128] images) ->
        (p2 [n,48,16,16], arg-max uint8 [n,48,16,16], saved); `saved` as enc_vanilla_fwd's (enc_saved_views / enc_routes)."""
        _chk(img, w1, b1, w2, b2)
        n = img.shape[0]
        saved = self._bytes(self.c.mlhot_enc_vanilla_saved_bytes(n), img)
        self._rc(self.c.mlhot_conv12_fwd(_ptr(img), n, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(saved), _stream(img)),
                 "mlhot_conv12_fwd")
        _, p2, am2, _ = self.enc_saved_views(saved, n)
        return p2, am2, saved

    def conv12_bwd(self, img, w1, b1, w2, dp2, saved):
        """d p2 [n,48,16,16] -> (dw1, db1, dw2, db2) of the block."""
        _chk(img, w1, b1, w2, dp2)
        n = img.shape[0]
        dw1, db1, dw2 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2)
        db2 = torch.empty(48, device=img.device)
        sb = self.c.mlhot_conv12_scratch_bytes(n)
        scratch = self._bytes(sb, img)
        self._rc(self.c.mlhot_conv12_bwd(_ptr(img), n, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(dp2), _ptr(saved), _ptr(dw1), _ptr(db1),
                                         _ptr(dw2), _ptr(db2), _ptr(scratch), sb, _stream(img)), "mlhot_conv12_bwd")
        return dw1, db1, dw2, db2

    @staticmethod
    def enc_saved_views(saved, n):
        """Test / diagnostic helper: the activations the encoder forward kept for its backward
        (csrc/encoder.h EncSaved: 256-byte aligned a1 | p2 | am2 | a3) as tensor views."""
        def al(x):
            return (x + 255) // 256 * 256
        o = 0
        a1 = saved[o:o + n * 32 * 4096 * 4].view(torch.float32).view(n, 32, 64, 64)
        o = al(o + n * 32 * 4096 * 4)
        p2 = saved[o:o + n * 48 * 256 * 4].view(torch.float32).view(n, 48, 16, 16)
        o = al(o + n * 48 * 256 * 4)
        am2 = saved[o:o + n * 48 * 256].view(n, 48, 16, 16)
        o = al(o + n * 48 * 256)
        a3 = saved[o:o + n * 4096 * 4].view(torch.float32).view(n, 64, 8, 8)
        return a1, p2, am2, a3

    @staticmethod
    def enc_routes(saved, n):
        """Test / diagnostic helper: the piecewise-linear routing decisions the fused encoder forward took, in the form
        oracle.ref_cpu.vanilla_encoder_routed takes them: (conv1 ReLU mask [n,32,64,64] unpacked from the forward's sign-bit
        words, pool arg-max [n,48,16,16], pooled-conv2 ReLU mask, conv3 ReLU mask) as CPU tensors."""
        def al(x):
            return (x + 255) // 256 * 256
        _, p2, am2, a3 = MlhotLib.enc_saved_views(saved, n)
        o = al(n * 32 * 4096 * 4)
        o = al(o + n * 48 * 256 * 4)
        o = al(o + n * 48 * 256)
        o = al(o + n * 4096 * 4)
        # records [img][row][column group cg][dword 4r + 2h + g], bit 16e + c = channel 16h + c of column 16cg + 4(2g + e) + r
        # (csrc/conv_tc.h m1_record)
        words = saved[o:o + n * 4096 * 4].view(torch.int32).view(n, 64, 4, 4, 2, 2, 1, 1).cpu()      # n, y, cg, r, h, g
        shifts = (16 * torch.arange(2, dtype=torch.int32).view(2, 1) + torch.arange(16, dtype=torch.int32).view(1, 16))
        bits = (words >> shifts) & 1                                                                # n, y, cg, r, h, g, e, c
        m1 = bits.permute(0, 4, 7, 1, 2, 5, 6, 3).reshape(n, 32, 64, 64).float()                   # n, (h c), y, (cg g e r)
        return m1, am2.cpu(), (p2 > 0).float().cpu(), (a3 > 0).float().cpu()

    @staticmethod
    def np_saved_views(saved, dims):
        """Test / diagnostic helper: activations mlhot_np_vanilla_fwd kept (csrc/np_vanilla.h np_saved_carve) as tensor views:
        the encoder blob (first; see enc_routes) and the task-side layers' post-ReLU outputs."""
        def al(x):
            return (x + 255) // 256 * 256
        Rc, Rq = dims.T * dims.Nc, dims.T * dims.Nq
        n, dw = Rc + Rq, dims.dim_w
        lib = MlhotLib._last
        o = al(lib.c.mlhot_enc_vanilla_saved_bytes(n))

        def take(rows, cols, dtype=torch.float32):
            nonlocal o
            v = saved[o:o + rows * cols * 4].view(dtype).view(rows, cols)
            o = al(o + rows * cols * 4)
            return v
        out = {"enc": saved, "n": n}
        out["dec_in"] = take(Rq, dw + dims.dim_z)
        out["d1"], out["d2"] = take(Rq, dims.dec_hidden), take(Rq, dims.dec_hidden)
        if dims.Nc > 0:
            out["cat_in"] = take(Rc, dw + dw // 4)
            out["h"] = [take(Rc, dims.hidden[i]) for i in range(dims.n_hidden)]
            out["rs"] = take(Rc, dims.dim_r)
            if dims.agg_mode != AGG["attention"]:
                out["r"], out["zt"] = take(dims.T, dims.dim_r), take(dims.T, dims.dim_z)
                out["amax"] = take(dims.T, dims.dim_r, torch.int32)
        return out

    def enc_vanilla_bwd(self, img0, img1, params, dim_w, dfeat0, dfeat1, saved):
        n0 = img0.shape[0]
        n1 = 0 if img1 is None else img1.shape[0]
        _chk(dfeat0, dfeat1)
        grads = [torch.empty_like(p) for p in params]
        sb = self.c.mlhot_enc_vanilla_scratch_bytes(n0 + n1, dim_w)
        scratch = self._bytes(sb, img0)
        ps, gs = self.enc_struct(params), self.enc_struct(grads)
        self._rc(self.c.mlhot_enc_vanilla_bwd(_ptr(img0), n0, _ptr(img1), n1, C.byref(ps), dim_w, _ptr(dfeat0), dim_w,
                                              _ptr(dfeat1), dim_w, _ptr(saved), C.byref(gs), _ptr(scratch), sb, _stream(img0)),
                 "mlhot_enc_vanilla_bwd")
        return grads

    # ---- E2 / D2 / B1 building blocks ----------------------------------------------------------
    def conv2d_fwd(self, x, w, b, stride, pad, relu):
        _chk(x, w, b)
        N, Cin, H, W = x.shape
        Cout, _, k, _ = w.shape
        HO, WO = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = torch.empty(N, Cout, HO, WO, device=x.device)
        self._rc(self.c.mlhot_conv2d_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), N, Cin, H, W, Cout, k, stride, pad, int(relu), _stream(x)),
                 "mlhot_conv2d_fwd")
        return y

    def conv2d_bwd(self, x, w, y, dy, stride, pad, relu, need_dx=True, has_bias=True):
        _chk(x, w, y, dy)
        N, Cin, H, W = x.shape
        Cout, _, k, _ = w.shape
        dx = torch.empty_like(x) if need_dx else None
        dw = torch.empty_like(w)
        db = torch.empty(Cout, device=x.device) if has_bias else None
        sb = self.c.mlhot_conv2d_bwd_scratch_bytes(N, Cin, H, W, Cout, k, stride, pad)
        scratch = self._bytes(sb, x)
        self._rc(self.c.mlhot_conv2d_bwd(_ptr(x), _ptr(w), _ptr(y), _ptr(dy), N, Cin, H, W, Cout, k, stride, pad, int(relu),
                                         _ptr(dx), _ptr(dw), _ptr(db), _ptr(scratch), sb, _stream(x)), "mlhot_conv2d_bwd")
        return dx, dw, db

    def add_relu_fwd(self, a, b):
        _chk(a, b)
        y = torch.empty_like(a)
        self._rc(self.c.mlhot_add_relu_fwd(_ptr(a), _ptr(b), _ptr(y), a.numel(), _stream(a)), "mlhot_add_relu_fwd")
        return y

    def add_relu_bwd(self, y, dy):
        _chk(y, dy)
        g = torch.empty_like(y)
        self._rc(self.c.mlhot_add_relu_bwd(_ptr(y), _ptr(dy), _ptr(g), y.numel(), _stream(y)), "mlhot_add_relu_bwd")
        return g

    def pool2_fwd(self, x):
        _chk(x)
        N, Cc, H, W = x.shape
        y = torch.empty(N, Cc, H // 2, W // 2, device=x.device)
        amax = torch.empty(N, Cc, H // 2, W // 2, dtype=torch.uint8, device=x.device)
        self._rc(self.c.mlhot_pool2_fwd(_ptr(x), _ptr(y), _ptr(amax), N * Cc, H, W, _stream(x)), "mlhot_pool2_fwd")
        return y, amax

    def pool2_bwd(self, dy, amax, H, W):
        _chk(dy, amax)
        N, Cc = dy.shape[:2]
        dx = torch.empty(N, Cc, H, W, device=dy.device)
        self._rc(self.c.mlhot_pool2_bwd(_ptr(dy), _ptr(amax), _ptr(dx), N * Cc, H, W, _stream(dy)), "mlhot_pool2_bwd")
        return dx

    def bbb_sample_fwd(self, mu, rho, eps):
        _chk(mu, rho, eps)
        w, klterm = torch.empty_like(mu), torch.empty_like(mu)
        kl = torch.empty((), device=mu.device)
        self._rc(self.c.mlhot_bbb_sample_fwd(_ptr(mu), _ptr(rho), _ptr(eps), _ptr(w), _ptr(klterm), _ptr(kl), mu.numel(), _stream(mu)),
                 "mlhot_bbb_sample_fwd")
        return w, kl

    def bbb_sample_bwd(self, mu, rho, eps, dw, dkl):
        _chk(dw, dkl)
        dmu, drho = _grad_like(mu), _grad_like(rho)
        self._rc(self.c.mlhot_bbb_sample_bwd(_ptr(mu), _ptr(rho), _ptr(eps), _ptr(dw), _ptr(dkl), _ptr(dmu), _ptr(drho), mu.numel(), _stream(mu)),
                 "mlhot_bbb_sample_bwd")
        return dmu, drho

    def _bbb_items(self, mus, rhos, epss, ws=None, dws=None, dmus=None, drhos=None, epss2=None, ws2=None, dws2=None):
        n = len(mus)
        if n > 32:
            raise MlhotError("bbb_sample_multi: at most 32 tensors per call")
        arr = (BbbItem * max(n, 1))()
        for i in range(n):
            it = arr[i]
            it.mu, it.rho, it.eps = mus[i].data_ptr(), rhos[i].data_ptr(), epss[i].data_ptr()
            it.w = ws[i].data_ptr() if ws is not None else None
            it.dw = dws[i].data_ptr() if dws is not None and dws[i] is not None else None
            it.dmu = dmus[i].data_ptr() if dmus is not None else None
            it.drho = drhos[i].data_ptr() if drhos is not None else None
            it.n = mus[i].numel()
            it.eps2 = epss2[i].data_ptr() if epss2 is not None else None
            it.w2 = ws2[i].data_ptr() if ws2 is not None else None
            it.dw2 = dws2[i].data_ptr() if dws2 is not None and dws2[i] is not None else None
        return arr

    def bbb_sample_multi_fwd(self, mus, rhos, epss, epss2=None):
        """Every (mu, rho, eps) triple sampled in ONE launch: returns ([w_i], kl = sum of all KL terms); with `epss2` a second
        independent sample of every tensor as well: ([w_i], [w2_i], kl)."""
        _chk(*mus, *rhos, *epss, *(epss2 or []))
        ws = [torch.empty_like(m) for m in mus]
        ws2 = [torch.empty_like(m) for m in mus] if epss2 is not None else None
        kl = torch.empty((), device=mus[0].device)
        items = self._bbb_items(mus, rhos, epss, ws=ws, epss2=epss2, ws2=ws2)
        self.c.mlhot_bbb_sample_multi_scratch_floats.restype = C.c_size_t
        partial = torch.empty(self.c.mlhot_bbb_sample_multi_scratch_floats(items, len(mus)), device=mus[0].device)
        self._rc(self.c.mlhot_bbb_sample_multi_fwd(items, len(mus), _ptr(partial), _ptr(kl), _stream(mus[0])), "mlhot_bbb_sample_multi_fwd")
        return (ws, kl) if epss2 is None else (ws, ws2, kl)

    def bbb_sample_multi_bwd(self, mus, rhos, epss, dws, dkl, epss2=None, dws2=None):
        _chk(*[d for d in dws if d is not None], dkl, *[d for d in (dws2 or []) if d is not None])
        dmus, drhos = [_grad_like(m) for m in mus], [_grad_like(r) for r in rhos]
        items = self._bbb_items(mus, rhos, epss, dws=dws, dmus=dmus, drhos=drhos, epss2=epss2, dws2=dws2)
        self._rc(self.c.mlhot_bbb_sample_multi_bwd(items, len(mus), _ptr(dkl), _stream(mus[0])), "mlhot_bbb_sample_multi_bwd")
        return dmus, drhos

    # ---- NT-Xent -----------------------------------------------------------------------------------
    def nt_xent_fwd(self, z, div, mod, t):
        """z [N, d] -> (loss scalar tensor, ws kept for the backward)."""
        _chk(z)
        if not z.is_cuda:
            raise MlhotError("mlhot_nt_xent_fwd: device tensors only")
        N, d = z.shape
        self.c.mlhot_nt_xent_ws_floats.restype = C.c_size_t
        self.c.mlhot_nt_xent_ws_floats.argtypes = [C.c_int]
        ws = torch.empty(self.c.mlhot_nt_xent_ws_floats(N), device=z.device)
        loss = torch.empty((), device=z.device)
        self.c.mlhot_nt_xent_fwd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        self._rc(self.c.mlhot_nt_xent_fwd(_ptr(z), N, d, div, mod, t, _ptr(ws), _ptr(loss), _stream(z)), "mlhot_nt_xent_fwd")
        return loss, ws

    def nt_xent_bwd(self, z, div, mod, t, ws, dloss):
        _chk(z, ws, dloss)
        N, d = z.shape
        dz = torch.empty_like(z)
        self.c.mlhot_nt_xent_bwd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]
        self._rc(self.c.mlhot_nt_xent_bwd(_ptr(z), N, d, div, mod, t, _ptr(ws), _ptr(dloss), _ptr(dz), _stream(z)), "mlhot_nt_xent_bwd")
        return dz

    # ---- torch's CPU normal_() stream on the device ---------------------------------------------
    def mt19937_normal(self, engine, uniform_ws, out, segs, nseg, total_outputs, total_groups):
        """engine: int32 [626] device tensor (state, left, next), advanced in place; segs: int64 [nseg, 4] device tensor."""
        if not (engine.is_cuda and uniform_ws.is_cuda and out.is_cuda and segs.is_cuda):
            raise MlhotError("mlhot_mt19937_normal: device tensors only")
        if engine.dtype != torch.int32 or engine.numel() != 626 or segs.dtype != torch.int64 or out.dtype != torch.float32:
            raise MlhotError("mlhot_mt19937_normal: engine int32[626], segs int64[nseg, 4], out float32")
        self.c.mlhot_mt19937_normal.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
        self._rc(self.c.mlhot_mt19937_normal(_ptr(engine), _ptr(uniform_ws), _ptr(out), _ptr(segs), nseg, total_outputs, total_groups,
                                             _stream(out)), "mlhot_mt19937_normal")

    def mt19937_normal_par(self, engine, uniform_ws, out, segs, nseg, total_outputs, total_groups, polys, n_sub, stride_blocks, jump_ws):
        """The same draw from n_sub jump-ahead sub-streams (polys: int32 [n_sub - 1, 624] device tensor from mlhot.mt_jump)."""
        if not (engine.is_cuda and uniform_ws.is_cuda and out.is_cuda and segs.is_cuda and polys.is_cuda and jump_ws.is_cuda):
            raise MlhotError("mlhot_mt19937_normal_par: device tensors only")
        if engine.dtype != torch.int32 or engine.numel() != 626 or segs.dtype != torch.int64 or out.dtype != torch.float32 or \
                polys.dtype != torch.int32 or tuple(polys.shape) != (n_sub - 1, 624) or jump_ws.dtype != torch.int32:
            raise MlhotError("mlhot_mt19937_normal_par: engine int32[626], segs int64[nseg, 4], out float32, polys int32[n_sub - 1, 624]")
        self.c.mlhot_mt19937_jump_ws_words.restype = C.c_size_t
        self.c.mlhot_mt19937_jump_ws_words.argtypes = [C.c_int]
        if jump_ws.numel() < self.c.mlhot_mt19937_jump_ws_words(n_sub):
            raise MlhotError("mlhot_mt19937_normal_par: jump workspace too small")
        self.c.mlhot_mt19937_normal_par.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self._rc(self.c.mlhot_mt19937_normal_par(_ptr(engine), _ptr(uniform_ws), _ptr(out), _ptr(segs), nseg, total_outputs, total_groups,
                                                 _ptr(polys), n_sub, stride_blocks, _ptr(jump_ws), _stream(out)), "mlhot_mt19937_normal_par")

    def mt19937_jump_ws_words(self, n_sub):
        self.c.mlhot_mt19937_jump_ws_words.restype = C.c_size_t
        self.c.mlhot_mt19937_jump_ws_words.argtypes = [C.c_int]
        return int(self.c.mlhot_mt19937_jump_ws_words(n_sub))

    def host_f32_to_u8_exact(self, src_ptr, dst_ptr, n, div=255.0, threads=1):
        """Raw host pointers (ints), n elements: bytes into dst, returns the number of elements that do NOT round-trip through
        (float)byte / div bit for bit.  Host only; `threads` native threads inside the call (the GIL is released for its duration)."""
        self.c.mlhot_host_f32_to_u8_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int, C.POINTER(C.c_int64)]
        bad = C.c_int64(0)
        self._rc(self.c.mlhot_host_f32_to_u8_exact(C.c_void_p(src_ptr), C.c_void_p(dst_ptr), int(n), float(div), int(threads), C.byref(bad)),
                 "mlhot_host_f32_to_u8_exact")
        return int(bad.value)

    def mt19937_advance(self, engine, n_outputs):
        """engine: numpy uint32[626] (state, left, next), advanced IN PLACE by n_outputs calls.  Host only: works without a GPU."""
        import numpy as np
        if not (isinstance(engine, np.ndarray) and engine.dtype == np.uint32 and engine.size == 626 and engine.flags.c_contiguous and engine.flags.writeable):
            raise MlhotError("mt19937_advance: engine must be a writeable contiguous numpy uint32[626]")
        self.c.mlhot_mt19937_advance.argtypes = [C.c_void_p, C.c_uint64]
        self._rc(self.c.mlhot_mt19937_advance(C.c_void_p(engine.ctypes.data), int(n_outputs)), "mlhot_mt19937_advance")
        return engine

    # ---- whole ResNet trunks -------------------------------------------------------------------
    @staticmethod
    def trunk_supported(C_, H):
        return (C_, H) in ((3, 64), (1, 128))

    @staticmethod
    def _trunk_structs(passes, wsets, grads=None, dfeats=None):
        """passes: [(img [n,C,H,H], wset index, [9 activation tensors])]; wsets: [([w0, b0, w1, b1, ...], skip_k)]"""
        wa = (TrunkWset * len(wsets))()
        for i, (tensors, skip_k) in enumerate(wsets):
            for c in range(13):
                wa[i].w[c], wa[i].b[c] = tensors[2 * c].data_ptr(), tensors[2 * c + 1].data_ptr()
                if grads is not None:
                    wa[i].dw[c], wa[i].db[c] = grads[i][2 * c].data_ptr(), grads[i][2 * c + 1].data_ptr()
            wa[i].skip_k = skip_k
        pa = (TrunkPass * len(passes))()
        for i, (img, wset, acts) in enumerate(passes):
            pa[i].img, pa[i].n_img, pa[i].wset = img.data_ptr(), img.shape[0], wset
            for k in range(9):
                pa[i].act[k] = acts[k].data_ptr()
            if dfeats is not None:
                pa[i].dfeat = dfeats[i].data_ptr()
        return pa, wa

    def trunk_acts(self, img):
        """The nine saved-activation tensors of one pass (a0, then (mid_i, y_i) of the four blocks) as views of one buffer."""
        n, C_, H, _ = img.shape
        self.c.mlhot_trunk_act_floats.restype = C.c_size_t
        self.c.mlhot_trunk_act_floats.argtypes = [C.c_int] * 4
        sizes = [self.c.mlhot_trunk_act_floats(C_, H, n, k) for k in range(9)]
        offs, tot = [], 0
        for sz in sizes:
            offs.append(tot)
            tot += (sz + 63) // 64 * 64
        flat = torch.empty(tot, device=img.device)
        out = []
        for k, (o, sz) in enumerate(zip(offs, sizes)):
            side = H // 2 if k == 0 else H >> ((k + 1) // 2 + 1)
            out.append(flat[o:o + sz].view(n, 64, side, side))
        return out

    def _trunk_scratch(self, pa, n_pass, wa, n_wset, C_, H, backward, like):
        self.c.mlhot_trunk_scratch_bytes.restype = C.c_size_t
        self.c.mlhot_trunk_scratch_bytes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        sb = self.c.mlhot_trunk_scratch_bytes(pa, n_pass, wa, n_wset, C_, H, backward)
        if sb == 0:
            raise MlhotError(f"mlhot_trunk: {self.c.mlhot_last_error().decode()}")
        return sb, self._bytes(sb, like)

    def trunk_fwd(self, passes, wsets):
        """passes: [(img, wset index)], wsets: [([w0, b0, ..., w12, b12], skip_k)] -> per pass the list of its 9 activations
        (the last one is the trunk's output map)."""
        imgs = [p[0] for p in passes]
        _chk(*imgs, *[t for ts, _ in wsets for t in ts])
        C_, H = imgs[0].shape[1], imgs[0].shape[2]
        full = [(img, w, self.trunk_acts(img)) for img, w in passes]
        pa, wa = self._trunk_structs(full, wsets)
        sb, scratch = self._trunk_scratch(pa, len(full), wa, len(wsets), C_, H, 0, imgs[0])
        self.c.mlhot_trunk_fwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        self._rc(self.c.mlhot_trunk_fwd(pa, len(full), wa, len(wsets), C_, H, _ptr(scratch), sb, _stream(imgs[0])), "mlhot_trunk_fwd")
        return [acts for _, _, acts in full]

    def trunk_bwd(self, passes, wsets, dfeats):
        """passes: [(img, wset index, acts)], dfeats: gradient wrt each pass's output map -> per weight set its 26 gradients."""
        _chk(*dfeats)
        imgs = [p[0] for p in passes]
        C_, H = imgs[0].shape[1], imgs[0].shape[2]
        grads = [[_grad_like(t) for t in ts] for ts, _ in wsets]
        pa, wa = self._trunk_structs(passes, wsets, grads=grads, dfeats=dfeats)
        sb, scratch = self._trunk_scratch(pa, len(passes), wa, len(wsets), C_, H, 1, imgs[0])
        self.c.mlhot_trunk_bwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        self._rc(self.c.mlhot_trunk_bwd(pa, len(passes), wa, len(wsets), C_, H, _ptr(scratch), sb, _stream(imgs[0])), "mlhot_trunk_bwd")
        return grads

    # ---- X1 building blocks --------------------------------------------------------------------
    def bn_relu_fwd(self, x, gamma, beta, run_mean, run_var, momentum=0.1, eps=1e-5):
        _chk(x, gamma, beta, run_mean, run_var)
        N, Cc = x.shape[:2]
        HW = x.numel() // (N * Cc)
        y = torch.empty_like(x)
        mean, var = torch.empty(Cc, device=x.device), torch.empty(Cc, device=x.device)
        self._rc(self.c.mlhot_bn_relu_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(run_mean), _ptr(run_var), momentum, eps, N, Cc, HW,
                                          _ptr(y), _ptr(mean), _ptr(var), _stream(x)), "mlhot_bn_relu_fwd")
        return y, mean, var

    def bn_relu_bwd(self, x, y, dy, gamma, mean, var, eps=1e-5):
        _chk(dy)
        N, Cc = x.shape[:2]
        HW = x.numel() // (N * Cc)
        dx, dgamma, dbeta = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
        self._rc(self.c.mlhot_bn_relu_bwd(_ptr(x), _ptr(y), _ptr(dy), _ptr(gamma), _ptr(mean), _ptr(var), eps, N, Cc, HW,
                                          _ptr(dx), _ptr(dgamma), _ptr(dbeta), _stream(x)), "mlhot_bn_relu_bwd")
        return dx, dgamma, dbeta

    def spatial_mean_fwd(self, x):
        _chk(x)
        N, Cc = x.shape[:2]
        HW = x.numel() // (N * Cc)
        y = torch.empty(N, Cc, device=x.device)
        self._rc(self.c.mlhot_spatial_mean_fwd(_ptr(x), _ptr(y), N * Cc, HW, _stream(x)), "mlhot_spatial_mean_fwd")
        return y

    def spatial_mean_bwd(self, dy, shape):
        _chk(dy)
        dx = torch.empty(shape, device=dy.device)
        N, Cc = shape[:2]
        self._rc(self.c.mlhot_spatial_mean_bwd(_ptr(dy), _ptr(dx), N * Cc, dx.numel() // (N * Cc), _stream(dy)), "mlhot_spatial_mean_bwd")
        return dx

    # ---- linear --------------------------------------------------------------------------------
    def linear_fwd(self, x, w, b, act="none"):
        _chk(x, w, b)
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, device=x.device)
        self._rc(self.c.mlhot_linear_fwd(_ptr(x), K, _ptr(w), _ptr(b), _ptr(y), N, M, K, N, ACT[act], _stream(x)), "mlhot_linear_fwd")
        return y

    def linear_bwd(self, x, w, y, dy, act="none", need_dx=True, b=None):
        """`b`: the layer's bias tensor, only to find its gradient's slot in an installed arena."""
        _chk(x, w, y, dy)
        M, K = x.shape
        N = w.shape[0]
        dx = torch.empty_like(x) if need_dx else None
        dw, db = _grad_like(w), (_grad_like(b) if b is not None else torch.empty(N, device=x.device))
        self._rc(self.c.mlhot_linear_bwd(_ptr(x), K, _ptr(w), _ptr(y), N, _ptr(dy), N, M, K, N, ACT[act], _ptr(dx), K, 0,
                                         _ptr(dw), _ptr(db), None, 0, _stream(x)), "mlhot_linear_bwd")
        return dx, dw, db

    # ---- chains of few-row linears / independent few-row linears in one launch (csrc/mlp_chain.h) --------------
    @staticmethod
    def chain_ok(x0, layers):
        """Shapes mlhot_mlp_chain_* take: <= 4 layers, <= 512 rows, input widths <= 512 (multiples of 4), outputs <= 256."""
        M = x0.shape[0]
        if not (1 <= len(layers) <= 4 and M <= 512 and x0.shape[1] % 4 == 0):
            return False
        prev = x0.shape[1]
        for k, (w, b, act, side, side_first) in enumerate(layers):
            sw = side.shape[1] if side is not None else 0
            N, K = w.shape
            if K != prev + sw or K > 512 or K % 4 or sw % 4 or N > 256 or (k + 1 < len(layers) and N % 4):
                return False
            prev = N
        return True

    def _chain_structs(self, x0, layers, ys):
        L = (ChainLayer * len(layers))()
        for k, (w, b, act, side, side_first) in enumerate(layers):
            _chk(w, b, side)
            N, K = w.shape
            L[k] = ChainLayer(_addr(w), _addr(b), K, N, ACT[act], _addr(side), side.shape[1] if side is not None else 0,
                              side.shape[1] if side is not None else 0, int(bool(side_first)), _addr(ys[k]), N)
        return L

    def mlp_chain_fwd(self, x0, layers):
        """x0 [M, K0]; layers: [(w [N, K], b, act, side [M, side_w] | None, side_first)] -> list of the layers' outputs [M, N]."""
        _chk(x0)
        M = x0.shape[0]
        ys = [torch.empty(M, w.shape[0], device=x0.device) for (w, *_rest) in layers]
        L = self._chain_structs(x0, layers, ys)
        self._rc(self.c.mlhot_mlp_chain_fwd(_ptr(x0), x0.shape[1], M, L, len(layers), _stream(x0)), "mlhot_mlp_chain_fwd")
        return ys

    def mlp_chain_bwd(self, x0, layers, ys, dy, need_dx0=True, need_dside=None):
        """-> (dx0 | None, [(dw, db, dside | None)] per layer)."""
        _chk(x0, dy)
        M = x0.shape[0]
        need_dside = need_dside or [False] * len(layers)
        L = self._chain_structs(x0, layers, ys)
        G = (ChainGrads * len(layers))()
        outs, keep = [], []
        for k, (w, b, act, side, side_first) in enumerate(layers):
            dw, db = _grad_like(w), (_grad_like(b) if b is not None else None)
            g = torch.empty(M, w.shape[0] + (-w.shape[0]) % 4, device=x0.device)
            ds = torch.empty_like(side) if (side is not None and need_dside[k]) else None
            G[k] = ChainGrads(_addr(dw), _addr(db), _addr(g), g.shape[1], _addr(ds), ds.shape[1] if ds is not None else 0, 0)
            outs.append((dw, db, ds))
            keep.append(g)
        dx0 = torch.empty_like(x0) if need_dx0 else None
        self._rc(self.c.mlhot_mlp_chain_bwd(_ptr(x0), x0.shape[1], M, L, G, len(layers), _ptr(dy), dy.shape[1], _ptr(dx0), x0.shape[1], 0,
                                            _stream(x0)), "mlhot_mlp_chain_bwd")
        return dx0, outs

    def linear_multi_fwd(self, jobs):
        """jobs: [(x [M, K1], w [N, K], b, act[, x2 [M, K - K1]])] - independent layers, one launch -> [y].  x2: the layer's input is
        cat([x, x2], -1) (folded into the kernel)."""
        J = (LinearJob * len(jobs))()
        ys = []
        for k, (x, w, b, act, *second) in enumerate(jobs):
            x2 = second[0] if second else None
            _chk(x, w, b, x2)
            y = torch.empty(x.shape[0], w.shape[0], device=x.device)
            J[k] = LinearJob(_addr(x), x.shape[1], _addr(w), _addr(b), _addr(y), w.shape[0], x.shape[0], w.shape[1], w.shape[0], ACT[act],
                             None, 0, None, 0, 0, None, None, _addr(x2), x2.shape[1] if x2 is not None else 0,
                             x2.shape[1] if x2 is not None else 0, None, 0)
            ys.append(y)
        self._rc(self.c.mlhot_linear_multi_fwd(J, len(jobs), _stream(jobs[0][0])), "mlhot_linear_multi_fwd")
        return ys

    def linear_multi_bwd(self, jobs):
        """jobs: [(x, w, y, dy, act[, bias[, x2, need_dx, need_dx2]])] -> [(dx, dw, db[, dx2])], all gradient bodies in one launch."""
        J = (LinearJob * len(jobs))()
        outs = []
        for k, (x, w, y, dy, act, *rest) in enumerate(jobs):
            b = rest[0] if rest else None
            x2, need_dx, need_dx2 = (rest[1], rest[2], rest[3]) if len(rest) > 1 else (None, True, False)
            _chk(x, w, y, dy, x2)
            dx = torch.empty_like(x) if need_dx else None
            dx2 = torch.empty_like(x2) if (x2 is not None and need_dx2) else None
            dw, db = _grad_like(w), (_grad_like(b) if b is not None else torch.empty(w.shape[0], device=x.device))
            J[k] = LinearJob(_addr(x), x.shape[1], _addr(w), None, _addr(y), w.shape[0], x.shape[0], w.shape[1], w.shape[0], ACT[act],
                             _addr(dy), dy.shape[1], _addr(dx), x.shape[1], 0, _addr(dw), _addr(db),
                             _addr(x2), x2.shape[1] if x2 is not None else 0, x2.shape[1] if x2 is not None else 0,
                             _addr(dx2), x2.shape[1] if x2 is not None else 0)
            outs.append((dx, dw, db) if x2 is None else (dx, dw, db, dx2))
        self._rc(self.c.mlhot_linear_multi_bwd(J, len(jobs), _stream(jobs[0][0])), "mlhot_linear_multi_bwd")
        return outs

    def axpy(self, a, x, alpha):
        """a + alpha * x (a None: alpha * x), elementwise, one launch."""
        _chk(a, x)
        y = torch.empty_like(x)
        self.c.mlhot_axpy.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]
        self._rc(self.c.mlhot_axpy(_ptr(a), _ptr(x), float(alpha), _ptr(y), x.numel(), _stream(x)), "mlhot_axpy")
        return y

    # ---- aggregators ---------------------------------------------------------------------------
    def agg_fwd(self, mode, rs, lv=None):
        _chk(rs, lv)
        T, Nc, R = rs.shape
        r = torch.empty(T, R, device=rs.device)
        sigma = torch.empty(T, R, device=rs.device)
        amax = torch.empty(T, R, dtype=torch.int32, device=rs.device)
        self._rc(self.c.mlhot_agg_fwd(AGG[mode], _ptr(rs), _ptr(lv), T, Nc, R, _ptr(r), _ptr(sigma), _ptr(amax), _stream(rs)), "mlhot_agg_fwd")
        return r, sigma, amax

    def agg_bwd(self, mode, rs, lv, r, sigma, amax, dr):
        _chk(dr)
        T, Nc, R = rs.shape
        drs = torch.empty_like(rs)
        dlv = torch.empty_like(rs) if mode == "baco" else None
        self._rc(self.c.mlhot_agg_bwd(AGG[mode], _ptr(rs), _ptr(lv), _ptr(r), _ptr(sigma), _ptr(amax), _ptr(dr), T, Nc, R,
                                      _ptr(drs), _ptr(dlv), _stream(rs)), "mlhot_agg_bwd")
        return drs, dlv

    # ---- FAVOR+ --------------------------------------------------------------------------------
    @staticmethod
    def _staged(call, exchange, direction):
        """Run `call(stage, xchg_ptr)` as the two staged halves around the caller's collective (include/mlhot.h, "strict sharded
        parity"): exchange = (object with forward(x) / backward(x), x = 4 floats on the device)."""
        ex, x = exchange
        _chk(x)
        call(0, _ptr(x))
        (ex.forward if direction == "fwd" else ex.backward)(x)
        call(1, _ptr(x))

    def favor_fwd(self, q, k, v, proj, exchange=None):
        """q [T,Nq,H,d], k/v [T,Nc,H,d], proj [m,d] -> out [T,Nq,d*H] (merged order), ws"""
        _chk(q, k, v, proj)
        T, Nq, H, d = q.shape
        Nc, m = k.shape[1], proj.shape[0]
        out = torch.empty(T, Nq, d * H, device=q.device)
        wb = self.c.mlhot_favor_ws_bytes(T, H, Nq, Nc, d, m)
        ws = self._bytes(wb, q)
        if exchange is not None:
            self._staged(lambda st, xp: self._rc(self.c.mlhot_favor_fwd_staged(
                _ptr(q), _ptr(k), _ptr(v), _ptr(proj), T, H, Nq, Nc, d, m, _ptr(out), _ptr(ws), wb, st, xp, _stream(q)),
                "mlhot_favor_fwd_staged"), exchange, "fwd")
            return out, ws
        self._rc(self.c.mlhot_favor_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(proj), T, H, Nq, Nc, d, m, _ptr(out), _ptr(ws), wb, _stream(q)),
                 "mlhot_favor_fwd")
        return out, ws

    def favor_bwd(self, q, k, v, proj, out, dout, ws, exchange=None):
        _chk(dout)
        T, Nq, H, d = q.shape
        Nc, m = k.shape[1], proj.shape[0]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        if exchange is not None:
            self._staged(lambda st, xp: self._rc(self.c.mlhot_favor_bwd_staged(
                _ptr(q), _ptr(k), _ptr(v), _ptr(proj), T, H, Nq, Nc, d, m, _ptr(out), _ptr(dout), _ptr(dq), _ptr(dk), _ptr(dv),
                _ptr(ws), ws.numel(), st, xp, _stream(q)), "mlhot_favor_bwd_staged"), exchange, "bwd")
            return dq, dk, dv
        self._rc(self.c.mlhot_favor_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(proj), T, H, Nq, Nc, d, m, _ptr(out), _ptr(dout),
                                        _ptr(dq), _ptr(dk), _ptr(dv), _ptr(ws), ws.numel(), _stream(q)), "mlhot_favor_bwd")
        return dq, dk, dv

    # ---- losses --------------------------------------------------------------------------------
    def loss_fwd(self, kind, mu, gt):
        _chk(mu, gt)
        rows = mu.numel() // mu.shape[-1]
        loss = torch.empty((), device=mu.device)
        self._rc(self.c.mlhot_loss_fwd(LOSS[kind], _ptr(mu), _ptr(gt), rows, mu.shape[-1], gt.shape[-1], _ptr(loss), _stream(mu)), "mlhot_loss_fwd")
        return loss

    def loss_bwd(self, kind, mu, gt, dloss):
        _chk(dloss)
        rows = mu.numel() // mu.shape[-1]
        dmu = torch.empty_like(mu)
        self._rc(self.c.mlhot_loss_bwd(LOSS[kind], _ptr(mu), _ptr(gt), rows, mu.shape[-1], gt.shape[-1], _ptr(dloss), _ptr(dmu), _stream(mu)),
                 "mlhot_loss_bwd")
        return dmu

    def loss_plus_fwd(self, kind, mu, gt, x, alpha):
        """loss(kind; mu, gt) + alpha * x (x a device scalar: the KL term) in the loss's launch; the total, a device scalar."""
        _chk(mu, gt, x)
        if x.numel() != 1:
            raise MlhotError(f"loss_plus_fwd: x must be a scalar, got {tuple(x.shape)}")
        rows = mu.numel() // mu.shape[-1]
        total = torch.empty((), device=mu.device)
        self._rc(self.c.mlhot_loss_plus_fwd(LOSS[kind], _ptr(mu), _ptr(gt), rows, mu.shape[-1], gt.shape[-1], _ptr(x), float(alpha), None,
                                            _ptr(total), _stream(mu)), "mlhot_loss_plus_fwd")
        return total

    def loss_plus_bwd(self, kind, mu, gt, dtotal, alpha, need_dx=True):
        """(d mu, d x) of loss_plus_fwd for the upstream scalar dtotal; d x = alpha * dtotal (None when not needed)."""
        _chk(dtotal)
        rows = mu.numel() // mu.shape[-1]
        dmu = torch.empty_like(mu)
        dx = torch.empty((), device=mu.device) if need_dx else None
        self._rc(self.c.mlhot_loss_plus_bwd(LOSS[kind], _ptr(mu), _ptr(gt), rows, mu.shape[-1], gt.shape[-1], _ptr(dtotal), float(alpha),
                                            _ptr(dmu), _ptr(dx) if dx is not None else None, _stream(mu)), "mlhot_loss_plus_bwd")
        return dmu, dx

    # ---- batch ingest ---------------------------------------------------------------------------
    def ingest_u8_nhwc(self, src, out=None, div=255.0):
        """src: uint8 [..., H, W, C] on the device (channel-last, as the data loaders hold images) ->
        fp32 [..., C, H, W] = src / div (dataset/shapenet_1d.py:189-190 + utils/utils.py:26-30)."""
        if src.dtype != torch.uint8 or src.dim() < 3:
            raise MlhotError(f"ingest_u8_nhwc expects a uint8 [..., H, W, C] tensor, got {src.dtype} {tuple(src.shape)}")
        _chk(src, out)
        *lead, H, W, Cc = src.shape
        shape = (*lead, Cc, H, W)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=src.device)
        elif out.dtype != torch.float32 or tuple(out.shape) != shape or out.device != src.device:
            raise MlhotError(f"ingest_u8_nhwc: out must be fp32 {shape} on {src.device}")
        n_img = 1
        for v in lead:
            n_img *= v
        self.c.mlhot_ingest_u8_nhwc.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        self._rc(self.c.mlhot_ingest_u8_nhwc(_ptr(src), _ptr(out), n_img, H, W, Cc, float(div), _stream(src)), "mlhot_ingest_u8_nhwc")
        return out

    # ---- fused Adam over flat buffers -------------------------------------------------------------
    def adam_step(self, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
        _chk(param, grad, exp_avg, exp_avg_sq)
        n = param.numel()
        if not (grad.numel() == exp_avg.numel() == exp_avg_sq.numel() == n):
            raise MlhotError("adam_step: buffers differ in size")
        f = C.c_float
        self.c.mlhot_adam_step.argtypes = [C.c_void_p] * 4 + [C.c_size_t] + [f] * 6 + [C.c_int, C.c_void_p]
        self._rc(self.c.mlhot_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), n, lr, beta1, beta2, eps,
                                        weight_decay, grad_scale, int(step), _stream(param)), "mlhot_adam_step")

    def adam_step_counter(self, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, grad_scale, step_counter):
        """adam_step with the step count in device memory (int32 tensor of one element, incremented by the call)."""
        _chk(param, grad, exp_avg, exp_avg_sq, step_counter)
        n = param.numel()
        if not (grad.numel() == exp_avg.numel() == exp_avg_sq.numel() == n) or step_counter.dtype != torch.int32:
            raise MlhotError("adam_step_counter: buffers differ in size, or the counter is not int32")
        f = C.c_float
        self.c.mlhot_adam_step_counter.argtypes = [C.c_void_p] * 4 + [C.c_size_t] + [f] * 6 + [C.c_void_p, C.c_void_p]
        self._rc(self.c.mlhot_adam_step_counter(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), n, lr, beta1, beta2, eps,
                                                weight_decay, grad_scale, _ptr(step_counter), _stream(param)), "mlhot_adam_step_counter")

    def np_grads_layout(self, dims):
        """(total floats, {state_dict key: float offset}) of the library's flat gradient buffer for `dims`."""
        offs = NpGrads()
        total = self.c.mlhot_np_grads_flat_layout(C.byref(dims), C.byref(offs))
        pm = vanilla_param_map(dims.n_hidden, dims.agg_mode == AGG["baco"], dims.agg_mode == AGG["attention"])
        return total, {key: (_get(offs, path) or 0) // 4 for path, key in pm}

    # ---- whole vanilla model -------------------------------------------------------------------
    @staticmethod
    def np_dims(T, Nc, Nq, label_dim, y_dim, dim_w, dim_r, dim_z, hidden, dec_hidden, agg_mode, out_tanh, m_feat):
        d = NpDims()
        d.T, d.Nc, d.Nq, d.label_dim, d.y_dim = T, Nc, Nq, label_dim, y_dim
        d.dim_w, d.dim_r, d.dim_z, d.n_hidden, d.dec_hidden = dim_w, dim_r, dim_z, len(hidden), dec_hidden
        for i, h in enumerate(hidden):
            d.hidden[i] = h
        d.agg_mode, d.out_tanh, d.m_feat = AGG[agg_mode], int(out_tanh), m_feat
        return d

    @staticmethod
    def np_struct(cls, dims, tensors, proj=None):
        """tensors: dict state_dict-key -> tensor (params or grads)"""
        s = cls()
        pm = vanilla_param_map(dims.n_hidden, dims.agg_mode == AGG["baco"], dims.agg_mode == AGG["attention"])
        for path, key in pm:
            _set(s, path, tensors[key].data_ptr())
        if proj is not None:
            s.proj = proj.data_ptr()
        return s

    def np_vanilla_fwd(self, dims, params, ctx_x, ctx_y, qry_x, proj=None, exchange=None):
        _chk(ctx_x, ctx_y, qry_x, proj, *params.values())
        mu = torch.empty(dims.T, dims.Nq, dims.y_dim, device=qry_x.device)
        saved = self._bytes(self.c.mlhot_np_saved_bytes(C.byref(dims)), qry_x)
        sb = self.c.mlhot_np_scratch_bytes(C.byref(dims))
        scratch = self._bytes(sb, qry_x)
        ps = self.np_struct(NpParams, dims, params, proj)
        if exchange is not None:
            self._staged(lambda st, xp: self._rc(self.c.mlhot_np_vanilla_fwd_staged(
                C.byref(dims), C.byref(ps), _ptr(ctx_x), _ptr(ctx_y), _ptr(qry_x), _ptr(mu), _ptr(saved), _ptr(scratch), sb, st, xp,
                _stream(qry_x)), "mlhot_np_vanilla_fwd_staged"), exchange, "fwd")
            return mu, saved, scratch
        self._rc(self.c.mlhot_np_vanilla_fwd(C.byref(dims), C.byref(ps), _ptr(ctx_x), _ptr(ctx_y), _ptr(qry_x), _ptr(mu),
                                             _ptr(saved), _ptr(scratch), sb, _stream(qry_x)), "mlhot_np_vanilla_fwd")
        return mu, saved, scratch

    def np_vanilla_bwd(self, dims, params, ctx_x, ctx_y, qry_x, mu, dmu, saved, scratch=None, proj=None, exchange=None, loss=None):
        """loss: None, or (kind, gt, dloss[, value]) - the loss whose gradient the call takes itself (mlhot_np_vanilla_bwd_loss): the
        gradient of mu is then dmu (may be None) + d loss / d mu * dloss; `value`: None, or the scalar tensor that receives the loss
        VALUE from the same call (what loss_fwd would have written)."""
        _chk(dmu, mu)
        if loss is not None and exchange is not None:
            raise MlhotError("np_vanilla_bwd: the staged pass takes dmu, not a loss descriptor")
        # one flat gradient buffer in the library's preferred order; the per-parameter gradients are views of it
        offs = NpGrads()
        total = self.c.mlhot_np_grads_flat_layout(C.byref(dims), C.byref(offs))
        flat = torch.empty(total, dtype=torch.float32, device=qry_x.device)
        pm = vanilla_param_map(dims.n_hidden, dims.agg_mode == AGG["baco"], dims.agg_mode == AGG["attention"])
        grads = {}
        for path, key in pm:
            off = (_get(offs, path) or 0) // 4
            grads[key] = flat[off:off + params[key].numel()].view_as(params[key])
        sb = self.c.mlhot_np_scratch_bytes(C.byref(dims))
        if scratch is None or scratch.numel() < sb:
            scratch = self._bytes(sb, qry_x)
        ps = self.np_struct(NpParams, dims, params, proj)
        gs = self.np_struct(NpGrads, dims, grads)
        if exchange is not None:
            self._staged(lambda st, xp: self._rc(self.c.mlhot_np_vanilla_bwd_staged(
                C.byref(dims), C.byref(ps), _ptr(ctx_x), _ptr(ctx_y), _ptr(qry_x), _ptr(mu), _ptr(dmu), C.byref(gs), _ptr(saved),
                _ptr(scratch), sb, st, xp, _stream(qry_x)), "mlhot_np_vanilla_bwd_staged"), exchange, "bwd")
            return grads
        if loss is not None:
            kind, gt, dloss, *rest = loss
            value = rest[0] if rest else None
            _chk(gt, dloss)
            if value is not None:
                _chk(value)
            if gt.numel() % (dims.T * dims.Nq) or LOSS[kind] == LOSS["degree"]:
                raise MlhotError(f"np_vanilla_bwd: loss {kind!r} with labels {tuple(gt.shape)} does not fit mu {tuple(mu.shape)}")
            ld = LossDesc(LOSS[kind], _ptr(gt), gt.numel() // (dims.T * dims.Nq), _ptr(dloss), _ptr(value))
            self._rc(self.c.mlhot_np_vanilla_bwd_loss(C.byref(dims), C.byref(ps), _ptr(ctx_x), _ptr(ctx_y), _ptr(qry_x), _ptr(mu), _ptr(dmu),
                                                      C.byref(ld), C.byref(gs), _ptr(saved), _ptr(scratch), sb, _stream(qry_x)), "mlhot_np_vanilla_bwd_loss")
            return grads
        self._rc(self.c.mlhot_np_vanilla_bwd(C.byref(dims), C.byref(ps), _ptr(ctx_x), _ptr(ctx_y), _ptr(qry_x), _ptr(mu), _ptr(dmu),
                                             C.byref(gs), _ptr(saved), _ptr(scratch), sb, _stream(qry_x)), "mlhot_np_vanilla_bwd")
        return grads
