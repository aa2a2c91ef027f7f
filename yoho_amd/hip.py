"""ctypes binding of libyoho_hip.so (include/yoho_hip.h) + a thin tensor-level wrapper.

PyTorch-ROCm tensors are the device-memory container only: every method passes
``tensor.data_ptr()`` and the current HIP stream to the C ABI.  There is NO CPU fallback: if the
library cannot be loaded, importing a GPU op raises.
"""
import ctypes as C
import os
import warnings
import numpy as np
import torch  # must be imported before the library so that both share one HIP runtime (libamdhip64.so.7)

from . import tables as _tables
from . import weights as _weights

# YOHO_LIB=exp: the experiments build (YOHO_EXPERIMENTS=1 python -m yoho_amd.build: the timing switches compiled in) - measurement tools only
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                         "libyoho_hip_exp.so" if os.environ.get("YOHO_LIB") == "exp" else "libyoho_hip.so")
_lib = None

SYMBOLS = [
    "yoho_last_error", "yoho_version", "yoho_ctx_create", "yoho_ctx_destroy", "yoho_load_partI",
    "yoho_load_partII", "yoho_partI_forward", "yoho_partI_forward_pair", "yoho_group_mean_np", "yoho_nn_search", "yoho_mutual_nn",
    "yoho_gconv_layer", "yoho_load_fcgf", "yoho_fcgf_voxelize", "yoho_fcgf_forward", "yoho_fcgf_forward_batch", "yoho_fcgf_voxelize_rotated", "yoho_rotate_select",
    "yoho_des2r", "yoho_des2r_indexed", "yoho_partII_forward", "yoho_partII_forward_indexed", "yoho_hyp_from_quat", "yoho_o_score", "yoho_c_ransac",
    "yoho_group_gather", "yoho_set_profiling", "yoho_get_kernel_ms", "yoho_set_gconv_mode", "yoho_set_partII_mode", "yoho_set_nn_grid",
    "yoho_range_status", "yoho_c_ransac_device", "yoho_group_scatter", "yoho_set_nn_prefilter", "yoho_set_fcgf_sort", "yoho_fcgf_voxelize_rotated_batch", "yoho_gconv_wgrad", "yoho_bn_stats", "yoho_bn_relu_apply", "yoho_bn_relu_backward", "yoho_set_partI_schedule", "yoho_clock_probe", "yoho_group_transfer_batch",
    "yoho_register_pair", "yoho_vote_order", "yoho_c_draw_np", "yoho_phase_profile", "yoho_phase_read",
]


class ConvW(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p)]


class BnW(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("var", C.c_void_p)]


class PartIWeights(C.Structure):
    _fields_ = [("conv_in", ConvW), ("res_in_bn", BnW), ("res_in", ConvW), ("res_out_bn", BnW),
                ("res_out", ConvW), ("out_bn", BnW), ("conv_out", ConvW)]


class PartIIWeights(C.Structure):
    _fields_ = [("init_bn", BnW), ("init", ConvW), ("res_in_bn", BnW), ("res_in", ConvW),
                ("res_out_bn", BnW), ("res_out", ConvW), ("fc0", ConvW), ("fc0_bn", BnW),
                ("fc1", ConvW), ("fc1_bn", BnW), ("fc2", ConvW)]


def lib_path():
    return _LIB_PATH


class FcgfConfig(C.Structure):
    _fields_ = [("channels", C.c_int * 5), ("tr_channels", C.c_int * 5), ("out_channels", C.c_int),
                ("conv1_kernel_size", C.c_int), ("in_channels", C.c_int), ("normalize_feature", C.c_int)]


class PairResultC(C.Structure):
    _fields_ = [("trans", C.c_double * 12), ("matches", C.c_int32), ("best_h", C.c_int32), ("best_count", C.c_int32),
                ("hypotheses", C.c_int32), ("range_flag", C.c_int32), ("reserved", C.c_int32)]


def load_library():
    """Load libyoho_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} not found: build it with `python -m yoho_amd.build` "
            "(there is no CPU fallback for the YOHO hot path)")
    lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise RuntimeError(f"libyoho_hip.so does not export {s}")
    lib.yoho_last_error.restype = C.c_char_p
    lib.yoho_version.restype = C.c_char_p
    vp, ci = C.c_void_p, C.c_int
    lib.yoho_ctx_create.argtypes = [ci, vp, vp, vp, C.POINTER(vp)]
    lib.yoho_ctx_destroy.argtypes = [vp]
    lib.yoho_load_partI.argtypes = [vp, C.POINTER(PartIWeights)]
    lib.yoho_load_partII.argtypes = [vp, C.POINTER(PartIIWeights)]
    lib.yoho_partI_forward.argtypes = [vp, vp, ci, vp, vp, vp, vp]
    lib.yoho_partI_forward_pair.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp, vp]
    lib.yoho_group_mean_np.argtypes = [vp, vp, ci, vp, vp]
    lib.yoho_nn_search.argtypes = [vp, vp, ci, vp, ci, ci, ci, vp, vp, vp]
    lib.yoho_mutual_nn.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp]
    lib.yoho_des2r.argtypes = [vp, vp, vp, ci, vp, vp, vp]
    lib.yoho_des2r_indexed.argtypes = [vp, vp, vp, vp, vp, ci, ci, vp, vp, vp]
    lib.yoho_partII_forward_indexed.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp]
    lib.yoho_partII_forward.argtypes = [vp, vp, vp, vp, vp, vp, ci, vp, vp]
    lib.yoho_hyp_from_quat.argtypes = [vp, vp, vp, vp, vp, ci, vp, vp]
    lib.yoho_o_score.argtypes = [vp, vp, vp, ci, vp, vp, ci, C.c_double, vp, vp, vp, vp]
    lib.yoho_c_ransac.argtypes = [vp, vp, vp, ci, vp, vp, ci, C.c_double, vp, vp, vp, vp, vp, vp]
    lib.yoho_group_gather.argtypes = [vp, vp, ci, vp, vp, ci, ci, vp, vp, vp, vp]
    lib.yoho_gconv_layer.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, vp, vp]
    lib.yoho_load_fcgf.argtypes = [vp, C.POINTER(FcgfConfig), C.POINTER(vp), ci]
    lib.yoho_fcgf_voxelize.argtypes = [vp, vp, ci, C.c_double, vp, vp, C.POINTER(ci), vp]
    lib.yoho_fcgf_forward.argtypes = [vp, vp, ci, vp, vp]
    lib.yoho_fcgf_forward_batch.argtypes = [vp, vp, vp, ci, vp, vp]
    lib.yoho_set_profiling.argtypes = [vp, ci]
    lib.yoho_set_gconv_mode.argtypes = [vp, ci]
    lib.yoho_set_partII_mode.argtypes = [vp, ci]
    lib.yoho_set_partI_schedule.argtypes = [vp, ci, ci]
    lib.yoho_clock_probe.argtypes = [vp, ci, vp, vp]
    lib.yoho_group_transfer_batch.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, vp, ci, vp, vp, vp, vp]
    lib.yoho_set_nn_grid.argtypes = [vp, C.c_double]
    lib.yoho_set_nn_prefilter.argtypes = [vp, ci]
    lib.yoho_set_fcgf_sort.argtypes = [vp, ci, ci]
    lib.yoho_fcgf_voxelize_rotated.argtypes = [vp, vp, ci, vp, C.c_double, vp, vp, vp, vp, vp]
    lib.yoho_rotate_select.argtypes = [vp, vp, vp, vp, ci, vp, vp]
    lib.yoho_fcgf_voxelize_rotated_batch.argtypes = [vp, vp, ci, vp, ci, C.c_double, vp, vp, vp, vp, vp]
    lib.yoho_get_kernel_ms.argtypes = [vp, ci, C.POINTER(C.c_float)]
    lib.yoho_bn_stats.argtypes = [vp, vp, ci, ci, vp, vp, vp]
    lib.yoho_bn_relu_apply.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    lib.yoho_bn_relu_backward.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp]
    lib.yoho_gconv_wgrad.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp]
    lib.yoho_group_scatter.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp]
    lib.yoho_range_status.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), vp]
    lib.yoho_c_ransac_device.argtypes = [vp, vp, vp, vp, vp, ci, vp, ci, ci, C.c_uint64, C.c_double, vp, vp, vp, vp, vp]
    lib.yoho_register_pair.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, C.c_double, C.c_uint64, ci, C.POINTER(PairResultC), vp]
    lib.yoho_vote_order.argtypes = [C.c_uint32, ci, vp]
    lib.yoho_c_draw_np.argtypes = [vp, C.POINTER(ci), vp, vp, vp, ci, vp, C.POINTER(ci), C.POINTER(ci)]
    lib.yoho_phase_profile.argtypes = [vp, ci]
    lib.yoho_phase_read.argtypes = [vp, vp, vp, vp, vp]
    for s in SYMBOLS[2:]:
        getattr(lib, s).restype = ci
    _lib = lib
    return lib


def vote_order(seed, M):
    """np.random.RandomState(seed).shuffle(np.arange(M)) as the library computes it on the host (yoho_vote_order)"""
    lib = load_library()
    out = np.empty((M,), dtype=np.int64)
    rc = lib.yoho_vote_order(int(seed) & 0xFFFFFFFF, int(M), out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise YohoError(f"libyoho_hip error {rc}: {lib.yoho_last_error().decode()}", rc)
    return out


def c_draw_np(prob, bucket_start, bucket_members, max_iter, rng=None):
    """The draws of the reference's YOHO-C loop (tests/estimator.py:113-128) taken from numpy's legacy stream in C (yoho_c_draw_np):
    rng is a np.random.RandomState or None for the global np.random state, which is advanced exactly as the reference's
    `np.random.choice(range(60), p=prob)` / `np.random.choice(bucket, 3)` calls advance it.  -> (triples (I,3) int64, draws)"""
    lib = load_library()
    src = np.random if rng is None else rng
    st = src.get_state()
    if st[0] != "MT19937":
        raise YohoError(f"c_draw_np continues numpy's legacy MT19937 stream; the generator is {st[0]!r}")
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = C.c_int(int(st[2]))
    prob = np.ascontiguousarray(prob, dtype=np.float64)
    start = np.ascontiguousarray(bucket_start, dtype=np.int64)
    members = np.ascontiguousarray(bucket_members, dtype=np.int64)
    if prob.shape != (60,) or start.shape != (61,) or members.ndim != 1 or int(start[-1]) > members.shape[0]:
        raise ValueError("c_draw_np: prob (60,), bucket_start (61,), bucket_members (>= bucket_start[60],)")
    tri = np.empty((max(int(max_iter), 0), 3), dtype=np.int64)
    n, draws = C.c_int(0), C.c_int(0)
    rc = lib.yoho_c_draw_np(key.ctypes.data_as(C.c_void_p), C.byref(pos), prob.ctypes.data_as(C.c_void_p), start.ctypes.data_as(C.c_void_p),
                            members.ctypes.data_as(C.c_void_p), int(max_iter), tri.ctypes.data_as(C.c_void_p), C.byref(n), C.byref(draws))
    if rc != 0:
        raise YohoError(f"libyoho_hip error {rc}: {lib.yoho_last_error().decode()}", rc)
    src.set_state((st[0], key, pos.value) + tuple(st[3:]))
    return tri[:n.value], draws.value


class YohoError(RuntimeError):
    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


GCONV_MODES = {"f32": 0, "bf16x3": 1, "fourier": 2, "fp16x2": 3, "fgemm": 4, "fgemm256": 5, "fgemm128": 6, "fgemm8": 7}
PARTII_MODES = {"f32": 0, "bf16x3": 1, "fp16x2": 2, "cgemm": 3, "cgemm8": 4}
FP16_PARTII_MODES = ("fp16x2", "cgemm", "cgemm8")       # fp16 planes: Fourier first layer, row-indexed entries, the fp16 range guard
FP16_GCONV_MODES = ("fp16x2", "fgemm", "fgemm256", "fgemm128", "fgemm8")      # arithmetic with fp16 planes: guarded by the range flag (yoho_range_status)
MAX_PAIR_KEYPOINTS = 16384                  # yoho_partI_forward_pair takes both fragments in one pass up to this many rows


def _check(rc):
    if rc != 0:
        raise YohoError(f"libyoho_hip error {rc}: {_lib.yoho_last_error().decode()}", rc)


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError(f"{name}: expected a CUDA/HIP tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


class Context:
    """One yoho_ctx per device: group tables + (optionally) PartI / PartII weights."""

    def __init__(self, device=None, so3_dir=None):
        lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("yoho_amd.hip.Context needs a visible MI355X (torch.cuda.is_available() is False)")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.tables = _tables.GroupTables(so3_dir)
        h = C.c_void_p()
        R = np.ascontiguousarray(self.tables.R32.reshape(60, 9))
        _check(lib.yoho_ctx_create(self.device, _np_ptr(R), _np_ptr(self.tables.N_u8), _np_ptr(self.tables.P_u8), C.byref(h)))
        self._h = h
        self._lib = lib
        # arithmetic modes are tracked here as well (the library reads the same variables at yoho_ctx_create)
        gm, pm = os.environ.get("YOHO_GCONV", "fgemm"), os.environ.get("YOHO_PARTII", "fp16x2")
        self.gconv_mode = gm if gm in GCONV_MODES else "fourier"
        self.partII_mode = pm if pm in PARTII_MODES else "bf16x3"
        self.set_gconv_mode(self.gconv_mode)
        self.set_partII_mode(self.partII_mode)
        self.range_fallbacks = 0            # passes repeated in bf16x3 because a value left the fp16 range (both networks, ever)
        # per network, since its checkpoint was loaded: repeats, and whether the network has been switched to bf16x3 for good
        self.range_repeats = {"gconv": 0, "partII": 0}
        self.range_sticky = {"gconv": None, "partII": None}       # None, or the mode the network ran in before it was switched
        self.range_sticky_after = int(os.environ.get("YOHO_RANGE_STICKY", "3"))    # 0: never switch, repeat every time
        self._range_pending = [False, False]  # flags read from the device but not yet consumed by a caller (range_status)
        self.partI_owner = self.partII_owner = self.fcgf_owner = None      # whose weights are resident (network objects)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.yoho_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_lib", None) is not None:
            self.close()

    # ---- weights ---------------------------------------------------------------------------
    @staticmethod
    def _conv(sd, k, keep):
        w = np.ascontiguousarray(sd[k + ".weight"], dtype=np.float32)
        b = np.ascontiguousarray(sd[k + ".bias"], dtype=np.float32)
        keep += [w, b]
        return ConvW(w.ctypes.data, b.ctypes.data)

    @staticmethod
    def _bn(sd, k, keep):
        arrs = [np.ascontiguousarray(sd[k + s], dtype=np.float32) for s in (".weight", ".bias", ".running_mean", ".running_var")]
        keep += arrs
        return BnW(*[a.ctypes.data for a in arrs])

    def load_partI(self, sd, owner=None):
        """owner: the network object these weights belong to; objects sharing the context compare ``ctx.partI_owner``
        with themselves before a forward pass and upload again when another checkpoint has been loaded since."""
        self.partI_owner = owner
        self._new_checkpoint("gconv")
        sd = _weights.to_numpy_state_dict(sd)
        _weights.check_state_dict(sd, _weights.PARTI_SPEC, strict=False)
        keep = []
        p, r = "PartI_net.", "PartI_net.SO3_Conv_layers.0."
        w = PartIWeights(self._conv(sd, p + "Conv_in.0", keep),
                         self._bn(sd, r + "comb_layer_in.0", keep), self._conv(sd, r + "comb_layer_in.2", keep),
                         self._bn(sd, r + "comb_layer_out.0", keep), self._conv(sd, r + "comb_layer_out.2", keep),
                         self._bn(sd, p + "Conv_out.comb_layer.0", keep), self._conv(sd, p + "Conv_out.comb_layer.2", keep))
        _check(self._lib.yoho_load_partI(self._h, C.byref(w)))

    def load_partII(self, sd, owner=None):
        self.partII_owner = owner
        self._new_checkpoint("partII")
        sd = _weights.to_numpy_state_dict(sd)
        _weights.check_state_dict(sd, _weights.PARTII_SPEC, strict=False)
        keep = []
        r, f = "PartII_SO3_Conv_layers.0.", "PartII_To_R_FC."
        w = PartIIWeights(self._bn(sd, "Conv_init.comb_layer.0", keep), self._conv(sd, "Conv_init.comb_layer.2", keep),
                          self._bn(sd, r + "comb_layer_in.0", keep), self._conv(sd, r + "comb_layer_in.2", keep),
                          self._bn(sd, r + "comb_layer_out.0", keep), self._conv(sd, r + "comb_layer_out.2", keep),
                          self._conv(sd, f + "0", keep), self._bn(sd, f + "1", keep),
                          self._conv(sd, f + "3", keep), self._bn(sd, f + "4", keep), self._conv(sd, f + "6", keep))
        _check(self._lib.yoho_load_partII(self._h, C.byref(w)))

    # ---- training path ------------------------------------------------------------------------
    def gconv_layer(self, x, weight, bias=None, transpose=False):
        """One (1,13) group-conv layer on device tensors.  weight (cout,cin,1,13), bias (cout) or None.
        transpose=False: x (B,cin,60) -> (B,cout,60); transpose=True: x = output gradient (B,cout,60) -> input gradient (B,cin,60)."""
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        B = x.shape[0]
        xc, yc = (cout, cin) if transpose else (cin, cout)
        if x.dim() != 3 or x.shape[1] != xc or x.shape[2] != 60:
            raise ValueError(f"expected ({'B'},{xc},60), got {tuple(x.shape)}")
        y = torch.empty((B, yc, 60), dtype=torch.float32, device=x.device)
        _check(self._lib.yoho_gconv_layer(self._h, _dev(x, torch.float32, "x"), B, cin, cout, _dev(weight, torch.float32, "weight"),
                                          _dev(bias, torch.float32, "bias") if (bias is not None and not transpose) else None,
                                          1 if transpose else 0, C.c_void_p(y.data_ptr()), _stream()))
        return y

    def bn_stats(self, x):
        """per-channel mean and biased variance of a (B,C,60) tensor"""
        B, Cn = x.shape[0], x.shape[1]
        mean = torch.empty((Cn,), dtype=torch.float32, device=x.device)
        var = torch.empty((Cn,), dtype=torch.float32, device=x.device)
        _check(self._lib.yoho_bn_stats(self._h, _dev(x, torch.float32, "x"), B, Cn, C.c_void_p(mean.data_ptr()), C.c_void_p(var.data_ptr()), _stream()))
        return mean, var

    def bn_relu_apply(self, x, scale, shift):
        y = torch.empty_like(x)
        _check(self._lib.yoho_bn_relu_apply(self._h, _dev(x, torch.float32, "x"), x.shape[0], x.shape[1], _dev(scale, torch.float32, "scale"),
                                            _dev(shift, torch.float32, "shift"), C.c_void_p(y.data_ptr()), _stream()))
        return y

    def bn_relu_backward(self, x, y, dy, gamma, mean, rstd, batch_stats):
        dx = torch.empty_like(x)
        dg = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
        db = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
        _check(self._lib.yoho_bn_relu_backward(self._h, _dev(x, torch.float32, "x"), _dev(y, torch.float32, "y"), _dev(dy, torch.float32, "dy"),
                                               x.shape[0], x.shape[1], _dev(gamma, torch.float32, "gamma"), _dev(mean, torch.float32, "mean"),
                                               _dev(rstd, torch.float32, "rstd"), 1 if batch_stats else 0, C.c_void_p(dx.data_ptr()),
                                               C.c_void_p(dg.data_ptr()), C.c_void_p(db.data_ptr()), _stream()))
        return dx, dg, db

    def gconv_wgrad(self, x, dy, want_bias=True):
        """weight / bias gradient of the (1,13) group conv: x (B,cin,60), dy (B,cout,60) -> (dW (cout,cin,1,13), db (cout) or None)"""
        B, cin, cout = x.shape[0], int(x.shape[1]), int(dy.shape[1])
        dW = torch.empty((cout, cin, 1, 13), dtype=torch.float32, device=x.device)
        db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_bias else None
        _check(self._lib.yoho_gconv_wgrad(self._h, _dev(x, torch.float32, "x"), _dev(dy, torch.float32, "dy"), B, cin, cout,
                                          C.c_void_p(dW.data_ptr()), C.c_void_p(db.data_ptr()) if want_bias else None, _stream()))
        return dW, db

    # ---- FCGF backbone ------------------------------------------------------------------------
    def load_fcgf(self, sd, channels=(0, 32, 64, 128, 256), tr_channels=(0, 64, 64, 64, 128), out_channels=32, conv1_kernel_size=7,
                  in_channels=1, normalize_feature=True, owner=None):
        """sd: FCGF backbone state_dict (torch tensors or ndarrays); architecture parameters as in fcgf_model/resunet.py."""
        self.fcgf_owner = owner
        sd = _weights.to_numpy_state_dict(sd)
        spec = _weights.fcgf_spec(tuple(channels), tuple(tr_channels), out_channels, conv1_kernel_size, in_channels)
        _weights.check_state_dict(sd, spec, strict=False)
        names = [n for n, _ in spec if not n.endswith("num_batches_tracked")]
        arrs = [np.ascontiguousarray(sd[n], dtype=np.float32) for n in names]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        cfg = FcgfConfig((C.c_int * 5)(*[int(v or 0) for v in channels]), (C.c_int * 5)(*[int(v or 0) for v in tr_channels]),
                         int(out_channels), int(conv1_kernel_size), int(in_channels), int(bool(normalize_feature)))
        _check(self._lib.yoho_load_fcgf(self._h, C.byref(cfg), ptrs, len(arrs)))
        self._fcgf_out = int(out_channels)

    def fcgf_voxelize(self, pts, voxel_size):
        """pts (n,3) f64 cuda -> (sel (m,) int64 ascending, coords (m,3) int32): first point of every voxel (fcgf_feat.py:33-43)."""
        n = pts.shape[0]
        sel = torch.empty((n,), dtype=torch.int64, device=pts.device)
        coords = torch.empty((n, 3), dtype=torch.int32, device=pts.device)
        cnt = C.c_int(0)
        _check(self._lib.yoho_fcgf_voxelize(self._h, _dev(pts, torch.float64, "pts"), n, float(voxel_size), C.c_void_p(sel.data_ptr()),
                                            C.c_void_p(coords.data_ptr()), C.byref(cnt), _stream()))
        return sel[:cnt.value], coords[:cnt.value]

    def fcgf_voxelize_rotated(self, pts, R, voxel_size, want_points=True):
        """voxelise the copy of pts (n,3) f64 cuda rotated by R (3,3) (p' = R p) without materialising it ->
        (sel, coords[, rotated selected points (m,3) f32])."""
        n = pts.shape[0]
        Rh = np.ascontiguousarray(np.asarray(R, dtype=np.float64).reshape(3, 3))
        sel = torch.empty((n,), dtype=torch.int64, device=pts.device)
        coords = torch.empty((n, 3), dtype=torch.int32, device=pts.device)
        ps = torch.empty((n, 3), dtype=torch.float32, device=pts.device) if want_points else None
        cnt = C.c_int(0)
        _check(self._lib.yoho_fcgf_voxelize_rotated(self._h, _dev(pts, torch.float64, "pts"), n, _np_ptr(Rh), float(voxel_size),
                                                    C.c_void_p(sel.data_ptr()), C.c_void_p(coords.data_ptr()),
                                                    C.c_void_p(ps.data_ptr()) if want_points else None, C.byref(cnt), _stream()))
        m = cnt.value
        return (sel[:m], coords[:m], ps[:m]) if want_points else (sel[:m], coords[:m])

    def fcgf_voxelize_rotated_batch(self, pts, Rs, voxel_size, want_points=True):
        """fcgf_voxelize_rotated for a list of rotations of the same cloud in one library call (one count read-back for all):
        -> list of (sel, coords[, rotated selected points]) views into shared (nb, n, ...) buffers."""
        n, nb = pts.shape[0], len(Rs)
        Rh = np.ascontiguousarray(np.stack([np.asarray(R, dtype=np.float64).reshape(3, 3) for R in Rs]))
        sel = torch.empty((nb, n), dtype=torch.int64, device=pts.device)
        coords = torch.empty((nb, n, 3), dtype=torch.int32, device=pts.device)
        ps = torch.empty((nb, n, 3), dtype=torch.float32, device=pts.device) if want_points else None
        cnt = (C.c_int * nb)()
        _check(self._lib.yoho_fcgf_voxelize_rotated_batch(self._h, _dev(pts, torch.float64, "pts"), n, _np_ptr(Rh), nb, float(voxel_size),
                                                          C.c_void_p(sel.data_ptr()), C.c_void_p(coords.data_ptr()),
                                                          C.c_void_p(ps.data_ptr()) if want_points else None, cnt, _stream()))
        out = []
        for b in range(nb):
            m = cnt[b]
            out.append((sel[b, :m], coords[b, :m], ps[b, :m]) if want_points else (sel[b, :m], coords[b, :m]))
        return out

    def rotate_select(self, pts, R, sel):
        """(float32)(R pts[sel]) for pts (n,3) f64 cuda, sel (m,) int64 cuda; R (3,3) or None."""
        m = sel.shape[0]
        out = torch.empty((m, 3), dtype=torch.float32, device=pts.device)
        Rh = None if R is None else np.ascontiguousarray(np.asarray(R, dtype=np.float64).reshape(3, 3))
        _check(self._lib.yoho_rotate_select(self._h, _dev(pts, torch.float64, "pts"), _np_ptr(Rh) if Rh is not None else None,
                                            _dev(sel, torch.int64, "sel"), m, C.c_void_p(out.data_ptr()), _stream()))
        return out

    def fcgf_forward(self, coords):
        """coords (n,3) int32 cuda, distinct voxels -> (n, out_channels) f32 unit rows."""
        n = coords.shape[0]
        out = torch.empty((n, getattr(self, "_fcgf_out", 32)), dtype=torch.float32, device=coords.device)
        _check(self._lib.yoho_fcgf_forward(self._h, _dev(coords, torch.int32, "coords"), n, C.c_void_p(out.data_ptr()), _stream()))
        return out

    def fcgf_forward_batch(self, coords_list):
        """several clouds ((n_b,3) int32 cuda each) in one pass -> list of (n_b, out_channels) feature tensors."""
        nb = len(coords_list)
        off = np.zeros(nb + 1, dtype=np.int32)
        off[1:] = np.cumsum([c.shape[0] for c in coords_list])
        allc = torch.cat(coords_list).contiguous()
        out = torch.empty((int(off[-1]), getattr(self, "_fcgf_out", 32)), dtype=torch.float32, device=allc.device)
        _check(self._lib.yoho_fcgf_forward_batch(self._h, _dev(allc, torch.int32, "coords"), off.ctypes.data_as(C.c_void_p), nb,
                                                 C.c_void_p(out.data_ptr()), _stream()))
        return [out[int(off[b]):int(off[b + 1])] for b in range(nb)]

    # ---- descriptor path -------------------------------------------------------------------
    # ---- fp16 range guard -------------------------------------------------------------------
    def range_status(self, consume=(True, True)):
        """(partI_overflow, partII_overflow): waits for the current stream and reads the device flags the fp16x2 kernels raise when
        a value does not fit an fp16 plane (include/yoho_hip.h).  A raised flag stays pending until a call CONSUMES that
        component: a caller that looks at one network only passes consume=(True, False) / (False, True), so a PartI overflow left
        behind by an unchecked pass is not wiped by a PartII check (and vice versa)."""
        a, b = C.c_int(0), C.c_int(0)
        rc = self._lib.yoho_range_status(self._h, C.byref(a), C.byref(b), _stream())
        if rc not in (0, -5):                      # YOHO_ERANGE is the report itself
            _check(rc)
        self._range_pending[0] |= bool(a.value)
        self._range_pending[1] |= bool(b.value)
        out = tuple(self._range_pending)
        for i in (0, 1):
            if consume[i]:
                self._range_pending[i] = False
        return out

    def partI_overflow(self):
        return self.range_status(consume=(True, False))[0]

    def partII_overflow(self):
        return self.range_status(consume=(False, True))[1]

    def _new_checkpoint(self, which):
        """another checkpoint for network `which`: its repeat count starts again, and a network that had been switched to bf16x3
        for the previous checkpoint goes back to the mode it was configured with"""
        self.range_repeats[which] = 0
        prev, self.range_sticky[which] = self.range_sticky[which], None
        if prev is not None:
            (self.set_gconv_mode if which == "gconv" else self.set_partII_mode)(prev)

    def _repeat_wider(self, which, fn):
        """repeat fn() with `which` ('gconv' / 'partII') switched to the bf16x3 planes (fp32 exponent range).

        A checkpoint whose activations do not fit the fp16 planes would otherwise pay fp16x2 pass + flag + bf16x3 pass on EVERY
        call, behind nothing but a RuntimeWarning per pass: after `range_sticky_after` repeats since the checkpoint was loaded the
        network STAYS in bf16x3 (about 3x the fp16x2 time, but once instead of 1 + 3), which is said once, loudly, and is visible
        in range_report(); loading another checkpoint restores the configured mode."""
        self.range_fallbacks += 1
        self.range_repeats[which] += 1
        attr, setter = ("gconv_mode", self.set_gconv_mode) if which == "gconv" else ("partII_mode", self.set_partII_mode)
        old = getattr(self, attr)
        stick = self.range_sticky_after > 0 and self.range_repeats[which] >= self.range_sticky_after and old != "bf16x3"
        if stick:
            warnings.warn(f"yoho_amd: {self.range_repeats[which]} passes of this checkpoint left the fp16 range of the {which} fp16x2 arithmetic; "
                          f"the network now STAYS in bf16x3 (fp32 exponent range, about 3x slower than {old}) until another checkpoint is "
                          f"loaded (YOHO_RANGE_STICKY=0 keeps repeating pass by pass instead)", RuntimeWarning)
        else:
            warnings.warn(f"yoho_amd: a value left the fp16 range of the {which} fp16x2 arithmetic; pass repeated in bf16x3", RuntimeWarning)
        setter("bf16x3")
        try:
            return fn()
        finally:
            if stick:
                self.range_sticky[which] = old
            else:
                setter(old)

    def range_report(self):
        """what the range guard has done on this context: repeats per network since its checkpoint was loaded, networks switched
        to bf16x3 for good, total repeats ever - for the stats of the dataset driver and the bench line"""
        return {"partI_repeats": self.range_repeats["gconv"], "partII_repeats": self.range_repeats["partII"],
                "partI_stays_bf16x3": self.range_sticky["gconv"] is not None, "partII_stays_bf16x3": self.range_sticky["partII"] is not None,
                "repeats_total": self.range_fallbacks}

    def supports_pair(self, n_rows):
        """yoho_partI_forward_pair (no concatenation copy) exists for the default arithmetic mode and one pass"""
        return self.gconv_mode in ("fgemm", "fgemm256", "fgemm128", "fgemm8") and n_rows <= MAX_PAIR_KEYPOINTS

    def supports_matched(self):
        """row-indexed PartII (yoho_partII_forward_indexed) exists for the default PartII mode"""
        return self.partII_mode in FP16_PARTII_MODES

    def partI_forward(self, x, want_inv=True, want_inv_np=False, check_range=True):
        """x (B,32,60) f32 cuda -> dict(eqv, inv[, inv_np]).  check_range: in the fp16x2 modes verify the range flag
        (one stream synchronisation) and repeat the pass in bf16x3 if it is raised; callers that pass False poll
        range_status() themselves (pipeline.run_pair does, at its first host read-back)."""
        if check_range and self.gconv_mode in FP16_GCONV_MODES:
            out = self.partI_forward(x, want_inv, want_inv_np, check_range=False)
            if self.partI_overflow():
                out = self._repeat_wider("gconv", lambda: self.partI_forward(x, want_inv, want_inv_np, check_range=False))
            return out
        B = x.shape[0]
        if x.dim() != 3 or x.shape[1] != 32 or x.shape[2] != 60:
            raise ValueError(f"group feature must be (B,32,60), got {tuple(x.shape)}")
        eqv = torch.empty_like(x)
        inv = torch.empty((B, 32), dtype=torch.float32, device=x.device) if want_inv else None
        inv_np = torch.empty((B, 32), dtype=torch.float32, device=x.device) if want_inv_np else None
        _check(self._lib.yoho_partI_forward(self._h, _dev(x, torch.float32, "x"), B, C.c_void_p(eqv.data_ptr()),
                                            C.c_void_p(inv.data_ptr()) if want_inv else None,
                                            C.c_void_p(inv_np.data_ptr()) if want_inv_np else None, _stream()))
        out = {"eqv": eqv}
        if want_inv:
            out["inv"] = inv
        if want_inv_np:
            out["inv_np"] = inv_np
        return out

    def partI_forward_pair(self, x0, x1, want_inv=True, want_inv_np=False, check_range=True):
        """both fragments in one pass, no concatenation copy: outputs have B0 + B1 rows (x0's first).  Default mode only
        (supports_pair)."""
        if check_range:
            out = self.partI_forward_pair(x0, x1, want_inv, want_inv_np, check_range=False)
            if self.partI_overflow():
                xc = torch.cat([x0, x1])
                out = self._repeat_wider("gconv", lambda: self.partI_forward(xc, want_inv, want_inv_np, check_range=False))
            return out
        B0, B1 = x0.shape[0], x1.shape[0]
        eqv = torch.empty((B0 + B1, 32, 60), dtype=torch.float32, device=x0.device)
        inv = torch.empty((B0 + B1, 32), dtype=torch.float32, device=x0.device) if want_inv else None
        inv_np = torch.empty((B0 + B1, 32), dtype=torch.float32, device=x0.device) if want_inv_np else None
        _check(self._lib.yoho_partI_forward_pair(self._h, _dev(x0, torch.float32, "x0"), B0, _dev(x1, torch.float32, "x1"), B1,
                                                 C.c_void_p(eqv.data_ptr()), C.c_void_p(inv.data_ptr()) if want_inv else None,
                                                 C.c_void_p(inv_np.data_ptr()) if want_inv_np else None, _stream()))
        out = {"eqv": eqv}
        if want_inv:
            out["inv"] = inv
        if want_inv_np:
            out["inv_np"] = inv_np
        return out

    def group_mean_np(self, eqv):
        B = eqv.shape[0]
        out = torch.empty((B, 32), dtype=torch.float32, device=eqv.device)
        _check(self._lib.yoho_group_mean_np(self._h, _dev(eqv, torch.float32, "eqv"), B, C.c_void_p(out.data_ptr()), _stream()))
        return out

    def nn_search(self, src, tgt, want_dist=True, squared=False):
        """src (Ns,D), tgt (Nt,D) f32, D in {32,3} -> (dist (Ns) f32 or None, idx (Ns) int64).
        squared=False: pdist 'L2' (sqrt(D2 + 1e-7)); True: 'SquareL2' (D2)."""
        Ns, D = src.shape
        Nt = tgt.shape[0]
        idx = torch.empty((Ns,), dtype=torch.int64, device=src.device)
        dist = torch.empty((Ns,), dtype=torch.float32, device=src.device) if want_dist else None
        _check(self._lib.yoho_nn_search(self._h, _dev(src, torch.float32, "src"), Ns, _dev(tgt, torch.float32, "tgt"), Nt, D, 1 if squared else 0,
                                        C.c_void_p(idx.data_ptr()), C.c_void_p(dist.data_ptr()) if want_dist else None, _stream()))
        return dist, idx

    def mutual_nn(self, a, b):
        """a (Na,32), b (Nb,32) -> (M,2) int64 mutual nearest neighbours, ascending in a."""
        Na, Nb = a.shape[0], b.shape[0]
        pairs = torch.empty((Na, 2), dtype=torch.int64, device=a.device)
        m = torch.zeros((1,), dtype=torch.int32, device=a.device)
        _check(self._lib.yoho_mutual_nn(self._h, _dev(a, torch.float32, "a"), Na, _dev(b, torch.float32, "b"), Nb,
                                        C.c_void_p(pairs.data_ptr()), C.c_void_p(m.data_ptr()), _stream()))
        return pairs[: int(m.item())]

    def des2r(self, d1, d2, want_cor=False):
        M = d1.shape[0]
        idx = torch.empty((M,), dtype=torch.int64, device=d1.device)
        cor = torch.empty((M, 60), dtype=torch.float32, device=d1.device) if want_cor else None
        _check(self._lib.yoho_des2r(self._h, _dev(d1, torch.float32, "d1"), _dev(d2, torch.float32, "d2"), M,
                                    C.c_void_p(idx.data_ptr()), C.c_void_p(cor.data_ptr()) if want_cor else None, _stream()))
        return (idx, cor) if want_cor else idx

    def des2r_matched(self, e1, e2, match):
        """Des2R(e1[match[:,1]], e2[match[:,0]]) with the rows read in place (match (M,2) int64 cuda, contiguous)."""
        M = match.shape[0]
        idx = torch.empty((M,), dtype=torch.int64, device=e1.device)
        mp = _dev(match, torch.int64, "match")
        _check(self._lib.yoho_des2r_indexed(self._h, _dev(e1, torch.float32, "e1"), C.c_void_p(match.data_ptr() + 8),
                                            _dev(e2, torch.float32, "e2"), mp, 2, M, C.c_void_p(idx.data_ptr()), None, _stream()))
        return idx

    def partII_forward_matched(self, feat0, feat1, eqv0, eqv1, match, pre_idx, check_range=True):
        """partII_forward(feat1[m1], feat0[m0], eqv1[m1], eqv0[m0], pre_idx) with m0, m1 = match[:,0], match[:,1], rows read
        in place.  Default PartII arithmetic mode only (supports_matched)."""
        if check_range:
            q = self.partII_forward_matched(feat0, feat1, eqv0, eqv1, match, pre_idx, check_range=False)
            if self.partII_overflow():
                m0, m1 = match[:, 0], match[:, 1]
                q = self._repeat_wider("partII", lambda: self.partII_forward(feat1[m1], feat0[m0], eqv1[m1], eqv0[m0], pre_idx, check_range=False))
            return q
        M = match.shape[0]
        quat = torch.empty((M, 4), dtype=torch.float32, device=feat0.device)
        m0p, m1p = C.c_void_p(_dev(match, torch.int64, "match").value), C.c_void_p(match.data_ptr() + 8)
        _check(self._lib.yoho_partII_forward_indexed(
            self._h, _dev(feat1, torch.float32, "feat1"), m1p, _dev(feat0, torch.float32, "feat0"), m0p,
            _dev(eqv1, torch.float32, "eqv1"), m1p, _dev(eqv0, torch.float32, "eqv0"), m0p, 2,
            _dev(pre_idx, torch.int64, "pre_idx"), M, C.c_void_p(quat.data_ptr()), _stream()))
        return quat

    def partII_forward(self, before_eqv0, before_eqv1, after_eqv0, after_eqv1, pre_idx, check_range=True):
        if check_range and self.partII_mode in FP16_PARTII_MODES:
            args = (before_eqv0, before_eqv1, after_eqv0, after_eqv1, pre_idx)
            q = self.partII_forward(*args, check_range=False)
            if self.partII_overflow():
                q = self._repeat_wider("partII", lambda: self.partII_forward(*args, check_range=False))
            return q
        M = before_eqv0.shape[0]
        quat = torch.empty((M, 4), dtype=torch.float32, device=before_eqv0.device)
        _check(self._lib.yoho_partII_forward(
            self._h, _dev(before_eqv0, torch.float32, "before_eqv0"), _dev(before_eqv1, torch.float32, "before_eqv1"),
            _dev(after_eqv0, torch.float32, "after_eqv0"), _dev(after_eqv1, torch.float32, "after_eqv1"),
            _dev(pre_idx, torch.int64, "pre_idx"), M, C.c_void_p(quat.data_ptr()), _stream()))
        return quat

    # ---- estimators ------------------------------------------------------------------------
    def hyp_from_quat(self, quat, idx, k0, k1):
        M = quat.shape[0]
        T = torch.empty((M, 3, 4), dtype=torch.float64, device=quat.device)
        _check(self._lib.yoho_hyp_from_quat(self._h, _dev(quat, torch.float32, "quat"), _dev(idx, torch.int64, "idx"),
                                            _dev(k0, torch.float64, "k0"), _dev(k1, torch.float64, "k1"), M,
                                            C.c_void_p(T.data_ptr()), _stream()))
        return T

    def o_score(self, k0, k1, T, order, H, d):
        """Returns (best_h, best_count, counts (H) int32) as device tensors."""
        M = k0.shape[0]
        res = torch.zeros((2,), dtype=torch.int32, device=k0.device)
        counts = torch.empty((H,), dtype=torch.int32, device=k0.device)
        _check(self._lib.yoho_o_score(self._h, _dev(k0, torch.float64, "k0"), _dev(k1, torch.float64, "k1"), M,
                                      _dev(T, torch.float64, "T"), _dev(order, torch.int64, "order") if order is not None else None,
                                      H, float(d), C.c_void_p(res.data_ptr()), C.c_void_p(res.data_ptr() + 4),
                                      C.c_void_p(counts.data_ptr()), _stream()))
        return res, counts

    def c_ransac(self, k0, k1, triples, reflect, d, want_all=False):
        M, I = k0.shape[0], triples.shape[0]
        best_T = torch.empty((3, 4), dtype=torch.float64, device=k0.device)
        res = torch.zeros((2,), dtype=torch.int32, device=k0.device)
        T_all = torch.empty((I, 3, 4), dtype=torch.float64, device=k0.device) if want_all else None
        counts = torch.empty((I,), dtype=torch.int32, device=k0.device) if want_all else None
        _check(self._lib.yoho_c_ransac(self._h, _dev(k0, torch.float64, "k0"), _dev(k1, torch.float64, "k1"), M,
                                       _dev(triples, torch.int64, "triples"),
                                       _dev(reflect, torch.uint8, "reflect") if reflect is not None else None, I, float(d),
                                       C.c_void_p(best_T.data_ptr()), C.c_void_p(res.data_ptr()), C.c_void_p(res.data_ptr() + 4),
                                       C.c_void_p(T_all.data_ptr()) if want_all else None,
                                       C.c_void_p(counts.data_ptr()) if want_all else None, _stream()))
        return best_T, res, T_all, counts

    def c_ransac_device(self, keys0, keys1, dr_index, max_iter, seed, d, match=None, want_triples=False):
        """YOHO-C with the sampling on the device (yoho_c_ransac_device): keys0 / keys1 (.,3) f64 cuda, addressed through
        the columns of match (M,2) int64 if given (else row m = match m), dr_index (M,) int64.
        Returns (best_T (3,4) f64, res int32[2] = (best_iter, best_count), triples (max_iter,3) int64 or None): device tensors."""
        M = dr_index.shape[0]
        best_T = torch.empty((3, 4), dtype=torch.float64, device=keys0.device)
        res = torch.zeros((2,), dtype=torch.int32, device=keys0.device)
        tri = torch.empty((max_iter, 3), dtype=torch.int64, device=keys0.device) if want_triples else None
        if match is not None:
            i0, i1, istride = C.c_void_p(_dev(match, torch.int64, "match").value), C.c_void_p(match.data_ptr() + 8), 2
        else:
            i0, i1, istride = None, None, 1
        _check(self._lib.yoho_c_ransac_device(self._h, _dev(keys0, torch.float64, "keys0"), i0, _dev(keys1, torch.float64, "keys1"), i1,
                                              istride, _dev(dr_index, torch.int64, "dr_index"), M, int(max_iter), int(seed) & (2 ** 64 - 1),
                                              float(d), C.c_void_p(best_T.data_ptr()), C.c_void_p(res.data_ptr()),
                                              C.c_void_p(res.data_ptr() + 4), C.c_void_p(tri.data_ptr()) if want_triples else None, _stream()))
        return best_T, res, tri

    def register_pair(self, feat0, feat1, eqv0, eqv1, inv0, inv1, keys0, keys1, estimator="yohoo", max_iter=1000, inlier_dist=0.09,
                      seed=0, selected=True):
        """One pair of described fragments in one library call (yoho_register_pair: mutual NN -> Des2R -> PartII + vote, or the
        device-sampled YOHO-C); the GIL is released for its whole duration.  feat / eqv (n,32,60) f32, inv (n,32) f32 = inv_np,
        keys (n,3) f64, all cuda.  The YOHO-O vote order is np.random.RandomState(seed & 0xFFFFFFFF)'s shuffle, the YOHO-C sampling
        stream is `seed`: the result equals pipeline.run_pair(..., order_rng=RandomState(seed & 0xFFFFFFFF), seed=seed).
        -> dict(trans (3,4) f64 or eye(4), best_h, best_count, matches, hypotheses, range_flag)"""
        if estimator not in ("yohoo", "yohoc"):
            raise ValueError(f"estimator must be 'yohoo' or 'yohoc', got {estimator!r}")
        res = PairResultC()
        est = 0 if estimator == "yohoo" else 1
        _check(self._lib.yoho_register_pair(
            self._h, _dev(feat0, torch.float32, "feat0") if est == 0 else None, _dev(feat1, torch.float32, "feat1") if est == 0 else None,
            _dev(eqv0, torch.float32, "eqv0"), _dev(eqv1, torch.float32, "eqv1"), _dev(inv0, torch.float32, "inv0"),
            _dev(inv1, torch.float32, "inv1"), _dev(keys0, torch.float64, "keys0"), _dev(keys1, torch.float64, "keys1"),
            eqv0.shape[0], eqv1.shape[0], est, int(max_iter), float(inlier_dist), int(seed) & (2 ** 64 - 1), 1 if selected else 0,
            C.byref(res), _stream()))
        ok = res.matches > 0 and res.best_count > 0
        trans = np.array(res.trans, dtype=np.float64).reshape(3, 4) if ok else np.eye(4)
        return {"trans": trans, "best_h": int(res.best_h), "best_count": int(res.best_count), "matches": int(res.matches),
                "hypotheses": int(res.hypotheses), "range_flag": bool(res.range_flag)}

    def group_gather(self, keys, pts, feat, g, out, want_idx=False):
        K, n = keys.shape[0], pts.shape[0]
        Rg = np.ascontiguousarray(self.tables.R64[g], dtype=np.float64)
        nn_idx = torch.empty((K,), dtype=torch.int64, device=keys.device) if want_idx else None
        _check(self._lib.yoho_group_gather(self._h, _dev(keys, torch.float64, "keys"), K, _dev(pts, torch.float32, "pts"),
                                           _dev(feat, torch.float32, "feat"), n, int(g), _np_ptr(Rg),
                                           _dev(out, torch.float32, "out"), C.c_void_p(nn_idx.data_ptr()) if want_idx else None,
                                           _stream()))
        return nn_idx

    def group_scatter(self, feat, idx, g, out):
        """out[:, :, g] = feat[idx] (feat (n,32) f32, idx (K,) int64, out (K,32,60) f32), in place"""
        if feat.dim() != 2 or feat.shape[1] != 32:
            raise ValueError(f"group_scatter: feat must be (n, 32) (the group feature's width, YOHO_testset.py:153-166), got {tuple(feat.shape)}")
        if idx.dim() != 1 or tuple(out.shape) != (idx.shape[0], 32, 60) or not 0 <= int(g) < 60:
            raise ValueError(f"group_scatter: idx (K,), out (K, 32, 60), 0 <= g < 60; got idx {tuple(idx.shape)}, out {tuple(out.shape)}, g {g}")
        _check(self._lib.yoho_group_scatter(self._h, _dev(feat, torch.float32, "feat"), feat.shape[0], _dev(idx, torch.int64, "idx"), idx.shape[0],
                                            int(g), _dev(out, torch.float32, "out"), _stream()))

    def group_transfer_batch(self, pts, kidx, Rs, ds_list, feat_list, g0, out):
        """The NN feature transfer of a backbone pass in one library call: for copy b, out[:, :, g0 + b] = feat_list[b][nn(R_b
        pts[kidx], ds_list[b])] (fp32 'SquareL2' search, through the hash grid when set_nn_grid is on).  pts (n,3) f64, kidx (K,)
        int64, Rs list of (3,3), ds_list[b] (m_b,3) f32, feat_list[b] (m_b,32) f32, out (K,32,60) f32; all on the device."""
        nb, K = len(Rs), kidx.shape[0]
        if not (1 <= nb <= 64 and len(ds_list) == nb and len(feat_list) == nb):
            raise ValueError("group_transfer_batch: 1..64 copies, one ds / feat tensor per rotation")
        if tuple(out.shape) != (K, 32, 60) or any(f.dim() != 2 or f.shape[1] != 32 or f.shape[0] != d.shape[0] for f, d in zip(feat_list, ds_list)):
            raise ValueError("group_transfer_batch: out (K,32,60), feat (m,32) and ds (m,3) per copy")
        Rh = np.ascontiguousarray(np.stack([np.asarray(R, dtype=np.float64).reshape(3, 3) for R in Rs]))
        dsp = (C.c_void_p * nb)(*[_dev(d, torch.float32, "ds").value for d in ds_list])
        fp = (C.c_void_p * nb)(*[_dev(f, torch.float32, "feat").value for f in feat_list])
        m = (C.c_int * nb)(*[int(d.shape[0]) for d in ds_list])
        q = torch.empty((K, 3), dtype=torch.float32, device=out.device)
        idx = torch.empty((K,), dtype=torch.int64, device=out.device)
        _check(self._lib.yoho_group_transfer_batch(self._h, _dev(pts, torch.float64, "pts"), _dev(kidx, torch.int64, "kidx"), K, _np_ptr(Rh), nb,
                                                   dsp, fp, m, int(g0), _dev(out, torch.float32, "out"), C.c_void_p(q.data_ptr()),
                                                   C.c_void_p(idx.data_ptr()), _stream()))

    def set_gconv_mode(self, mode):
        """'f32' (direct conv, fp32 MFMA), 'bf16x3' (direct conv, fp32-accurate 3-way bf16 split MFMA),
        'fourier' (group-Fourier domain conv, fp32 MFMA) or 'fp16x2' (direct conv, 2-way fp16 split MFMA) for PartI."""
        _check(self._lib.yoho_set_gconv_mode(self._h, GCONV_MODES[mode]))
        self.gconv_mode = mode

    def set_partI_schedule(self, chunk_kp=0, streams=1):
        """PartI pass of the default mode breadth-first (chunk_kp = 0) or depth-first over chunks of chunk_kp keypoints, the
        chunks on one stream or alternating over two (include/yoho_hip.h); bit-identical results."""
        _check(self._lib.yoho_set_partI_schedule(self._h, int(chunk_kp), int(streams)))

    def set_partII_mode(self, mode):
        """'f32', 'bf16x3', 'fp16x2' (direct cone kernels), 'cgemm' (13-element cone layer as an implicit GEMM) or 'cgemm8' (cgemm with fp8
        correction products) for PartII (include/yoho_hip.h: yoho_set_partII_mode)."""
        _check(self._lib.yoho_set_partII_mode(self._h, PARTII_MODES[mode]))
        self.partII_mode = mode

    def set_nn_prefilter(self, on=True):
        """mutual_nn on large sets through the MFMA pre-filter (default) or by brute force; identical match lists"""
        _check(self._lib.yoho_set_nn_prefilter(self._h, 1 if on else 0))

    def set_fcgf_sort(self, parity=True, cells=1):
        """internal row orders of the FCGF backbone: transposed convolutions over parity-sorted rows (default on), level-0 rows
        grouped by 8^3-voxel cell (0 never, 1 = default: passes of >= 2^18 voxels, 2 always; + 4: coordinate maps through hash
        tables even when the clouds fit rank-ordered bitmaps - the cell sort only exists on that path); identical outputs in the
        caller's row order either way"""
        _check(self._lib.yoho_set_fcgf_sort(self._h, 1 if parity else 0, int(cells)))

    def set_nn_grid(self, cell):
        """3-D nearest-neighbour searches (nn_search with 3 columns, group_gather) through a hash grid with this cell size
        (0 = brute force).  Same answers for any cell; pass the voxel size the target cloud was down-sampled with."""
        _check(self._lib.yoho_set_nn_grid(self._h, float(cell)))

    def clock_probe(self, microseconds, stream=None, out=None):
        """queue the one-wave clock probe (include/yoho_hip.h) on `stream` (a torch stream; default: the current one);
        returns the (3,) int64 device tensor it fills - shader MHz = t[0] / t[1] * t[2] / 1000 once the stream has run"""
        if out is None:
            out = torch.zeros(3, dtype=torch.int64, device=f"cuda:{self.device}")
            torch.cuda.current_stream().synchronize()        # the fill must not run after the probe (it is on another stream)
        st = stream.cuda_stream if stream is not None else _stream()
        _check(self._lib.yoho_clock_probe(self._h, int(microseconds), out.data_ptr(), st))
        return out

    # ---- profiling hook (bench.py) ------------------------------------------------------------
    def set_profiling(self, on=True):
        _check(self._lib.yoho_set_profiling(self._h, 1 if on else 0))

    PHASES = ("voxelise", "coordinate_maps", "kernel_maps", "conv1", "conv3x3_level0", "conv3x3_level1", "conv3x3_level2", "conv3x3_level3",
              "strided_into_level1", "strided_into_level2", "strided_into_level3", "transposed_to_level0", "transposed_to_level1",
              "transposed_to_level2", "heads_1x1_normalise", "nn_feature_transfer")

    def phase_profile(self, on=True):
        """accumulating phase timer of the raw-cloud path (include/yoho_hip.h: yoho_phase_profile); read with phase_read()"""
        _check(self._lib.yoho_phase_profile(self._h, 1 if on else 0))

    def phase_read(self):
        """-> {phase: {"ms", "mfma_flops", "launches"}} accumulated since the last read (waits for the current stream)"""
        ms, fl, ln = (np.zeros(16, dtype=np.float64) for _ in range(3))
        _check(self._lib.yoho_phase_read(self._h, _np_ptr(ms), _np_ptr(fl), _np_ptr(ln), _stream()))
        return {n: {"ms": float(ms[i]), "mfma_flops": float(fl[i]), "launches": int(ln[i])} for i, n in enumerate(self.PHASES)}

    def kernel_ms(self, which):
        ms = C.c_float(-1.0)
        _check(self._lib.yoho_get_kernel_ms(self._h, int(which), C.byref(ms)))
        return float(ms.value)


_ctx_cache = {}


def streams_overlap(ctx, s1, s2, microseconds=300):
    """True when work queued on the torch streams s1 and s2 runs concurrently.  HIP multiplexes a process's streams over a few hardware
    queues (four by default, dealt round-robin in creation order, the null stream among them): two streams that land on the same queue
    run their kernels one after the other however independent they are - about one stream in four shares the null stream's queue
    (measured: a 20 us kernel on such a stream waits for everything queued on the null stream).  Measured, not assumed: the one-wave
    clock probe (yoho_clock_probe spins for `microseconds`) on both streams, timed together - ~1 x the probe when they overlap, 2 x when
    they share a queue."""
    import time
    buf = torch.zeros((2, 3), dtype=torch.int64, device=f"cuda:{ctx.device}")
    torch.cuda.synchronize(ctx.device)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.clock_probe(microseconds, stream=s1, out=buf[0])
        ctx.clock_probe(microseconds, stream=s2, out=buf[1])
        s1.synchronize()
        s2.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best < 1.6e-6 * microseconds


def concurrent_stream(ctx, others, tries=8):
    """a new torch stream (normal priority) that demonstrably overlaps every stream of `others` (streams_overlap): the first candidate
    that passes is kept.  Falls back to the last candidate - correct, merely serialised - when none does."""
    cand = None
    for _ in range(tries):
        cand = torch.cuda.Stream(device=ctx.device)
        if all(streams_overlap(ctx, o, cand) for o in others):
            return cand
    return cand


def get_context(device=None, so3_dir=None, lane=0):
    """Process-wide context per (device, table directory, lane).  lane > 0: a further context with a workspace of its own for work
    queued on another stream while lane 0's runs (the backbone lanes of yoho_extractor / testset_create); it is kept, like lane 0's,
    so that its workspace is sized once per process."""
    dev = torch.cuda.current_device() if device is None else int(device)
    key = (dev, os.path.abspath(so3_dir) if so3_dir else None) + ((int(lane),) if lane else ())
    if key not in _ctx_cache:
        _ctx_cache[key] = Context(dev, so3_dir)
    return _ctx_cache[key]
