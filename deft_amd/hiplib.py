"""ctypes binding of libdeft_hip.so (include/deft_hip.h).

The library is the product: there is NO CPU or PyTorch fallback.  `get_lib()`
raises if deft_amd/lib/libdeft_hip.so (built by `python -m deft_amd.build` /
__graft_entry__.build()) is missing.  The unit tests may hand an explicitly
built test object to `HipLib(path)`; nothing in this package
ever selects it.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdeft_hip.so")

c_fp = C.c_void_p


class GemmDesc(C.Structure):
    """Mirror of DeftGemmDesc (include/deft_hip.h).  `flop_k` (python-side only) overrides
    Ktot in the algorithmic FLOP count when input channels are padding (the 3->4 image)."""
    flop_k = 0
    flop_n = 0          # likewise for output columns that are padding (the 27 -> 32 offset/mask conv)
    _fields_ = [
        ("x", c_fp), ("x2", c_fp), ("w", c_fp), ("scale", c_fp), ("shift", c_fp), ("res", c_fp), ("y", c_fp),
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("ldx", C.c_int),
        ("OH", C.c_int), ("OW", C.c_int), ("Cout", C.c_int), ("ldy", C.c_int), ("ldr", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("Ktot", C.c_int), ("Kpad", C.c_int), ("cin_log2", C.c_int), ("M", C.c_int),
        ("relu", C.c_int), ("Q", C.c_int), ("ldom", C.c_int), ("tile", C.c_int),
        ("Tper", C.c_int), ("u0", C.c_int), ("du", C.c_int), ("v0", C.c_int), ("dv", C.c_int),
        ("stride_w", C.c_int), ("korder", C.c_int),
        ("rowmap", c_fp),
        ("splitk", C.c_int), ("ws", c_fp), ("ws_cnt", c_fp), ("prec", C.c_int),
        ("x3", c_fp), ("w3", c_fp), ("y3", c_fp), ("ldx3", C.c_int), ("ldy3", C.c_int), ("p3_kernel", C.c_int),
        ("fold_w", c_fp), ("fold_y", c_fp), ("fold_n", C.c_int), ("fold_ld", C.c_int),
    ]


class PairMlpDesc(C.Structure):
    """Mirror of DeftPairMlp (include/deft_hip.h)."""
    _fields_ = [("U", c_fp), ("V", c_fp), ("wimg", c_fp), ("s2", c_fp), ("t2", c_fp), ("s3", c_fp), ("t3", c_fp), ("s4", c_fp), ("t4", c_fp),
                ("w5", c_fp), ("out", c_fp), ("b5", C.c_float), ("ldu", C.c_int), ("M", C.c_int), ("Q", C.c_int),
                ("Tper", C.c_int), ("u0", C.c_int), ("du", C.c_int), ("v0", C.c_int), ("dv", C.c_int)]


_SIGS = {
    "deft_version": (C.c_int, []),
    "deft_pieces": (C.c_int, []),
    "deft_last_error": (C.c_char_p, []),
    "deft_conv2d_nhwc": (C.c_int, [C.POINTER(GemmDesc), c_fp]),
    "deft_conv2d_group": (C.c_int, [C.POINTER(GemmDesc), c_fp, C.c_int, c_fp]),
    "deft_dcn_v2_nhwc": (C.c_int, [C.POINTER(GemmDesc), c_fp]),
    "deft_pair_layer": (C.c_int, [C.POINTER(GemmDesc), c_fp]),
    "deft_pair_mlp": (C.c_int, [C.POINTER(PairMlpDesc), c_fp]),
    "deft_pair_mlp_image_bytes": (C.c_int, []),
    "deft_nchw_to_nhwc": (C.c_int, [c_fp, c_fp] + [C.c_int] * 5 + [c_fp]),
    "deft_nhwc_to_nchw": (C.c_int, [c_fp, c_fp] + [C.c_int] * 5 + [c_fp]),
    "deft_preprocess_u8": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp]),
    "deft_maxpool2x2": (C.c_int, [c_fp, c_fp] + [C.c_int] * 6 + [c_fp]),
    "deft_upsample_add": (C.c_int, [c_fp] * 4 + [C.c_int] * 8 + [c_fp, C.c_int, c_fp]),
    "deft_hm_peaks": (C.c_int, [c_fp] + [C.c_int] * 6 + [c_fp] * 3 + [C.c_int, c_fp]),
    "deft_topk": (C.c_int, [c_fp] * 3 + [C.c_int] * 4 + [c_fp] * 3 + [c_fp]),
    "deft_heads_at_peaks": (C.c_int, [c_fp] + [C.c_int] * 5 + [c_fp, C.c_int] + [c_fp] * 5 + [C.c_int] * 2 + [c_fp, c_fp]),
    "deft_peak_rows": (C.c_int, [c_fp] + [C.c_int] * 4 + [c_fp, c_fp]),
    "deft_heads_finish": (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, c_fp, c_fp, C.c_int, c_fp, c_fp]),
    "deft_decode_boxes": (C.c_int, [c_fp, c_fp] + [C.c_int] * 8 + [c_fp] * 4),
    "deft_embed_map": (C.c_int, [c_fp] + [C.c_int] * 5 + [c_fp, c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int, C.c_int, C.c_int, c_fp]),
    "deft_embed_rows": (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, C.c_int, c_fp, c_fp, C.c_int, c_fp]),
    "deft_embed_blend": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp]),
    "deft_affinity_finish": (C.c_int, [c_fp, C.c_int, C.c_int, c_fp, C.c_float, c_fp] + [C.c_int] * 4 + [c_fp, c_fp]),
    "deft_lstm_step": (C.c_int, [c_fp] * 3 + [C.c_int] * 3 + [c_fp] * 8 + [c_fp]),
    "deft_gemm_plan": (C.c_int, [C.POINTER(GemmDesc), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "deft_motion_step": (C.c_int, [c_fp, c_fp] + [C.c_int] * 3 + [c_fp] * 3 + [C.c_int] * 2 + [c_fp] * 7 + [c_fp, c_fp, c_fp]),
    "deft_split_planes": (C.c_int, [c_fp, c_fp, C.c_longlong, C.c_int, C.c_int, C.c_int, c_fp]),
    "deft_split_weights": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, c_fp]),
    "deft_split_weights_halo": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, c_fp]),
    "deft_split_weights_dcn": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, c_fp]),
    "deft_fold_finish": (C.c_int, [c_fp, C.c_int, C.c_longlong, C.c_int, C.c_int, c_fp, c_fp, C.c_int, c_fp]),
    "deft_conv_direct": (C.c_int, [C.POINTER(GemmDesc), c_fp]),
    "deft_split_weights_direct": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_fp]),
    "deft_direct_weight_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "deft_track_similarity": (C.c_int, [c_fp] + [C.c_int] * 2 + [c_fp] * 3 + [C.c_int] * 2 + [c_fp, c_fp]),
    # host-side association helpers (HOST pointers, synchronous)
    "deft_lapjv": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_double, c_fp, c_fp, c_fp]),
    "deft_iou3d_matrix": (C.c_int, [c_fp, C.c_int, c_fp, C.c_int, c_fp]),
    "deft_associate_ddd": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_double, C.c_double, C.c_double,
                                     c_fp, c_fp, c_fp, C.c_double, C.c_double, C.c_double, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "deft_kf_predict": (C.c_int, [c_fp, c_fp, C.c_int]),
    "deft_kf_update": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, c_fp]),
    "deft_greedy_nms": (C.c_int, [c_fp, c_fp, C.c_int, C.c_double, C.c_int, c_fp, c_fp]),
    "deft_track_nodes": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int,
                                   c_fp, c_fp, c_fp, c_fp]),
    "deft_associate_2d": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_double, C.c_double, C.c_double, C.c_int,
                                    c_fp, c_fp, c_fp, C.c_double, C.c_double, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
}
EXPORTS = tuple(_SIGS)
ABI_VERSION = 14


class DeftHipError(RuntimeError):
    pass


TWIN_SUFFIX = "_p3"       # deft_amd/build.py: every entry point of the device-code sources exists a second time, compiled with three bf16 pieces


class HipLib:
    """One arithmetic of one shared object.  libdeft_hip.so carries two (include/deft_hip.h): the entry points `name` (two fp16 pieces per operand:
    three matrix instructions per fp32 product, activations |x| < 4094) and their twins `name_p3` (three bf16 pieces: six products, no range
    limit).  `HipLib(path)` binds the first set, `.twin()` the second -- same library, same process; piece buffers, weight images and plans belong
    to ONE of them (the piece formats differ), so the choice is made per plan (engine._Plan(lib=...))."""

    def __init__(self, path, suffix="", cdll=None):
        if not os.path.exists(path):
            raise DeftHipError(
                "HIP extension missing: %s -- run `python -m deft_amd.build` "
                "(hipcc --offload-arch=gfx950); deft_amd has no CPU fallback" % path)
        self.path = path
        self.suffix = suffix
        self.cdll = C.CDLL(path) if cdll is None else cdll
        self._fn = {}
        for name, (res, args) in _SIGS.items():
            fn = getattr(self.cdll, name + suffix, None) if suffix else None
            if fn is None:
                fn = getattr(self.cdll, name)      # AttributeError if a symbol is not exported (host-only helpers have no twin)
            fn.restype = res
            fn.argtypes = args
            self._fn[name] = fn
        # a build whose kernels run on host memory (the unit-test build of the same sources) exports this marker
        self.host_pointers = hasattr(self.cdll, "deft_host_pointers")
        v = self._fn["deft_version"]()
        if v != ABI_VERSION:
            raise DeftHipError("libdeft_hip ABI version %d, expected %d -- rebuild: python -m deft_amd.build" % (v, ABI_VERSION))
        # operand pieces of the split arithmetic of this set: 3 = bf16 x 3 (six products), 2 = fp16 x 2 (three products); sizes the piece buffers
        self.pieces = int(self._fn["deft_pieces"]())
        self._twin = None

    @property
    def has_twin(self):
        return not self.suffix and hasattr(self.cdll, "deft_pieces" + TWIN_SUFFIX)

    def twin(self):
        """The three-bf16-piece entry points of the same library (None when this set is already range-free or the library carries one set)."""
        if self._twin is None and self.has_twin and self.pieces == 2:
            self._twin = HipLib(self.path, TWIN_SUFFIX, self.cdll)
            assert self._twin.pieces == 3
        return self._twin

    def last_error(self):
        return self._fn["deft_last_error"]().decode()

    profile = None      # set to a list to record (entry, algorithmic_flops, event0, event1, shape, algorithmic_bytes) per call

    def call(self, name, *args):
        prof = self.profile
        if prof is not None:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = self._fn[name](*args)
        if rc != 0:
            raise DeftHipError("%s failed (%d): %s" % (name + self.suffix, rc, self.last_error()))
        if prof is not None:
            e1.record()
            fl, info, nbytes = 0.0, "", 0.0

            def alg_bytes(d):      # operands read once + output written once (fp32)
                if name == "deft_pair_layer":
                    return 4.0 * (d.M * d.Cout + d.Cout * d.Ktot + 2.0 * (d.M // max(d.Q, 1) + d.Q) * d.Ktot)
                rows_in = d.M * d.KH * d.KW if d.rowmap else d.N * d.H * d.W
                return 4.0 * (min(rows_in, d.N * d.H * d.W) * d.Cin + d.M * d.Cout * (2 if d.res else 1) + d.Cout * d.Ktot
                              + (d.M * 27 if name == "deft_dcn_v2_nhwc" else 0))
            ceil = 157.3            # TFLOP/s ceiling of the instructions the launch issues: v_mfma_f32_32x32x2_f32 ...
            if name == "deft_conv2d_group":
                ds = args[0]
                fl = sum(2.0 * ds[i].M * ds[i].Cout * ds[i].Ktot for i in range(args[2]))
                nbytes = sum(alg_bytes(ds[i]) for i in range(args[2]))
                info = "group of %d" % args[2]
            elif name == "deft_pair_mlp":
                d = args[0]._obj
                fl = 2.0 * d.M * (512 * 256 + 256 * 128 + 128 * 64 + 64)
                nbytes = 4.0 * (d.M + 2.0 * (d.M // max(d.Q, 1) + d.Q) * 512) + 672.0 * 1024
                ceil = 2500.0 / (6 if self.pieces == 3 else 3)
                info = "M=%d pair MLP 512-256-128-64-1 fused split" % d.M
            elif name in ("deft_conv2d_nhwc", "deft_dcn_v2_nhwc", "deft_pair_layer", "deft_conv_direct"):
                d = args[0]._obj
                fl = 2.0 * d.M * (d.flop_n if d.flop_n else d.Cout) * (d.flop_k if d.flop_k else d.Ktot)
                nbytes = alg_bytes(d)
                info = "M=%d N=%d K=%d %dx%d s%d @%dx%d" % (d.M, d.Cout, d.Ktot, d.KH, d.KW, d.stride, d.H, d.W)
                if name == "deft_conv_direct":
                    ceil = 2500.0 / (6 if self.pieces == 3 else 3)   # six v_mfma_f32_16x16x32_bf16 / three v_mfma_f32_16x16x32_f16 per fp32 product
                    info += " split direct"
                elif self.split_arithmetic(name, d):
                    # ... or six v_mfma_f32_32x32x16_bf16 (three bf16 pieces) / three v_mfma_f32_32x32x16_f16 (two fp16 pieces) per fp32 product
                    ceil = 2500.0 / (6 if self.pieces == 3 else 3)
                    info += " split" + (" patch" if d.p3_kernel in (2, 3) else " halo" if d.p3_kernel else (" x3" if d.x3 else ""))
            prof.append((name, fl, e0, e1, info, nbytes, ceil))

    def split_arithmetic(self, name, d):
        """Does this launch run on the bf16 matrix instructions (prec 1 honoured)?  The pre-split kernels always do; igemm.hip
        only on its 1-stage tiles with BN >= 64 (launch_igemm) -- the tile is the library's own choice when tile == 0."""
        if d.prec != 1:
            return False
        if d.x3 or d.p3_kernel in (2, 3):
            return True
        tile = d.tile
        if (tile & 0xffff) == 0:
            if name == "deft_pair_layer":
                return True                                  # 128 x 128 or 128 x 64
            t, s, wf, wt = C.c_int(), C.c_int(), C.c_longlong(), C.c_int()
            if self._fn["deft_gemm_plan"](C.byref(d), 0 if name == "deft_conv2d_nhwc" else 1, C.byref(t), C.byref(s), C.byref(wf), C.byref(wt)) != 0:
                return False
            tile = t.value
        return (tile & 0xffff) >= 64 and not (tile >> 29) & 1


_lib = None


def load(path):
    """Load a specific shared object (tests use this with the emulator build)."""
    global _lib
    _lib = HipLib(path)
    return _lib


def get_lib():
    global _lib
    if _lib is None:
        lib = HipLib(os.environ.get("DEFT_HIP_LIB", LIB_PATH))     # override: A/B builds of the same HIP sources
        if lib.host_pointers and os.environ.get("DEFT_TEST_HOST_POINTERS") != "1":
            # the unit-test build of the kernels (the SIMT emulator under tests/) runs on host memory: never a product path -- tests hand it over
            # explicitly (hiplib.load / the `lib=` arguments), an environment variable alone must not select it
            raise DeftHipError("%s is the host-memory test build of the kernels, not libdeft_hip.so (deft_amd has no CPU path)" % lib.path)
        if os.environ.get("DEFT_ARITH", "") == "bf16x3" and lib.twin() is not None:
            lib = lib.twin()                    # the whole process on the three-bf16-piece entry points (A/B runs, bench.py's side line)
        _lib = lib
        if not lib.host_pointers:
            import atexit
            atexit.register(_idle_at_exit)
    return _lib


def ptr(t):
    """Device (or, under the test emulator, host) address of a tensor; None -> NULL."""
    if t is None:
        return None
    assert t.dtype in (torch.float32, torch.int32, torch.float64, torch.uint8), t.dtype
    return C.c_void_p(t.data_ptr())


def _idle_at_exit():
    """Interpreter shutdown with launches, graph replays or copies still in flight tears down streams, graphs and pinned buffers under
    them: wait for the device first (registered when the library is first loaded on a GPU)."""
    try:
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
    except Exception:
        pass


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # one C call instead of building a torch.cuda.Stream object per launch


def stream_ptr(device):
    """The current HIP stream of `device` as the void* every entry point takes (None on the host test build)."""
    if device.type == "cuda":
        if _raw_stream is not None:
            idx = device.index
            return C.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return None
