"""ctypes binding of the C ABI declared in ``include/howl_hip.h``.

``SIGNATURES`` is the single table of entry points (name -> argtypes); ``tests/test_cabi.py`` checks it against the
header and against the symbols the built library exports.  ``get()`` loads the in-tree ``libhowl_hip.so`` and fails
loudly when it is absent -- there is no fallback path.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p
from pathlib import Path

# HOWL_HIP_LIBRARY selects another build of the same C ABI (diagnostic kernel variants, tools/variants.py)
LIB_PATH = Path(os.environ.get("HOWL_HIP_LIBRARY") or Path(__file__).resolve().parent / "libhowl_hip.so")
MAX_MELS = 96                                                                       # HOWL_MAX_MELS
FB_COLS = 48
FB_PACKED_FLOATS = 260 * FB_COLS + 17 * 64 * 4 + 17 * (FB_COLS // 4) * 64 + 32      # HOWL_FB_PACKED_FLOATS: one bank


def fb_packed_floats(n_mels: int) -> int:
    """``howl_fb_packed_floats``: a packed filterbank is one bank of up to 48 mel bins, or two back to back."""
    return FB_PACKED_FLOATS * (1 if n_mels <= FB_COLS else 2)


P = c_void_p  # device pointer
STREAM = c_void_p


class HowlMelPoints(ctypes.Structure):
    _fields_ = [("f", c_float * (MAX_MELS + 2))]


class HowlRes8Params(ctypes.Structure):
    _fields_ = [("conv0_w", P), ("conv_w", P * 6), ("bn_running_mean", P * 6), ("bn_running_var", P * 6),
                ("bn_num_batches", P * 6), ("out_w", P), ("out_b", P)]


class HowlRes8Grads(ctypes.Structure):
    _fields_ = [("conv0_w", P), ("conv_w", P * 6), ("out_w", P), ("out_b", P)]


class HowlRes8Saved(ctypes.Structure):
    _fields_ = [("s", P * 7), ("bn_stats", P), ("pooled", P), ("mask0", P)]


class HowlLstmParams(ctypes.Structure):
    _fields_ = [("w_ih", P), ("w_hh", P), ("b_ih", P), ("b_hh", P)]


class HowlLstmGrads(ctypes.Structure):
    _fields_ = [("w_ih", P), ("w_hh", P), ("b_ih", P), ("b_hh", P)]


class HowlHeadParams(ctypes.Structure):
    _fields_ = [("w1", P), ("b1", P), ("w2", P), ("b2", P)]


class HowlHeadGrads(ctypes.Structure):
    _fields_ = [("w1", P), ("b1", P), ("w2", P), ("b2", P)]


class HowlCtcMean(ctypes.Structure):
    _fields_ = [("nll", P), ("target_lengths", P), ("B", c_int), ("loss", P)]


class HowlLogmelArgs(ctypes.Structure):
    _fields_ = [("pcm", P), ("B", c_int), ("L", c_int), ("ld", c_long), ("fbp", P), ("M", c_int), ("log_eps", c_float), ("zmuv", P),
                ("out", P), ("layout", c_int)]


class HowlAdamW(ctypes.Structure):
    _fields_ = [("p", P), ("g", P), ("m", P), ("v", P), ("n", c_size_t), ("lr", c_float), ("beta1", c_float), ("beta2", c_float),
                ("eps", c_float), ("weight_decay", c_float), ("step", c_int), ("grad_scale", c_float)]


class HowlLstmSaved(ctypes.Structure):
    _fields_ = [("gx", P), ("gates", P), ("c", P), ("hseq", P), ("dgates", P), ("t_out", c_int), ("x_frames", c_int)]


class HowlMbLayer(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("kind", "cin", "cout", "stride", "pad_h", "pad_w", "act", "bias", "pool", "res_src",
                                     "feat", "sub", "wrapped")] + \
               [(n, ctypes.c_longlong) for n in ("w_off", "b_off", "gamma_off", "beta_off", "rmean_off", "rvar_off")]


SIGNATURES = {
    "howl_version": [POINTER(c_int), POINTER(c_int)],
    "howl_profile_enable": [c_int],
    "howl_profile_read": [c_char_p, POINTER(c_double), POINTER(c_int), c_int],
    "howl_profile_read_work": [c_char_p, POINTER(c_double), POINTER(c_int), POINTER(c_double), c_int],
    "howl_shutdown": [],
    "howl_fb_pack": [P, c_int, P, STREAM],
    "howl_fb_from_points": [POINTER(HowlMelPoints), c_int, c_float, P, STREAM],
    "howl_logmel_fwd": [P, c_int, c_int, c_long, P, c_int, c_float, P, P, c_int, STREAM],
    "howl_deltas_fwd": [P, c_int, c_int, c_int, P, P, STREAM],
    "howl_zmuv_update": [P, c_size_t, P, P, P, P, STREAM],
    "howl_zmuv_update_masked": [P, P, c_size_t, c_double, P, P, P, P, STREAM],
    "howl_zmuv_pair": [P, P, P, STREAM],
    "howl_zmuv_apply": [P, c_size_t, P, P, STREAM],
    "howl_collate_augment": [P, c_long, P, P, P, P, P, P, ctypes.c_ulonglong, c_int, c_int, P, STREAM],
    "howl_collate_augment_mix": [P, c_long, P, P, P, P, P, P, ctypes.c_ulonglong, P, c_long, P, P, P, c_int, c_int, P, STREAM],
    "howl_collate_augment_window": [P, c_long, P, P, P, P, P, P, ctypes.c_ulonglong, P, c_long, P, P, P, P, c_int, c_int, P,
                                    STREAM],
    "howl_gather_windows": [P, c_long, P, P, P, P, c_int, c_int, P, STREAM],
    "howl_specaug_mask": [P, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_long, P, P, P, P, STREAM],
    "howl_res8_fwd": [POINTER(HowlRes8Params), P, c_long, c_long, c_long, c_int, c_int, c_int, c_int, c_int,
                      POINTER(HowlRes8Saved), P, P, c_size_t, STREAM],
    "howl_res8_fwd_long": [POINTER(HowlRes8Params), P, c_long, c_long, c_long, c_int, c_int, c_int, c_int, P, P, c_size_t, STREAM],
    "howl_res8_bwd": [POINTER(HowlRes8Params), P, c_long, c_long, c_long, c_int, c_int, c_int, c_int,
                      POINTER(HowlRes8Saved), P, POINTER(HowlRes8Grads), P, c_size_t, STREAM],
    "howl_res8_bwd_part": [POINTER(HowlRes8Params), P, c_long, c_long, c_long, c_int, c_int, c_int, c_int,
                           POINTER(HowlRes8Saved), P, POINTER(HowlRes8Grads), P, c_size_t, c_int, STREAM],
    "howl_res8_fwd_xent": [POINTER(HowlRes8Params), P, c_long, c_long, c_long, c_int, c_int, c_int, c_int,
                           POINTER(HowlRes8Saved), P, P, P, P, P, c_size_t, STREAM],
    "howl_res8_bwd_xent": [POINTER(HowlRes8Params), P, c_long, c_long, c_long, c_int, c_int, c_int, c_int,
                           POINTER(HowlRes8Saved), P, P, P, POINTER(HowlRes8Grads), P, c_size_t, c_int, POINTER(HowlAdamW), STREAM],
    "howl_dropout_mask": [P, c_size_t, c_float, ctypes.c_ulonglong, STREAM],
    "howl_xent_fwd_bwd": [P, P, c_int, c_int, P, P, STREAM],
    "howl_ctc_loss": [P, c_long, c_long, c_int, c_int, c_int, P, c_long, c_int, P, P, c_int, P, P, P, c_long, c_long, P, c_size_t,
                      STREAM],
    "howl_lstm_fwd": [POINTER(HowlLstmParams), P, c_int, c_int, c_int, P, P, P, POINTER(HowlLstmSaved), P, P, P, c_size_t,
                      STREAM],
    "howl_lstm_fwd_next": [POINTER(HowlLstmParams), P, c_int, c_int, c_int, P, P, P, POINTER(HowlLstmSaved), P, P, P, c_size_t,
                           POINTER(HowlLogmelArgs), STREAM],
    "howl_lstm_bwd": [POINTER(HowlLstmParams), P, c_int, c_int, c_int, P, P, POINTER(HowlLstmSaved), P, P, P,
                      POINTER(HowlLstmGrads), P, c_size_t, STREAM],
    "howl_head_fwd": [POINTER(HowlHeadParams), P, c_int, c_long, c_long, c_int, c_int, c_int, c_int, P, P, STREAM],
    "howl_head_bwd": [POINTER(HowlHeadParams), P, c_int, c_long, c_long, c_int, c_int, c_int, c_int, P, P, P, P,
                      POINTER(HowlHeadGrads), POINTER(HowlCtcMean), P, c_size_t, STREAM],
    "howl_seq_head_ctc": [POINTER(HowlHeadParams), P, c_long, c_long, c_int, c_int, c_int, c_int, c_int, P, c_long, c_int, P, P, c_int, P, P,
                          P, P, P, c_size_t, STREAM],
    "howl_seq_head_ctc_supported": [c_int, c_int, c_int, c_int, c_int, c_int],
    "howl_seq_lstm_bwd": [POINTER(HowlHeadParams), c_int, c_int, P, P, P, P, POINTER(HowlHeadGrads), POINTER(HowlCtcMean), P,
                          c_size_t, POINTER(HowlLstmParams), P, c_int, c_int, c_int, P, P, POINTER(HowlLstmSaved),
                          POINTER(HowlLstmGrads), P, c_size_t, POINTER(HowlAdamW), STREAM],
    "howl_adamw_step": [P, P, P, P, c_size_t, c_float, c_float, c_float, c_float, c_float, c_int, c_float, STREAM],
    "howl_mobilenet_layer": [c_int, POINTER(HowlMbLayer)],
    "howl_mobilenet_fwd": [P, P, c_int, P, c_long, c_long, c_long, c_int, c_int, c_int, c_int, P, c_float, P, P, c_size_t,
                           STREAM],
    "howl_mobilenet_bwd": [P, c_int, P, c_long, c_long, c_long, c_int, c_int, c_int, P, c_float, P, P, P, c_size_t, STREAM],
}
# entry points that do not return an int status
SIZE_FUNCS = {"howl_fb_packed_floats": [c_int], "howl_res8_workspace_bytes": [c_int, c_int], "howl_res8_long_workspace_bytes": [c_int, c_int],
              "howl_res8_workspace_bytes_mels": [c_int, c_int, c_int], "howl_res8_saved_floats": [c_int, c_int, c_int],
              "howl_res8_eval_workspace_bytes_mels": [c_int, c_int, c_int], "howl_res8_row_strips": [c_int], "howl_res8_long_workspace_bytes_mels": [c_int, c_int, c_int], "howl_lstm_workspace_bytes": [c_int, c_int],
              "howl_lstm_needs_gx": [POINTER(HowlLstmParams), c_int, c_int, c_int, c_int],
              "howl_head_workspace_bytes": [c_int, c_int, c_int],
              "howl_mobilenet_num_layers": [],
              "howl_mobilenet_param_floats": [c_int], "howl_mobilenet_buffer_floats": [],
              "howl_mobilenet_workspace_bytes": [c_int, c_int, c_int, c_int],
              "howl_ctc_supported": [c_int, c_int, c_int], "howl_ctc_workspace_floats": [c_int, c_int]}


class HowlHipError(RuntimeError):
    pass


class Library:
    def __init__(self, path):
        path = Path(path)
        if not path.exists():
            raise HowlHipError(
                f"{path} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
                "(build()) first; howl_amd has no CPU fallback.")
        self.path = path
        self.cdll = ctypes.CDLL(str(path))
        self.cdll.howl_last_error.restype = c_char_p
        self.cdll.howl_last_error.argtypes = []
        for name, argtypes in SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = c_int
            fn.argtypes = argtypes
        for name, argtypes in SIZE_FUNCS.items():
            fn = getattr(self.cdll, name)
            fn.restype = c_size_t
            fn.argtypes = argtypes

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise HowlHipError(f"{name} failed ({rc}): {self.cdll.howl_last_error().decode()}")


_LIB = None


def get() -> Library:
    global _LIB
    if _LIB is None:
        _LIB = Library(LIB_PATH)
        import atexit
        atexit.register(_LIB.cdll.howl_shutdown)     # side HIP queues / events go before the runtime does
    return _LIB
