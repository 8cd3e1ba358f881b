"""ctypes binding of libfsraft.so (the C ABI declared in include/fsraft.h).

This is the only place the shared library is loaded.  There is NO fallback: if the
library is missing or a kernel launch fails, the caller gets a RuntimeError.
`import torch` happens first on purpose so that libfsraft's NEEDED libamdhip64.so.7
resolves to the HIP runtime torch already loaded (one runtime per process).
"""
import ctypes
import functools
import os
from ctypes import POINTER, Structure, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FSRAFT_LIB_PATH") or os.path.join(_HERE, "libfsraft.so")   # (override: experiment builds)
_lib = None

c_float_p = c_void_p      # device pointers travel as integers


class ConvDesc(Structure):
    """Mirror of struct fsraft_conv_desc (include/fsraft.h)."""
    _fields_ = [
        ("src", c_void_p * 3), ("srcC", c_int * 3), ("srcld", c_int * 3), ("nsrc", c_int),
        ("wpk", c_void_p), ("bias", c_void_p), ("wpk_split", c_void_p),
        ("B", c_int), ("H", c_int), ("W", c_int), ("KH", c_int), ("KW", c_int), ("N", c_int),
        ("dst", c_void_p * 3), ("dst_bs", c_int64 * 3), ("dst_ps", c_int64 * 3), ("dst_cs", c_int64 * 3),
        ("dst_n0", c_int * 3), ("dst_acc", c_int * 3), ("ndst", c_int),
        ("relu", c_int), ("alpha", c_float),
        ("epi", c_int),
        ("h", c_void_p), ("ldh", c_int), ("z", c_void_p), ("ldz", c_int),
        ("aux1", c_void_p), ("ld1", c_int), ("aux2", c_void_p), ("ld2", c_int), ("hid", c_int),
        ("pre", c_void_p), ("ldpre", c_int),
        ("rmask", c_void_p * 3), ("ldmask", c_int * 3), ("maskc", c_int * 3),
        ("wpk_frag", c_void_p), ("pad_h1", c_int), ("pad_w1", c_int),
        ("ws", c_void_p), ("ws_floats", c_int64),
        ("src_amax", c_void_p * 3), ("w_amax", c_void_p), ("dst_amax", c_void_p * 3),
    ]


class PackJob(Structure):
    """Mirror of fsraft_pack_job (include/fsraft.h)."""
    _fields_ = [
        ("w", c_void_p * 3), ("rows", c_int * 3), ("npiece", c_int),
        ("wpk", c_void_p),
        ("cin_full", c_int), ("kh", c_int), ("kw", c_int),
        ("srcC", c_int * 3), ("srcOff", c_int * 3), ("nsrc", c_int),
        ("mode", c_int), ("flags", c_int),
        ("scale", c_float), ("accumulate", c_int),
        ("amax", c_void_p),
    ]


_PP = POINTER(c_void_p)
_IP = POINTER(c_int)
_S = c_void_p   # hipStream_t

SIGNATURES = {
    "fsraft_corr_build": [c_void_p, c_void_p, _PP, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, _S],
    "fsraft_corr_unpool_bwd": [_PP, c_int, c_int, c_int, c_int, _S],
    "fsraft_corr_lookup_fwd": [_PP, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_corr_lookup_bwd": [_PP, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_altcorr_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_altcorr_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_upsample_fwd": [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, _S],
    "fsraft_upsample_bwd": [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_upflow8_fwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, _S],
    "fsraft_upflow8_bwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, _S],
    "fsraft_conv_ktot": [_IP, c_int, c_int, c_int],
    "fsraft_conv_forward": [POINTER(ConvDesc), _S],
    "fsraft_conv_forward_stats": [POINTER(ConvDesc), c_void_p, c_void_p, c_int, POINTER(c_int), _S],
    "fsraft_conv_wgrad": [c_void_p, c_int, c_int, _PP, _IP, _IP, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, _PP, _S],
    "fsraft_conv_wgrad_multi": [_PP, c_int, c_int, c_int, _PP, _IP, _IP, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, _PP, _PP, _S],
    "fsraft_amax_jobs": [_PP, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), _PP, c_int, _S],
    "fsraft_amax": [c_void_p, c_int64, c_int64, c_int64, c_void_p, _S],
    "fsraft_abi_version": [],
    "fsraft_conv_small_fwd": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_conv_small_wgrad": [_PP, _PP, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_conv_small_dgrad": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_pack_conv_weights": [POINTER(PackJob), c_int, _S],
    "fsraft_pack_conv_weight": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, _IP, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_set_rec_mfma16": [c_int],
    "fsraft_set_build_kernel": [c_int],
    "fsraft_set_dvol_policy": [c_int],
    "fsraft_set_dvol_box": [c_int],
    "fsraft_set_ktile_exact": [c_int],
    "fsraft_set_lookup_policy": [c_int],
    "fsraft_set_upsample_kernel": [c_int],
    "fsraft_conv_workspace": [c_void_p, c_int64],
    "fsraft_set_arithmetic": [c_int],
    "fsraft_get_arithmetic": [],
    "fsraft_set_tuning": [c_int, c_int],
    "fsraft_adamw_flat": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_float, c_float, c_float,
                          c_float, c_void_p, c_void_p, _S],
    "fsraft_stream_capture_id": [_S, c_void_p],
    "fsraft_stem_slots": [],
    "fsraft_stem7x7s2_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_stem7x7s2_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, _S],
    "fsraft_set_lookup_qb": [c_int],
    "fsraft_set_build_split": [c_int],
    "fsraft_set_gemm_split": [c_int],
    "fsraft_gemm_tn_split": [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, _S],
    "fsraft_gemm_f32": [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, _S],
    "fsraft_nchw_to_nhwc": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_nhwc_to_nchw": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_im2col7": [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_col2im7": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, _S],
    "fsraft_flow_to_nhwc": [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_nhwc_to_flow": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, _S],
    "fsraft_relu_bwd": [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, _S],
    "fsraft_gru_bwd1": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, _S],
    "fsraft_gru_bwd2": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, _S],
    "fsraft_col_sum": [c_void_p, c_int, c_int64, c_int, c_void_p, c_float, _S],
    "fsraft_softmax_rows": [c_void_p, c_int64, c_int, _S],
    "fsraft_softmax_rows_bwd": [c_void_p, c_void_p, c_int64, c_int, _S],
    "fsraft_softmax_rows_rec": [c_void_p, c_int64, c_int, _S],
    "fsraft_softmax_rows_bwd_rec": [c_void_p, c_void_p, c_int64, c_int, c_void_p, _S],
    "fsraft_gma_mix_fwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, _S],
    "fsraft_gma_mix_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, _S],
    "fsraft_inorm_relu_fwd": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, _S],
    "fsraft_inorm_relu_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, _S],
    "fsraft_affine_relu_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, _S],
    "fsraft_affine_relu_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, _S],
    "fsraft_sequence_loss": [_PP, _PP, POINTER(c_float), c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_corr_pool_pyramid": [_PP, c_int, c_int64, c_int, c_int, _S],
    "fsraft_corr_pool_pyramid_same": [_PP, c_int, c_int64, c_int, c_int, _S],
    "fsraft_corr_lookup_fwd_same": [_PP, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_set_norm_blocks": [c_int],
    "fsraft_get_tuning": [c_int],
    "fsraft_space_to_depth2": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_forward_interpolate": [c_void_p, c_void_p, c_int, c_int, _S],
    "fsraft_inorm_relu_cl_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_inorm_relu_cl_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_int, c_int, c_void_p, _S],
    "fsraft_affine_relu_cl_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_affine_relu_cl_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_int, c_int, c_int, c_void_p, _S],
    "fsraft_axpby": [c_void_p, c_void_p, c_float, c_float, c_int64, _S],
    "fsraft_sum_n": [_PP, c_int, c_void_p, c_int64, c_int, _S],
    "fsraft_bn_fold": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, _S],
    "fsraft_bn_fold_bwd": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, _S],
    "fsraft_vol_layout": [c_int, c_int, c_int, _IP],
    "fsraft_corr_build_tiled": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, _S],
    "fsraft_corr_build_rec": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, _S],
    "fsraft_corr_lookup_tiled_fwd": [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_corr_dvol_build": [_PP, _PP, POINTER(c_int64), c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64,
                               c_int64, c_void_p, c_void_p, c_void_p, _S],
    "fsraft_amax_scaled": [c_void_p, c_float, c_void_p, _S],
    "fsraft_set_alt_tile": [c_int],
    "fsraft_set_alt_rough_pct": [c_int],
    "fsraft_altcorr_fused_fwd": [c_void_p, _PP, c_int, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_altcorr_mfma_fwd": [c_void_p, _PP, c_void_p, _PP, c_int, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, _PP, c_void_p, _S],
    "fsraft_corr_f2cat": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_corr_dfmap2": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, _S],
    "fsraft_corr_f2cat_rec": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, _S],
    "fsraft_to_records": [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, _S],
    "fsraft_gemm_rec_tn": [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int,
                           c_float, c_int, c_int, c_void_p, c_void_p, _S],
    "fsraft_gemm_rec_nt": [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int,
                           c_float, c_int, c_int, c_void_p, c_void_p, _S],
    "fsraft_gemm_rec_nt_list": [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                c_float, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, _S],
    "fsraft_gemm_rec_tn_list": [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                c_float, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, _S],
    "fsraft_corr_bwd_ktiles": [_PP, POINTER(c_int64), c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p,
                               c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, _S],
}


def load():
    """Load libfsraft.so once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C flow_supervisor_amd/csrc`). There is no CPU/eager fallback for the RAFT hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = lib
    # Arithmetic per GEMM family: FSRAFT_ARITHMETIC=0 exact-fp32 MFMA everywhere, 1 (default) fp16x3 products of scaled operands;
    # FSRAFT_CONV_SPLIT / FSRAFT_WGRAD_SPLIT / FSRAFT_BUILD_SPLIT switch one family (the parity suite runs both modes of each).
    arith = os.environ.get("FSRAFT_ARITHMETIC")
    if arith is not None:
        lib.fsraft_set_arithmetic(int(arith))
    split = os.environ.get("FSRAFT_CONV_SPLIT")
    if split is not None:
        lib.fsraft_set_tuning(3, int(split))
    bsplit = os.environ.get("FSRAFT_BUILD_SPLIT")
    if bsplit is not None:
        lib.fsraft_set_build_split(int(bsplit))
    # FSRAFT_TUNING="key=value,key=value": fsraft_set_tuning keys of include/fsraft_tuning.h for A/B runs of scripts/ (round 3 had one
    # environment variable per key)
    for kv in filter(None, os.environ.get("FSRAFT_TUNING", "").split(",")):
        key, val = (int(v) for v in kv.split("="))
        if lib.fsraft_set_tuning(key, val) != 0:
            raise RuntimeError(f"FSRAFT_TUNING: key {key} is not a tuning key of this libfsraft (include/fsraft_tuning.h)")
    wsplit = os.environ.get("FSRAFT_WGRAD_SPLIT")
    if wsplit is not None:
        lib.fsraft_set_tuning(4, int(wsplit))
    return lib


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"libfsraft: {what} failed with status {rc} "
                           f"({'bad argument' if rc == 1 else 'kernel launch error'})")


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def ptr_array(tensors):
    arr = (c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return ctypes.cast(arr, _PP), arr


def int_array(vals):
    return (c_int * len(vals))(*vals)


def require_cuda_f32(*tensors):
    """Every tensor fp32, on ONE cuda device, and that device the calling thread's current one: the kernels are enqueued on
    the current device's current stream (`stream()`), so a tensor living elsewhere would be dereferenced by the wrong GPU.  The
    module-level entry points (`on_tensor_device`) make their input's device current themselves; a caller of `ops.*` does it
    with `torch.cuda.device(t.device)`, as the reference's nn.DataParallel worker threads do (pytorch/train.py:192)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("fsraft ops need CUDA (ROCm) tensors; the hot path has no CPU implementation")
        if t.dtype != torch.float32:
            raise RuntimeError(f"fsraft ops are fp32-only, got {t.dtype}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"fsraft ops need all tensors on one device, got {dev} and {t.device}")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError(f"fsraft op called with tensors on {dev} while cuda:{torch.cuda.current_device()} is the current device; "
                           f"wrap the call in `with torch.cuda.device({dev.index}):`")


def _first_cuda_device(values):
    for v in values:
        if isinstance(v, torch.Tensor):
            if v.is_cuda:
                return v.device
        elif isinstance(v, (list, tuple)):
            d = _first_cuda_device(v)
            if d is not None:
                return d
    return None


def on_tensor_device(fn):
    """Decorator of the path's public entry points (the reference's callables of SURVEY.md 8b): run `fn` with the device of its
    first CUDA tensor argument as the thread's current device, so that a single process driving several GPUs -- one host thread
    per device, the reference's `nn.DataParallel(L2L(args))`, pytorch/train.py:192 -- launches on the tensors' device and its
    current stream whatever device the thread had selected (the reference's extension launches on the legacy default stream
    without a guard, alt_cuda_corr/correlation_kernel.cu:260-323)."""
    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        dev = _first_cuda_device(args)
        if dev is None and kwargs:
            dev = _first_cuda_device(kwargs.values())
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return guarded
