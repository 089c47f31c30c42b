"""ctypes binding of libs2p_hip.so (C ABI declared in include/s2p_hip.h).

The product path has NO CPU fallback: `lib()` raises if the shared library is missing, and every op
raises if a tensor is not on a HIP device.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "csrc", "libs2p_hip.so")     # (tools/uselib.py points this at a second build for an A/B; no environment variable is read here)

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SWISH = 0, 1, 2, 3, 4
EPI_STORE, EPI_ADD, EPI_MUL_ACTGRAD = 0, 1, 2

c_int, c_float, c_void_p, c_int64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "dtype", "N", "H", "W", "Cin", "x_pitch", "Ho", "Wo", "Cout", "y_pitch",
        "KH", "KW", "stride", "pad", "transposed", "reflect", "groups", "x_gstride", "y_gstride", "cin_real")]


class L1Job(ctypes.Structure):
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("grad_a", c_void_p), ("count", c_int64), ("scale", c_float),
                ("loss_out", c_void_p)]


class WgradJob(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("db", c_void_p)]


class PackJob(ctypes.Structure):
    _fields_ = [("src", c_void_p), ("dst_fwd", c_void_p), ("dst_bwd", c_void_p),
                ("R", ctypes.c_int32), ("T", ctypes.c_int32), ("C", ctypes.c_int32),
                ("Cpad", ctypes.c_int32), ("Rrow", ctypes.c_int32), ("r_off", ctypes.c_int32),
                ("dtype", ctypes.c_int32)]


# name -> argtypes; every entry returns int unless listed in _RESTYPE
_P = c_void_p
_DESC = ctypes.POINTER(ConvDesc)
SIGNATURES = {
    "s2p_version": [],
    "s2p_last_error": [],
    "s2p_conv2d_fwd": [_DESC, _P, _P, _P, _P, _P, c_int, c_float, c_int, _P],
    "s2p_conv2d_dgrad": [_DESC, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P],
    "s2p_conv2d_fwd_workspace": [_DESC, c_int],
    "s2p_conv2d_fwd_ws": [_DESC, _P, _P, _P, _P, _P, c_int, c_float, c_int, _P, ctypes.c_size_t, _P],
    "s2p_conv2d_fwd_mat": [_DESC, _P, _P, _P, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_float, c_float, _P, c_int, _P, _P,
                           ctypes.c_size_t, _P],
    "s2p_conv2d_dgrad_mat": [_DESC, _P, _P, _P, _P, _P, c_int, _P, _P, c_int, _P, c_int, c_int, c_float, c_float, _P, _P, c_int, _P, c_int,
                             _P, c_int, _P, c_int, _P, ctypes.c_size_t, _P],
    "s2p_conv2d_mat_is_fused": [_DESC, c_int, c_int],
    "s2p_conv2d_dgrad_workspace": [_DESC],
    "s2p_conv2d_dgrad_ws": [_DESC, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P, ctypes.c_size_t, _P],
    "s2p_conv2d_wgrad": [_DESC, _P, _P, _P, _P, c_int, c_int, c_int64, c_int, _P],
    "s2p_conv2d_wgrad_workspace": [_DESC, c_int, c_int],
    "s2p_conv2d_wgrad_ws": [_DESC, _P, _P, _P, _P, c_int, c_int, c_int64, c_int, _P, ctypes.c_size_t, _P],
    "s2p_conv2d_wgrad_batched_workspace": [_DESC, c_int, c_int, c_int],
    "s2p_conv2d_wgrad_batched": [_DESC, ctypes.POINTER(WgradJob), c_int, c_int, c_int, _P, ctypes.c_size_t, _P],
    "s2p_reflect_pad_bwd": [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P],
    "s2p_channel_sum": [c_int, _P, c_int64, c_int, c_int, _P, _P],
    "s2p_in_stats": [c_int, _P, c_int, c_int, c_int, c_int, c_float, _P, _P],
    "s2p_in_apply_fwd": [c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, c_int, c_float,
                         c_float, _P, c_int, _P],
    "s2p_in_norm_fwd": [c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, c_float, c_float, _P, c_int,
                        _P, _P],
    "s2p_in_bwd_reduce": [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, c_int,
                          c_float, c_float, _P, _P],
    "s2p_in_bwd_apply": [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, c_int,
                         c_float, c_float, _P, _P, c_int, _P, c_int, _P, c_int, _P],
    "s2p_in_norm_bwd": [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, c_int,
                        c_float, c_float, _P, _P, c_int, _P, c_int, _P, c_int, _P],
    "s2p_in_norm_bwd_res": [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, c_int,
                            c_float, c_float, _P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P],
    "s2p_in_stats_floats": [c_int, c_int, c_int],
    "s2p_in_bwd_sums_floats": [c_int, c_int, c_int],
    "s2p_linear_fwd": [_P, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, c_float, _P, c_int, c_int, _P],
    "s2p_linear_bwd_workspace": [c_int, c_int, c_int],
    "s2p_linear_bwd": [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_float, _P, c_int, _P, _P,
                       c_int, _P, ctypes.c_size_t, _P],
    "s2p_posenc_fwd": [_P, c_int, c_int, c_int, _P, c_int, _P],
    "s2p_avgpool3x3s2_fwd": [c_int, _P, c_int, c_int, c_int, c_int, _P, _P],
    "s2p_avgpool3x3s2_bwd": [c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "s2p_maxpool2x2_fwd": [c_int, _P, c_int, c_int, c_int, c_int, _P, _P],
    "s2p_maxpool2x2_bwd": [c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P],
    "s2p_resize_nearest": [c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P],
    "s2p_nchw_to_nhwc": [c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P],
    "s2p_nhwc_to_nchw": [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "s2p_cast": [c_int, _P, c_int, _P, c_int64, _P],
    "s2p_u8_to_nhwc": [c_int, _P, c_int64, c_int, _P, c_int, _P],
    "s2p_nhwc_to_u8": [c_int, _P, c_int, c_int64, c_int, _P, _P],
    "s2p_l1_loss": [c_int, _P, _P, c_int64, c_float, _P, _P, c_int, _P],
    "s2p_l1_loss_multi": [c_int, ctypes.POINTER(L1Job), c_int, _P],
    "s2p_hinge_loss": [c_int, _P, c_int64, c_int, c_float, _P, _P, _P],
    "s2p_hinge_loss_strided": [c_int, _P, c_int64, c_int, c_int, c_float, _P, _P, _P],
    "s2p_adam_step": [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int, c_float, _P],
    "s2p_ensemble_head": [_P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P,
                          _P, _P, _P],
    "s2p_adam_step_dev": [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, _P, c_float, _P],
    "s2p_adam_step_dev_part": [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, _P, c_float, c_int, _P],
    "s2p_pack_weights": [_P, c_int, c_int, _P],
    "s2p_act_bwd": [c_int, _P, _P, c_int64, c_int, c_float, _P, _P],
    "s2p_scale": [c_int, _P, c_int64, _P, _P],
    "s2p_add": [c_int, _P, _P, _P, c_int64, _P],
    "s2p_copy_channels": [c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int64, c_int, _P],
    "s2p_image_metrics": [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P],
}
_RESTYPE = {"s2p_last_error": ctypes.c_char_p, "s2p_conv2d_wgrad_batched_workspace": ctypes.c_size_t, "s2p_conv2d_wgrad_workspace": ctypes.c_size_t, "s2p_conv2d_fwd_workspace": ctypes.c_size_t, "s2p_conv2d_dgrad_workspace": ctypes.c_size_t, "s2p_linear_bwd_workspace": ctypes.c_size_t, "s2p_in_stats_floats": c_int64, "s2p_in_bwd_sums_floats": c_int64}

_lib = None


def build(force=False):
    """Compile libs2p_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    env = dict(os.environ)
    if force:
        env["FORCE"] = "1"                 # recompile every .hip even when the shipped .so is newer than the sources
    subprocess.check_call(["bash", script], env=env)
    return os.path.join(_HERE, "csrc", "libs2p_hip.so")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                f"libs2p_hip.so not found at {_SO}: the S2P hot path has no CPU fallback; run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc) first")
        L = ctypes.CDLL(_SO)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the export is missing
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, c_int)
        if L.s2p_version() < 121:
            raise RuntimeError("libs2p_hip.so is older than this package")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().s2p_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def dtype_id(t):
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t}")


def chunk_elems(t):
    return 4 if t == torch.float32 else 8


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("S2P HIP op called with a non-HIP tensor: the product path has no CPU fallback")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
