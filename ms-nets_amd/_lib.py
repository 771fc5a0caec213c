"""ctypes binding of libmsnet_hip.so (include/msnet_hip.h).  There is no CPU fallback: if the library is
missing or a call fails, a RuntimeError is raised -- the product path never silently degrades."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSNET_HIP_LIB lets an A/B experiment point at another build of the same ABI (tools_layer_bench.py)
LIB_PATH = os.environ.get("MSNET_HIP_LIB") or os.path.join(_HERE, "libmsnet_hip.so")
_lib = None


class VolumeParams(ctypes.Structure):
    """msnet_volume_params (include/msnet_hip.h); defaults = cbmv_generator.py:434-462 of the reference."""
    _fields_ = [("censw", c_int), ("nccw", c_int), ("sadw", c_int), ("sobelw", c_int),
                ("cens_sigma", c_float), ("ncc_sigma", c_float), ("sad_sigma", c_float),
                ("border_h", c_int), ("border_w", c_int)]


P = c_void_p
# name -> (restype, argtypes); must list every symbol include/msnet_hip.h declares (tests/test_abi.py checks).
SIGNATURES = {
    "msnet_version": (c_int, []),
    "msnet_last_error": (c_char_p, []),
    "msnet_prof_enable": (c_int, [c_int]),
    "msnet_prof_select": (c_int, [c_char_p]),
    "msnet_prof_collect": (c_long, [c_char_p, c_size_t]),
    "msnet_set_overflow_flag": (c_int, [P]),
    "msnet_set_exact_tails": (c_int, [c_int]),
    "msnet_peak_copy": (c_int, [P, P, c_size_t, P]),
    "msnet_peak_mfma_f16": (ctypes.c_double, [P, c_int, P]),
    "msnet_peak_mfma_f16_16x16": (ctypes.c_double, [P, c_int, P]),
    "msnet_peak_mfma_f16_rand": (ctypes.c_double, [P, c_int, c_int, P]),
    "msnet_clock_probe": (c_int, [P, P]),
    "msnet_census": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "msnet_census_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "msnet_ncc": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "msnet_zsad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "msnet_sobel": (c_int, [P, P, c_int, c_int, P]),
    "msnet_sadsob": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "msnet_sadsob_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "msnet_swap_axes": (c_int, [P, P, c_int, c_int, c_int, P]),
    "msnet_get_right_cost": (c_int, [P, P, c_int, c_int, c_int, P]),
    "msnet_extract_likelihood": (c_int, [P, P, c_long, c_int, c_float, P]),
    "msnet_extract_features_left": (c_int, [P, P, P, P, P, c_long, c_int, c_float, c_float, c_float, P]),
    "msnet_volume_default_params": (None, [ctypes.POINTER(VolumeParams)]),
    "msnet_build_volume": (c_int, [P, P, c_int, c_int, c_int, ctypes.POINTER(VolumeParams), P, P, P]),
    "msnet_build_volume_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "msnet_build_volume_ndhwc_supported": (c_int, [c_int, c_int, c_int, ctypes.POINTER(VolumeParams)]),
    "msnet_build_volume_ndhwc": (c_int, [P, P, c_int, c_int, c_int, ctypes.POINTER(VolumeParams), P, P, P]),
    "msnet_preprocess_out_shape": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "msnet_preprocess_image": (c_int, [P, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_double), P, P, P]),
    "msnet_ncdhw_to_ndhwc": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_ndhwc_to_ncdhw": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_check_input_range": (c_int, [P, c_size_t, P]),
    "msnet_packed_weight_floats": (c_size_t, [c_int, c_int]),
    "msnet_pack_conv_weight": (c_int, [P, P, c_int, c_int, P]),
    "msnet_pack_deconv_weight": (c_int, [P, P, c_int, c_int, P]),
    "msnet_conv3d_k3": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_pack_conv_weight_f16s": (c_int, [P, P, c_int, c_int, c_int, P]),
    "msnet_conv3d_k3_f16s_supported": (c_int, [c_int, c_int, c_int]),
    "msnet_conv3d_k3_f16s": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_deconv3d_k3s2_f16s_supported": (c_int, [c_int, c_int]),
    "msnet_pack_deconv_weight_f16s": (c_int, [P, P, c_int, c_int, P]),
    "msnet_deconv3d_k3s2_f16s": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_deconv3d_k3s2": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_pack_conv_weight_wd_f16s": (c_int, [P, P, P]),
    "msnet_conv3d_k3_wd_f16s_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "msnet_conv3d_k3_wd_f16s": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_conv3d_k3_wd_f16s_strided": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_conv3d_k3_c8_ncdhw_f16s": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_conv3d_k3_c8_in_f16s": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_conv3d_k3_in_f16s": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_conv3d_k3_cout1": (c_int, [P, P, c_float, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_softargmin": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "msnet_deconv5_softargmin": (c_int, [P, P, c_float, c_float, P, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_deconv5_softargmin_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "msnet_deconv5_softargmin_ws": (c_int, [P, P, c_float, c_float, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "msnet_deconv3d_cout1": (c_int, [P, P, c_float, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_trilinear_softargmin": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "msnet_epe_badx": (c_int, [P, P, c_size_t, c_float, c_float, P, P]),
}


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmsnet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback for this path." % LIB_PATH)
    # torch FIRST: the library needs libamdhip64.so.7 and PyTorch-ROCm loads its own bundled copy of it by path.  With torch's copy
    # already in the process the loader hands that one to this library too (same SONAME) and both share ONE HIP runtime; loaded the
    # other way round the process ends up with two runtimes, and the second one to touch the device reports "no ROCm-capable
    # device is detected" on its first launch (round 6: __graft_entry__.build() followed by smoke() in one process did that).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    variant = bool(os.environ.get("MSNET_HIP_LIB"))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)      # AttributeError => ABI mismatch, fail loudly
        except AttributeError:
            # only an A/B library named through MSNET_HIP_LIB (an older build of the same ABI) may lack a newer entry point:
            # calling it raises; the shipped library must export every symbol (tests/test_abi.py)
            if not variant:
                raise
            continue
        fn.restype = res
        fn.argtypes = args
    if lib.msnet_version() != 1:
        raise RuntimeError("libmsnet_hip.so ABI version %d != 1" % lib.msnet_version())
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().msnet_last_error()
        raise RuntimeError("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))


def stream_ptr():
    """The current torch HIP stream as a void* for the C ABI."""
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def require_gpu_f32(t, name, dtype=None):
    import torch
    dtype = dtype or torch.float32
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on an MI355X device (got %s); this path has no CPU implementation"
                           % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if t.device.index != torch.cuda.current_device():
        # kernels launch on the CURRENT device's stream; a tensor of another GPU would be a foreign pointer there
        raise RuntimeError("%s lives on %s but the current device is cuda:%d (use torch.cuda.set_device / torch.cuda.device)"
                           % (name, t.device, torch.cuda.current_device()))
    return t.contiguous()


def prof_enable(on=True, prefix=None):
    """Per-launch HIP-event timing; `prefix` restricts it to kernel families whose name starts with it."""
    load().msnet_prof_select(prefix.encode() if prefix else None)
    load().msnet_prof_enable(1 if on else 0)


def prof_collect():
    """-> {kernel: dict(calls, ms, flops, bytes)} for launches since the last collect."""
    buf = ctypes.create_string_buffer(1 << 16)
    n = load().msnet_prof_collect(buf, len(buf))
    if n < 0:
        check(1, "msnet_prof_collect")
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms, fl, by = line.split()
        out[name] = dict(calls=int(calls), ms=float(ms), flops=float(fl), bytes=float(by))
    return out
