"""ctypes binding of libvae_tagger_hip.so (C ABI in include/vae_tagger_hip.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VAE_TAGGER_HIP_LIB: an alternative build of the same library, for A/B runs of the test suite)
LIB_PATH = os.environ.get("VAE_TAGGER_HIP_LIB") or os.path.join(_HERE, "csrc", "libvae_tagger_hip.so")

VT_F32, VT_BF16, VT_F16 = 0, 1, 2
VT_STATUS_NONFINITE, VT_STATUS_FP8_SATURATED = 1, 2      # bits of vt_status (include/vae_tagger_hip.h)
ENCODE_MOMENTS, ENCODE_MODE, ENCODE_MODE_SCALED = 0, 1, 2

_c = ctypes
_vp, _i, _f, _sz, _ll = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t, _c.c_longlong

# name -> (restype, argtypes); every symbol include/vae_tagger_hip.h declares
PROTOTYPES = {
    "vt_version": (_c.c_char_p, []),
    "vt_create": (_i, [_i, _c.POINTER(_vp)]),
    "vt_destroy": (None, [_vp]),
    "vt_last_error": (_c.c_char_p, [_vp]),
    "vt_encoder_configure": (_i, [_vp, _i, _i, _c.POINTER(_i), _i, _i, _i, _f, _i, _f, _i]),
    "vt_set_weight": (_i, [_vp, _c.c_char_p, _vp, _i, _c.POINTER(_c.c_int64), _i]),
    "vt_encoder_finalize": (_i, [_vp]),
    "vt_decoder_configure": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i]),
    "vt_decoder_finalize": (_i, [_vp]),
    "vt_encode_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "vt_encode": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "vt_decode_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "vt_decode_logits": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "vt_get_confidence": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "vt_summarize_confidence": (_i, [_vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "vt_status": (_i, [_vp, _i, _c.POINTER(_i), _vp]),
    "vt_status_async": (_i, [_vp, _i, _vp, _vp]),
    "vt_encode_tag_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "vt_encode_tag": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "vt_preprocess_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "vt_resize_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "vt_resize_table": (_i, [_i, _i, _i, _c.POINTER(_i), _i]),
    "vt_resize_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "vt_encoder_flops": (_c.c_double, [_vp, _i, _i]),
    "vt_set_flag": (_i, [_vp, _i, _i]),
    "vt_debug_trace": (_i, [_vp, _i, _c.POINTER(_c.c_ulonglong), _i, _c.POINTER(_i)]),
    "vt_profile_num_configs": (_i, []),
    "vt_profile_begin": (_i, [_vp]),
    "vt_profile_end": (_i, [_vp, _i, _c.POINTER(_ll), _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _c.POINTER(_c.c_char_p)]),
    "vt_op_conv2d": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vt_op_norm_silu_conv3x3": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vt_op_conv2d_gn_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "vt_op_conv2d_gn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "vt_op_conv3x3_fp8_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "vt_op_conv3x3_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vt_op_gemm_nt": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ll, _ll, _ll, _f, _i, _vp]),
    "vt_op_conv_in": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "vt_op_groupnorm_workspace_bytes": (_sz, [_i, _i, _i]),
    "vt_op_groupnorm": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _vp]),
    "vt_op_softmax_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vt_op_attention_workspace_bytes": (_sz, [_i, _i, _i]),
    "vt_op_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()


class VTError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every prototype.  Raises if it is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise VTError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C {os.path.dirname(LIB_PATH)}`. "
                "There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)     # AttributeError here == header/library drift; let it surface
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


class Context:
    """One vt_context per device.  Owns packed weights on the device; buffers are the caller's."""

    def __init__(self, device_index=0):
        self.lib = load()
        h = _vp()
        rc = self.lib.vt_create(int(device_index), ctypes.byref(h))
        if rc != 0 or not h.value:
            raise VTError(f"vt_create(device={device_index}) failed with code {rc} (no usable HIP device?)")
        self.handle = h
        self.device_index = int(device_index)

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.vt_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.vt_last_error(self.handle)
            raise VTError(f"{what} failed (code {rc}): {msg.decode('utf-8', 'replace') if msg else ''}")

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self.handle, *args), name)

    def status(self, clear=True, stream=None):
        """Sticky device health word (synchronises the stream): bit 0 (VT_STATUS_NONFINITE) = non-finite GroupNorm statistics were
        seen; bit 1 (VT_STATUS_FP8_SATURATED, fp8 mode) = an activation exceeded the e4m3 range and was clamped."""
        v = _i(0)
        if stream is None:
            # torch's CURRENT stream of this device, so that the read is ordered after the work it is meant to check
            import torch
            stream = _vp(torch.cuda.current_stream(self.device_index).cuda_stream)
        self.call("vt_status", int(clear), ctypes.byref(v), stream)
        return v.value

    def set_weight(self, name, tensor):
        """tensor: torch tensor on any device; copied to host fp32/bf16/f16 and handed over by pointer."""
        import torch
        t = tensor.detach().to("cpu")
        if t.dtype == torch.bfloat16:
            dt, t = VT_BF16, t.contiguous().view(torch.int16)
        elif t.dtype == torch.float16:
            dt, t = VT_F16, t.contiguous().view(torch.int16)
        else:
            dt, t = VT_F32, t.to(torch.float32).contiguous()
        shape = (ctypes.c_int64 * max(1, tensor.dim()))(*tensor.shape)
        self.call("vt_set_weight", name.encode("utf-8"), _vp(t.data_ptr()), dt, shape, tensor.dim())
