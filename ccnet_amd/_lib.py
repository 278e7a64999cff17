"""ctypes binding of libccnet_hip.so (include/ccnet_hip.h).  There is no CPU fallback: if the library is missing
or a call fails, a RuntimeError says so."""
import ctypes as C
import os

import torch

F32, BF16 = 0, 1
ACT = {"none": 0, None: 0, "silu": 1, "relu": 2, "relu6": 3, "lrelu": 4}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CCNET_HIP_LIB") or os.path.join(_HERE, "libccnet_hip.so")


class Slice(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ldc", C.c_int32), ("coff", C.c_int32), ("C", C.c_int32), ("shift", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("out_dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("nseg", C.c_int32),
                ("inp", Slice * 2), ("weight", C.c_void_p), ("Kpad", C.c_int32), ("Cout_pad", C.c_int32),
                ("bias", C.c_void_p), ("act", C.c_int32), ("residual", Slice), ("out", Slice)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("ccnet_amd: %s is missing -- build it with `make -C ccnet_amd/csrc` (or "
                               "__graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t
        L.ccnet_last_error.restype = C.c_char_p
        L.ccnet_version.restype = i32
        L.ccnet_device_count.restype = i32
        L.ccnet_conv_pack_dims.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        L.ccnet_conv2d.argtypes = [C.POINTER(ConvDesc), vp]
        L.ccnet_conv_halo_stream_elems.argtypes = [i32, i32]
        L.ccnet_conv_halo_stream_elems.restype = sz
        L.ccnet_conv_halo_s2_stream_elems.argtypes = [i32, i32]
        L.ccnet_conv_halo_s2_stream_elems.restype = sz
        L.ccnet_conv_halo.argtypes = [C.POINTER(Slice), i32, i32, i32, i32, i32, i32, vp, vp, i32, Slice, Slice, vp]
        L.ccnet_conv_pw_stream_elems.argtypes = [i32, i32]
        L.ccnet_conv_pw_stream_elems.restype = sz
        L.ccnet_conv_pw.argtypes = [C.POINTER(Slice), i32, i32, i32, i32, i32, vp, vp, i32, Slice, Slice, vp]
        L.ccnet_bneck_fused_stream_elems.argtypes = [i32]
        L.ccnet_bneck_fused_stream_elems.restype = sz
        L.ccnet_bneck_fused.argtypes = [Slice, i32, i32, i32, vp, vp, vp, i32, i32, Slice, vp]
        L.ccnet_csp_fused_weight_elems.argtypes = [i32, i32]
        L.ccnet_csp_fused_weight_elems.restype = sz
        L.ccnet_csp_fused.argtypes = [C.POINTER(Slice), i32, i32, i32, i32, i32, i32, i32, vp, vp, Slice, vp]
        L.ccnet_stem_conv.argtypes = [vp, i32, i32, i32, vp, vp, i32, i32, Slice, i32, vp]
        L.ccnet_dwconv3x3.argtypes = [Slice, i32, i32, i32, i32, vp, vp, Slice, vp]
        L.ccnet_sppf_pool.argtypes = [Slice, i32, i32, i32, i32, vp]
        L.ccnet_maxpool3x3s2.argtypes = [Slice, i32, i32, i32, i32, Slice, vp]
        L.ccnet_gate_scratch_floats.argtypes = [i32, i32, i32, i32]
        L.ccnet_gate_scratch_floats.restype = sz
        L.ccnet_cbam.argtypes = [Slice, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, Slice, vp]
        L.ccnet_eca.argtypes = [Slice, i32, i32, i32, i32, vp, vp, Slice, vp]
        L.ccnet_asff_fuse.argtypes = [Slice, Slice, Slice, Slice, i32, i32, i32, i32, Slice, vp]
        L.ccnet_layernorm.argtypes = [vp, vp, vp, vp, vp, i32, i64, i32, f32, vp]
        L.ccnet_attention.argtypes = [Slice, Slice, Slice, i32, i32, i32, i32, i32, i32, Slice, vp]
        L.ccnet_transformer_fused_frags.argtypes = [i32, i32, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.ccnet_transformer_fused.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, i32, vp, C.c_longlong, vp, i32, i32, vp]
        L.ccnet_decode.argtypes = [C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32),
                                   i32, i32, i32, f32, f32, vp, vp]
        L.ccnet_decode_scored.argtypes = [C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32),
                                          i32, i32, i32, f32, f32, vp, vp, vp]
        L.ccnet_filter_scored.argtypes = [vp, vp, i32, i32, i32, f32, vp, vp, vp, vp]
        L.ccnet_filter.argtypes = [vp, i32, i32, i32, f32, vp, vp, vp]
        L.ccnet_filter_workspace_ints.argtypes = [i32, i32]
        L.ccnet_filter_workspace_ints.restype = sz
        L.ccnet_filter_ws.argtypes = [vp, i32, i32, i32, f32, vp, vp, vp, vp]
        L.ccnet_nms_workspace_bytes.argtypes = [i32, i32]
        L.ccnet_nms_workspace_bytes.restype = sz
        L.ccnet_nms.argtypes = [vp, vp, i32, i32, i32, f64, vp, vp, vp, vp, vp, vp]
        L.ccnet_nms_noalloc.argtypes = [vp, vp, i32, i32, i32, f64, vp, vp, vp, vp, vp]
        L.ccnet_nms_noalloc_max_bytes.restype = sz
        L.ccnet_nms_noalloc_init.argtypes = []
        ll = C.c_longlong
        L.ccnet_resample_ksize.argtypes = [i32, i32]
        L.ccnet_resample_coeffs.argtypes = [i32, i32, vp, vp, i32]
        L.ccnet_preprocess_workspace_bytes.argtypes = [i32, i32, i32]
        L.ccnet_preprocess_workspace_bytes.restype = sz
        L.ccnet_preprocess.argtypes = [vp, i32, i32, i32, i32, ll, ll, i32, vp, i32, i32, ll, ll, i32, i32, i32, i32,
                                       vp, vp, i32, vp, vp, i32, vp, i32, vp, sz, vp]
        L.ccnet_plan_begin.argtypes = [C.POINTER(vp)]
        L.ccnet_plan_end.argtypes = []
        L.ccnet_plan_num_ops.argtypes = [vp]
        L.ccnet_plan_set_lane.argtypes = [i32]
        L.ccnet_plan_wait.argtypes = [i32, i32]
        L.ccnet_plan_run.argtypes = [vp, vp, vp]
        L.ccnet_plan_destroy.argtypes = [vp]
        L.ccnet_plan_op_info.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(f64), C.POINTER(f64)]
        L.ccnet_plan_run_timed.argtypes = [vp, vp, vp, C.POINTER(f32)]
        L.ccnet_plan_destroy.restype = None
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libccnet_hip: %s (code %d)" % (lib().ccnet_last_error().decode("utf-8", "replace"), rc))


def dtype_code(t):
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError("ccnet_amd supports float32 and bfloat16 activations, got %s" % t)


def torch_dtype(code):
    return torch.bfloat16 if code == BF16 else torch.float32


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def sl(t, coff=0, C_=None, shift=0):
    """Channel slice of an NHWC tensor (..., ldc)."""
    ldc = t.shape[-1]
    return Slice(t.data_ptr(), ldc, coff, ldc - coff if C_ is None else C_, shift)


def null_slice():
    return Slice(None, 0, 0, 0, 0)


def conv_pack_dims(dtype, cout, k):
    a, b = C.c_int32(), C.c_int32()
    check(lib().ccnet_conv_pack_dims(dtype, cout, k, C.byref(a), C.byref(b)))
    return a.value, b.value
