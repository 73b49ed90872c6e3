"""ctypes loader for libbcnn_hip.so -- the C-ABI declared in include/bcnn_hip.h.

There is NO fallback: if the HIP library is missing or a symbol is absent, importing/using the
back-end raises. (The CPU restatement under oracle/ is test infrastructure and is never loaded here.)
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BCNN_HIP_LIB", os.path.join(_HERE, "lib", "libbcnn_hip.so"))  # override: kernel experiments
CSRC = os.path.join(_HERE, "csrc")

vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, [argtypes]); mirrors include/bcnn_hip.h one to one
SIGNATURES = {
    "bcnn_hip_device_count": (i, []),
    "bcnn_hip_set_device": (None, [i]),
    "bcnn_hip_get_device": (i, []),
    "bcnn_hip_device_name": (C.c_char_p, []),
    "bcnn_hip_malloc_f32": (vp, [sz]),
    "bcnn_hip_malloc_i32": (vp, [sz]),
    "bcnn_hip_free": (None, [vp]),
    "bcnn_hip_memcpy_h2d": (None, [vp, vp, sz]),
    "bcnn_hip_memcpy_d2h": (None, [vp, vp, sz]),
    "bcnn_hip_memcpy_d2d": (None, [vp, vp, sz]),
    "bcnn_hip_fill_f32": (None, [vp, sz, f]),
    "bcnn_hip_sync": (None, []),
    "bcnn_hip_stream_create": (vp, []),
    "bcnn_hip_stream_destroy": (None, [vp]),
    "bcnn_hip_set_stream": (None, [vp]),
    "bcnn_hip_get_stream": (vp, []),
    "bcnn_hip_event_create": (vp, []),
    "bcnn_hip_event_destroy": (None, [vp]),
    "bcnn_hip_event_record": (None, [vp]),
    "bcnn_hip_event_sync": (None, [vp]),
    "bcnn_hip_event_elapsed_ms": (f, [vp, vp]),
    "bcnn_hip_profile_enable": (None, [i]),
    "bcnn_hip_profile_reset": (None, []),
    "bcnn_hip_profile_num_classes": (i, []),
    "bcnn_hip_profile_class_name": (C.c_char_p, [i]),
    "bcnn_hip_profile_read": (None, [i, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "bcnn_hip_profile_read_useful_flops": (C.c_double, [i]),
    "bcnn_hip_conv_side_stream_mode": (i, [i]),
    "bcnn_hip_conv_side_join": (None, []),
    "bcnn_hip_trace_enable": (None, [i]),
    "bcnn_hip_trace_read": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "bcnn_hip_axpy": (None, [sz, f, vp, vp]),
    "bcnn_hip_scal": (None, [sz, f, vp]),
    "bcnn_hip_copy_f32": (None, [sz, vp, vp]),
    "bcnn_hip_add_bias": (None, [vp, vp, i, i, i]),
    "bcnn_hip_grad_bias": (None, [vp, vp, i, i, i]),
    "bcnn_hip_scales": (None, [vp, vp, i, i, i]),
    "bcnn_hip_grad_scales": (None, [vp, vp, i, i, i, vp]),
    "bcnn_hip_gemm": (None, [i, i, i, i, i, f, vp, i, vp, i, f, vp, i]),
    "bcnn_hip_im2col": (None, [vp, i, i, i, i, i, i, vp]),
    "bcnn_hip_col2im": (None, [vp, i, i, i, i, i, i, vp]),
    "bcnn_hip_activation_forward": (None, [vp, sz, i, vp, i, i]),
    "bcnn_hip_activation_backward": (None, [vp, vp, sz, i, vp, vp, i, i]),
    "bcnn_hip_batchnorm_forward": (None, [vp] * 10 + [i, i, i, i, i]),
    "bcnn_hip_batchnorm_backward": (None, [vp, vp, vp, i] + [vp] * 9 + [i, i, i]),
    "bcnn_hip_conv_workspace_size": (sz, [i] * 9),
    "bcnn_hip_conv_forward": (None, [vp, vp, vp, vp] + [i] * 10 + [vp, i, vp, vp, vp, vp, vp, vp, vp, i]),
    "bcnn_hip_conv_backward": (None, [vp] * 8 + [i] * 10 + [vp, vp, i] + [vp] * 8 + [vp, sz]),
    "bcnn_hip_maxpool_forward": (None, [vp, vp, vp] + [i] * 8),
    "bcnn_hip_maxpool_backward": (None, [vp, vp, vp] + [i] * 9),
    "bcnn_hip_avgpool_forward": (None, [vp, vp, i, i, i, i]),
    "bcnn_hip_avgpool_backward": (None, [vp, vp, i, i, i, i]),
    "bcnn_hip_depthwise_forward": (None, [vp, vp, vp, vp] + [i] * 8),
    "bcnn_hip_depthwise_backward": (None, [vp] * 7 + [i] * 9),
    "bcnn_hip_depthwise_stats_size": (sz, [i] * 7),
    "bcnn_hip_depthwise_forward_stats": (i, [vp, vp, vp, vp] + [i] * 8 + [vp, sz]),
    "bcnn_hip_batchnorm_forward_stats": (None, [vp] * 10 + [i, i, i, i, i, vp, i]),
    "bcnn_hip_batchnorm_forward_stats_only": (None, [vp] * 7 + [i, i, i, vp, i]),
    "bcnn_hip_conv_bnfold_fusable": (i, [i] * 5), "bcnn_hip_conv_set_input_bnfold": (None, [vp] * 4),
    "bcnn_hip_depthwise_bn_fusable": (i, [i] * 8),
    "bcnn_hip_batchnorm_backward_sums": (None, [vp] * 9 + [i, i, i]),
    "bcnn_hip_depthwise_backward_bn": (None, [vp] * 7 + [i] * 9 + [vp] * 5),
    "bcnn_hip_batchnorm_backward_apply": (None, [vp] * 8 + [i, i, i]),
    "bcnn_hip_depthwise_bnin_fusable": (i, [i] * 9),
    "bcnn_hip_conv_bnsums_size": (sz, [i] * 4),
    "bcnn_hip_conv_backward_bnsums": (i, [vp] * 8 + [i] * 10 + [vp, vp, i] + [vp] * 8 + [vp, sz] + [vp, vp, vp, sz]),
    "bcnn_hip_batchnorm_backward_finalize": (None, [vp, i] + [vp] * 6 + [i]),
    "bcnn_hip_depthwise_forward_bnin": (i, [vp, vp, vp, vp] + [i] * 8 + [vp, sz] + [vp] * 4 + [i]),
    "bcnn_hip_depthwise_backward_bnin": (None, [vp] * 7 + [i] * 9 + [vp] * 4 + [i]),
    "bcnn_hip_depthwise_backward_bn_bnin": (None, [vp] * 7 + [i] * 9 + [vp] * 5 + [vp] * 4 + [i]),
    "bcnn_hip_conv_residual_fusable": (i, [i, i, i, i, vp, vp, vp]),
    "bcnn_hip_conv_forward_residual": (None, [vp, vp, vp] + [i] * 9 + [vp] * 7 + [sz, i, vp]),
    "bcnn_hip_conv_backward_residual": (None, [vp] * 7 + [i] * 9 + [vp] * 7 + [vp, sz, vp, vp, i, vp, vp, sz]),
    "bcnn_hip_batchnorm_apply": (None, [vp] * 6 + [i, i, i, i]),
    "bcnn_hip_conv_prepack": (None, [vp, i, i]),
    "bcnn_hip_maxpool_forward_bn_keep": (None, [vp, vp, vp] + [i] * 8 + [vp] * 4 + [i, vp]),
    "bcnn_hip_maxpool_bn_backward_fusable": (i, [i] * 9 + [vp] * 4),
    "bcnn_hip_maxpool_bn_backward": (None, [vp] * 5 + [i] * 8 + [vp] * 8 + [i]),
    "bcnn_hip_conv_backward_bn_done": (None, [vp] * 5 + [i] * 9 + [vp, sz]),
    "bcnn_hip_depthwise_insums_size": (sz, [i] * 7),
    "bcnn_hip_depthwise_backward_bnin_sums": (i, [vp] * 7 + [i] * 9 + [vp] * 5 + [vp] * 4 + [i] + [vp, sz]),
    "bcnn_hip_conv_backward_presummed": (i, [vp] * 8 + [i] * 10 + [vp, vp, i] + [vp] * 8 + [vp, sz] + [vp, i] + [vp, vp, vp, sz]),
    "bcnn_hip_conv_prepack_reset": (None, []),
    "bcnn_hip_conv_prepack_discard": (None, []),
    "bcnn_hip_maxpool_bn_fusable": (i, [i] * 9 + [vp]),
    "bcnn_hip_conv_forward_stats_only": (None, [vp, vp, vp] + [i] * 9 + [vp] * 6),
    "bcnn_hip_maxpool_forward_bn": (None, [vp, vp, vp] + [i] * 8 + [vp] * 4 + [i]),
    "bcnn_hip_sgd_update": (None, [vp, vp, vp, vp, sz, sz, i, f, f, f]),
    "bcnn_hip_sgd_update_chunks": (None, [vp, i, i, f, f, f]),
    "bcnn_hip_zero_chunks": (None, [vp, i]),
    "bcnn_hip_adam_update": (None, [vp, vp, vp, vp, vp, vp, sz, sz, i, i, f, f, f, f, f]),
    "bcnn_hip_eltwise_forward": (None, [vp, vp, vp, sz, sz, i]),
    "bcnn_hip_eltwise_backward": (None, [vp, vp, vp, vp, sz, sz, i, i]),
    "bcnn_hip_cost_metric": (None, [i, vp, vp, vp, i, i, vp]),
    "bcnn_hip_axpy_strided": (None, [i, f, vp, vp] + [i] * 11),
    "bcnn_hip_add_rowvec": (None, [vp, vp, i, i]),
    "bcnn_hip_softmax_forward": (None, [vp, vp, i, i, i]),
    "bcnn_hip_comm_init": (None, [i, i, C.c_char_p]),
    "bcnn_hip_comm_destroy": (None, []),
    "bcnn_hip_comm_retain": (None, []),
    "bcnn_hip_rendezvous_publish": (i, [C.c_char_p, vp, sz, i]),
    "bcnn_hip_rendezvous_fetch": (i, [C.c_char_p, vp, sz, i, i]),
    "bcnn_hip_comm_world": (i, []),
    "bcnn_hip_comm_rank": (i, []),
    "bcnn_hip_allreduce_sum": (None, [vp, sz]),
    "bcnn_hip_broadcast": (None, [vp, sz, i]),
    "bcnn_hip_comm_join": (None, []),
}

_lib = None


def build(verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU) into lib/libbcnn_hip.so."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=out)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch wheels bundle their own HIP runtime; importing torch first makes the process share ONE
        # runtime (two runtimes in a process cannot both own the device). torch stays plumbing only.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "bcnn_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def declared_symbols():
    """Function names declared in include/bcnn_hip.h (parsed from the header text)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "bcnn_hip.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bcnn_hip_[a-z0-9_]+)\s*\(", text)))
