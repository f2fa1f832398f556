"""Loader + ctypes prototypes for libstm_hip.so (include/stm_hip.h).

There is deliberately no CPU fallback: if the HIP library is missing, importing a stage raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libstm_hip.so")
# STM_LIB=timing: the tools/*_time.py experiments load the -DSTM_TIMING build of the same sources (kernels with parts
# switched off; results NOT valid).  Nothing else ever loads it.
if os.environ.get("STM_LIB") == "timing":
    LIB_PATH = os.path.join(HERE, "libstm_hip_timing.so")

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
u8pp = C.POINTER(u8p)
f32pp = C.POINTER(f32p)
vp = C.c_void_p
i, f = C.c_int, C.c_float

# name -> argtypes, in the order of include/stm_hip.h.  Device-flavour entries take raw addresses (c_void_p).
PROTOS = {
    "stm_version": ([], i),
    "stm_set_stream": ([vp], None),
    "stm_get_stream": ([], vp),
    "stm_set_error_mode": ([i], None),
    "stm_last_error": ([], C.c_char_p),
    "stm_release_workspace": ([], None),
    "stm_prof_enable": ([i], None),
    "stm_prof_reset": ([], None),
    "stm_prof_read": ([C.c_char_p, f32p], i),
    "stm_set_agg_variant": ([i], None),
    "stm_set_irv_paper_ratio": ([i], None),
    "stm_set_ref_quirks": ([i], None),
    "stm_ci_adcensus": ([u8p, u8p, f32pp, f32pp, f, f, i, i, i, i, i], None),
    "stm_d_ci_adcensus": ([vp, vp, vp, vp, f32pp, f32pp, vp, f, f, i, i, i, i, i], None),
    "stm_ca_cross": ([u8p, u8pp, f32pp, f32pp, f, f, i, i, i, i, i, i], None),
    "stm_d_ca_cross": ([vp, vp, vp, f32pp, vp, vp, f, f, i, i, i, i, i, i], None),
    "stm_dc_wta": ([f32pp, f32p, i, i, i, i], None),
    "stm_d_dc_wta": ([vp, vp, i, i, i, i], None),
    "stm_dc_hslo": ([f32pp, f32p, u8p, u8p, f, f, f, i, i, i, i, i], None),
    "stm_d_dc_hslo": ([vp, vp, vp, vp, f, f, f, i, i, i, i, i], None),
    "stm_dr_dcc": ([u8p, u8p, f32p, f32p, i, i], None),
    "stm_d_dr_dcc": ([vp, vp, vp, vp, i, i], None),
    "stm_dr_irv": ([f32p, u8p, u8pp, i, f, i, i, i, i, i, i], None),
    "stm_d_dr_irv": ([vp, vp, vp, i, f, i, i, i, i, i, i], None),
    "stm_filter_bilateral_1": ([f32p, i, f, f, i, i, i], None),
    "stm_d_filter_bilateral_1": ([vp, i, f, f, i, i, i], None),
    "stm_filter_gaussian_1": ([f32p, i, f, i, i], None),
    "stm_d_filter_gaussian_1": ([vp, i, f, i, i], None),
    "stm_filter_bleed_1": ([u8p, i, i, i], None),
    "stm_d_filter_bleed_1": ([vp, i, i, i], None),
    "stm_filter_median": ([f32p, i, i], None),
    "stm_generate_gaussian_kernel": ([f32p, i, f], None),
    "stm_d_filter_median": ([vp, i, i], None),
    "stm_dibr_occl": ([u8p, u8p, f32p, f32p, i, i], None),
    "stm_d_dibr_occl": ([vp, vp, vp, vp, i, i], None),
    "stm_dibr_occl_to_mask": ([f32p, f32p, u8p, u8p, i, i], None),
    "stm_d_dibr_occl_to_mask": ([vp, vp, vp, vp, i, i], None),
    "stm_dibr_dbm": ([u8p, u8p, u8p, f32p, f32p, u8p, u8p, f32p, f32p, f, i, i, i], None),
    "stm_d_dibr_dbm": ([vp, vp, vp, vp, vp, vp, vp, vp, vp, f, i, i, i], None),
    "stm_dibr_dfm": ([u8p, u8p, u8p, f32p, f32p, f, i, i, i], None),
    "stm_d_dibr_dfm": ([vp, vp, vp, vp, vp, f, i, i, i], None),
    "stm_mux_multiview": ([u8pp, u8p, i, f, i, i, i, i, i], None),
    "stm_d_mux_multiview": ([vp, vp, i, f, i, i, i, i, i], None),
    "stm_d_demux_sbs": ([vp, vp, vp, i, i, i, i], None),
    "stm_adcensus_stm": ([u8p, f32p, f32p, u8p, i, i, i, i, i, i, i, f, i, i, f, f, f, f, i, i, i, f], None),
    "stm_d_adcensus_stm": ([vp, vp, vp, vp, i, i, i, i, i, i, i, f, i, i, f, f, f, f, i, i, i, f, i], None),
    "stm_adcensus_stm_2": ([u8p, f32p, f32p, u8p, i, i, i, i, i, i, i, i, f, i, f, i, i, f, f, f, f, i, i, i, f], None),
    "stm_d_adcensus_stm_2": ([vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, i, f, i, i, f, f, f, f, i, i, i, f], None),
    "stm_d_tx_scale": ([u8p, u8p, i, i, i, i, i], None),
    "stm_stream_create": ([i, i, i, i, i, i, i, f, i, i, f, f, f, f, i, i, i, f], C.c_void_p),
    "stm_stream_submit": ([C.c_void_p, u8p], C.c_long),
    "stm_stream_collect": ([C.c_void_p, f32p, f32p, u8p], C.c_long),
    "stm_stream_input_buffer": ([C.c_void_p], C.c_void_p),
    "stm_stream_collect_view": ([C.c_void_p, C.POINTER(f32p), C.POINTER(f32p), C.POINTER(u8p)], C.c_long),
    "stm_stream_destroy": ([C.c_void_p], None),
    "stm_bmp_read": ([C.c_char_p, C.POINTER(i), C.POINTER(i)], C.c_void_p),
    "stm_bmp_write": ([C.c_char_p, u8p, i, i], i),
    "stm_bmp_free": ([C.c_void_p], None),
}

_lib = None


def lib():
    """The loaded C-ABI library.  Raises (no fallback) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libstm_hip.so is missing (%s): build it with __graft_entry__.build() / make -C csrc; "
                "there is no CPU fallback for the product path" % LIB_PATH)
        # torch bundles its own libamdhip64.so.7; it must be in the process BEFORE this library is loaded so that
        # both share ONE HIP runtime (same device context, torch streams valid for stm_set_stream).  Loading in the
        # other order starts two runtimes and the second one finds "no ROCm-capable device".
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (args, res) in PROTOS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = res
        _lib = l
    return _lib
