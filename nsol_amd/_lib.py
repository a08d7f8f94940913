"""ctypes binding of libnsol_hip.so (the C ABI declared in include/nsol_hip.h).

There is NO CPU fallback: if the shared library is missing or does not export a
symbol, every compute entry point raises.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libnsol_hip.so")
HEADER = os.path.join(HERE, "..", "include", "nsol_hip.h")

_lib = None

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_dbl = ctypes.c_double
c_ptr = ctypes.c_void_p

_CTYPE = {"int": c_int, "int64_t": c_i64, "double": c_dbl}


class NsolHipError(RuntimeError):
    pass


def declared_symbols(header=HEADER):
    """Parse include/nsol_hip.h: {name: (restype, [argtypes])}."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\bint\s+(nsol_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(c_ptr)
                else:
                    argtypes.append(_CTYPE[a.split()[-2]])
        out[name] = (c_int, argtypes)
    out.pop("nsol_hip_set_param", None)
    return out


def load():
    """Load the library once and set argtypes from the header."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NsolHipError(
            "libnsol_hip.so not found at %s -- build it with "
            "`python -m nsol_amd.build` (there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, argtypes) in declared_symbols().items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise NsolHipError("libnsol_hip.so does not export %s" % name)
        fn.restype = res
        fn.argtypes = argtypes
    # experiment knob, not part of the reference-facing ABI
    lib.nsol_hip_set_param.restype = c_int
    lib.nsol_hip_set_param.argtypes = [ctypes.c_char_p, c_int]
    lib.nsol_lb_sort_tmp_bytes.restype = c_i64
    lib.nsol_lb_sort_tmp_bytes.argtypes = [c_int, c_int]
    lib.nsol_lb_walk_table_doubles.restype = c_i64
    lib.nsol_lb_walk_table_doubles.argtypes = [c_int, c_int]
    lib.nsol_lb_walk_tmp_bytes.restype = c_i64
    lib.nsol_lb_walk_tmp_bytes.argtypes = [c_int]
    lib.nsol_lb_gram_ws_doubles.restype = c_i64
    lib.nsol_lb_gram_ws_doubles.argtypes = []
    lib.nsol_pd_persist_ws_bytes.restype = c_i64
    lib.nsol_pd_persist_ws_bytes.argtypes = [c_int, c_int, c_i64, c_i64, c_i64,
                                             c_int]
    lib.nsol_hip_set_param_conv.restype = c_int
    lib.nsol_hip_set_param_conv.argtypes = [ctypes.c_char_p, c_int]
    lib.nsol_hip_set_param_pd2.restype = c_int
    lib.nsol_hip_set_param_pd2.argtypes = [ctypes.c_char_p, c_int]
    lib.nsol_hip_set_param_pdk.restype = c_int
    lib.nsol_hip_set_param_pdk.argtypes = [ctypes.c_char_p, c_int]
    lib.nsol_hip_set_param_lb.restype = c_int
    lib.nsol_hip_set_param_lb.argtypes = [ctypes.c_char_p, c_int]
    lib.nsol_hip_set_param_sort.restype = c_int
    lib.nsol_hip_set_param_sort.argtypes = [ctypes.c_char_p, c_int]
    lib.nsol_hip_set_param_pdp.restype = c_int
    lib.nsol_hip_set_param_pdp.argtypes = [ctypes.c_char_p, c_int]
    if lib.nsol_hip_abi_version() != 1:
        raise NsolHipError("libnsol_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        if rc == -1:
            raise ValueError("%s: invalid argument" % what)
        raise NsolHipError("%s failed with hipError_t %d" % (what, rc))


# every experiment knob of the library with the default it is built with
# (nsol_pd.hip PdTuning, nsol_pd2.hip / nsol_pdk.hip Tuning, nsol_conv.hip globals)
PARAM_DEFAULTS = {
    "pd_zchunk": 0, "pd_ry": 0, "pd_two_pass": 0, "pd_xcd_map": 1, "pd_rag": 1,
    "max_grid_blocks": 2048, "stencil_slabs": 1, "stencil_blocks": 65536,
    "pd2_zchunk": 0, "pd2_enable": 1, "pd2_variant": 0, "pd2_xcd_map": 1,
    "pdk_enable": 1, "pdk_kmax": 3, "pdk_nw": 0, "pdk_zchunk": 0, "pdk_ntx": 0,
    "pdk_xcd_map": 1, "pdk_verbose": 0, "pdk_autotune": 1, "pdk_pf2": -1,
    "pdk_split": -1, "pdk_tail2": 1, "pdk_min_kvox": 1024, "pdk_tune_min_mvox": 16,
    "pdp_max_spin": 1 << 21, "pdp_mute_tile": -1,
    "corr_ra": 8, "corr_xv": 1, "corr_blur3_lxb": 16, "corr_blur3_zchunk": 0, "corr_blur3_dma": 1, "corr_blur3_dma_rag": 1, "lb_gram_dma": 1, "lb_gram_mfma": 1, "sort_walk_by_key": 1,
}
_touched = set()


def reset_params():
    """Put every knob a caller changed back to its built-in default."""
    for name in sorted(_touched):
        set_param(name, PARAM_DEFAULTS[name])
    _touched.clear()


def set_param(name, value):
    lib = load()
    if name in PARAM_DEFAULTS:
        _touched.add(name)
    if name.startswith("pd2_"):
        fn = lib.nsol_hip_set_param_pd2
    elif name.startswith("pdk_"):
        fn = lib.nsol_hip_set_param_pdk
    elif name.startswith("pdp_"):
        fn = lib.nsol_hip_set_param_pdp
    elif name.startswith("corr_"):
        fn = lib.nsol_hip_set_param_conv
    elif name.startswith("lb_"):
        fn = lib.nsol_hip_set_param_lb
    elif name.startswith("sort_"):
        fn = lib.nsol_hip_set_param_sort
    else:
        fn = lib.nsol_hip_set_param
    check(fn(name.encode(), int(value)), "set_param")
