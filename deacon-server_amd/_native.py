"""ctypes binding of lib/libdeacon_hip.so (the C ABI declared in include/deacon_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing this module raises.
"""
import ctypes as C
import os
import re
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCN_LIB_PATH") or os.path.join(_PKG, "lib", "libdeacon_hip.so")  # env: experiment builds
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "deacon_hip.h")

DCN_OK = 0
DCN_ERR_ARG = -1
DCN_ERR_HIP = -2
DCN_ERR_NOMEM = -3
DCN_ERR_IO = -4
DCN_ERR_FORMAT = -5
DCN_ERR_CAPACITY = -6
DCN_ERR_INTERNAL = -7
N_STATS = 6
N_STAGES = 5
STAGE_NAMES = ("pack", "plan", "scan", "distinct", "finish")
STAT_NAMES = ("total_seqs", "filtered_seqs", "total_bp", "output_bp", "filtered_bp", "output_seq_counter")


class DeaconHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"deacon_hip error {code}: {message}")
        self.code = code
        self.message = message


class Params(C.Structure):
    _fields_ = [
        ("abs_threshold", C.c_uint64),
        ("rel_threshold", C.c_double),
        ("prefix_length", C.c_uint64),
        ("deplete", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


def build(force=False, jobs=6):
    """Compile every HIP source for gfx950 into lib/libdeacon_hip.so (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_PKG, "csrc")
    cmd = ["make", "-C", csrc, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def declared_symbols():
    """Names of every function declared in include/deacon_hip.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcn_[a-z0-9_]+)\s*\(", text)))


_u8p, _u32p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_vp = C.c_void_p

_SIGNATURES = {
    "dcn_abi_version": (C.c_int, [_u32p, _u32p]),
    "dcn_version": (C.c_char_p, []),
    "dcn_last_error": (C.c_char_p, []),
    "dcn_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dcn_set_minimizer_variant": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "dcn_get_minimizer_variant": (C.c_int, [_u32p, _u32p, _u32p]),
    "dcn_index_from_keys": (C.c_int, [_vp, C.c_uint64, C.c_uint8, C.c_uint8, C.c_int, C.POINTER(_vp)]),
    "dcn_index_from_file": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "dcn_index_build": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint8, C.c_uint8, C.c_float, C.c_uint64, C.c_int,
                                  C.POINTER(_vp)]),
    "dcn_index_keys": (C.c_int, [_vp, _vp, C.c_uint64, _u64p]),
    "dcn_index_union": (C.c_int, [C.POINTER(_vp), C.c_uint32, C.POINTER(_vp)]),
    "dcn_index_diff": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "dcn_index_write_file": (C.c_int, [_vp, C.c_char_p]),
    "dcn_index_header": (C.c_int, [_vp, _u8p, _u8p, _u64p]),
    "dcn_index_contains": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "dcn_index_contains_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "dcn_index_probe_ceiling": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    "dcn_index_destroy": (None, [_vp]),
    "dcn_ctx_create": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.POINTER(_vp)]),
    "dcn_ctx_destroy": (None, [_vp]),
    "dcn_filter_batch": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(Params), _vp, _vp, _vp]),
    "dcn_filter_batch_submit": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(Params), _vp, _vp, _vp, _u64p]),
    "dcn_filter_batch_wait": (C.c_int, [_vp, C.c_uint64]),
    "dcn_filter_batch_packed": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(Params), _vp, _vp, _vp]),
    "dcn_filter_batch_packed_submit": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(Params), _vp, _vp, _vp,
                                                 _u64p]),
    "dcn_pack_ascii": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _u32p]),
    "dcn_stats_allreduce": (C.c_int, [C.POINTER(_vp), C.c_int, _u64p]),
    "dcn_comm_available": (C.c_int, []),
    "dcn_comm_unique_id": (C.c_int, [_vp]),
    "dcn_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "dcn_stats_allreduce_rccl": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, _u64p]),
    "dcn_comm_destroy": (None, [_vp]),
    "dcn_index_device": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "dcn_index_memory": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "dcn_index_clone": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "dcn_filter_batch_device": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint64, C.c_uint32,
                                          C.POINTER(Params), _vp, _vp, _vp]),
    "dcn_ctx_synchronize": (C.c_int, [_vp]),
    "dcn_ctx_reserve_records": (C.c_int, [_vp, C.c_uint64]),
    "dcn_ctx_stream": (_vp, [_vp]),
    "dcn_host_alloc": (C.c_int, [C.c_uint64, C.POINTER(_vp)]),
    "dcn_host_free": (None, [_vp]),
    "dcn_minimizer_hashes_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint64, _vp, _vp, _vp, C.c_uint64]),
    "dcn_should_keep_hashes": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(Params), _vp, _vp, _vp]),
    "dcn_ctx_stats": (C.c_int, [_vp, _u64p]),
    "dcn_ctx_reset_stats": (C.c_int, [_vp]),
    "dcn_ctx_set_profiling": (C.c_int, [_vp, C.c_int]),
    "dcn_ctx_profile": (C.c_int, [_vp, C.POINTER(C.c_double), _u64p]),
}

_lib = None
ABI = (1, 1)  # DCN_ABI_MAJOR, the DCN_ABI_MINOR these signatures need


def lib():
    """Load the shared library (once).  Raises if it has not been built: the product has no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
            "(make -C deacon-server_amd/csrc). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    major, minor = C.c_uint32(), C.c_uint32()
    if L.dcn_abi_version(C.byref(major), C.byref(minor)) != 0 or (major.value, minor.value >= ABI[1]) != (ABI[0], True):
        raise ImportError(f"{LIB_PATH} has ABI {major.value}.{minor.value}; these bindings were written against {ABI[0]}.{ABI[1]} "
                          "(include/deacon_hip.h: DCN_ABI_MAJOR / DCN_ABI_MINOR)")
    _lib = L
    return L


def check(rc):
    if rc != DCN_OK:
        msg = lib().dcn_last_error()
        raise DeaconHipError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
