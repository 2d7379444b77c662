"""ctypes binding of libsplicedice_hip.so (include/sdice.h).

There is no CPU backend and no fallback: if the shared library is missing or cannot be
loaded, importing the engine raises, and creating a context without a gfx950 device fails
with the library's error message.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPLICEDICE_HIP_LIB", os.path.join(_HERE, "lib", "libsplicedice_hip.so"))

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_i8p = C.POINTER(C.c_int8)
c_u8p = C.POINTER(C.c_uint8)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
vp = C.c_void_p
ctxp = C.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/sdice.h
SIGNATURES = {
    "sdice_version": [],
    "sdice_last_error": [],
    "sdice_ctx_create": [C.c_int, C.POINTER(ctxp)],
    "sdice_ctx_destroy": [ctxp],
    "sdice_sync": [ctxp],
    "sdice_trim": [ctxp],
    "sdice_device_info": [ctxp, C.c_char_p, C.c_int, C.POINTER(C.c_int), c_i64p],
    "sdice_dmalloc": [ctxp, C.c_int64, C.POINTER(vp)],
    "sdice_dfree": [ctxp, vp],
    "sdice_h2d": [ctxp, vp, vp, C.c_int64],
    "sdice_d2h": [ctxp, vp, vp, C.c_int64],
    "sdice_dmemset": [ctxp, vp, C.c_int, C.c_int64],
    "sdice_cluster": [ctxp, C.c_int64, vp, vp, vp, vp, vp, vp, c_i64p],
    "sdice_cluster_col": [ctxp, vp, C.c_int64],
    "sdice_cluster_dev": [ctxp, C.c_int64, vp, vp, vp, vp, vp, vp, c_i64p],
    "sdice_cluster_col_dev": [ctxp, C.POINTER(vp), c_i64p],
    "sdice_cluster_status": [ctxp, c_i64p, C.POINTER(C.c_int32)],
    "sdice_ps": [ctxp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp],
    "sdice_ps_dev": [ctxp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp],
    "sdice_ps_f64": [ctxp, C.c_int64, C.c_int64, C.c_int32, vp, vp, vp, vp],
    "sdice_ps_f64_dev": [ctxp, C.c_int64, C.c_int64, C.c_int32, vp, vp, vp, vp],
    "sdice_excl_f64": [ctxp, C.c_int64, C.c_int64, C.c_int32, vp, vp, vp, vp],
    "sdice_excl_f64_dev": [ctxp, C.c_int64, C.c_int64, C.c_int32, vp, vp, vp, vp],
    "sdice_mark_low": [ctxp, C.c_int64, vp, vp, C.c_int64],
    "sdice_mark_low_dev": [ctxp, C.c_int64, vp, vp, C.c_int64],
    "sdice_quantize3": [ctxp, C.c_int64, vp],
    "sdice_quantize3_dev": [ctxp, C.c_int64, vp],
    "sdice_ranksum": [ctxp, C.c_int64, C.c_int32, vp, vp, C.c_int32, vp, C.c_int32] + [vp] * 8,
    "sdice_ranksum_dev": [ctxp, C.c_int64, C.c_int32, vp, vp, C.c_int32, vp, C.c_int32] + [vp] * 8,
    "sdice_fisher_pairs": [ctxp, C.c_int64, C.c_int32, vp, vp, vp],
    "sdice_fisher_pairs_dev": [ctxp, C.c_int64, C.c_int32, vp, vp, vp],
    "sdice_fisher_tables": [ctxp, C.c_int64, vp, vp],
    "sdice_fisher_step_stats": [ctxp, vp, vp],
    "sdice_chi2_pairs": [ctxp, C.c_int64, C.c_int32, vp, vp, vp, c_i64p],
    "sdice_chi2_pairs_dev": [ctxp, C.c_int64, C.c_int32, vp, vp, vp, vp],
    "sdice_bh": [ctxp, C.c_int64, vp, vp],
    "sdice_bh_dev": [ctxp, C.c_int64, vp, vp],
    "sdice_bh_masked_dev": [ctxp, C.c_int64, vp, vp, vp],
    "sdice_bh_columns": [ctxp, C.c_int64, C.c_int64, vp],
    "sdice_bh_columns_dev": [ctxp, C.c_int64, C.c_int64, vp],
    "sdice_bh_columns_pitched_dev": [ctxp, C.c_int64, C.c_int64, C.c_int64, vp],
    "sdice_textio_stats": [vp, C.c_int],
    "sdice_write_table": [C.c_char_p, C.c_char_p, C.c_int64, C.c_int32, vp, vp, vp, C.c_int, C.c_int, C.c_int],
    "sdice_write_clusters": [C.c_char_p, C.c_int64, vp, vp, vp, vp, C.c_int],
    "sdice_write_columns_sfx": [C.c_char_p, C.c_char_p, C.c_int64, vp, vp, C.c_int32, vp, vp, vp, vp, vp, C.c_int],
    "sdice_interval_overlaps": [C.c_int64, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, C.c_int64, C.c_int],
    "sdice_write_junction_bed": [C.c_char_p, C.c_int64, vp, vp, C.c_int32, vp, vp, vp, vp, C.c_int],
    "sdice_textio_trim": [],
    "sdice_junction_names": [C.c_int64, vp, vp, C.c_int32, vp, vp, vp, vp, vp, C.c_int64, vp, c_i64p],
    "sdice_write_columns": [C.c_char_p, C.c_char_p, C.c_int64, vp, vp, C.c_int32, vp, vp, vp, C.c_int],
    "sdice_table_open": [C.c_char_p, C.POINTER(vp), c_i64p, C.POINTER(C.c_int32), c_i64p, c_i64p],
    "sdice_table_read": [vp, vp, vp, vp, vp, C.c_int, C.c_int],
    "sdice_table_close": [vp],
    "sdice_junc_open": [C.c_char_p, C.c_int, C.POINTER(vp), c_i64p, C.POINTER(C.c_int32), c_i64p],
    "sdice_junc_read": [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                        C.c_int],
    "sdice_junc_close": [vp],
    "sdice_junc_lookup": [C.c_int64, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, C.c_int],
    "sdice_junc_count_column": [C.c_int64, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, C.c_int32, vp, vp],
    "sdice_junc_pack_keys": [C.c_int64, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp, c_i64p, C.POINTER(C.c_int32)],
    "sdice_transpose_i32": [C.c_int64, C.c_int64, vp, vp, C.c_int],
    "sdice_host_threads": [],
    "sdice_sort_unique_u64": [ctxp, C.c_int64, vp, c_i64p],
    "sdice_similarity": [ctxp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp],
    "sdice_similarity_dev": [ctxp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp],
    "sdice_rowstats": [ctxp, C.c_int64, C.c_int32, vp, C.c_int, vp, C.c_int32, vp, vp, vp],
    "sdice_rowstats_dev": [ctxp, C.c_int64, C.c_int32, vp, C.c_int, vp, C.c_int32, vp, vp, vp],
    "sdice_shard_plan": [C.c_int64, vp, vp, C.c_int32, C.c_double, vp],
    "sdice_shard_plan_junctions": [C.c_int64, vp, vp, vp, vp, C.c_int32, C.c_double, vp],
    "sdice_comm_unique_id": [ctxp, vp],
    "sdice_comm_init": [ctxp, vp, C.c_int, C.c_int],
    "sdice_comm_destroy": [ctxp],
    "sdice_comm_fork": [ctxp],
    "sdice_comm_join": [ctxp],
    "sdice_allgather_dev": [ctxp, vp, vp, C.c_int64],
    "sdice_alltoall_dev": [ctxp, vp, vp, C.c_int64],
    "sdice_copy2d_dev": [ctxp, vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int64],
    "sdice_prof_enable": [ctxp, C.c_int],
    "sdice_prof_reset": [ctxp],
    "sdice_prof_query": [ctxp, C.c_char_p, c_i64p, c_f64p],
    "sdice_prof_report": [ctxp, C.c_char_p, C.c_int],
    "sdice_timer_start": [ctxp],
    "sdice_timer_stop": [ctxp, c_f64p],
    "sdice_set_param": [ctxp, C.c_char_p, C.c_int64],
}
_RESTYPE = {"sdice_last_error": C.c_char_p}

_lib = None


class SdiceError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdiceError(
            f"HIP extension not built: {LIB_PATH} is missing. Build it with "
            f"`make -C splicedice_amd/csrc` (or python -c 'import __graft_entry__ as g; g.build()'). "
            f"There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().sdice_last_error()
        raise SdiceError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")
