"""ctypes loader for libcolorid_hip.so.  Fails loudly when the HIP extension is missing."""
import ctypes as C
import os

try:
    # PyTorch wheels bundle their own libamdhip64; whichever HIP runtime is loaded first serves the whole
    # process.  Loading torch's first keeps torch (streams, RCCL) and this library on ONE runtime; loading
    # ours first leaves torch without a device.  Without torch the system ROCm runtime is used.
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COLORID_HIP_LIB") or os.path.join(_HERE, "libcolorid_hip.so")  # override: A/B experiments
_LIB = None

u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
vp = C.c_void_p

# name -> (restype, argtypes): every symbol include/colorid_hip.h declares
SIGNATURES = {
    "cid_last_error": (C.c_char_p, []),
    "cid_abi_version": (C.c_int, []),
    "cid_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "cid_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "cid_pinned_alloc": (C.c_int, [C.c_size_t, C.POINTER(vp)]),
    "cid_pinned_free": (None, [vp]),
    "cid_ctx_set_stream": (C.c_int, [vp, vp]),
    "cid_ctx_synchronize": (C.c_int, [vp]),
    "cid_ctx_destroy": (None, [vp]),
    "cid_index_create": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "cid_index_set_minimizer": (C.c_int, [vp, C.c_uint32]),
    "cid_index_set_hash_variant": (C.c_int, [vp, C.c_int]),
    "cid_index_put_rows": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "cid_index_put_records": (C.c_int, [vp, vp, C.c_size_t]),
    "cid_index_device_matrix": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]),
    "cid_index_finalize": (C.c_int, [vp]),
    "cid_index_get_rows": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "cid_index_get_records": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, vp]),
    "cid_index_insert_kmers_dev": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "cid_index_insert_kmers": (C.c_int, [vp, vp, C.c_uint32, C.c_size_t]),
    "cid_index_destroy": (None, [vp]),
    "cid_search_count": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, vp, vp, vp]),
    "cid_search_count_dev": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, vp, vp, vp]),
    "cid_search_perfect": (C.c_int, [vp, vp, vp, C.c_size_t, vp, C.POINTER(C.c_int)]),
    "cid_search_count_codes_dev": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, vp, vp, vp]),
    "cid_search_count_stripe_dev": (C.c_int, [vp, vp, vp, vp, C.c_size_t, C.c_uint32, vp, vp]),
    "cid_search_unique_finalize_dev": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_uint32, vp, vp, vp]),
    "cid_search_perfect_stripe_dev": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, vp]),
    "cid_index_row_stride_words": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "cid_readid_stripe_zero_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_size_t, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp, vp]),
    "cid_readid_stripe_count_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                              C.c_int, vp, vp, vp, vp]),
    "cid_readid_stripe_mask_words": (C.c_int, [C.c_uint32, C.c_uint32, vp, vp, C.c_size_t, C.POINTER(C.c_uint64)]),
    "cid_readid_stripe_zero": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint32, vp, vp, vp]),
    "cid_readid_stripe_count": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                          vp, vp, vp, vp]),
    "cid_warmup": (C.c_int, [vp, C.c_uint]),
    "cid_bgzf_inflate": (C.c_int, [vp, vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "cid_bgzf_inflate_start": (C.c_int, [vp, vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t, C.c_size_t]),
    "cid_bgzf_inflate_finish": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "cid_kmerset_create": (C.c_int, [vp, C.c_uint32, C.POINTER(vp)]),
    "cid_kmerset_add_seqs": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_int]),
    "cid_kmerset_add_seqs_dev": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_uint64, C.c_int]),
    "cid_kmerset_finalize": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "cid_kmerset_size": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "cid_kmerset_count_histogram": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "cid_kmerset_clean": (C.c_int, [vp, C.c_uint64]),
    "cid_kmerset_order_for_index": (C.c_int, [vp, vp]),
    "cid_kmerset_set_target_index": (C.c_int, [vp, vp]),
    "cid_order_codes_for_index_dev": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, vp]),
    "cid_kmerset_download": (C.c_int, [vp, vp, vp]),
    "cid_kmerset_device_arrays": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]),
    "cid_kmerset_device_ascii": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]),
    "cid_kmerset_destroy": (None, [vp]),
    "cid_index_insert_kmerset": (C.c_int, [vp, vp, C.c_uint32]),
    "cid_search_count_set": (C.c_int, [vp, vp, vp, vp, vp, vp, vp]),
    "cid_search_perfect_set": (C.c_int, [vp, vp, vp, vp, C.POINTER(C.c_int)]),
    "cid_search_count_set_report": (C.c_int, [vp, vp, vp, vp, vp, vp, vp]),
    "cid_unique_freq_modes_dev": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_uint32, vp]),
    "cid_readid_count": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "cid_readid_count_sparse": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(C.c_uint64)]),
    "cid_readid_sparse_fetch": (C.c_int, [vp, vp, vp, vp]),
    "cid_readid_count_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp, vp]),
    "cid_readid_count_resident": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "cid_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]),
    "cid_group_size": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "cid_group_ctx": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "cid_group_uses_rccl": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "cid_group_destroy": (None, [vp]),
    "cid_group_replicate_index": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "cid_group_search_count": (C.c_int, [vp, C.POINTER(vp), vp, vp, C.c_size_t, vp, vp, vp, vp]),
    "cid_group_search_count_set": (C.c_int, [vp, C.POINTER(vp), vp, vp, vp, vp, vp]),
    "cid_group_search_perfect": (C.c_int, [vp, C.POINTER(vp), vp, C.c_size_t, vp, C.POINTER(C.c_int)]),
    "cid_group_search_perfect_set": (C.c_int, [vp, C.POINTER(vp), vp, vp, C.POINTER(C.c_int)]),
    "cid_group_readid_count_sparse": (C.c_int, [vp, C.POINTER(vp), vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp, vp,
                                                C.POINTER(C.c_uint64)]),
    "cid_group_readid_sparse_fetch": (C.c_int, [vp, vp, vp, vp]),
    "cid_group_kmerset_create": (C.c_int, [vp, C.c_uint32, C.POINTER(vp)]),
    "cid_group_kmerset_add_seqs": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_int]),
    "cid_group_kmerset_finalize": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "cid_group_kmerset_size": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "cid_group_kmerset_part_sizes": (C.c_int, [vp, vp]),
    "cid_group_kmerset_count_histogram": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "cid_group_kmerset_clean": (C.c_int, [vp, C.c_uint64]),
    "cid_group_kmerset_download": (C.c_int, [vp, vp, vp]),
    "cid_group_kmerset_destroy": (None, [vp]),
    "cid_group_search_count_parts": (C.c_int, [vp, C.POINTER(vp), vp, vp, vp, vp, vp]),
    "cid_group_search_count_parts_report": (C.c_int, [vp, C.POINTER(vp), vp, vp, vp, vp, vp]),
    "cid_group_search_perfect_parts": (C.c_int, [vp, C.POINTER(vp), vp, vp, C.POINTER(C.c_int)]),
    "cid_group_stripes_create": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "cid_group_stripes_base": (C.c_int, [vp, C.POINTER(vp), vp]),
    "cid_group_stripes_put_records": (C.c_int, [vp, C.POINTER(vp), vp, C.c_size_t]),
    "cid_group_stripes_put_rows": (C.c_int, [vp, C.POINTER(vp), vp, vp, C.c_size_t]),
    "cid_group_stripes_search_count": (C.c_int, [vp, C.POINTER(vp), vp, vp, C.c_size_t, vp, vp, vp, vp]),
    "cid_group_stripes_search_count_set": (C.c_int, [vp, C.POINTER(vp), vp, vp, vp, vp, vp]),
    "cid_group_stripes_search_count_set_report": (C.c_int, [vp, C.POINTER(vp), vp, vp, vp, vp, vp]),
    "cid_group_stripes_search_perfect": (C.c_int, [vp, C.POINTER(vp), vp, C.c_size_t, vp, C.POINTER(C.c_int)]),
    "cid_group_stripes_search_perfect_set": (C.c_int, [vp, C.POINTER(vp), vp, vp, C.POINTER(C.c_int)]),
    "cid_group_stripes_readid_count_sparse": (C.c_int, [vp, C.POINTER(vp), vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp, vp,
                                                        C.POINTER(C.c_uint64)]),
    "cid_ctx_tune": (C.c_int, [vp, C.c_char_p, C.c_long]),
    "cid_fastq_create": (C.c_int, [vp, C.c_int, C.c_uint32, C.POINTER(vp)]),
    "cid_fastq_push_bgzf": (C.c_int, [vp, C.c_int, vp, C.c_size_t, vp, vp, vp, C.c_size_t, C.c_int]),
    "cid_fastq_push_text": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.c_int]),
    "cid_fastq_classify": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cid_fastq_count_kmers": (C.c_int, [vp, vp, C.c_int, C.POINTER(C.c_uint64)]),
    "cid_fastq_classify_begin": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_int]),
    "cid_fastq_classify_end": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cid_fastq_fetch": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "cid_fastq_destroy": (None, [vp]),
    "cid_timer_start": (C.c_int, [vp]),
    "cid_timer_stop_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
}


class CidError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcolorid_hip error {code}: {msg}")
        self.code = code


def open_library(path):
    """A configured handle on one build of the library (the shipped one, or libcolorid_hip_tune.so for the A/B tests and tools)."""
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        if path != os.path.join(_HERE, "libcolorid_hip.so") and not hasattr(lib, name):
            continue  # an older build loaded on purpose for an A/B measurement
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


TUNE_LIB_PATH = os.path.join(_HERE, "libcolorid_hip_tune.so")   # `make -C colorid_amd/csrc tune`: + the two rejected schedulings


def load_library():
    """Return the loaded C-ABI library; raise if it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  colorid_amd has no CPU fallback.")
    _LIB = open_library(LIB_PATH)
    return _LIB


def check(rc):
    if rc != 0:
        raise CidError(rc, load_library().cid_last_error().decode(errors="replace"))
