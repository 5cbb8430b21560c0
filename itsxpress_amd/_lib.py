"""ctypes binding of the engine's C ABI (include/itsx_hip.h -> itsxpress_amd/libitsx_hip.so).

The library is built in-tree by `__graft_entry__.build()` / `make -C itsxpress_amd/csrc`.
There is no fallback: if the shared object is missing, or no gfx950 device is usable, the
calls below raise -- FileNotFoundError for a missing engine (the class the reference raises
when vsearch/hmmsearch are missing, itsxpress/SeqSample.py:127-131,221-225), EngineError for
everything the engine reports.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libitsx_hip.so")
ABI_VERSION = 6          # include/itsx_hip.h: ITSX_ABI_VERSION
_LIB = None


class EngineError(subprocess.SubprocessError):
    """Raised when the HIP engine reports an error (plays the part of CalledProcessError)."""

    def __init__(self, code, message):
        super().__init__("itsx_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


DOMAIN_DTYPE = np.dtype([("rep", "<i8"), ("prof", "<i4"), ("tlen", "<i4"), ("ienv", "<i4"), ("jenv", "<i4"),
                         ("dom_idx", "<i4"), ("ndom", "<i4"), ("flags", "<i4"), ("envsc", "<f4"),
                         ("domcorrection", "<f4"), ("dombias", "<f4"), ("bitscore", "<f4"), ("lnP", "<f8"),
                         ("seq_score", "<f4"), ("seq_bias", "<f4"), ("seq_reported", "<i4"),
                         ("dom_reported", "<i4")], align=True)
TRACE_DTYPE = np.dtype([("rep", "<i8"), ("prof", "<i4"), ("msv_xj", "<i4"), ("pass_msv", "<i4"),
                        ("pass_bias", "<i4"), ("pass_fwd", "<i4"), ("msv_sc", "<f4"), ("filtersc", "<f4"),
                        ("fwdsc", "<f4"), ("bcksc", "<f4"), ("nullsc", "<f4"), ("nregions", "<i4"),
                        ("ndom", "<i4"), ("ran_vit", "<i4"), ("pass_vit", "<i4"), ("vitsc", "<f4"), ("pad", "<i4")], align=True)
STATS_FIELDS = [("n_reads", "<i8"), ("n_unique", "<i8"), ("n_dropped_short", "<i8"), ("n_pairs", "<i8"),
                ("n_past_msv", "<i8"), ("n_past_bias", "<i8"), ("n_past_fwd", "<i8"), ("n_regions", "<i8"),
                ("n_multidomain", "<i8"), ("n_domains", "<i8"), ("n_domain_overflow", "<i8"),
                ("n_profiles", "<i4"), ("hash_reseeds", "<i4"), ("ms_derep", "<f4"), ("ms_msv", "<f4"),
                ("ms_filters", "<f4"), ("ms_domains", "<f4"), ("ms_finalize", "<f4"), ("ms_msv_kernel", "<f4"),
                ("msv_cells", "<i8"), ("msv_launches", "<i8"), ("ms_fwd_kernel", "<f4"), ("ms_bwd_kernel", "<f4"),
                ("fwd_rows", "<i8"), ("ms_env_kernel", "<f4"), ("ms_bias_kernel", "<f4"), ("env_rows", "<i8"),
                ("n_env_unique", "<i8"), ("ms_decode_kernel", "<f4"), ("n_batches", "<i4"), ("ms_cluster", "<f4"),
                ("pad0", "<i4"), ("cl_windows", "<i8"), ("cl_cuts", "<i8"), ("cl_alignments", "<i8"), ("ms_merge", "<f4"),
                ("pad1", "<i4"), ("cl_certified", "<i8"), ("ms_pack", "<f4"), ("pad2", "<i4"),
                ("n_uniq_multi_winner", "<i8"), ("n_reads_multi_winner", "<i8"), ("n_uniq_region_cap", "<i8"),
                ("n_reads_region_cap", "<i8"), ("n_mr_clustered", "<i8"), ("n_mr_failed", "<i8"), ("n_mr_envelopes", "<i8"),
                ("ms_ensemble", "<f4"), ("pad3", "<i4"), ("n_mr_distinct", "<i8"), ("n_slab_shrinks", "<i8"), ("ms_vit_kernel", "<f4"), ("pad4", "<i4"),
                ("n_mr_fail_kind", "<i8", (8,)), ("n_rows_resident", "<i8"),
                ("lazy", "<i4"), ("n_bound_launches", "<i4"), ("n_lazy_pending_profiles", "<i8"), ("n_lazy_completed", "<i8"), ("n_lazy_completed_profiles", "<i8"), ("n_mr_overflow", "<i8"), ("ms_lazy_complete", "<f4"), ("lazy_bound_maxdiff", "<f4"), ("n_lazy_evaluated", "<i8"), ("n_lazy_round1", "<i8"),
                ("n_lazy_pending", "<i8"), ("n_lazy_reruns", "<i8"), ("bound_rows", "<i8"), ("ms_bound_kernel", "<f4"),
                ("ms_lazy_select", "<f4"),
                ("share_B", "<i4"), ("share_batches", "<i4"), ("share_nodes", "<i8"), ("share_chains", "<i8"), ("msv_rows", "<i8"), ("msv_rows_full", "<i8"),
                ("bound_rows_full", "<i8"), ("n_share_helpers", "<i8"), ("share_mismatch", "<i8"), ("ms_share_build", "<f4"), ("share_frac", "<f4"),
                ("two_sided", "<i4"), ("n_bwd_launches", "<i4"), ("n_joined", "<i8"), ("bwd_chains", "<i8"), ("gamma_nodes", "<i8"), ("bwd_rows", "<i8"),
                ("two_fwd_rows", "<i8"), ("two_bwd_rows", "<i8"), ("two_rows_full", "<i8"), ("join_maxdiff", "<f4"), ("ms_bwd_bound", "<f4"),
                ("n_lazy_topup", "<i8"), ("ms_lazy_topup", "<f4"), ("ms_lazy_topup_stages", "<f4")]
STATS_DTYPE = np.dtype(STATS_FIELDS, align=True)

# every symbol include/itsx_hip.h declares
EXPORTS = ["itsx_abi_version", "itsx_last_error", "itsx_create", "itsx_destroy", "itsx_load_profiles_file",
           "itsx_load_profiles_mem", "itsx_profile_name", "itsx_profile_tables", "itsx_set_reads", "itsx_set_reads_view", "itsx_set_reads_device", "itsx_domz_device", "itsx_trim_coords_device",
           "itsx_rep_coords_device", "itsx_derep_device", "itsx_unique_keys128_device",
           "itsx_load_reads_file", "itsx_derep", "itsx_cluster", "itsx_get_cluster", "itsx_get_derep", "itsx_unique_keys", "itsx_set_active_uniques", "itsx_get_uniques",
           "itsx_search", "itsx_get_domz", "itsx_set_domz", "itsx_search_finalize", "itsx_num_domains",
           "itsx_get_domains", "itsx_num_pairtraces", "itsx_get_pairtraces", "itsx_trim_coords",
           "itsx_rep_coords", "itsx_write_uc", "itsx_write_rep_fasta", "itsx_write_domtbl", "itsx_get_stats", "itsx_switches", "itsx_switch_registry", "itsx_release_scratch",
           "itsx_debug_read_hashes", "itsx_debug_packed_read", "itsx_debug_detmath", "itsx_debug_logf", "itsx_debug_dust", "itsx_debug_calibrate", "itsx_debug_issue", "itsx_shard_text", "itsx_shard_last_error", "itsx_owner_verdicts",
           "itsx_write_trimmed_fastq", "itsx_write_trimmed_paired", "itsx_trim_last_error",
           "itsx_merge_buffers", "itsx_merge_pairs_files", "itsx_merge_pairs_load", "itsx_merge_tables",
           "itsx_orient_load_db", "itsx_orient", "itsx_write_oriented_fastq",
           "itsx_io_read", "itsx_io_free", "itsx_io_codecs", "itsx_fastq_ids",
           "itsx_load_reads_files", "itsx_set_samples", "itsx_num_samples", "itsx_select_sample",
           "itsx_io_parallel_inflates", "itsx_io_cache_clear", "itsx_get_read_names",
           "itsx_set_rows_mode", "itsx_lazy_pending", "itsx_domz_count", "itsx_lazy_pending_profiles", "itsx_lazy_complete",
           "itsx_load_reads_file_shard", "itsx_unique_keys128", "itsx_write_derep_arrays", "itsx_write_domtbl_arrays",
           "itsx_writers_last_error", "itsx_profile_params", "itsx_get_unique_seqs",
           "itsx_load_reads_text", "itsx_stream_open", "itsx_stream_next", "itsx_stream_close", "itsx_stream_last_error", "itsx_stream_records_bound",
           "itsx_stream_open_shared", "itsx_stream_open_threads", "itsx_stream_base", "itsx_stream_progress", "itsx_stream_next_records", "itsx_count_records",
           "itsx_merge_pairs_load_text", "itsx_merge_pair_index", "itsx_twriter_set_mode", "itsx_write_range",
           "itsx_keyset_create", "itsx_keyset_destroy", "itsx_keyset_size", "itsx_keyset_assign",
           "itsx_twriter_open", "itsx_twriter_text", "itsx_twriter_coords", "itsx_twriter_update", "itsx_twriter_close",
           "itsx_lazy_pending_uniques", "itsx_set_partial_coords", "itsx_set_kept_rows"]


def lib():
    """Load libitsx_hip.so (once).  Raises FileNotFoundError if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s not found: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C itsxpress_amd/csrc). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f64, cp = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_char_p
    sig = {
        "itsx_abi_version": (i32, []),
        "itsx_last_error": (cp, [vp]),
        "itsx_create": (vp, [i32, i32]),
        "itsx_destroy": (None, [vp]),
        "itsx_load_profiles_file": (i32, [vp, cp, vp]),
        "itsx_load_profiles_mem": (i32, [vp, cp, i64, vp]),
        "itsx_profile_name": (i32, [vp, i32, vp, i32]),
        "itsx_profile_tables": (i32, [vp, i32, vp, vp, vp, vp]),
        "itsx_set_reads": (i32, [vp, vp, vp, i64, vp, vp]),
        "itsx_set_reads_view": (i32, [vp, vp, vp, i64, vp, vp]),
        "itsx_set_reads_device": (i32, [vp, vp, vp, i64, vp, vp]),
        "itsx_domz_device": (i32, [vp, vp, vp]),
        "itsx_trim_coords_device": (i32, [vp, cp, cp, vp, vp]),
        "itsx_rep_coords_device": (i32, [vp, cp, cp, vp, vp]),
        "itsx_derep_device": (i32, [vp, vp, vp, vp, vp]),
        "itsx_unique_keys128_device": (i32, [vp, C.c_uint64, C.c_uint64, i64, vp, vp]),
        "itsx_load_reads_file": (i32, [vp, cp, vp]),
        "itsx_load_reads_file_shard": (i32, [vp, cp, i32, i32, vp, vp, vp]),
        "itsx_load_reads_text": (i32, [vp, vp, i64, vp]),
        "itsx_stream_open": (i32, [cp, vp]),
        "itsx_stream_next": (i32, [vp, i64, vp, vp, vp]),
        "itsx_stream_close": (i32, [vp, i32]),
        "itsx_stream_records_bound": (i64, [vp]),
        "itsx_stream_open_shared": (i32, [cp, cp, vp, vp]),
        "itsx_stream_open_threads": (i32, [cp, i32, vp]),
        "itsx_stream_base": (vp, [vp]),
        "itsx_stream_progress": (i32, [vp, vp, vp, vp]),
        "itsx_stream_next_records": (i32, [vp, i64, vp, vp, vp, vp]),
        "itsx_count_records": (i64, [vp, i64]),
        "itsx_merge_pairs_load_text": (i32, [vp, vp, i64, vp, i64, i32, f64, i32, vp, vp]),
        "itsx_merge_pair_index": (i32, [vp, vp, i64]),
        "itsx_twriter_set_mode": (i32, [vp, i32]),
        "itsx_write_range": (i32, [cp, vp, i64]),
        "itsx_stream_last_error": (cp, []),
        "itsx_keyset_create": (vp, []),
        "itsx_keyset_destroy": (None, [vp]),
        "itsx_keyset_size": (i64, [vp]),
        "itsx_keyset_assign": (i32, [vp, vp, i64, i32, vp, vp]),
        "itsx_twriter_open": (i32, [cp, i32, i32, vp]),
        "itsx_twriter_text": (i32, [vp, vp, i64, i32]),
        "itsx_twriter_coords": (i32, [vp, i64, i64, vp, vp, vp]),
        "itsx_twriter_update": (i32, [vp, vp, i64, vp, vp]),
        "itsx_twriter_close": (i32, [vp, vp, vp]),
        "itsx_lazy_pending_uniques": (i32, [vp, vp]),
        "itsx_set_partial_coords": (i32, [vp, i32]),
        "itsx_set_kept_rows": (i32, [vp, i32]),
        "itsx_unique_keys128": (i32, [vp, C.c_uint64, C.c_uint64, i64, vp]),
        "itsx_write_derep_arrays": (i32, [cp, cp, i64, vp, vp, vp, vp, vp, vp, vp, i64]),
        "itsx_write_domtbl_arrays": (i32, [cp, vp, i64, i64, vp, i32, vp, vp, vp, vp, vp, vp, vp]),
        "itsx_writers_last_error": (cp, []),
        "itsx_profile_params": (i32, [vp, i32, vp, vp]),
        "itsx_get_unique_seqs": (i32, [vp, vp, i64, vp]),
        "itsx_derep": (i32, [vp, i32, i32, vp]),
        "itsx_cluster": (i32, [vp, f64, i32, vp]),
        "itsx_get_cluster": (i32, [vp, vp, vp, vp]),
        "itsx_merge_buffers": (i32, [vp, cp, cp, vp, cp, cp, vp, i64, i32, f64, i32, vp, vp, vp, vp, vp, vp]),
        "itsx_merge_pairs_files": (i32, [vp, cp, cp, cp, i32, f64, i32, vp, vp]),
        "itsx_merge_pairs_load": (i32, [vp, cp, cp, i32, f64, i32, vp, vp]),
        "itsx_merge_tables": (i32, [vp, vp, vp, vp, vp]),
        "itsx_orient_load_db": (i32, [vp, cp, vp]),
        "itsx_orient": (i32, [vp, vp, vp, vp]),
        "itsx_write_oriented_fastq": (i32, [cp, cp, vp, i64, vp]),
        "itsx_unique_keys": (i32, [vp, C.c_uint64, vp, vp]),
        "itsx_set_active_uniques": (i32, [vp, vp]),
        "itsx_get_derep": (i32, [vp, vp, vp, vp]),
        "itsx_get_uniques": (i32, [vp, vp, vp]),
        "itsx_search": (i32, [vp, f64, f64, f64, f64]),
        "itsx_set_rows_mode": (i32, [vp, i32]),
        "itsx_lazy_pending": (i64, [vp]),
        "itsx_lazy_pending_profiles": (i32, [vp, vp]),
        "itsx_lazy_complete": (i32, [vp, vp]),
        "itsx_domz_count": (i64, [vp]),
        "itsx_get_domz": (i32, [vp, vp]),
        "itsx_set_domz": (i32, [vp, vp]),
        "itsx_search_finalize": (i32, [vp, f64]),
        "itsx_num_domains": (i64, [vp]),
        "itsx_get_domains": (i32, [vp, vp]),
        "itsx_num_pairtraces": (i64, [vp]),
        "itsx_get_pairtraces": (i32, [vp, vp, i64]),
        "itsx_trim_coords": (i32, [vp, cp, cp, vp, vp, vp, vp]),
        "itsx_rep_coords": (i32, [vp, cp, cp, vp, vp, vp, vp]),
        "itsx_write_uc": (i32, [vp, cp]),
        "itsx_write_rep_fasta": (i32, [vp, cp]),
        "itsx_write_domtbl": (i32, [vp, cp]),
        "itsx_get_stats": (i32, [vp, vp, i64]),
        "itsx_switches": (i64, [vp, vp, i64]),
        "itsx_release_scratch": (i32, [vp]),
        "itsx_switch_registry": (i64, [vp, i64]),
        "itsx_debug_read_hashes": (i32, [vp, vp, vp]),
        "itsx_debug_packed_read": (i32, [vp, i64, vp, vp, vp, vp]),
        "itsx_debug_detmath": (i32, [vp, vp, i64, vp, vp]),
        "itsx_debug_logf": (i32, [vp, vp, i64, vp]),
        "itsx_debug_dust": (i32, [vp, vp]),
        "itsx_debug_calibrate": (i32, [vp, i32, f64, i32, vp, vp]),
        "itsx_debug_issue": (i32, [vp, i32, i32, i32, vp, vp]),
        "itsx_shard_text": (i32, [C.c_char_p, i32, vp, C.c_char_p, vp, vp]),
        "itsx_shard_last_error": (C.c_char_p, []),
        "itsx_owner_verdicts": (i32, [vp, vp, i64, vp]),
        "itsx_write_trimmed_fastq": (i32, [cp, cp, i32, i32, vp, vp, i64, vp, vp]),
        "itsx_write_trimmed_paired": (i32, [cp, cp, cp, cp, i32, i32, vp, vp, i64, vp, vp, vp, vp]),
        "itsx_trim_last_error": (cp, []),
        "itsx_io_read": (i32, [cp, vp, vp]),
        "itsx_fastq_ids": (i32, [cp, vp, vp, vp]),
        "itsx_io_free": (None, [vp]),
        "itsx_io_codecs": (i32, []),
        "itsx_io_parallel_inflates": (i64, []),
        "itsx_io_cache_clear": (None, []),
        "itsx_get_read_names": (i32, [vp, vp, i64, vp]),
        "itsx_load_reads_files": (i32, [vp, vp, i32, vp]),
        "itsx_set_samples": (i32, [vp, vp, i32]),
        "itsx_num_samples": (i32, [vp]),
        "itsx_select_sample": (i32, [vp, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.itsx_abi_version() != ABI_VERSION:
        raise EngineError(-1, "ABI version mismatch: %s reports %d, this package binds version %d" % (LIB_PATH, L.itsx_abi_version(), ABI_VERSION))
    _LIB = L
    return L
