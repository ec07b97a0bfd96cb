"""ctypes / numpy mirrors of the C-ABI structs declared in include/snapgpu.h."""
from __future__ import annotations

import ctypes as C

import numpy as np

# snapgpu_single_result (POD mirror of SingleAlignmentResult, SNAPLib/AlignmentResult.h:48-76)
RESULT_DTYPE = np.dtype([
    ("status", np.int32),
    ("direction", np.int32),
    ("location", np.int64),
    ("orig_location", np.int64),
    ("score", np.int32),
    ("score_prior_to_clipping", np.int32),
    ("mapq", np.int32),
    ("clipping_for_read_adjustment", np.int32),
    ("used_affine_gap_scoring", np.int32),
    ("bases_clipped_before", np.int32),
    ("bases_clipped_after", np.int32),
    ("ag_score", np.int32),
    ("supplementary", np.int32),
    ("seed_offset", np.int32),
    ("match_probability", np.float64),
    ("probability_all_candidates", np.float64),
    ("popular_seeds_skipped", np.uint32),
    ("reserved", np.uint32),
], align=True)
assert RESULT_DTYPE.itemsize == 88, RESULT_DTYPE.itemsize

# snapgpu_paired_result (POD mirror of PairedAlignmentResult, SNAPLib/AlignmentResult.h:87-128)
PAIRED_RESULT_DTYPE = np.dtype([
    ("status", np.int32, 2),
    ("direction", np.int32, 2),
    ("location", np.int64, 2),
    ("orig_location", np.int64, 2),
    ("score", np.int32, 2),
    ("score_prior_to_clipping", np.int32, 2),
    ("mapq", np.int32, 2),
    ("clipping_for_read_adjustment", np.int32, 2),
    ("used_affine_gap_scoring", np.int32, 2),
    ("bases_clipped_before", np.int32, 2),
    ("bases_clipped_after", np.int32, 2),
    ("ag_score", np.int32, 2),
    ("supplementary", np.int32, 2),
    ("seed_offset", np.int32, 2),
    ("lv_indels", np.int32, 2),
    ("match_probability", np.float64, 2),
    ("probability_all_pairs", np.float64),
    ("popular_seeds_skipped", np.uint32, 2),
    ("used_gapless_clipping", np.int32, 2),
    ("ref_span", np.int32, 2),
    ("liftover", np.int32, 2),
    ("aligned_as_pair", np.int32),
    ("ag_forced_single_aligner_call", np.int32),
    ("reserved", np.uint32),
    ("flags", np.uint32),
], align=True)
assert PAIRED_RESULT_DTYPE.itemsize == 208, PAIRED_RESULT_DTYPE.itemsize

NOT_FOUND, SINGLE_HIT, MULTIPLE_HITS = 0, 1, 2
SCORE_ABOVE_LIMIT = -1
INVALID_GENOME_LOCATION_32 = 0xFFFFFFFF
MAX_K = 127


class IndexView(C.Structure):
    _fields_ = [
        ("seed_len", C.c_uint32),
        ("key_bytes", C.c_uint32),
        ("n_hash_tables", C.c_uint32),
        ("large_hash_table", C.c_uint32),
        ("location_size", C.c_uint32),
        ("chromosome_padding", C.c_uint32),
        ("overflow_table_size", C.c_uint64),
        ("hash_blob", C.c_void_p),
        ("hash_blob_bytes", C.c_uint64),
        ("table_offset", C.c_void_p),
        ("table_size", C.c_void_p),
        ("overflow", C.c_void_p),
        ("genome", C.c_void_p),
        ("n_bases", C.c_uint64),
        ("genome_pad", C.c_uint32),
        ("contig_begin", C.c_void_p),
        ("n_contigs", C.c_uint32),
        ("first_alt_location", C.c_uint64),
        ("on_device", C.c_uint32),
        ("contig_proj_begin", C.c_void_p),
        ("contig_proj_rc", C.c_void_p),
        ("contig_cigar_start", C.c_void_p),
        ("cigar_ops", C.c_void_p),
    ]


class IndexBuildParams(C.Structure):          # snapgpu_index_build_params
    _fields_ = [
        ("seed_len", C.c_uint32),
        ("slack", C.c_double),
        ("key_bytes", C.c_uint32),
        ("chromosome_padding", C.c_uint32),
        ("space_terminates_name", C.c_uint32),
        ("name_terminators", C.c_char_p),
        ("auto_alt", C.c_uint32),
        ("max_alt_contig_size", C.c_int64),
        ("alt_contig_names", C.POINTER(C.c_char_p)), ("n_alt_contig_names", C.c_uint32),
        ("non_alt_contig_names", C.POINTER(C.c_char_p)), ("n_non_alt_contig_names", C.c_uint32),
        ("alt_liftover_file", C.c_char_p),
    ]


class IndexBuildStats(C.Structure):           # snapgpu_index_build_stats
    _fields_ = [(n, C.c_uint64) for n in ("n_bases", "n_seed_locations", "n_distinct_seeds", "n_repeated_seeds", "overflow_table_size",
                                          "hash_table_slots", "hash_blob_bytes")] + \
               [(n, C.c_double) for n in ("ms_keys", "ms_sort", "ms_runs", "ms_tables", "ms_total_device", "s_fasta")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Params(C.Structure):
    _fields_ = [
        ("max_hits", C.c_uint32),
        ("max_k", C.c_uint32),
        ("num_seeds", C.c_uint32),
        ("seed_coverage", C.c_double),
        ("min_weight_to_check", C.c_uint32),
        ("extra_search_depth", C.c_uint32),
        ("use_affine_gap", C.c_uint32),
        ("match_reward", C.c_uint32),
        ("sub_penalty", C.c_uint32),
        ("gap_open_penalty", C.c_uint32),
        ("gap_extend_penalty", C.c_uint32),
        ("five_prime_end_bonus", C.c_uint32),
        ("three_prime_end_bonus", C.c_uint32),
        ("alt_awareness", C.c_uint32),
        ("emit_alt_alignments", C.c_uint32),
        ("max_score_gap_to_prefer_non_alt", C.c_int32),
        ("max_read_len", C.c_uint32),
    ]


class PairedParams(C.Structure):
    _fields_ = [
        ("min_spacing", C.c_int32),
        ("max_spacing", C.c_uint32),
        ("force_spacing", C.c_uint32),
        ("max_big_hits", C.c_uint32),
        ("max_candidate_pool_size", C.c_uint32),
        ("num_seeds", C.c_uint32),
        ("seed_coverage", C.c_double),
        ("max_k_for_indels", C.c_uint32),
        ("min_read_length", C.c_uint32),
        ("use_soft_clipping", C.c_uint32),
        ("flatten_mapq_at_or_below", C.c_int32),
        ("min_score_realignment", C.c_int32),
        ("min_score_gap_realignment_alt", C.c_int32),
        ("min_ag_score_improvement", C.c_int32),
        ("enable_hamming_scoring_base_aligner", C.c_uint32),
        ("max_single_seeds", C.c_uint32),
    ]


class SecondaryParams(C.Structure):
    """snapgpu_secondary_params: -om / -mpc / -omax / -ae (AlignerOptions.cpp:70-72, 96)."""
    _fields_ = [
        ("max_edit_distance", C.c_int32),
        ("max_per_contig", C.c_int32),
        ("max_results", C.c_int64),
        ("adjust_alignments", C.c_uint32),
    ]


def secondary_params(max_edit_distance: int, max_results: int = 0x7fffffff, max_per_contig: int = -1,
                     adjust_alignments: int = 0) -> SecondaryParams:
    return SecondaryParams(max_edit_distance=max_edit_distance, max_per_contig=max_per_contig, max_results=max_results,
                           adjust_alignments=adjust_alignments)


def default_paired_params(**overrides) -> PairedParams:
    """PairedAlignerOptions defaults, PairedAligner.cpp:55-57, 227-242; AlignerOptions.cpp:103-110."""
    p = PairedParams(min_spacing=0, max_spacing=1000, force_spacing=0, max_big_hits=4000,
                     max_candidate_pool_size=1000000, num_seeds=8, seed_coverage=0.0, max_k_for_indels=40,
                     min_read_length=50, use_soft_clipping=1, flatten_mapq_at_or_below=3,
                     min_score_realignment=3, min_score_gap_realignment_alt=3, min_ag_score_improvement=24,
                     enable_hamming_scoring_base_aligner=1, max_single_seeds=25)
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Counters(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64),
        ("n_hash_table_lookups", C.c_uint64),
        ("n_hash_slots_probed", C.c_uint64),
        ("n_hits_consumed", C.c_uint64),
        ("n_overflow_lists", C.c_uint64),
        ("n_lv_locations", C.c_uint64),
        ("n_ag_locations", C.c_uint64),
        ("n_lv_ref_bytes", C.c_uint64),
        ("cycles_lookup", C.c_uint64),
        ("cycles_hits", C.c_uint64),
        ("cycles_lv", C.c_uint64),
        ("cycles_ag", C.c_uint64),
        ("cycles_total", C.c_uint64),
        ("reserved", C.c_uint64 * 3),
    ]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}
        d["cycles_single_fallback"] = int(self.reserved[0])      # paired-end path only
        d["help_watchdog_events"] = int(self.reserved[1])        # Phase-4 help waits that were given up (paired_dev.h); 0 in a healthy run
        # reserved[2]: lists published << 32 | speculative answers the ordered walks took; after a watchdog event (top nibble set) what it saw
        r2 = int(self.reserved[2])
        d["help_watchdog_last"] = r2 if (r2 >> 60) else 0
        d["help_lists_published"] = 0 if (r2 >> 60) else (r2 >> 32)
        d["help_answers_used"] = 0 if (r2 >> 60) else (r2 & 0xffffffff)
        return d


def default_params(max_k: int = 27, max_read_len: int = 400, **overrides) -> Params:
    """The reference's single-end defaults, AlignerOptions.cpp:39-117."""
    p = Params(max_hits=300, max_k=max_k, num_seeds=25, seed_coverage=0.0, min_weight_to_check=1,
               extra_search_depth=1, use_affine_gap=1, match_reward=1, sub_penalty=4,
               gap_open_penalty=6, gap_extend_penalty=1, five_prime_end_bonus=10,
               three_prime_end_bonus=7, alt_awareness=1, emit_alt_alignments=0,
               max_score_gap_to_prefer_non_alt=64, max_read_len=max_read_len)
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def ptr(a: np.ndarray):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
