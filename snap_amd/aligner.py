"""Host-side mirror of the reference's interface for the hot path, over the C ABI.

Class / method names follow the reference (SNAPLib/BaseAligner.h:44-90, GenomeIndex.h:33-80,
LandauVishkin.h:100): `BaseAligner.AlignRead` here takes a *batch* of reads because a GPU
aligner is fed batches, but argument meaning, result fields (SingleAlignmentResult) and error
behaviour follow the reference.  All computation happens in libsnapgpu.so (HIP, gfx950);
there is no CPU path -- if the library or a GPU is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .abi import (Counters, IndexView, Params, RESULT_DTYPE, default_params, ptr)
from .index import GENOME_PAD, GenomeIndex

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsnapgpu.so")      # the product opens this file and nothing else

_lib = None


class SnapGpuError(RuntimeError):
    pass


def load_library():
    """dlopen libsnapgpu.so (built by __graft_entry__.build()); fails loudly if missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SnapGpuError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.snapgpu_last_error.restype = C.c_char_p
        lib.snapgpu_last_error.argtypes = [C.c_void_p]
        lib.snapgpu_create.argtypes = [C.POINTER(IndexView), C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
        lib.snapgpu_destroy.argtypes = [C.c_void_p]
        lib.snapgpu_destroy.restype = None
        lib.snapgpu_align_single_device.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
        _lib = lib
    return _lib


EXPORTED_SYMBOLS = [
    "snapgpu_abi_version", "snapgpu_last_error", "snapgpu_default_params", "snapgpu_create",
    "snapgpu_destroy", "snapgpu_create_from_directory", "snapgpu_device_count", "snapgpu_create_replica", "snapgpu_broadcast_index", "snapgpu_default_paired_params", "snapgpu_enable_paired",
    "snapgpu_align_paired", "snapgpu_align_paired_device", "snapgpu_index_device_ptrs", "snapgpu_lookup_seeds",
    "snapgpu_lookup_seeds_device", "snapgpu_landau_vishkin", "snapgpu_affine_gap", "snapgpu_align_single",
    "snapgpu_align_single_device", "snapgpu_get_counters", "snapgpu_kernel_time",
    "snapgpu_enable_secondary", "snapgpu_align_single_secondary", "snapgpu_align_single_secondary_device",
    "snapgpu_align_paired_secondary", "snapgpu_align_paired_secondary_device",
    "snapgpu_compute_cigar_lv", "snapgpu_compute_cigar_ag", "snapgpu_adjust_alignments", "snapgpu_set_aligner_flags", "snapgpu_affine_gap_sequence", "snapgpu_sam_fields_single", "snapgpu_sam_fields_single_device", "snapgpu_sam_fields_paired", "snapgpu_align_sam_single", "snapgpu_create_replica_with_params",
    "snapgpu_default_index_build_params", "snapgpu_index_build", "snapgpu_index_build_from_fasta", "snapgpu_built_index_view",
    "snapgpu_built_index_save", "snapgpu_built_index_stats", "snapgpu_built_index_destroy",
]


def _pack(strings):
    lens = np.array([len(s) for s in strings], dtype=np.int32)
    offs = np.zeros(len(strings), dtype=np.uint32)
    if len(strings):
        offs[1:] = np.cumsum(lens[:-1])
    buf = np.frombuffer(b"".join(strings) + b"\0" * 16, dtype=np.uint8).copy()
    return buf, offs, lens


def make_index_view(index: GenomeIndex, device_index_ptrs=None):
    """snapgpu_index_view over the arrays of a loaded index; returns (view, arrays to keep alive)."""
    keep = []
    v = IndexView()
    v.seed_len = index.seed_len
    v.key_bytes = index.key_bytes
    v.n_hash_tables = index.n_hash_tables
    v.large_hash_table = 1 if index.large else 0
    v.location_size = index.location_size
    v.chromosome_padding = index.chromosome_padding
    sizes = getattr(index, "_device_sizes", None)      # set by dist.broadcast_index on every rank
    v.overflow_table_size = sizes[1] if sizes else index.overflow.size
    v.hash_blob_bytes = sizes[0] if sizes else index.hash_blob.size
    toff = np.ascontiguousarray(index.table_offset, dtype=np.uint64)
    tsz = np.ascontiguousarray(index.table_size, dtype=np.uint64)
    cb = np.ascontiguousarray(index.contig_begin, dtype=np.uint64)
    keep += [toff, tsz, cb]
    v.table_offset = toff.ctypes.data
    v.table_size = tsz.ctypes.data
    v.contig_begin = cb.ctypes.data
    v.n_contigs = len(index.contigs)
    v.n_bases = index.n_bases
    v.genome_pad = GENOME_PAD
    v.first_alt_location = index.first_alt_location
    pb, prc, cst, cops = index.projection_arrays()
    keep += [pb, prc, cst, cops]
    v.contig_proj_begin = pb.ctypes.data
    v.contig_proj_rc = prc.ctypes.data
    v.contig_cigar_start = cst.ctypes.data
    v.cigar_ops = cops.ctypes.data
    if device_index_ptrs is None:
        v.hash_blob = index.hash_blob.ctypes.data
        v.overflow = index.overflow.ctypes.data
        v.genome = index.genome_padded.ctypes.data + GENOME_PAD
        v.on_device = 0
    else:                                   # (hash, overflow, genome_padded) device addresses
        v.hash_blob, v.overflow = int(device_index_ptrs[0]), int(device_index_ptrs[1])
        v.genome = int(device_index_ptrs[2]) + GENOME_PAD
        v.on_device = 1
    return v, keep


class BaseAligner:
    """One aligner context on one GPU (the analogue of one BaseAligner per CPU thread)."""

    def __init__(self, index: GenomeIndex, params: Params | None = None, device: int = 0,
                 device_index_ptrs=None):
        self.lib = load_library()
        self.index = index
        self.params = params if params is not None else default_params()
        self._keep = []
        v, keep = make_index_view(index, device_index_ptrs)
        self._keep += keep
        handle = C.c_void_p()
        rc = self.lib.snapgpu_create(C.byref(v), C.byref(self.params), device, C.byref(handle))
        if rc != 0:
            raise SnapGpuError("snapgpu_create failed (%d): %s" % (rc, self.lib.snapgpu_last_error(None).decode()))
        self.handle = handle
        self.device = device

    @classmethod
    def from_built_index(cls, built, index: GenomeIndex | None = None, params: Params | None = None, device: int = 0, paired_params=None):
        """A context over an index that snapgpu_index_build* left in HBM (snap_amd.index.BuiltIndex): the view is adopted as it is
        (on_device = 1), nothing is copied and no file is read.  `built` must outlive the aligner."""
        self = cls.__new__(cls)
        self.lib = load_library()
        self.index = index
        self.params = params if params is not None else default_params()
        self._keep = [built]
        v = built.view()
        handle = C.c_void_p()
        rc = self.lib.snapgpu_create(C.byref(v), C.byref(self.params), device, C.byref(handle))
        if rc != 0:
            raise SnapGpuError("snapgpu_create over a built index failed (%d): %s" % (rc, self.lib.snapgpu_last_error(None).decode()))
        self.handle = handle
        self.device = device
        self._after_built(paired_params)
        return self

    @classmethod
    def from_directory(cls, directory: str, params: Params | None = None, device: int = 0):
        """snapgpu_create_from_directory: the library's own loader of the reference's four index files (what the shim and snapgpu-sam use)."""
        self = cls.__new__(cls)
        self.lib = load_library()
        self.index = None
        self.params = params if params is not None else default_params()
        self._keep = []
        self.lib.snapgpu_create_from_directory.argtypes = [C.c_char_p, C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
        handle = C.c_void_p()
        rc = self.lib.snapgpu_create_from_directory(directory.encode(), C.byref(self.params), device, C.byref(handle))
        if rc != 0:
            raise SnapGpuError("snapgpu_create_from_directory failed (%d): %s" % (rc, self.lib.snapgpu_last_error(None).decode()))
        self.handle = handle
        self.device = device
        return self

    def _after_built(self, paired_params):
        pass

    def replica(self, device: int | None = None, share_index: bool = True, params: Params | None = None):
        """Another context over the same index (include/snapgpu.h: snapgpu_create_replica): on this GPU sharing the resident blobs -- a
        second feeder, so that two batches can be in flight -- or on another GPU with blobs of its own (to be filled by
        snapgpu_broadcast_index).  The analogue of the reference's one-aligner-per-thread over one shared GenomeIndex
        (SNAPLib/AlignerContext.cpp: runTask).  `params`: options of its own for the new context (snapgpu_create_replica_with_params),
        e.g. another -d over the one resident index."""
        self.lib.snapgpu_create_replica_with_params.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        dev = self.device if device is None else device
        rc = self.lib.snapgpu_create_replica_with_params(self.handle, C.c_int(dev), C.c_int(1 if share_index else 0),
                                                         C.byref(params) if params is not None else None, C.byref(h))
        if rc != 0:
            raise SnapGpuError("snapgpu_create_replica failed (%d): %s" % (rc, self.lib.snapgpu_last_error(self.handle).decode()))
        other = type(self).__new__(type(self))
        other.__dict__.update(self.__dict__)
        other.handle = h
        other.device = dev
        if params is not None:
            other.params = params
        other._after_replica()
        return other

    def _after_replica(self):
        pass

    def close(self):
        if getattr(self, "handle", None):
            self.lib.snapgpu_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise SnapGpuError("%s failed (%d): %s" % (what, rc, self.lib.snapgpu_last_error(self.handle).decode()))

    # ---- GenomeIndex::lookupSeed32 ------------------------------------------------------
    def lookupSeed32(self, seeds: np.ndarray, max_hits_out: int = 512):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint8)
        n = seeds.shape[0]
        n_hits = np.zeros((n, 2), dtype=np.int64)
        hits = np.zeros((n, 2, max_hits_out), dtype=np.uint32)
        self._check(self.lib.snapgpu_lookup_seeds(self.handle, C.c_uint32(n), ptr(seeds), ptr(n_hits), ptr(hits),
                                                  C.c_uint32(max_hits_out)), "snapgpu_lookup_seeds")
        return n_hits, hits

    def lookup_device(self, n: int, d_seeds: int, d_n_hits: int, d_hits: int = 0, max_hits_out: int = 300, stream: int = 0):
        """GenomeIndex::lookupSeed32 for n seeds already in HBM (device pointers); d_hits = 0: hit counts only (lists read, not stored).
        Timed and counted like the align kernels: the stand-alone index-probe kernel."""
        self._check(self.lib.snapgpu_lookup_seeds_device(self.handle, C.c_uint32(n), C.c_void_p(d_seeds), C.c_void_p(d_n_hits),
                                                         C.c_void_p(d_hits) if d_hits else None, C.c_uint32(max_hits_out),
                                                         C.c_void_p(stream) if stream else None), "snapgpu_lookup_seeds_device")

    # ---- LandauVishkin<dir>::computeEditDistance ---------------------------------------
    def computeEditDistance(self, direction: int, texts, patterns, quals, k):
        n = len(texts)
        tbuf, toff, tlen = _pack(texts)
        if direction == -1:
            toff = (toff + tlen.astype(np.uint32)).astype(np.uint32)
        pbuf, poff, plen = _pack(patterns)
        qbuf, _, _ = _pack(quals)
        k = np.ascontiguousarray(k, dtype=np.int32)
        score = np.zeros(n, np.int32); prob = np.zeros(n, np.float64)
        net = np.zeros(n, np.int32); tot = np.zeros(n, np.int32); span = np.zeros(n, np.int32)
        self._check(self.lib.snapgpu_landau_vishkin(
            self.handle, C.c_int(direction), C.c_uint32(n), ptr(tbuf), C.c_uint64(tbuf.size), ptr(toff), ptr(tlen),
            ptr(pbuf), ptr(qbuf), C.c_uint64(pbuf.size), ptr(poff), ptr(plen), ptr(k),
            ptr(score), ptr(prob), ptr(net), ptr(tot), ptr(span)), "snapgpu_landau_vishkin")
        return dict(score=score, match_probability=prob, net_indel=net, total_indels=tot, text_span=span)

    # ---- AffineGapVectorized<dir>::computeScore[Banded] --------------------------------
    def computeScoreAffine(self, direction: int, texts, patterns, quals, w, score_init, is_rc, banded, use_clip=None, sequence: bool = False):
        """sequence=True: the problems are calls in order on ONE newly constructed object (snapgpu_affine_gap_sequence); the result then
        also has stale_steps."""
        n = len(texts)
        tbuf, toff, tlen = _pack(texts)
        if direction == -1:
            toff = (toff + tlen.astype(np.uint32)).astype(np.uint32)
        pbuf, poff, plen = _pack(patterns)
        qbuf, _, _ = _pack(quals)
        w = np.ascontiguousarray(w, dtype=np.int32)
        score_init = np.ascontiguousarray(score_init, dtype=np.int32)
        is_rc = np.ascontiguousarray(is_rc, dtype=np.uint8)
        banded = np.ascontiguousarray(banded, dtype=np.uint8)
        use_clip = np.zeros(n, np.uint8) if use_clip is None else np.ascontiguousarray(use_clip, dtype=np.uint8)
        ag = np.zeros(n, np.int32); to = np.zeros(n, np.int32); po = np.zeros(n, np.int32)
        ne = np.zeros(n, np.int32); prob = np.zeros(n, np.float64)
        if sequence:
            stale = np.zeros(n, np.int32)
            self._check(self.lib.snapgpu_affine_gap_sequence(
                self.handle, C.c_int(direction), C.c_uint32(n), ptr(tbuf), C.c_uint64(tbuf.size), ptr(toff), ptr(tlen),
                ptr(pbuf), ptr(qbuf), C.c_uint64(pbuf.size), ptr(poff), ptr(plen), ptr(w), ptr(score_init),
                ptr(is_rc), ptr(banded), ptr(use_clip), ptr(ag), ptr(to), ptr(po), ptr(ne), ptr(prob), ptr(stale)), "snapgpu_affine_gap_sequence")
            return dict(ag_score=ag, text_offset=to, pattern_offset=po, n_edits=ne, match_probability=prob, stale_steps=stale)
        self._check(self.lib.snapgpu_affine_gap(
            self.handle, C.c_int(direction), C.c_uint32(n), ptr(tbuf), C.c_uint64(tbuf.size), ptr(toff), ptr(tlen),
            ptr(pbuf), ptr(qbuf), C.c_uint64(pbuf.size), ptr(poff), ptr(plen), ptr(w), ptr(score_init),
            ptr(is_rc), ptr(banded), ptr(use_clip), ptr(ag), ptr(to), ptr(po), ptr(ne), ptr(prob)), "snapgpu_affine_gap")
        return dict(ag_score=ag, text_offset=to, pattern_offset=po, n_edits=ne, match_probability=prob)

    # ---- SAMFormat::computeCigar (Landau-Vishkin variant) over a batch ------------------
    def computeCigar(self, data, off, length, loc, extra_before, use_m: bool = False, ops_stride: int = 64):
        """SAM.cpp:2354-2467 for a batch of written reads (see snapgpu_compute_cigar_lv in include/snapgpu.h): data = the clipped
        reads in reference orientation (one uint8 buffer), item i = data[off[i] : off[i] + length[i]] at genome location loc[i].
        Returns dict(ops uint32[n, ops_stride], n_ops, edit_distance, add_front_clipping, extra_clipped_after)."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(off, dtype=np.uint64); length = np.ascontiguousarray(length, dtype=np.int32)
        loc = np.ascontiguousarray(loc, dtype=np.int64); extra_before = np.ascontiguousarray(extra_before, dtype=np.int32)
        n = off.size
        ops = np.zeros((n, ops_stride), dtype=np.uint32); n_ops = np.zeros(n, np.int32); ed = np.zeros(n, np.int32)
        afc = np.zeros(n, np.int32); after = np.zeros(n, np.int64)
        self._check(self.lib.snapgpu_compute_cigar_lv(
            self.handle, C.c_uint32(n), ptr(data), C.c_uint64(data.size), ptr(off), ptr(length), ptr(loc), ptr(extra_before),
            C.c_int(1 if use_m else 0), ptr(ops), C.c_uint32(ops_stride), ptr(n_ops), ptr(ed), ptr(afc), ptr(after)),
            "snapgpu_compute_cigar_lv")
        return dict(ops=ops, n_ops=n_ops, edit_distance=ed, add_front_clipping=afc, extra_clipped_after=after)

    def AdjustAlignments(self, data, off, length, results):
        """AlignmentAdjuster::AdjustAlignment (AlignmentAdjuster.cpp:33-190) for a batch (snapgpu_adjust_alignments): read i =
        data[off[i] : off[i] + length[i]] as given to AlignRead; results (RESULT_DTYPE) carry status / direction / location / score in and
        come back with status / location / score / clipping_for_read_adjustment adjusted.  Returns the adjusted copy."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(off, dtype=np.uint64); length = np.ascontiguousarray(length, dtype=np.int32)
        out = np.ascontiguousarray(results, dtype=RESULT_DTYPE).copy()
        self._check(self.lib.snapgpu_adjust_alignments(self.handle, C.c_uint32(off.size), ptr(data), C.c_uint64(data.size), ptr(off), ptr(length), ptr(out)),
                    "snapgpu_adjust_alignments")
        return out

    def computeCigarAffineGap(self, data, quals, off, length, loc, extra_before, score, use_m: bool = False, ops_stride: int = 64):
        """SAM.cpp:2470-2588 (the CIGAR of a read scored with affine gap) for a batch; see snapgpu_compute_cigar_ag.  score = the
        alignment's edit distance.  Returns the fields of computeCigar plus back_clipping_missed and stale (the reference's answer
        for that item depends on what its object computed before)."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(off, dtype=np.uint64); length = np.ascontiguousarray(length, dtype=np.int32)
        loc = np.ascontiguousarray(loc, dtype=np.int64); extra_before = np.ascontiguousarray(extra_before, dtype=np.int32)
        score = np.ascontiguousarray(score, dtype=np.int32)
        n = off.size
        ops = np.zeros((n, ops_stride), dtype=np.uint32); n_ops = np.zeros(n, np.int32); ed = np.zeros(n, np.int32)
        afc = np.zeros(n, np.int32); after = np.zeros(n, np.int64); tail = np.zeros(n, np.int32); stale = np.zeros(n, np.int32)
        self._check(self.lib.snapgpu_compute_cigar_ag(
            self.handle, C.c_uint32(n), ptr(data), ptr(quals), C.c_uint64(data.size), ptr(off), ptr(length), ptr(loc), ptr(extra_before),
            ptr(score), C.c_int(1 if use_m else 0), ptr(ops), C.c_uint32(ops_stride), ptr(n_ops), ptr(ed), ptr(afc), ptr(after),
            ptr(tail), ptr(stale)), "snapgpu_compute_cigar_ag")
        return dict(ops=ops, n_ops=n_ops, edit_distance=ed, add_front_clipping=afc, extra_clipped_after=after,
                    back_clipping_missed=tail, stale=stale)

    def samFields(self, bases, quals, offsets, front_clip, data_len, results, use_m: bool = False, ops_stride: int = 64):
        """SimpleReadWriter::writeReads + SAMFormat::writeRead up to the point of printing, for the primary results of a batch of
        single-end reads (see snapgpu_sam_fields_single).  bases / quals / offsets: the unclipped reads; front_clip / data_len:
        Read::clip's outcome; results: RESULT_DTYPE array.  Returns dict(flag, contig, pos, mapq, ops, n_ops, nm, stale)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1); quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        front_clip = np.ascontiguousarray(front_clip, dtype=np.int32); data_len = np.ascontiguousarray(data_len, dtype=np.int32)
        results = np.ascontiguousarray(results, dtype=RESULT_DTYPE)
        n = offsets.size - 1
        flag = np.zeros(n, np.int32); contig = np.zeros(n, np.int32); pos = np.zeros(n, np.int64); mapq = np.zeros(n, np.int32)
        ops = np.zeros((n, ops_stride), dtype=np.uint32); n_ops = np.zeros(n, np.int32); nm = np.zeros(n, np.int32); stale = np.zeros(n, np.int32)
        self._check(self.lib.snapgpu_sam_fields_single(
            self.handle, C.c_uint32(n), ptr(bases), ptr(quals), ptr(offsets), ptr(front_clip), ptr(data_len), ptr(results),
            C.c_int(1 if use_m else 0), ptr(flag), ptr(contig), ptr(pos), ptr(mapq), ptr(ops), C.c_uint32(ops_stride), ptr(n_ops), ptr(nm),
            ptr(stale)), "snapgpu_sam_fields_single")
        return dict(flag=flag, contig=contig, pos=pos, mapq=mapq, ops=ops, n_ops=n_ops, nm=nm, stale=stale)

    def alignSam(self, bases, quals, offsets, front_clip, data_len, skip, use_m: bool = False, ops_stride: int = 64):
        """The single-end path of a SAM writer in one call (snapgpu_align_sam_single): BaseAligner::AlignRead over the clipped reads
        (SingleAligner.cpp:250) and SimpleReadWriter::writeReads' computed fields for each primary result, the batch uploaded once.
        bases / quals / offsets: the unclipped reads; front_clip / data_len: Read::clip's outcome; skip[i] != 0: the read is not given to
        the aligner (SingleAligner.cpp:211-232).  Returns (results, first_alt, dict(flag, contig, pos, mapq, ops, n_ops, nm, stale))."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1); quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        front_clip = np.ascontiguousarray(front_clip, dtype=np.int32); data_len = np.ascontiguousarray(data_len, dtype=np.int32)
        skip = np.ascontiguousarray(skip, dtype=np.uint8)
        n = offsets.size - 1
        results = np.zeros(n, dtype=RESULT_DTYPE); first_alt = np.zeros(n, dtype=RESULT_DTYPE)
        flag = np.zeros(n, np.int32); contig = np.zeros(n, np.int32); pos = np.zeros(n, np.int64); mapq = np.zeros(n, np.int32)
        ops = np.zeros((n, ops_stride), dtype=np.uint32); n_ops = np.zeros(n, np.int32); nm = np.zeros(n, np.int32); stale = np.zeros(n, np.int32)
        self._check(self.lib.snapgpu_align_sam_single(
            self.handle, C.c_uint32(n), ptr(bases), ptr(quals), ptr(offsets), ptr(front_clip), ptr(data_len), ptr(skip),
            C.c_int(1 if use_m else 0), ptr(results), ptr(first_alt), ptr(flag), ptr(contig), ptr(pos), ptr(mapq), ptr(ops),
            C.c_uint32(ops_stride), ptr(n_ops), ptr(nm), ptr(stale)), "snapgpu_align_sam_single")
        return results, first_alt, dict(flag=flag, contig=contig, pos=pos, mapq=mapq, ops=ops, n_ops=n_ops, nm=nm, stale=stale)

    def samFieldsPaired(self, bases, quals, offsets, front_clip, data_len, results, use_m: bool = False, ops_stride: int = 64):
        """SAMFormat::writePairs + fillMateInfo up to the point of printing (see snapgpu_sam_fields_paired).  offsets has 2 n + 1
        entries (read 0 and read 1 of each pair); results: PAIRED_RESULT_DTYPE array of n."""
        from .abi import PAIRED_RESULT_DTYPE
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1); quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        front_clip = np.ascontiguousarray(front_clip, dtype=np.int32); data_len = np.ascontiguousarray(data_len, dtype=np.int32)
        results = np.ascontiguousarray(results, dtype=PAIRED_RESULT_DTYPE)
        n = offsets.size - 1; npairs = n // 2
        flag = np.zeros(n, np.int32); contig = np.zeros(n, np.int32); pos = np.zeros(n, np.int64); mapq = np.zeros(n, np.int32)
        ops = np.zeros((n, ops_stride), dtype=np.uint32); n_ops = np.zeros(n, np.int32); nm = np.zeros(n, np.int32); stale = np.zeros(n, np.int32)
        rnext = np.zeros(n, np.int32); pnext = np.zeros(n, np.int64); tlen = np.zeros(n, np.int64); first = np.zeros(npairs, np.int32)
        self._check(self.lib.snapgpu_sam_fields_paired(
            self.handle, C.c_uint32(npairs), ptr(bases), ptr(quals), ptr(offsets), ptr(front_clip), ptr(data_len), ptr(results),
            C.c_int(1 if use_m else 0), ptr(flag), ptr(contig), ptr(pos), ptr(mapq), ptr(ops), C.c_uint32(ops_stride), ptr(n_ops), ptr(nm),
            ptr(rnext), ptr(pnext), ptr(tlen), ptr(first), ptr(stale)), "snapgpu_sam_fields_paired")
        return dict(flag=flag, contig=contig, pos=pos, mapq=mapq, ops=ops, n_ops=n_ops, nm=nm, rnext=rnext, pnext=pnext, tlen=tlen,
                    first_written=first, stale=stale)

    # ---- BaseAligner::AlignRead over a batch -------------------------------------------
    def AlignRead(self, bases: np.ndarray, quals: np.ndarray, offsets: np.ndarray):
        """Returns (primaryResult[n], firstALTResult[n]) as RESULT_DTYPE arrays."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1)
        quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        primary = np.zeros(n, dtype=RESULT_DTYPE)
        first_alt = np.zeros(n, dtype=RESULT_DTYPE)
        self._check(self.lib.snapgpu_align_single(self.handle, C.c_uint32(n), ptr(bases), ptr(quals), ptr(offsets),
                                                  ptr(primary), ptr(first_alt)), "snapgpu_align_single")
        return primary, first_alt

    # ---- BaseAligner::AlignRead with secondary results (-om / -omax / -mpc) --------------
    def set_flags(self, stop_on_first_hit: bool = False, explore_popular_seeds: bool = False):
        """-f / -x: BaseAligner::setStopOnFirstHit / setExplorePopularSeeds (SingleAligner.cpp:179-180) for this context's single-end calls."""
        self.lib.snapgpu_set_aligner_flags.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self._check(self.lib.snapgpu_set_aligner_flags(self.handle, C.c_int(1 if stop_on_first_hit else 0), C.c_int(1 if explore_popular_seeds else 0)),
                    "snapgpu_set_aligner_flags")

    def enable_secondary(self, max_edit_distance: int, max_results: int = 0x7fffffff, max_per_contig: int = -1,
                         adjust_alignments: int = 0):
        from .abi import secondary_params
        sp = secondary_params(max_edit_distance, max_results, max_per_contig, adjust_alignments)
        self._check(self.lib.snapgpu_enable_secondary(self.handle, C.byref(sp)), "snapgpu_enable_secondary")
        self._sec_max = max_results

    def AlignReadSecondary(self, bases: np.ndarray, quals: np.ndarray, offsets: np.ndarray, stride: int = 16):
        """Returns (primary[n], firstALT[n], secondary[n, stride'], nSecondary[n]); like the reference's caller
        (SingleAligner.cpp:250-263) it grows the buffer and calls again when a read has more results than fit."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1)
        quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        primary = np.zeros(n, dtype=RESULT_DTYPE)
        first_alt = np.zeros(n, dtype=RESULT_DTYPE)
        while True:
            secondary = np.zeros((n, stride), dtype=RESULT_DTYPE)
            n_sec = np.zeros(n, dtype=np.uint32)
            rc = self.lib.snapgpu_align_single_secondary(self.handle, C.c_uint32(n), ptr(bases), ptr(quals), ptr(offsets),
                                                         ptr(primary), ptr(first_alt), ptr(secondary), C.c_uint32(stride), ptr(n_sec))
            if rc == 1:                                   # SNAPGPU_W_SECONDARY_TRUNCATED
                stride = int(n_sec.max())
                continue
            self._check(rc, "snapgpu_align_single_secondary")
            return primary, first_alt, secondary, n_sec

    def align_device(self, n: int, d_bases: int, d_quals: int, d_offsets: int, d_primary: int, d_first_alt: int = 0,
                     stream: int = 0):
        """Device-pointer form (inputs already in HBM, results left in HBM)."""
        self._check(self.lib.snapgpu_align_single_device(self.handle, C.c_uint32(n), C.c_void_p(d_bases),
                                                         C.c_void_p(d_quals), C.c_void_p(d_offsets), C.c_void_p(d_primary),
                                                         C.c_void_p(d_first_alt) if d_first_alt else None,
                                                         C.c_void_p(stream) if stream else None),
                    "snapgpu_align_single_device")

    def counters(self, reset: bool = False) -> dict:
        c = Counters()
        self._check(self.lib.snapgpu_get_counters(self.handle, C.byref(c), C.c_int(1 if reset else 0)), "snapgpu_get_counters")
        return c.as_dict()

    def launch_profile(self):
        """Diagnostics of the last single-end launch of a context created under SNAPGPU_PHASE_TIMERS=1: dict(read_cycles_log2_hist[64],
        wave_start[S], wave_finish[S], wave_worst_read_cycles[S], wave_worst_read_ag_calls[S])."""
        cap = 64 + 3 * 65536
        buf = np.zeros(cap, dtype=np.uint64)
        ns = C.c_uint32(0)
        self.lib.snapgpu_debug_launch_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]
        self._check(self.lib.snapgpu_debug_launch_profile(self.handle, buf.ctypes.data, C.c_uint64(cap), C.byref(ns)), "snapgpu_debug_launch_profile")
        S = int(ns.value)
        w = buf[64 + 2 * S:64 + 3 * S]
        return dict(read_cycles_log2_hist=buf[:64].copy(), wave_start=buf[64:64 + S].copy(), wave_finish=buf[64 + S:64 + 2 * S].copy(),
                    wave_worst_read_cycles=(w >> np.uint64(24)).copy(), wave_worst_read_ag_calls=(w & np.uint64(0xffffff)).copy())

    def kernel_time(self, reset: bool = False):
        ms = C.c_double(0); nl = C.c_uint64(0)
        self._check(self.lib.snapgpu_kernel_time(self.handle, C.byref(ms), C.byref(nl), C.c_int(1 if reset else 0)),
                    "snapgpu_kernel_time")
        return ms.value, int(nl.value)

    def device_index_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self.lib.snapgpu_index_device_ptrs(self.handle, C.byref(a), C.byref(b), C.byref(c)),
                    "snapgpu_index_device_ptrs")
        return a.value, b.value, c.value

    def debug_tables(self):
        phred = np.zeros(256); indel = np.zeros(1001); perfect = np.zeros(1001)
        sp = C.c_double(0); thr = np.zeros(72); wrapped = np.zeros(33, dtype=np.uint32)
        self._check(self.lib.snapgpu_debug_tables(self.handle, ptr(phred), ptr(indel), C.c_uint32(1001), ptr(perfect),
                                                  C.c_uint32(1001), C.byref(sp), ptr(thr), ptr(wrapped)), "snapgpu_debug_tables")
        return dict(phred=phred, indel=indel, perfect=perfect, seed_prob=sp.value, mapq_threshold=thr, wrapped=wrapped)


class ChimericPairedEndAligner(BaseAligner):
    """Paired-end host mirror: ChimericPairedEndAligner over IntersectingPairedEndAligner
    (SNAPLib/ChimericPairedEndAligner.cpp:126, SNAPLib/IntersectingPairedEndAligner.cpp:169), built the way
    PairedAlignerContext builds them (SNAPLib/PairedAligner.cpp:556-625).  `params` are the options shared with the
    single-end aligner (maxHits, maxDist, affine-gap scores ...), `paired_params` the PairedAlignerOptions."""

    def __init__(self, index: GenomeIndex, params: Params | None = None, paired_params=None, device: int = 0,
                 device_index_ptrs=None):
        from .abi import PairedParams, default_paired_params
        super().__init__(index, params, device, device_index_ptrs)
        self.paired_params = paired_params if paired_params is not None else default_paired_params()
        self.lib.snapgpu_enable_paired.argtypes = [C.c_void_p, C.POINTER(PairedParams)]
        self.lib.snapgpu_align_paired_device.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
        self._check(self.lib.snapgpu_enable_paired(self.handle, C.byref(self.paired_params)), "snapgpu_enable_paired")

    def _after_replica(self):               # (a replica starts as a single-end context)
        self._check(self.lib.snapgpu_enable_paired(self.handle, C.byref(self.paired_params)), "snapgpu_enable_paired")

    @classmethod
    def over(cls, base: "BaseAligner", paired_params=None, params: Params | None = None):
        """A paired-end context over the index another context (single-end or paired) already has resident on its GPU:
        snapgpu_create_replica(share_index = 1) + snapgpu_enable_paired.  `base` must outlive it (it owns the blobs)."""
        from .abi import PairedParams, default_paired_params
        proto = cls.__new__(cls)
        proto.__dict__.update(base.__dict__)
        proto.paired_params = paired_params if paired_params is not None else default_paired_params()
        proto.lib.snapgpu_enable_paired.argtypes = [C.c_void_p, C.POINTER(PairedParams)]
        proto.lib.snapgpu_align_paired_device.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
        try:
            other = BaseAligner.replica(proto, params=params)
        finally:
            proto.handle = None             # (the prototype borrowed base's handle: it must never destroy it, whatever the replica call did)
        return other

    def _after_built(self, paired_params):
        from .abi import PairedParams, default_paired_params
        self.paired_params = paired_params if paired_params is not None else default_paired_params()
        self.lib.snapgpu_enable_paired.argtypes = [C.c_void_p, C.POINTER(PairedParams)]
        self.lib.snapgpu_align_paired_device.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
        self._check(self.lib.snapgpu_enable_paired(self.handle, C.byref(self.paired_params)), "snapgpu_enable_paired")

    def align(self, bases: np.ndarray, quals: np.ndarray, offsets: np.ndarray):
        """ChimericPairedEndAligner::align over a batch: offsets has 2n+1 entries (read 0 / read 1 of each pair
        interleaved).  Returns (result[n], firstALTResult[n]) as PAIRED_RESULT_DTYPE arrays."""
        from .abi import PAIRED_RESULT_DTYPE
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1)
        quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if (offsets.size - 1) % 2:
            raise ValueError("offsets must have 2*n_pairs + 1 entries")
        n = (offsets.size - 1) // 2
        primary = np.zeros(n, dtype=PAIRED_RESULT_DTYPE)
        first_alt = np.zeros(n, dtype=PAIRED_RESULT_DTYPE)
        self._check(self.lib.snapgpu_align_paired(self.handle, C.c_uint32(n), ptr(bases), ptr(quals), ptr(offsets),
                                                  ptr(primary), ptr(first_alt)), "snapgpu_align_paired")
        return primary, first_alt

    def align_secondary(self, bases: np.ndarray, quals: np.ndarray, offsets: np.ndarray, stride: int = 8, single_stride: int = 16):
        """ChimericPairedEndAligner::align with secondary results (after enable_secondary(...)): returns
        (result[n], firstALT[n], secondary[n, stride'], nSecondary[n], singleSecondary[n, single_stride'], nSingleSecondary[n, 2]).
        Like PairedAligner.cpp:727-779 it grows a buffer and calls again when a pair has more results than fit."""
        from .abi import PAIRED_RESULT_DTYPE
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1)
        quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if (offsets.size - 1) % 2:
            raise ValueError("offsets must have 2*n_pairs + 1 entries")
        n = (offsets.size - 1) // 2
        primary = np.zeros(n, dtype=PAIRED_RESULT_DTYPE)
        first_alt = np.zeros(n, dtype=PAIRED_RESULT_DTYPE)
        while True:
            sec = np.zeros((n, stride), dtype=PAIRED_RESULT_DTYPE)
            nsec = np.zeros(n, dtype=np.uint32)
            ssec = np.zeros((n, single_stride), dtype=RESULT_DTYPE)
            nssec = np.zeros((n, 2), dtype=np.uint32)
            rc = self.lib.snapgpu_align_paired_secondary(self.handle, C.c_uint32(n), ptr(bases), ptr(quals), ptr(offsets), ptr(primary),
                                                         ptr(first_alt), ptr(sec), C.c_uint32(stride), ptr(nsec),
                                                         ptr(ssec), C.c_uint32(single_stride), ptr(nssec))
            if rc == 1:                                   # SNAPGPU_W_SECONDARY_TRUNCATED
                stride = max(stride, int(nsec.max()))
                single_stride = max(single_stride, int(nssec.sum(axis=1).max()))
                continue
            self._check(rc, "snapgpu_align_paired_secondary")
            return primary, first_alt, sec, nsec, ssec, nssec

    def align_device(self, n_pairs: int, d_bases: int, d_quals: int, d_offsets: int, d_primary: int, d_first_alt: int = 0,
                     stream: int = 0):
        self._check(self.lib.snapgpu_align_paired_device(self.handle, C.c_uint32(n_pairs), C.c_void_p(d_bases),
                                                         C.c_void_p(d_quals), C.c_void_p(d_offsets), C.c_void_p(d_primary),
                                                         C.c_void_p(d_first_alt) if d_first_alt else None,
                                                         C.c_void_p(stream) if stream else None),
                    "snapgpu_align_paired_device")
