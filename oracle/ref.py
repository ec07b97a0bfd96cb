"""TEST INFRASTRUCTURE ONLY -- ctypes front end for oracle/_ref/libsnapref.so.

libsnapref.so is the *unmodified* reference (SNAP 2.0.5) compiled by oracle/Makefile from
the sources under /root/reference, plus oracle/ref_driver.cpp.  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this module, and only as the
checker / reported baseline.  The product (snap_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from snap_amd.abi import RESULT_DTYPE, Params, ptr

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
LIB_PATH = os.path.join(REF_DIR, "libsnapref.so")
CLI_PATH = os.path.join(REF_DIR, "snap-aligner")


def available() -> bool:
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libsnapref.so missing: run `make -C oracle ref` "
                               "where /root/reference exists")
        _lib = C.CDLL(LIB_PATH)
        _lib.snapref_load_index.restype = C.c_void_p
        _lib.snapref_load_index.argtypes = [C.c_char_p]
        _lib.snapref_init()
    return _lib


class adjust_alignments:
    """`with ref.adjust_alignments():` -- the reference aligners are constructed with ignoreAlignmentAdjustmentsForOm = false, i.e. `-ae`
    (AlignerOptions.cpp:476): AlignmentAdjuster::AdjustAlignment runs on the primary and on every secondary result before the -om filter
    (BaseAligner.cpp:2444-2463).  Single-end entry points only."""
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = lib().snapref_get_adjust_alignments()
        lib().snapref_set_adjust_alignments(1 if self.on else 0)
        return self

    def __exit__(self, *a):
        lib().snapref_set_adjust_alignments(self.prev)


class aligner_flags:
    """`with ref.aligner_flags(stop_on_first_hit=True, explore_popular_seeds=False):` -- -f / -x for the BaseAligner objects snapref_align_single*
    construct inside the block (BaseAligner::setStopOnFirstHit / setExplorePopularSeeds, as SingleAligner.cpp:179-180 calls them)."""

    def __init__(self, stop_on_first_hit: bool = False, explore_popular_seeds: bool = False):
        self.f, self.x = stop_on_first_hit, explore_popular_seeds

    def __enter__(self):
        lib().snapref_set_aligner_flags(C.c_int(1 if self.f else 0), C.c_int(1 if self.x else 0))
        return self

    def __exit__(self, *exc):
        lib().snapref_set_aligner_flags(C.c_int(0), C.c_int(0))
        return False


class fresh_objects:
    """`with ref.fresh_objects():` -- every read / pair is aligned by reference aligner objects newly constructed in zero-filled memory
    (oracle/ref_driver.cpp: ZeroedArena), so the reference's answer is a function of the read alone and EVERY read can be compared
    (no exclusion of reads whose banded affine-gap traceback walks through cells an earlier read left behind)."""

    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        self.prev = lib().snapref_get_fresh_objects()
        lib().snapref_set_fresh_objects(1 if self.on else 0)
        return self

    def __exit__(self, *a):
        lib().snapref_set_fresh_objects(self.prev)
        return False


def build_index(fasta: str, out_dir: str, seed_len: int = 20, threads: int = 8, large: bool = False,
                extra=()) -> None:
    """`snap-aligner index <fasta> <dir> -s N` with the reference's own builder."""
    cmd = [CLI_PATH, "index", fasta, out_dir, "-s", str(seed_len), "-t%d" % max(1, min(int(threads), 100))]   # GenomeIndex.cpp:213 caps -t at 100
    if large:
        cmd.append("-large")
    cmd += list(extra)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0 or not os.path.exists(os.path.join(out_dir, "GenomeIndex")):
        raise RuntimeError("reference index build failed:\n" + r.stdout.decode(errors="replace")[-2000:])


class RefIndex:
    def __init__(self, directory: str):
        self.handle = C.c_void_p(lib().snapref_load_index(directory.encode()))
        if not self.handle:
            raise RuntimeError("reference failed to load index " + directory)
        self.directory = directory

    def lookup_seeds(self, seeds: np.ndarray, max_hits_out: int = 512):
        """seeds: uint8 [n, seed_len].  Returns (n_hits int64[n,2], hits uint32[n,2,max])."""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint8)
        n = seeds.shape[0]
        n_hits = np.zeros((n, 2), dtype=np.int64)
        hits = np.zeros((n, 2, max_hits_out), dtype=np.uint32)
        rc = lib().snapref_lookup_seeds(self.handle, C.c_uint32(n), ptr(seeds), ptr(n_hits), ptr(hits),
                                        C.c_uint32(max_hits_out))
        if rc != 0:
            raise RuntimeError("snapref_lookup_seeds rc=%d" % rc)
        return n_hits, hits

    def compute_cigar_lv(self, data: np.ndarray, off: np.ndarray, length: np.ndarray, loc: np.ndarray, extra_before: np.ndarray,
                         use_m: bool, ops_stride: int = 64):
        """SAMFormat::computeCigar (Landau-Vishkin variant, BAM_CIGAR_OPS) for a batch; see snapref_compute_cigar_lv."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(off, dtype=np.uint64); length = np.ascontiguousarray(length, dtype=np.int32)
        loc = np.ascontiguousarray(loc, dtype=np.int64); extra_before = np.ascontiguousarray(extra_before, dtype=np.int32)
        n = off.size
        ops = np.zeros((n, ops_stride), dtype=np.uint32); n_ops = np.zeros(n, dtype=np.int32)
        ed = np.zeros(n, dtype=np.int32); afc = np.zeros(n, dtype=np.int32); after = np.zeros(n, dtype=np.int64)
        rc = lib().snapref_compute_cigar_lv(self.handle, C.c_uint32(n), ptr(data), ptr(off), ptr(length), ptr(loc), ptr(extra_before),
                                            C.c_int(1 if use_m else 0), ptr(ops), C.c_uint32(ops_stride), ptr(n_ops), ptr(ed), ptr(afc),
                                            ptr(after))
        if rc != 0:
            raise RuntimeError("snapref_compute_cigar_lv rc=%d" % rc)
        return dict(ops=ops, n_ops=n_ops, edit_distance=ed, add_front_clipping=afc, extra_clipped_after=after)

    def adjust_alignments(self, data, off, length, results):
        """AlignmentAdjuster::AdjustAlignment for a batch; see snapref_adjust_alignments.  Returns the adjusted copy of `results`."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(off, dtype=np.uint64); length = np.ascontiguousarray(length, dtype=np.int32)
        out = np.ascontiguousarray(results, dtype=RESULT_DTYPE).copy()
        rc = lib().snapref_adjust_alignments(self.handle, C.c_uint32(off.size), ptr(data), ptr(off), ptr(length), ptr(out))
        if rc != 0:
            raise RuntimeError("snapref_adjust_alignments rc=%d" % rc)
        return out

    def compute_cigar_ag(self, data, quals, off, length, loc, extra_before, score, use_m: bool, fresh_object: bool = False,
                         ops_stride: int = 64, agparams=(1, 4, 6, 1)):
        """SAMFormat::computeCigar (affine-gap variant, BAM_CIGAR_OPS) for a batch; see snapref_compute_cigar_ag."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1); quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(off, dtype=np.uint64); length = np.ascontiguousarray(length, dtype=np.int32)
        loc = np.ascontiguousarray(loc, dtype=np.int64); extra_before = np.ascontiguousarray(extra_before, dtype=np.int32)
        score = np.ascontiguousarray(score, dtype=np.int32); agp = np.array(agparams, dtype=np.int32)
        n = off.size
        ops = np.zeros((n, ops_stride), dtype=np.uint32); n_ops = np.zeros(n, dtype=np.int32)
        ed = np.zeros(n, dtype=np.int32); afc = np.zeros(n, dtype=np.int32); after = np.zeros(n, dtype=np.int64); tail = np.zeros(n, dtype=np.int32)
        rc = lib().snapref_compute_cigar_ag(self.handle, ptr(agp), C.c_uint32(n), ptr(data), ptr(quals), ptr(off), ptr(length), ptr(loc),
                                            ptr(extra_before), ptr(score), C.c_int(1 if use_m else 0), C.c_int(1 if fresh_object else 0),
                                            ptr(ops), C.c_uint32(ops_stride), ptr(n_ops), ptr(ed), ptr(afc), ptr(after), ptr(tail))
        if rc != 0:
            raise RuntimeError("snapref_compute_cigar_ag rc=%d" % rc)
        return dict(ops=ops, n_ops=n_ops, edit_distance=ed, add_front_clipping=afc, extra_clipped_after=after, back_clipping_missed=tail)

    def align_single(self, params: Params, bases: np.ndarray, quals: np.ndarray, offsets: np.ndarray,
                     threads: int = 1):
        """BaseAligner::AlignRead over a batch; returns (primary, first_alt, counters, seconds)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1)
        quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        primary = np.zeros(n, dtype=RESULT_DTYPE)
        first_alt = np.zeros(n, dtype=RESULT_DTYPE)
        counters = np.zeros(3, dtype=np.int64)
        secs = C.c_double(0)
        rc = lib().snapref_align_single(self.handle, C.byref(params), C.c_uint32(n), ptr(bases), ptr(quals),
                                        ptr(offsets), C.c_int(threads), ptr(primary), ptr(first_alt),
                                        ptr(counters), C.byref(secs))
        if rc != 0:
            raise RuntimeError("snapref_align_single rc=%d" % rc)
        return primary, first_alt, dict(lookups=int(counters[0]), lv=int(counters[1]), ag=int(counters[2])), secs.value


    def align_single_secondary(self, params: Params, om: int, bases: np.ndarray, quals: np.ndarray, offsets: np.ndarray,
                               omax: int = 0x7fffffff, mpc: int = -1, threads: int = 1, stride: int = 64):
        """BaseAligner::AlignRead with -om/-omax/-mpc, called as SingleAligner.cpp:250 calls it.
        Returns (primary, first_alt, secondary[n, stride], n_secondary[n])."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1)
        quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        primary = np.zeros(n, dtype=RESULT_DTYPE)
        first_alt = np.zeros(n, dtype=RESULT_DTYPE)
        while True:
            secondary = np.zeros((n, stride), dtype=RESULT_DTYPE)
            n_sec = np.zeros(n, dtype=np.uint32)
            rc = lib().snapref_align_single_secondary(self.handle, C.byref(params), C.c_int(om), C.c_int64(omax), C.c_int(mpc),
                                                      C.c_uint32(n), ptr(bases), ptr(quals), ptr(offsets), C.c_int(threads),
                                                      ptr(primary), ptr(first_alt), ptr(secondary), C.c_uint32(stride), ptr(n_sec))
            if rc != 0:
                raise RuntimeError("snapref_align_single_secondary rc=%d" % rc)
            if n and int(n_sec.max()) > stride:
                stride = int(n_sec.max())
                continue
            return primary, first_alt, secondary, n_sec

    def align_paired_secondary(self, params: Params, pparams, om: int, bases: np.ndarray, quals: np.ndarray, offsets: np.ndarray,
                               omax: int = 0x7fffffff, mpc: int = -1, threads: int = 1, stage: int = 0, stride: int = 32,
                               single_stride: int = 64):
        """ChimericPairedEndAligner::align (stage 0) / IntersectingPairedEndAligner::align (stage 1) with -om / -omax / -mpc, called
        as PairedAligner.cpp:727 calls it.  Returns (primary, first_alt, secondary[n, stride], n_secondary[n],
        single_secondary[n, single_stride], n_single_secondary[n, 2])."""
        from snap_amd.abi import PAIRED_RESULT_DTYPE
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1)
        quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = (offsets.size - 1) // 2
        primary = np.zeros(n, dtype=PAIRED_RESULT_DTYPE)
        first_alt = np.zeros(n, dtype=PAIRED_RESULT_DTYPE)
        while True:
            sec = np.zeros((n, stride), dtype=PAIRED_RESULT_DTYPE)
            nsec = np.zeros(n, dtype=np.uint32)
            ssec = np.zeros((n, single_stride), dtype=RESULT_DTYPE)
            nssec = np.zeros((n, 2), dtype=np.uint32)
            rc = lib().snapref_align_paired_secondary(self.handle, C.byref(params), C.byref(pparams), C.c_int(stage), C.c_int(om), C.c_int64(omax),
                                                      C.c_int(mpc), C.c_uint32(n), ptr(bases), ptr(quals), ptr(offsets), C.c_int(threads),
                                                      ptr(primary), ptr(first_alt), ptr(sec), C.c_uint32(stride), ptr(nsec),
                                                      ptr(ssec), C.c_uint32(single_stride), ptr(nssec))
            if rc != 0:
                raise RuntimeError("snapref_align_paired_secondary rc=%d" % rc)
            grow = False
            if n and int(nsec.max()) > stride:
                stride = int(nsec.max()); grow = True
            if n and int(nssec.sum(axis=1).max()) > single_stride:
                single_stride = int(nssec.sum(axis=1).max()); grow = True
            if not grow:
                return primary, first_alt, sec, nsec, ssec, nssec

    def align_paired(self, params: Params, pparams, bases: np.ndarray, quals: np.ndarray, offsets: np.ndarray,
                     threads: int = 1, stage: int = 0):
        """ChimericPairedEndAligner::align (stage 0) or IntersectingPairedEndAligner::align only (stage 1)
        over a batch of pairs; offsets has 2n+1 entries (read 0 and read 1 of each pair interleaved).
        Returns (primary, first_alt, counters, seconds)."""
        from snap_amd.abi import PAIRED_RESULT_DTYPE
        bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1)
        quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        assert (offsets.size - 1) % 2 == 0
        n = (offsets.size - 1) // 2
        primary = np.zeros(n, dtype=PAIRED_RESULT_DTYPE)
        first_alt = np.zeros(n, dtype=PAIRED_RESULT_DTYPE)
        counters = np.zeros(2, dtype=np.int64)
        secs = C.c_double(0)
        rc = lib().snapref_align_paired(self.handle, C.byref(params), C.byref(pparams), C.c_int(stage), C.c_uint32(n),
                                        ptr(bases), ptr(quals), ptr(offsets), C.c_int(threads), ptr(primary),
                                        ptr(first_alt), ptr(counters), C.byref(secs))
        if rc != 0:
            raise RuntimeError("snapref_align_paired rc=%d" % rc)
        return primary, first_alt, dict(lv=int(counters[0]), ag=int(counters[1])), secs.value


def _pack(strings):
    """list of bytes -> (uint8 buffer, uint32 offsets, int32 lengths)."""
    lens = np.array([len(s) for s in strings], dtype=np.int32)
    offs = np.zeros(len(strings), dtype=np.uint32)
    if len(strings):
        offs[1:] = np.cumsum(lens[:-1])
    buf = np.frombuffer(b"".join(strings) + b"\0", dtype=np.uint8).copy()
    return buf, offs, lens


def landau_vishkin(direction: int, texts, patterns, quals, k):
    """Batched LandauVishkin<direction>::computeEditDistance on python byte strings.

    For direction -1, texts[i] is given in *memory order*; the reference travels it from its
    last byte backwards (text pointer = one past the end).
    """
    n = len(texts)
    tbuf, toff, tlen = _pack(texts)
    if direction == -1:
        toff = (toff + tlen.astype(np.uint32)).astype(np.uint32)
    pbuf, poff, plen = _pack(patterns)
    qbuf, _, _ = _pack(quals)
    k = np.ascontiguousarray(k, dtype=np.int32)
    score = np.zeros(n, np.int32); prob = np.zeros(n, np.float64)
    net = np.zeros(n, np.int32); tot = np.zeros(n, np.int32); span = np.zeros(n, np.int32)
    rc = lib().snapref_landau_vishkin(C.c_int(direction), C.c_uint32(n), ptr(tbuf), ptr(toff), ptr(tlen),
                                      ptr(pbuf), ptr(qbuf), ptr(poff), ptr(plen), ptr(k),
                                      ptr(score), ptr(prob), ptr(net), ptr(tot), ptr(span))
    assert rc == 0
    return dict(score=score, match_probability=prob, net_indel=net, total_indels=tot, text_span=span)


def affine_gap(direction: int, texts, patterns, quals, w, score_init, is_rc, banded, use_clip=None,
               agparams=(1, 4, 6, 1, 10, 7)):
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
    agp = np.array(agparams, dtype=np.int32)
    ag = np.zeros(n, np.int32); to = np.zeros(n, np.int32); po = np.zeros(n, np.int32)
    ne = np.zeros(n, np.int32); prob = np.zeros(n, np.float64)
    rc = lib().snapref_affine_gap(C.c_int(direction), C.c_uint32(n), ptr(agp), ptr(tbuf), ptr(toff), ptr(tlen),
                                  ptr(pbuf), ptr(qbuf), ptr(poff), ptr(plen), ptr(w), ptr(score_init),
                                  ptr(is_rc), ptr(banded), ptr(use_clip),
                                  ptr(ag), ptr(to), ptr(po), ptr(ne), ptr(prob))
    assert rc == 0
    return dict(ag_score=ag, text_offset=to, pattern_offset=po, n_edits=ne, match_probability=prob)


def tables(n_indel: int = 1001, n_perfect: int = 1001):
    phred = np.zeros(256); indel = np.zeros(n_indel); perfect = np.zeros(n_perfect)
    lib().snapref_tables(ptr(phred), ptr(indel), C.c_uint32(n_indel), ptr(perfect), C.c_uint32(n_perfect))
    return phred, indel, perfect


def compute_mapq(p_all: float, p_best: float, score: int, popular_skipped: int) -> int:
    f = lib().snapref_compute_mapq
    f.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    return int(f(p_all, p_best, score, popular_skipped))


def wrapped_next_seed(seed_len: int, wrap_count: int) -> int:
    return int(lib().snapref_wrapped_next_seed(C.c_uint(seed_len), C.c_uint(wrap_count)))


def seed_prob(seed_len: int) -> float:
    f = lib().snapref_seed_prob
    f.restype = C.c_double
    f.argtypes = [C.c_int]
    return float(f(seed_len))
