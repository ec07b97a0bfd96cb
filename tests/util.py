"""Shared helpers for the test-suite (test infrastructure; may use oracle/)."""
import ctypes as C
import os

import numpy as np

from snap_amd.index import Contig, GenomeIndex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# fields of SingleAlignmentResult the reference leaves undefined when status == NotFound
UNDEFINED_WHEN_NOT_FOUND = ("match_probability", "probability_all_candidates", "orig_location",
                            "popular_seeds_skipped")


def load_golden_index(name: str = "tiny_index.npz") -> GenomeIndex:
    z = np.load(os.path.join(GOLDEN, name))
    m = z["meta"]
    contigs = [Contig(int(b), bool(a), i, str(n)) for i, (b, a, n) in
               enumerate(zip(z["contig_begin"], z["contig_is_alt"], z["contig_names"]))]
    if "contig_proj_begin" in z.files:                      # index built with -altLiftoverFile
        for c, pb, rc, cg in zip(contigs, z["contig_proj_begin"], z["contig_proj_rc"], z["contig_proj_cigar"]):
            c.proj_begin, c.proj_rc, c.proj_cigar = int(pb), bool(rc), str(cg)
    return GenomeIndex(seed_len=int(m[0]), key_bytes=int(m[1]), n_hash_tables=int(m[2]), large=bool(m[3]),
                       location_size=int(m[4]), chromosome_padding=int(m[5]), overflow=z["overflow"],
                       hash_blob=z["hash_blob"], table_offset=z["table_offset"], table_size=z["table_size"],
                       genome_padded=z["genome_padded"], n_bases=int(m[6]), contigs=contigs)


def compare_results(ref, got, what="primary", exclude=None):
    """Field-by-field, bit-exact comparison of two RESULT_DTYPE arrays; returns list of problems.

    `exclude` masks reads for which the reference's own answer is not a function of the read
    (banded affine-gap traceback through stale cells; flagged by the GPU path in `reserved`)."""
    problems = []
    found = ref["status"] != 0
    for f in ref.dtype.names:
        if f == "reserved":
            continue
        ne = ref[f] != got[f]
        if exclude is not None:
            ne &= ~exclude
        if f in UNDEFINED_WHEN_NOT_FOUND:
            ne &= found
        if ne.any():
            i = int(np.nonzero(ne)[0][0])
            problems.append("%s.%s differs for %d reads, first at %d: ref=%r got=%r" %
                            (what, f, int(ne.sum()), i, ref[f][i], got[f][i]))
    return problems


# ---------------------------------------------------------------- C restatement (oracle/liboracle.so)
class _AGP(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("m", "s", "o", "e", "f", "t")]


class _OIndex(C.Structure):
    _fields_ = [("seed_len", C.c_uint32), ("key_bytes", C.c_uint32), ("n_hash_tables", C.c_uint32),
                ("large", C.c_uint32), ("hash_blob", C.c_void_p), ("table_offset", C.c_void_p),
                ("table_size", C.c_void_p), ("overflow", C.c_void_p), ("n_bases", C.c_uint64)]


_olib = None


def oracle_lib():
    global _olib
    if _olib is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run __graft_entry__.build()")
        lib = C.CDLL(path)
        lib.oracle_ag.restype = C.c_int
        lib.oracle_lv.restype = C.c_int
        lib.oracle_seed_prob.restype = C.c_double
        lib.oracle_seed_prob.argtypes = [C.c_int]
        lib.oracle_compute_mapq.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
        lib.oracle_phred_table.restype = C.POINTER(C.c_double)
        lib.oracle_indel_table.restype = C.POINTER(C.c_double)
        lib.oracle_perfect_table.restype = C.POINTER(C.c_double)
        lib.oracle_init()
        _olib = lib
    return _olib


def _padded(text: bytes, direction: int):
    tb = b"n" * 64 + text + b"n" * 64
    buf = C.create_string_buffer(tb, len(tb))
    base = C.addressof(buf) + 64
    return buf, (base if direction == 1 else base + len(text))


def oracle_lv(direction, text, pattern, quality, k):
    lib = oracle_lib()
    buf, tp = _padded(text, direction)
    pb = C.create_string_buffer(pattern + b"\0" * 8)
    qb = C.create_string_buffer(quality + b"\0" * 8)
    mp = C.c_double(); ni = C.c_int(); ti = C.c_int(); ts = C.c_int()
    s = lib.oracle_lv(direction, C.c_void_p(tp), len(text), pb, qb, len(pattern), int(k), C.byref(mp), C.byref(ni),
                      C.byref(ti), C.byref(ts))
    return dict(score=s, match_probability=mp.value, net_indel=ni.value, total_indels=ti.value, text_span=ts.value)


def oracle_ag(direction, banded, text, pattern, quality, w, score_init, is_rc, use_clip, params=(1, 4, 6, 1, 10, 7)):
    lib = oracle_lib()
    buf, tp = _padded(text, direction)
    pb = C.create_string_buffer(pattern + b"\0" * 8)
    qb = C.create_string_buffer(quality + b"\0" * 8)
    to = C.c_int(); po = C.c_int(); ne = C.c_int(); mp = C.c_double(); st = C.c_int()
    prm = _AGP(*params)
    s = lib.oracle_ag(direction, int(banded), C.byref(prm), C.c_void_p(tp), len(text), pb, qb, len(pattern), int(w),
                      int(score_init), int(is_rc), int(use_clip), C.byref(to), C.byref(po), C.byref(ne), C.byref(mp), C.byref(st))
    return dict(ag_score=s, text_offset=to.value, pattern_offset=po.value, n_edits=ne.value,
                match_probability=mp.value, stale_reads=st.value)


def oracle_lookup(index: GenomeIndex, seed: bytes):
    """GenomeIndex::lookupSeed32 through the C restatement; returns None if not a seed."""
    lib = oracle_lib()
    bases = C.c_uint64(); rc = C.c_uint64()
    if not lib.oracle_pack_seed(seed, index.seed_len, C.byref(bases), C.byref(rc)):
        return None
    ix = _OIndex(index.seed_len, index.key_bytes, index.n_hash_tables, 1 if index.large else 0,
                 index.hash_blob.ctypes.data, index.table_offset.ctypes.data, index.table_size.ctypes.data,
                 index.overflow.ctypes.data, index.n_bases)
    nh = (C.c_int64 * 2)(); hp = (C.c_void_p * 2)(); sg = (C.c_uint32 * 2)(); sl = (C.c_uint32 * 2)()
    lib.oracle_lookup_seed(C.byref(ix), bases, rc, nh, hp, sg, sl)
    out = []
    for d in range(2):
        n = int(nh[d])
        if n <= 0:
            hits = np.zeros(0, np.uint32)
        else:
            hits = np.ctypeslib.as_array(C.cast(hp[d], C.POINTER(C.c_uint32)), shape=(n,)).copy()
        out.append((n, hits, int(sl[d])))
    return out


class _OGenome(C.Structure):
    _fields_ = [("genome", C.c_void_p), ("n_bases", C.c_uint64), ("genome_pad", C.c_uint32), ("chromosome_padding", C.c_uint32),
                ("contig_begin", C.c_void_p), ("n_contigs", C.c_uint32), ("first_alt_location", C.c_uint64)]


def oracle_align_reads(index: GenomeIndex, params, bases, quals, offsets, secondary=None, sec_stride: int = 64):
    """BaseAligner::AlignRead through the C restatement (oracle/align_oracle.c), one read after the other.
    secondary = snap_amd.abi.SecondaryParams or None.  Returns (primary, first_alt[, secondary, n_secondary])."""
    from snap_amd.abi import RESULT_DTYPE
    lib = oracle_lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1)
    quals = np.ascontiguousarray(quals, dtype=np.uint8).reshape(-1)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    ix = _OIndex(index.seed_len, index.key_bytes, index.n_hash_tables, 1 if index.large else 0,
                 index.hash_blob.ctypes.data, index.table_offset.ctypes.data, index.table_size.ctypes.data,
                 index.overflow.ctypes.data, index.n_bases)
    pad = (index.genome_padded.size - index.n_bases) // 2
    cb = np.ascontiguousarray(index.contig_begin, dtype=np.uint64)
    alts = [c.begin for c in index.contigs if c.is_alt]
    g = _OGenome(index.genome_padded.ctypes.data + pad, index.n_bases, pad, index.chromosome_padding, cb.ctypes.data, len(index.contigs),
                 min(alts) if alts else index.n_bases + (1 << 40))
    prim = np.zeros(n, dtype=RESULT_DTYPE); alt = np.zeros(n, dtype=RESULT_DTYPE)
    if secondary is None:
        rc = lib.oracle_align_reads(C.byref(ix), C.byref(g), C.byref(params), C.c_uint32(n), C.c_void_p(bases.ctypes.data),
                                    C.c_void_p(quals.ctypes.data), C.c_void_p(offsets.ctypes.data), C.c_void_p(prim.ctypes.data),
                                    C.c_void_p(alt.ctypes.data), None, None, C.c_uint32(0), None)
        assert rc == 0
        return prim, alt
    sec = np.zeros((n, sec_stride), dtype=RESULT_DTYPE); nsec = np.zeros(n, dtype=np.uint32)
    rc = lib.oracle_align_reads(C.byref(ix), C.byref(g), C.byref(params), C.c_uint32(n), C.c_void_p(bases.ctypes.data),
                                C.c_void_p(quals.ctypes.data), C.c_void_p(offsets.ctypes.data), C.c_void_p(prim.ctypes.data),
                                C.c_void_p(alt.ctypes.data), C.byref(secondary), C.c_void_p(sec.ctypes.data), C.c_uint32(sec_stride),
                                C.c_void_p(nsec.ctypes.data))
    assert rc == 0
    return prim, alt, sec, nsec


def oracle_genome(index: GenomeIndex):
    """(struct, keep-alive tuple) for the C restatement's oracle_genome."""
    pad = (index.genome_padded.size - index.n_bases) // 2
    cb = np.ascontiguousarray(index.contig_begin, dtype=np.uint64)
    alts = [c.begin for c in index.contigs if c.is_alt]
    g = _OGenome(index.genome_padded.ctypes.data + pad, index.n_bases, pad, index.chromosome_padding, cb.ctypes.data, len(index.contigs),
                 min(alts) if alts else index.n_bases + (1 << 40))
    return g, (cb, index.genome_padded)


def oracle_compute_cigar_lv(index: GenomeIndex, data, off, length, loc, extra_before, use_m, ops_stride: int = 64):
    """SAMFormat::computeCigar (Landau-Vishkin variant) through the C restatement (oracle/cigar_oracle.c), item by item."""
    lib = oracle_lib()
    g, keep = oracle_genome(index)
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    n = len(off)
    ops = np.zeros((n, ops_stride), dtype=np.uint32); n_ops = np.zeros(n, dtype=np.int32)
    ed = np.zeros(n, dtype=np.int32); afc = np.zeros(n, dtype=np.int32); after = np.zeros(n, dtype=np.int64)
    for i in range(n):
        no, e, a, x = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int64(0)
        rc = lib.oracle_compute_cigar_lv(C.byref(g), C.c_void_p(data.ctypes.data + int(off[i])), C.c_int64(int(length[i])),
                                         C.c_int64(int(extra_before[i])), C.c_int64(int(loc[i])), C.c_int(1 if use_m else 0),
                                         C.c_void_p(ops[i].ctypes.data), C.c_int(ops_stride), C.byref(no), C.byref(e), C.byref(a), C.byref(x))
        assert rc == 0
        n_ops[i], ed[i], afc[i], after[i] = no.value, e.value, a.value, x.value
    return dict(ops=ops, n_ops=n_ops, edit_distance=ed, add_front_clipping=afc, extra_clipped_after=after)


CIGAR_CODES = "MIDNSHP=X"


def cigar_text(ops, n_ops):
    """BAMAlignment::decodeCigar (Bam.cpp:350-376): "%u%c" per op; "*" for n_ops < 0."""
    if n_ops < 0:
        return "*"
    return "".join("%d%s" % (int(o) >> 4, CIGAR_CODES[int(o) & 15]) for o in ops[:n_ops])


# fields of a secondary result the reference never writes (BaseAligner.cpp:2182-2199); both sides hold 0
UNSET_IN_SECONDARY = ("probability_all_candidates", "popular_seeds_skipped", "reserved")


def compare_secondary(ref_sec, ref_n, got_sec, got_n, exclude):
    problems = []
    ne = (ref_n != got_n) & ~exclude
    if ne.any():
        i = int(np.nonzero(ne)[0][0])
        problems.append("nSecondaryResults differs for %d reads, first at %d: ref=%d got=%d" % (int(ne.sum()), i, ref_n[i], got_n[i]))
    width = min(ref_sec.shape[1], got_sec.shape[1])
    live = (np.arange(width)[None, :] < np.minimum(ref_n, got_n)[:, None]) & ~exclude[:, None]
    for f in ref_sec.dtype.names:
        if f in UNSET_IN_SECONDARY:
            continue
        d = (ref_sec[f][:, :width] != got_sec[f][:, :width]) & live
        if d.any():
            i, k = [int(x[0]) for x in np.nonzero(d)]
            problems.append("secondary[%d].%s differs for %d records, first at read %d: ref=%r got=%r" %
                            (k, f, int(d.sum()), i, ref_sec[f][i, k], got_sec[f][i, k]))
    return problems


# ---------------------------------------------------------------- device buffers without torch
class HipBuffers:
    """hipMalloc / hipMemcpy through ctypes on the HIP runtime that is ALREADY mapped into this process (the one
    libsnapgpu.so resolved), so that tests of the `_device` entry points do not depend on a second runtime (torch
    bundles its own libamdhip64) being able to initialise the GPU.  With the wavefront emulator (SNAPGPU_TEST_LIB)
    "device" memory is host memory and the emulator library's own hipMalloc shims are used."""

    def __init__(self, emu=None):
        from snap_amd.aligner import load_library
        load_library()
        path = None
        if emu:                             # the caller knows its library is the emulator's (libamdhip64 may be mapped by an earlier test)
            self.emu, self.rt, self.live = True, None, []
            return
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        self.emu = path is None
        self.rt = C.CDLL(path) if path else None
        self.live = []
        if self.rt is not None:
            self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            self.rt.hipFree.argtypes = [C.c_void_p]
            self.rt.hipDeviceSynchronize.argtypes = []

    def upload(self, x):
        """numpy array -> device pointer (int) holding the same bytes."""
        a = np.ascontiguousarray(x)
        nbytes = max(16, a.nbytes)
        if self.emu:                    # emulator: host memory is device memory
            buf = np.zeros(nbytes, np.uint8)
            buf[:a.nbytes] = a.view(np.uint8).reshape(-1)
            self.live.append(buf)
            return buf.ctypes.data
        p = C.c_void_p()
        rc = self.rt.hipMalloc(C.byref(p), nbytes)
        assert rc == 0 and p.value, "hipMalloc(%d) -> %d" % (nbytes, rc)
        rc = self.rt.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1)      # hipMemcpyHostToDevice
        assert rc == 0, "hipMemcpy H2D -> %d" % rc
        self.live.append(p.value)
        return p.value

    def download(self, dptr, like):
        """device pointer -> numpy array shaped / typed like `like`."""
        out = np.empty_like(np.ascontiguousarray(like))
        if self.emu:
            C.memmove(out.ctypes.data, dptr, out.nbytes)
            return out
        assert self.rt.hipDeviceSynchronize() == 0
        rc = self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), out.nbytes, 2)   # hipMemcpyDeviceToHost
        assert rc == 0, "hipMemcpy D2H -> %d" % rc
        return out

    def free_all(self):
        if not self.emu:
            for p in self.live:
                self.rt.hipFree(C.c_void_p(p))
        self.live = []


# ---------------------------------------------------------------- fresh-object reference answers (no parity exclusions)
_fresh_overrides = None


def fresh_overrides():
    global _fresh_overrides
    if _fresh_overrides is None:
        _fresh_overrides = np.load(os.path.join(GOLDEN, "fresh_overrides.npz"))
    return _fresh_overrides


def with_fresh_overrides(committed, key):
    """The fixture array `committed` (what one shared reference aligner object answered) with the records patched in that a reference
    aligner NEWLY CONSTRUCTED IN ZERO-FILLED MEMORY answers differently (scripts/make_golden_fresh.py): the reference's answer as a
    function of the read alone, so every read is compared and none excluded.  Returns (patched copy, indices that were patched)."""
    fo = fresh_overrides()
    out = committed.copy()
    idx = fo[key + "_idx"]
    if idx.size:
        out[idx] = fo[key + "_rec"]
    return out, idx
