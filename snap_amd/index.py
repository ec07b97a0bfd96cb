"""Loader for a SNAP index directory (the reference's on-disk format, unchanged).

Mirrors GenomeIndex::loadFromDirectory (SNAPLib/GenomeIndex.cpp:1839-2093),
Genome::loadFromFile (SNAPLib/Genome.cpp:277-438) and SNAPHashTable::loadCommon
(SNAPLib/HashTable.cpp:98-175); layout summary in SURVEY.md Appendix B.

The four files are parsed into flat numpy arrays that are handed to the C ABI as a
`snapgpu_index_view`.  The hash-table slot bytes are kept exactly as the reference's index
builder wrote them, so the device probe sequence is the reference's probe sequence.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field

import numpy as np

HASH_MAGIC = 0xB111B010  # SNAPLib/HashTable.cpp:343
GENOME_PAD = 1024        # >= Genome::N_PADDING (1000), SNAPLib/Genome.h:446; keeps 16-byte alignment


@dataclass
class Contig:
    begin: int
    is_alt: bool
    original_number: int
    name: str
    # ALT-to-primary projection (Genome.h:386-400): only meaningful for ALT contigs of an index built with -altLiftoverFile
    proj_begin: int = 0
    proj_rc: bool = False
    proj_cigar: str = "*"


def parse_proj_cigar(cigar: str):
    """Genome.cpp:389-403: repeated sscanf("%d%c"); returns [(count, action), ...] ('*' gives none)."""
    return [(int(c), a) for c, a in re.findall(r"(\d+)([A-Za-z=])", cigar)]


@dataclass
class GenomeIndex:
    """Host-side image of a SNAP index (32-bit locations only in this build)."""
    seed_len: int
    key_bytes: int
    n_hash_tables: int
    large: bool
    location_size: int
    chromosome_padding: int
    overflow: np.ndarray              # uint32[overflow_table_size]
    hash_blob: np.ndarray             # uint8[...] concatenated slot arrays (headers stripped)
    table_offset: np.ndarray          # uint64[n_hash_tables] byte offsets into hash_blob
    table_size: np.ndarray            # uint64[n_hash_tables] slots
    genome_padded: np.ndarray         # uint8[GENOME_PAD + n_bases + GENOME_PAD], 'n' padded
    n_bases: int
    contigs: list = field(default_factory=list)
    directory: str = ""

    @property
    def entry_bytes(self) -> int:
        return 4 * (2 if self.large else 1) + self.key_bytes

    @property
    def genome(self) -> np.ndarray:
        return self.genome_padded[GENOME_PAD:GENOME_PAD + self.n_bases]

    @property
    def contig_begin(self) -> np.ndarray:
        return np.array([c.begin for c in self.contigs], dtype=np.uint64)

    def projection_arrays(self):
        """(proj_begin u64[n], proj_rc u8[n], cigar_start u32[n+1], cigar_ops u32[...]) for snapgpu_index_view."""
        pb = np.array([c.proj_begin for c in self.contigs], dtype=np.uint64)
        rc = np.array([1 if c.proj_rc else 0 for c in self.contigs], dtype=np.uint8)
        start = [0]; ops = []
        for c in self.contigs:
            for cnt, act in parse_proj_cigar(c.proj_cigar):
                ops.append((cnt << 8) | ord(act))
            start.append(len(ops))
        return pb, rc, np.array(start, dtype=np.uint32), np.array(ops if ops else [0], dtype=np.uint32)

    @property
    def first_alt_location(self) -> int:
        # Genome::sortContigsByName, SNAPLib/Genome.cpp:462-476
        alts = [c.begin for c in self.contigs if c.is_alt]
        return min(alts) if alts else (1 << 62)

    def nbytes(self) -> int:
        return self.overflow.nbytes + self.hash_blob.nbytes + self.genome_padded.nbytes

    @staticmethod
    def load_from_directory(directory: str) -> "GenomeIndex":
        with open(os.path.join(directory, "GenomeIndex"), "rb") as f:
            fields = f.read().split()
        if len(fields) < 10:
            raise ValueError("GenomeIndex header has %d fields, expected 10" % len(fields))
        (major, minor, n_tables, overflow_size, seed_len, padding, key_bytes,
         hash_file_size, small, location_size) = [int(x) for x in fields[:10]]
        if major != 7:
            raise ValueError("index major version %d != 7 (GenomeIndex.h:170)" % major)
        if not 4 <= location_size <= 8:
            raise ValueError("GenomeIndex header: location size %d" % location_size)
        # 5 .. 8 bytes per location (what the indexer picks for seeds shorter than 20, GenomeIndex.cpp:446-453, or -locationSize): the
        # reference then goes through lookupSeed / overflowTable64 (GenomeIndex.cpp:2205-2328).  Where every value fits 32 bits the
        # tables are narrowed below, slot for slot, as snapgpu_create_from_directory does; a genome that needs wider values is refused.
        wide = location_size > 4

        # --- Genome ---------------------------------------------------------------
        with open(os.path.join(directory, "Genome"), "rb") as f:
            hdr = f.readline().split()
            n_bases, n_contigs = int(hdr[0]), int(hdr[1])
            contigs = []
            for _ in range(n_contigs):
                line = f.readline().rstrip(b"\n")
                parts = line.split(b" ", 7)
                begin = int(parts[0]); flags = int(parts[1], 16); orig = int(parts[2])
                name_len = int(parts[5]); cigar_len = int(parts[6])
                name = parts[7][:name_len].decode()
                cigar = parts[7][name_len + 1:name_len + 1 + cigar_len].decode()       # "%s %s": name, blank, CIGAR (Genome.cpp:226)
                contigs.append(Contig(begin, bool(flags & 1), orig, name, proj_begin=int(parts[3]),
                                      proj_rc=bool(int(parts[4], 16) & 1), proj_cigar=cigar or "*"))
            genome_padded = np.full(n_bases + 2 * GENOME_PAD, ord("n"), dtype=np.uint8)
            got = f.readinto(memoryview(genome_padded)[GENOME_PAD:GENOME_PAD + n_bases])
            if got != n_bases:
                raise ValueError("Genome file truncated: %d of %d bases" % (got, n_bases))

        # --- OverflowTable --------------------------------------------------------
        overflow = np.fromfile(os.path.join(directory, "OverflowTable"), dtype=np.uint64 if wide else np.uint32)
        if overflow.size != overflow_size:
            raise ValueError("OverflowTable has %d words, header says %d" % (overflow.size, overflow_size))
        if wide:
            if n_bases + overflow_size >= 0xfffffffe or (overflow.size and int(overflow.max()) > 0xffffffff):
                raise NotImplementedError("index has %d-byte genome locations and its values do not fit 32 bits: only indexes whose "
                                          "genome + overflow table stay below 2^32 - 2 are supported (narrowed on load)" % location_size)
            overflow = overflow.astype(np.uint32)
        if overflow.size == 0:
            overflow = np.zeros(1, dtype=np.uint32)   # keep a valid pointer

        # --- GenomeIndexHash ------------------------------------------------------
        value_count = 1 if small else 2
        if not wide:
            # 4-byte locations (the north star's shape; ~25 GB at GRCh38 scale): every table's slots are read straight into their place in
            # ONE array -- the file is never held a second time (reading it whole and concatenating the tables cost twice its size in host
            # memory on the rank that loads the directory before the broadcast)
            entry = 4 * value_count + key_bytes
            path = os.path.join(directory, "GenomeIndexHash")
            fsize = os.path.getsize(path)
            hdr_bytes = n_tables * (32 + location_size)
            hash_blob = np.zeros(fsize - hdr_bytes + 16, dtype=np.uint8)        # 16 bytes of slack: 8-byte device loads of the last slot stay in bounds
            offs = np.zeros(n_tables, dtype=np.uint64)
            sizes = np.zeros(n_tables, dtype=np.uint64)
            out_pos = 0
            with open(path, "rb") as f:
                for t in range(n_tables):
                    h = np.frombuffer(f.read(32 + location_size), dtype=np.uint8)
                    if h.size != 32 + location_size or int(h[0:4].view(np.uint32)[0]) != HASH_MAGIC:
                        raise ValueError("hash table %d: bad header" % t)
                    table_size = int(h[4:12].view(np.uint64)[0])
                    ks, vs, vc = [int(x) for x in h[20:32].view(np.uint32)]
                    if ks != key_bytes or vs != location_size or vc != value_count:
                        raise ValueError("hash table %d: key/value sizes %d/%d/%d do not match header" % (t, ks, vs, vc))
                    nbytes = table_size * entry
                    got = f.readinto(memoryview(hash_blob)[out_pos:out_pos + nbytes])
                    if got != nbytes:
                        raise ValueError("GenomeIndexHash truncated in table %d" % t)
                    offs[t] = out_pos; sizes[t] = table_size
                    out_pos += nbytes
                if f.read(1):
                    raise ValueError("GenomeIndexHash: trailing bytes")
            if out_pos + 16 != hash_blob.size:
                raise ValueError("GenomeIndexHash: %d bytes of tables, file says %d" % (out_pos, hash_blob.size - 16))
            return GenomeIndex(seed_len=seed_len, key_bytes=key_bytes, n_hash_tables=n_tables, large=not small, location_size=4,
                               chromosome_padding=padding, overflow=overflow, hash_blob=hash_blob, table_offset=offs, table_size=sizes,
                               genome_padded=genome_padded, n_bases=n_bases, contigs=contigs, directory=directory)
        raw = np.fromfile(os.path.join(directory, "GenomeIndexHash"), dtype=np.uint8)
        entry = 4 * value_count + key_bytes
        src_entry = location_size * value_count + key_bytes
        all_ones = (1 << (8 * location_size)) - 1
        offs = np.zeros(n_tables, dtype=np.uint64)
        sizes = np.zeros(n_tables, dtype=np.uint64)
        pieces = []
        pos = 0
        out_pos = 0
        for t in range(n_tables):
            magic = int(raw[pos:pos + 4].view(np.uint32)[0])
            if magic != HASH_MAGIC:
                raise ValueError("hash table %d: bad magic %#x" % (t, magic))
            table_size = int(raw[pos + 4:pos + 12].view(np.uint64)[0])
            ks, vs, vc = [int(x) for x in raw[pos + 20:pos + 32].view(np.uint32)]
            if ks != key_bytes or vs != location_size or vc != value_count:
                raise ValueError("hash table %d: key/value sizes %d/%d/%d do not match header" % (t, ks, vs, vc))
            pos += 32 + vs                          # header + invalidValue
            src_bytes = table_size * src_entry
            nbytes = table_size * entry
            if not wide:
                pieces.append(raw[pos:pos + nbytes])
            else:                                   # [value x value_count][key]: values of location_size bytes -> 4 (all ones = unused, all ones - 1 = other strand only)
                src = raw[pos:pos + src_bytes].reshape(table_size, src_entry)
                dst = np.zeros((table_size, entry), dtype=np.uint8)
                for k in range(value_count):
                    v = np.zeros(table_size, dtype=np.uint64)
                    for b in range(location_size):
                        v |= src[:, k * location_size + b].astype(np.uint64) << np.uint64(8 * b)
                    w = v.copy()
                    w[v == all_ones] = 0xffffffff
                    w[v == all_ones - 1] = 0xfffffffe
                    if ((v < all_ones - 1) & (v >= 0xfffffffe)).any():
                        raise NotImplementedError("hash table value does not fit 32 bits")
                    dst[:, 4 * k:4 * k + 4] = w.astype(np.uint32).view(np.uint8).reshape(table_size, 4)
                dst[:, 4 * value_count:] = src[:, location_size * value_count:]
                pieces.append(dst.reshape(-1))
            offs[t] = out_pos
            sizes[t] = table_size
            pos += src_bytes
            out_pos += nbytes
        if pos != raw.size:
            raise ValueError("GenomeIndexHash: %d trailing bytes" % (raw.size - pos))
        # 16 bytes of slack so that 8-byte device loads of the last slot stay in bounds
        hash_blob = np.concatenate(pieces + [np.zeros(16, dtype=np.uint8)])

        return GenomeIndex(seed_len=seed_len, key_bytes=key_bytes, n_hash_tables=n_tables,
                           large=not small, location_size=4,             # (narrowed above where the files have wider values)
                           chromosome_padding=padding, overflow=overflow, hash_blob=hash_blob,
                           table_offset=offs, table_size=sizes, genome_padded=genome_padded,
                           n_bases=n_bases, contigs=contigs, directory=directory)


def build_index(fasta: str, out_dir: str | None, seed_len: int = 20, slack: float = 0.3, key_bytes: int = 0, padding: int = 2000,
                device: int = 0, alt_liftover_file: str | None = None, alt_contig_names=(), non_alt_contig_names=(), auto_alt: bool = True,
                max_alt_contig_size: int = -1, space_terminates_name: bool = True, lib=None, keep: bool = False):
    """`snap-aligner index <fasta> <out_dir> -s <seed_len> ...` with the build on the GPU (GenomeIndex::runIndexer,
    SNAPLib/GenomeIndex.cpp:126-506; include/snapgpu.h: snapgpu_index_build_from_fasta / snapgpu_built_index_save).
    Writes the reference's four files into out_dir (None: nothing is written) and returns the build's statistics; with keep=True returns
    (stats, BuiltIndex) -- the index still resident in HBM, to hand to an aligner without going through the files."""
    import ctypes as C
    from .abi import IndexBuildParams, IndexBuildStats
    if lib is None:
        from .aligner import load_library
        lib = load_library()
    lib.snapgpu_index_build_from_fasta.argtypes = [C.c_char_p, C.POINTER(IndexBuildParams), C.c_int, C.POINTER(C.c_void_p)]
    lib.snapgpu_built_index_save.argtypes = [C.c_void_p, C.c_char_p]
    lib.snapgpu_built_index_stats.argtypes = [C.c_void_p, C.POINTER(IndexBuildStats)]
    lib.snapgpu_built_index_destroy.argtypes = [C.c_void_p]
    lib.snapgpu_built_index_destroy.restype = None
    lib.snapgpu_default_index_build_params.argtypes = [C.POINTER(IndexBuildParams)]
    lib.snapgpu_default_index_build_params.restype = None
    lib.snapgpu_last_error.restype = C.c_char_p
    lib.snapgpu_last_error.argtypes = [C.c_void_p]
    bp = IndexBuildParams()
    lib.snapgpu_default_index_build_params(C.byref(bp))
    bp.seed_len, bp.slack, bp.key_bytes, bp.chromosome_padding = seed_len, slack, key_bytes, padding
    bp.auto_alt = 1 if auto_alt else 0
    bp.max_alt_contig_size = max_alt_contig_size
    bp.space_terminates_name = 1 if space_terminates_name else 0
    keepalive = []
    if alt_liftover_file:
        bp.alt_liftover_file = alt_liftover_file.encode()
    for field, names in (("alt_contig_names", alt_contig_names), ("non_alt_contig_names", non_alt_contig_names)):
        if names:
            arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
            keepalive.append(arr)
            setattr(bp, field, arr)
            setattr(bp, "n_" + field, len(names))
    h = C.c_void_p()
    rc = lib.snapgpu_index_build_from_fasta(fasta.encode(), C.byref(bp), C.c_int(device), C.byref(h))
    if rc != 0:
        raise RuntimeError("snapgpu_index_build_from_fasta failed (%d): %s" % (rc, (lib.snapgpu_last_error(None) or b"").decode()))
    try:
        if out_dir is not None:
            rc = lib.snapgpu_built_index_save(h, out_dir.encode())
            if rc != 0:
                raise RuntimeError("snapgpu_built_index_save failed (%d): %s" % (rc, (lib.snapgpu_last_error(None) or b"").decode()))
        st = IndexBuildStats()
        lib.snapgpu_built_index_stats(h, C.byref(st))
    except BaseException:
        lib.snapgpu_built_index_destroy(h)
        raise
    if keep:
        return st.as_dict(), BuiltIndex(lib, h)
    lib.snapgpu_built_index_destroy(h)
    return st.as_dict()


class BuiltIndex:
    """An index built by snapgpu_index_build* and still resident in HBM (include/snapgpu.h: snapgpu_built_index)."""

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle

    def view(self):
        import ctypes as C
        from .abi import IndexView
        v = IndexView()
        self.lib.snapgpu_built_index_view.argtypes = [C.c_void_p, C.POINTER(IndexView)]
        rc = self.lib.snapgpu_built_index_view(self.handle, C.byref(v))
        if rc != 0:
            raise RuntimeError("snapgpu_built_index_view failed (%d)" % rc)
        return v

    def save(self, out_dir: str):
        import ctypes as C
        self.lib.snapgpu_built_index_save.argtypes = [C.c_void_p, C.c_char_p]
        rc = self.lib.snapgpu_built_index_save(self.handle, out_dir.encode())
        if rc != 0:
            raise RuntimeError("snapgpu_built_index_save failed (%d)" % rc)

    def close(self):
        if self.handle:
            self.lib.snapgpu_built_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
