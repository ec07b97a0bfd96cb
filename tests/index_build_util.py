"""Shared body of the index-builder parity tests (test infrastructure; uses oracle/_ref = the reference compiled in place).

The GPU builder (include/snapgpu.h: snapgpu_index_build_from_fasta, snap_amd/index.build_index) and the reference's own
`snap-aligner index` are run on the same FASTA; then
  * `Genome` must be the same file, byte for byte, and `GenomeIndex` must agree in every field but the hash-file size
    (table sizes: the reference estimates distinct seeds with approximate counters unless -exact);
  * the REFERENCE loads both directories and must answer every probed seed identically -- hit counts and hit lists, both strands --
    for every seed of the genome (or a sample) plus absent seeds;
  * the reference aligns reads over both directories with identical results.
"""
import os

import numpy as np

from snap_amd import abi, synth
from snap_amd.index import GenomeIndex, build_index


def hard_fasta(path, seed=7, size=120_000):
    """Three regular contigs with planted repeats, an N run, lower-case stretches, IUPAC codes, a header with a description after a
    blank, CRLF line ends in one contig, a short last line -- and an `_alt` contig that must move to the end of the genome."""
    rng = np.random.default_rng(seed)
    contigs = synth.make_genome(seed, size, n_contigs=3, repeat_frac=0.35, max_copies=60, repeat_len=(60, 900), max_divergence=0.04,
                                n_run_frac=0.01)
    alt = contigs[0][1][5000:9000].copy()
    alt[::97] = ord("A")
    with open(path, "wb") as f:
        for k, (name, g) in enumerate(contigs):
            g = g.copy()
            f.write(b">" + name.encode() + (b" description of " + name.encode() if k == 0 else b"") + b"\n")
            lo = int(rng.integers(100, 2000))
            g[lo:lo + 700] = np.frombuffer(bytes(g[lo:lo + 700]).lower(), dtype=np.uint8)           # soft-masked stretch
            if k == 1:
                g[3000:3005] = np.frombuffer(b"RYKMS", dtype=np.uint8)                             # IUPAC codes -> N
            eol = b"\r\n" if k == 2 else b"\n"
            for i in range(0, len(g), 61):
                f.write(bytes(g[i:i + 61]) + eol)
            if k == 0:                                                                              # the ALT contig sits in the middle of the file
                f.write(b">chrA_fix_alt\n")
                for i in range(0, len(alt), 80):
                    f.write(bytes(alt[i:i + 80]) + b"\n")
    return contigs


def compare_with_reference(tmpdir, lib=None, seed_len=20, fasta=None, n_reads=3000, extra_ref=(), **build_kw):
    from oracle import ref
    tmpdir = str(tmpdir)
    fasta = fasta or os.path.join(tmpdir, "g.fa")
    if not os.path.exists(fasta):
        hard_fasta(fasta)
    d_ref, d_gpu = os.path.join(tmpdir, "idx_ref"), os.path.join(tmpdir, "idx_gpu")
    ref.build_index(fasta, d_ref, seed_len, threads=8, extra=["-exact"] + list(extra_ref))
    stats = build_index(fasta, d_gpu, seed_len=seed_len, lib=lib, **build_kw)

    # ---- files
    assert open(os.path.join(d_ref, "Genome"), "rb").read() == open(os.path.join(d_gpu, "Genome"), "rb").read(), "Genome files differ"
    ha = open(os.path.join(d_ref, "GenomeIndex")).read().split()
    hb = open(os.path.join(d_gpu, "GenomeIndex")).read().split()
    assert len(ha) == len(hb) == 10
    for i, (x, y) in enumerate(zip(ha, hb)):
        if i != 7:                                        # [7] = hash file size
            assert x == y, "GenomeIndex field %d: reference %s, GPU builder %s" % (i, x, y)
    # -exact: the reference sizes its tables from exact distinct-seed counts too, with the same formula
    assert ha[7] == hb[7], "hash table sizes differ from the reference's -exact build"
    assert os.path.getsize(os.path.join(d_ref, "OverflowTable")) == os.path.getsize(os.path.join(d_gpu, "OverflowTable"))
    ia, ib = GenomeIndex.load_from_directory(d_ref), GenomeIndex.load_from_directory(d_gpu)
    assert (ia.table_size == ib.table_size).all()
    eb = ib.entry_bytes

    def used_slots(o, n):
        e = np.asarray(ib.hash_blob[int(o):int(o) + int(n) * eb]).reshape(int(n), eb)
        return int(np.count_nonzero((e[:, :4] != 0xff).any(axis=1)))

    assert stats["n_distinct_seeds"] == sum(used_slots(o, n) for o, n in zip(ib.table_offset, ib.table_size))

    # ---- lookups, answered by the reference over both directories
    ra, rb = ref.RefIndex(d_ref), ref.RefIndex(d_gpu)
    g = ia.genome
    n_pos = len(g) - seed_len
    pos = np.arange(0, n_pos, max(1, n_pos // 60000))
    seeds = np.lib.stride_tricks.sliding_window_view(g, seed_len)[pos]
    ok = np.isin(seeds, np.frombuffer(b"ACGT", dtype=np.uint8)).all(axis=1)
    seeds = np.ascontiguousarray(seeds[ok])
    rng = np.random.default_rng(5)
    absent = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(4000, seed_len))]
    allseeds = np.concatenate([seeds, absent])
    na, ha_ = ra.lookup_seeds(allseeds, 128)
    nb, hb_ = rb.lookup_seeds(allseeds, 128)
    assert (na == nb).all(), "hit counts differ for %d seeds" % int((na != nb).any(axis=1).sum())
    assert (ha_ == hb_).all(), "hit lists differ"
    assert (na[:len(seeds), 0] >= 1).all()

    # ---- alignments by the reference over both directories
    contigs = [(c.name, ia.genome[c.begin:(ia.contigs[i + 1].begin - ia.chromosome_padding if i + 1 < len(ia.contigs) else ia.n_bases - ia.chromosome_padding)])
               for i, c in enumerate(ia.contigs)]
    reads = synth.make_reads(11, [c for c in contigs if len(c[1]) > 400], n_reads, 100)
    params = abi.default_params(max_k=8, max_read_len=112)
    pa = ra.align_single(params, reads["bases"], reads["quals"], reads["offsets"], threads=2)[0]
    pb = rb.align_single(params, reads["bases"], reads["quals"], reads["offsets"], threads=2)[0]
    from tests import util
    assert not util.compare_results(pa, pb), util.compare_results(pa, pb)
    assert (pa["status"] != 0).mean() > 0.9
    return stats, d_ref, d_gpu
