"""Seeded synthetic genomes and reads.

GRCh38 is not available in this environment (no network, 62 GB hosts), so every config is
run against seeded synthetic genomes: uniform-random ACGT contigs with planted repeat
families (copy number / divergence configurable) and optional N runs, and reads sampled
from them with the error model BASELINE.md names (1 % substitution, 0.05 % insertion,
0.05 % deletion, 50 % reverse strand, Phred 20-40).

Everything is numpy-vectorised so that 1 M x 150 bp reads take a couple of seconds.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.full(256, ord("N"), dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def make_genome(seed: int, n_bases: int, n_contigs: int = 1, repeat_frac: float = 0.0,
                max_copies: int = 200, repeat_len=(200, 3000), max_divergence: float = 0.05,
                n_run_frac: float = 0.0):
    """Return a list of (name, uint8 array) contigs totalling ~n_bases bases.

    repeat_frac of the bases are overwritten by copies of repeat-family consensus
    sequences, each copy independently diverged by a per-family rate in [0, max_divergence].
    """
    rng = np.random.default_rng(seed)
    sizes = np.full(n_contigs, n_bases // n_contigs, dtype=np.int64)
    sizes[-1] += n_bases - sizes.sum()
    contigs = []
    for c, size in enumerate(sizes):
        g = _ACGT[rng.integers(0, 4, size=int(size), dtype=np.uint8)]
        contigs.append(g)
    if repeat_frac > 0:
        target = int(repeat_frac * n_bases)
        planted = 0
        while planted < target:
            flen = int(rng.integers(repeat_len[0], repeat_len[1] + 1))
            # copy numbers are log-uniform in [2, max_copies]
            copies = int(np.exp(rng.uniform(np.log(2), np.log(max_copies + 1))))
            copies = max(2, min(copies, max_copies, (target - planted) // flen + 2))
            div = rng.uniform(0, max_divergence)
            cons = _ACGT[rng.integers(0, 4, size=flen, dtype=np.uint8)]
            for _ in range(copies):
                ci = int(rng.integers(0, n_contigs))
                g = contigs[ci]
                if len(g) <= flen + 2:
                    continue
                pos = int(rng.integers(0, len(g) - flen))
                copy = cons.copy()
                nmut = rng.binomial(flen, div)
                if nmut:
                    where = rng.integers(0, flen, size=nmut)
                    copy[where] = _ACGT[rng.integers(0, 4, size=nmut, dtype=np.uint8)]
                g[pos:pos + flen] = copy
                planted += flen
    if n_run_frac > 0:
        for g in contigs:
            nruns = max(1, int(n_run_frac * len(g) / 500))
            for _ in range(nruns):
                rl = int(rng.integers(50, 1000))
                pos = int(rng.integers(0, max(1, len(g) - rl)))
                g[pos:pos + rl] = ord("N")
    return [("chr%s" % chr(ord("A") + i) if n_contigs <= 26 else "chr%d" % i, g)
            for i, g in enumerate(contigs)]


def write_fasta(path: str, contigs, width: int = 100) -> None:
    with open(path, "wb") as f:
        for name, g in contigs:
            f.write(b">" + name.encode() + b"\n")
            n = len(g)
            full = (n // width) * width
            if full:
                block = np.empty((full // width, width + 1), dtype=np.uint8)
                block[:, :width] = g[:full].reshape(-1, width)
                block[:, width] = ord("\n")
                f.write(block.tobytes())
            if n > full:
                f.write(g[full:].tobytes() + b"\n")


def make_reads(seed: int, contigs, n_reads: int, read_len: int, sub: float = 0.01,
               ins: float = 0.0005, dele: float = 0.0005, rc_frac: float = 0.5,
               qmin: int = 20, qmax: int = 40, n_frac: float = 0.0):
    """Sample reads; returns dict(bases[n,L] uint8, quals[n,L] uint8, contig, pos, rc).

    Indels are single-base events: a deletion skips one reference base, an insertion emits
    one random base without consuming reference.
    """
    rng = np.random.default_rng(seed)
    L = read_len
    lens = np.array([len(g) for _, g in contigs], dtype=np.int64)
    span = L + 16
    ok = lens > span + 1
    w = np.where(ok, lens - span, 0).astype(np.float64)
    ci = rng.choice(len(contigs), size=n_reads, p=w / w.sum())
    pos = (rng.random(n_reads) * (lens[ci] - span)).astype(np.int64)
    cat = np.concatenate([g for _, g in contigs])
    cstart = np.concatenate([[0], np.cumsum(lens)[:-1]])
    gpos = cstart[ci] + pos

    ev = np.zeros((n_reads, L), dtype=np.int8)          # +1 deletion before base j, -1 insertion at j
    r = rng.random((n_reads, L))
    ev[r < dele] = 1
    is_ins = (r >= dele) & (r < dele + ins)
    ev[is_ins] = -1
    ev[:, 0] = 0
    is_ins[:, 0] = False
    shift = np.cumsum(ev, axis=1, dtype=np.int64)
    # an inserted base does not consume reference: bases after it shift back by one
    idx = gpos[:, None] + np.arange(L)[None, :] + shift
    idx = np.clip(idx, 0, len(cat) - 1)
    bases = cat[idx]
    rnd = _ACGT[rng.integers(0, 4, size=(n_reads, L), dtype=np.uint8)]
    bases = np.where(is_ins, rnd, bases)
    subm = rng.random((n_reads, L)) < sub
    # substitute with a *different* base
    alt = _ACGT_SORTED[(np.searchsorted(_ACGT_SORTED, np.where(bases == ord("N"), ord("A"), bases))
                        + rng.integers(1, 4, size=(n_reads, L))) % 4]
    bases = np.where(subm & (bases != ord("N")), alt, bases)
    if n_frac > 0:
        bases = np.where(rng.random((n_reads, L)) < n_frac, np.uint8(ord("N")), bases)
    rc = rng.random(n_reads) < rc_frac
    bases[rc] = _COMP[bases[rc][:, ::-1]]
    quals = rng.integers(qmin + 33, qmax + 34, size=(n_reads, L), dtype=np.uint8)
    return dict(bases=np.ascontiguousarray(bases), quals=quals, contig=ci, pos=pos, rc=rc,
                offsets=np.arange(n_reads + 1, dtype=np.uint64) * L)


_ACGT_SORTED = np.sort(_ACGT)


def make_pairs(seed: int, contigs, n_pairs: int, read_len: int, insert_mean: float = 400.0, insert_sd: float = 50.0,
               insert_min: int = 150, insert_max: int = 1000, sub: float = 0.01, ins: float = 0.0005,
               dele: float = 0.0005, qmin: int = 20, qmax: int = 40, n_frac: float = 0.0,
               long_indel_frac: float = 0.0, long_indel_max: int = 10):
    """FR pairs: fragment length ~ N(insert_mean, insert_sd^2) clipped to
    [insert_min, insert_max]; mate 0 is the fragment's left end on the fragment strand, mate 1 the
    reverse complement of its right end; the fragment strand is forward for half of the pairs.
    Returns a dict with interleaved reads: bases[2n, L], quals[2n, L], offsets[2n+1], plus truth."""
    rng = np.random.default_rng(seed)
    L = read_len
    lens = np.array([len(g) for _, g in contigs], dtype=np.int64)
    frag = np.clip(np.rint(rng.normal(insert_mean, insert_sd, size=n_pairs)), max(insert_min, L), insert_max).astype(np.int64)
    span = frag + 32
    w = np.where(lens > span.max() + 1, lens - span.max(), 0).astype(np.float64)
    ci = rng.choice(len(contigs), size=n_pairs, p=w / w.sum())
    pos = (rng.random(n_pairs) * (lens[ci] - span)).astype(np.int64)
    cat = np.concatenate([g for _, g in contigs])
    cstart = np.concatenate([[0], np.cumsum(lens)[:-1]])
    gpos = cstart[ci] + pos

    def sample(start):
        ev = np.zeros((n_pairs, L), dtype=np.int64)
        r = rng.random((n_pairs, L))
        ev[r < dele] = 1
        is_ins = (r >= dele) & (r < dele + ins)
        ev[is_ins] = -1
        if long_indel_frac > 0:                       # one longer deletion in some reads (exercises affine gap)
            who = rng.random(n_pairs) < long_indel_frac
            at = rng.integers(L // 4, 3 * L // 4, size=n_pairs)
            ln = rng.integers(2, long_indel_max + 1, size=n_pairs)
            ev[who, at[who]] += ln[who]
        ev[:, 0] = 0
        is_ins[:, 0] = False
        shift = np.cumsum(ev, axis=1)
        idx = np.clip(start[:, None] + np.arange(L)[None, :] + shift, 0, len(cat) - 1)
        b = cat[idx]
        rnd = _ACGT[rng.integers(0, 4, size=(n_pairs, L), dtype=np.uint8)]
        b = np.where(is_ins, rnd, b)
        subm = rng.random((n_pairs, L)) < sub
        alt = _ACGT_SORTED[(np.searchsorted(_ACGT_SORTED, np.where(b == ord("N"), ord("A"), b))
                            + rng.integers(1, 4, size=(n_pairs, L))) % 4]
        b = np.where(subm & (b != ord("N")), alt, b)
        if n_frac > 0:
            b = np.where(rng.random((n_pairs, L)) < n_frac, np.uint8(ord("N")), b)
        return np.ascontiguousarray(b)

    left = sample(gpos)                               # forward-strand bases of the fragment's left end
    right = sample(gpos + frag - L)                   # forward-strand bases of its right end
    right_rc = _COMP[right[:, ::-1]]
    left_rc = _COMP[left[:, ::-1]]
    flip = rng.random(n_pairs) < 0.5                  # fragment from the reverse strand: mates swap roles
    m0 = np.where(flip[:, None], right_rc, left)
    m1 = np.where(flip[:, None], left, right_rc)
    bases = np.empty((2 * n_pairs, L), dtype=np.uint8)
    bases[0::2] = m0
    bases[1::2] = m1
    quals = rng.integers(qmin + 33, qmax + 34, size=(2 * n_pairs, L), dtype=np.uint8)
    return dict(bases=bases, quals=quals, offsets=np.arange(2 * n_pairs + 1, dtype=np.uint64) * L,
                contig=ci, pos=pos, frag=frag, flip=flip)


def write_fastq(path: str, reads, prefix: str = "r") -> None:
    bases, quals = reads["bases"], reads["quals"]
    with open(path, "wb") as f:
        for i in range(bases.shape[0]):
            f.write(b"@%s%d\n" % (prefix.encode(), i))
            f.write(bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
