"""Shared helpers of the paired-end tests: a seeded 'hard' pair set and the field-wise comparison."""
import numpy as np

from snap_amd import synth

# fields that are meaningful for every aligned read; fields of NotFound reads are not compared (the reference leaves
# most of them as they were)
PAIR_FIELDS = ["status", "direction", "location", "score", "mapq", "used_affine_gap_scoring", "bases_clipped_before",
               "bases_clipped_after", "ag_score", "aligned_as_pair"]


def hard_pairs(seed, contigs, n, L, **kw):
    """FR pairs with substitutions, short and long indels and Ns, plus: chimeric pairs, unalignable mates, unalignable
    pairs, ragged/short reads (some below -mrl), and reads whose head or tail is garbage (soft clipping)."""
    pr = synth.make_pairs(seed, contigs, n, L, sub=0.02, ins=0.003, dele=0.003, long_indel_frac=0.15, n_frac=0.003, **kw)
    rng = np.random.default_rng(seed + 1)
    b = pr["bases"]; q = pr["quals"]
    other = synth.make_reads(seed + 2, contigs, n, L)
    kind = rng.integers(0, 20, size=n)
    chim = kind == 0
    b[1::2][chim] = other["bases"][chim]
    rnd = kind == 1
    b[0::2][rnd] = synth._ACGT[rng.integers(0, 4, size=(int(rnd.sum()), L), dtype=np.uint8)]
    both = kind == 2
    b[0::2][both] = synth._ACGT[rng.integers(0, 4, size=(int(both.sum()), L), dtype=np.uint8)]
    b[1::2][both] = synth._ACGT[rng.integers(0, 4, size=(int(both.sum()), L), dtype=np.uint8)]
    lens = np.full(2 * n, L, dtype=np.int64)
    short = np.nonzero(kind == 3)[0]
    lens[2 * short] = rng.integers(15, L, size=short.size)
    short2 = np.nonzero(kind == 4)[0]
    lens[2 * short2 + 1] = rng.integers(15, L, size=short2.size)
    lens[2 * short2] = rng.integers(15, 60, size=short2.size)
    for i in np.nonzero(kind == 5)[0]:
        k = int(rng.integers(5, 50))
        b[2 * i, :k] = synth._ACGT[rng.integers(0, 4, size=k, dtype=np.uint8)]
    for i in np.nonzero(kind == 6)[0]:
        k = int(rng.integers(5, 60))
        b[2 * i + 1, L - k:] = synth._ACGT[rng.integers(0, 4, size=k, dtype=np.uint8)]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bb = np.concatenate([b[i, :lens[i]] for i in range(2 * n)])
    qq = np.concatenate([q[i, :lens[i]] for i in range(2 * n)])
    return dict(bases=bb, quals=qq, offsets=offs)


def compare_paired(ref_r, got, verbose=3, exclude=None, fields=PAIR_FIELDS):
    """Returns the boolean mask of pairs that differ in any compared field."""
    bad = np.zeros(ref_r.size, dtype=bool)
    why = {}
    for f in fields:
        a, b = ref_r[f], got[f]
        if a.ndim == 2:
            m = a != b
            if f != "status":
                m &= ref_r["status"] != 0
            m = m.any(axis=1)
        else:
            m = a != b
        if exclude is not None:
            m &= ~exclude
        if m.any():
            why[f] = int(m.sum())
        bad |= m
    if verbose:
        print("pairs", ref_r.size, "mismatching", int(bad.sum()), why)
        for i in np.nonzero(bad)[0][:verbose]:
            print("--- pair", i)
            for f in ref_r.dtype.names:
                if not np.array_equal(ref_r[f][i], got[f][i]):
                    print("   ", f, "ref", ref_r[f][i], "got", got[f][i])
    return bad
