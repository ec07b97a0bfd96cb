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


def alt_liftover_genome():
    """The genome of tests/golden/paired_alt_index.npz: three primary contigs plus
        chrA_alt1  forward copy of chrA[20000:32000] with a 10-base deletion and a 20-base insertion   5000M10D2990M20I4000M
        chrB_alt2  reverse-complement copy of chrB[40000:49000] behind 100 novel bases                  100S9000M (flag 16)
    Returns (contigs, text of the -altLiftoverFile, extra `snap-aligner index` arguments without the file name)."""
    g = synth.make_genome(20260927, 240_000, n_contigs=3, repeat_frac=0.25, max_copies=30, repeat_len=(150, 1200))
    rng = np.random.default_rng(5)

    def mutate(a, rate):
        a = a.copy(); m = rng.random(a.size) < rate
        a[m] = synth._ACGT[rng.integers(0, 4, size=int(m.sum()))]
        return a
    A, B = g[0][1], g[1][1]
    alt1 = np.concatenate([mutate(A[20000:25000], 0.01), mutate(A[25010:28000], 0.01), synth._ACGT[rng.integers(0, 4, size=20)],
                           mutate(A[28000:32000], 0.01)])
    alt2 = np.concatenate([synth._ACGT[rng.integers(0, 4, size=100)], synth._COMP[mutate(B[40000:49000], 0.012)[::-1]]])
    sam = ("@HD\tVN:1.0\nchrA_alt1\t0\tchrA\t20001\t255\t5000M10D2990M20I4000M\t*\t0\t0\t*\t*\n"
           "chrB_alt2\t16\tchrB\t40001\t255\t100S9000M\t*\t0\t0\t*\t*\n")
    return g + [("chrA_alt1", alt1), ("chrB_alt2", alt2)], sam, ["-altContigName", "chrA_alt1", "-altContigName", "chrB_alt2"]
