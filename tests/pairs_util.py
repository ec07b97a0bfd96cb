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


# ---- secondary results (-om): comparison of (primary, alt, secondary, nsec, single_secondary, nssec) tuples
# fields of a secondary result the reference never writes: they hold whatever its buffer held (0 on this side)
PAIRED_SECONDARY_UNSET = ("ref_span", "reserved", "flags", "probability_all_pairs", "liftover", "supplementary",
                          "clipping_for_read_adjustment", "aligned_as_pair", "ag_forced_single_aligner_call")
SINGLE_SECONDARY_UNSET = ("reserved", "probability_all_candidates", "popular_seeds_skipped")


def load_paired_secondary_sets(z):
    import ast
    return [(str(r[0]), ast.literal_eval(str(r[1])), ast.literal_eval(str(r[2])), int(r[3]), int(r[4]), int(r[5])) for r in z["sets"]]


def compare_paired_secondary(ref_t, got_t, exclude):
    """Position-by-position comparison of the secondary results of two runs; returns a list of problems."""
    _, _, rsec, rn, rssec, rns = ref_t
    _, _, gsec, gn, gssec, gns = got_t
    problems = []
    m = (rn != gn) & ~exclude
    if m.any():
        i = int(np.nonzero(m)[0][0])
        problems.append("nSecondaryResults differs for %d pairs, first %d: ref %d got %d" % (int(m.sum()), i, rn[i], gn[i]))
    m = (rns != gns).any(axis=1) & ~exclude
    if m.any():
        i = int(np.nonzero(m)[0][0])
        problems.append("nSingleEndSecondaryResults differs for %d pairs, first %d: ref %s got %s" % (int(m.sum()), i, rns[i], gns[i]))
    w = min(rsec.shape[1], gsec.shape[1])
    live = (np.arange(w)[None, :] < np.minimum(rn, gn)[:, None]) & ~exclude[:, None]
    for f in rsec.dtype.names:
        if f in PAIRED_SECONDARY_UNSET:
            continue
        d = rsec[f][:, :w] != gsec[f][:, :w]
        if d.ndim == 3:
            d = d.any(axis=2)
        d &= live
        if d.any():
            i, k = [int(x[0]) for x in np.nonzero(d)]
            problems.append("secondary[%d].%s differs for %d records, first at pair %d: ref %r got %r" % (k, f, int(d.sum()), i, rsec[f][i, k], gsec[f][i, k]))
    w = min(rssec.shape[1], gssec.shape[1])
    live = (np.arange(w)[None, :] < np.minimum(rns.sum(axis=1), gns.sum(axis=1))[:, None]) & ~exclude[:, None]
    for f in rssec.dtype.names:
        if f in SINGLE_SECONDARY_UNSET:
            continue
        d = (rssec[f][:, :w] != gssec[f][:, :w]) & live
        if d.any():
            i, k = [int(x[0]) for x in np.nonzero(d)]
            problems.append("single_secondary[%d].%s differs for %d records, first at pair %d: ref %r got %r" % (k, f, int(d.sum()), i, rssec[f][i, k], gssec[f][i, k]))
    return problems
