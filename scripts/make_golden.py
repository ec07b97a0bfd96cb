"""Generate tests/golden/ fixtures with the compiled reference (oracle/_ref).  Run here (where
/root/reference exists); the fixtures travel to the GPU box, the reference sources do not.

  tiny_index.npz      a complete SNAP index (seed 20) of a 120 kb two-contig + one ALT contig
                      synthetic genome, as the arrays snap_amd.index.GenomeIndex holds
  tiny_reads.npz      3000 x 100 bp + 1000 x 150 bp reads and the reference's
                      SingleAlignmentResult for each under three option sets
  reference_kats.json the reference's own known-answer vectors for LV / affine gap
                      (tests/LandauVishkinTest.cpp:11-32, tests/AffineGapVectorizedTest.cpp:39-67)
  primitives.npz      seeded LV / affine-gap / seed-lookup problems with the reference's answers
"""
import json, os, sys, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from oracle import ref

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
W = '/tmp/snap_golden'
shutil.rmtree(W, ignore_errors=True); os.makedirs(W); os.makedirs(OUT, exist_ok=True)

# ---- genome: two primary contigs with repeats and N runs + an ALT contig that copies part of chrA
g = synth.make_genome(20260925, 100_000, n_contigs=2, repeat_frac=0.4, max_copies=60, repeat_len=(150, 1200), n_run_frac=0.004)
rng = np.random.default_rng(99)
alt = g[0][1][20_000:32_000].copy()
mut = rng.random(alt.size) < 0.01
alt[mut] = synth._ACGT[rng.integers(0, 4, size=int(mut.sum()))]
g.append(('chrA_alt1', alt))
synth.write_fasta(W + '/ref.fa', g)
ref.build_index(W + '/ref.fa', W + '/idx', 20, threads=4, extra=['-altContigName', 'chrA_alt1'])
idx = GenomeIndex.load_from_directory(W + '/idx')
assert any(c.is_alt for c in idx.contigs), idx.contigs
np.savez_compressed(OUT + '/tiny_index.npz',
                    meta=np.array([idx.seed_len, idx.key_bytes, idx.n_hash_tables, int(idx.large), idx.location_size,
                                   idx.chromosome_padding, idx.n_bases], dtype=np.int64),
                    overflow=idx.overflow, hash_blob=idx.hash_blob, table_offset=idx.table_offset,
                    table_size=idx.table_size, genome_padded=idx.genome_padded,
                    contig_begin=idx.contig_begin, contig_is_alt=np.array([c.is_alt for c in idx.contigs]),
                    contig_names=np.array([c.name for c in idx.contigs]))

# ---- reads
ri = ref.RefIndex(W + '/idx')
sets = {}
r100 = synth.make_reads(1, g, 3000, 100, sub=0.012, ins=0.002, dele=0.002, n_frac=0.001)
r150 = synth.make_reads(2, g, 1000, 150, sub=0.02, ins=0.004, dele=0.004)
# edge cases appended to the 100 bp set: all-N read, read with > maxK Ns, poly-A, exact copy at contig start / end
b, q = r100['bases'], r100['quals']
cat0 = g[0][1]
b[0] = ord('N')
b[1, :12] = ord('N')
b[2] = ord('A')
b[3] = cat0[:100]; b[4] = cat0[-100:]
b[5] = synth._COMP[cat0[500:600][::-1]]
opts = dict(default_d8=dict(max_k=8), lvonly_d8=dict(max_k=8, use_affine_gap=0), default_d27=dict(max_k=27),
            emitalt_d8=dict(max_k=8, emit_alt_alignments=1))
out = dict(b100=r100['bases'], q100=r100['quals'], b150=r150['bases'], q150=r150['quals'])
for name, kw in opts.items():
    p = abi.default_params(max_read_len=160, **kw)
    for tag, rd in (('100', r100), ('150', r150)):
        prim, alt_r, cnt, _ = ri.align_single(p, rd['bases'], rd['quals'], rd['offsets'], threads=1)
        # The reference's banded affine gap can trace back through cells an *earlier* call left in
        # the aligner object (AffineGapVectorized.h:743), so a few results depend on which reads
        # the same thread aligned before.  Find them by varying the history.
        unstable = np.zeros(len(prim), bool)
        n = len(prim)
        rev = np.arange(n)[::-1].copy()
        variants = [(np.arange(n), 3), (rev, 1), (np.random.default_rng(5).permutation(n), 2)]
        for order, th in variants:
            L = rd['bases'].shape[1]
            pv, _, _, _ = ri.align_single(p, rd['bases'][order], rd['quals'][order], np.arange(n + 1, dtype=np.uint64) * L, threads=th)
            back = np.empty_like(pv); back[order] = pv
            for f in prim.dtype.names:
                unstable |= prim[f] != back[f]
        print(name, tag, 'reference-unstable reads:', np.nonzero(unstable)[0].tolist())
        out['%s_%s_unstable' % (name, tag)] = unstable
        out['%s_%s_primary' % (name, tag)] = prim
        out['%s_%s_alt' % (name, tag)] = alt_r
        out['%s_%s_counters' % (name, tag)] = np.array([cnt['lookups'], cnt['lv'], cnt['ag']], dtype=np.int64)
np.savez_compressed(OUT + '/tiny_reads.npz', **out)

# ---- primitive problems
def mutate(s, rate):
    o = bytearray()
    for c in s:
        r = rng.random()
        if r < rate: o.append(b'ACGT'[rng.integers(0, 4)])
        elif r < rate * 1.4:
            if rng.random() < 0.5: continue
            o.append(c)
            for _ in range(rng.integers(1, 4)): o.append(b'ACGT'[rng.integers(0, 4)])
        else: o.append(c)
    return bytes(o)
prim = {}
cat = np.concatenate([x for _, x in g])
N = 600
texts, pats, quals, ks = [], [], [], []
for i in range(N):
    L = int(rng.integers(1, 150)); s = int(rng.integers(0, len(cat) - 400))
    t = cat[s:s + L + 40].tobytes()
    pt = mutate(t[:L], rng.choice([0.0, 0.01, 0.03, 0.1]))[:L] or b'A'
    texts.append(t); pats.append(pt); quals.append(bytes(rng.integers(35, 74, size=len(pt), dtype=np.uint8))); ks.append(int(rng.integers(0, 31)))
prim['lv_texts'] = np.array(texts, dtype=object); prim['lv_pats'] = np.array(pats, dtype=object)
prim['lv_quals'] = np.array(quals, dtype=object); prim['lv_k'] = np.array(ks, dtype=np.int32)
for d in (1, -1):
    tt = texts if d == 1 else [x[::-1] for x in texts]
    r = ref.landau_vishkin(d, tt, pats, quals, ks)
    for key, v in r.items(): prim['lv%+d_%s' % (d, key)] = v
ag = dict(texts=[], pats=[], quals=[], w=[], si=[], rc=[], clip=[], banded=[])
for i in range(N):
    L = int(rng.integers(1, 140))
    gg = bytes(rng.choice(list(b'ACGT'), size=L + 130).astype(np.uint8))
    if rng.random() < 0.1: gg = gg[:10] + b'N' + gg[11:]
    pt = mutate(gg[:L], rng.choice([0, 0.01, 0.03, 0.08, 0.2]))[:L] or b'A'
    if rng.random() < 0.15:
        cut = int(rng.integers(0, len(pt))); pt = (pt[:cut] + pt[cut + int(rng.integers(1, 12)):]) or b'A'
    w = int(rng.integers(3, 30)); tl = min(len(pt) + (127 if rng.random() < 0.7 else w), len(gg))
    ag['texts'].append(gg[:tl]); ag['pats'].append(pt); ag['quals'].append(bytes(rng.integers(35, 74, size=len(pt), dtype=np.uint8)))
    ag['w'].append(w); ag['si'].append(int(rng.choice([150, 100, len(pt) + 20, 30]))); ag['rc'].append(int(rng.integers(0, 2)))
    ag['clip'].append(int(rng.integers(0, 2))); ag['banded'].append(int(rng.integers(0, 2)))
for k2 in ('texts', 'pats', 'quals'): prim['ag_' + k2] = np.array(ag[k2], dtype=object)
for k2 in ('w', 'si', 'rc', 'clip', 'banded'): prim['ag_' + k2] = np.array(ag[k2], dtype=np.int32)
# keep only problems on which the C restatement sees no stale traceback read (reference result well defined)
import ctypes as C
lib = C.CDLL(os.path.join(os.path.dirname(OUT), '..', 'oracle', 'liboracle.so'))
for d in (1, -1):
    tt = ag['texts'] if d == 1 else [x[::-1] for x in ag['texts']]
    r = ref.affine_gap(d, tt, ag['pats'], ag['quals'], ag['w'], ag['si'], ag['rc'], ag['banded'], ag['clip'])
    for key, v in r.items(): prim['ag%+d_%s' % (d, key)] = v
pos = rng.integers(0, len(cat) - 20, size=2000)
seeds = np.stack([cat[s:s + 20] for s in pos]); seeds[::5] = synth._ACGT[rng.integers(0, 4, size=seeds[::5].shape)]
nh, h = ri.lookup_seeds(seeds, 128)
prim['seeds'] = seeds; prim['seed_n_hits'] = nh; prim['seed_hits'] = h
np.savez_compressed(OUT + '/primitives.npz', **prim)

kats = dict(
    lv=[dict(text=t, pattern=p, k=k, expect=e) for (e, t, p, k) in [
        (0, "abcde", "abcde", 2), (0, "abcde", "abcd", 2), (0, "abcde", "abc", 2), (0, "abcde", "ab", 2),
        (1, "abcde", "abcdX", 2), (1, "abcde", "abde", 2), (1, "abcde", "bcde", 2), (1, "abcde", "abcXde", 2),
        (2, "abcde", "abXXe", 2), (2, "abcde", "abcXXde", 2), (-1, "abcde", "XXXXX", 2)]],
    ag=[dict(text=t, pattern=p, w=w, score_init=si, expect=e, params=[1, 4, 6, 1, 10, 5]) for (e, t, p, w, si) in [
        (25, "ACGTA", "ACGTA", 16, 20), (21, "AACGTACGT", "ACGTACGT", 16, 20), (26, "ACGTAAAAACGTACGTACGT", "ACGTACGTACGTACGT", 16, 20),
        (104, "CCGTCTCAACAATAACAACAACAACAACAAAAACCAGTCACTGTGTTAGGGACAGTCAGAACATGGGGGGATGGGAAAGAGGAGTTACAGGGAGACTT", "CCGTCTCAACAATAACAACAACAACAACAACAAAAGCCAGTCACTGTGTTAGGGACAGTCAGAACATGGGGGGATGGGAAAGAGGAGTTACAGGGAGACTT", 16, 20),
        (103, "ACAATTAGGCAAAAAATCAATGGGATTCAGACAAATATGGGACAATTTTCTCTCTCTGTCTCTCTCTCTGTCTCTCTCTCTGACACACACACACA", "ACAATTAGGCAAAAAATCAATGGGATTCAGACAAATATGGGACAATTTTCTCTCTCTGTCTCTCTCTCTGTCTCTGTCTCTCTCTCTGACACACACACACA", 16, 20),
        (72, "CATTGGCCAGGCTGGTCTCGAACTCCTGACCTCATGATCCACACGCCTCGA", "TGTTGGTCAGGCTGGTCTCGAACTCCT", 16, 60),
        (80, "CAAAAATTAGCTGGGCACGGTGGCAGGCGCCTGTAATCCCAGCTACTCAGGAGACTGAGGCAGGAGAA", "GAAAAATTAGCTGTGCACGGTGGCAGGCGCCTGTA", 16, 55),
        (41, "CAAAAAATTAGCCACGCATGGTGGCATATCCCTGTAGTCCCAGCTACTCGGGGCTGAGGCAGGAG", "GAAAAATTAGCTGTGCACGGTGGCAGGCGCCTGTA", 16, 40),
        (95, "TAACCAATTAGACAGCTTCTTCCCACCCCAGACCCCAGAGACCTGGCCCAAGCCTGGAGAAGACATCCTGTTTCCCCTGAGGAAGTGGCCCAGATTG", "AAACCAATTAGACAGCTTCTTC", 16, 78),
        (83, "CTCTGTCTCTCTCTCTGTCTCTCTCTTTTAACAGGGTATAAACAGACTTAGGGTAACTAAAAAACGGATTAACAATAAGTGATACGA", "CTCTGTCTCTGTCTCTCTCTCTGTCTCTCTCTTTTAACAGGGTATAAACAGACTTAGGGTAACTAAAAAACGGATTAACA", 8, 21)]],
    source="tests/LandauVishkinTest.cpp:11-32 and tests/AffineGapVectorizedTest.cpp:39-67 of the reference (SNAP 2.0.5); quality string all '2'")
json.dump(kats, open(OUT + '/reference_kats.json', 'w'), indent=1)
print({f: os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)})
