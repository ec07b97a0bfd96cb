"""Generate tests/golden/cigar_ag.npz with the compiled reference (oracle/_ref): SAMFormat::computeCigar, affine-gap variant
(SAM.cpp:2470-2588), on the golden index of make_golden.py.  Items as in make_golden_cigar.py, each with its qualities and the
edit distance the reference's aligner reported (the k of the band).  Every item is answered twice: by a fresh, zero-filled
AffineGapVectorizedWithCigar (a function of the item alone: the expectation) and by one object serving the whole batch in order
(what a SAM writer thread does); items on which the two differ are recorded as `unstable` -- the reference's banded traceback read
cells an earlier call left behind (AffineGapVectorized.cpp:811 over AffineGapVectorized.h:1441)."""
import os, sys, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth
from snap_amd.index import GenomeIndex
from oracle import ref
from tests import util

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
W = '/tmp/snap_golden_cigar_ag'
shutil.rmtree(W, ignore_errors=True); os.makedirs(W)
g = synth.make_genome(20260925, 100_000, n_contigs=2, repeat_frac=0.4, max_copies=60, repeat_len=(150, 1200), n_run_frac=0.004)
rng = np.random.default_rng(99)
alt = g[0][1][20_000:32_000].copy()
mut = rng.random(alt.size) < 0.01
alt[mut] = synth._ACGT[rng.integers(0, 4, size=int(mut.sum()))]
g.append(('chrA_alt1', alt))
synth.write_fasta(W + '/ref.fa', g)
ref.build_index(W + '/ref.fa', W + '/idx', 20, threads=4, extra=['-altContigName', 'chrA_alt1'])
idx = GenomeIndex.load_from_directory(W + '/idx')
gold = util.load_golden_index()
assert (idx.contig_begin == gold.contig_begin).all() and (idx.genome_padded == gold.genome_padded).all()
ri = ref.RefIndex(W + '/idx')
z = np.load(OUT + '/tiny_reads.npz')
rng = np.random.default_rng(20260927)
items = []          # (data, quals, loc, extra_before, score)
for name, tag in (('default_d8', '100'), ('default_d27', '100'), ('default_d27', '150'), ('default_d8', '150')):
    b, q = z['b' + tag], z['q' + tag]; prim = z['%s_%s_primary' % (name, tag)]
    for i in np.nonzero(prim['status'] != 0)[0][:600]:
        rc = int(prim['direction'][i])
        d = synth._COMP[b[i][::-1]] if rc else b[i]
        qq = q[i][::-1] if rc else q[i]
        if rng.random() < 0.3:
            qq = qq.copy(); lowq = rng.random(qq.size) < 0.3; qq[lowq] = rng.integers(35, 64, size=int(lowq.sum()))     # qualities < 65 enable the "flip" passes
        sc = int(prim['score'][i])
        items.append((d.tobytes(), qq.tobytes(), int(prim['location'][i]), 0, sc))
        r = rng.random()
        if r < 0.12:
            items.append((d.tobytes(), qq.tobytes(), int(prim['location'][i]) + int(rng.integers(-4, 5)), 0, sc + 4))
        elif r < 0.18:
            items.append((d.tobytes(), qq.tobytes(), int(prim['location'][i]), int(rng.integers(1, 12)), sc))
        elif r < 0.24:
            items.append((d.tobytes(), qq.tobytes(), int(prim['location'][i]), 0, max(0, sc - 2)))                        # band too narrow: the full form takes over
        elif r < 0.30:
            items.append((d.tobytes(), qq.tobytes(), int(prim['location'][i]), 0, int(rng.integers(17, 40))))            # pattern shorter than 3 (2k + 1): full form
nb = idx.n_bases
cb = [int(x) for x in idx.contig_begin] + [int(nb)]
pad = idx.chromosome_padding
G = idx.genome_padded[(idx.genome_padded.size - nb) // 2:]
for c in range(len(cb) - 1):
    real_end = cb[c + 1] - pad
    for L in (100, 150):
        for hang in (1, 3, 10, 40):
            start = real_end - L + hang
            d = G[start:start + L].copy(); d[L - hang:] = synth._ACGT[rng.integers(0, 4, size=hang)]
            qq = rng.integers(40, 74, size=L).astype(np.uint8)
            items.append((d.tobytes(), qq.tobytes(), start, 0, 3))
            d2 = np.append(np.delete(d, L // 2), synth._ACGT[rng.integers(0, 4)])
            items.append((d2.tobytes(), qq.tobytes(), start, 0, 4))
            d3 = np.insert(d, L // 2, synth._ACGT[rng.integers(0, 4)])[:L]
            items.append((d3.tobytes(), qq.tobytes(), start, 0, 4))
    items.append((G[cb[c + 1] - 60:cb[c + 1] + 40].tobytes(), bytes([60] * 100), cb[c + 1] - 60, 0, 2))
for _ in range(30):                       # random locations: nothing aligns
    L = int(rng.choice([100, 150]))
    items.append((synth._ACGT[rng.integers(0, 4, size=L)].tobytes(), bytes([60] * L), int(rng.integers(cb[0], cb[1] - 2000)), 0, int(rng.integers(0, 12))))

data = np.frombuffer(b''.join(x[0] for x in items), dtype=np.uint8).copy()
quals = np.frombuffer(b''.join(x[1] for x in items), dtype=np.uint8).copy()
length = np.array([len(x[0]) for x in items], dtype=np.int32)
off = np.zeros(len(items), dtype=np.uint64); off[1:] = np.cumsum(length)[:-1]
loc = np.array([x[2] for x in items], dtype=np.int64)
extra = np.array([x[3] for x in items], dtype=np.int32)
score = np.array([x[4] for x in items], dtype=np.int32)
out = dict(data=data, quals=quals, off=off, length=length, loc=loc, extra_before=extra, score=score)
for use_m in (0, 1):
    r = ri.compute_cigar_ag(data, quals, off, length, loc, extra, score, bool(use_m), fresh_object=True, ops_stride=256)
    shared = ri.compute_cigar_ag(data, quals, off, length, loc, extra, score, bool(use_m), fresh_object=False, ops_stride=256)
    unstable = np.zeros(len(items), bool)
    for k in r:
        unstable |= (r[k] != shared[k]).reshape(len(items), -1).any(axis=1)
    for k, v in r.items():
        out['m%d_%s' % (use_m, k)] = v
    out['m%d_unstable' % use_m] = unstable
    print('use_m', use_m, 'items', len(items), 'star', int((r['n_ops'] < 0).sum()), 'leading-D', int((r['add_front_clipping'] > 0).sum()),
          'leading-I', int((r['add_front_clipping'] < 0).sum()), 'hanging', int((r['extra_clipped_after'] > 0).sum()),
          'failed', int((r['edit_distance'] < 0).sum()), 'tail ins', int((r['back_clipping_missed'] > 0).sum()),
          'unstable', int(unstable.sum()), 'max ops', int(r['n_ops'].max()))
np.savez_compressed(OUT + '/cigar_ag.npz', **out)
print('written', OUT + '/cigar_ag.npz')
