"""Generate tests/golden/cigar_lv.npz with the compiled reference (oracle/_ref): SAMFormat::computeCigar (Landau-Vishkin variant,
SAM.cpp:2354-2467) on the golden index of make_golden.py -- the genome is rebuilt from the same seeds, so locations are those of
tiny_index.npz.  Items: the golden reads at the locations the reference aligned them to (three option sets), the same reads
shifted by a few bases (leading D / I: the addFrontClipping convention), reads hanging off the end of a contig, reads with
extraBasesClippedBefore, and reads at random locations (edit distance above MAX_K - 1: "no cigar")."""
import os, sys, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth
from snap_amd.index import GenomeIndex
from oracle import ref
from tests import util

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
W = '/tmp/snap_golden_cigar'
shutil.rmtree(W, ignore_errors=True); os.makedirs(W)

# ---- the genome of make_golden.py
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
items = []          # (data bytes, loc, extra_before)
rng = np.random.default_rng(20260926)


def oriented(read, direction):
    return synth._COMP[read[::-1]] if direction else read


for name, tag in (('default_d8', '100'), ('default_d27', '100'), ('default_d27', '150'), ('lvonly_d8', '150')):
    b = z['b' + tag]; prim = z['%s_%s_primary' % (name, tag)]
    for i in np.nonzero(prim['status'] != 0)[0][:700]:
        d = oriented(b[i], int(prim['direction'][i]))
        items.append((d.tobytes(), int(prim['location'][i]), 0))
        r = rng.random()
        if r < 0.12:                       # shifted: the alignment now starts with an insertion or a deletion
            items.append((d.tobytes(), int(prim['location'][i]) + int(rng.integers(-4, 5)), 0))
        elif r < 0.18:                     # extra clipping in front
            items.append((d.tobytes(), int(prim['location'][i]), int(rng.integers(1, 12))))
# reads hanging off the end of each contig (exact copies of the contig tail + bases past it), with and without a net indel near the end
nb = idx.n_bases
cb = [int(x) for x in idx.contig_begin] + [int(nb)]
pad = idx.chromosome_padding
G = idx.genome_padded[(idx.genome_padded.size - nb) // 2:]
for c in range(len(cb) - 1):
    real_end = cb[c + 1] - pad
    for L in (100, 150):
        for hang in (1, 3, 10, 40):
            start = real_end - L + hang
            d = G[start:start + L].copy()
            d[L - hang:] = synth._ACGT[rng.integers(0, 4, size=hang)]
            items.append((d.tobytes(), start, 0))
            d2 = np.delete(d, L // 2)                                   # a deletion in the read: net indel +1 moves the hang
            d2 = np.append(d2, synth._ACGT[rng.integers(0, 4)])
            items.append((d2.tobytes(), start, 0))
            d3 = np.insert(d, L // 2, synth._ACGT[rng.integers(0, 4)])[:L]   # an insertion: net indel -1
            items.append((d3.tobytes(), start, 0))
    items.append((G[cb[c]:cb[c] + 100].tobytes(), cb[c], 0))            # first bases of the contig
    items.append((G[real_end - 100:real_end].tobytes(), real_end - 100, 0))
    items.append((G[real_end - 30:real_end + 70].tobytes(), real_end - 30, 0))   # mostly in the padding
    items.append((G[cb[c + 1] - 60:cb[c + 1] + 40].tobytes(), cb[c + 1] - 60, 0))  # crosses into the next contig: getSubstring fails, "*"
# random locations: far more than MAX_K - 1 edits
for _ in range(40):
    L = int(rng.choice([100, 150]))
    items.append((synth._ACGT[rng.integers(0, 4, size=L)].tobytes(), int(rng.integers(cb[0], cb[1] - 2000)), 0))

data = np.frombuffer(b''.join(x[0] for x in items), dtype=np.uint8).copy()
length = np.array([len(x[0]) for x in items], dtype=np.int32)
off = np.zeros(len(items), dtype=np.uint64); off[1:] = np.cumsum(length)[:-1]
loc = np.array([x[1] for x in items], dtype=np.int64)
extra = np.array([x[2] for x in items], dtype=np.int32)
out = dict(data=data, off=off, length=length, loc=loc, extra_before=extra)
for use_m in (0, 1):
    r = ri.compute_cigar_lv(data, off, length, loc, extra, bool(use_m), ops_stride=256)
    for k, v in r.items():
        out['m%d_%s' % (use_m, k)] = v
    n_ops = r['n_ops']
    print('use_m', use_m, 'items', len(items), 'star', int((n_ops < 0).sum()), 'leading-D', int((r['add_front_clipping'] > 0).sum()),
          'leading-I', int((r['add_front_clipping'] < 0).sum()), 'hanging', int((r['extra_clipped_after'] > 0).sum()),
          'above limit', int((r['edit_distance'] == -1).sum()), 'max ops', int(n_ops.max()))
    # the restatement must agree before the fixture is written
    o = util.oracle_compute_cigar_lv(gold, data, off, length, loc, extra, bool(use_m), ops_stride=256)
    for k in r:
        if k == 'ops':
            for i in range(len(items)):
                assert (o['ops'][i, :max(0, o['n_ops'][i])] == r['ops'][i, :max(0, r['n_ops'][i])]).all(), (i, util.cigar_text(o['ops'][i], o['n_ops'][i]), util.cigar_text(r['ops'][i], r['n_ops'][i]))
        else:
            bad = np.nonzero(o[k] != r[k])[0]
            assert bad.size == 0, (k, bad[:5], o[k][bad[:5]], r[k][bad[:5]])
np.savez_compressed(OUT + '/cigar_lv.npz', **out)
print('written', OUT + '/cigar_lv.npz')
