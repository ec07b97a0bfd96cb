"""Generate tests/golden/sam_fields.npz: FLAG / RNAME / POS / MAPQ / CIGAR / NM as the unmodified reference CLI printed them
(oracle/_ref/snap-aligner single ... -o out.sam), together with what snapgpu_sam_fields_single needs to compute them: the reads
as written to the FASTQ, Read::clip's outcome (ClipBack of '#', the CLI default) and the reference aligner's result for each read
(oracle/_ref/libsnapref.so, same options).  Genome = the golden genome of make_golden.py (locations are tiny_index.npz's).
Six option sets: default (affine-gap cigars, M), -G- (Landau-Vishkin cigars only), both with -= (= / X instead of M), and both with -C++ (the reader
also clips a leading run of '#': front_clip > 0)."""
import os, sys, shutil, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from oracle import ref
from tests import util

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
W = '/tmp/snap_golden_samf'
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
rng = np.random.default_rng(20260928)
ACGT = synth._ACGT

reads = []          # (bases, quals) uint8 arrays
for tag, n in (('100', 1400), ('150', 1000)):
    b, q = z['b' + tag], z['q' + tag]
    for i in range(n):
        bb, qq = b[i].copy(), q[i].copy()
        kind = i % 25
        if kind == 1: L = int(rng.integers(30, len(bb))); bb, qq = bb[:L], qq[:L]                   # ragged, some below -mrl 50
        elif kind == 2: qq[len(qq) - int(rng.integers(1, 40)):] = ord('#')                           # '#' tail: clipped by the reader
        elif kind == 3: k = int(rng.integers(1, 4)); bb = np.concatenate([ACGT[rng.integers(0, 4, size=k)], bb])[:len(qq)]   # bases prepended: leading insertion
        elif kind == 4: k = int(rng.integers(1, 4)); bb = np.concatenate([bb[k:], ACGT[rng.integers(0, 4, size=k)]])       # first bases dropped
        elif kind == 5: bb[rng.integers(0, len(bb), size=12)] = ord('N')                             # too many Ns: not aligned
        elif kind == 6: j = int(rng.integers(2, 8)); bb = np.delete(bb, j); qq = qq[:len(bb)]         # deletion right after the start
        elif kind == 7: j = int(rng.integers(2, 8)); bb = np.insert(bb, j, ACGT[rng.integers(0, 4)])[:len(qq)]   # insertion right after the start
        elif kind == 8: lowq = rng.random(len(qq)) < 0.4; qq[lowq] = rng.integers(35, 64, size=int(lowq.sum()))
        elif kind == 9: qq[:int(rng.integers(1, 25))] = ord('#')                                   # '#' head: clipped only with -C++
        elif kind == 10: qq[:int(rng.integers(1, 15))] = ord('#'); qq[len(qq) - int(rng.integers(1, 15)):] = ord('#')
        reads.append((bb, qq))
# reads around contig boundaries: starting before the first base of a contig (the aligner may place them in the padding), ending past the last
nb = idx.n_bases
cb = [int(x) for x in idx.contig_begin] + [int(nb)]
pad = idx.chromosome_padding
G = idx.genome_padded[(idx.genome_padded.size - nb) // 2:]
for c in range(len(cb) - 1):
    real_end = cb[c + 1] - pad
    for L in (100, 150):
        for k in (1, 2, 5, 12):
            d = np.concatenate([ACGT[rng.integers(0, 4, size=k)], G[cb[c]:cb[c] + L - k]])           # k foreign bases, then the contig's first bases
            reads.append((d, rng.integers(45, 74, size=L).astype(np.uint8)))
            d = np.concatenate([G[real_end - (L - k):real_end], ACGT[rng.integers(0, 4, size=k)]])   # the contig's last bases, then k foreign ones
            reads.append((d, rng.integers(45, 74, size=L).astype(np.uint8)))
            reads.append((synth._COMP[d[::-1]], rng.integers(45, 74, size=L).astype(np.uint8)))
n = len(reads)
names = ['r%d' % i for i in range(n)]
with open(W + '/r.fq', 'wb') as f:
    for nm, (b, q) in zip(names, reads):
        f.write(b'@' + nm.encode() + b'\n' + b.tobytes() + b'\n+\n' + q.tobytes() + b'\n')

bases = np.concatenate([r[0] for r in reads]); quals = np.concatenate([r[1] for r in reads])
offsets = np.concatenate([[0], np.cumsum([len(r[0]) for r in reads])]).astype(np.uint64)
def clip_all(front_too):                                 # Read::clip (Read.h:567-620): ClipBack drops the trailing run of '#', ClipFrontAndBack then the leading one
    fc = np.zeros(n, dtype=np.int32); dl = np.zeros(n, dtype=np.int32)
    for i, (b, q) in enumerate(reads):
        m = len(q)
        while m > 0 and q[m - 1] == ord('#'):
            m -= 1
        f = 0
        if front_too:
            while f < m and q[f] == ord('#'):
                f += 1
        fc[i] = f; dl[i] = m - f
    return fc, dl


out = dict(bases=bases, quals=quals, offsets=offsets, contig_names=np.array([c.name for c in idx.contigs]))
contig_of = {c.name: i for i, c in enumerate(idx.contigs)}
CIG = {c: i for i, c in enumerate('MIDNSHP=X')}
for tag, cli, kw, use_m in (('default', [], {}, 1), ('lvonly', ['-G-'], dict(use_affine_gap=0), 1), ('eqx', ['-='], {}, 0), ('lvonly_eqx', ['-G-', '-='], dict(use_affine_gap=0), 0),
                            ('clipfront', ['-C++'], {}, 1), ('clipfront_lvonly', ['-C++', '-G-'], dict(use_affine_gap=0), 1)):
    front_clip, data_len = clip_all(tag.startswith('clipfront'))
    out['%s_front_clip' % tag] = front_clip; out['%s_data_len' % tag] = data_len
    sam = W + '/out_%s.sam' % tag
    r = subprocess.run([ref.CLI_PATH, 'single', W + '/idx', W + '/r.fq', '-o', sam, '-t', '1'] + cli, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    # what the CLI's aligner saw: clipped reads that pass the filters (SingleAligner.cpp:213-233: -mrl 50, more Ns than -d 14? no: maxDist)
    p = abi.default_params(max_read_len=400, **kw)
    keep = [i for i in range(n) if data_len[i] >= 50 and int((reads[i][0][front_clip[i]:front_clip[i] + data_len[i]] == ord('N')).sum()) <= int(p.max_k)]
    kb = np.concatenate([reads[i][0][front_clip[i]:front_clip[i] + data_len[i]] for i in keep]); kq = np.concatenate([reads[i][1][front_clip[i]:front_clip[i] + data_len[i]] for i in keep])
    ko = np.concatenate([[0], np.cumsum([data_len[i] for i in keep])]).astype(np.uint64)
    prim, _, _, _ = ri.align_single(p, kb, kq, ko, threads=1)
    results = np.zeros(n, dtype=abi.RESULT_DTYPE)
    results['status'] = 0; results['location'] = 0xFFFFFFFF; results['score'] = -1
    results[keep] = prim
    flag = np.zeros(n, np.int32); contig = np.full(n, -1, np.int32); pos = np.zeros(n, np.int64); mapq = np.zeros(n, np.int32)
    nmv = np.zeros(n, np.int32); n_ops = np.full(n, -1, np.int32); ops = np.zeros((n, 64), np.uint32)
    seen = 0
    for line in open(sam):
        if line.startswith('@'):
            continue
        t = line.rstrip('\n').split('\t')
        i = int(t[0][1:]); seen += 1
        flag[i] = int(t[1]); contig[i] = contig_of.get(t[2], -1); pos[i] = int(t[3]); mapq[i] = int(t[4])
        nmv[i] = int([x for x in t[11:] if x.startswith('NM:i:')][0][5:])
        if t[5] != '*':
            num = ''; k = 0
            for ch in t[5]:
                if ch.isdigit(): num += ch
                else: ops[i, k] = (int(num) << 4) | CIG[ch]; k += 1; num = ''
            n_ops[i] = k
    assert seen == n, (seen, n)
    for k, v in (('results', results), ('flag', flag), ('contig', contig), ('pos', pos), ('mapq', mapq), ('nm', nmv), ('n_ops', n_ops), ('ops', ops)):
        out['%s_%s' % (tag, k)] = v
    out['%s_use_m' % tag] = np.int32(use_m)
    cig0 = ops[:, 0] & 15
    print(tag, 'reads', n, 'unmapped', int((flag & 4 != 0).sum()), 'rc', int((flag & 16 != 0).sum()), 'leading S', int(((n_ops > 0) & (cig0 == 4)).sum()),
          'star among mapped', int(((flag & 4 == 0) & (n_ops < 0)).sum()), 'filtered', n - len(keep))
np.savez_compressed(OUT + '/sam_fields.npz', **out)
print('written', OUT + '/sam_fields.npz')
