"""Generate tests/golden/sam_fields_paired.npz: the 9 computed SAM fields (FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN NM) of both records
of each pair and the order of the two records, as the unmodified reference CLI printed them (oracle/_ref/snap-aligner paired ... -o out.sam
-t 1), with what snapgpu_sam_fields_paired needs to compute them: the reads as written to the two FASTQ files, Read::clip's outcome and the
reference's PairedAlignmentResult (ChimericPairedEndAligner::align, oracle/_ref/libsnapref.so, same options) for each pair.
Genome = the paired golden genome of make_golden_paired.py (locations are paired_index.npz's).  Pairs in which exactly one mate is
"useless" (too short / too many Ns: PairedAligner.cpp:708-760 aligns the other one alone) are not generated.
Option sets: default, -G- (Landau-Vishkin cigars), -= (= / X)."""
import os, sys, shutil, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from oracle import ref
from tests import util
from tests.pairs_util import hard_pairs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
W = '/tmp/snap_golden_samfp'
shutil.rmtree(W, ignore_errors=True); os.makedirs(W)
g = synth.make_genome(20260926, 240_000, n_contigs=3, repeat_frac=0.35, max_copies=40, repeat_len=(150, 1500), n_run_frac=0.003)
synth.write_fasta(W + '/ref.fa', g)
ref.build_index(W + '/ref.fa', W + '/idx', 20, threads=4)
idx = GenomeIndex.load_from_directory(W + '/idx')
gold = util.load_golden_index('paired_index.npz')
assert (idx.contig_begin == gold.contig_begin).all() and (idx.genome_padded == gold.genome_padded).all()
ri = ref.RefIndex(W + '/idx')
rng = np.random.default_rng(20260929)
ACGT = synth._ACGT
reads = []                               # [read0, read1, read0, read1, ...] as (bases, quals)
for seed, npairs, L, ins in ((17, 900, 150, 380), (18, 400, 100, 300)):
    pr = hard_pairs(seed, g, npairs, L, insert_mean=ins, **({} if L == 150 else dict(insert_min=100)))
    o = pr['offsets'].astype(np.int64)
    for i in range(npairs):
        mates = []
        for w in (0, 1):
            bb = pr['bases'][o[2 * i + w]:o[2 * i + w + 1]].copy(); qq = pr['quals'][o[2 * i + w]:o[2 * i + w + 1]].copy()
            kind = (i * 2 + w) % 23
            if kind == 2: qq[len(qq) - int(rng.integers(1, 30)):] = ord('#')
            elif kind == 3: k = int(rng.integers(1, 4)); bb = np.concatenate([ACGT[rng.integers(0, 4, size=k)], bb])[:len(qq)]
            elif kind == 4: k = int(rng.integers(1, 4)); bb = np.concatenate([bb[k:], ACGT[rng.integers(0, 4, size=k)]])
            elif kind == 6: j = int(rng.integers(2, 8)); bb = np.delete(bb, j); qq = qq[:len(bb)]
            elif kind == 7: j = int(rng.integers(2, 8)); bb = np.insert(bb, j, ACGT[rng.integers(0, 4)])[:len(qq)]
            elif kind == 8: lowq = rng.random(len(qq)) < 0.4; qq[lowq] = rng.integers(35, 64, size=int(lowq.sum()))
            elif kind == 9: bb = ACGT[rng.integers(0, 4, size=len(bb))]                      # this mate aligns nowhere
            mates.append((bb, qq))
        if i % 97 == 5: mates = [(m[0][:40], m[1][:40]) for m in mates]                          # both mates too short: written unaligned
        if i % 101 == 7: mates = [(ACGT[rng.integers(0, 4, size=len(m[1]))], m[1]) for m in mates]       # neither aligns
        reads += mates
n = len(reads); npairs = n // 2


def _useful(r, max_k=27):                 # SingleAligner / PairedAligner filter: clipped length >= -mrl 50, at most maxDist Ns (the smallest -d used below)
    b, q = r
    m = len(q)
    while m > 0 and q[m - 1] == ord('#'):
        m -= 1
    return m >= 50 and int((b[:m] == ord('N')).sum()) <= 8


for i in range(npairs):                   # no pair with exactly one useless mate
    if _useful(reads[2 * i]) != _useful(reads[2 * i + 1]):
        reads[2 * i] = (reads[2 * i][0][:40], reads[2 * i][1][:40]); reads[2 * i + 1] = (reads[2 * i + 1][0][:40], reads[2 * i + 1][1][:40])
for w in (0, 1):
    with open(W + '/r%d.fq' % (w + 1), 'wb') as f:
        for i in range(npairs):
            b, q = reads[2 * i + w]
            f.write(b'@p%d/%d\n' % (i, w + 1) + b.tobytes() + b'\n+\n' + q.tobytes() + b'\n')
bases = np.concatenate([r[0] for r in reads]); quals = np.concatenate([r[1] for r in reads])
offsets = np.concatenate([[0], np.cumsum([len(r[0]) for r in reads])]).astype(np.uint64)
front_clip = np.zeros(n, dtype=np.int32); data_len = np.zeros(n, dtype=np.int32)
for i, (b, q) in enumerate(reads):
    m = len(q)
    while m > 0 and q[m - 1] == ord('#'):
        m -= 1
    data_len[i] = m
out = dict(bases=bases, quals=quals, offsets=offsets, front_clip=front_clip, data_len=data_len, contig_names=np.array([c.name for c in idx.contigs]))
contig_of = {c.name: i for i, c in enumerate(idx.contigs)}
CIG = {c: i for i, c in enumerate('MIDNSHP=X')}
for tag, cli, kw, use_m in (('default', [], {}, 1), ('lvonly', ['-G-'], dict(use_affine_gap=0), 1), ('eqx', ['-='], {}, 0)):
    sam = W + '/out_%s.sam' % tag
    r = subprocess.run([ref.CLI_PATH, 'paired', W + '/idx', W + '/r1.fq', W + '/r2.fq', '-o', sam, '-t', '1'] + cli, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    p = abi.default_params(max_read_len=400, **kw); pp = abi.default_paired_params()
    useful = np.array([data_len[i] >= 50 and int((reads[i][0][:data_len[i]] == ord('N')).sum()) <= int(p.max_k) for i in range(n)])
    assert (useful[0::2] == useful[1::2]).all(), 'a pair with exactly one useless mate slipped in'
    keep = np.nonzero(useful[0::2])[0]
    kb = np.concatenate([reads[2 * i + w][0][:data_len[2 * i + w]] for i in keep for w in (0, 1)])
    kq = np.concatenate([reads[2 * i + w][1][:data_len[2 * i + w]] for i in keep for w in (0, 1)])
    ko = np.concatenate([[0], np.cumsum([data_len[2 * i + w] for i in keep for w in (0, 1)])]).astype(np.uint64)
    prim, _, _, _ = ri.align_paired(p, pp, kb, kq, ko, threads=1, stage=0)
    results = np.zeros(npairs, dtype=abi.PAIRED_RESULT_DTYPE)
    results['status'] = 0; results['location'] = 0xFFFFFFFF; results['score'] = -1
    results[keep] = prim
    F = {k: np.zeros(n, np.int64) for k in ('flag', 'contig', 'pos', 'mapq', 'nm', 'n_ops', 'rnext', 'pnext', 'tlen')}
    F['contig'][:] = -1; F['n_ops'][:] = -1
    ops = np.zeros((n, 64), np.uint32); first_written = np.full(npairs, -1, np.int32)
    seen = 0
    for line in open(sam):
        if line.startswith('@'):
            continue
        t = line.rstrip('\n').split('\t')
        pi = int(t[0][1:]); flag = int(t[1]); w = 0 if flag & 0x40 else 1
        i = 2 * pi + w; seen += 1
        if first_written[pi] < 0: first_written[pi] = w
        F['flag'][i] = flag; F['contig'][i] = contig_of.get(t[2], -1); F['pos'][i] = int(t[3]); F['mapq'][i] = int(t[4])
        F['rnext'][i] = -2 if t[6] == '=' else contig_of.get(t[6], -1); F['pnext'][i] = int(t[7]); F['tlen'][i] = int(t[8])
        F['nm'][i] = int([x for x in t[11:] if x.startswith('NM:i:')][0][5:])
        if t[5] != '*':
            num = ''; k = 0
            for ch in t[5]:
                if ch.isdigit(): num += ch
                else: ops[i, k] = (int(num) << 4) | CIG[ch]; k += 1; num = ''
            F['n_ops'][i] = k
    assert seen == n, (seen, n)
    for k, v in F.items():
        out['%s_%s' % (tag, k)] = v
    out['%s_ops' % tag] = ops; out['%s_first_written' % tag] = first_written; out['%s_results' % tag] = results; out['%s_use_m' % tag] = np.int32(use_m)
    fl = F['flag']
    print(tag, 'pairs', npairs, 'proper', int((fl & 2 != 0).sum()) // 2, 'unmapped reads', int((fl & 4 != 0).sum()), 'one mate unmapped', int(((fl & 4 != 0) ^ (fl & 8 != 0)).sum()) // 2,
          'second written first', int((first_written == 1).sum()), 'filtered pairs', npairs - len(keep))
np.savez_compressed(OUT + '/sam_fields_paired.npz', **out)
print('written', OUT + '/sam_fields_paired.npz')
