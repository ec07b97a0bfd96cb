"""Generate tests/golden/paired_secondary.npz with the compiled reference (oracle/_ref): ChimericPairedEndAligner::align called as
PairedAligner.cpp:727 calls it with -om / -omax / -mpc, for the pairs of tests/golden/paired_reads.npz against the index of
tests/golden/paired_index.npz (rebuilt here as a directory: the reference loads directories).  Primary, firstALT, the paired
secondary results and the single-end secondary results of the chimeric fallback.  Run in the build container."""
import os, sys, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from oracle import ref

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
W = '/tmp/snap_golden_psec'
shutil.rmtree(W, ignore_errors=True); os.makedirs(W)
g = synth.make_genome(20260926, 240_000, n_contigs=3, repeat_frac=0.35, max_copies=40, repeat_len=(150, 1500), n_run_frac=0.003)
synth.write_fasta(W + '/ref.fa', g)
ref.build_index(W + '/ref.fa', W + '/idx', 20, threads=4)
idx = GenomeIndex.load_from_directory(W + '/idx')
z = np.load(OUT + '/paired_index.npz')
assert np.array_equal(z['genome_padded'], idx.genome_padded) and np.array_equal(z['contig_begin'], idx.contig_begin)
ri = ref.RefIndex(W + '/idx')
rd = np.load(OUT + '/paired_reads.npz')

# (name, BaseAligner options, paired options, -om, -omax, -mpc)
sets = [
    ('om1_d8',           dict(max_k=8), {}, 1, 0x7fffffff, -1),
    ('om3_D3_d12_omax3', dict(max_k=12, extra_search_depth=3), {}, 3, 3, -1),
    ('om3_D3_d12_mpc2',  dict(max_k=12, extra_search_depth=3), {}, 3, 0x7fffffff, 2),
    ('om2_D2_lvonly',    dict(max_k=10, extra_search_depth=2, use_affine_gap=0), {}, 2, 5, 3),
    ('om0_spacing_mpc1', dict(max_k=8), dict(min_spacing=100, max_spacing=600, num_seeds=12), 0, 0x7fffffff, 1),
]
PFIELDS = ('status', 'location', 'score', 'mapq', 'ag_score', 'direction')
out = {}
for name, kw, pkw, om, omax, mpc in sets:
    p = abi.default_params(max_read_len=160, **kw)
    pp = abi.default_paired_params(**pkw)
    for tag in ('150', '100'):
        b, q, o = rd['b' + tag], rd['q' + tag], rd['o' + tag]
        if tag == '150':                                    # 600 of the 1500 hard pairs keep the fixture small
            o = o[:1201]; b = b[:int(o[-1])]; q = q[:int(o[-1])]
        R = ri.align_paired_secondary(p, pp, om, b, q, o, omax=omax, mpc=mpc, stage=0, threads=1)
        prim, alt, sec, nsec, ssec, nssec = R
        n = prim.size
        # what the reference answers differently when the same thread saw other pairs before (stale affine-gap traceback cells, the
        # size its secondary buffers had grown to, ChimericPairedEndAligner.cpp:339): vary the history and mark what moves
        unstable = np.zeros(n, bool)
        lens = np.diff(o.astype(np.int64)); starts = o[:-1].astype(np.int64)
        for order, th in ((np.arange(n)[::-1].copy(), 1), (np.random.default_rng(5).permutation(n), 3)):
            ro = np.empty(2 * n, dtype=np.int64); ro[0::2] = 2 * order; ro[1::2] = 2 * order + 1
            bb = np.concatenate([b[starts[i]:starts[i] + lens[i]] for i in ro]); qq = np.concatenate([q[starts[i]:starts[i] + lens[i]] for i in ro])
            oo = np.concatenate([[0], np.cumsum(lens[ro])]).astype(np.uint64)
            V = ri.align_paired_secondary(p, pp, om, bb, qq, oo, omax=omax, mpc=mpc, stage=0, threads=th, stride=sec.shape[1], single_stride=ssec.shape[1])
            back = [np.zeros_like(x) for x in R]
            for k in range(6):
                w = min(V[k].shape[1], R[k].shape[1]) if V[k].ndim == 2 and k in (2, 4) else None
                if w is None: back[k][order] = V[k]
                else: back[k][order, :w] = V[k][:, :w]
            for f in PFIELDS:
                m = (prim[f] != back[0][f])
                if f != 'status': m &= prim['status'] != 0
                unstable |= m.any(axis=1)
            unstable |= (nsec != back[3]) | (nssec != back[5]).any(axis=1)
            for f in ('location', 'score', 'direction', 'ag_score', 'match_probability'):
                unstable |= (sec[f] != back[2][f]).any(axis=(1, 2))
            for f in ('location', 'score', 'direction', 'ag_score', 'match_probability'):
                unstable |= (ssec[f] != back[4][f]).any(axis=1)
        print(name, tag, 'pairs with paired secondaries', int((nsec > 0).sum()), 'total', int(nsec.sum()), '| single-end secondaries', int(nssec.sum()),
              '| reference-unstable', np.nonzero(unstable)[0].tolist())
        key = '%s_%s_' % (name, tag)
        out[key + 'primary'] = prim; out[key + 'alt'] = alt
        out[key + 'secondary'] = sec[:, :max(1, int(nsec.max()))].copy(); out[key + 'nsec'] = nsec
        out[key + 'single_secondary'] = ssec[:, :max(1, int(nssec.sum(axis=1).max()))].copy(); out[key + 'nssec'] = nssec
        out[key + 'unstable'] = unstable
out['sets'] = np.array([[s[0], repr(s[1]), repr(s[2]), str(s[3]), str(s[4]), str(s[5])] for s in sets])
np.savez_compressed(OUT + '/paired_secondary.npz', **out)
print('wrote', OUT + '/paired_secondary.npz', os.path.getsize(OUT + '/paired_secondary.npz'))
