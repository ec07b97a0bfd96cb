"""Drop-in test at the outermost boundary: SNAP's own CLI with shim/GpuAlignerExtension.cpp
installed (oracle/_ref/snap-aligner-gpu: reference readers, filters and SAM writer, alignment by
libsnapgpu.so through the C ABI) must write the same SAM records as the unmodified reference CLI
(oracle/_ref/snap-aligner) on the same FASTQ and index.  Both binaries are prebuilt in the
container (oracle/Makefile) and travel with the snapshot; nothing here reads /root/reference."""
import os
import subprocess

import numpy as np
import pytest

from snap_amd import abi, synth
from snap_amd.index import GenomeIndex
from oracle import ref

pytestmark = pytest.mark.gpu

REF_CLI = ref.CLI_PATH
# (SNAPGPU_TEST_SHIM: the same objects linked against the wavefront emulator's build of the C ABI -- tests/emu/README.md -- to run this file on the host)
GPU_CLI = os.environ.get("SNAPGPU_TEST_SHIM") or os.path.join(os.path.dirname(REF_CLI), "snap-aligner-gpu")


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=600)
    assert r.returncode == 0, "%s failed:\n%s" % (cmd[0], r.stdout.decode(errors="replace")[-3000:])
    return r.stdout.decode(errors="replace")


def _sam(path):
    header, records = [], []
    for line in open(path):
        if line.startswith("@"):
            if not line.startswith("@PG"):                  # @PG carries the command line
                header.append(line)
        else:
            records.append(line)
    return header, sorted(records)


@pytest.fixture(scope="module")
def workload(tmp_path_factory):
    if not (os.path.exists(REF_CLI) and os.path.exists(GPU_CLI)):
        pytest.skip("oracle/_ref CLIs were not built (run __graft_entry__.build() in the container)")
    d = str(tmp_path_factory.mktemp("shim"))
    contigs = synth.make_genome(77, 3_000_000, n_contigs=3, repeat_frac=0.08)
    fasta = os.path.join(d, "g.fa")
    synth.write_fasta(fasta, contigs)
    index_dir = os.path.join(d, "index")
    ref.build_index(fasta, index_dir, seed_len=20, threads=8)
    reads = synth.make_reads(5, contigs, 20000, 150, sub=0.015, ins=0.002, dele=0.002, n_frac=0.002)
    rng = np.random.default_rng(9)
    fastq = os.path.join(d, "r.fq")
    names, seqs = [], []
    with open(fastq, "wb") as f:
        for i in range(reads["bases"].shape[0]):
            b = reads["bases"][i].copy(); q = reads["quals"][i].copy()
            kind = i % 50
            L = 150
            if kind == 1:
                L = int(rng.integers(30, 150))              # ragged lengths, some below -mrl 50
            elif kind == 2:
                b[rng.integers(0, 150, size=12)] = ord("N")   # more Ns than -d: "useless" read
            elif kind == 3:
                q[150 - int(rng.integers(1, 40)):] = ord("#")   # '#' tail: back-clipped by the reader
            elif kind == 4:
                b = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=150)   # unalignable
            f.write(b"@read%d\n" % i + b[:L].tobytes() + b"\n+\n" + q[:L].tobytes() + b"\n")
            names.append("read%d" % i); seqs.append((b[:L], q[:L]))
    return dict(dir=d, index=index_dir, fastq=fastq, names=names, seqs=seqs)


def _unstable_names(workload, extra_params):
    """Reads whose reference result depends on the aligner object's history (DESIGN.md
    'Reference nondeterminism'): the device flags them in `reserved`."""
    from snap_amd.aligner import BaseAligner
    idx = GenomeIndex.load_from_directory(workload["index"])
    extra_params = dict(extra_params)
    secondary = extra_params.pop("secondary", None)         # (-om, -omax, -mpc)
    flags = extra_params.pop("flags", None)                 # (-f, -x)
    max_hits = extra_params.pop("max_hits", None)
    p = abi.default_params(max_read_len=400, **extra_params)
    if max_hits is not None:
        p.max_hits = max_hits
    a = BaseAligner(idx, p)
    if flags:
        a.set_flags(stop_on_first_hit=flags[0], explore_popular_seeds=flags[1])
    def clip(s):                                            # ClipBack (the CLI default, -C-+): drop the trailing '#' run
        n = len(s[1])
        while n > 0 and s[1][n - 1] == ord("#"):
            n -= 1
        return s[0][:n], s[1][:n]
    keep = [(n, clip(s)) for n, s in zip(workload["names"], workload["seqs"])]
    keep = [(n, s) for n, s in keep if len(s[0]) >= 50]
    bases = np.concatenate([s[0] for _, s in keep]); quals = np.concatenate([s[1] for _, s in keep])
    offs = np.concatenate([[0], np.cumsum([len(s[0]) for _, s in keep])]).astype(np.uint64)
    if secondary:
        a.enable_secondary(secondary[0], max_results=secondary[1], max_per_contig=secondary[2])
        prim = a.AlignReadSecondary(bases, quals, offs)[0]
    else:
        prim, alt = a.AlignRead(bases, quals, offs)
    a.close()
    return {n for (n, _), r in zip(keep, prim["reserved"]) if r != 0}


@pytest.mark.parametrize("opts,params", [
    ([], {}),
    (["-d", "12", "-G-"], {"max_k": 12, "use_affine_gap": 0}),
    (["-ea", "-D", "2"], {"emit_alt_alignments": 1, "extra_search_depth": 2}),
    (["-om", "1", "-omax", "4"], {"secondary": (1, 4, -1)}),                 # secondary alignments (SingleAligner.cpp:250-318)
    (["-D", "2", "-om", "2", "-mpc", "2"], {"extra_search_depth": 2, "secondary": (2, 0x7fffffff, 2)}),
    (["-ae"], {}),                                                            # AlignmentAdjuster on the primary (BaseAligner.cpp:2444-2452)
    (["-ae", "-om", "1"], {"secondary": (1, 0x7fffffff, -1)}),                # ... and on every secondary result, before the -om filter
    (["-f"], {"flags": (True, False)}),                                       # stopOnFirstHit (snapgpu_set_aligner_flags, round 4)
    (["-x", "-h", "20"], {"flags": (False, True), "max_hits": 20}),           # explorePopularSeeds, with a -h low enough for it to matter
])
def test_sam_identical_to_reference_cli(workload, opts, params):
    d = workload["dir"]
    tag = "_".join(o.strip("-") for o in opts) or "default"
    out_ref = os.path.join(d, "ref_%s.sam" % tag)
    out_gpu = os.path.join(d, "gpu_%s.sam" % tag)
    _run([REF_CLI, "single", workload["index"], workload["fastq"], "-o", out_ref, "-t", "8"] + opts)
    log = _run([GPU_CLI, "single", workload["index"], workload["fastq"], "-o", out_gpu, "-t", "4"] + opts)
    h_ref, r_ref = _sam(out_ref)
    h_gpu, r_gpu = _sam(out_gpu)
    assert h_ref == h_gpu
    assert len(r_ref) == len(r_gpu) and len(r_ref) >= len(workload["names"]), log[-2000:]
    if r_ref != r_gpu:
        def by_name(records):
            m = {}
            for line in records:
                m.setdefault(line.split("\t")[0], []).append(line)
            return m
        m_ref, m_gpu = by_name(r_ref), by_name(r_gpu)
        assert m_ref.keys() == m_gpu.keys()
        differing = [n for n in m_ref if m_ref[n] != m_gpu[n]]
        unstable = _unstable_names(workload, params)
        bad = [n for n in differing if n not in unstable]
        assert not bad, "first differing read %s:\nref: %sgpu: %s" % (bad[0], "".join(m_ref[bad[0]]), "".join(m_gpu[bad[0]]))
        assert len(differing) <= 1 + len(m_ref) // 2000


def test_shim_fails_loudly_on_unsupported_option(workload):
    # (-ins, inferSpacing, is not implemented for the paired-end path: the shim must say so, not run the reference's aligner instead)
    r = subprocess.run([GPU_CLI, "paired", workload["index"], workload["fastq"], workload["fastq"], "-o", os.path.join(workload["dir"], "x.sam"),
                        "-ins"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=300)
    assert r.returncode != 0
    assert b"libsnapgpu" in r.stdout


# ---------------------------------------------------------------------------------------- paired end

@pytest.fixture(scope="module")
def paired_workload(workload):
    from tests.pairs_util import hard_pairs
    d = workload["dir"]
    contigs = synth.make_genome(77, 3_000_000, n_contigs=3, repeat_frac=0.08)        # the genome of `workload`
    pr = hard_pairs(11, contigs, 6000, 150, insert_mean=380)
    o = pr["offsets"].astype(np.int64)
    fq = [os.path.join(d, "p1.fq"), os.path.join(d, "p2.fq")]
    with open(fq[0], "wb") as f0, open(fq[1], "wb") as f1:
        for i in range(o.size // 2):
            for r, f in ((0, f0), (1, f1)):
                s, e = o[2 * i + r], o[2 * i + r + 1]
                f.write(b"@pair%d\n" % i + pr["bases"][s:e].tobytes() + b"\n+\n" + pr["quals"][s:e].tobytes() + b"\n")
    return dict(dir=d, index=workload["index"], fq=fq, pairs=pr)


@pytest.mark.parametrize("opts,params,pparams", [
    ([], {}, {}),
    (["-d", "10", "-s", "50", "800", "-H", "2000"], {"max_k": 10}, {"min_spacing": 50, "max_spacing": 800, "max_big_hits": 2000}),
    (["-om", "1", "-omax", "3"], {"secondary": (1, 3, -1)}, {}),            # paired + single-end secondary alignments (PairedAligner.cpp:843-890)
])
def test_paired_sam_identical_to_reference_cli(paired_workload, opts, params, pparams):
    w = paired_workload
    d = w["dir"]
    tag = "_".join(o.strip("-") for o in opts) or "default"
    out_ref = os.path.join(d, "pref_%s.sam" % tag)
    out_gpu = os.path.join(d, "pgpu_%s.sam" % tag)
    _run([REF_CLI, "paired", w["index"], w["fq"][0], w["fq"][1], "-o", out_ref, "-t", "8"] + opts)
    log = _run([GPU_CLI, "paired", w["index"], w["fq"][0], w["fq"][1], "-o", out_gpu, "-t", "4"] + opts)
    h_ref, r_ref = _sam(out_ref)
    h_gpu, r_gpu = _sam(out_gpu)
    assert h_ref == h_gpu
    assert min(len(r_ref), len(r_gpu)) >= 2 * 6000, log[-2000:]       # (record counts are compared per pair below)
    if r_ref != r_gpu:
        def by_name(records):
            m = {}
            for line in records:
                m.setdefault(line.split("\t")[0], []).append(line)
            return m
        m_ref, m_gpu = by_name(r_ref), by_name(r_gpu)
        assert m_ref.keys() == m_gpu.keys()
        differing = [n for n in m_ref if m_ref[n] != m_gpu[n]]
        # pairs the device flags as depending on the reference aligner's history (DESIGN.md "Reference nondeterminism")
        from snap_amd.aligner import ChimericPairedEndAligner
        params = dict(params)
        secondary = params.pop("secondary", None)
        a = ChimericPairedEndAligner(GenomeIndex.load_from_directory(w["index"]), abi.default_params(max_read_len=400, **params),
                                     abi.default_paired_params(**pparams))
        if secondary:       # ... or on the size its secondary-result buffer happened to have (SNAPGPU_PAIR_REF_BUFFER_DEPENDENT)
            a.enable_secondary(secondary[0], max_results=secondary[1], max_per_contig=secondary[2])
            prim = a.align_secondary(w["pairs"]["bases"], w["pairs"]["quals"], w["pairs"]["offsets"])[0]
        else:
            prim, _ = a.align(w["pairs"]["bases"], w["pairs"]["quals"], w["pairs"]["offsets"])
        a.close()
        unstable = {"pair%d" % i for i in np.nonzero((prim["reserved"] != 0) | ((prim["flags"] & 2) != 0))[0]}
        bad = [n for n in differing if n not in unstable]
        assert not bad, "first differing pair %s:\nref: %sgpu: %s" % (bad[0], "".join(m_ref[bad[0]]), "".join(m_gpu[bad[0]]))
        assert len(differing) <= 2 + len(m_ref) // 50


def test_paired_sam_identical_on_alt_liftover_index(tmp_path):
    """An index built with -altLiftoverFile, pairs drawn from its ALT contigs: exercises snapgpu_create_from_directory's parsing of the
    projection data (Genome.cpp:353-403) and the reference SAM writer on lifted-over results."""
    from tests.pairs_util import alt_liftover_genome, hard_pairs
    if not (os.path.exists(REF_CLI) and os.path.exists(GPU_CLI)):
        pytest.skip("oracle/_ref CLIs were not built")
    d = str(tmp_path)
    g, sam, alt_args = alt_liftover_genome()
    synth.write_fasta(d + "/ref.fa", g)
    open(d + "/lift.sam", "w").write(sam)
    ref.build_index(d + "/ref.fa", d + "/idx", 20, threads=8, extra=alt_args + ["-altLiftoverFile", d + "/lift.sam"])
    pa = hard_pairs(21, g[3:], 1200, 150, insert_mean=380)
    pb = hard_pairs(22, g, 1200, 150, insert_mean=380)
    fq = [d + "/p1.fq", d + "/p2.fq"]
    with open(fq[0], "wb") as f0, open(fq[1], "wb") as f1:
        i = 0
        for pr in (pa, pb):
            o = pr["offsets"].astype(np.int64)
            for j in range(o.size // 2):
                for r, f in ((0, f0), (1, f1)):
                    s, e = o[2 * j + r], o[2 * j + r + 1]
                    f.write(b"@pair%d\n" % i + pr["bases"][s:e].tobytes() + b"\n+\n" + pr["quals"][s:e].tobytes() + b"\n")
                i += 1
    _run([REF_CLI, "paired", d + "/idx", fq[0], fq[1], "-o", d + "/ref.sam", "-t", "8", "-d", "8"])
    _run([GPU_CLI, "paired", d + "/idx", fq[0], fq[1], "-o", d + "/gpu.sam", "-t", "4", "-d", "8"])
    h_ref, r_ref = _sam(d + "/ref.sam")
    h_gpu, r_gpu = _sam(d + "/gpu.sam")
    assert h_ref == h_gpu
    differing = sum(a != b for a, b in zip(r_ref, r_gpu))
    assert len(r_ref) == len(r_gpu) and differing <= 2 + len(r_ref) // 100, "first differing records:\n%s%s" % next(
        ((a, b) for a, b in zip(r_ref, r_gpu) if a != b), ("", ""))
