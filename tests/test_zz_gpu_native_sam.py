"""The native FASTQ -> SAM path end to end: snap_amd/snapgpu-sam (C++ host program over the C ABI: FASTQ batcher, snapgpu_align_single,
snapgpu_sam_fields_single, SAM text) must write the same file as the unmodified reference CLI (oracle/_ref/snap-aligner) -- every line
but @PG -- on the same FASTQ and index, under several option sets.

Also `-o x.bam`: the BAM header, reference table and records (decompressed BGZF) must equal the reference CLI's (SNAPLib/Bam.cpp).
First verified with the program linked against the wavefront emulator (tests/test_emu_kernels.py::test_emu_native_*), on hardware since round 2."""
import os
import subprocess

import numpy as np
import pytest

from snap_amd import synth
from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (SNAPGPU_TEST_TOOL: the same program linked against the wavefront emulator, to run this file on the host -- tests/emu/README.md)
TOOL = os.environ.get("SNAPGPU_TEST_TOOL") or os.path.join(ROOT, "snap_amd", "snapgpu-sam")


def make_workload(d, n_reads, genome_bases=600_000):
    """A genome with repeats, its index (the reference's builder) and a FASTQ with the awkward cases: ragged lengths, '#' tails, reads
    with too many Ns, unalignable reads, bases prepended / dropped at the start, names with a comment."""
    contigs = synth.make_genome(177, genome_bases, n_contigs=3, repeat_frac=0.1)
    rng0 = np.random.default_rng(7)                                   # an ALT contig: a 1 % diverged copy of a stretch of the first contig (-ea, ALT-aware scoring)
    alt = contigs[0][1][genome_bases // 20:genome_bases // 20 + genome_bases // 25].copy()
    m = rng0.random(alt.size) < 0.01; alt[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng0.integers(0, 4, size=int(m.sum()))]
    contigs.append(("alt1", alt))
    fasta = os.path.join(d, "g.fa"); synth.write_fasta(fasta, contigs)
    index_dir = os.path.join(d, "index")
    ref.build_index(fasta, index_dir, seed_len=20, threads=max(1, min(8, os.cpu_count() or 1)), extra=["-altContigName", "alt1"])
    reads = synth.make_reads(15, contigs, n_reads, 150, sub=0.015, ins=0.003, dele=0.003, n_frac=0.002)
    rng = np.random.default_rng(19)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    fastq = os.path.join(d, "r.fq")
    with open(fastq, "wb") as f:
        for i in range(n_reads):
            b = reads["bases"][i].copy(); q = reads["quals"][i].copy()
            kind = i % 40
            L = 150
            name = b"read%d" % i
            if kind == 1: L = int(rng.integers(30, 150))
            elif kind == 2: b[rng.integers(0, 150, size=16)] = ord("N")
            elif kind == 3: q[150 - int(rng.integers(1, 40)):] = ord("#")
            elif kind == 4: b = rng.choice(acgt, size=150)
            elif kind == 5: k = int(rng.integers(1, 4)); b = np.concatenate([rng.choice(acgt, size=k), b])[:150]
            elif kind == 6: k = int(rng.integers(1, 4)); b = np.concatenate([b[k:], rng.choice(acgt, size=k)])
            elif kind == 7: name += b" some comment"
            f.write(b"@" + name + b"\n" + b[:L].tobytes() + b"\n+\n" + q[:L].tobytes() + b"\n")
        k = n_reads
        for L2 in (250, 380, 60):                                      # other lengths in the same file: other affine-gap kernel variants, longer cigars
            extra = synth.make_reads(20 + L2, contigs, max(10, n_reads // 25), L2, sub=0.02, ins=0.004, dele=0.004, n_frac=0.001)
            for i in range(extra["bases"].shape[0]):
                f.write(b"@read%d\n" % k + extra["bases"][i].tobytes() + b"\n+\n" + extra["quals"][i].tobytes() + b"\n"); k += 1
    return index_dir, fastq


def sam_lines(path, mask_0x800_on_secondary=False):
    """mask_0x800_on_secondary: with -ea the reference writes `firstALTResult` as one more (secondary) record, but nothing on that path ever
    assigns firstALTResult.supplementary (BaseAligner.cpp:1041 -> fillInSingleAlignmentResult :2301-2323; the variable is an uninitialised
    stack object, SingleAligner.cpp:249), so bit 0x800 of that record is whatever the stack held.  Not compared."""
    out = []
    for line in open(path):
        if line.startswith("@PG"):
            continue
        if mask_0x800_on_secondary and not line.startswith("@"):
            t = line.split("\t")
            if int(t[1]) & 0x100:
                t[1] = str(int(t[1]) & ~0x800); line = "\t".join(t)
        out.append(line)
    return sorted(out)


def run_and_compare(tool, d, index_dir, fastq, opts, env=None, ref_opts=None):
    tag = "_".join(o.strip("-") or "eq" for o in opts) or "default"
    out_ref, out_new = os.path.join(d, "ref_%s.sam" % tag), os.path.join(d, "new_%s.sam" % tag)
    for cmd, e in (([ref.CLI_PATH, "single", index_dir, fastq, "-o", out_ref, "-t", "1"] + (opts if ref_opts is None else ref_opts), None), ([tool, "single", index_dir, fastq, "-o", out_new] + opts, env)):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=1800, env=e)
        assert r.returncode == 0, "%s failed:\n%s" % (cmd[0], r.stdout.decode(errors="replace")[-3000:])
    a, b = sam_lines(out_ref, "-ea" in opts), sam_lines(out_new, "-ea" in opts)
    assert len(a) == len(b)
    diff = [(x, y) for x, y in zip(a, b) if x != y]
    assert not diff, "%d of %d lines differ, first:\n%s%s" % (len(diff), len(a), diff[0][0], diff[0][1])
    return len(a)


@pytest.fixture(scope="module")
def single_workload(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("native"))
    return (d,) + make_workload(d, 8000, genome_bases=2_000_000)


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
@pytest.mark.parametrize("opts", [[], ["-G-"], ["-="], ["-d", "8"], ["-G-", "-=", "-d", "20"], ["-C++"], ["-om", "1", "-omax", "4"], ["-D", "2", "-om", "2", "-mpc", "2"], ["-ea", "-om", "1"], ["-ae"], ["-ae", "-om", "1"]])
def test_native_fastq_to_sam_identical_to_reference_cli(single_workload, opts):
    assert os.path.exists(TOOL), "snap_amd/snapgpu-sam not built: run __graft_entry__.build()"
    d, index_dir, fastq = single_workload
    assert run_and_compare(TOOL, d, index_dir, fastq, opts) > 8000           # (with -om: secondary records too, flag 0x100)


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
def test_native_fastq_to_sam_with_page_locked_group_buffers(single_workload):
    """The tool registers the buffers it hands to snapgpu_align_sam_single with the HIP runtime (hipHostRegister) when they are big; here
    every one is (SNAPGPU_SAM_PIN_MIN=1), over groups of two 1 000-read batches and two feeders -- and with the registration off."""
    d, index_dir, fastq = single_workload
    for env in (dict(os.environ, SNAPGPU_SAM_PIN_MIN="1"), dict(os.environ, SNAPGPU_SAM_PIN="0")):
        assert run_and_compare(TOOL, d, index_dir, fastq, ["-b", "1000", "-g", "2", "-q", "2"], env=env, ref_opts=[]) > 8000


def make_paired_workload(d, n_pairs, genome_bases=600_000):
    from tests.pairs_util import hard_pairs
    contigs = synth.make_genome(178, genome_bases, n_contigs=3, repeat_frac=0.15)
    fasta = os.path.join(d, "g.fa"); synth.write_fasta(fasta, contigs)
    index_dir = os.path.join(d, "index")
    ref.build_index(fasta, index_dir, seed_len=20, threads=max(1, min(8, os.cpu_count() or 1)))
    pr = hard_pairs(27, contigs, n_pairs, 150, insert_mean=380)
    o = pr["offsets"].astype(np.int64)
    rng = np.random.default_rng(29)
    fq = [os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")]
    files = [open(f, "wb") for f in fq]
    for i in range(n_pairs):
        short = i % 53 == 7                                    # both mates below -mrl: the pair is written unaligned
        for w in (0, 1):
            b = pr["bases"][o[2 * i + w]:o[2 * i + w + 1]].copy(); q = pr["quals"][o[2 * i + w]:o[2 * i + w + 1]].copy()
            if (2 * i + w) % 19 == 3: q[len(q) - int(rng.integers(1, 30)):] = ord("#")
            if short: b, q = b[:40], q[:40]
            if i % 59 == 11 and w == 1: b, q = b[:35], q[:35]                                   # exactly one useless mate: the pair is still aligned (PairedAligner.cpp:680-682)
            if i % 61 == 13 and w == 0: b[rng.integers(0, len(b), size=min(40, len(b)))] = ord("N")
            files[w].write(b"@pair%d/%d\n" % (i, w + 1) + b.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    for f in files:
        f.close()
    return index_dir, fq


def make_contig_start_workload(d, reps, seed=5):
    """Pairs whose one mate is the reverse complement of a contig's FIRST 150 bases (its record: flag 0x10, POS 1) while the other lies a few hundred
    bases further on in an orientation that is no proper pair, over a genome in which the start of two contigs also occurs, slightly mutated,
    inside another one: under -om the read has further records after the one at POS 1 -- the case in which the record alone does not say what
    back clipping it left on the Read (snapgpu_sam.cpp: leave_behind), which stopped the program until round 4."""
    rng = np.random.default_rng(seed)
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    acgt = np.frombuffer(b"ACGT", np.uint8)
    g = [(n, b.copy()) for n, b in synth.make_genome(11, 300_000, n_contigs=3, repeat_frac=0.0)]
    for ci, at in ((1, 50_000), (2, 70_000)):
        blk = g[ci][1][:700].copy()
        mut = rng.random(700) < 0.01
        blk[mut] = acgt[rng.integers(0, 4, size=int(mut.sum()))]
        g[0][1][at:at + 700] = blk
    fasta = os.path.join(d, "g.fa"); synth.write_fasta(fasta, g)
    index_dir = os.path.join(d, "index")
    ref.build_index(fasta, index_dir, seed_len=20, threads=max(1, min(8, os.cpu_count() or 1)))

    def mutate(seq, nsub):
        s = seq.copy()
        for j in rng.integers(20, len(s) - 20, size=nsub):
            s[j] = acgt[(np.searchsorted(np.sort(acgt), s[j]) + 1) % 4]
        return s
    fq = [os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")]
    n = 0
    with open(fq[0], "wb") as f1, open(fq[1], "wb") as f2:
        for ci in (1, 2):
            c = g[ci][1]
            for rep in range(reps):
                nsub, li = int(rng.integers(1, 4)), int(rng.integers(0, 3))
                left = mutate(c[0:150], nsub)
                if li:                                                      # extra bases at the reference-left end: a leading insertion of the RC alignment
                    left = np.concatenate([acgt[rng.integers(0, 4, size=li)], left[:150 - li]])
                b = comp[left[::-1]]
                off = int(rng.integers(260, 420))
                a = mutate(c[off:off + 150], int(rng.integers(0, 3)))
                if rep % 2: a = comp[a[::-1]]
                x, y = (a, b) if rep % 4 < 2 else (b, a)
                q = b"I" * 150
                f1.write(b"@c%d/1\n" % n + x.tobytes() + b"\n+\n" + q + b"\n"); f2.write(b"@c%d/2\n" % n + y.tobytes() + b"\n+\n" + q + b"\n"); n += 1
    return index_dir, fq


def run_and_compare_paired(tool, d, index_dir, fq, opts, env=None):
    tag = "_".join(o.strip("-") or "eq" for o in opts) or "default"
    out_ref, out_new = os.path.join(d, "pref_%s.sam" % tag), os.path.join(d, "pnew_%s.sam" % tag)
    for cmd, e in (([ref.CLI_PATH, "paired", index_dir, fq[0], fq[1], "-o", out_ref, "-t", "1"] + opts, None), ([tool, "paired", index_dir, fq[0], fq[1], "-o", out_new] + opts, env)):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=3000, env=e)
        assert r.returncode == 0, "%s failed:\n%s" % (cmd[0], r.stdout.decode(errors="replace")[-3000:])
    a = [line for line in open(out_ref) if not line.startswith("@PG")]          # not sorted: with -t 1 the reference keeps the input order,
    b = [line for line in open(out_new) if not line.startswith("@PG")]          # and the order of the two records of a pair is part of the contract
    assert len(a) == len(b)
    diff = [(x, y) for x, y in zip(a, b) if x != y]
    assert not diff, "%d of %d lines differ, first:\n%s%s" % (len(diff), len(a), diff[0][0], diff[0][1])
    return len(a)


@pytest.fixture(scope="module")
def paired_workload(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("nativep"))
    return (d,) + make_paired_workload(d, 2500, genome_bases=2_000_000)


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
@pytest.mark.parametrize("opts", [[], ["-G-"], ["-="]])
def test_native_paired_fastq_to_sam_identical_to_reference_cli(paired_workload, opts):
    assert os.path.exists(TOOL), "snap_amd/snapgpu-sam not built: run __graft_entry__.build()"
    d, index_dir, fq = paired_workload
    assert run_and_compare_paired(TOOL, d, index_dir, fq, opts) > 5000


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
@pytest.mark.parametrize("opts", [["-om", "1"], ["-D", "2", "-om", "2", "-omax", "3", "-mpc", "2", "-="]])
def test_native_paired_secondary_records_identical_to_reference_cli(paired_workload, opts):
    """-om with `paired`: every further PairedAlignmentResult of a pair as two more records (flag 0x100), then each mate's single-end secondary
    results as records of their own -- unpaired flags, the whole read id -- in the order SimpleReadWriter::writePairs writes them
    (ReadWriter.cpp:345-590).  The workload has the reads whose earlier record leaves additional back clipping on the Read, incl. the ones
    where a later record of the same read clips a leading insertion again (`-om 1`): those go through the host record loop."""
    d, index_dir, fq = paired_workload
    n = run_and_compare_paired(TOOL, d, index_dir, fq, opts)
    tag = "_".join(o.strip("-") or "eq" for o in opts)
    flags = [int(line.split("\t")[1]) for line in open(os.path.join(d, "pnew_%s.sam" % tag)) if not line.startswith("@")]
    assert n > 5000 and any(f & 0x100 and f & 0x1 for f in flags) and any(f & 0x100 and not f & 0x1 for f in flags)      # both kinds are in the workload


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
def test_native_paired_secondary_records_host_record_loop(tmp_path):
    """SNAPGPU_SAM_HOST_LOOP=1: every record of a multi-result batch through the host record loop (the path a read takes when an earlier record
    left back clipping on it) instead of the device's: same file as the reference CLI, so the two loops agree record for record."""
    d = str(tmp_path)
    index_dir, fq = make_paired_workload(d, 800, genome_bases=400_000)
    assert run_and_compare_paired(TOOL, d, index_dir, fq, ["-om", "1"], env=dict(os.environ, SNAPGPU_SAM_HOST_LOOP="1")) > 2000


def make_paired_alt_workload(d, n_pairs, genome_bases=500_000):
    """Pairs over a genome whose first contig has a 1 % diverged ALT copy of one stretch (index built with -altContigName)."""
    contigs = synth.make_genome(177, genome_bases, n_contigs=2, repeat_frac=0.1)
    rng0 = np.random.default_rng(7)
    alt = contigs[0][1][genome_bases // 20:genome_bases // 20 + genome_bases // 10].copy()
    m = rng0.random(alt.size) < 0.01; alt[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng0.integers(0, 4, size=int(m.sum()))]
    contigs.append(("alt1", alt))
    fasta = os.path.join(d, "g.fa"); synth.write_fasta(fasta, contigs)
    index_dir = os.path.join(d, "index")
    ref.build_index(fasta, index_dir, seed_len=20, threads=max(1, min(8, os.cpu_count() or 1)), extra=["-altContigName", "alt1"])
    p = synth.make_pairs(5, [contigs[0], contigs[2]], n_pairs, 150)
    fq = [os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")]
    with open(fq[0], "wb") as f1, open(fq[1], "wb") as f2:
        for k in range(n_pairs):
            f1.write(b"@p%d/1\n" % k + p["bases"][2 * k].tobytes() + b"\n+\n" + p["quals"][2 * k].tobytes() + b"\n")
            f2.write(b"@p%d/2\n" % k + p["bases"][2 * k + 1].tobytes() + b"\n+\n" + p["quals"][2 * k + 1].tobytes() + b"\n")
    return index_dir, fq


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
def test_native_paired_first_alt_records_identical_to_reference_cli(tmp_path):
    """-ea with `paired`: the first ALT result of a pair is written as a pair of its own after the pair's other records (PairedAligner.cpp:877-879)."""
    d = str(tmp_path)
    index_dir, fq = make_paired_alt_workload(d, 1500)
    assert run_and_compare_paired(TOOL, d, index_dir, fq, ["-ea"]) > 3000


# ---------------------------------------------------------------------------------------- BAM (-o x.bam)
def bam_parts(path):
    """(header text without @PG, [(name, length)] of the reference table, [record bytes]) of a BAM file; BGZF is concatenated gzip members."""
    import gzip
    import struct
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\1"
    lt = struct.unpack_from("<i", raw, 4)[0]
    text = b"\n".join(l for l in raw[8:8 + lt].split(b"\n") if not l.startswith(b"@PG"))
    at = 8 + lt
    n_ref = struct.unpack_from("<i", raw, at)[0]; at += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, at)[0]
        refs.append((raw[at + 4:at + 4 + ln], struct.unpack_from("<i", raw, at + 4 + ln)[0])); at += 8 + ln
    recs = []
    while at < len(raw):
        bs = struct.unpack_from("<i", raw, at)[0]
        recs.append(raw[at:at + 4 + bs]); at += 4 + bs
    return text, refs, recs


def run_and_compare_bam(tool, d, mode, index_dir, fqs, opts, env=None):
    """Both programs write BAM; header text, reference table and every record (bin, cigar ops, 4-bit SEQ, QUAL, aux tags ...) must be the
    same bytes, in the same order (`-t 1` keeps the input order in the reference)."""
    tag = mode + "_" + ("_".join(o.strip("-") or "eq" for o in opts) or "default")
    out_ref, out_new = os.path.join(d, "ref_%s.bam" % tag), os.path.join(d, "new_%s.bam" % tag)
    for cmd, e in (([ref.CLI_PATH, mode, index_dir] + fqs + ["-o", out_ref, "-t", "1"] + opts, None), ([tool, mode, index_dir] + fqs + ["-o", out_new] + opts, env)):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=3000, env=e)
        assert r.returncode == 0, "%s failed:\n%s" % (cmd[0], r.stdout.decode(errors="replace")[-3000:])
    a, b = bam_parts(out_ref), bam_parts(out_new)
    assert a[0] == b[0] and a[1] == b[1]
    assert len(a[2]) == len(b[2])
    diff = [k for k, (x, y) in enumerate(zip(a[2], b[2])) if x != y]
    assert not diff, "%d of %d BAM records differ, first: %r vs %r" % (len(diff), len(a[2]), a[2][diff[0]][:80], b[2][diff[0]][:80])
    return len(a[2])


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
@pytest.mark.parametrize("opts", [[], ["-=", "-om", "1", "-omax", "3"]])
def test_native_fastq_to_bam_identical_to_reference_cli(single_workload, opts):
    d, index_dir, fastq = single_workload
    assert run_and_compare_bam(TOOL, d, "single", index_dir, [fastq], opts) > 8000


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
def test_native_paired_fastq_to_bam_identical_to_reference_cli(paired_workload):
    d, index_dir, fq = paired_workload
    assert run_and_compare_bam(TOOL, d, "paired", index_dir, fq, []) == 5000
