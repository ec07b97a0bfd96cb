"""The GPU index builder (include/snapgpu.h: snapgpu_index_build*) on the hardware, against the reference's own
`snap-aligner index` on the same FASTA: same `Genome` file, same table sizes as the reference's -exact build, the reference answers every
probed seed identically over both directories and aligns reads identically over both (tests/index_build_util.py).  Then the product
path over the built index -- from the files and straight from HBM -- against the reference."""
import os
import subprocess

import numpy as np
import pytest

from snap_amd import abi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _need_ref():
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built")
    return ref


@pytest.mark.parametrize("seed_len,kw,extra", [(20, {}, []), (18, {}, ["-locationSize", "4"]), (22, dict(key_bytes=4), ["-keysize", "4"]),
                                               # key sizes other than 4 (index_build.h: k_ib_insert_wide); -s 24 -> 5 is the reference's default shape (GenomeIndex.cpp:437)
                                               (24, {}, []), (26, {}, []), (17, {}, ["-locationSize", "4"])])
def test_built_directory_equals_the_reference_indexer(tmp_path, seed_len, kw, extra):
    _need_ref()
    from tests.index_build_util import compare_with_reference
    stats, _, _ = compare_with_reference(tmp_path, seed_len=seed_len, extra_ref=extra, **kw)
    assert stats["n_repeated_seeds"] > 0


def test_larger_genome_and_the_product_path_over_the_built_index(tmp_path):
    """16 Mb with planted repeats (thousands of tiles per pass, overflow lists up to hundreds of hits): the directory test again, then
    BaseAligner over the built index (files, and the HBM-resident view) must equal the reference over the REFERENCE-built directory."""
    ref = _need_ref()
    from tests.index_build_util import compare_with_reference
    from snap_amd.aligner import BaseAligner
    from snap_amd.index import GenomeIndex, build_index
    g = synth.make_genome(31, 16_000_000, n_contigs=5, repeat_frac=0.3, max_copies=800, repeat_len=(200, 3000), max_divergence=0.05, n_run_frac=0.001)
    fasta = os.path.join(str(tmp_path), "g.fa")
    synth.write_fasta(fasta, g)
    stats, d_ref, d_gpu = compare_with_reference(tmp_path, fasta=fasta, n_reads=4000)
    reads = synth.make_reads(5, g, 20000, 150)
    params = abi.default_params(max_k=8, max_read_len=160)
    with ref.fresh_objects():
        exp = ref.RefIndex(d_ref).align_single(params, reads["bases"], reads["quals"], reads["offsets"], threads=16)[0]
    a = BaseAligner(GenomeIndex.load_from_directory(d_gpu), params)
    got, _ = a.AlignRead(reads["bases"], reads["quals"], reads["offsets"])
    a.close()
    assert not util.compare_results(exp, got), util.compare_results(exp, got)
    st2, built = build_index(fasta, None, keep=True)
    assert st2["n_distinct_seeds"] == stats["n_distinct_seeds"] and st2["overflow_table_size"] == stats["overflow_table_size"]
    b = BaseAligner.from_built_index(built, None, params)
    got2, _ = b.AlignRead(reads["bases"], reads["quals"], reads["offsets"])
    b.close(); built.close()
    assert not util.compare_results(exp, got2)


def test_snapgpu_index_command_line(tmp_path):
    """`snapgpu-index <fasta> <dir> -s 20` (snap_amd/csrc/host/snapgpu_index.cpp): the directory it writes is loaded by the reference CLI,
    whose SAM over it equals its SAM over its own index."""
    ref = _need_ref()
    from tests.index_build_util import hard_fasta
    tool = os.path.join(util.ROOT, "snap_amd", "snapgpu-index")
    if not os.path.exists(tool):
        pytest.skip("snap_amd/snapgpu-index not built")
    fasta = os.path.join(str(tmp_path), "g.fa")
    contigs = hard_fasta(fasta)
    d_ref, d_gpu = os.path.join(str(tmp_path), "r"), os.path.join(str(tmp_path), "g")
    ref.build_index(fasta, d_ref, 20, threads=8)
    r = subprocess.run([tool, fasta, d_gpu, "-s", "20"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    reads = synth.make_reads(9, contigs, 3000, 100)
    fq = os.path.join(str(tmp_path), "r.fq")
    synth.write_fastq(fq, reads)
    outs = []
    for d in (d_ref, d_gpu):
        sam = os.path.join(str(tmp_path), os.path.basename(d) + ".sam")
        r = subprocess.run([ref.CLI_PATH, "single", d, fq, "-t", "1", "-d", "8", "-o", sam], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout.decode()
        outs.append([l for l in open(sam) if not l.startswith("@PG")])
    assert outs[0] == outs[1]
