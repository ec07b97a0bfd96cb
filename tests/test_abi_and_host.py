"""CPU tests: the C-ABI library loads and exports what include/snapgpu.h declares, it refuses to
run without a GPU (no CPU fallback), and the host-side logic (index loader, synthetic data,
read sharding) behaves."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from snap_amd import abi, synth
from snap_amd.aligner import EXPORTED_SYMBOLS, LIB_PATH, load_library
from tests import util

ROOT = util.ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "snapgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(snapgpu_[a-z_0-9]+)\s*\(", hdr)))


def test_header_and_python_symbol_lists_agree():
    assert _declared_symbols() == sorted(EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB_PATH), "libsnapgpu.so not built: run __graft_entry__.build()"
    lib = load_library()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.snapgpu_abi_version() == 4


def test_struct_layouts_match_header():
    # sizes the C side static_asserts / documents
    assert abi.RESULT_DTYPE.itemsize == 88
    assert C.sizeof(abi.Counters) == 16 * 8
    p = abi.default_params()
    assert (p.max_hits, p.max_k, p.num_seeds, p.extra_search_depth, p.use_affine_gap) == (300, 27, 25, 1, 1)
    assert (p.match_reward, p.sub_penalty, p.gap_open_penalty, p.gap_extend_penalty) == (1, 4, 6, 1)
    assert (p.five_prime_end_bonus, p.three_prime_end_bonus, p.max_score_gap_to_prefer_non_alt) == (10, 7, 64)
    q = abi.Params()
    load_library().snapgpu_default_params(C.byref(q))
    for f, _ in abi.Params._fields_:
        assert getattr(p, f) == getattr(q, f), f


def test_no_cpu_fallback_without_gpu(golden_index):
    """Without a HIP device the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from snap_amd.aligner import BaseAligner, SnapGpuError
    with pytest.raises(SnapGpuError) as e:
        BaseAligner(golden_index, abi.default_params(max_k=8))
    assert "no HIP device" in str(e.value) or "NODEVICE" in str(e.value) or "-2" in str(e.value)


def test_create_rejects_bad_arguments():
    lib = load_library()
    h = C.c_void_p()
    rc = lib.snapgpu_create(None, None, 0, C.byref(h))
    assert rc == -1 and not h.value


def test_golden_index_is_consistent(golden_index):
    ix = golden_index
    assert ix.seed_len == 20 and ix.key_bytes == 4 and ix.n_hash_tables == 256 and not ix.large
    assert ix.genome_padded[:1024].tobytes() == b"n" * 1024 and ix.genome_padded[-1024:].tobytes() == b"n" * 1024
    assert ix.genome.size == ix.n_bases
    assert sum(c.is_alt for c in ix.contigs) == 1 and ix.first_alt_location == max(c.begin for c in ix.contigs)
    # every contig starts after chromosome padding of 'n'
    for c in ix.contigs:
        assert ix.genome[c.begin - 1] == ord("n") and ix.genome[c.begin] != ord("n")
    assert int(ix.table_offset[-1] + ix.table_size[-1] * ix.entry_bytes) + 16 == ix.hash_blob.size


def test_index_loader_roundtrip(tmp_path, golden_index):
    """Write the golden arrays back in the reference's on-disk layout and load them again."""
    from snap_amd.index import GENOME_PAD, GenomeIndex, HASH_MAGIC
    ix = golden_index
    d = tmp_path / "idx"
    d.mkdir()
    (d / "GenomeIndex").write_text("7 1 %d %d %d %d %d %d 1 4" % (ix.n_hash_tables, ix.overflow.size, ix.seed_len,
                                   ix.chromosome_padding, ix.key_bytes, 0))
    with open(d / "Genome", "wb") as f:
        f.write(b"%d %d 1\n" % (ix.n_bases, len(ix.contigs)))
        for c in ix.contigs:
            f.write(b"%d %x %d 0 0 %d 1 %s *\n" % (c.begin, 1 if c.is_alt else 0, c.original_number, len(c.name), c.name.encode()))
        f.write(ix.genome.tobytes())
    ix.overflow.tofile(d / "OverflowTable")
    with open(d / "GenomeIndexHash", "wb") as f:
        for t in range(ix.n_hash_tables):
            f.write(np.uint32(HASH_MAGIC).tobytes() + np.uint64(ix.table_size[t]).tobytes() + np.uint64(0).tobytes())
            f.write(np.array([ix.key_bytes, 4, 1], dtype=np.uint32).tobytes() + np.uint32(0xffffffff).tobytes())
            o = int(ix.table_offset[t])
            f.write(ix.hash_blob[o:o + int(ix.table_size[t]) * ix.entry_bytes].tobytes())
    ix2 = GenomeIndex.load_from_directory(str(d))
    assert (ix2.hash_blob == ix.hash_blob).all() and (ix2.overflow == ix.overflow).all()
    assert (ix2.genome_padded == ix.genome_padded).all() and (ix2.table_offset == ix.table_offset).all()
    assert [(c.begin, c.is_alt, c.name) for c in ix2.contigs] == [(c.begin, c.is_alt, c.name) for c in ix.contigs]


def test_synth_is_seeded_and_sane():
    g1 = synth.make_genome(5, 50_000, n_contigs=2, repeat_frac=0.3)
    g2 = synth.make_genome(5, 50_000, n_contigs=2, repeat_frac=0.3)
    assert all((a[1] == b[1]).all() for a, b in zip(g1, g2))
    r = synth.make_reads(6, g1, 500, 100)
    assert r["bases"].shape == (500, 100) and set(np.unique(r["bases"])) <= set(b"ACGTN")
    assert r["quals"].min() >= 53 and r["quals"].max() <= 73
    # an error-free forward read is a verbatim copy of the genome
    r0 = synth.make_reads(6, g1, 200, 100, sub=0, ins=0, dele=0, rc_frac=0)
    for i in range(200):
        gsrc = g1[int(r0["contig"][i])][1]
        assert (r0["bases"][i] == gsrc[int(r0["pos"][i]):int(r0["pos"][i]) + 100]).all()


def test_read_sharding_covers_every_read_once():
    from snap_amd.dist import shard_range
    for n in (0, 1, 7, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_product_code_never_reaches_for_the_oracle_or_the_emulator():
    """The oracle (oracle/) and the wavefront emulator (tests/emu/) are test infrastructure.  No product source may import, include, open or
    link them: Python under snap_amd/, the HIP / C++ sources under snap_amd/csrc/, the build hook's product part."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    pats = [re.compile(r"^\s*(from|import)\s+(oracle|tests)\b"), re.compile(r"#include\s+[\"<][^\">]*(oracle|tests)/"), re.compile(r"libsnapgpu_emu|liboracle|libsnapref|wave_emu|SNAPGPU_TEST_LIB")]
    for sub in ("snap_amd", os.path.join("snap_amd", "csrc"), os.path.join("snap_amd", "csrc", "host")):
        d = os.path.join(root, sub)
        for f in sorted(os.listdir(d)):
            p = os.path.join(d, f)
            if not os.path.isfile(p) or not f.endswith((".py", ".h", ".hip", ".cpp")):
                continue
            for ln, line in enumerate(open(p, errors="replace"), 1):
                if any(r.search(line) for r in pats):
                    bad.append("%s:%d: %s" % (os.path.relpath(p, root), ln, line.strip()))
    assert not bad, bad
    from snap_amd import aligner
    assert aligner.LIB_PATH == os.path.join(root, "snap_amd", "libsnapgpu.so") or os.environ.get("SNAPGPU_TEST_LIB")


def _multi_ctx_contract_worker(rank, q):
    """One of two processes (the shape the 8-GPU driver run has: one process per GPU) driving the argument contract of the multi-context
    exports without a GPU: every call must come back with the documented error, none may crash or block."""
    import ctypes as C
    from snap_amd.aligner import load_library
    lib = load_library()
    lib.snapgpu_device_count.restype = C.c_int
    lib.snapgpu_create_replica.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.snapgpu_broadcast_index.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    out = C.c_void_p(1234)
    res = dict(rank=rank, n_dev=lib.snapgpu_device_count(),
               replica_null=lib.snapgpu_create_replica(None, rank, 0, C.byref(out)), replica_out=out.value,
               bcast_null=lib.snapgpu_broadcast_index(None, 2), bcast_zero=lib.snapgpu_broadcast_index((C.c_void_p * 2)(), 0),
               bcast_null_ctx=lib.snapgpu_broadcast_index((C.c_void_p * 2)(None, None), 2),
               err=lib.snapgpu_last_error(None).decode())
    q.put(res)


def test_multi_context_exports_argument_contract_two_processes():
    """snapgpu_device_count / snapgpu_create_replica / snapgpu_broadcast_index (include/snapgpu.h) from two processes at once: the error
    convention (negative code + message, nothing created) holds where there is no GPU; the GPU side of the same exports is
    tests/test_gpu_multi_ctx.py."""
    import multiprocessing as mp
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by tests/test_gpu_multi_ctx.py")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_multi_ctx_contract_worker, args=(r, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted((q.get(timeout=120) for _ in ps), key=lambda d: d["rank"])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for d in got:
        assert d["n_dev"] == 0
        assert d["replica_null"] == -1 and d["replica_out"] in (None, 1234)          # SNAPGPU_E_INVALID, *out untouched or NULL
        assert d["bcast_null"] == -1 and d["bcast_zero"] == -1 and d["bcast_null_ctx"] == -1
        assert "snapgpu_broadcast_index" in d["err"]


def test_kernel_source_hash_covers_the_timed_kernels_sources_only(tmp_path, monkeypatch):
    """bench.py replays PMC counters only next to the build they were taken on: the hash must move with every source the align kernels are
    compiled from (their translation units' include closure + the host launcher) and must NOT move with the SAM-side / index-builder
    sources, which are not part of those kernels."""
    import shutil
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "repo"
    shutil.copytree(os.path.join(root, "snap_amd", "csrc"), fake / "snap_amd" / "csrc", ignore=shutil.ignore_patterns("host"))
    shutil.copytree(os.path.join(root, "include"), fake / "include")
    monkeypatch.delenv("SNAPGPU_BUILD_FLAGS", raising=False)
    monkeypatch.setattr(bench, "ROOT", str(fake))
    h0 = bench.kernel_source_hash()
    assert h0 == hashlib_of_repo(root, bench)

    def touched(name):
        p = fake / "snap_amd" / "csrc" / name
        if name.startswith("include/"):
            p = fake / name
        old = p.read_bytes()
        p.write_bytes(old + b"\n// x\n")
        h = bench.kernel_source_hash()
        p.write_bytes(old)
        return h
    for name in ("ag_win.h", "lv.h", "align_single.h", "dev_common.h", "paired.h", "paired_dev.h", "single_kernel.h", "probe.h", "bucket.h", "snapgpu.hip", "single_timed_k.hip", "paired_k.hip"):
        assert touched(name) != h0, name
    for name in ("cigar_ag.h", "sam_fields.h", "cigar_k.hip", "index_build.h", "index_build.hip"):
        assert touched(name) == h0, name
    assert touched("include/snapgpu.h") != h0                 # (the ABI structs and constants the kernels are compiled against)
    monkeypatch.setenv("SNAPGPU_BUILD_FLAGS", "-DSNAPGPU_WAVES_PER_SIMD(AGC)=5")
    assert bench.kernel_source_hash() != h0                   # another build of the same sources is another build


def hashlib_of_repo(root, bench):
    saved = bench.ROOT
    try:
        bench.ROOT = root
        return bench.kernel_source_hash()
    finally:
        bench.ROOT = saved
