"""Several contexts over one index (include/snapgpu.h: snapgpu_create_replica / snapgpu_broadcast_index) on the GPU box:
a feeder context that shares the first one's index blobs must answer exactly like it, concurrently; the broadcast of a one-context list is
a no-op; two contexts on one device are refused by the broadcast (one rank per GPU).  The N > 1 broadcast itself needs N GPUs: the driver's
8-GPU run exercises the torch.distributed form (snap_amd/dist.py), the C++ form is what snapgpu-sam -gpus N calls."""
import ctypes as C
import threading

import numpy as np
import pytest

from snap_amd import abi
from tests import util


@pytest.mark.gpu
def test_replica_sharing_the_index_answers_like_the_original(golden_index, golden_reads):
    from snap_amd.aligner import BaseAligner
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    lib = a.lib
    lib.snapgpu_device_count.restype = C.c_int
    lib.snapgpu_create_replica.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.snapgpu_broadcast_index.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    assert lib.snapgpu_device_count() >= 1
    h2 = C.c_void_p()
    assert lib.snapgpu_create_replica(a.handle, 0, 1, C.byref(h2)) == 0 and h2.value
    b = BaseAligner.__new__(BaseAligner)
    b.__dict__.update(a.__dict__)
    b.handle = h2
    try:
        z = golden_reads
        reads, quals = z["b100"], z["q100"]
        offs = np.arange(reads.shape[0] + 1, dtype=np.uint64) * 100
        out = {}

        def run(al, key):
            out[key] = al.AlignRead(reads, quals, offs)[0]
        ts = [threading.Thread(target=run, args=(a, "a")), threading.Thread(target=run, args=(b, "b"))]       # two feeder threads, one GPU
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not util.compare_results(out["a"], out["b"])
        exp, _ = util.with_fresh_overrides(z["default_d8_100_primary"], "default_d8_100_primary")
        assert not util.compare_results(exp, out["b"])
        # broadcast: a single context is a no-op; a replica on the SAME device is not a second rank; a sharing context has no blobs of its own
        assert lib.snapgpu_broadcast_index((C.c_void_p * 1)(a.handle), 1) == 0
        h3 = C.c_void_p()
        assert lib.snapgpu_create_replica(a.handle, 0, 0, C.byref(h3)) == 0
        try:
            assert lib.snapgpu_broadcast_index((C.c_void_p * 2)(a.handle, h3), 2) == -1
            assert b"one device" in lib.snapgpu_last_error(a.handle)
            assert lib.snapgpu_broadcast_index((C.c_void_p * 2)(a.handle, h2), 2) == -1
        finally:
            lib.snapgpu_destroy(h3)
    finally:
        lib.snapgpu_destroy(h2)
        b.handle = None                     # (b borrowed a's Python state; its context is gone)
        a.close()


@pytest.mark.gpu
def test_paired_feeders_two_batches_in_flight():
    """bench.py --feeders 2 / snapgpu-sam's two feeders per GPU on the paired-end path: a replica made by the Python mirror
    (BaseAligner.replica -> snapgpu_create_replica + snapgpu_enable_paired) aligns the same pairs, at the same time, to the same bytes."""
    from snap_amd.aligner import ChimericPairedEndAligner
    from tests.pairs_util import compare_paired
    import os
    zp = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
    a = ChimericPairedEndAligner(util.load_golden_index("paired_index.npz"), abi.default_params(max_k=8, max_read_len=160),
                                 abi.default_paired_params())
    b = a.replica()
    try:
        n_pairs = 200
        o = zp["o150"][:2 * n_pairs + 1]
        bases, quals = zp["b150"][:int(o[-1])], zp["q150"][:int(o[-1])]
        out = {}

        def run(al, key):
            out[key] = al.align(bases, quals, o)[0]
        ts = [threading.Thread(target=run, args=(a, "a")), threading.Thread(target=run, args=(b, "b"))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert out["a"].tobytes() == out["b"].tobytes()
        exp, _ = util.with_fresh_overrides(zp["default_d8_150_s0_primary"], "pe_default_d8_150_s0_primary")
        assert not compare_paired(exp[:n_pairs], out["b"], verbose=0).any()
    finally:
        b.close()
        a.close()


@pytest.mark.gpu
def test_paired_grid_share_three_feeders_same_bytes(monkeypatch):
    """snapgpu.hip: paired_grid_share -- a paired-end launch asks for its share of the chip (calls in flight on the device, at most three
    shares; SNAPGPU_PAIRED_GRID_SHARE fixes the divisor).  The grid's size is scheduling only: three feeders aligning the same 1 500 pairs
    at the same time, and one context under divisors 1, 3 and 7, produce the bytes of the fixture's reference results."""
    from snap_amd.aligner import ChimericPairedEndAligner
    from tests.pairs_util import compare_paired
    import os
    zp = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
    a = ChimericPairedEndAligner(util.load_golden_index("paired_index.npz"), abi.default_params(max_k=8, max_read_len=160),
                                 abi.default_paired_params())
    feeders = [a, a.replica(), a.replica()]
    try:
        n_avail = (len(zp["o150"]) - 1) // 2
        n_pairs = min(3000, n_avail)
        o = zp["o150"][:2 * n_pairs + 1]
        bases, quals = zp["b150"][:int(o[-1])], zp["q150"][:int(o[-1])]
        exp, _ = util.with_fresh_overrides(zp["default_d8_150_s0_primary"], "pe_default_d8_150_s0_primary")
        out = {}

        def run(al, key):
            for rep in range(2):            # (the second call starts while the others' first calls are in flight)
                out[key, rep] = al.align(bases, quals, o)[0]
        ts = [threading.Thread(target=run, args=(f, i)) for i, f in enumerate(feeders)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for k, v in out.items():
            assert v.tobytes() == out[0, 0].tobytes(), k
        assert not compare_paired(exp[:n_pairs], out[0, 0], verbose=0).any()
        for share in ("1", "3", "7"):
            monkeypatch.setenv("SNAPGPU_PAIRED_GRID_SHARE", share)
            assert a.align(bases, quals, o)[0].tobytes() == out[0, 0].tobytes(), share
    finally:
        for f in feeders[1:]:
            f.close()
        a.close()
