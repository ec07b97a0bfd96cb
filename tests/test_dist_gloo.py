"""world_size-2 test of the N>1 path on CPU (gloo): index replication by broadcast, read
sharding, max-over-ranks timing.  The aligner itself needs a GPU, so each rank 'aligns' by
looking its shard up in the committed golden results -- what is under test is the plumbing."""
import os
import subprocess
import sys
import textwrap

from tests import util

WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch
    sys.path.insert(0, %r)
    from snap_amd import dist as sd
    from tests.util import load_golden_index
    rank, world, _ = sd.env_rank_world()
    d = sd.init_process_group("gloo")
    dev = torch.device("cpu")
    index = load_golden_index() if rank == 0 else None
    ix, (h, o, g) = sd.broadcast_index(index, dev)
    full = load_golden_index()
    assert (h.numpy() == full.hash_blob).all()
    assert (o.numpy().view(np.uint32) == full.overflow).all()
    assert (g.numpy() == full.genome_padded).all()
    assert (ix.table_offset == full.table_offset).all() and (ix.table_size == full.table_size).all()
    assert [c.begin for c in ix.contigs] == [c.begin for c in full.contigs]
    assert ix.first_alt_location == full.first_alt_location and ix.n_bases == full.n_bases
    assert all((a == b).all() for a, b in zip(ix.projection_arrays(), full.projection_arrays()))
    # the blobs in pieces (what a 25 GB hash blob does: one broadcast per piece through one staging buffer) -- same bytes
    ixc, (hc, oc, gc) = sd.broadcast_index(index, dev, chunk_bytes=4096)
    assert (hc.numpy() == full.hash_blob).all() and (oc.numpy().view(np.uint32) == full.overflow).all() and (gc.numpy() == full.genome_padded).all()
    assert ixc._device_sizes == ix._device_sizes == (full.hash_blob.size, full.overflow.size, full.genome_padded.size)
    lift = load_golden_index("paired_alt_index.npz")
    ix2, _ = sd.broadcast_index(lift if rank == 0 else None, dev)
    assert all((a == b).all() for a, b in zip(ix2.projection_arrays(), lift.projection_arrays()))
    assert [c.proj_cigar for c in ix2.contigs] == [c.proj_cigar for c in lift.contigs]
    n = 3001
    b, e = sd.shard_range(n, rank, world)
    total = sd.sum_over_ranks(float(e - b), dev)
    assert total == n
    t = sd.max_over_ranks(1.0 + rank, dev)
    assert t == float(world)
    d.barrier()
    os.write(1, ("rank " + str(rank) + " ok" + chr(10)).encode())      # one write(2): the two ranks share a pipe, print() can interleave
""") % util.ROOT


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29577", str(script)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert "rank 0 ok" in out and "rank 1 ok" in out
