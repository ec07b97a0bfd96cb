"""Multi-GPU plumbing: one process per GPU, reads sharded, index replicated once.

The path shards embarrassingly (every read is aligned independently against a read-only index;
the reference does the same across threads, SNAPLib/SingleAligner.cpp:197, ParallelTask.h:128-138),
so there is no data-path collective.  The only exchange is the one-time replication of the index
blobs (hash tables, overflow table, genome): rank 0 reads the index directory and broadcasts the
three blobs over RCCL/xGMI straight into each rank's HBM.

torch.distributed is used as plumbing only (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests).
"""
from __future__ import annotations

import os

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced [begin, end) slice of n_items for `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: str):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend)
    return dist


INDEX_META_FIELDS = ("seed_len", "key_bytes", "n_hash_tables", "large", "location_size", "chromosome_padding", "n_bases")


def broadcast_index(index, device, src: int = 0):
    """Replicate a GenomeIndex from rank `src` to every rank.

    `index` is the loaded GenomeIndex on rank src and None elsewhere.  Returns
    (index_with_small_host_arrays, (hash_t, overflow_t, genome_padded_t)): the three big blobs as
    uint8/int32 torch tensors on `device` (each rank's HBM for nccl), filled by dist.broadcast.
    Small metadata (table offsets/sizes, contig table) travels as one int64 tensor.
    """
    import torch
    import torch.distributed as dist
    from .index import Contig, GenomeIndex
    rank = dist.get_rank()
    is_src = rank == src
    if is_src:
        hdr = np.array([getattr(index, f) if f != "large" else int(index.large) for f in INDEX_META_FIELDS]
                       + [index.hash_blob.size, index.overflow.size, index.genome_padded.size, len(index.contigs)],
                       dtype=np.int64)
    else:
        hdr = np.zeros(len(INDEX_META_FIELDS) + 4, dtype=np.int64)
    t = torch.from_numpy(hdr).to(device)
    dist.broadcast(t, src)
    hdr = t.cpu().numpy()
    meta = dict(zip(INDEX_META_FIELDS, hdr[:len(INDEX_META_FIELDS)].tolist()))
    hash_n, ovf_n, gen_n, n_contigs = [int(x) for x in hdr[len(INDEX_META_FIELDS):]]
    n_tables = int(meta["n_hash_tables"])
    if is_src:
        small = np.concatenate([index.table_offset.astype(np.int64), index.table_size.astype(np.int64),
                                index.contig_begin.astype(np.int64),
                                np.array([int(c.is_alt) for c in index.contigs], dtype=np.int64)])
    else:
        small = np.zeros(2 * n_tables + 2 * n_contigs, dtype=np.int64)
    ts = torch.from_numpy(small).to(device)
    dist.broadcast(ts, src)
    small = ts.cpu().numpy()
    # ALT-to-primary projections of the contigs (paired-end ALT liftover): proj_begin, proj_rc, CIGAR op table
    if is_src:
        pb, prc, cst, cops = index.projection_arrays()
        proj = np.concatenate([[cops.size], pb.astype(np.int64), prc.astype(np.int64), cst.astype(np.int64), cops.astype(np.int64)])
        plen = np.array([proj.size], dtype=np.int64)
    else:
        plen = np.zeros(1, dtype=np.int64)
    tl = torch.from_numpy(plen).to(device)
    dist.broadcast(tl, src)
    if not is_src:
        proj = np.zeros(int(tl.cpu().numpy()[0]), dtype=np.int64)
    tp = torch.from_numpy(proj).to(device)
    dist.broadcast(tp, src)
    proj = tp.cpu().numpy()

    blobs = []
    for name, n, dt in (("hash_blob", hash_n, torch.uint8), ("overflow", ovf_n, torch.int32), ("genome_padded", gen_n, torch.uint8)):
        if is_src:
            arr = getattr(index, name)
            src_t = torch.from_numpy(arr.view(np.int32) if dt == torch.int32 else arr)
            buf = src_t.to(device)                       # host -> this rank's HBM
        else:
            buf = torch.empty(n, dtype=dt, device=device)
        dist.broadcast(buf, src)                         # RCCL broadcast over xGMI, HBM to HBM
        blobs.append(buf)

    if is_src:
        out_index = index
    else:
        contigs = [Contig(int(b), bool(a), i, "contig%d" % i) for i, (b, a) in
                   enumerate(zip(small[2 * n_tables:2 * n_tables + n_contigs], small[2 * n_tables + n_contigs:]))]
        n_ops = int(proj[0])
        pb = proj[1:1 + n_contigs]; prc = proj[1 + n_contigs:1 + 2 * n_contigs]
        cst = proj[1 + 2 * n_contigs:2 + 3 * n_contigs]; cops = proj[2 + 3 * n_contigs:2 + 3 * n_contigs + n_ops]
        for i, c in enumerate(contigs):
            c.proj_begin, c.proj_rc = int(pb[i]), bool(prc[i])
            ops = cops[int(cst[i]):int(cst[i + 1])]
            c.proj_cigar = "".join("%d%s" % (int(o) >> 8, chr(int(o) & 0xff)) for o in ops) or "*"
        out_index = GenomeIndex(seed_len=int(meta["seed_len"]), key_bytes=int(meta["key_bytes"]), n_hash_tables=n_tables,
                                large=bool(meta["large"]), location_size=int(meta["location_size"]),
                                chromosome_padding=int(meta["chromosome_padding"]),
                                overflow=np.zeros(ovf_n, dtype=np.uint32)[:0], hash_blob=np.zeros(0, dtype=np.uint8),
                                table_offset=small[:n_tables].astype(np.uint64), table_size=small[n_tables:2 * n_tables].astype(np.uint64),
                                genome_padded=np.zeros(0, dtype=np.uint8), n_bases=int(meta["n_bases"]), contigs=contigs)
        # sizes the C ABI needs even though the blobs live on the device
        out_index._device_sizes = (hash_n, ovf_n, gen_n)
    if is_src:
        out_index._device_sizes = (hash_n, ovf_n, gen_n)
    return out_index, tuple(blobs)


def max_over_ranks(value: float, device) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
