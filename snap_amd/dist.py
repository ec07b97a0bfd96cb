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


class _DevicePtrArray:
    """A device allocation somebody else owns (the index blobs of a context) as something torch.as_tensor can wrap without copying."""

    def __init__(self, ptr: int, nbytes: int, typestr: str, itemsize: int):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(nbytes) // itemsize,), "typestr": typestr, "version": 2}


def index_meta_from_view(view):
    """The small host-side part of a GenomeIndex (header fields, table offsets / sizes, contig table) from a snapgpu_index_view whose
    blobs live on the device -- what snap_amd.index.BuiltIndex.view() returns for an index the GPU has just built.  The blob arrays
    of the result are empty; `_device_sizes` carries their sizes."""
    import ctypes as C
    from .index import Contig, GenomeIndex
    nt, nc = int(view.n_hash_tables), int(view.n_contigs)
    toff = np.ctypeslib.as_array(C.cast(view.table_offset, C.POINTER(C.c_uint64)), shape=(nt,)).copy()
    tsz = np.ctypeslib.as_array(C.cast(view.table_size, C.POINTER(C.c_uint64)), shape=(nt,)).copy()
    cb = np.ctypeslib.as_array(C.cast(view.contig_begin, C.POINTER(C.c_uint64)), shape=(nc,)).copy() if nc else np.zeros(0, np.uint64)
    first_alt = int(view.first_alt_location)
    contigs = [Contig(int(b), int(b) >= first_alt, i, "contig%d" % i) for i, b in enumerate(cb)]
    if view.contig_proj_begin and view.contig_cigar_start and nc:          # ALT-to-primary projections (an index built with an -altLiftoverFile)
        pb = np.ctypeslib.as_array(C.cast(view.contig_proj_begin, C.POINTER(C.c_uint64)), shape=(nc,))
        prc = np.ctypeslib.as_array(C.cast(view.contig_proj_rc, C.POINTER(C.c_uint8)), shape=(nc,)) if view.contig_proj_rc else np.zeros(nc, np.uint8)
        cst = np.ctypeslib.as_array(C.cast(view.contig_cigar_start, C.POINTER(C.c_uint32)), shape=(nc + 1,))
        cops = np.ctypeslib.as_array(C.cast(view.cigar_ops, C.POINTER(C.c_uint32)), shape=(max(1, int(cst[nc])),)) if view.cigar_ops else np.zeros(1, np.uint32)
        for i, c in enumerate(contigs):
            c.proj_begin, c.proj_rc = int(pb[i]), bool(prc[i])
            c.proj_cigar = "".join("%d%s" % (int(o) >> 8, chr(int(o) & 0xff)) for o in cops[int(cst[i]):int(cst[i + 1])]) or "*"
    ix = GenomeIndex(seed_len=int(view.seed_len), key_bytes=int(view.key_bytes), n_hash_tables=nt, large=bool(view.large_hash_table),
                     location_size=int(view.location_size), chromosome_padding=int(view.chromosome_padding),
                     overflow=np.zeros(0, dtype=np.uint32), hash_blob=np.zeros(0, dtype=np.uint8), table_offset=toff, table_size=tsz,
                     genome_padded=np.zeros(0, dtype=np.uint8), n_bases=int(view.n_bases), contigs=contigs)
    ix._device_sizes = (int(view.hash_blob_bytes), int(view.overflow_table_size), int(view.n_bases) + 2 * int(view.genome_pad))
    return ix


def broadcast_index(index, device, src: int = 0, src_device_ptrs=None, chunk_bytes: int = 1 << 30):
    """Replicate a GenomeIndex from rank `src` to every rank.

    `index` is the GenomeIndex on rank src and None elsewhere.  Returns
    (index_with_small_host_arrays, (hash_t, overflow_t, genome_padded_t)): the three big blobs as
    uint8/int32 torch tensors on `device` (each rank's HBM for nccl), filled by dist.broadcast.
    Small metadata (table offsets/sizes, contig table) travels as one int64 tensor.

    Where the blobs come from on rank src:
      src_device_ptrs = (hash, overflow, genome_with_pad) device pointers: the index is ALREADY in src's HBM (it built it there, or
        a context holds it): the broadcast reads it in place -- no host copy, no file -- and the returned tensors on src are views
        of that memory (the owner must outlive them);
      otherwise `index` holds them in host memory: they go up in pieces of `chunk_bytes` through one pinned staging buffer, each piece
        broadcast as soon as it is up, so that rank src never holds more than the index arrays plus one piece (the one-tensor
        `.to(device)` of a 25 GB pageable array took a second host copy of it: 91.6 GB peak at GRCh38 scale, profiles/r04z).
    Either way a blob travels in pieces of at most chunk_bytes: one RCCL broadcast per piece.
    """
    import torch
    import torch.distributed as dist
    from .index import Contig, GenomeIndex
    rank = dist.get_rank()
    is_src = rank == src
    if is_src:
        sizes = getattr(index, "_device_sizes", None) if src_device_ptrs is not None else None
        if sizes is None:
            sizes = (index.hash_blob.size, index.overflow.size, index.genome_padded.size)
        hdr = np.array([getattr(index, f) if f != "large" else int(index.large) for f in INDEX_META_FIELDS]
                       + [int(sizes[0]), int(sizes[1]), int(sizes[2]), len(index.contigs)],
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
    on_gpu = torch.device(device).type == "cuda"
    staging = None
    for k, (name, n, dt) in enumerate((("hash_blob", hash_n, torch.uint8), ("overflow", ovf_n, torch.int32), ("genome_padded", gen_n, torch.uint8))):
        item = 4 if dt == torch.int32 else 1
        host = None
        if is_src and src_device_ptrs is not None:
            try:
                buf = torch.as_tensor(_DevicePtrArray(src_device_ptrs[k], n * item, "<i4" if item == 4 else "|u1", item), device=device)
                if buf.data_ptr() != int(src_device_ptrs[k]) or buf.numel() != n:
                    raise RuntimeError("torch.as_tensor copied the blob")
            except Exception:          # noqa: BLE001 -- a torch build without the array interface: one device-to-device copy into a tensor of its own
                import ctypes as C
                buf = torch.empty(n, dtype=dt, device=device)
                hip = C.CDLL("libamdhip64.so")
                hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
                torch.cuda.synchronize()
                if hip.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(int(src_device_ptrs[k])), C.c_size_t(n * item), 3) != 0:      # hipMemcpyDeviceToDevice
                    raise RuntimeError("hipMemcpy of the index blob failed")
        else:
            buf = torch.empty(n, dtype=dt, device=device)
            if is_src:
                arr = getattr(index, name)
                host = torch.from_numpy(arr.view(np.int32) if dt == torch.int32 else arr)
        step = max(1, int(chunk_bytes) // item)
        for i in range(0, n, step):
            j = min(n, i + step)
            if host is not None:                         # host -> this rank's HBM, one piece through the pinned staging buffer
                if on_gpu:
                    if staging is None:
                        staging = torch.empty(int(chunk_bytes), dtype=torch.uint8, pin_memory=True)
                    st = staging[:(j - i) * item].view(dt)
                    st.copy_(host[i:j])
                    buf[i:j].copy_(st, non_blocking=False)
                else:
                    buf[i:j].copy_(host[i:j])
            dist.broadcast(buf[i:j], src)                # RCCL broadcast over xGMI, HBM to HBM
        blobs.append(buf)
    del staging

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
