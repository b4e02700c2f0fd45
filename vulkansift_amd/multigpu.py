"""Multi-GPU layer above the single-instance vksift API: one process per GPU.

The reference is single-GPU (vulkansift.h:32-34). Two things shard naturally:

  * detection — images are independent: rank r takes images [r*B/G, (r+1)*B/G); no collective
  * matching  — the QUERY set A is sharded by rows; every rank needs all of B, which arrives through ONE all-gather of
    the uint8 descriptor matrix (M x 128 bytes). Each query row still scans the whole of B in index order, so the 2-NN
    result (including the reference's tie rules) is bit-identical for every world size.

The product path is the C entry vksift_ext_matchSharded (csrc/host/vksift_sharded.c): RCCL all-gather issued inside the
library, overlapped with the norm pre-pass of the local rows, then the MFMA matcher. `ShardGroup` below is its ctypes
handle; torch.distributed is used only to carry the 128-byte RCCL id from rank 0 to the other ranks and to collect
results. `sharded_match_reference` is the same data flow over any torch.distributed backend with an injectable compute
step — the form the CPU tests run (gloo, world size 2, the oracle standing in for the kernel).
"""
import ctypes as C

import numpy as np


def shard_range(n, world, rank):
    """Contiguous, balanced split of n items: the first n % world ranks get one extra item (query rows, images)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_layout(n, world, rank):
    """vksift_ext_shardGroupLayout: (block_rows, first_row, nb_rows) of rank `rank` in the equal-block layout of the reference set
    that vksift_ext_matchSharded all-gathers. The arithmetic lives in the C library (vksift_sharded.c); this is its binding."""
    from . import api

    blk, lo, cnt = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    api.lib().vksift_ext_shardGroupLayout(n, world, rank, C.byref(blk), C.byref(lo), C.byref(cnt))
    return int(blk.value), int(lo.value), int(cnt.value)


def shard_bounds(n, world, rank):
    """Equal blocks of ceil(n / world) rows — the layout of the reference set B that vksift_ext_matchSharded all-gathers
    (the last blocks may be short or empty; the caller pads them to the block size)."""
    _, lo, cnt = shard_layout(n, world, rank)
    return lo, lo + cnt


def split_batch(items, world, rank):
    s, e = shard_range(len(items), world, rank)
    return items[s:e]


def hip_match_fn(desc_a, a_index_base, desc_b):
    """2-NN of the dense uint8 descriptor tensors (torch, on the current GPU) through the kernel C-ABI (single GPU)."""
    import torch

    from . import api

    na, nb = desc_a.shape[0], desc_b.shape[0]
    assert desc_a.is_cuda and desc_b.is_cuda and desc_a.dtype == torch.uint8 and desc_b.dtype == torch.uint8
    assert desc_a.is_contiguous() and desc_b.is_contiguous() and nb >= 2
    out = torch.empty((na, 5), dtype=torch.int32, device=desc_a.device)
    # the library's own figure (norms, row list, per-row lists of the decomposed kernels): never a formula restated here
    n_words = int(api.lib().vksift_hip_match_scratch_u32(na, nb))
    scratch = torch.empty(n_words, dtype=torch.int32, device=desc_a.device)
    stream = torch.cuda.current_stream(desc_a.device).cuda_stream
    err = api.lib().vksift_hip_match_2nn_desc(desc_a.data_ptr(), na, a_index_base, desc_b.data_ptr(), nb, scratch.data_ptr(), n_words, out.data_ptr(), stream)
    if err != 0:
        raise RuntimeError(f"vksift_hip_match_2nn_desc failed with HIP error {err}")
    return out


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p)


def host_staged_transport(dist, group=None):
    """A vksift_ext_AllGatherFn over ANY torch.distributed backend (gloo included): synchronise the stream, copy this rank's block
    to the host, all-gather on the host, copy the gathered set back. Slow by construction — it exists so that the world-size > 1
    code path of vksift_ext_matchSharded (block layout, in-place send slot, a_index_base, stream fork / join) can run where RCCL
    cannot: several ranks sharing one GPU. Returns the Python callable for ShardGroup(transport=...)."""
    import torch

    from . import api

    def all_gather(user, d_send, d_recv, nbytes, rank, world, stream):
        try:
            L = api.lib()
            if L.vksift_hip_stream_sync(stream) != 0:
                return 1
            mine = np.empty(nbytes, np.uint8)
            if L.vksift_hip_memcpy_d2h(mine.ctypes.data, d_send, nbytes, stream) != 0 or L.vksift_hip_stream_sync(stream) != 0:
                return 2
            full = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(full, torch.from_numpy(mine), group=group)
            host = full.numpy()
            if L.vksift_hip_memcpy_h2d(d_recv, host.ctypes.data, world * nbytes, stream) != 0 or L.vksift_hip_stream_sync(stream) != 0:
                return 3
            return 0
        except Exception:  # noqa: BLE001 - an exception must not unwind through the C frames of the library
            import traceback

            traceback.print_exc()
            return -1

    return all_gather


class ShardGroup:
    """vksift_ext_ShardGroup: the library's own RCCL communicator (`dist` — torch.distributed, initialised — only broadcasts the
    id), or, with `transport` (a Python callable with the vksift_ext_AllGatherFn signature), the application's own all-gather."""

    def __init__(self, device_index, world, rank, dist=None, transport=None):
        import torch

        from . import api

        self._api = api
        api.load()
        self.world, self.rank = world, rank
        self._h = C.c_void_p(None)
        if transport is not None:
            self._cb = ALL_GATHER_FN(transport)      # keep the thunk alive as long as the group
            r = api.lib().vksift_ext_shardGroupCreateWithTransport(C.byref(self._h), device_index, world, rank, self._cb, None)
            if r != 0:
                raise RuntimeError(f"vksift_ext_shardGroupCreateWithTransport failed ({r})")
            return
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            if api.lib().vksift_ext_shardGetUniqueId(buf) != 0:
                raise RuntimeError("vksift_ext_shardGetUniqueId failed (RCCL missing?)")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if world > 1:
            ident = ident.to(torch.device("cuda", device_index)) if dist.get_backend() == "nccl" else ident
            dist.broadcast(ident, src=0)
            ident = ident.cpu()
        raw = (C.c_uint8 * 128)(*ident.tolist())
        r = api.lib().vksift_ext_shardGroupCreate(C.byref(self._h), device_index, world, rank, raw)
        if r != 0:
            raise RuntimeError(f"vksift_ext_shardGroupCreate failed ({r})")

    def info(self):
        """{"world", "rank", "rccl_ranks", "rccl_rank"}: the last two are ncclCommCount / ncclCommUserRank of the group's own communicator
        (0 for a transport group)"""
        v = [C.c_uint32(0) for _ in range(4)]
        self._api.lib().vksift_ext_shardGroupInfo(self._h, *[C.byref(x) for x in v])
        return dict(zip(("world", "rank", "rccl_ranks", "rccl_rank"), (int(x.value) for x in v)))

    def reserve(self, max_na, max_nb_total):
        """vksift_ext_shardGroupReserve: device scratch up front, so that match() within these sizes allocates nothing (a rank that
        runs out of memory inside a collective call cannot leave it without stranding its peers). Local; returns the result code —
        agree on it across the ranks (e.g. an all-reduce) before the first match()."""
        return self._api.lib().vksift_ext_shardGroupReserve(self._h, max_na, max_nb_total)

    def match(self, d_a, a_index_base, d_b_shard, nb_total):
        """d_a: this rank's query rows (n x 128 uint8, cuda); d_b_shard: its block of B padded to ceil(nb_total / world) rows.
        Returns (records (n x 5 int32, cuda), milliseconds of the whole pipeline on this rank)."""
        import torch

        assert d_a.is_cuda and d_a.dtype == torch.uint8 and d_a.is_contiguous() and d_b_shard.is_contiguous()
        out = torch.empty((d_a.shape[0], 5), dtype=torch.int32, device=d_a.device)
        torch.cuda.synchronize(d_a.device)      # the group's stream is the library's own: inputs must be complete
        r = self._api.lib().vksift_ext_matchSharded(self._h, d_a.data_ptr(), d_a.shape[0], a_index_base, d_b_shard.data_ptr(), d_b_shard.shape[0], nb_total,
                                                    out.data_ptr())
        if r != 0:
            raise RuntimeError(f"vksift_ext_matchSharded failed ({r})")
        ms = C.c_float(0)
        if self._api.lib().vksift_ext_shardGroupSynchronize(self._h, C.byref(ms)) != 0:
            raise RuntimeError("vksift_ext_shardGroupSynchronize failed")
        return out, float(ms.value)

    def close(self):
        if self._h:
            self._api.lib().vksift_ext_shardGroupDestroy(C.byref(self._h))
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pad_rows(rows, n):
    """zero-pad a (k x 128) tensor to n rows (padding rows are never addressed: nb_total bounds the scan)"""
    import torch

    if rows.shape[0] == n:
        return rows.contiguous()
    out = torch.zeros((n,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    out[: rows.shape[0]] = rows
    return out


def sharded_match_timed(d_a, a_index_base, d_b_block, nb_total, world, rank, repeats=3, group_factory=None):
    """bench.py helper: best-of-`repeats` time of vksift_ext_matchSharded on this rank + its records. group_factory(world, rank)
    (default: an RCCL ShardGroup on d_a's device) lets the CPU tests run this exact flow — block size from the C layout, padding,
    barriers, repeats — with a stand-in group."""
    import torch.distributed as dist

    blk = shard_layout(nb_total, world, rank)[0]
    grp = group_factory(world, rank) if group_factory else ShardGroup(d_a.device.index, world, rank, dist if world > 1 else None)
    try:
        d_b = pad_rows(d_b_block, blk)
        best, rec = None, None
        sharded_match_timed.last_info = grp.info() if hasattr(grp, "info") else None
        for _ in range(repeats):
            if world > 1:
                dist.barrier()
            rec, ms = grp.match(d_a, a_index_base, d_b, nb_total)
            best = ms if best is None else min(best, ms)
        return best, rec
    finally:
        grp.close()


def gather_records(rec_local, n_total, world, rank):
    """all ranks' (n_local x 5) int32 records -> rank order concatenation (on every rank); A was split with shard_bounds"""
    import torch
    import torch.distributed as dist

    if world == 1:
        return rec_local
    blk = shard_layout(n_total, world, rank)[0]
    pad = torch.zeros((blk, 5), dtype=rec_local.dtype, device=rec_local.device)
    pad[: rec_local.shape[0]] = rec_local
    full = torch.empty((world * blk, 5), dtype=rec_local.dtype, device=rec_local.device)
    dist.all_gather_into_tensor(full, pad)
    return full[:n_total]


def sharded_match_reference(desc_a_local, a_index_base, desc_b_block, nb_total, match_fn, group=None):
    """The data flow of vksift_ext_matchSharded over torch.distributed (any backend) with an injectable compute step:
    equal B blocks (padded once), ONE all_gather_into_tensor, no host round trip, then match_fn(local A rows, base, all of B)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    blk = (nb_total + world - 1) // world
    b_pad = pad_rows(desc_b_block, blk)
    b_full = torch.empty((world * blk,) + tuple(b_pad.shape[1:]), dtype=b_pad.dtype, device=b_pad.device)
    dist.all_gather_into_tensor(b_full, b_pad, group=group)
    return match_fn(desc_a_local, a_index_base, b_full[:nb_total])


def records_to_struct(rec_int32):
    """(n x 5) int32 -> numpy structured array with the vksift_Match_2NN layout."""
    from .api import MATCH_DTYPE

    a = np.ascontiguousarray(rec_int32, dtype=np.int32)
    return a.view(np.uint8).reshape(-1, 20).copy().view(MATCH_DTYPE).reshape(-1)
