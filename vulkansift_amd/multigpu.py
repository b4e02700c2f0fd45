"""Multi-GPU layer above the single-instance vksift API: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-GPU (vulkansift.h:32-34). Two things shard naturally:

  * detection — images are independent: rank r takes images [r*B/G, (r+1)*B/G); no collective
  * matching  — the QUERY set A is sharded by rows; every rank needs all of B, which arrives through
    ONE all-gather of the uint8 descriptor matrix (M x 128 bytes). Each query row still scans the
    whole of B in index order, so the 2-NN result (including the reference's tie rules) is
    bit-identical for every world size; results are concatenated in rank order.

torch is used for device memory, the all-gather and nothing else; the distances/top-2 are computed
by the HIP matcher behind the C-ABI (vksift_hip_match_2nn_desc). The compute step is injectable so
that the collective logic can be exercised on CPU (gloo) with the oracle standing in for the kernel.
"""
import numpy as np


def shard_range(n, world, rank):
    """Contiguous, balanced split of n items: the first n % world ranks get one extra item."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def split_batch(items, world, rank):
    s, e = shard_range(len(items), world, rank)
    return items[s:e]


def hip_match_fn(desc_a, a_index_base, desc_b):
    """2-NN of the dense uint8 descriptor tensors (torch, on the current GPU) through the C-ABI."""
    import torch

    from . import api

    na, nb = desc_a.shape[0], desc_b.shape[0]
    assert desc_a.is_cuda and desc_b.is_cuda and desc_a.dtype == torch.uint8 and desc_b.dtype == torch.uint8
    assert desc_a.is_contiguous() and desc_b.is_contiguous() and nb >= 2
    out = torch.empty((na, 5), dtype=torch.int32, device=desc_a.device)
    # norms + (large N_A) the partial top-2 lists of the B-chunked kernel, see include/vksift_hip.h
    scratch = torch.empty(2 * na + nb + 64 + (na * 5 * 8 if na > 32768 else 0), dtype=torch.int32, device=desc_a.device)
    stream = torch.cuda.current_stream(desc_a.device).cuda_stream
    err = api.lib().vksift_hip_match_2nn_desc(desc_a.data_ptr(), na, a_index_base, desc_b.data_ptr(), nb, scratch.data_ptr(), out.data_ptr(), stream)
    if err != 0:
        raise RuntimeError(f"vksift_hip_match_2nn_desc failed with HIP error {err}")
    return out


def all_gather_rows(local_rows, group=None):
    """All-gather of row-sharded 2-D tensors with (possibly) different row counts; returns the
    concatenation in rank order. One size exchange + one padded all_gather (a single collective on
    the data path: RCCL all-gather over xGMI)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_local = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes + [1])
    pad = torch.zeros((mx,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    gathered = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad, group=group)
    return torch.cat([g[:n] for g, n in zip(gathered, sizes)], dim=0), sizes


def sharded_match(desc_a_local, a_index_base, desc_b_local, match_fn=hip_match_fn, group=None):
    """Query-sharded 2-NN.

    desc_a_local : this rank's rows of A (n_a_local x 128, uint8), global row index of its first row = a_index_base
    desc_b_local : this rank's shard of B (rows in global order by rank)
    returns      : (n_a_local x 5) int32 match records {idx_a, idx_b1, idx_b2, dist1 bits, dist2 bits} for the
                   local A rows against the FULL B.
    """
    b_full, _ = all_gather_rows(desc_b_local, group=group)
    return match_fn(desc_a_local, a_index_base, b_full)


def records_to_struct(rec_int32):
    """(n x 5) int32 -> numpy structured array with the vksift_Match_2NN layout."""
    from .api import MATCH_DTYPE

    a = np.ascontiguousarray(rec_int32, dtype=np.int32)
    return a.view(np.uint8).reshape(-1, 20).copy().view(MATCH_DTYPE).reshape(-1)
