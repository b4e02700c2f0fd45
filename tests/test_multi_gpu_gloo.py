"""world_size-2 run of the multi-GPU data flow on CPU (gloo): batch splitting needs no collective, the query-sharded
matcher needs exactly one all-gather of equal (padded) blocks of B and gives the single-process result bit for bit.
There is no GPU here, so the compute step is the oracle; the product path — the C entry vksift_ext_matchSharded with its
own RCCL all-gather — is exercised on the GPU box by tests/test_gpu_sharded.py (world size 1) and by bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_match_fn(desc_a, a_index_base, desc_b):
    from oracle import oracle as O

    m = O.match_2nn(desc_a.numpy(), desc_b.numpy())
    m["idx_a"] += a_index_base
    return torch.from_numpy(m.view(np.uint8).reshape(-1, 20).copy().view(np.int32).reshape(-1, 5))


def _worker(rank, world, port, na, nb, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vulkansift_amd import api, multigpu

    a = api.gen_synthetic_descriptors(31, na)
    b = api.gen_synthetic_descriptors(32, nb)
    b[1] = b[0]
    a0, a1 = multigpu.shard_range(na, world, rank)
    b0, b1 = multigpu.shard_bounds(nb, world, rank)
    rec = multigpu.sharded_match_reference(torch.from_numpy(a[a0:a1]), a0, torch.from_numpy(b[b0:b1]), nb, match_fn=_oracle_match_fn)
    # detection-side splitting: no collective, just a partition
    imgs = list(range(13))
    mine = multigpu.split_batch(imgs, world, rank)
    q.put((rank, a0, rec.numpy(), mine))
    dist.barrier()
    dist.destroy_process_group()


class _StubGroup:
    """Stands in for multigpu.ShardGroup where there is no GPU: the same match() contract — this rank's query rows, its PADDED block of
    B, nb_total — with the exchange over gloo and the oracle as the compute step. Everything around it (block size from the C layout,
    padding, barriers, repeats, record gather) is multigpu.sharded_match_timed's own code."""

    def __init__(self, world, rank):
        self.world, self.rank, self.calls = world, rank, 0

    def match(self, d_a, a_index_base, d_b_shard, nb_total):
        from vulkansift_amd import multigpu

        blk = multigpu.shard_layout(nb_total, self.world, self.rank)[0]
        assert d_b_shard.shape[0] == blk and d_b_shard.is_contiguous()      # what vksift_ext_matchSharded validates as nb_shard
        full = torch.empty((self.world * blk, 128), dtype=torch.uint8)
        dist.all_gather_into_tensor(full, d_b_shard)
        self.calls += 1
        return _oracle_match_fn(d_a, a_index_base, full[:nb_total]), 1.0 / self.calls

    def close(self):
        pass


def _timed_worker(rank, world, port, na, nb, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vulkansift_amd import api, multigpu

    a = api.gen_synthetic_descriptors(41, na)
    b = api.gen_synthetic_descriptors(42, nb)
    b[1] = b[0]
    # bench.py's sharded_match(): A and B both split with the C block layout
    lo, hi = multigpu.shard_bounds(na, world, rank)
    blo, bhi = multigpu.shard_bounds(nb, world, rank)
    groups = []

    def factory(w, r):
        groups.append(_StubGroup(w, r))
        return groups[-1]

    ms, rec = multigpu.sharded_match_timed(torch.from_numpy(a[lo:hi]), lo, torch.from_numpy(b[blo:bhi]), nb, world, rank, repeats=3, group_factory=factory)
    rec_all = multigpu.gather_records(rec, na, world, rank)
    q.put((rank, lo, hi, ms, groups[0].calls, rec_all.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("na,nb", [(101, 77), (9, 3), (64, 64)])
def test_sharded_match_timed_flow_world2(oracle, vk, na, nb):
    """bench.py's sharded leg at world size 2 with only the C entry's device work stubbed: C block layout (vksift_ext_shardGroupLayout)
    for A and B, padding of the short last block, a_index_base = first row, best-of-repeats, and the all-gather of the records."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timed_worker, args=(r, world, port, na, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from vulkansift_amd import multigpu

    a = vk.gen_synthetic_descriptors(41, na)
    b = vk.gen_synthetic_descriptors(42, nb)
    b[1] = b[0]
    ref = oracle.match_2nn(a, b)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == na          # the two A shards tile [0, na)
    for rank, lo, hi, ms, calls, rec_all in res:
        assert calls == 3 and abs(ms - 1.0 / 3) < 1e-9                             # three repeats, the best one reported
        got = multigpu.records_to_struct(rec_all)                                  # every rank holds ALL records after the gather
        assert got.tobytes() == ref.tobytes()


def test_shard_layout_is_the_c_arithmetic(vk):
    from vulkansift_amd import multigpu

    for n in (0, 1, 2, 3, 7, 64, 50000, 50001, 2**32 - 1):
        for world in (1, 2, 3, 8):
            blk = (n + world - 1) // world
            rows = 0
            for r in range(world):
                b, lo, cnt = multigpu.shard_layout(n, world, r)
                assert b == blk and lo == min(n, r * blk) and cnt == min(n, lo + blk) - lo
                rows += cnt
            assert rows == n


@pytest.mark.parametrize("na,nb", [(101, 77), (8, 3), (5, 2)])
def test_sharded_match_equals_single_process(oracle, vk, na, nb):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, na, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from vulkansift_amd import multigpu

    got = np.concatenate([multigpu.records_to_struct(r[2]) for r in res])
    a = vk.gen_synthetic_descriptors(31, na)
    b = vk.gen_synthetic_descriptors(32, nb)
    b[1] = b[0]
    ref = oracle.match_2nn(a, b)
    for name in ref.dtype.names:
        assert np.array_equal(got[name], ref[name]), name
    assert sorted(sum([r[3] for r in res], [])) == list(range(13))


def test_shard_range_partitions():
    from vulkansift_amd import multigpu

    for n in (0, 1, 7, 64, 50000):
        for world in (1, 2, 3, 8):
            parts = [multigpu.shard_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in parts]
            assert max(sizes) - min(sizes) <= 1


def test_shard_bounds_are_equal_blocks():
    from vulkansift_amd import multigpu

    for n in (2, 3, 7, 64, 50000, 50001):
        for world in (1, 2, 3, 8):
            blk = (n + world - 1) // world
            parts = [multigpu.shard_bounds(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and max(e for _, e in parts) == n
            assert all(e - s <= blk for s, e in parts)
            assert all(s == min(n, r * blk) for r, (s, _) in enumerate(parts))
