"""vksift_ext_matchSharded at WORLD SIZE 2 on the one GPU of the test box. RCCL refuses two ranks on one device, so the exchange
goes through the library's transport hook (vksift_ext_shardGroupCreateWithTransport) with a host-staged all-gather over gloo;
everything else is the product's N > 1 path as the 8-GPU run executes it: two processes, the C block layout
(vksift_ext_shardGroupLayout), a short last block, rank 1's rows landing in slot 1 of the gathered set, a_index_base != 0, ties
(quirk Q7) across the block border, the stream fork / join around the exchange, and the failure discipline (a rank with a local
error still enters the exchange; its peer completes; the group stays usable). Only ncclAllGather itself is not executed here
(tests/test_gpu_sharded.py runs it at world size 1)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(5000, 3001), (1201, 778), (40000, 2500), (7, 2)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(api, na, nb):
    a = api.gen_synthetic_descriptors(91, na)
    b = api.gen_synthetic_descriptors(92, nb)
    blk = (nb + 1) // 2
    b[1] = b[0]                      # a tie inside block 0
    if nb > blk:
        b[blk] = b[blk - 1]          # a tie across the block border: index order must survive the gather
        a[na - 1] = b[blk]           # an exact hit whose two best are on the two sides of the border
    a[0] = b[0]
    return a, b


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes as C

    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vulkansift_amd import api, multigpu

    torch.cuda.set_device(0)
    grp = multigpu.ShardGroup(0, world, rank, transport=multigpu.host_staged_transport(dist))
    out = {}
    try:
        for na, nb in CASES:
            a, b = _inputs(api, na, nb)
            lo, hi = multigpu.shard_bounds(na, world, rank)
            blo, bhi = multigpu.shard_bounds(nb, world, rank)
            blk = multigpu.shard_layout(nb, world, rank)[0]
            d_a = torch.from_numpy(a[lo:hi]).cuda()
            d_b = multigpu.pad_rows(torch.from_numpy(b[blo:bhi]).cuda(), blk)
            rec, ms = grp.match(d_a, lo, d_b, nb)
            out[(na, nb)] = (lo, hi, rec.cpu().numpy(), ms)
        # failure discipline at world size 2: rank 1 passes no shard (rank-local invalid input). It must still enter the exchange —
        # otherwise rank 0 would block in the all-gather for ever — and report its error afterwards; rank 0 completes.
        na, nb = 2000, 1000
        a, b = _inputs(api, na, nb)
        lo, hi = multigpu.shard_bounds(na, world, rank)
        blo, bhi = multigpu.shard_bounds(nb, world, rank)
        d_a = torch.from_numpy(a[lo:hi]).cuda()
        d_b = multigpu.pad_rows(torch.from_numpy(b[blo:bhi]).cuda(), 500)
        rec = torch.zeros((hi - lo, 5), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        L = api.lib()
        r = L.vksift_ext_matchSharded(grp._h, d_a.data_ptr(), hi - lo, lo, d_b.data_ptr() if rank == 0 else None, 500, nb, rec.data_ptr())
        ms = C.c_float(0)
        rs = L.vksift_ext_shardGroupSynchronize(grp._h, C.byref(ms))
        out["failure"] = (r, rs)
        rec2, _ = grp.match(d_a, lo, d_b, nb)            # both ranks again: the group is still usable
        out["after_failure"] = (lo, hi, rec2.cpu().numpy())
    finally:
        grp.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_match_sharded_world2_on_one_gpu(vk, oracle):
    import torch.multiprocessing as mp

    from vulkansift_amd import multigpu

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for na, nb in CASES:
        a, b = _inputs(vk, na, nb)
        ref = oracle.match_2nn(a, b)
        parts = [res[r][(na, nb)] for r in range(world)]
        assert parts[0][0] == 0 and parts[0][1] == parts[1][0] and parts[1][1] == na
        got = np.concatenate([multigpu.records_to_struct(p[2]) for p in parts])
        assert got.tobytes() == ref.tobytes(), (na, nb)          # idx_a, both neighbours, both distances: bit for bit
        assert got["idx_b1"][0] == 1 and got["idx_b2"][0] == 0    # Q7 inside block 0
        if nb > 2:
            blk = (nb + 1) // 2
            # the pair across the border, strict '<': the earlier index (rank 0's last row) stays the best, rank 1's first row second
            assert (int(got["idx_b1"][na - 1]), int(got["idx_b2"][na - 1])) == (blk - 1, blk)
        assert all(p[3] > 0 for p in parts)
    # rank 1 reported VKSIFT_INVALID_INPUT_ERROR after the exchange, rank 0 succeeded, both synchronised
    assert res[0]["failure"] == (0, 0) and res[1]["failure"] == (1, 0)
    a, b = _inputs(vk, 2000, 1000)
    ref = oracle.match_2nn(a, b)
    got = np.concatenate([multigpu.records_to_struct(res[r]["after_failure"][2]) for r in range(world)])
    assert got.tobytes() == ref.tobytes()
