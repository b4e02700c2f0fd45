"""The sharded matcher's C entry (vksift_ext_matchSharded: RCCL all-gather inside the library + MFMA matcher) on the GPU box.
One GPU is available to the tests, so the communicator has world size 1 — the collective, the stream fork/join and the
record layout are the real ones; the N > 1 data flow is covered on CPU (tests/test_multi_gpu_gloo.py) and by the CRC that
bench.py --gpus N prints for N = 1, 2, 4, 8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("na,nb,base", [(1000, 777, 0), (9000, 2500, 123), (40000, 3000, 7)])
def test_match_sharded_world1_equals_oracle(vk, oracle, na, nb, base):
    import torch
    from vulkansift_amd import multigpu

    a = vk.gen_synthetic_descriptors(71, na)
    b = vk.gen_synthetic_descriptors(72, nb)
    b[1] = b[0]
    b[nb // 2] = b[5]
    a[3] = b[0]
    grp = multigpu.ShardGroup(0, 1, 0)
    try:
        rec, ms = grp.match(torch.from_numpy(a).cuda(), base, torch.from_numpy(b).cuda(), nb)
        rec2, _ = grp.match(torch.from_numpy(a[:100]).cuda(), 0, torch.from_numpy(b).cuda(), nb)     # scratch reuse, smaller call
    finally:
        grp.close()
    assert ms > 0
    got = multigpu.records_to_struct(rec.cpu().numpy())
    ref = oracle.match_2nn(a, b)
    assert np.array_equal(got["idx_a"], ref["idx_a"] + base)
    for name in ("idx_b1", "idx_b2"):
        assert np.array_equal(got[name], ref[name]), name
    for name in ("dist_a_b1", "dist_a_b2"):
        assert np.array_equal(got[name].view(np.uint32), ref[name].view(np.uint32)), name
    assert got["idx_b1"][3] == 1 and got["idx_b2"][3] == 0      # quirk Q7 survives the gather
    got2 = multigpu.records_to_struct(rec2.cpu().numpy())
    assert np.array_equal(got2["idx_b1"], ref["idx_b1"][:100])


def test_match_sharded_equals_instance_matcher(vk):
    """the same descriptors through vksift_matchFeatures of an instance and through the sharded entry: identical records"""
    import torch
    from vulkansift_amd import multigpu

    img_a = vk.gen_synthetic_image(801, 480, 360)
    img_b = vk.gen_synthetic_image(802, 480, 360)
    with vk.Instance(vk.default_config()) as inst:
        inst.detectFeatures(img_a, 0)
        inst.detectFeatures(img_b, 1)
        fa, fb = inst.downloadFeatures(0), inst.downloadFeatures(1)
        inst.matchFeatures(0, 1)
        m = inst.downloadMatches()
        da = torch.empty((len(fa), 128), dtype=torch.uint8, device="cuda")
        db = torch.empty((len(fb), 128), dtype=torch.uint8, device="cuda")
        assert inst.exportDescriptorsDevice(0, da.data_ptr()) == len(fa)
        assert inst.exportDescriptorsDevice(1, db.data_ptr()) == len(fb)
    grp = multigpu.ShardGroup(0, 1, 0)
    try:
        rec, _ = grp.match(da, 0, db, len(fb))
    finally:
        grp.close()
    got = multigpu.records_to_struct(rec.cpu().numpy())
    assert got.tobytes() == m.tobytes()


def test_invalid_arguments_are_rejected(vk):
    import ctypes as C
    h = C.c_void_p(None)
    ident = (C.c_uint8 * 128)()
    assert vk.lib().vksift_ext_shardGroupCreate(C.byref(h), 0, 2, 2, ident) == 1       # rank >= world: VKSIFT_INVALID_INPUT_ERROR
    assert vk.lib().vksift_ext_shardGroupCreate(C.byref(h), 99, 1, 0, ident) == 2      # no such device: VKSIFT_VULKAN_ERROR
    assert vk.lib().vksift_ext_matchSharded(None, None, 0, 0, None, 0, 0, None) == 1


def test_local_failure_still_enters_the_collective(vk, oracle):
    """a rank that returns before ncclAllGather strands its peers. Rank-local invalid input (a NULL query pointer) must
    be reported AFTER the all-gather was entered, and must leave the group usable; reservations make later calls allocation-free"""
    import ctypes as C
    import torch
    from vulkansift_amd import multigpu

    na, nb = 3000, 2000
    a = vk.gen_synthetic_descriptors(81, na)
    b = vk.gen_synthetic_descriptors(82, nb)
    d_a, d_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.zeros((na, 5), dtype=torch.int32, device="cuda")
    grp = multigpu.ShardGroup(0, 1, 0)
    try:
        assert grp.reserve(na, nb) == 0
        L = vk.lib()
        # arguments every rank shares: rejected before anything is queued
        assert L.vksift_ext_matchSharded(grp._h, d_a.data_ptr(), na, 0, d_b.data_ptr(), 0, nb, out.data_ptr()) == 1
        assert L.vksift_ext_matchSharded(grp._h, d_a.data_ptr(), na, 0, d_b.data_ptr(), nb - 1, nb, out.data_ptr()) == 1
        # rank-local: NULL rows with na > 0 -> the collective runs (ev_gathered is recorded, synchronize returns), error code after
        assert L.vksift_ext_matchSharded(grp._h, None, na, 0, d_b.data_ptr(), nb, nb, out.data_ptr()) == 1
        ms = C.c_float(0)
        assert L.vksift_ext_shardGroupSynchronize(grp._h, C.byref(ms)) == 0
        assert int(out.abs().sum().item()) == 0          # no records were produced
        rec, _ = grp.match(d_a, 0, d_b, nb)              # the group is still usable
    finally:
        grp.close()
    got = multigpu.records_to_struct(rec.cpu().numpy())
    assert got.tobytes() == oracle.match_2nn(a, b).tobytes()
