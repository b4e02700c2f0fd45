"""The arithmetic of the single-pair matcher's stream decomposition (vulkansift_amd/csrc/hip/match.hip: stream_span, the piece loop of
k_match_mfma, the piece count of k_match_merge), restated in Python and checked over many shapes: every (row block, B tile) pair
is visited exactly once, the pieces of a row block are numbered 0..n-1 without gaps in increasing tile order, and no row block
needs more partial lists than VKSIFT_HIP_MATCH_CHUNKS. (The kernels themselves are compared with the oracle by the -m gpu tests.)"""
import re
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _chunks():
    hdr = open(os.path.join(ROOT, "include", "vksift_hip.h")).read()
    return int(re.search(r"#define VKSIFT_HIP_MATCH_CHUNKS (\d+)", hdr).group(1))


CHUNKS = _chunks()


def stream_span(nblocks, tiles, G):
    even = (nblocks * tiles + G - 1) // G
    floor_ = (tiles + CHUNKS - 3) // (CHUNKS - 2)
    return max(even, floor_, 1)


def pieces(na, nb, G, block_rows=256, tile_rows=128):
    """what each workgroup of the grid does: (workgroup, row block, first tile, tile count, partial-list slot)"""
    nblocks, tiles = (na + block_rows - 1) // block_rows, (nb + tile_rows - 1) // tile_rows
    span = stream_span(nblocks, tiles, G)
    out = []
    for w in range(G):
        pos, end = w * span, min((w + 1) * span, nblocks * tiles)
        while pos < end:
            rb = pos // tiles
            t_first = pos - rb * tiles
            t_cnt = min(tiles - t_first, end - pos)
            out.append((w, rb, t_first, t_cnt, w - (rb * tiles) // span))
            pos += t_cnt
    return out, nblocks, tiles, span


def merge_count(rb, tiles, span):
    return ((rb + 1) * tiles - 1) // span - (rb * tiles) // span + 1


@pytest.mark.parametrize("na,nb", [(1, 2), (300, 33000), (1537, 2), (1537, 4097), (4000, 4000), (9000, 700), (33000, 5000), (50000, 50000),
                                   (70001, 130), (100000, 100000), (257, 128), (256, 129), (1000000, 300)])
@pytest.mark.parametrize("G", [512, 608, 64])
def test_every_tile_once_and_slots_dense(na, nb, G):
    ps, nblocks, tiles, span = pieces(na, nb, G)
    seen = np.zeros((nblocks, tiles), np.int32)
    slots = {}
    for w, rb, t0, cnt, slot in ps:
        seen[rb, t0:t0 + cnt] += 1
        slots.setdefault(rb, []).append((t0, slot))
    assert (seen == 1).all()
    for rb, lst in slots.items():
        lst.sort()
        assert [s for _, s in lst] == list(range(len(lst))), (rb, lst)          # slot order == tile order: ties keep arrival order
        assert len(lst) == merge_count(rb, tiles, span) <= CHUNKS, (rb, len(lst))
        assert lst[0][0] == 0                                                    # slot 0 holds tile 0 (the Q7 swap flag lives there)
    assert set(slots) == set(range(nblocks))


def test_random_shapes():
    rng = np.random.default_rng(5)
    for _ in range(300):
        na, nb, G = int(rng.integers(1, 200000)), int(rng.integers(2, 200000)), int(rng.choice([64, 128, 512, 608]))
        nblocks, tiles = (na + 255) // 256, (nb + 127) // 128
        span = stream_span(nblocks, tiles, G)
        assert span * G >= nblocks * tiles                                       # the grid covers the list
        worst = max(merge_count(rb, tiles, span) for rb in {0, nblocks // 2, nblocks - 1})
        assert worst <= CHUNKS
