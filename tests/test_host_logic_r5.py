"""Host-side restatements of small device-side index maps introduced in round 5 (no GPU needed)."""
import numpy as np
import pytest


def xcd_block(b, G):
    """findnpropagate_amd/csrc/common.h fnp_xcd_block(): the logical block of hardware workgroup b of a G-workgroup launch."""
    if G < 16:
        return b
    per, rem, x, sl = G >> 3, G & 7, b & 7, b >> 3
    return (x * (per + 1) if x < rem else rem * (per + 1) + (x - rem) * per) + sl


@pytest.mark.parametrize("G", [1, 7, 15, 16, 17, 23, 64, 255, 256, 257, 2047, 2048])
def test_xcd_contiguous_workgroup_order_is_a_bijection_with_one_run_per_xcd(G):
    """Workgroup b runs on XCD b & 7; the renumbering must hit every logical block exactly once (results cannot change) and
    give every XCD ONE contiguous run of logical blocks (that is what it is for)."""
    lb = np.array([xcd_block(b, G) for b in range(G)])
    assert np.array_equal(np.sort(lb), np.arange(G))
    if G >= 16:
        for x in range(8):
            run = np.sort(lb[np.arange(G) % 8 == x])
            assert len(run) in (G // 8, G // 8 + 1) and np.array_equal(run, np.arange(run[0], run[0] + len(run)))
        firsts = [lb[np.arange(G) % 8 == x].min() for x in range(8)]
        assert firsts == sorted(firsts)          # XCD 0 owns the first run, XCD 7 the last
