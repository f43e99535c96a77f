"""Host-side restatements of small device-side index maps introduced in round 5 (no GPU needed)."""
import numpy as np
import pytest


_XCD_SRC = r"""
#include <cstdio>
#include <cstdlib>
#include "xcdmap.h"
int main(int argc, char **argv) {
    for (int a = 1; a < argc; ++a) {
        const unsigned G = (unsigned)atoi(argv[a]);
        for (unsigned b = 0; b < G; ++b) printf("%u ", fnp_xcd_map(G, b));
        printf("\n");
    }
    return 0;
}
"""
_XCD_GS = [1, 7, 15, 16, 17, 23, 64, 255, 256, 257, 2047, 2048]


@pytest.fixture(scope="module")
def xcd_tables(tmp_path_factory):
    """fnp_xcd_map AS SHIPPED (findnpropagate_amd/csrc/xcdmap.h, the function fnp_xcd_block() calls on the device), compiled for
    the host with g++ and run for every grid size of the test: {G: [logical block of hardware workgroup b]}"""
    import os
    import subprocess
    d = tmp_path_factory.mktemp("xcd")
    src, exe = d / "xcd.cpp", d / "xcd"
    src.write_text(_XCD_SRC)
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "findnpropagate_amd", "csrc")
    subprocess.run(["g++", "-O1", "-std=c++17", f"-I{inc}", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)] + [str(g) for g in _XCD_GS], check=True, capture_output=True, text=True).stdout
    return {g: [int(v) for v in line.split()] for g, line in zip(_XCD_GS, out.splitlines())}


@pytest.mark.parametrize("G", _XCD_GS)
def test_xcd_contiguous_workgroup_order_is_a_bijection_with_one_run_per_xcd(G, xcd_tables):
    """Workgroup b runs on XCD b & 7; the renumbering must hit every logical block exactly once (results cannot change) and
    give every XCD ONE contiguous run of logical blocks (that is what it is for)."""
    lb = np.array(xcd_tables[G])
    assert np.array_equal(np.sort(lb), np.arange(G))
    if G >= 16:
        for x in range(8):
            run = np.sort(lb[np.arange(G) % 8 == x])
            assert len(run) in (G // 8, G // 8 + 1) and np.array_equal(run, np.arange(run[0], run[0] + len(run)))
        firsts = [lb[np.arange(G) % 8 == x].min() for x in range(8)]
        assert firsts == sorted(firsts)          # XCD 0 owns the first run, XCD 7 the last
