// fnp_xcd_map: the XCD-contiguous workgroup renumbering of common.h (see the comment there), kept free of HIP headers so that
// tests/test_host_logic_r5.py compiles THIS file with g++ and checks the bijection and the one-run-per-XCD property on the
// shipped arithmetic (a hand copy in the test could drift).
#pragma once
#ifndef FNP_XCD_SWZ
#define FNP_XCD_SWZ 1
#endif
// ASSUMES A 1-D GRID, or a 2-D one whose gridDim.x is a multiple of 8: the hardware's round-robin runs over the LINEAR workgroup
// id blockIdx.x + blockIdx.y * gridDim.x, so with gridDim.x % 8 != 0 the rows y > 0 start on another XCD and the runs are no
// longer one per XCD (still a bijection per row — results cannot change, only the locality the renumbering exists for).  The one
// 2-D launch that uses it, rg_clear_multi_kernel, rounds its gridDim.x up to a multiple of 8.
#if defined(__HIPCC__) || defined(__CUDACC__)
#define FNP_HD __host__ __device__
#else
#define FNP_HD
#endif
FNP_HD inline unsigned fnp_xcd_map(unsigned G, unsigned b) {
    if (!FNP_XCD_SWZ || G < 16u) return b;
    const unsigned per = G >> 3, rem = G & 7u, x = b & 7u, sl = b >> 3;
    return (x < rem ? x * (per + 1u) : rem * (per + 1u) + (x - rem) * per) + sl;
}
