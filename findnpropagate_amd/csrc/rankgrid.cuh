// Rank grid: the collision-free voxel index used by voxelisation and rulebook building.
//
// A (B, D, H, W) cell grid is cut into 4x4x4 blocks.  Block w owns
//   bits[w]  u64 occupancy, bit = (z&3)*16 + (y&3)*4 + (x&3)
//   base[w]  u32 number of occupied cells in blocks < w          (exclusive popcount scan)
// so   rank(cell) = base[w] + popc(bits[w] & ((1<<bit) - 1))     iff the bit is set.
// A 3x3x3 neighbourhood touches at most 8 blocks (3.4 on average) instead of 27 hash probes,
// lookups never collide, and ranks enumerate the cells in a spatially blocked order that the
// strided convolutions adopt as their output row order (good L2 locality for the gathers).
// An optional perm[] maps rank -> row for tensors whose row order is fixed by someone else
// (the voxeliser's first-come order).
#pragma once
#include "common.h"

struct RankGridDims {
    int B, D, H, W;       // cells
    int bd, bh, bw;       // blocks per axis
};

__host__ __device__ inline RankGridDims fnp_make_dims(int B, int D, int H, int W) {
    RankGridDims g;
    g.B = B; g.D = D; g.H = H; g.W = W;
    g.bd = (D + 3) >> 2; g.bh = (H + 3) >> 2; g.bw = (W + 3) >> 2;
    return g;
}

__host__ __device__ inline long long fnp_num_blocks(const RankGridDims &g) {
    return (long long)g.B * g.bd * g.bh * g.bw;
}

__device__ __forceinline__ long long rg_block_of(const RankGridDims &g, int b, int z, int y, int x) {
    return (((long long)b * g.bd + (z >> 2)) * g.bh + (y >> 2)) * g.bw + (x >> 2);
}
__device__ __forceinline__ int rg_bit_of(int z, int y, int x) { return ((z & 3) << 4) | ((y & 3) << 2) | (x & 3); }

// rank -> (b,z,y,x) decode for a (block, bit) pair
__device__ __forceinline__ void rg_decode(const RankGridDims &g, long long blk, int bit, int &b, int &z, int &y, int &x) {
    const int bx = (int)(blk % g.bw); blk /= g.bw;
    const int by = (int)(blk % g.bh); blk /= g.bh;
    const int bz = (int)(blk % g.bd); blk /= g.bd;
    b = (int)blk;
    z = (bz << 2) | (bit >> 4);
    y = (by << 2) | ((bit >> 2) & 3);
    x = (bx << 2) | (bit & 3);
}

// Row of cell (b,z,y,x) or -1.  Caller guarantees the cell is inside the grid.
__device__ __forceinline__ int rg_lookup(const RankGridDims &g, const unsigned long long *__restrict__ bits,
                                         const unsigned *__restrict__ base, const int *__restrict__ perm,
                                         int b, int z, int y, int x) {
    const long long blk = rg_block_of(g, b, z, y, x);
    const unsigned long long w = bits[blk];
    const int bit = rg_bit_of(z, y, x);
    if (!((w >> bit) & 1ull)) return -1;
    const int r = (int)base[blk] + __popcll(w & ((1ull << bit) - 1ull));
    return perm ? perm[r] : r;
}

// Device-wide exclusive scan (three launches).  See scan.hip.
namespace fnp_scan {
constexpr int kTile = 4096;  // elements per workgroup (256 threads x 16)
// out[i] = exclusive prefix of popc(bits[i]); *total = sum.  ws: fnp_scan_workspace_bytes(n).
int popcount_u64(const unsigned long long *bits, long long n, unsigned *out, int *total, void *ws, hipStream_t s);
// out[i] = exclusive prefix of flags[i] (int32 0/1 or counts); *total = sum.
int int32(const int *in, long long n, int *out, int *total, void *ws, hipStream_t s);
long long workspace_bytes(long long n);
}  // namespace fnp_scan
