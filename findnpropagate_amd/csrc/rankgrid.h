// Rank grid: the collision-free voxel index used by voxelisation and rulebook building.
//
// A (B, D, H, W) cell grid is cut into 4x4x4 blocks, numbered patch by patch (rg_block_of).  Block w owns
//   bits[w]  u64 occupancy, bit = (z&3)*16 + (y&3)*4 + (x&3)
//   base[w]  u32 number of occupied cells in blocks < w   (exclusive popcount scan; defined only
//            where bits[w] != 0 — an empty word is never ranked)
// so   rank(cell) = base[w] + popc(bits[w] & ((1<<bit) - 1))     iff the bit is set.
// summary[i] bit j says bits[64 i + j] != 0: scans, coordinate emission and clearing walk the
// summary and touch only occupied blocks.
// A 3x3x3 neighbourhood touches at most 8 blocks (3.4 on average) instead of 27 hash probes,
// lookups never collide, and ranks enumerate the cells in a spatially blocked order that the
// strided convolutions adopt as their output row order (good L2 locality for the gathers).
// An optional perm[] maps rank -> row for tensors whose row order is fixed by someone else
// (the voxeliser's first-come order).
#pragma once
#include "common.h"
#include <stdlib.h>

struct RankGridDims {
    int B, D, H, W;       // cells
    int bd, bh, bw;       // blocks per axis
    int th, tw;           // 8x8-block (32x32-cell) patches per H / W axis (bh, bw rounded up)
};

__host__ __device__ inline RankGridDims fnp_make_dims(int B, int D, int H, int W) {
    RankGridDims g;
    g.B = B; g.D = D; g.H = H; g.W = W;
    g.bd = (D + 3) >> 2; g.bh = (H + 3) >> 2; g.bw = (W + 3) >> 2;
    g.th = (g.bh + 7) >> 3; g.tw = (g.bw + 7) >> 3;
    return g;
}

__host__ __device__ inline long long fnp_num_blocks(const RankGridDims &g) {
    return (long long)g.B * g.th * g.tw * 64 * g.bd;
}

// device view of a fnp_rankgrid
struct RG {
    RankGridDims d;
    unsigned long long *bits;
    unsigned *base;
    unsigned long long *summ;
    int *perm;
    long long nblk, nsum;
    // counted marks (round 6; fnp.h fnp_rankgrid.counters): cells per UNIT of `wpw` summary words, per group of 64 units, per chunk
    // of 1024 units, laid out cnt[nunits] | gtot[ngroups] | ctot[nchunks]; nullptr = not counted (the prefix counts by itself)
    unsigned *ctr;
    int wpw;
    long long nunits;
    int ctr_levels;   // 1: the marks count per unit only (the totals kernel stays), 3: per unit, group and chunk (development)
};

// summary words per unit of the rank prefix (scan.hip): 64 for the large ~98 % empty grids, 8 / 1 for the small dense ones.  The
// marking kernels and the prefix pass must agree on it: it is a function of the grid's size alone.
// (round 6: 16 instead of 64 for the large grids — with the count pass gone the prefix pass is alone, and 4 x the waves with a
//  quarter of the serial rounds each take it from 87 to 72 us per large grid at 128 scenes; 8: 66 us, but +3 us on the marks)
__host__ __device__ inline int fnp_rg_wpw(long long nsum) { return nsum >= (1ll << 18) ? 16 : nsum >= (1ll << 15) ? 8 : 1; }
// (allocation: sized for the finest split a grid of this size may be given — 8 summary words per unit from 2^15 words on, else 1)
__host__ __device__ inline long long fnp_rg_counter_words(long long nsum) {
    const long long nunits = nsum >= (1ll << 15) ? (nsum + 7) / 8 : nsum;
    return nunits + ((nunits + 63) >> 6) + ((nunits + 1023) >> 10);
}

inline bool fnp_rg_valid(const fnp_rankgrid *g, bool need_perm = false) {
    return g && g->B > 0 && g->D > 0 && g->H > 0 && g->W > 0 && g->bits && g->base && g->summary && (!need_perm || g->perm);
}

inline int fnp_count_levels() {
    static const int v = [] { const char *e = getenv("FNP_COUNT_LEVELS"); return (e && e[0] == '3') ? 3 : 1; }();
    return v;
}

inline RG fnp_rg_view(const fnp_rankgrid *g) {
    RG r;
    r.d = fnp_make_dims(g->B, g->D, g->H, g->W);
    r.bits = (unsigned long long *)g->bits;
    r.base = (unsigned *)g->base;
    r.summ = (unsigned long long *)g->summary;
    r.perm = g->perm;
    r.nblk = fnp_num_blocks(r.d);
    r.nsum = (r.nblk + 63) >> 6;
    r.ctr = (unsigned *)g->counters;
    r.wpw = fnp_rg_wpw(r.nsum);
    {   // development: FNP_WPW_BIG=8 / 16 / 64 — the unit size of the large grids
        static const int big = [] { const char *e = getenv("FNP_WPW_BIG"); return e ? atoi(e) : 0; }();
        if (big > 0 && r.wpw == 16) r.wpw = big;
    }
    r.nunits = (r.nsum + r.wpw - 1) / r.wpw;
    r.ctr_levels = fnp_count_levels();
    return r;
}

// Block order: scene, then 8x8-block patches of the (H, W) plane row-major, then the 64 block
// columns of a patch along a Z-order curve, then the blocks of a column bottom to top.  Ranks (and
// with them the strided convolutions' output rows) therefore run through compact 3-D patches: the
// rows of one convolution tile share most of their neighbours, which the gathers find in L1/L2.
__device__ __forceinline__ unsigned rg_morton3(unsigned v) {   // 3 bits abc -> 0a0b0c
    return (v & 1u) | ((v & 2u) << 1) | ((v & 4u) << 2);
}
__device__ __forceinline__ unsigned rg_unmorton3(unsigned m) { // 0a0b0c -> abc
    return (m & 1u) | ((m >> 1) & 2u) | ((m >> 2) & 4u);
}
__device__ __forceinline__ long long rg_block_of(const RankGridDims &g, int b, int z, int y, int x) {
    const int by = y >> 2, bx = x >> 2;
    const unsigned col = (rg_morton3(by & 7) << 1) | rg_morton3(bx & 7);
    return ((((long long)b * g.th + (by >> 3)) * g.tw + (bx >> 3)) * 64 + col) * g.bd + (z >> 2);
}
__device__ __forceinline__ int rg_bit_of(int z, int y, int x) { return ((z & 3) << 4) | ((y & 3) << 2) | (x & 3); }

// rank -> (b,z,y,x) decode for a (block, bit) pair
__device__ __forceinline__ void rg_decode(const RankGridDims &g, long long blk, int bit, int &b, int &z, int &y, int &x) {
    const int bz = (int)(blk % g.bd); blk /= g.bd;
    const unsigned col = (unsigned)(blk & 63); blk >>= 6;
    const int tx = (int)(blk % g.tw); blk /= g.tw;
    const int ty = (int)(blk % g.th); blk /= g.th;
    b = (int)blk;
    const int by = (ty << 3) | (int)rg_unmorton3(col >> 1), bx = (tx << 3) | (int)rg_unmorton3(col);
    z = (bz << 2) | (bit >> 4);
    y = (by << 2) | ((bit >> 2) & 3);
    x = (bx << 2) | (bit & 3);
}

// Mark the cells `m` of block `blk` occupied.  The plain read first keeps already-set bits (the common
// case once a block has been touched) off the atomic path; exactly one thread sees the word go
// 0 -> non-zero and publishes the block in the summary level.  (Measured: publishing from every thread
// that reads the summary bit as unset — two independent no-return atomics instead of a dependent pair —
// is 2.5x slower; same-address atomics serialise in L2.)
// Returns the number of cells this call newly set (every bit goes 0 -> 1 exactly once, in exactly one call: the returns of all
// calls of a build add up to the number of occupied cells).
__device__ __forceinline__ int rg_mark_mask(const RG &g, long long blk, unsigned long long m) {
    if ((g.bits[blk] & m) == m) return 0;
    const unsigned long long old = atomicOr(&g.bits[blk], m);
    if (old == 0ull) atomicOr(&g.summ[blk >> 6], 1ull << (blk & 63));
    return __popcll(m & ~old);
}
__device__ __forceinline__ void rg_mark(const RG &g, long long blk, int bit) { rg_mark_mask(g, blk, 1ull << bit); }

// COUNTED MARKS (round 6).  The rank prefix used to open with a pass of its own that counted the occupied cells per unit (scan.hip
// PASS 0: 32 us on the stage-1 and stage-2 grids of a 128-scene batch, five launches per forward) and a totals kernel over those
// counts (five more).  The marking kernels know what they set: rg_mark_mask returns the new cells of a call, a workgroup adds them
// up per unit in a small LDS table (its 256 rows are a compact patch: 1-3 units of the large grids, a dozen of the small ones) and
// issues one atomicAdd per unit, per group of 64 units and per chunk of 1024 units it touched — a few hundred adds per counter
// word over a launch.  The prefix is then ONE launch (scan.hip PASS 1 reading these counters).
constexpr int kCntTab = 32;
struct CntTab {
    unsigned key[kCntTab];   // unit id + 1, 0 = free
    unsigned val[kCntTab];
};
__device__ __forceinline__ void rg_count_global(const RG &g, unsigned U, unsigned inc) {
    atomicAdd(&g.ctr[U], inc);
    if (g.ctr_levels >= 3) {
        atomicAdd(&g.ctr[g.nunits + (U >> 6)], inc);
        atomicAdd(&g.ctr[g.nunits + ((g.nunits + 63) >> 6) + (U >> 10)], inc);
    }
}
// Workgroup-level merging of the marks (round 3).  The marking kernels spend three quarters of their time in the atomics
// themselves — read-modify-writes of words of a zeroed, sparse grid that miss L2 (1.4 M misses for 1.05 M atomics per
// launch; a probe that finds every bit already set runs in 25 of 100 us, and fire-and-forget atomics take as long as returning
// ones) — and issue 3.6 x more of them than their rows have distinct (wave, output block) pairs: the run merge along the
// wave covers the first outputs of a row only, not the outputs across a block border.  A workgroup's rows of one pass are
// 256 consecutive ranks, a compact patch: ALL their (block, bits) pairs meet in an LDS table first (compare-and-swap on the
// block id, OR on the bits), and each distinct block costs one atomic in memory.
constexpr int kMarkTabLog = 9, kMarkTab = 1 << kMarkTabLog;
struct MarkTab {
    unsigned long long key[kMarkTab];   // block id, ~0 = free
    unsigned long long val[kMarkTab];
    CntTab cnt;                         // new cells per unit (counted marks, below mark_tab_flush)
};
#ifndef FNP_MARK_TAB
#define FNP_MARK_TAB 1
#endif
__device__ __forceinline__ void rg_count_put(MarkTab *tab, const RG &g, long long blk, int inc) {
    if (!g.ctr || inc <= 0) return;
    const unsigned U = (unsigned)((blk >> 6) / g.wpw);
    unsigned h = (U * 0x9E3779B1u) >> 27;
    for (int probe = 0; probe < 4; ++probe) {
        const unsigned old = atomicCAS(&tab->cnt.key[h], 0u, U + 1u);
        if (old == 0u || old == U + 1u) {
            atomicAdd(&tab->cnt.val[h], (unsigned)inc);
            return;
        }
        h = (h + 1) & (unsigned)(kCntTab - 1);
    }
    rg_count_global(g, U, (unsigned)inc);   // (a crowded table: straight to memory)
}
__device__ __forceinline__ void mark_put(MarkTab *tab, const RG &go, long long blk, unsigned long long m) {
    if (!FNP_MARK_TAB || !tab) {
        const int inc = rg_mark_mask(go, blk, m);
        if (go.ctr && inc > 0) rg_count_global(go, (unsigned)((blk >> 6) / go.wpw), (unsigned)inc);
        return;
    }
    unsigned h = ((unsigned)blk * 0x9E3779B1u) >> (32 - kMarkTabLog);
    for (int probe = 0; probe < 8; ++probe) {
        const unsigned long long old = atomicCAS(&tab->key[h], ~0ull, (unsigned long long)blk);
        if (old == ~0ull || old == (unsigned long long)blk) {
            atomicOr(&tab->val[h], m);
            return;
        }
        h = (h + 1) & (unsigned)(kMarkTab - 1);
    }
    // (a crowded table: straight to memory, the count too.  Not through rg_count_put: with a second inlined copy of its LDS
    //  compare-and-swap loop behind this function's null test of `tab`, hipcc 7.2 fails in instruction selection —
    //  "V_CMP_NE_U32_e32 0, $src_shared_base: Operand has incorrect register class")
    const int inc = rg_mark_mask(go, blk, m);
    if (go.ctr && inc > 0) rg_count_global(go, (unsigned)((blk >> 6) / go.wpw), (unsigned)inc);
}
// every thread of the workgroup: empty the table / write its blocks out and empty it (barriers inside)
__device__ __forceinline__ void mark_tab_init(MarkTab *tab, int tid, int nthreads) {
    for (int i = tid; i < kMarkTab; i += nthreads) {
        tab->key[i] = ~0ull;
        tab->val[i] = 0ull;
    }
    if (tid < kCntTab) tab->cnt.key[tid] = tab->cnt.val[tid] = 0u;
    __syncthreads();
}
__device__ __forceinline__ void mark_tab_flush(MarkTab *tab, const RG &go, int tid, int nthreads) {
    __syncthreads();
    for (int i = tid; i < kMarkTab; i += nthreads) {
        const unsigned long long k = tab->key[i];
        if (k != ~0ull) {
            rg_count_put(tab, go, (long long)k, rg_mark_mask(go, (long long)k, tab->val[i]));
            tab->key[i] = ~0ull;
            tab->val[i] = 0ull;
        }
    }
    __syncthreads();
    if (go.ctr) {   // (uniform) the workgroup's new cells per unit: one add per unit, group and chunk
        if (tid < kCntTab && tab->cnt.key[tid] != 0u) {
            rg_count_global(go, tab->cnt.key[tid] - 1u, tab->cnt.val[tid]);
            tab->cnt.key[tid] = tab->cnt.val[tid] = 0u;
        }
        __syncthreads();
    }
}


// Row of cell (b,z,y,x) or -1.  Caller guarantees the cell is inside the grid.
__device__ __forceinline__ int rg_lookup(const RG &g, int b, int z, int y, int x) {
    const long long blk = rg_block_of(g.d, b, z, y, x);
    const unsigned long long w = g.bits[blk];
    const int bit = rg_bit_of(z, y, x);
    if (!((w >> bit) & 1ull)) return -1;
    const int r = (int)g.base[blk] + __popcll(w & ((1ull << bit) - 1ull));
    return g.perm ? g.perm[r] : r;
}

// One thread per output row for all K = KZ*KY*KX offsets (kernel sizes 1 or 3 per axis): the input
// cells of a row span at most two blocks per axis, so the thread loads the <= 8 occupancy words and
// prefixes once (independent loads, all in flight together) and resolves the K cells with bit
// operations, instead of K threads each re-reading the coordinates and one word.  Stores stay
// coalesced: for a fixed offset, consecutive threads write consecutive entries of nbr[k][.].
// lo = first input cell per axis (SubM: c - k/2; strided: c*s - p).
// The K entries of a row go to a wave-private LDS strip ([k][sstride]); rulebook.hip's nbr_flush then writes the strip out 16 bytes
// per lane, four offsets (4 x 256 contiguous bytes) per wave instruction instead of one: the 4-byte form was
// bound by the number of store instructions, not by bytes.
// mask_out (optional): bit k = the row has a neighbour at offset k (K <= 32).
// DEFER (default): all K ranks are formed first and the permutation loads of a stage-1 grid leave together (the record builders, the
// plain table kernels, the in-kernel rulebooks of the strided convolutions: -13 ... -24 % / -15 % / -2 ... -3 %).  DEFER = false: every
// entry leaves for the strip as soon as it is known — the tile-rulebook kernels, where 27 live ranks cost a wave per SIMD (+5 ... +7 %)
// and the grids carry no permutation.
template <int KZ, int KY, int KX, bool DEFER = true>
__device__ __forceinline__ void nbr_row(const RG &g, int b, int loz, int loy, int lox, int *strip, int sstride, unsigned *mask_out = nullptr) {   // (strip: LDS shared by the lanes of a wave - no __restrict__)
    // Round 5: BRANCH-FREE.  The first form tested every entry under three nested `if`s and, for a grid with a rank -> row
    // permutation (stage 1), loaded perm[rank] inside the innermost one: hipcc turned that into ~200 exec-mask branches per row and,
    // worse, a load + s_waitcnt vmcnt(0) PER ENTRY — 27 serial memory round trips per row in the stage-1 record builders.  Now every
    // rank is computed unconditionally (bit arithmetic on words already in registers), absent entries are selected to -1, and the
    // permutation loads of all K entries are issued together behind ONE uniform branch.
    const int bz0 = loz >> 2, by0 = loy >> 2, bx0 = lox >> 2;   // (arithmetic shift: -1 -> block -1, outside)
    unsigned long long w[2][2][2];
    unsigned base[2][2][2];
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
            const int by = by0 + cy, bx = bx0 + cx;
            const bool need_yx = (cy == 0 || ((loy + KY - 1) >> 2) != by0) && (cx == 0 || ((lox + KX - 1) >> 2) != bx0);
            const bool in_yx = by >= 0 && by < g.d.bh && bx >= 0 && bx < g.d.bw;
            // the blocks of a column are numbered bottom to top: one column base, + bz
            const unsigned col = (rg_morton3((unsigned)by & 7u) << 1) | rg_morton3((unsigned)bx & 7u);
            const long long colbase = ((((long long)b * g.d.th + (by >> 3)) * g.d.tw + (bx >> 3)) * 64 + col) * g.d.bd;
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                const int bz = bz0 + cz;
                const bool need = need_yx && (cz == 0 || ((loz + KZ - 1) >> 2) != bz0);
                const bool in = in_yx && bz >= 0 && bz < g.d.bd;
                unsigned long long ww = 0ull;
                unsigned bb = 0u;
                if (need && in) {
                    ww = g.bits[colbase + bz];
                    bb = g.base[colbase + bz];   // (defined only where ww != 0; unused otherwise)
                }
                w[cz][cy][cx] = ww;
                base[cz][cy][cx] = bb;
            }
        }
    constexpr int K = KZ * KY * KX;
    unsigned msk = 0u;
    int rk[DEFER ? K : 1];
    // A LINE of the block (fixed z, y) is a nibble of its word; the KX cells of a row's line lie in the nibbles of the two x-adjacent
    // words: one byte `lb` (cell lox - (lox & 3) + i at bit i).  Per line: the byte and the two "cells before the line" prefixes
    // (a 64-bit mask and popcount each); per entry a 32-bit and / popcount / select (round 5, after the branch-free form: the 27
    // entries each built their own 64-bit mask and popcount — ~22 vector instructions per entry, 7 now).
    const int px0 = lox & 3;
    unsigned bitx[KX], mA[KX];
    bool selx[KX], vx[KX];
#pragma unroll
    for (int jx = 0; jx < KX; ++jx) {
        const int pj = px0 + jx, x = lox + jx;             // position in the byte: 0 .. 3 + KX - 1
        selx[jx] = pj >= 4;                                // the cell lies in the second word
        bitx[jx] = 1u << pj;
        mA[jx] = selx[jx] ? ((bitx[jx] - 1u) & 0xF0u) : (bitx[jx] - 1u);   // the cells of ITS nibble before it
        vx[jx] = x >= 0 && x < g.d.W;
    }
#pragma unroll
    for (int jz = 0; jz < KZ; ++jz) {
        const int z = loz + jz;
        const bool cz = (z >> 2) != bz0;
        const bool vz = z >= 0 && z < g.d.D;
        unsigned long long wz[2][2];
        unsigned bzv[2][2];
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                wz[cy][cx] = cz ? w[1][cy][cx] : w[0][cy][cx];
                bzv[cy][cx] = cz ? base[1][cy][cx] : base[0][cy][cx];
            }
#pragma unroll
        for (int jy = 0; jy < KY; ++jy) {
            const int y = loy + jy;
            const bool cy = (y >> 2) != by0;
            const bool vy = vz && y >= 0 && y < g.d.H;
            const unsigned long long wy0 = cy ? wz[1][0] : wz[0][0], wy1 = cy ? wz[1][1] : wz[0][1];
            const unsigned by0v = cy ? bzv[1][0] : bzv[0][0], by1v = cy ? bzv[1][1] : bzv[0][1];
            const int line = ((z & 3) << 4) | ((y & 3) << 2);
            const unsigned long long below = (1ull << line) - 1ull;
            const unsigned pre0 = by0v + (unsigned)__popcll(wy0 & below), pre1 = by1v + (unsigned)__popcll(wy1 & below);
            const unsigned lb = ((unsigned)(wy0 >> line) & 0xFu) | (((unsigned)(wy1 >> line) & 0xFu) << 4);
#pragma unroll
            for (int jx = 0; jx < KX; ++jx) {
                const bool hit = vy && vx[jx] && (lb & bitx[jx]) != 0u;
                const int r = (int)((selx[jx] ? pre1 : pre0) + (unsigned)__popc(lb & mA[jx]));
                const int e = (jz * KY + jy) * KX + jx;
                if constexpr (DEFER) {
                    rk[e] = hit ? r : -1;
                } else {
                    int rr = hit ? r : -1;
                    if (g.perm && hit) rr = g.perm[r];
                    strip[e * sstride] = rr;
                    if (mask_out && hit) msk |= 1u << (e & 31);
                }
            }
        }
    }
    if constexpr (DEFER) {
        if (g.perm) {   // (uniform) rank -> row: all K loads in flight together
            int pr[K];
#pragma unroll
            for (int k = 0; k < K; ++k) pr[k] = g.perm[rk[k] < 0 ? 0 : rk[k]];
#pragma unroll
            for (int k = 0; k < K; ++k) rk[k] = rk[k] < 0 ? -1 : pr[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            strip[k * sstride] = rk[k];
            if (mask_out && rk[k] >= 0) msk |= 1u << (k & 31);
        }
    }
    if (mask_out) *mask_out = msk;
}


// strip -> nbr[k][o0 .. o0 + 63] for the K offsets (table stride `cap`); rows >= n are never written
template <int K>
__device__ __forceinline__ void nbr_flush(const int *__restrict__ strip_wave, int o0, int n, int cap, int *__restrict__ nbr) {
    const int lane = fnp_lane();
    if ((cap & 3) == 0) {   // 16-byte aligned rows of the table (wave-uniform)
#pragma unroll
        for (int j = 0; j < (K * 16 + 63) / 64; ++j) {
            const int e = j * 64 + lane, k = e >> 4, c = e & 15;
            if (k < K) {
                const int4 v = *reinterpret_cast<const int4 *>(strip_wave + k * 64 + c * 4);
                const int o = o0 + c * 4;
                if (o + 3 < n) *reinterpret_cast<int4 *>(nbr + (size_t)k * cap + o) = v;
                else {
                    if (o < n) nbr[(size_t)k * cap + o] = v.x;
                    if (o + 1 < n) nbr[(size_t)k * cap + o + 1] = v.y;
                    if (o + 2 < n) nbr[(size_t)k * cap + o + 2] = v.z;
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (o0 + lane < n) nbr[(size_t)k * cap + o0 + lane] = strip_wave[k * 64 + lane];
    }
}


// Device-wide exclusive scans (scan.hip).
namespace fnp_scan {
constexpr int kTile = 4096;  // elements per workgroup (256 threads x 16)
// out[i] = exclusive prefix of in[i] (int32 flags or counts); *total = sum.  in == out allowed.
int int32(const int *in, long long n, int *out, int *total, void *ws, hipStream_t s);
long long workspace_bytes(long long n);
// base[w] for every occupied block of the grid, *total = number of occupied cells.  With out_coords the
// (b, z, y, x) of every occupied cell is written at its rank (rows >= cap_out dropped) in the same sweep.
// counted: the grid's counters (g.ctr) hold the cells per unit / group / chunk already (counted marks): one launch instead of three.
int rank_grid(const RG &g, int *total, void *ws, hipStream_t s, int *out_coords = nullptr, int cap_out = 0, bool counted = false);
long long rank_grid_workspace_bytes(long long nsum);
}  // namespace fnp_scan
