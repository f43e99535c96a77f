// Geometry of the tile rulebook (include/fnp.h: fnp_tile_rulebook_build) shared by its two writers — the stand-alone
// restatement of an int32 table (spconv_tile.hip) and the SubM rulebook kernel that emits both forms at once
// (rulebook.hip) — and by the convolutions that read it (spconv_tile.hip).
#pragma once
#include "common.h"

#ifndef FNP_TILE32_MB
#define FNP_TILE32_MB 2   // (4: four consumer waves x 64 rows — built, bit-identical, 5 % slower: round 5, spconv_tile.hip)
#endif

namespace tilerb {

constexpr int kK = 27;                  // 3x3x3 kernels only
constexpr unsigned kEscape = 0xFFFFu;   // entry: not in the tile image — fetch through the int32 table

// An entry is the LDS byte address, inside the tile image, of a feature row with the row's swizzle in its low bits: a
// lane reads logical 16-byte chunk c at entry ^ (c << 4).  Image rows: the window (rows [tile - HALO, tile + TILE + HALO)
// of the input, even rows first, then odd rows: the neighbours of a consumer wave's 16 columns — every other row — are then
// 16 consecutive image rows), OVF overflow rows (far neighbours, one slot per distinct row), one row of zeros.
struct G32 {   // 32 channels: 64-byte rows.  (window +-64 with 128 overflow rows fits LDS too and was slower: 0.20 vs 0.18 ms per
               // layer, rulebook pass 0.27 vs 0.20 ms — 39 distinct far rows crowd a 128-slot table)
    // SPLIT = 16-row blocks a consumer wave of spconv_tile32_kernel owns (FNP_TILE32_MB): its MFMA column l15 of block mb is tile row
    // SPLIT l15 + mb of the wave's rows, so the neighbours of one block are rows of ONE residue mod SPLIT: the window keeps the
    // residues in separate parts, and what a fragment read touches is 16 consecutive image rows.
    static constexpr int TILE = FNP_TILE_ROWS, HALO = 32, WIN = TILE + 2 * HALO, OVF = 256, ZERO = WIN + OVF, ROWB = 64, SPLIT = FNP_TILE32_MB;
    static constexpr int REC_FAR = kK * TILE * 2, REC_ESC = REC_FAR + OVF * 4, REC = REC_ESC + 16;
    // the row at slot rs stores logical chunk c (0..3) at chunk c ^ (-(rs >> 2) & 3)
    __host__ __device__ static constexpr unsigned code(unsigned rs) { return rs * ROWB + (((0u - (rs >> 2)) & 3u) << 4); }
};
struct G64 {   // 64 channels: 128-byte rows
    static constexpr int TILE = FNP_TILE64_ROWS, HALO = 64, WIN = TILE + 2 * HALO, OVF = 128, ZERO = WIN + OVF, ROWB = 128, SPLIT = 2;
    static constexpr int REC_FAR = kK * TILE * 2, REC_ESC = REC_FAR + OVF * 4, REC = REC_ESC + 16;
    // the row at slot rs stores logical chunk c (0..7) at chunk c ^ ((rs >> 1) & 7)
    __host__ __device__ static constexpr unsigned code(unsigned rs) { return rs * ROWB + (((rs >> 1) & 7u) << 4); }
};
static_assert(G32::REC == FNP_TILE_RECORD_BYTES && G32::REC % 16 == 0 && G32::ZERO * G32::ROWB + 48 < 0xFFFF, "32-channel tile record");
static_assert(G64::REC == FNP_TILE64_RECORD_BYTES && G64::REC % 16 == 0 && G64::ZERO * G64::ROWB + 112 < 0xFFFF, "64-channel tile record");
static_assert(G32::HALO % 32 == 0 && (G32::WIN / G32::SPLIT) % 4 == 0 && G64::HALO % 32 == 0 && (G64::WIN / 2) % 8 == 0, "window parts keep the swizzle period");
template <typename G> __host__ __device__ constexpr unsigned win_slot(unsigned d) { return (d % (unsigned)G::SPLIT) * (G::WIN / G::SPLIT) + d / (unsigned)G::SPLIT; }
// first slot a far row probes: multiplicative, so that the runs of consecutive row ids far neighbours come in do not pile up
// into one long cluster of the linear-probing table (with id & (OVF - 1) the 64-channel build spent 0.12 ms in probes)
template <typename G> __host__ __device__ constexpr unsigned far_hash(int id) {
    const unsigned h = (unsigned)id * 0x9E3779B1u;
    return h >> (32 - (G::OVF == 256 ? 8 : G::OVF == 128 ? 7 : 6));
}
template <typename G> __host__ __device__ constexpr unsigned far_next(unsigned h) { return h + 1 == (unsigned)G::OVF ? 0u : h + 1; }
static_assert((G32::OVF == 256 || G32::OVF == 128) && G64::OVF == 128, "far_hash knows these table sizes");

// The entry of neighbour row `id` (-1: none) of a row of the tile whose window starts at row wlo.  Far rows take a slot
// of `table` (G::OVF ints, -1 = free, shared by the threads working on the tile): compare-and-swap with linear probing —
// which slot a row gets depends on the order the threads arrive in, the BYTES the convolution reads through it do not.
template <typename G>
__device__ __forceinline__ unsigned entry_of(int id, int wlo, int *table) {
    if (id < 0) return G::code(G::ZERO);
    const unsigned d = (unsigned)(id - wlo);
    if (d < (unsigned)G::WIN) return G::code(win_slot<G>(d));
    unsigned h = far_hash<G>(id);
    for (int probe = 0; probe < 64; ++probe) {
        const int old = atomicCAS(&table[h], -1, id);
        if (old == -1 || old == id) return G::code((unsigned)G::WIN + h);
        h = far_next<G>(h);
    }
    return kEscape;
}

// The 27 entries of one row at once, for the kernels that restate whole rows: window rows and absent neighbours through a
// table (lut[min(id - wlo, WIN)], G::WIN + 1 entries made by fill_lut), the first probes of all far rows issued back to
// back (one LDS round trip for the row instead of one per far entry), stragglers one by one.  Returns whether any entry
// escaped.
template <typename G>
__device__ __forceinline__ void fill_lut(unsigned short *lut, int tid, int nthreads) {
    for (int d = tid; d <= G::WIN; d += nthreads) lut[d] = (unsigned short)(d < G::WIN ? G::code(win_slot<G>((unsigned)d)) : G::code(G::ZERO));
}
template <typename G>
__device__ __forceinline__ bool entries_of_row(const int (&id)[kK], int wlo, int *table, const unsigned short *lut, unsigned (&code)[kK]) {
    int old[kK];
    unsigned far = 0;
#pragma unroll
    for (int k = 0; k < kK; ++k) {
        const unsigned d = (unsigned)(id[k] - wlo);
        code[k] = lut[min(d, (unsigned)G::WIN)];
        old[k] = 0;
        if (d >= (unsigned)G::WIN && id[k] != -1) {
            far |= 1u << k;
            old[k] = atomicCAS(&table[far_hash<G>(id[k])], -1, id[k]);
        }
    }
    bool esc = false;
    if (far) {
#pragma unroll
        for (int k = 0; k < kK; ++k) {
            if (far & (1u << k)) {
                unsigned h = far_hash<G>(id[k]);
                bool ok = old[k] == -1 || old[k] == id[k];
#pragma unroll 1
                for (int probe = 1; probe < 64 && !ok; ++probe) {
                    h = far_next<G>(h);
                    const int o2 = atomicCAS(&table[h], -1, id[k]);
                    ok = o2 == -1 || o2 == id[k];
                }
                code[k] = ok ? G::code((unsigned)G::WIN + h) : kEscape;
                esc |= !ok;
            }
        }
    }
    return esc;
}

}  // namespace tilerb
