// Device-wide exclusive scans (wave shuffles + LDS) used for order-preserving compaction:
// first-point flags of the voxeliser (int32 scan: tile reduce -> scan of tile sums -> tile scan)
// and the rank-grid popcount prefix, which walks the summary level so that only occupied
// blocks are read.
#include "rankgrid.h"
#include <cstdint>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int kThreads = 256;
constexpr int kItems = fnp_scan::kTile / kThreads;  // 16

struct LoadInt {
    const int *p;
    __device__ __forceinline__ unsigned operator()(long long i) const { return (unsigned)p[i]; }
};

__device__ __forceinline__ unsigned wave_inclusive(unsigned v) {
    const int lane = fnp_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across the 256-thread workgroup; returns the
// exclusive prefix, *total receives the workgroup sum.  wsum: 4 LDS words.
__device__ __forceinline__ unsigned block_exclusive(unsigned v, unsigned *wsum, unsigned &total) {
    const int wave = threadIdx.x >> 6, lane = fnp_lane();
    const unsigned inc = wave_inclusive(v);
    __syncthreads();  // protect wsum reuse
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned off = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
        const unsigned s = wsum[w];
        if (w < wave) off += s;
    }
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return off + inc - v;
}

template <class L>
__global__ __launch_bounds__(kThreads) void tile_reduce_kernel(L load, long long n, unsigned *__restrict__ partial) {
    __shared__ unsigned wsum[4];
    const long long base = (long long)blockIdx.x * fnp_scan::kTile;
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const long long i = base + j * kThreads + threadIdx.x;
        if (i < n) s += load(i);
    }
    unsigned total;
    block_exclusive(s, wsum, total);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// one workgroup: in-place exclusive scan of the tile sums, grand total to *total.
__global__ __launch_bounds__(kThreads) void partial_scan_kernel(unsigned *__restrict__ partial, int np,
                                                                int *__restrict__ total_out) {
    __shared__ unsigned wsum[4];
    unsigned carry = 0;
    for (int base = 0; base < np; base += kThreads) {
        const int i = base + threadIdx.x;
        const unsigned v = i < np ? partial[i] : 0u;
        unsigned total;
        const unsigned ex = block_exclusive(v, wsum, total);
        if (i < np) partial[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = (int)carry;
}

template <class L, class TOut>
__global__ __launch_bounds__(kThreads) void tile_scan_kernel(L load, long long n, const unsigned *__restrict__ partial,
                                                             TOut *__restrict__ out) {
    __shared__ unsigned wsum[4];
    const long long base = (long long)blockIdx.x * fnp_scan::kTile;
    unsigned carry = partial[blockIdx.x];
#pragma unroll 1
    for (int j = 0; j < kItems; ++j) {
        const long long i = base + j * kThreads + threadIdx.x;
        const unsigned v = i < n ? load(i) : 0u;
        unsigned total;
        const unsigned ex = block_exclusive(v, wsum, total);
        if (i < n) out[i] = (TOut)(carry + ex);
        carry += total;
    }
}

// SMALL INPUTS (one scene: the reference's extraction runs batch size 1, where a forward is a chain of ~50 launches of a few
// microseconds each): up to kSmallTiles tiles in ONE launch.  Workgroup b sums everything in front of its tile itself (16-byte
// loads, <= 60 per thread, all L2 hits) instead of waiting for a tile-sum pass and a scan of the tile sums, then scans its tile
// four elements per thread and round; the last workgroup writes the grand total.  Same integers as the three-launch form.
constexpr int kSmallTiles = 16;
__global__ __launch_bounds__(kThreads) void small_scan_kernel(const int *__restrict__ in, int n, int *__restrict__ out, int *__restrict__ total_out) {
    __shared__ unsigned wsum[4];
    const int base = blockIdx.x * fnp_scan::kTile;
    const int4 *in4 = reinterpret_cast<const int4 *>(in);
    unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int c = threadIdx.x; c < base / 4; c += 4 * kThreads) {   // (base is a multiple of 4096: whole 16-byte chunks, all < n)
        const int4 a = in4[c];
        const int4 b = c + kThreads < base / 4 ? in4[c + kThreads] : make_int4(0, 0, 0, 0);
        const int4 d = c + 2 * kThreads < base / 4 ? in4[c + 2 * kThreads] : make_int4(0, 0, 0, 0);
        const int4 e = c + 3 * kThreads < base / 4 ? in4[c + 3 * kThreads] : make_int4(0, 0, 0, 0);
        s0 += (unsigned)(a.x + a.y + a.z + a.w);
        s1 += (unsigned)(b.x + b.y + b.z + b.w);
        s2 += (unsigned)(d.x + d.y + d.z + d.w);
        s3 += (unsigned)(e.x + e.y + e.z + e.w);
    }
    unsigned carry;
    block_exclusive(s0 + s1 + s2 + s3, wsum, carry);   // (carry = the workgroup total = everything in front of the tile)
    int4 v[kItems / 4];
#pragma unroll
    for (int j = 0; j < kItems / 4; ++j) {
        const int i = base + (j * kThreads + threadIdx.x) * 4;
        v[j] = make_int4(0, 0, 0, 0);
        if (i + 3 < n) v[j] = in4[i >> 2];
        else {
            if (i < n) v[j].x = in[i];
            if (i + 1 < n) v[j].y = in[i + 1];
            if (i + 2 < n) v[j].z = in[i + 2];
        }
    }
#pragma unroll
    for (int j = 0; j < kItems / 4; ++j) {
        const int i = base + (j * kThreads + threadIdx.x) * 4;
        unsigned tot;
        const unsigned ex = carry + block_exclusive((unsigned)(v[j].x + v[j].y + v[j].z + v[j].w), wsum, tot);
        const int4 o = make_int4((int)ex, (int)(ex + (unsigned)v[j].x), (int)(ex + (unsigned)(v[j].x + v[j].y)), (int)(ex + (unsigned)(v[j].x + v[j].y + v[j].z)));
        if (i + 3 < n) reinterpret_cast<int4 *>(out)[i >> 2] = o;
        else {
            if (i < n) out[i] = o.x;
            if (i + 1 < n) out[i + 1] = o.y;
            if (i + 2 < n) out[i + 2] = o.z;
        }
        carry += tot;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = (int)carry;
}

template <class L, class TOut>
int run_scan(L load, long long n, TOut *out, int *total, void *ws, hipStream_t s) {
    if (n < 0 || !total) return FNP_ERR_ARG;
    if (n == 0) {
        return fnp_fill_words(total, 1, 0u, s);
    }
    if (!out || !ws) return FNP_ERR_ARG;
    const int tiles = fnp_divup(n, fnp_scan::kTile);
    if constexpr (std::is_same<L, LoadInt>::value && std::is_same<TOut, int>::value) {
        static const bool small_ok = [] { const char *e = getenv("FNP_SMALL_SCAN"); return !e || atoi(e) != 0; }();   // (development switch)
        if (small_ok && tiles <= kSmallTiles && (const int *)out != load.p && (((uintptr_t)load.p | (uintptr_t)out) & 15) == 0) {   // (not in place: a workgroup reads the tiles in front of its own)
            hipLaunchKernelGGL(small_scan_kernel, dim3(tiles), dim3(kThreads), 0, s, load.p, (int)n, out, total);
            FNP_LAUNCH_CHECK();
            return FNP_OK;
        }
    }
    unsigned *partial = (unsigned *)ws;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_reduce_kernel<L>), dim3(tiles), dim3(kThreads), 0, s, load, n, partial);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_scan_kernel, dim3(1), dim3(kThreads), 0, s, partial, tiles, total);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_scan_kernel<L, TOut>), dim3(tiles), dim3(kThreads), 0, s, load, n, partial, out);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// ---- rank-grid prefix through the summary level ---------------------------------------------
// One wave per UNIT of WPW consecutive summary words (64 WPW blocks): lane j < WPW loads word j of the
// unit (one coalesced read), a ballot names the non-zero words, and only those are visited — four at a
// time, so that the occupancy-word loads of a round are in flight together.  WPW = 64 for the large,
// ~98 % empty grids (the 41 x 1440 x 1440 lidar grid of a 32-scene batch has 713 k summary words: a wave
// per word spent 85 us launching waves that retire after one load); WPW = 1 for the small dense grids of
// the later stages, where a wave per word is the parallel form.  Three kernels, none of them a scan over
// the per-unit counts (round 3, measured and dropped: one-workgroup forms for the small grids and the 30 k point flags of a
// one-scene forward — a launch fewer or two each, but a lone 1024-thread workgroup needs 16-65 us for what these three launches
// do in 15: its two dependent memory round trips and three barriers run with nothing to hide them):
//   PASS 0 : cnt[U]  = occupied cells in the blocks of unit U             (no scan: a per-lane sum)
//   TOTALS : gtot[g] = sum of cnt over group g (64 units), ctot[c] = over chunk c (16 groups);
//            one workgroup per chunk, coalesced
//   PASS 1 : base[w] = cells before U + cells in the earlier words of U + cells in the blocks of w's word
//            before w, where "cells before U" = sum(ctot[< c]) + sum(gtot[16c .. g)) + sum(cnt[64g .. U))
//            (lane-parallel reads, reduced once per wave) and the words of U are walked in order with a
//            running sum;  *total = sum(ctot)  (wave of U = 0).
//            With out_coords the coordinates of the occupied cells are written at their ranks in the
//            same sweep (lane = block: decode the block origin once, one int4 per set bit).
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(kThreads) void rank_totals_kernel(const unsigned *__restrict__ cnt, long long nunits,
                                                               unsigned *__restrict__ gtot, unsigned *__restrict__ ctot) {
    __shared__ unsigned part[16];
    const int wave = threadIdx.x >> 6, lane = fnp_lane();
    const long long c = blockIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // wave w sums groups 4w .. 4w+3 of the chunk
        const long long grp = c * 16 + wave * 4 + i, U = grp * 64 + lane;
        const unsigned t = wave_sum(U < nunits ? cnt[U] : 0u);
        if (lane == 0) {
            part[wave * 4 + i] = t;
            if (grp * 64 < nunits) gtot[grp] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i];
        ctot[c] = t;
    }
}

template <int PASS, int WPW>
__global__ __launch_bounds__(kThreads) void summary_pass_kernel(RG g, long long nunits, unsigned *__restrict__ cnt,
                                                                const unsigned *__restrict__ gtot,
                                                                const unsigned *__restrict__ ctot, int nchunks,
                                                                int *__restrict__ total, int cap_out,
                                                                int *__restrict__ out_coords) {
    constexpr int ROUND = WPW < 4 ? WPW : 4;   // non-zero summary words visited per round
    const int lane = fnp_lane();
    const long long U = ((long long)fnp_xcd_block() * kThreads + threadIdx.x) >> 6;
    if (U >= nunits) return;
    if (PASS == 1 && U == 0) {
        unsigned t = 0;
        for (int c = lane; c < nchunks; c += 64) t += ctot[c];
        t = wave_sum(t);
        if (lane == 0) *total = (int)t;
    }
    const long long S0 = U * WPW;
    unsigned long long swl, todo;
    if (WPW == 1) {
        swl = g.summ[S0];          // wave-uniform
        todo = swl != 0ull ? 1ull : 0ull;
    } else {
        swl = (lane < WPW && S0 + lane < g.nsum) ? g.summ[S0 + lane] : 0ull;
        todo = __ballot(swl != 0ull);
    }
    if (todo == 0ull) {
        if (PASS == 0 && lane == 0) cnt[U] = 0u;
        return;
    }
    unsigned run = 0;   // PASS 0: per-lane sum; PASS 1: cells before the current word (same in every lane)
    if (PASS == 1) {
        const long long grp = U >> 6, chunk = U >> 10;
        unsigned p = 0;
        for (int i = lane; i < (int)chunk; i += 64) p += ctot[i];
        if (lane < (int)(grp - chunk * 16)) p += gtot[chunk * 16 + lane];
        if (lane < (int)(U - grp * 64)) p += cnt[grp * 64 + lane];
        run = wave_sum(p);
    }
    while (todo) {
        int j[ROUND];
        bool v[ROUND];
        unsigned long long bits[ROUND];
#pragma unroll
        for (int u = 0; u < ROUND; ++u) {
            v[u] = todo != 0ull;
            j[u] = v[u] ? __builtin_ctzll(todo) : 0;
            if (v[u]) todo &= todo - 1ull;
            const unsigned long long sw = WPW == 1 ? swl : readlane64(swl, j[u]);
            const bool occ = v[u] && ((sw >> lane) & 1ull);
            bits[u] = occ ? g.bits[(S0 + j[u]) * 64 + lane] : 0ull;   // (the loads of a round are issued together)
        }
#pragma unroll
        for (int u = 0; u < ROUND; ++u) {
            const unsigned c = (unsigned)__popcll(bits[u]);
            if (PASS == 0) {
                run += c;
            } else {
                if (!v[u]) break;   // uniform
                const unsigned inc = wave_inclusive(c);
                const long long blk = (S0 + j[u]) * 64 + lane;
                int r = (int)(run + inc - c);
                if (bits[u]) {
                    g.base[blk] = (unsigned)r;
                    if (out_coords) {
                        int b, z0, y0, x0;
                        rg_decode(g.d, blk, 0, b, z0, y0, x0);
                        unsigned long long m = bits[u];
                        while (m) {
                            const int bit = __ffsll((long long)m) - 1;
                            m &= m - 1;
                            if (r < cap_out)
                                reinterpret_cast<int4 *>(out_coords)[r] = make_int4(b, z0 | (bit >> 4), y0 | ((bit >> 2) & 3), x0 | (bit & 3));
                            ++r;
                        }
                    }
                }
                if (WPW > 1) run += (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
            }
        }
    }
    if (PASS == 0) {
        run = wave_sum(run);
        if (lane == 0) cnt[U] = run;
    }
}

}  // namespace

namespace fnp_scan {
long long workspace_bytes(long long n) { return ((n + kTile - 1) / kTile + 1) * 4 + 64; }
int int32(const int *in, long long n, int *out, int *total, void *ws, hipStream_t s) {
    return run_scan(LoadInt{in}, n, out, total, ws, s);
}
// (sized for one unit per summary word, the finest split)
long long rank_grid_workspace_bytes(long long nsum) {
    const long long ngroups = (nsum + 63) >> 6, nchunks = (nsum + 1023) >> 10;
    return ((nsum * 4 + 255) & ~255ll) + (((ngroups + nchunks) * 4 + 255) & ~255ll) + 256;
}
template <int WPW>
static int rank_grid_w(const RG &g, int *total, void *ws, hipStream_t s, int *out_coords, int cap_out, bool counted) {
    const long long nunits = (g.nsum + WPW - 1) / WPW;
    const long long ngroups = (nunits + 63) >> 6, nchunks = (nunits + 1023) >> 10;
    if (nchunks > 0x7fffffffll) return FNP_ERR_ARG;
    unsigned *cnt = (unsigned *)ws;   // (nunits) cells per unit
    unsigned *gtot = (unsigned *)((char *)ws + ((nunits * 4 + 255) & ~255ll));
    unsigned *ctot = gtot + ngroups;
    const int grid = fnp_divup(nunits * 64, kThreads);
    if (counted) {   // the marking kernels counted (rankgrid.h): cnt | gtot | ctot are the grid's own, the prefix pass is all there is
        if (!g.ctr || g.wpw != WPW || g.nunits != nunits) return FNP_ERR_ARG;
        if (g.ctr_levels < 3) {   // units counted by the marks, groups and chunks by the totals kernel (into the grid's own counter words)
            hipLaunchKernelGGL(rank_totals_kernel, dim3((unsigned)nchunks), dim3(kThreads), 0, s, (const unsigned *)g.ctr, nunits, g.ctr + nunits,
                               g.ctr + nunits + ngroups);
            FNP_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(summary_pass_kernel<1, WPW>), dim3(grid), dim3(kThreads), 0, s, g, nunits, g.ctr,
                           (const unsigned *)(g.ctr + nunits), (const unsigned *)(g.ctr + nunits + ngroups), (int)nchunks, total, cap_out, out_coords);
        FNP_LAUNCH_CHECK();
        return FNP_OK;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(summary_pass_kernel<0, WPW>), dim3(grid), dim3(kThreads), 0, s, g, nunits, cnt,
                       (const unsigned *)nullptr, (const unsigned *)nullptr, (int)nchunks, total, 0, (int *)nullptr);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(rank_totals_kernel, dim3((unsigned)nchunks), dim3(kThreads), 0, s, (const unsigned *)cnt, nunits, gtot, ctot);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(HIP_KERNEL_NAME(summary_pass_kernel<1, WPW>), dim3(grid), dim3(kThreads), 0, s, g, nunits, cnt,
                       (const unsigned *)gtot, (const unsigned *)ctot, (int)nchunks, total, cap_out, out_coords);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
int rank_grid(const RG &g, int *total, void *ws, hipStream_t s, int *out_coords, int cap_out, bool counted) {
    // the split only changes who computes what, never the ranks (fnp_rg_wpw: the marking kernels count per the same units)
    const int wpw = g.wpw;
    if (wpw == 64) return rank_grid_w<64>(g, total, ws, s, out_coords, cap_out, counted);
    if (wpw == 16) return rank_grid_w<16>(g, total, ws, s, out_coords, cap_out, counted);
    if (wpw == 8) return rank_grid_w<8>(g, total, ws, s, out_coords, cap_out, counted);
    return rank_grid_w<1>(g, total, ws, s, out_coords, cap_out, counted);
}
}  // namespace fnp_scan

extern "C" int64_t fnp_rankgrid_counter_words(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const long long nblk = fnp_num_blocks(fnp_make_dims(B, D, H, W));
    return fnp_rg_counter_words((nblk + 63) >> 6) + 64;
}

extern "C" int64_t fnp_rankgrid_workspace_bytes(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const long long nblk = fnp_num_blocks(fnp_make_dims(B, D, H, W));
    return fnp_scan::rank_grid_workspace_bytes((nblk + 63) >> 6) + 256;
}
