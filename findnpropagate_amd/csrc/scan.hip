// Device-wide exclusive scans (wave shuffles + LDS) used for order-preserving compaction:
// first-point flags of the voxeliser (int32 scan: tile reduce -> scan of tile sums -> tile scan)
// and the rank-grid popcount prefix, which walks the summary level so that only occupied
// blocks are read.
#include "rankgrid.cuh"

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

template <class L, class TOut>
int run_scan(L load, long long n, TOut *out, int *total, void *ws, hipStream_t s) {
    if (n < 0 || !total) return FNP_ERR_ARG;
    if (n == 0) {
        FNP_HIP_TRY(hipMemsetAsync(total, 0, sizeof(int), s));
        return FNP_OK;
    }
    if (!out || !ws) return FNP_ERR_ARG;
    const int tiles = fnp_divup(n, fnp_scan::kTile);
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
// One wave per summary word S (64 blocks); a wave whose word is zero retires after one load, so
// the cost follows the occupied blocks.  Lane j owns block 64*S + j.  Three kernels, none of them a
// scan over the 360 k per-word counts:
//   PASS 0 : cnt[S]  = occupied cells in the 64 blocks of S
//   TOTALS : gtot[g] = sum of cnt over group g (64 words), ctot[c] = over chunk c (16 groups);
//            one workgroup per chunk, coalesced
//   PASS 1 : base[w] = cells before S + occupied cells in the blocks of S before w   (w occupied),
//            where "cells before S" = sum(ctot[< c]) + sum(gtot[16c .. g)) + sum(cnt[64g .. S)):
//            <= 350 + 15 + 63 values, read lane-parallel and reduced once per occupied word
//            *total = sum(ctot)   (wave of S = 0)
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ __launch_bounds__(kThreads) void rank_totals_kernel(const unsigned *__restrict__ cnt, long long nsum,
                                                               unsigned *__restrict__ gtot, unsigned *__restrict__ ctot) {
    __shared__ unsigned part[16];
    const int wave = threadIdx.x >> 6, lane = fnp_lane();
    const long long c = blockIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // wave w sums groups 4w .. 4w+3 of the chunk
        const long long grp = c * 16 + wave * 4 + i, S = grp * 64 + lane;
        const unsigned t = wave_sum(S < nsum ? cnt[S] : 0u);
        if (lane == 0) {
            part[wave * 4 + i] = t;
            if (grp * 64 < nsum) gtot[grp] = t;
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

template <int PASS>
__global__ __launch_bounds__(kThreads) void summary_pass_kernel(RG g, unsigned *__restrict__ cnt, const unsigned *__restrict__ gtot,
                                                                const unsigned *__restrict__ ctot, int nchunks,
                                                                int *__restrict__ total) {
    const int lane = fnp_lane();
    const long long S = ((long long)blockIdx.x * kThreads + threadIdx.x) >> 6;
    if (S >= g.nsum) return;
    if (PASS == 1 && S == 0) {
        unsigned t = 0;
        for (int c = lane; c < nchunks; c += 64) t += ctot[c];
        t = wave_sum(t);
        if (lane == 0) *total = (int)t;
    }
    const unsigned long long sw = g.summ[S];   // wave-uniform
    if (sw == 0ull) {
        if (PASS == 0 && lane == 0) cnt[S] = 0;
        return;
    }
    const long long blk = S * 64 + lane;
    const bool occ = (sw >> lane) & 1ull;
    const unsigned c = occ ? (unsigned)__popcll(g.bits[blk]) : 0u;
    const unsigned inc = wave_inclusive(c);
    if (PASS == 0) {
        if (lane == 63) cnt[S] = inc;
    } else {
        const long long grp = S >> 6, chunk = S >> 10;
        unsigned p = 0;
        for (int i = lane; i < (int)chunk; i += 64) p += ctot[i];
        if (lane < (int)(grp - chunk * 16)) p += gtot[chunk * 16 + lane];
        if (lane < (int)(S - grp * 64)) p += cnt[grp * 64 + lane];
        p = wave_sum(p);
        if (occ) g.base[blk] = p + inc - c;
    }
}

}  // namespace

namespace fnp_scan {
long long workspace_bytes(long long n) { return ((n + kTile - 1) / kTile + 1) * 4 + 64; }
int int32(const int *in, long long n, int *out, int *total, void *ws, hipStream_t s) {
    return run_scan(LoadInt{in}, n, out, total, ws, s);
}
long long rank_grid_workspace_bytes(long long nsum) {
    const long long ngroups = (nsum + 63) >> 6, nchunks = (nsum + 1023) >> 10;
    return ((nsum * 4 + 255) & ~255ll) + (((ngroups + nchunks) * 4 + 255) & ~255ll) + 256;
}
int rank_grid(const RG &g, int *total, void *ws, hipStream_t s) {
    const long long ngroups = (g.nsum + 63) >> 6, nchunks = (g.nsum + 1023) >> 10;
    if (nchunks > 0x7fffffffll) return FNP_ERR_ARG;
    unsigned *cnt = (unsigned *)ws;   // (nsum) cells per summary word
    unsigned *gtot = (unsigned *)((char *)ws + ((g.nsum * 4 + 255) & ~255ll));
    unsigned *ctot = gtot + ngroups;
    const int grid = fnp_divup(g.nsum * 64, kThreads);
    hipLaunchKernelGGL(summary_pass_kernel<0>, dim3(grid), dim3(kThreads), 0, s, g, cnt, (const unsigned *)nullptr,
                       (const unsigned *)nullptr, (int)nchunks, total);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(rank_totals_kernel, dim3((unsigned)nchunks), dim3(kThreads), 0, s, (const unsigned *)cnt, g.nsum, gtot, ctot);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(summary_pass_kernel<1>, dim3(grid), dim3(kThreads), 0, s, g, cnt, (const unsigned *)gtot,
                       (const unsigned *)ctot, (int)nchunks, total);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
}  // namespace fnp_scan

extern "C" int64_t fnp_rankgrid_workspace_bytes(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const long long nblk = fnp_num_blocks(fnp_make_dims(B, D, H, W));
    return fnp_scan::rank_grid_workspace_bytes((nblk + 63) >> 6) + 256;
}
