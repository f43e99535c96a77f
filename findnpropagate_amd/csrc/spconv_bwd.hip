// Backward of the sparse convolution (spconv's autograd behind SubMConv3d /
// SparseConv3d in the self-training step, tools/train_st.py; call sites spconv_backbone.py:12-17,39-46).
//
//   forward   out[o] = sum_k W_k^T x[nbr[k][o]]                        W_k: (Cout, Cin)
//   dgrad     dx[i]  = sum_k W_k  dy[nbrT[k][i]]    nbrT[k][i] = o  <=>  nbr[k][o] = i
//             = the forward kernel on the TRANSPOSED rulebook with the (Cin, Cout) transposed slabs, so
//               it runs on the same MFMA / VALU implicit-GEMM kernels (fnp_spconv_forward); only the
//               transposition of the rulebook is new (one scatter, each (k, i) has at most one o)
//   wgrad     dW_k   = sum_o dy[o] (x) x[nbr[k][o]]                     (Cout, Cin) per offset
//             two stages, no atomics: every (row chunk, offset) workgroup accumulates its Cout x Cin
//             block in registers from LDS-staged row tiles (f32), writes a partial, and a second kernel
//             adds the partials in chunk order -> bit-reproducible gradients.
#include "common.h"
#include <type_traits>

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void transpose_rulebook_kernel(const int *__restrict__ nbr, int nbr_stride, int K,
                                                                      const int *__restrict__ n_out, int cap_out,
                                                                      int *__restrict__ nbr_t, int cap_in) {
    const int n = min(*n_out, cap_out);
    const int k = blockIdx.y;
    for (int o = blockIdx.x * kThreads + threadIdx.x; o < n; o += gridDim.x * kThreads) {
        const int i = nbr[(size_t)k * nbr_stride + o];
        if (i >= 0 && i < cap_in) nbr_t[(size_t)k * cap_in + i] = o;
    }
}

__device__ __forceinline__ float ld_f32(const float *p) { return *p; }
__device__ __forceinline__ float ld_f32(const __bf16 *p) { return (float)*p; }
__device__ __forceinline__ float ld_f32(const _Float16 *p) { return (float)*p; }

// grid (chunks, K).  TR rows per tile; LDS: dy tile (TR x Cout) + x tile (TR x Cin), f32.
// Thread t owns the pairs p = t + 256 j (co = p / Cin, ci = p % Cin), at most PMAX of them.
template <typename TX, typename TY, int PMAX>
__global__ __launch_bounds__(kThreads) void wgrad_partial_kernel(const TX *__restrict__ x, const TY *__restrict__ dy,
                                                                 const int *__restrict__ nbr, int nbr_stride,
                                                                 const int *__restrict__ n_out, int cap_out, int /*unused*/,
                                                                 int Cin, int Cout, float *__restrict__ partial) {
    constexpr int TR = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char fnp_wg_smem[];
    float *sy = reinterpret_cast<float *>(fnp_wg_smem);   // [TR][Cout]
    float *sx = sy + TR * Cout;                           // [TR][Cin]
    __shared__ int sidx[TR];
    const int n = min(*n_out, cap_out);
    const int k = blockIdx.y, chunk = blockIdx.x, K = gridDim.y;
    // the row count is only known on the device: cut the n rows (not the capacity) into gridDim.x chunks
    const int rows_per_chunk = ((n + (int)gridDim.x - 1) / (int)gridDim.x + 127) / 128 * 128;
    const int r0 = min(n, chunk * rows_per_chunk), r1 = min(n, r0 + rows_per_chunk);
    const int pairs = Cin * Cout;
    float acc[PMAX];
#pragma unroll
    for (int j = 0; j < PMAX; ++j) acc[j] = 0.f;
    for (int t0 = r0; t0 < r1; t0 += TR) {
        __syncthreads();
        if (threadIdx.x < TR) {
            const int r = t0 + threadIdx.x;
            sidx[threadIdx.x] = r < r1 ? nbr[(size_t)k * nbr_stride + r] : -1;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TR * Cout; e += kThreads) {
            const int rr = e / Cout, c = e % Cout, r = t0 + rr;
            sy[e] = (r < r1 && sidx[rr] >= 0) ? ld_f32(dy + (size_t)r * Cout + c) : 0.f;
        }
        for (int e = threadIdx.x; e < TR * Cin; e += kThreads) {
            const int rr = e / Cin, c = e % Cin, id = sidx[rr];
            sx[e] = id >= 0 ? ld_f32(x + (size_t)id * Cin + c) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PMAX; ++j) {
            const int p = threadIdx.x + j * kThreads;
            if (p < pairs) {
                const int co = p / Cin, ci = p % Cin;
                float a = acc[j];
#pragma unroll
                for (int rr = 0; rr < TR; ++rr) a = fmaf(sy[rr * Cout + co], sx[rr * Cin + ci], a);
                acc[j] = a;
            }
        }
    }
    float *out = partial + ((size_t)chunk * K + k) * pairs;
#pragma unroll
    for (int j = 0; j < PMAX; ++j) {
        const int p = threadIdx.x + j * kThreads;
        if (p < pairs) out[p] = acc[j];
    }
}

// PAIR LISTS, balanced (round 3, late).  With one row-chunk grid per offset, (chunks, K), every offset got the same number of
// workgroups — but the centre offset lists EVERY row and the others 1/4 to 1/30 of them (3.6-14 neighbours of 27): its
// workgroups ran 2-7 times longer than the rest, alone on a tenth of the CUs.  The W = chunks x K workgroups of a launch now
// share the pairs evenly: R = rows per workgroup = ceil(total pairs / (W - K)) rounded up to 128, offset k takes ceil(n_k / R)
// consecutive workgroups (sum <= W; the spare ones leave at once), and workgroup w writes its partial sum at partial[w].  The
// split is a function of pair_count alone, recomputed by every workgroup and by the reduction: same order every run.
// MEASURED (16 scenes): conv_input 137 -> 63 us, 16 x 16 32.7 -> 25.6 us — and the wide layers SLOWER (128 x 128 139 -> 172 us,
// 64 x 64 117 -> 166, 32 x 32 124 -> 172, 32 x 64 49 -> 76): those are bound by the L2 gather rate, not by their longest
// workgroup (the centre offset's workgroups ran alone at the end, but at the full rate of an idle L2), and ~580 equal
// workgroups at two per CU make two rounds.  Balanced for Cin x Cout <= 256 only (kPairBalancedMax).
constexpr int kPairBalancedMax = 256;
struct PairSplit {
    int k, chunk, R, n, first, count;   // offset, chunk index inside it, rows per chunk, pairs of the offset, its first workgroup and how many
    bool live;
};
// (spc: the K <= 64 pair counts, clamped, in LDS — loaded by K threads at once: a lone thread walking pair_count in memory
//  took 80 us in the reduction kernel)
__device__ __forceinline__ void pair_counts_to_lds(const int *__restrict__ pair_count, int K, int cap_out, int *spc) {
    if ((int)threadIdx.x < K) spc[threadIdx.x] = min(pair_count[threadIdx.x], cap_out);
    __syncthreads();
}
__device__ __forceinline__ int pair_rows_per_wg(const int *spc, int K, int W) {
    long long total = 0;
    for (int k = 0; k < K; ++k) total += spc[k];
    const int budget = W - K > 0 ? W - K : 1;
    const int R = (int)(((total + budget - 1) / budget + 127) / 128 * 128);
    return R < 128 ? 128 : R;
}
__device__ __forceinline__ PairSplit pair_split_of_wg(const int *spc, int K, int W, int w) {
    PairSplit ps;
    ps.R = pair_rows_per_wg(spc, K, W);
    ps.live = false;
    ps.k = 0; ps.chunk = 0; ps.n = 0; ps.first = 0; ps.count = 0;
    int start = 0;
    for (int k = 0; k < K; ++k) {
        const int n = spc[k], c = (n + ps.R - 1) / ps.R;
        if (!ps.live && w < start + c) {
            ps.k = k; ps.chunk = w - start; ps.n = n; ps.first = start; ps.count = c; ps.live = true;
        }
        start += c;
    }
    return ps;
}
// dw (K, Cout, Cin) (or the module's layout) = the partial sums of each offset's workgroups, in workgroup order (K <= 64)
__global__ __launch_bounds__(kThreads) void wgrad_reduce_pairs_kernel(const float *__restrict__ partial, const int *__restrict__ pair_count, int K,
                                                                      int cap_out, int W, int pairs, float *__restrict__ dw, int module_cc, int Cin) {
    __shared__ int sfirst[65], spc[64];
    pair_counts_to_lds(pair_count, K, cap_out, spc);
    if (threadIdx.x == 0) {   // (the split once per workgroup)
        const int R = pair_rows_per_wg(spc, K, W);
        int start = 0;
        for (int j = 0; j < K; ++j) {
            sfirst[j] = start;
            start += (spc[j] + R - 1) / R;
        }
        sfirst[K] = start;
    }
    __syncthreads();
    // eight lanes per element: lane j adds the partials j, j + 8, ... in order, a fixed butterfly adds the eight sums (the centre
    // offset owns a quarter of the workgroups: one thread adding its ~230 partials in turn took 80 us)
    const long long total = (long long)K * pairs;
    const int sub = threadIdx.x & 7;
    for (long long e0 = ((long long)blockIdx.x * kThreads + threadIdx.x) >> 3; e0 < ((total + 31) & ~31ll); e0 += ((long long)gridDim.x * kThreads) >> 3) {
        const bool live = e0 < total;   // (whole waves reach the shuffles: 8 elements per wave, total rounded up to 32)
        const long long e = live ? e0 : total - 1;
        const int k = (int)(e / pairs), p = (int)(e % pairs);
        const int first = sfirst[k], count = sfirst[k + 1] - first;
        float s = 0.f;
        for (int c = sub; c < count; c += 8) s += partial[(size_t)(first + c) * pairs + p];   // fixed order
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (live && sub == 0) {
            long long o = e;
            if (module_cc) {
                const int co = p / Cin, ci = p % Cin;
                o = ((long long)co * K + k) * Cin + ci;
            }
            dw[o] = s;
        }
    }
}

// The same sum for layers with few (Cin x Cout <= 128) pairs — conv_input, 5 -> 16 — where the kernel above leaves two thirds
// of its threads idle and meets three barriers per 16 rows: 128-row tiles, and the 256 threads form G = 256 / pairs groups
// that each take every G-th row of a tile; the groups' sums are added in group order at the end (fixed order).
// PAIRS (round 3): the rows of offset k come from the rulebook's pair lists (nbr = pair_o, pair_i, pair_count[k] of them) instead
// of a sweep over every row of the table: conv_input's rows have 3.6 of 27 neighbours, so seven tiles of eight staged zeros only.
template <typename TX, typename TY, bool PAIRS>
__global__ __launch_bounds__(kThreads) void wgrad_small_kernel(const TX *__restrict__ x, const TY *__restrict__ dy,
                                                               const int *__restrict__ nbr, int nbr_stride,
                                                               const int *__restrict__ n_out, int cap_out,
                                                               int Cin, int Cout, float *__restrict__ partial,
                                                               const int *__restrict__ pair_i, const int *__restrict__ pair_count) {
    constexpr int TR = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char fnp_wg_smem[];
    float *sy = reinterpret_cast<float *>(fnp_wg_smem);   // [TR][Cout]
    float *sx = sy + TR * Cout;                           // [TR][Cin]
    __shared__ int sidx[TR];
    __shared__ int sout[TR];
    __shared__ float sred[kThreads];
    int k = blockIdx.y, chunk = blockIdx.x;
    const int K = gridDim.y;
    int n, rows_per_chunk;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const bool bal = PAIRS && K <= 64;   // (uniform; the reduction's table of first workgroups holds 64 offsets)
    __shared__ int spc[64];
    if (bal) {   // (entries of offset k, shared evenly over the launch's workgroups)
        pair_counts_to_lds(pair_count, K, cap_out, spc);
        const PairSplit ps = pair_split_of_wg(spc, K, gridDim.x * gridDim.y, wg);
        if (!ps.live) return;   // (whole workgroup, before any barrier)
        k = ps.k; chunk = ps.chunk; n = ps.n; rows_per_chunk = ps.R;
    } else {
        n = PAIRS ? min(pair_count[k], cap_out) : min(*n_out, cap_out);
        rows_per_chunk = ((n + (int)gridDim.x - 1) / (int)gridDim.x + 127) / 128 * 128;
    }
    const int r0 = min(n, chunk * rows_per_chunk), r1 = min(n, r0 + rows_per_chunk);
    const int pairs = Cin * Cout, G = kThreads / pairs;
    const int g = threadIdx.x / pairs, p = threadIdx.x % pairs, co = p / Cin, ci = p % Cin;
    float acc = 0.f;
    for (int t0 = r0; t0 < r1; t0 += TR) {
        __syncthreads();
        if (threadIdx.x < TR) {
            const int r = t0 + threadIdx.x;
            if (PAIRS) {
                sidx[threadIdx.x] = r < r1 ? pair_i[(size_t)k * nbr_stride + r] : -1;
                sout[threadIdx.x] = r < r1 ? nbr[(size_t)k * nbr_stride + r] : 0;
            } else {
                sidx[threadIdx.x] = r < r1 ? nbr[(size_t)k * nbr_stride + r] : -1;
                sout[threadIdx.x] = r;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TR * Cout; e += kThreads) {
            const int rr = e / Cout, c = e % Cout, r = t0 + rr;
            sy[e] = (r < r1 && sidx[rr] >= 0) ? ld_f32(dy + (size_t)sout[rr] * Cout + c) : 0.f;
        }
        for (int e = threadIdx.x; e < TR * Cin; e += kThreads) {
            const int rr = e / Cin, c = e % Cin, id = sidx[rr];
            sx[e] = id >= 0 ? ld_f32(x + (size_t)id * Cin + c) : 0.f;
        }
        __syncthreads();
        if (g < G) {
            for (int rr = g; rr < TR; rr += G) acc = fmaf(sy[rr * Cout + co], sx[rr * Cin + ci], acc);
        }
    }
    __syncthreads();
    sred[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < pairs) {
        float a = 0.f;
        for (int gg = 0; gg < G; ++gg) a += sred[gg * pairs + threadIdx.x];
        partial[(bal ? (size_t)wg : (size_t)chunk * K + k) * pairs + threadIdx.x] = a;
    }
}

// module_cc = Cout * Cin when dw is the MODULE's layout (Cout, K, Cin) (the parameter's .grad, written here instead of by a
// permute-copy afterwards), 0 for the packed (K, Cout, Cin)
__global__ __launch_bounds__(kThreads) void wgrad_reduce_kernel(const float *__restrict__ partial, int chunks, long long total,
                                                                float *__restrict__ dw, int module_cc = 0, int Cin = 1, int K = 1) {
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kThreads) {
        float s = 0.f;
        for (int c = 0; c < chunks; ++c) s += partial[(size_t)c * total + e];   // fixed order
        long long o = e;
        if (module_cc) {
            const int k = (int)(e / module_cc), r = (int)(e % module_cc), co = r / Cin, ci = r % Cin;
            o = ((long long)co * K + k) * Cin + ci;
        }
        dw[o] = s;
    }
}

// ---- MFMA weight gradient (bf16 x bf16 -> f32) ---------------------------------------------------
// dW_k (COUT x CIN) = dY^T (COUT x rows) . X_k (rows x CIN): rows are the GEMM's K dimension, so both
// MFMA operands are columns of row-major tensors.  A 32-row tile of dy and of the gathered x rows is staged
// ROW-major in LDS with 16-byte stores (rows padded by 16 bytes) and the fragments are read with gfx950's
// transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block and each lane
// receives one column of it): lane (l15, q) of an operand needs rows 8q .. 8q+7 of column c0 + l15, i.e. two
// such reads.  (The first version scattered every 16-byte chunk to a [channel][row] image with eight 2-byte
// LDS writes, 16-way bank-conflicted: 1.03 ms for the 128 x 128 layer against 0.12 ms for its forward.)
// Workgroup = 4 waves = WB waves across the COUT/16 row blocks of dW x WK = 4/WB row slices of the tile
// (k-split); tiles are double-buffered, one barrier per tile; the WK partial sums meet in LDS at the end.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename T> struct Frag8 { typedef T type __attribute__((ext_vector_type(8))); };
__device__ __forceinline__ f32x4_t wg_mfma(Frag8<__bf16>::type a, Frag8<__bf16>::type b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t wg_mfma(Frag8<_Float16>::type a, Frag8<_Float16>::type b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// T16: __bf16 or _Float16 (the reference's AMP mode): the images are moved as raw 16-bit elements, only the matrix
// instruction differs
// PAIR LISTS (round 3).  The weight gradient of offset k only sums over the output rows that HAVE a neighbour at k — 44 %
// (stage 2) to 54 % (stage 4) of the rows — but a sweep over all rows of the table stages a 32-row tile of dy and of (mostly
// zero) x rows for every (tile, offset) and multiplies the zeros.  fnp_rulebook_pairs compacts every offset's column once
// per rulebook (its two to four convolutions share it): pair_o[k][j], pair_i[k][j] = the j-th output row with a neighbour
// at k, in ascending order, and that neighbour; count[k].  Three small launches, order-preserving (a count per 1,024-row tile,
// a scan per offset, an emit pass), so the gradient stays bit-reproducible.
constexpr int kPairTile = 1024;
__global__ __launch_bounds__(kThreads) void pairs_count_kernel(const int *__restrict__ nbr, int nbr_stride, const int *__restrict__ n_out, int cap,
                                                               int ntiles, int *__restrict__ counts) {
    __shared__ int wsum[kThreads / 64];
    const int n = min(*n_out, cap), k = blockIdx.y, t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int c = 0;
#pragma unroll
    for (int j = 0; j < kPairTile / kThreads; ++j) {
        const int o = t * kPairTile + j * kThreads + threadIdx.x;
        c += (o < n && nbr[(size_t)k * nbr_stride + o] >= 0) ? 1 : 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < kThreads / 64; ++w) tot += wsum[w];
        counts[(size_t)k * ntiles + t] = tot;
    }
}
// grid (K): exclusive scan of counts[k][0 .. ntiles) in place, total -> pair_count[k]
__global__ __launch_bounds__(kThreads) void pairs_scan_kernel(int ntiles, int *__restrict__ counts, int *__restrict__ pair_count) {
    __shared__ int wsum[kThreads / 64];
    __shared__ int carry_s;
    int *row = counts + (size_t)blockIdx.x * ntiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b = 0; b < ntiles; b += kThreads) {
        const int i = b + threadIdx.x;
        const int v = i < ntiles ? row[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int off = carry_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (i < ntiles) row[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == kThreads - 1) carry_s = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) pair_count[blockIdx.x] = carry_s;
}
__global__ __launch_bounds__(kThreads) void pairs_emit_kernel(const int *__restrict__ nbr, int nbr_stride, const int *__restrict__ n_out, int cap,
                                                              int ntiles, const int *__restrict__ base, int *__restrict__ pair_o,
                                                              int *__restrict__ pair_i, int pair_stride) {
    __shared__ int wsum[kThreads / 64];
    const int n = min(*n_out, cap), k = blockIdx.y, t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pos = base[(size_t)k * ntiles + t];
    for (int j = 0; j < kPairTile / kThreads; ++j) {   // (ascending rows: pass j covers rows t*1024 + j*256 ..)
        const int o = t * kPairTile + j * kThreads + threadIdx.x;
        const int id = o < n ? nbr[(size_t)k * nbr_stride + o] : -1;
        const unsigned long long m = __ballot(id >= 0);
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = pos, tot = 0;
        for (int w = 0; w < kThreads / 64; ++w) {
            if (w < wave) off += wsum[w];
            tot += wsum[w];
        }
        if (id >= 0) {
            const int p = off + __popcll(m & ((1ull << lane) - 1ull));
            pair_o[(size_t)k * pair_stride + p] = o;
            pair_i[(size_t)k * pair_stride + p] = id;
        }
        pos += tot;
        __syncthreads();
    }
}

// PAIRS: the rows of offset k come from the pair lists (pair_o / pair_i / pair_count) instead of a sweep over the table column
#ifndef FNP_WGRAD_TR_NARROW
#define FNP_WGRAD_TR_NARROW 128
#endif
#ifndef FNP_WGRAD_TR_WIDE
#define FNP_WGRAD_TR_WIDE 64   // (128 x 128; 32 -> 64 rows per tile at the end of round 3: twice the MFMAs between two barriers, still two
                                //  workgroups per CU: 142 -> 122 us per launch at 16 scenes — the kernel is bound by its per-tile round trip)
#endif
template <int CIN, int COUT> struct WgradTile {
    static constexpr int NBO = COUT / 16, WB = NBO >= 4 ? 4 : NBO, WK = 4 / WB;
    // (two buffers x WK slices x rows x (dy row + x row + padding) must leave room for several workgroups per CU)
    static constexpr int rows = (CIN + COUT >= 256) ? FNP_WGRAD_TR_WIDE : (CIN + COUT >= 192 || WK > 1) ? 64 : FNP_WGRAD_TR_NARROW;
};
template <int CIN, int COUT, typename T16, bool PAIRS = false>
__global__ __launch_bounds__(kThreads) void wgrad_mfma_kernel(const T16 *__restrict__ x, const T16 *__restrict__ dy,
                                                              const int *__restrict__ nbr, int nbr_stride,
                                                              const int *__restrict__ n_out, int cap_out, int balanced,
                                                              float *__restrict__ partial, const int *__restrict__ pair_i = nullptr,
                                                              const int *__restrict__ pair_count = nullptr) {
    constexpr int NBO = COUT / 16, NBI = CIN / 16;
    constexpr int WB = NBO >= 4 ? 4 : NBO;          // waves across output-channel blocks
    constexpr int WK = 4 / WB;                      // row slices (k-split)
    constexpr int OB = NBO / WB;                    // output-channel blocks per wave
    // rows per slice and tile: 64 where a 32-row step is 16 MFMAs per wave (128 x 128; 32 until the end of round 3); the narrow layers do
    // 2-8 MFMAs per 32 rows and were bound by the tile's barrier + LDS round trip, not by rows: they take 128 / 64 rows per
    // tile (4 / 2 MFMA steps between two barriers)
    constexpr int TR = WgradTile<CIN, COUT>::rows;
    constexpr int GT = WB * 64;                     // threads of one k-split group
    constexpr int XCH = CIN / 8, YCH = COUT / 8;    // 16-byte chunks per row
    constexpr int SY = COUT * 2 + 16, SX = CIN * 2 + 16;   // bytes per staged row (dy / x)
    constexpr int SLICE = (SY + SX) * TR / 2;       // 16-bit elements of one slice image: dy rows then x rows
    using bf16x8_t = typename Frag8<T16>::type;
    static_assert(NBO % WB == 0 && CIN % 16 == 0 && COUT % 16 == 0, "channel counts");
    extern __shared__ __attribute__((aligned(16))) unsigned char fnp_wg_smem[];
    __bf16 *lds = reinterpret_cast<__bf16 *>(fnp_wg_smem);     // [2 buffers][WK slices][SLICE]

    int k = blockIdx.y, chunk = blockIdx.x;
    const int K = gridDim.y;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    // (PAIRS: `nbr` is pair_o, n the number of pairs of this offset, and the launch's workgroups share ALL pairs evenly)
    int n, rows_per_chunk;
    const bool bal = PAIRS && balanced;   // (uniform)
    __shared__ int spc[64];
    if (bal) {
        pair_counts_to_lds(pair_count, K, cap_out, spc);
        const PairSplit ps = pair_split_of_wg(spc, K, gridDim.x * gridDim.y, wg);
        if (!ps.live) return;   // (whole workgroup, before any barrier)
        k = ps.k; chunk = ps.chunk; n = ps.n; rows_per_chunk = ps.R;
    } else {
        n = PAIRS ? min(pair_count[k], cap_out) : min(*n_out, cap_out);
        rows_per_chunk = ((n + (int)gridDim.x - 1) / (int)gridDim.x + 127) / 128 * 128;   // from n, not the capacity
    }
    const int r0 = min(n, chunk * rows_per_chunk), r1 = min(n, r0 + rows_per_chunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = wave / WB, wb = wave % WB;       // this wave's row slice and block column
    const int gtid = tid - ks * GT;                 // thread index inside the k-split group
    const int l15 = lane & 15, kq = lane >> 4;

    f32x4_t acc[OB][NBI];
#pragma unroll
    for (int a = 0; a < OB; ++a)
#pragma unroll
        for (int b = 0; b < NBI; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    constexpr int YL = (TR * YCH + GT - 1) / GT, XL = (TR * XCH + GT - 1) / GT;   // 16-byte loads per thread and tile
    uint4 ry[YL], rx[XL];
    // The row indices run a tile ahead of the rows (round 3): a tile's time was two dependent memory round trips — index, then
    // row — against ~0.3 us of matrix work, with the rows requested one tile ahead only (135 tiles x 1.9 us for the 128 x 128
    // layer at 16 scenes).  fetch_idx(t) loads what fetch(t) dereferences; the loop keeps it one tile further ahead.
    int iy[YL], ix[XL];
    // tile t of this slice covers rows r0 + (t*WK + ks)*TR .. + TR
    auto fetch_idx = [&](int t) {
        const int base = r0 + (t * WK + ks) * TR;
#pragma unroll
        for (int j = 0; j < YL; ++j) {
            const int e = gtid + j * GT, rr = e / YCH, r = base + rr;
            iy[j] = -1;
            if (e < TR * YCH && r < r1) iy[j] = PAIRS ? nbr[(size_t)k * nbr_stride + r] : r;   // (PAIRS: the r-th output row with a neighbour at k)
        }
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int e = gtid + j * GT, rr = e / XCH, r = base + rr;
            ix[j] = -1;
            if (e < TR * XCH && r < r1) ix[j] = PAIRS ? pair_i[(size_t)k * nbr_stride + r] : nbr[(size_t)k * nbr_stride + r];
        }
    };
    auto fetch = [&]() {   // the rows behind iy / ix
#pragma unroll
        for (int j = 0; j < YL; ++j) {
            const int c = (gtid + j * GT) % YCH;
            ry[j] = make_uint4(0u, 0u, 0u, 0u);
            if (iy[j] >= 0) ry[j] = *reinterpret_cast<const uint4 *>(dy + (size_t)iy[j] * COUT + c * 8);
        }
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int c = (gtid + j * GT) % XCH;
            rx[j] = make_uint4(0u, 0u, 0u, 0u);
            if (ix[j] >= 0) rx[j] = *reinterpret_cast<const uint4 *>(x + (size_t)ix[j] * CIN + c * 8);
        }
    };
    auto stage = [&](int buf) {
        unsigned char *img = reinterpret_cast<unsigned char *>(lds + (size_t)(buf * WK + ks) * SLICE);
#pragma unroll
        for (int j = 0; j < YL; ++j) {
            const int e = gtid + j * GT, rr = e / YCH, c = e % YCH;
            if (e < TR * YCH) *reinterpret_cast<uint4 *>(img + rr * SY + c * 16) = ry[j];
        }
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int e = gtid + j * GT, rr = e / XCH, c = e % XCH;
            if (e < TR * XCH) *reinterpret_cast<uint4 *>(img + TR * SY + rr * SX + c * 16) = rx[j];
        }
    };
    // operand fragment of column c0 + l15, rows 8 kq .. 8 kq + 7, from a row-major image with row stride S:
    // lane 4 q' + p of a 16-lane group supplies the address of (row r0 + q', columns c0 + 4 p ..) and
    // receives column c0 + (its index in the group) of the four rows (every lane takes part: EXEC is full here)
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    const int trq = l15 >> 2, trp = l15 & 3;
    auto tr_frag = [&](const unsigned char *img, int S, int c0) -> bf16x8_t {
        const unsigned char *a0 = img + (8 * kq + trq) * S + (c0 + 4 * trp) * 2;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(a0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(a0 + 4 * S));
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return *reinterpret_cast<const bf16x8_t *>(&v);
    };

    const int tiles = (r1 - r0 + WK * TR - 1) / (WK * TR);   // (workgroup-uniform)
    if (tiles > 0) {
        fetch_idx(0);
        fetch();
        stage(0);
        fetch_idx(1);   // (past the last tile: every index -1)
    }
    __syncthreads();
    for (int t = 0; t < tiles; ++t) {
        if (t + 1 < tiles) {
            fetch();            // rows of tile t + 1 (its indices arrived a tile ago): in flight under this tile's MFMAs
            fetch_idx(t + 2);   // indices of tile t + 2
        }
        const unsigned char *img = reinterpret_cast<const unsigned char *>(lds + (size_t)((t & 1) * WK + ks) * SLICE);
#pragma unroll
        for (int sub = 0; sub < TR / 32; ++sub) {   // 32 rows = one MFMA step
            bf16x8_t bfr[NBI];
#pragma unroll
            for (int b = 0; b < NBI; ++b) bfr[b] = tr_frag(img + TR * SY + sub * 32 * SX, SX, b * 16);
#pragma unroll
            for (int a = 0; a < OB; ++a) {
                const bf16x8_t afr = tr_frag(img + sub * 32 * SY, SY, (wb * OB + a) * 16);
#pragma unroll
                for (int b = 0; b < NBI; ++b) acc[a][b] = wg_mfma(afr, bfr[b], acc[a][b]);
            }
        }
        if (t + 1 < tiles) stage((t + 1) & 1);
        __syncthreads();
    }

    // lane holds dW[co = blk*16 + kq*4 + r][ci = b*16 + l15]; the WK slices are added through LDS (slice 0 writes)
    float *red = reinterpret_cast<float *>(fnp_wg_smem);
    float *out = partial + (bal ? (size_t)wg : (size_t)chunk * K + k) * (COUT * CIN);
    for (int pass = 1; pass < WK; ++pass) {   // (WK is 1, 2 or 4)
        __syncthreads();
        if (ks == pass) {
#pragma unroll
            for (int a = 0; a < OB; ++a)
#pragma unroll
                for (int b = 0; b < NBI; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[((wb * OB + a) * 16 + kq * 4 + r) * CIN + b * 16 + l15] = acc[a][b][r];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int a = 0; a < OB; ++a)
#pragma unroll
                for (int b = 0; b < NBI; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[a][b][r] += red[((wb * OB + a) * 16 + kq * 4 + r) * CIN + b * 16 + l15];
        }
    }
    if (ks == 0) {
#pragma unroll
        for (int a = 0; a < OB; ++a)
#pragma unroll
            for (int b = 0; b < NBI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[((wb * OB + a) * 16 + kq * 4 + r) * CIN + b * 16 + l15] = acc[a][b][r];
    }
}

template <int CIN, int COUT>
constexpr size_t wgrad_mfma_lds() {
    constexpr int NBO = COUT / 16, WB = NBO >= 4 ? 4 : NBO, WK = 4 / WB;
    const size_t tiles = (size_t)2 * WK * (COUT * 2 + 16 + CIN * 2 + 16) * WgradTile<CIN, COUT>::rows, red = (size_t)COUT * CIN * 4;
    return tiles > red ? tiles : red;
}

template <int CIN, int COUT, typename T16>
void launch_wgrad_mfma(dim3 grid, hipStream_t s, const T16 *x, const T16 *dy, const int *nbr, int nbr_stride,
                       const int *n_out, int cap_out, int rows_per_chunk, float *partial, const int *pair_i = nullptr,
                       const int *pair_count = nullptr) {
    const size_t lds = wgrad_mfma_lds<CIN, COUT>();
    if (pair_i) {
        auto kern = wgrad_mfma_kernel<CIN, COUT, T16, true>;
        hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, x, dy, nbr, nbr_stride, n_out, cap_out, (CIN * COUT <= kPairBalancedMax && grid.y <= 64) ? 1 : 0, partial, pair_i,
                           pair_count);
    } else {
        auto kern = wgrad_mfma_kernel<CIN, COUT, T16, false>;
        hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, x, dy, nbr, nbr_stride, n_out, cap_out, 0, partial, pair_i, pair_count);
    }
}

// row chunks per offset (round 3: 128 / 64 / 32 -> 32 / 32 / 24 once the pair lists had halved the rows of an offset and the
// tiles had grown: fewer partials to write and to add — the reduction pass reads chunks x K x Cin x Cout floats — and
// 24 x 27 workgroups of the 128 x 128 layer are resident at once instead of a full round and a tail; -2 % of the step)
#ifndef FNP_WG_CH1
#define FNP_WG_CH1 32
#endif
#ifndef FNP_WG_CH2
#define FNP_WG_CH2 32
#endif
#ifndef FNP_WG_CH3
#define FNP_WG_CH3 24
#endif
static inline int max_chunks(int Cin, int Cout) {
    if ((long long)Cin * Cout <= 128) return 128;   // (wgrad_small_kernel, conv_input: one accumulator per thread — it needs the workgroups)
    return (long long)Cin * Cout <= 4096 ? FNP_WG_CH1 : (long long)Cin * Cout <= 8192 ? FNP_WG_CH2 : FNP_WG_CH3;
}

template <typename TX, typename TY>
int run_wgrad(const void *x, const void *dy, const int *nbr, int nbr_stride, int K, const int *n_out, int cap_out, float *dw,
              int Cin, int Cout, void *ws, int64_t ws_bytes, hipStream_t s, int layout, const int *pair_i = nullptr, const int *pair_count = nullptr) {
    const int module_cc = layout ? Cin * Cout : 0;
    const long long pairs = (long long)Cin * Cout, total = pairs * K;
    if (pairs > 64 * kThreads) return FNP_ERR_ARG;   // <= 128 x 128
    int chunks = fnp_divup(cap_out, 2048);
    if (chunks > max_chunks(Cin, Cout)) chunks = max_chunks(Cin, Cout);
    if (chunks < 1) chunks = 1;
    const int rows_per_chunk = fnp_divup(fnp_divup(cap_out, chunks), 128) * 128;   // multiple of every tile height
    if ((long long)chunks * total * 4 > ws_bytes) return FNP_ERR_WORKSPACE;
    const size_t lds = (size_t)16 * (Cin + Cout) * 4;
    const dim3 grid(chunks, K);
    float *partial = (float *)ws;
    if constexpr (sizeof(TX) == 2 && std::is_same<TX, TY>::value) {
        bool done = true;
#define FNP_WM(CI, CO)                                                                                         \
    else if (Cin == CI && Cout == CO) launch_wgrad_mfma<CI, CO, TX>(grid, s, (const TX *)x, (const TX *)dy, nbr,         \
                                                                nbr_stride, n_out, cap_out, rows_per_chunk, partial, pair_i, pair_count)
        if (false) {}
        FNP_WM(16, 16);
        FNP_WM(16, 32);
        FNP_WM(32, 32);
        FNP_WM(32, 64);
        FNP_WM(64, 64);
        FNP_WM(64, 128);
        FNP_WM(128, 128);
        else done = false;
#undef FNP_WM
        if (!done && pair_i) return FNP_ERR_ARG;   // (pair lists: the MFMA shapes only)
        if (done) {
            FNP_LAUNCH_CHECK();
            if (pair_i && pairs <= kPairBalancedMax && K <= 64)
                hipLaunchKernelGGL(wgrad_reduce_pairs_kernel, dim3(fnp_grid_for(total * 8, kThreads)), dim3(kThreads), 0, s, (const float *)partial,
                                   pair_count, K, cap_out, chunks * K, (int)pairs, dw, module_cc, Cin);
            else
                hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(fnp_grid_for(total, kThreads)), dim3(kThreads), 0, s,
                                   (const float *)partial, chunks, total, dw, module_cc, Cin, K);
            FNP_LAUNCH_CHECK();
            return FNP_OK;
        }
    }
    if (pair_i && pairs > 128) return FNP_ERR_ARG;
    if (pairs <= 128) {
        if (pair_i)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_small_kernel<TX, TY, true>), grid, dim3(kThreads), (size_t)128 * (Cin + Cout) * 4, s, (const TX *)x,
                               (const TY *)dy, nbr, nbr_stride, n_out, cap_out, Cin, Cout, partial, pair_i, pair_count);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_small_kernel<TX, TY, false>), grid, dim3(kThreads), (size_t)128 * (Cin + Cout) * 4, s, (const TX *)x,
                               (const TY *)dy, nbr, nbr_stride, n_out, cap_out, Cin, Cout, partial, pair_i, pair_count);
        FNP_LAUNCH_CHECK();
        if (pair_i && K <= 64)
            hipLaunchKernelGGL(wgrad_reduce_pairs_kernel, dim3(fnp_grid_for(total * 8, kThreads)), dim3(kThreads), 0, s, (const float *)partial,
                               pair_count, K, cap_out, chunks * K, (int)pairs, dw, module_cc, Cin);
        else
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(fnp_grid_for(total, kThreads)), dim3(kThreads), 0, s, (const float *)partial, chunks,
                               total, dw, module_cc, Cin, K);
        FNP_LAUNCH_CHECK();
        return FNP_OK;
    }
#define FNP_WG(P)                                                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_partial_kernel<TX, TY, P>), grid, dim3(kThreads), lds, s, (const TX *)x,         \
                       (const TY *)dy, nbr, nbr_stride, n_out, cap_out, rows_per_chunk, Cin, Cout, partial)
    if (pairs <= 4 * kThreads) FNP_WG(4);
    else if (pairs <= 16 * kThreads) FNP_WG(16);
    else FNP_WG(64);
#undef FNP_WG
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(fnp_grid_for(total, kThreads)), dim3(kThreads), 0, s, (const float *)partial, chunks,
                       total, dw, module_cc, Cin, K);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

}  // namespace

// The module's weight (Cout, K, Cin) f32 -> the packed slabs (K, Cout, Cin) the convolutions read, in the activation dtype, and
// (training) the slabs the data gradient reads, in the SAME launch: mirror 1 = offsets mirrored (K - 1 - k, Cout, Cin) for the
// SubM data gradient on the forward's table, mirror 2 = mirrored and transposed (K - 1 - k, Cin, Cout): what the forward kernel
// reads when it runs on the gradient of a SubM layer, mirror 3 = transposed only (k, Cin, Cout): the same for a strided layer
// on its transposed table.
// (torch: a permute-copy and a cast per layer and step, plus a flip and a transpose-copy in the backward.)
template <typename T>
__global__ __launch_bounds__(kThreads) void pack_weight_kernel(const float *__restrict__ w, int Cout, int K, int Cin, T *__restrict__ packed,
                                                               T *__restrict__ mirror, int mode) {
    const long long total = (long long)Cout * K * Cin;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kThreads) {
        const int ci = (int)(e % Cin), co = (int)((e / Cin) % Cout), k = (int)(e / ((long long)Cin * Cout));   // e indexes `packed`
        const T v = (T)w[((long long)co * K + k) * Cin + ci];
        packed[e] = v;
        if (mode == 1) mirror[((long long)(K - 1 - k) * Cout + co) * Cin + ci] = v;
        else if (mode == 2) mirror[((long long)(K - 1 - k) * Cin + ci) * Cout + co] = v;
        else if (mode == 3) mirror[((long long)k * Cin + ci) * Cout + co] = v;
    }
}
extern "C" int fnp_pack_weight(const float *weight, int Cout, int K, int Cin, int dtype, void *packed, void *mirror, int mirror_mode,
                               fnp_stream_t stream) {
    if (!weight || !packed || Cout <= 0 || K <= 0 || Cin <= 0 || mirror_mode < 0 || mirror_mode > 3 || (mirror_mode != 0) != (mirror != nullptr))
        return FNP_ERR_ARG;
    const long long total = (long long)Cout * K * Cin;
    const dim3 grid(fnp_grid_for(total, kThreads, 1024));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FNP_F32)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(pack_weight_kernel<float>), grid, dim3(kThreads), 0, s, weight, Cout, K, Cin, (float *)packed, (float *)mirror, mirror_mode);
    else if (dtype == FNP_BF16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(pack_weight_kernel<__bf16>), grid, dim3(kThreads), 0, s, weight, Cout, K, Cin, (__bf16 *)packed, (__bf16 *)mirror, mirror_mode);
    else if (dtype == FNP_F16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(pack_weight_kernel<_Float16>), grid, dim3(kThreads), 0, s, weight, Cout, K, Cin, (_Float16 *)packed, (_Float16 *)mirror, mirror_mode);
    else
        return FNP_ERR_ARG;
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// fnp_pack_weight for up to 32 layers in ONE launch (blockIdx.y = layer): the training forward is bound by the host's launches
struct PackJobs {
    const float *w[32];
    void *packed[32], *mirror[32];
    int cout[32], k[32], cin[32], mode[32];
};
template <typename T>
__global__ __launch_bounds__(kThreads) void pack_weight_multi_kernel(PackJobs j) {
    const int l = blockIdx.y;
    const float *__restrict__ w = j.w[l];
    T *__restrict__ packed = (T *)j.packed[l];
    T *__restrict__ mirror = (T *)j.mirror[l];
    const int Cout = j.cout[l], K = j.k[l], Cin = j.cin[l], mode = j.mode[l];
    const long long total = (long long)Cout * K * Cin;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kThreads) {
        const int ci = (int)(e % Cin), co = (int)((e / Cin) % Cout), k = (int)(e / ((long long)Cin * Cout));
        const T v = (T)w[((long long)co * K + k) * Cin + ci];
        packed[e] = v;
        if (mode == 1) mirror[((long long)(K - 1 - k) * Cout + co) * Cin + ci] = v;
        else if (mode == 2) mirror[((long long)(K - 1 - k) * Cin + ci) * Cout + co] = v;
        else if (mode == 3) mirror[((long long)k * Cin + ci) * Cout + co] = v;
    }
}
extern "C" int fnp_pack_weight_multi(int count, const float *const *weights, const int *couts, const int *ks, const int *cins, int dtype,
                                     void *const *packed, void *const *mirrors, const int *mirror_modes, fnp_stream_t stream) {
    if (count <= 0 || count > 32 || !weights || !couts || !ks || !cins || !packed || !mirrors || !mirror_modes) return FNP_ERR_ARG;
    PackJobs j{};
    long long most = 1;
    for (int i = 0; i < count; ++i) {
        if (!weights[i] || !packed[i] || couts[i] <= 0 || ks[i] <= 0 || cins[i] <= 0 || mirror_modes[i] < 0 || mirror_modes[i] > 3 ||
            (mirror_modes[i] != 0) != (mirrors[i] != nullptr))
            return FNP_ERR_ARG;
        j.w[i] = weights[i]; j.packed[i] = packed[i]; j.mirror[i] = mirrors[i];
        j.cout[i] = couts[i]; j.k[i] = ks[i]; j.cin[i] = cins[i]; j.mode[i] = mirror_modes[i];
        const long long t = (long long)couts[i] * ks[i] * cins[i];
        if (t > most) most = t;
    }
    const dim3 grid(fnp_grid_for(most, kThreads, 256), count);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FNP_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(pack_weight_multi_kernel<float>), grid, dim3(kThreads), 0, s, j);
    else if (dtype == FNP_BF16) hipLaunchKernelGGL(HIP_KERNEL_NAME(pack_weight_multi_kernel<__bf16>), grid, dim3(kThreads), 0, s, j);
    else if (dtype == FNP_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(pack_weight_multi_kernel<_Float16>), grid, dim3(kThreads), 0, s, j);
    else return FNP_ERR_ARG;
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_rulebook_transpose(const int *nbr, int nbr_stride, int K, const int *n_out, int cap_out, int *nbr_t,
                                      int cap_in, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!nbr || !n_out || !nbr_t || K <= 0 || cap_out <= 0 || cap_in <= 0 || nbr_stride < cap_out) return FNP_ERR_ARG;
    {
        const int frc = fnp_fill_words(nbr_t, (long long)K * cap_in, 0xffffffffu, s);
        if (frc) return frc;
    }
    hipLaunchKernelGGL(transpose_rulebook_kernel, dim3(fnp_grid_for(cap_out, kThreads, 1024), K), dim3(kThreads), 0, s, nbr,
                       nbr_stride, K, n_out, cap_out, nbr_t, cap_in);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int64_t fnp_spconv_wgrad_workspace_bytes(int K, int Cin, int Cout) {
    if (K <= 0 || Cin <= 0 || Cout <= 0) return 0;
    return (int64_t)max_chunks(Cin, Cout) * K * Cin * Cout * 4;
}

extern "C" int fnp_spconv_wgrad(const void *feat_in, int in_dtype, const void *grad_out, int grad_dtype, const int *nbr,
                                int nbr_stride, int K, const int *n_out, int cap_out, float *grad_weight, int grad_layout, int Cin, int Cout,
                                void *workspace, int64_t workspace_bytes, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!feat_in || !grad_out || !nbr || !n_out || !grad_weight || !workspace || K <= 0 || Cin <= 0 || Cout <= 0 ||
        cap_out <= 0 || nbr_stride < cap_out || (grad_layout != 0 && grad_layout != 1))
        return FNP_ERR_ARG;
    if (in_dtype == FNP_F32 && grad_dtype == FNP_F32)
        return run_wgrad<float, float>(feat_in, grad_out, nbr, nbr_stride, K, n_out, cap_out, grad_weight, Cin, Cout, workspace,
                                       workspace_bytes, s, grad_layout);
    if (in_dtype == FNP_BF16 && grad_dtype == FNP_BF16)
        return run_wgrad<__bf16, __bf16>(feat_in, grad_out, nbr, nbr_stride, K, n_out, cap_out, grad_weight, Cin, Cout, workspace,
                                         workspace_bytes, s, grad_layout);
    if (in_dtype == FNP_F16 && grad_dtype == FNP_F16)
        return run_wgrad<_Float16, _Float16>(feat_in, grad_out, nbr, nbr_stride, K, n_out, cap_out, grad_weight, Cin, Cout,
                                             workspace, workspace_bytes, s, grad_layout);
    if (in_dtype == FNP_F32 && grad_dtype == FNP_BF16)
        return run_wgrad<float, __bf16>(feat_in, grad_out, nbr, nbr_stride, K, n_out, cap_out, grad_weight, Cin, Cout, workspace,
                                        workspace_bytes, s, grad_layout);
    if (in_dtype == FNP_BF16 && grad_dtype == FNP_F32)
        return run_wgrad<__bf16, float>(feat_in, grad_out, nbr, nbr_stride, K, n_out, cap_out, grad_weight, Cin, Cout, workspace,
                                        workspace_bytes, s, grad_layout);
    return FNP_ERR_ARG;
}

// Pair lists of a rulebook (see wgrad_mfma_kernel): pair_o, pair_i (K, pair_stride) int32 with pair_stride >= cap_out — the rows
// the caller knows to exist, not the table's stride (a strided layer's table is sized for 27 outputs per input) —, pair_count (K).
extern "C" int64_t fnp_rulebook_pairs_workspace_bytes(int K, int cap_out) {
    if (K <= 0 || cap_out <= 0) return 0;
    return (int64_t)K * fnp_divup(cap_out, kPairTile) * 4;
}

extern "C" int fnp_rulebook_pairs(const int *nbr, int nbr_stride, int K, const int *n_out, int cap_out, int *pair_o, int *pair_i,
                                  int pair_stride, int *pair_count, void *workspace, int64_t workspace_bytes, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!nbr || !n_out || !pair_o || !pair_i || !pair_count || !workspace || K <= 0 || K > 65535 || cap_out <= 0 || nbr_stride < cap_out ||
        pair_stride < cap_out)
        return FNP_ERR_ARG;
    if (fnp_rulebook_pairs_workspace_bytes(K, cap_out) > workspace_bytes) return FNP_ERR_WORKSPACE;
    const int ntiles = fnp_divup(cap_out, kPairTile);
    int *counts = (int *)workspace;
    hipLaunchKernelGGL(pairs_count_kernel, dim3(ntiles, K), dim3(kThreads), 0, s, nbr, nbr_stride, n_out, cap_out, ntiles, counts);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(pairs_scan_kernel, dim3(K), dim3(kThreads), 0, s, ntiles, counts, pair_count);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(pairs_emit_kernel, dim3(ntiles, K), dim3(kThreads), 0, s, nbr, nbr_stride, n_out, cap_out, ntiles, (const int *)counts, pair_o,
                       pair_i, pair_stride);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// fnp_spconv_wgrad on the pair lists: 16-bit features and gradients of the MFMA channel pairs; FNP_ERR_ARG otherwise (take
// fnp_spconv_wgrad).  Same sum in another grouping of the rows: equal to fnp_spconv_wgrad's up to f32 rounding, run-to-run identical.
extern "C" int fnp_spconv_wgrad_pairs(const void *feat_in, int in_dtype, const void *grad_out, int grad_dtype, const int *pair_o,
                                      const int *pair_i, const int *pair_count, int pair_stride, int K, const int *n_out, int cap_out,
                                      float *grad_weight, int grad_layout, int Cin, int Cout, void *workspace, int64_t workspace_bytes,
                                      fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!feat_in || !grad_out || !pair_o || !pair_i || !pair_count || !n_out || !grad_weight || !workspace || K <= 0 || Cin <= 0 || Cout <= 0 ||
        cap_out <= 0 || pair_stride < cap_out || (grad_layout != 0 && grad_layout != 1))
        return FNP_ERR_ARG;
    if (in_dtype == FNP_F32 && grad_dtype == FNP_F32 && (long long)Cin * Cout <= 128)   // (conv_input: the few-pairs kernel on the lists)
        return run_wgrad<float, float>(feat_in, grad_out, pair_o, pair_stride, K, n_out, cap_out, grad_weight, Cin, Cout, workspace,
                                       workspace_bytes, s, grad_layout, pair_i, pair_count);
    if (in_dtype == FNP_BF16 && grad_dtype == FNP_BF16)
        return run_wgrad<__bf16, __bf16>(feat_in, grad_out, pair_o, pair_stride, K, n_out, cap_out, grad_weight, Cin, Cout, workspace,
                                         workspace_bytes, s, grad_layout, pair_i, pair_count);
    if (in_dtype == FNP_F16 && grad_dtype == FNP_F16)
        return run_wgrad<_Float16, _Float16>(feat_in, grad_out, pair_o, pair_stride, K, n_out, cap_out, grad_weight, Cin, Cout,
                                             workspace, workspace_bytes, s, grad_layout, pair_i, pair_count);
    return FNP_ERR_ARG;
}
