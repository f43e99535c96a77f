// Sparse convolution forward for gfx950: output-stationary implicit GEMM over the rulebook
// nbr[k][o] (rulebook.hip) with the BatchNorm1d(eval) / residual / ReLU epilogue of
// SparseBasicBlock and post_act_block fused in
// (pcdet/models/backbones_3d/spconv_backbone.py:8-27,51-67).  Replaces spconv's
// SubMConv3d / SparseConv3d forward (gather -> GEMM -> scatter-add).
//
//   out[o, :] = act( (sum_k W_k^T x[nbr[k][o], :]) * scale + shift + residual[o, :] )
//
// Two code paths:
//   * bf16 features/weights, fp32 accumulate on MFMA (v_mfma_f32_16x16x32_bf16).  The product is
//     formed transposed, D = W_k^T (A operand, 16 out-channels x 32 in-channels) times
//     X^T (B operand, 32 in-channels x 16 sites): both fragments are 16 contiguous bytes per
//     lane straight from HBM/L2 (weights are pre-packed [K][Cout][Cin]; a gathered feature row
//     is contiguous in Cin), so no transpose is needed, and each lane ends up with 4
//     consecutive output channels of one site -> 8-byte bf16 stores.  One wave owns MB*16 sites
//     and all Cout; the weight slab of each kernel offset is shared by the workgroup's waves
//     through a swizzled LDS image (details at the kernel).
//   * f32 validation path on the VALU: a k-ascending, cin-ascending fmaf chain per output
//     element, the same chain the CPU oracle evaluates, so it is bit-comparable.
// No atomics anywhere: every output row is written once, results are run-to-run identical.
#include "rankgrid.h"
#include <type_traits>
#include <cstdlib>

namespace {

// 16-bit activation types of the MFMA path: bf16 (default) and fp16 (the reference's AMP mode, train_utils.py:172:
// autocast makes spconv run fp16 features with fp32 accumulation).  Same kernel, same fragment layouts
// (v_mfma_f32_16x16x32_bf16 / _f16 take the same cycles); only the matrix instruction and the conversions differ.
template <typename T> struct Vec16 {
    typedef T v8 __attribute__((ext_vector_type(8)));
    typedef T v4 __attribute__((ext_vector_type(4)));
};
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma16(Vec16<__bf16>::v8 a, Vec16<__bf16>::v8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma16(Vec16<_Float16>::v8 a, Vec16<_Float16>::v8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(_Float16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ _Float16 from_f32<_Float16>(float v) { return (_Float16)v; }

// ------------------------------------------------------------------------------------------
// VALU path (any Cin/Cout, any dtype mix): thread per (row, cout).
// ------------------------------------------------------------------------------------------
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void spconv_valu_kernel(const TIn *__restrict__ x, const TIn *__restrict__ w,
                                                          const int *__restrict__ nbr, int nbr_stride, int K,
                                                          const int *__restrict__ n_out, int cap,
                                                          TOut *__restrict__ y, const float *__restrict__ scale,
                                                          const float *__restrict__ shift,
                                                          const TOut *__restrict__ residual, int relu, int Cin, int Cout) {
    const int n = min(*n_out, cap);
    const long long total = (long long)n * Cout;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / Cout), co = (int)(t % Cout);
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const int idx = nbr[(size_t)k * nbr_stride + row];
            if (idx < 0) continue;
            const TIn *xr = x + (size_t)idx * Cin;
            const TIn *wr = w + ((size_t)k * Cout + co) * Cin;
            for (int ci = 0; ci < Cin; ++ci) acc = fmaf(to_f32(xr[ci]), to_f32(wr[ci]), acc);
        }
        float v = acc;
        if (scale) v = v * scale[co] + shift[co];
        if (residual) v = v + to_f32(residual[(size_t)row * Cout + co]);
        if (relu && v < 0.f) v = 0.f;
        y[(size_t)row * Cout + co] = from_f32<TOut>(v);
    }
}

// ------------------------------------------------------------------------------------------
// First layer (conv_input: f32 point features, Cin <= 8 -> COUT): one thread per output row keeps
// all COUT accumulators in registers; the whole weight set (K x Cin x COUT f32, 8.6 KB for
// 27 x 5 x 16) sits in LDS and is read as wave-wide broadcasts.  Per (row, cout) the fmaf chain
// is the same k-ascending, cin-ascending chain as spconv_valu_kernel / the oracle.
// ------------------------------------------------------------------------------------------
template <int COUT, typename TOut>
__global__ __launch_bounds__(256) void spconv_first_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                           const int *__restrict__ nbr, int nbr_stride, int K,
                                                           const int *__restrict__ n_out, int cap,
                                                           TOut *__restrict__ y, const float *__restrict__ scale,
                                                           const float *__restrict__ shift,
                                                           const TOut *__restrict__ residual, int relu, int Cin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fnp_smem[];
    float *wl = reinterpret_cast<float *>(fnp_smem);   // [k][ci][co]
    for (int i = threadIdx.x; i < K * Cin * COUT; i += 256) {
        const int k = i / (Cin * COUT), r = i % (Cin * COUT);
        const int ci = r / COUT, co = r % COUT;
        wl[i] = w[((size_t)k * COUT + co) * Cin + ci];
    }
    __syncthreads();
    const int n = min(*n_out, cap);
    for (int row = blockIdx.x * 256 + threadIdx.x; row < n; row += gridDim.x * 256) {
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
        // Branch-free sweep in groups of 9 offsets: the rulebook entries of a group are loaded
        // together, the feature rows unconditionally (row 0 stands in for an absent neighbour and is
        // multiplied by 0: fmaf(0, w, acc) == acc, the chain of the present neighbours is unchanged),
        // so the loads of a group are all in flight before its first fma instead of one dependent
        // load per offset.
        for (int k0 = 0; k0 < K; k0 += 9) {
            int idx[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) idx[j] = k0 + j < K ? nbr[(size_t)(k0 + j) * nbr_stride + row] : -1;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                if (k0 + j >= K) break;   // uniform
                const bool ok = idx[j] >= 0;
                const float *xr = x + (size_t)(ok ? idx[j] : 0) * Cin;
                const float *wk = wl + (k0 + j) * Cin * COUT;
                float xv[8];   // (Cin <= 8 by dispatch)
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) xv[ci] = ci < Cin ? xr[ci] : 0.f;
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) {
                    if (ci >= Cin) break;   // uniform
                    const float xc = ok ? xv[ci] : 0.f;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xc, wk[ci * COUT + co], acc[co]);
                }
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float v = acc[co];
            if (scale) v = v * scale[co] + shift[co];
            if (residual) v = v + to_f32(residual[(size_t)row * COUT + co]);
            if (relu && v < 0.f) v = 0.f;
            y[(size_t)row * COUT + co] = from_f32<TOut>(v);
        }
    }
}

template <typename TOut>
int launch_first(const void *x, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap, void *y,
                 const float *scale, const float *shift, const void *residual, int relu, int Cin, hipStream_t s) {
    const int grid = fnp_grid_for(cap, 256, 2048);
    const size_t lds = sizeof(float) * (size_t)K * Cin * 16;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(spconv_first_kernel<16, TOut>), dim3(grid), dim3(256), lds, s, (const float *)x,
                       (const float *)w, nbr, nbr_stride, K, n_out, cap, (TOut *)y, scale, shift, (const TOut *)residual,
                       relu, Cin);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// ------------------------------------------------------------------------------------------
// MFMA path.
//
// Workgroup = NW waves (4; 8 for 128 -> 128 channels); wave w owns MB*16 output sites and all COUT
// channels, accumulators in registers for the whole sweep over the K kernel offsets.  The weight slab
// W_k (COUT x CIN bf16) is shared by the waves through LDS:
//   * ALLK  (K*slab <= 64 KiB: the 16/32-channel layers): every slab is staged once per
//     persistent workgroup and the offset loop runs without barriers;
//   * else  (64/128-channel layers): slabs are double-buffered; W_{k+1} is fetched a few 16-byte
//     chunks per MFMA step into registers and written to the other LDS buffer one step later
//     (one barrier per offset), so its HBM/L2 latency hides under the matrix work.
// LDS image: row r (one output channel, CIN*2 bytes = CH 16-byte chunks) stores logical chunk c
// at physical chunk c ^ ((r >> SW) & (CH-1)); with that XOR every 16-lane group of the
// ds_read_b128 fragment reads (16 rows x one logical chunk) hits 16 distinct 16-byte slots of the
// 256-byte bank row: conflict-free (SQ_LDS_BANK_CONFLICT = 0 measured).
// Feature fragments are gathered straight from HBM/L2 through a buffer descriptor (16 contiguous
// bytes per lane; rows absent from the rulebook present an out-of-range offset and read as zeros
// without touching memory).  The kernel is bound by the per-CU rate of these gathers,
// so they run PFK whole kernel offsets ahead of the matrix work: the fragments of offset k + PFK
// are requested into the registers that offset k has just finished with, and the rulebook
// indices they need were themselves fetched PFK offsets earlier.
// KVOL: kernel volume known at compile time (27) or 0 = runtime K; it also gives the 3x3x3
// layers and conv_out (K = 3) distinct kernel names for per-layer-class profiler statistics.
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int KVOL>
struct MfmaCfg {
    // PAIR (16 input channels, 3x3x3): two kernel offsets share one MFMA step.  A 16-channel row fills only
    // half of the 32-wide K step and of a gather instruction's lanes, and these layers are bound by the NUMBER
    // of gather instructions (15.6 clk per 64-lane b128 instruction per CU, whether or not its lanes are in
    // range): lanes q < 2 take offset 2p, lanes q >= 2 offset 2p + 1, the weight image holds [W_2p | W_2p+1]
    // per output channel, and the sweep has 14 steps instead of 27.
#ifndef FNP_PAIR16
#define FNP_PAIR16 1
#endif
    static constexpr bool PAIR = FNP_PAIR16 && CIN == 16 && KVOL == 27;
    static constexpr int KEFF = PAIR ? (KVOL + 1) / 2 : KVOL;   // steps of the offset sweep
    static constexpr int CH = PAIR ? 4 : CIN / 8;         // 16-byte chunks per weight row (image row)
    static constexpr int SLAB = COUT * CH;                // chunks per slab
    static constexpr int SW = (CH == 8 || CH == 4) ? 1 : 0;
    static constexpr bool ALLK = KVOL > 0 && (long long)KEFF * SLAB * 16 <= 65536;
    // WPAIR (128 -> 128, 8-wave workgroups): a ring of four slab buffers and ONE barrier per two offsets
#ifndef FNP_WPAIR
#define FNP_WPAIR 0
#endif
    static constexpr bool WPAIR = FNP_WPAIR && !ALLK && CIN == 128 && COUT == 128;
    static constexpr int LDS_BYTES = (ALLK ? KEFF : WPAIR ? 4 : 2) * SLAB * 16;
    static constexpr int KS = (CIN + 31) / 32;            // 32-wide K steps of the MFMA
    // gather prefetch distance in kernel offsets: 16 gathers in flight per wave
#ifndef FNP_PFK128
#define FNP_PFK128 1   // (probe, round 5: fragments of the 128 -> 128 sweep requested TWO offsets ahead)
#endif
    static constexpr int PFK = ALLK ? (KS == 1 ? 4 : 2) : (CIN == 128 && COUT == 128 && KVOL == 27) ? FNP_PFK128 : 1;
    // feature window (WIN kernels, 32/64-channel layers): rows [tile - WH, tile + rows + WH) of the
    // input tensor are staged in LDS once per tile; WZERO bytes of zeros follow them
#ifndef FNP_WH32
#define FNP_WH32 64
#endif
    static constexpr int WH = CH == 4 ? FNP_WH32 : 64;
    static constexpr int WZERO = 128;
    static constexpr int XLB = KS == 1 ? 2 : 1;           // window fragments are read XLB offsets ahead
    static constexpr int win_rows(int nw, int mb) { return nw * mb * 16 + 2 * WH; }
    // wide epilogue (bf16 outputs, >= 32 channels): one 16-site block of the wave's tile is transposed through a
    // wave-private LDS strip so that residual reads and output stores move 16 bytes per lane over whole rows
    // (128-byte cache lines) instead of 8 bytes per lane over 32-byte row pieces.
#ifndef FNP_WIDE_EPI
#define FNP_WIDE_EPI 1
#endif
    static constexpr bool WIDE = FNP_WIDE_EPI && COUT >= 32 && !WPAIR;   // (WPAIR: no LDS left for the strips -> swap form)
    static constexpr int ESTRIDE = COUT * 2 + 16;   // bytes per staged row
#ifndef FNP_EPI_SITES
#define FNP_EPI_SITES 16
#endif
    static constexpr int epi_sites(bool win) { return win ? 8 : (COUT >= 128 ? FNP_EPI_SITES : 16); }   // sites per strip pass (window kernels: LDS is tight)
    static constexpr int epi_bytes(int nw, bool win, bool out16) { return (WIDE && out16 && !win) ? nw * epi_sites(win) * ESTRIDE : 0; }
    static constexpr int lds_bytes(int nw, int mb, bool win) { return LDS_BYTES + (win ? win_rows(nw, mb) * CH * 16 + WZERO : 0); }
};

// Development-only ablation bit mask (tools/bench_conv.py with FNP_LIB_PATH): 1 = no feature gathers,
// 2 = no weight staging, 4 = no MFMA, 8 = window kernels issue no global gathers, 16 = no window reads,
// 32 = rulebook entries read from a 64 KiB (cache-resident) slice of the table, 64 = window address taken
// from the entry without arithmetic (timing probe for pre-computed addresses), 128 = no output stores,
// 256 = no residual loads, 512 = no rulebook loads (every neighbour row + k - 13 present).  The shipped library is built with FNP_ABLATE == 0.
#ifndef FNP_ABLATE
#define FNP_ABLATE 0
#endif
#ifndef FNP_NT_STORE
#define FNP_NT_STORE 0   // (development: output rows stored with the non-temporal hint — measured within the noise of a box, round 4)
#endif
// L2 line-touch prefetch of the rows above a tile (bit 0) — see run_tile
#ifndef FNP_PF
#define FNP_PF 0
#endif
// window kernel epilogue: 16-byte accesses through lane-row swaps (see the epilogue)
#ifndef FNP_SWAP_EPI
#define FNP_SWAP_EPI 1
#endif

// waves per SIMD the register budget is held to: 3 (<= 168 VGPRs) where it measured faster on
// MI355X (the channel-doubling strided layer 32->64: a third resident workgroup per CU
// shortens the last, partly filled round of tiles), 4 for 16->32, 2 elsewhere (3 costs spills there)
template <int CIN, int COUT> struct MfmaOcc;

// waves per workgroup: 8 for the 128-channel layers (one 32 KiB weight slab per offset then serves 384 sites
// instead of 192: the slab stream through L2 is the largest term of those layers), 4 elsewhere
#ifndef FNP_NW128
#define FNP_NW128 8
#endif
#ifndef FNP_NW_MINCIN
#define FNP_NW_MINCIN 128
#endif
#ifndef FNP_NW32
#define FNP_NW32 8
#endif
#ifndef FNP_NW64
#define FNP_NW64 8
#endif
template <int CIN, int COUT> struct MfmaWg { static constexpr int NW = (COUT == 128 && CIN >= FNP_NW_MINCIN) ? FNP_NW128 : (CIN == 32 && COUT == 32) ? FNP_NW32 : (CIN == 64 && COUT == 64) ? FNP_NW64 : 4; };

#ifndef FNP_OCC1616
#define FNP_OCC1616 4
#endif
#ifndef FNP_OCC3232
#define FNP_OCC3232 4
#endif
#ifndef FNP_OCC6464
#define FNP_OCC6464 4
#endif
#ifndef FNP_OCC128
#define FNP_OCC128 2
#endif
template <int CIN, int COUT> struct MfmaOcc { static constexpr int WAVES = (CIN == 128 && COUT == 128) ? FNP_OCC128 : (CIN == 16 && COUT == 16) ? FNP_OCC1616 : (CIN == 32 && COUT == 32) ? FNP_OCC3232 : (CIN == 64 && COUT == 64) ? FNP_OCC6464 : (CIN == 16 && COUT == 32) ? 4 : ((CIN < COUT && COUT <= 64) || MfmaWg<CIN, COUT>::NW == 6) ? 3 : 2; };

// FUSED (strided 3x3x3 layers of the fused backbone): the rulebook of such a layer is used exactly once, so the
// kernel computes the 27 entries of its rows itself — each wave, at the top of a tile, one lane per row with the
// same nbr_row() the rulebook kernels use (coordinates -> <= 8 occupancy words of the INPUT grid -> bit
// arithmetic), into a wave-private LDS strip that the offset sweep then reads instead of the table.  No
// 108-byte-per-row table is written and read back (0.5 GB per step at 64 scenes), and the index fetches leave
// the in-order VMEM queue.
struct FusedRb {
    RG gi;                    // input rank grid (rows = ranks: perm unused unless the grid carries one)
    int s[3], p[3];           // stride, padding (kernel 3x3x3)
    const int *out_coords;    // (cap, 4) [b, z, y, x] of the output rows
    // OR (ell_rec != nullptr): the rows' entries come from the COMPACT rulebook (spconv_ell.hip: 32-byte records, the entries
    // that exist in ascending offset order, chained extension records behind the ell_cap row records).  The 16-channel
    // layers have 2-4 of 27 neighbours: their (27, cap) table is 108 bytes per row against 32 + 32 of features, the largest
    // stream of those launches, and its 27 entry loads per row sit in the same in-order VMEM queue as the gathers.  A wave
    // expands the records of its tile rows into its LDS strip at the top of a tile (two 16-byte loads per row, a chain
    // step for the 8 % of rows with more than 8 neighbours) and the sweep runs as on the table: same products, same order.
    const unsigned *ell_rec;
    int ell_cap;
};

// SORTED (the 128 -> 128 SubM layers of stage 4): the rows a workgroup owns are processed in an order sorted by
// neighbourhood class (fnp_rulebook_classsort: no neighbour below / above / both / neither in z), so that most tiles hold
// rows of one class and the tile sweeps only the kernel offsets at least one of its rows has a neighbour at — the
// offset's slab load, barrier, gathers and matrix work all go (20 % of the (tile, offset) pairs on lidar scenes;
// unsorted, every tile needs every offset).  `perm` maps a processing position to its row (rulebook entries, residual
// and output rows are addressed through it), `blockmask` holds the union of the 27-bit neighbour masks of each 16
// positions.  Every row still sums its own neighbours in ascending offset order: same values as the unsorted sweep.
struct SortedRb {
    const int *perm;
    const unsigned *blockmask;
};

// f32 outputs only (the bf16x3 engine's main product, fnp_spconv_forward_split): a 16-bit addend (its two cross terms, which need
// 16-bit precision only) joins the sum before the ReLU, and the f32 result also leaves as a (hi, lo) pair of 16-bit rows with
// hi + lo == y to 2^-17 |y| — the next layer's operands, written where they are made instead of by a pass of their own.
struct SplitOut {
    const void *addend;
    void *hi, *lo;
    int relu;   // (applied after the addend; the kernel's own `relu` is 0 then)
};

// rows [row_begin, row_end) of workgroup range `range` of `G` when n rows are cut at 16-row blocks: the split of
// spconv_mfma_kernel, shared with the class-sort pass (which must sort exactly the rows a workgroup will own)
__device__ __forceinline__ void fnp_range_rows(int n, int range, int G, int &row_begin, int &row_end) {
    const long long nblk16 = (n + 15) >> 4;
    row_begin = (int)((nblk16 * range) / G) << 4;
    row_end = min(n, (int)((nblk16 * (range + 1)) / G) << 4);
}

// SORTED work split.  Blocks b and b + 8 share an XCD (and its L2); the slots of one such group own ONE contiguous run of rows
// [X0, X1) together and take its tiles round-robin: round j = positions [X0 + j S T, X0 + (j + 1) S T), slot s its s-th
// tile of T rows.  At any time the S workgroups of a group sweep S consecutive tiles — one contiguous region of the
// feature map, about the XCD's L2 in size — and the class sort orders the rows of each ROUND: of its S tiles all but the
// two or three at the class boundaries hold rows of one class.  (Sorting the rows of a private per-workgroup range
// instead made every tile gather from the whole range: L2 misses + 31 %, and most of the skipped offsets' time went back
// into gather latency.)  The last, partial round is cut into S tiles of fewer blocks per wave, so that the matrix work of
// the tail stays proportional to its rows.  Placement only affects speed, never results.
struct XcdRows {
    int X0, X1, S;   // rows of the group, number of slots
};
__device__ __forceinline__ XcdRows fnp_xcd_rows(int n, int G, int xcd) {
    const int per = G >> 3, rem = G & 7;
    XcdRows x;
    x.S = per + (xcd < rem ? 1 : 0);
    x.X0 = x.X1 = 0;
    if (x.S > 0) {
        const int first = xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per;
        int t;
        fnp_range_rows(n, first, G, x.X0, t);
        fnp_range_rows(n, first + x.S - 1, G, t, x.X1);
    }
    return x;
}
// blocks per wave (0 = no tile) of the partial last round of `rows_left` rows cut into S tiles of NW waves
__device__ __forceinline__ int fnp_tail_blocks(int rows_left, int S, int NW) {
    const int nb = (rows_left + 15) >> 4;
    return ((nb + S - 1) / S + NW - 1) / NW;
}

#ifdef FNP_MFMA_STAMP
__device__ unsigned long long g_mfma_stamps[8];
#define FNP_MS_NOW(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define FNP_MS(ph) do { if (CIN == 128 && COUT == 128 && KVOL == 27) { unsigned long long n_; FNP_MS_NOW(n_); ms_acc[ph] += n_ - ms_prev; ms_prev = n_; } } while (0)
#else
#define FNP_MS(ph)
#endif
// NWO: waves per workgroup when not the layer class's own (MfmaWg): the SMALL-INPUT form of the 128 -> 128 layers.  With a few
// thousand rows (one scene: the reference's extraction runs batch size 1) every workgroup holds one 16-row block per wave, and an
// offset then costs what eight waves need to read the whole 32 KB slab as A fragments — 256 KB of LDS reads for 128 rows;
// four waves read half of that per row and the rows spread over twice the workgroups.
template <int CIN, int COUT, int NWO> struct NwOf { static constexpr int value = NWO ? NWO : MfmaWg<CIN, COUT>::NW; };
template <int CIN, int COUT, int MB, int KVOL, bool WIN, typename TOut, bool FUSED = false, typename TAct = __bf16, bool SORTED = false, int NWO = 0>
__global__ __launch_bounds__((NwOf<CIN, COUT, NWO>::value * 64), (MfmaOcc<CIN, COUT>::WAVES)) void spconv_mfma_kernel(const TAct *__restrict__ x, int x_bytes,
                                                             const TAct *__restrict__ w,
                                                             const int *__restrict__ nbr, int nbr_stride, int Krt,
                                                             const int *__restrict__ n_out, int cap,
                                                             TOut *__restrict__ y, const float *__restrict__ scale,
                                                             const float *__restrict__ shift,
                                                             const TOut *__restrict__ residual, int relu, int hints, FusedRb frb, SortedRb srb, SplitOut so) {
    using Cfg = MfmaCfg<CIN, COUT, KVOL>;
    constexpr int NWX = NwOf<CIN, COUT, NWO>::value;
    static_assert(!SORTED || (KVOL == 27 && !WIN && !FUSED && !Cfg::PAIR && !Cfg::ALLK), "sorted sweep: wide double-buffered 3x3x3 layers");
    using bf16x8 = typename Vec16<TAct>::v8;   // (named after the default activation type)
    using bf16x4 = typename Vec16<TAct>::v4;
    static_assert(sizeof(TOut) == 4 || std::is_same<TOut, TAct>::value, "16-bit outputs have the activation type");
    static_assert(!FUSED || (KVOL == 27 && !WIN), "fused rulebook: 3x3x3 strided layers");
    constexpr int CH = Cfg::CH, SLAB = Cfg::SLAB, SW = Cfg::SW, KS = Cfg::KS, PFK = Cfg::PFK;
    constexpr bool ALLK = Cfg::ALLK;
    constexpr int XLB = Cfg::XLB, WROWS = Cfg::win_rows(NWX, MB), WH = Cfg::WH;
    static_assert(!WIN || ((CH == 4 || CH == 8) && XLB <= PFK && PFK % XLB == 0), "window path: 32/64 input channels");
    constexpr int NB = COUT / 16;         // 16-channel output blocks
    constexpr int NBH = NB < 4 ? NB : 4;  // A fragments held at once
    constexpr int ROWS_PER_WAVE = MB * 16;
    constexpr int NW = NWX, NT = NW * 64;   // waves / threads per workgroup
    constexpr int ROWS_PER_WG = NW * ROWS_PER_WAVE;
    // weight staging of the double-buffered path: NCH chunks per thread per slab, WST per MFMA step
    constexpr int NCH = (SLAB + NT - 1) / NT;
    constexpr int WST = (NCH + KS - 1) / KS;
    // small slabs (<= 2 chunks per thread): W_{k+2} is requested at the top of offset k and W_{k+1}
    // (requested one offset earlier) is written to LDS at its end, so the weights get two offsets of
    // matrix work to arrive; larger slabs cannot afford the registers and go step by step
    constexpr bool WDEEP = !ALLK && NCH <= 1;
    // 8-wave workgroups (128 -> 128): one chunk per thread and MFMA step, each held in a register for a whole
    // offset: chunk ks of W_{k+2} is requested at step ks of offset k and written to LDS at step ks of offset
    // k + 1, so the slab stream never waits for its own L2 latency (a step is ~0.15 us, the latency ~0.5-1 us)
#ifndef FNP_WD4
#define FNP_WD4 1
#endif
#ifndef FNP_WD4_64128
#define FNP_WD4_64128 1
#endif
    // (64 -> 128, four waves: 4 chunks per thread and slab, two per MFMA step — the same two-offsets-ahead register pipeline; its
    //  step-by-step staging left a one-scene launch waiting for the L2 once per offset: 33 us, 18.6 with no staging at all)
    constexpr bool WD4 = FNP_WD4 && !ALLK && (((NW == 8 || (FNP_WD4_64128 && NW == 4 && CIN == 128 && COUT == 64)) && NCH == KS && NCH == 4) || (NWO == 4 && NCH == 2 * KS && KS == 4) ||
                                              (FNP_WD4_64128 && NW == 4 && NCH == 2 * KS && KS == 2 && CIN == 64 && COUT == 128));
    constexpr int WDN = WD4 ? NCH / KS : 1;   // chunks per thread and MFMA step (2 in the four-wave small-input form)
    constexpr bool WPAIR = Cfg::WPAIR && WD4;
    static_assert(CIN % 16 == 0 && COUT % 16 == 0, "channel counts must be multiples of 16");
    static_assert(ALLK || SLAB % NT == 0 || SLAB < NT, "unsupported slab size");

    extern __shared__ __attribute__((aligned(16))) unsigned char fnp_smem[];
    uint4 *wl = reinterpret_cast<uint4 *>(fnp_smem);
    // window image: row d (relative to the window start) stores logical chunk c at physical chunk
    // c ^ sw(d); sw is chosen so that the four 16-lane groups of a ds_read_b128 B-fragment read
    // (16 consecutive rows x one logical chunk per quarter wave) are conflict-free
    unsigned char *const win = fnp_smem + Cfg::LDS_BYTES;
    constexpr unsigned WIN_ZERO = (unsigned)(Cfg::LDS_BYTES + WROWS * CH * 16);   // byte address of the zeros
    auto win_sw = [](unsigned d) -> unsigned { return CH == 4 ? ((0u - (d >> 2)) & 3u) : ((d >> 1) & 7u); };

    constexpr bool PAIR = Cfg::PAIR;
#ifndef FNP_CONV_PRIO
#define FNP_CONV_PRIO 0
#endif
#ifndef FNP_SWEEP_PRIO
#define FNP_SWEEP_PRIO 0
#endif
#ifndef FNP_GPAIR
#define FNP_GPAIR 1   // (round 5: 128 -> 128 class-sorted 0.765 -> 0.751 ms per launch at 128 scenes, 0.397 -> 0.390 at 64, conv_out -5 %; bit-identical)
#endif
    constexpr bool GPAIR = FNP_GPAIR && !WIN && KS % 2 == 0 && PFK == 1;
#ifdef FNP_KLIM   // timing probe only (results are wrong): sweep the first FNP_KLIM offsets
    const int K = KVOL > 0 ? (Cfg::KEFF < FNP_KLIM ? Cfg::KEFF : FNP_KLIM) : Krt;
#else
    const int K = KVOL > 0 ? Cfg::KEFF : Krt;   // (PAIR: offset pairs)
#endif
#if FNP_CONV_PRIO
    __builtin_amdgcn_s_setprio(FNP_CONV_PRIO);   // (probe: convolution waves ahead of the index kernels that share the CUs in the replayed step)
#endif
    const int n = min(*n_out, cap);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const bool kvalid0 = PAIR || (q * 8) < CIN;  // for CIN == 16 without pairing only lanes 0..31 carry data in a K step
    const int qk = PAIR ? (q >> 1) : 0;          // PAIR: which offset of the pair this lane works on

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    // byte offset of (row id, this lane's 16-byte chunk of MFMA step 0); absent rows get an offset
    // that stays out of range after the + ks*64 of the later steps
    auto row_off = [&](int id) -> unsigned {
        return ((FNP_ABLATE & 1) || id < 0 || !kvalid0) ? 0x80000000u
                                                        : (unsigned)id * (unsigned)(CIN * 2) + (unsigned)(PAIR ? (q & 1) : q) * 16u;
    };
    auto gather = [&](unsigned roff, int ks) -> bf16x8 {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (!(WIN && (FNP_ABLATE & 8))) v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, roff + (unsigned)ks * 64u, 0, 0);
        return *reinterpret_cast<bf16x8 *>(&v);
    };
    // rulebook entry of row r for offset k; rows past the range and offsets past K read a valid
    // address and yield -1 (no data-dependent branch around a load)
    // (PAIR: step k stands for the offsets 2k and 2k + 1; the lane's own one is 2k + qk, absent past KVOL - 1)
    // (FUSED: the entries of the wave's tile rows sit in its LDS strip [offset][row of the tile], -1 for rows past
    //  the range)
    constexpr int SR = MB * 16;   // rows of a wave tile
    int *const fstrip = reinterpret_cast<int *>(fnp_smem + Cfg::lds_bytes(NWX, MB, WIN) +
                                                Cfg::epi_bytes(NWX, WIN, sizeof(TOut) == 2)) + wave * (27 * SR);
    int frow0 = 0;                // first row of the wave's current tile (set by the tile body)
    // BatchNorm scale and shift in LDS (behind every other region): an epilogue reads them per tile, and from L2 that is a
    // ~800-cycle round trip at the end of a sweep that takes 3-6 k cycles on the narrow layers
    float *const ss_lds = reinterpret_cast<float *>(fnp_smem + Cfg::lds_bytes(NWX, MB, WIN) +
                                                    Cfg::epi_bytes(NWX, WIN, sizeof(TOut) == 2) +
                                                    (FUSED ? NWX * 27 * MB * 16 * 4 : 0));
    auto nbr_at = [&](int k, int r, int r_end) -> int {
        const int rc = r < r_end ? r : r_end - 1;
        const int kr = PAIR ? 2 * k + qk : k, KR = PAIR ? KVOL : K;
        const int kc = kr < KR ? kr : KR - 1;
        if constexpr (FUSED) return kr < KR ? fstrip[kc * SR + (r - frow0)] : -1;
        const int v = nbr[(FNP_ABLATE & 32) ? (size_t)((kc * 64 + rc) & 0x3fff) : (size_t)kc * nbr_stride + rc];
        return (r < r_end && kr < KR) ? v : -1;
    };
    // same load, but the value is NOT touched here: the validity select happens where the entry is
    // consumed (two rounds later).  Any arithmetic on a freshly loaded entry makes the compiler wait
    // for it — and, VMEM returning in order, for every gather and weight load in flight.
    auto nbr_raw = [&](int k, int r, int r_end) -> int {
        const int rc = r < r_end ? r : r_end - 1;
        const int kr = PAIR ? 2 * k + qk : k, KR = PAIR ? KVOL : K;
        const int kc = kr < KR ? kr : KR - 1;
        if (FNP_ABLATE & 512) return rc + kc - 13;   // (probe: no rulebook loads at all)
        if constexpr (FUSED) return fstrip[kc * SR + (r - frow0)];
        return nbr[(FNP_ABLATE & 32) ? (size_t)((kc * 64 + rc) & 0x3fff) : (size_t)kc * nbr_stride + rc];
    };
    // is the lane's offset of step k a real one (raw entries are validated where they are consumed)
    auto k_live = [&](int k) -> bool { return PAIR ? (2 * k + qk < KVOL) : (k < K); };

    // window path: LDS byte address of (row id, this lane's chunk of MFMA step 0) when the row is in
    // the window [wlo, wlo + WROWS), else the zeros; step ks reads at (address ^ ks*64).  The global
    // gather of the same fragment is then suppressed (out-of-range offset), so every fragment is
    // the OR of an LDS read and a buffer load of which at most one is non-zero.
    auto win_off = [&](int id, int wlo) -> unsigned {
        if (FNP_ABLATE & 64)   // timing probe: the entry is taken as a pre-computed LDS address
            return (unsigned)Cfg::LDS_BYTES + ((((unsigned)id << 6) & 0x3fc0u) ^ ((unsigned)q << 4));
        const unsigned d = (unsigned)(id - wlo);
        return d < (unsigned)WROWS ? (unsigned)Cfg::LDS_BYTES + ((d * CH + ((unsigned)q ^ win_sw(d))) << 4) : WIN_ZERO;
    };
    auto row_off_w = [&](int id, int wlo) -> unsigned {
        const unsigned d = (unsigned)(id - wlo);
        return ((FNP_ABLATE & 1) || id < 0 || d < (unsigned)WROWS) ? 0x80000000u
                                                                    : (unsigned)id * (unsigned)(CIN * 2) + (unsigned)q * 16u;
    };
    auto win_read = [&](unsigned off) -> u32x4 {
        if (FNP_ABLATE & 16) return u32x4{off, 0u, 0u, 0u};
        return *reinterpret_cast<const u32x4 *>(fnp_smem + off);
    };

    // Work split: the n rows are cut into gridDim.x contiguous ranges of (almost) equal numbers of
    // 16-row blocks, so every workgroup finishes at about the same time whatever n is.  Range r
    // goes to the workgroups of one XCD in contiguous runs (blocks b and b + 8 share an XCD): each
    // XCD's L2 then serves 1/8 of the feature map.  Placement only affects speed, never results.
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per = G >> 3, rem = G & 7;
    const int range = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;  // bijective
    int row_begin, row_end;
    int xslots = 1;   // SORTED: slots of this workgroup's XCD group
    if constexpr (SORTED) {
        const XcdRows xr = fnp_xcd_rows(n, G, xcd);   // (row_begin .. row_end: the rows of the whole group)
        row_begin = xr.X0;
        row_end = xr.X1;
        xslots = xr.S;
    } else {
        fnp_range_rows(n, range, G, row_begin, row_end);
    }
    if (row_begin >= row_end) return;  // before any barrier: safe early exit
    for (int c = tid; c < 2 * COUT; c += NWX * 64)   // (a barrier — weight staging or the first slab — lies before any epilogue)
        ss_lds[c] = scale ? (c < COUT ? scale[c] : shift[c - COUT]) : (c < COUT ? 1.f : 0.f);

#define FNP_LDS_POS(row, chunk) ((row) * CH + ((chunk) ^ (((row) >> SW) & (CH - 1))))
    // LDS addressing with compile-time immediates: the swizzle term of row nb*16 + l15 depends on
    // l15 only, and that of staging position tid + c*256 on tid only, so one VGPR per MFMA step
    // (fragment reads) and one per thread (staging writes) carry the lane part; nb, c, k are added
    // as constants.
    int aoff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) aoff[ks] = FNP_LDS_POS(l15, kvalid0 ? ks * 4 + q : (q & (CH - 1)));   // (idle lanes mirror the conflict-free pattern)
    const int st_pos0 = FNP_LDS_POS(tid / CH, tid % CH);
    static_assert(SLAB < NT || ((NT / CH) % (CH << SW) == 0), "staging swizzle must be periodic in NT chunks");
    if (WIN && tid < Cfg::WZERO / 16) reinterpret_cast<uint4 *>(fnp_smem + WIN_ZERO)[tid] = make_uint4(0u, 0u, 0u, 0u);
    if (ALLK) {
        // narrow layers: all K slabs resident in LDS for the lifetime of the workgroup
        for (int p = tid; p < K * SLAB; p += NT) {
            const int kk = p / SLAB, r = p % SLAB;
            if constexpr (PAIR) {
                // image row of output channel co: [W_2kk (2 chunks) | W_2kk+1 (2 chunks)], zeros past the last offset
                const int co = r / CH, c = r % CH, kr = 2 * kk + (c >> 1);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (kr < KVOL) v = reinterpret_cast<const uint4 *>(w)[((size_t)kr * COUT + co) * 2 + (c & 1)];
                wl[kk * SLAB + FNP_LDS_POS(co, c)] = v;
            } else {
                wl[kk * SLAB + FNP_LDS_POS(r / CH, r % CH)] = reinterpret_cast<const uint4 *>(w)[p];
            }
        }
        __syncthreads();
    }

    // One tile = MBT 16-site blocks per wave (4 * MBT per workgroup), accumulators in registers for the
    // sweep over the K offsets.  The body is instantiated for MBT = MBT (full tiles) and for the smaller
    // block counts of a range's last, partial tile: a workgroup whose range ends with t blocks runs
    // that tile with ceil(t / 4) blocks per wave instead of MBT, so the matrix work of the tail is
    // proportional to its rows (every wave of the workgroup still passes the same barriers).
#ifdef FNP_MFMA_STAMP
    unsigned long long ms_acc[4] = {0, 0, 0, 0}, ms_prev;
    FNP_MS_NOW(ms_prev);
#endif
    bool first_tile = true;
    auto run_tile = [&](auto mbt_tag, const int tile_base) __attribute__((always_inline)) {
        constexpr int MBT = decltype(mbt_tag)::value;
        const int row0 = tile_base + wave * (MBT * 16);
        // SORTED: the offsets this tile sweeps.  Every wave forms the union of the tile's block masks itself (no LDS, no
        // barrier: the value is the same in all of them); lane l then holds the l-th live offset, read back with readlane.
        int Kt = K;           // offsets swept by this tile
        int kl = 0;           // SORTED: lane l: the l-th live offset (lanes >= Kt: the last one)
        int prow[MBT];        // SORTED: row of this lane's position in block mb
        if constexpr (SORTED) {
            const int nbt = min(NW * MBT, (row_end - tile_base + 15) >> 4);
            unsigned m = lane < nbt ? srb.blockmask[(tile_base >> 4) + lane] : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m |= (unsigned)__shfl_xor((int)m, d);
            m = (unsigned)__builtin_amdgcn_readfirstlane((int)m) & 0x7ffffffu;
            if (m == 0u) m = 1u << 13;   // (cannot happen for a tile with rows: a row is its own neighbour at offset 13)
            Kt = __popc(m);
            int cnt = 0;
#pragma unroll
            for (int pbit = 0; pbit < 27; ++pbit) {
                if ((m >> pbit) & 1u) {   // (uniform)
                    if (lane >= cnt) kl = pbit;
                    ++cnt;
                }
            }
#pragma unroll
            for (int mb = 0; mb < MBT; ++mb) prow[mb] = srb.perm[min(row0 + mb * 16 + l15, row_end - 1)];
        }
        // offset of sweep step i (clamped to the last step); rulebook entries of (step, block mb) for this lane's row
        auto koff = [&](int i) -> int {
            if constexpr (SORTED) return __builtin_amdgcn_readlane(kl, i < Kt ? i : Kt - 1);
            else return i;
        };
        auto ent_raw = [&](int k, int mb) -> int {
            if constexpr (SORTED) {
                if (FNP_ABLATE & 512) return prow[mb] + koff(k) - 13;   // (probe: no rulebook loads)
                // (round 5, measured and removed: a copy of the table in PROCESSING order — one 64-byte read per block and offset instead of 16
                //  scattered words through perm — takes 3.4 % off this launch and costs its own permute pass, 61 us per 128-scene step for
                //  4 x 25 us: +0.6 % / -0.3 % end to end at 64 / 128 scenes, a wash)
                return nbr[(size_t)koff(k) * nbr_stride + prow[mb]];
            }
            else return nbr_raw(k, row0 + mb * 16 + l15, row_end);
        };
        auto ent_at = [&](int k, int mb) -> int {
            if constexpr (SORTED) {
                const int v = ent_raw(k, mb);
                return (row0 + mb * 16 + l15 < row_end && k < Kt) ? v : -1;
            } else return nbr_at(k, row0 + mb * 16 + l15, row_end);
        };
        auto live = [&](int k) -> bool {
            if constexpr (SORTED) return k < Kt;
            else return k_live(k);
        };
        auto wslab = [&](int i) -> const uint4 * {   // weight slab of sweep step i (past the end: the last step's, never used)
            return reinterpret_cast<const uint4 *>(w + (size_t)koff(i < Kt ? i : Kt - 1) * COUT * CIN);
        };
        if constexpr (FUSED) {
            // rulebook rows of this wave's tile: lane j resolves the 27 input cells of row row0 + j.  The strip is
            // written by lane j and read by every lane of the wave (and overwritten by the next tile): wave-level
            // fences + barriers on both sides pin the order of those LDS accesses whatever the compiler schedules.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            frow0 = row0;
            if (frb.ell_rec) {   // (uniform) entries from the compact rulebook
                if (lane < MBT * 16) {
                    const int r = row0 + lane;
                    uint4 ra = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu), rb4 = ra;
                    if (r < row_end) {
                        const uint4 *rp = reinterpret_cast<const uint4 *>(frb.ell_rec + (size_t)r * 8);
                        ra = rp[0];
                        rb4 = rp[1];
                    }
#pragma unroll
                    for (int k = 0; k < 27; ++k) fstrip[k * SR + lane] = -1;
                    while (true) {
                        const unsigned v[8] = {ra.x, ra.y, ra.z, ra.w, rb4.x, rb4.y, rb4.z, rb4.w};
                        unsigned link = 0xffffffffu;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const unsigned kk = v[j] >> 27;
                            if (v[j] == 0xffffffffu) continue;
                            if (kk < 27u) fstrip[kk * SR + lane] = (int)(v[j] & 0x7ffffffu);
                            else link = v[j] & 0x7ffffffu;   // (slot 7 of a full record: the next record of the chain)
                        }
                        if (link == 0xffffffffu) break;
                        const uint4 *rp = reinterpret_cast<const uint4 *>(frb.ell_rec + ((size_t)frb.ell_cap + link) * 8);
                        ra = rp[0];
                        rb4 = rp[1];
                    }
                }
            } else if (lane < MBT * 16) {
                const int r = row0 + lane;
                if (r < row_end) {
                    const int4 c = reinterpret_cast<const int4 *>(frb.out_coords)[r];
                    nbr_row<3, 3, 3>(frb.gi, c.x, c.y * frb.s[0] - frb.p[0], c.z * frb.s[1] - frb.p[1], c.w * frb.s[2] - frb.p[2],
                                     fstrip + lane, SR);
                } else {
#pragma unroll
                    for (int k = 0; k < 27; ++k) fstrip[k * SR + lane] = -1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // feature window of this tile: WROWS consecutive input rows around the tile's own rows.  With
        // rows in rank-grid order ~96 % of a tile's neighbours lie in it, each fetched once instead
        // of once per (site, offset) pair that references it.
        const int wlo = max(0, tile_base - WH);
        if constexpr (WIN) {
            constexpr int NST = WROWS * CH / NT;
            static_assert(WROWS * CH % NT == 0, "window staging");
            if (ALLK && !first_tile) __syncthreads();  // every wave is done with the previous window
            u32x4 st[NST];
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const unsigned p = (unsigned)tid + j * (unsigned)NT;   // chunk p of the window, in memory order: coalesced
                st[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (unsigned)wlo * (unsigned)(CIN * 2) + p * 16u, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const unsigned p = (unsigned)tid + j * (unsigned)NT, d = p / CH, c = p % CH;
                *reinterpret_cast<u32x4 *>(win + ((d * CH + (c ^ win_sw(d))) << 4)) = st[j];
            }
            if (ALLK) __syncthreads();  // (double-buffered layers: the slab-0 barrier below)
        }
        f32x4 acc[NB][MBT];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MBT; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // prologue: fragments of offsets 0..PFK-1, rulebook indices of offsets PFK..2*PFK-1
        bf16x8 xb[PFK][KS][MBT];
        int rawq[PFK][MBT];   // raw rulebook entries of the offsets PFK..2*PFK-1 ahead (gathers of this round)
        int rawr[PFK][MBT];   // ... of the offsets 2*PFK..3*PFK-1 ahead (gathers of the next round)
        unsigned loff[WIN ? PFK : 1][WIN ? MBT : 1];          // window addresses of the fragments in xb
        u32x4 xl[WIN ? XLB : 1][WIN ? KS : 1][WIN ? MBT : 1];  // window reads, XLB offsets ahead
#pragma unroll
        for (int u = 0; u < PFK; ++u)
#pragma unroll
            for (int mb = 0; mb < MBT; ++mb) {
                const int id0 = ent_at(u, mb);
                const unsigned ro = WIN ? row_off_w(id0, wlo) : row_off(id0);
                if (WIN) loff[u][mb] = win_off(id0, wlo);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) xb[u][ks][mb] = gather(ro, ks);
                rawq[u][mb] = ent_raw(PFK + u, mb);
                rawr[u][mb] = ent_raw(2 * PFK + u, mb);
            }
        // L2 line touch (ranked rows): the input rows just above this tile — the next tile's own rows, first
        // reached by the upper-neighbour offsets of this one — are requested one dword per 128-byte line now,
        // so the gathers that reach them later in the sweep hit L2 instead of each paying a fabric round trip
        // (VMEM returns in order: one late row holds back every younger gather of the wave, and the
        // per-offset barrier passes that wait on to the whole workgroup).  The values are never used.
        constexpr int NPF = (FNP_PF & 1) && !WIN && CIN == COUT ? (ROWS_PER_WG * CIN * 2 / 128 + NT - 1) / NT : 0;
        unsigned pfv[NPF > 0 ? NPF : 1];
        if constexpr (NPF > 0) {
            const unsigned pbase = ((hints & FNP_HINT_ROWS_RANKED) ? (unsigned)(tile_base + NW * MBT * 16) * (unsigned)(CIN * 2) : 0x80000000u);
#pragma unroll
            for (int j = 0; j < NPF; ++j)
                pfv[j] = __builtin_amdgcn_raw_buffer_load_b32(xrsrc, pbase + ((unsigned)tid + j * (unsigned)NT) * 128u, 0, 0);
        }
        if (!ALLK) {
            if (!(FNP_ABLATE & 2)) {
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int p = tid + j * NT;
                    if (SLAB % NT == 0 || p < SLAB) wl[st_pos0 + j * NT] = wslab(0)[p];
                    if (WPAIR) wl[SLAB + st_pos0 + j * NT] = wslab(1)[p];
                }
            }
            __syncthreads();
        }
        if constexpr (WIN) {
#pragma unroll
            for (int j = 0; j < XLB; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int mb = 0; mb < MBT; ++mb) xl[j][ks][mb] = win_read(loff[j][mb] ^ (unsigned)(ks << 6));
        }
        // (named scalars, not an array: a conditionally written array lands in scratch memory)
        uint4 wcur0 = make_uint4(0u, 0u, 0u, 0u), wcur1 = make_uint4(0u, 0u, 0u, 0u);
        if (WDEEP && !(FNP_ABLATE & 2)) {
            const uint4 *w1 = wslab(1);
            wcur0 = w1[tid < SLAB ? tid : 0];
            if (NCH > 1) wcur1 = w1[tid + NT];
        }

        uint4 wd0 = make_uint4(0u, 0u, 0u, 0u), wd1 = wd0, wd2 = wd0, wd3 = wd0;   // (named: see wcur0)
        uint4 we0 = wd0, we1 = wd0, we2 = wd0, we3 = wd0;                          // (WDN == 2: the second chunk of a step)
        if (WD4 && !(FNP_ABLATE & 2)) {
            const uint4 *w1 = wslab(WPAIR ? 2 : 1);
            wd0 = w1[tid]; wd1 = w1[tid + WDN * NT];
            if constexpr (KS > 2) { wd2 = w1[tid + 2 * WDN * NT]; wd3 = w1[tid + 3 * WDN * NT]; }
            if constexpr (WDN == 2) {
                we0 = w1[tid + NT]; we1 = w1[tid + 3 * NT];
                if constexpr (KS > 2) { we2 = w1[tid + 5 * NT]; we3 = w1[tid + 7 * NT]; }
            }
        }
        FNP_MS(2);
#if FNP_SWEEP_PRIO
        __builtin_amdgcn_s_setprio(FNP_SWEEP_PRIO);   // (probe: a sweeping workgroup's waves ahead of those of a workgroup in its prologue / epilogue)
#endif
        for (int k0 = 0; k0 < Kt; k0 += PFK) {
#pragma unroll
            for (int u = 0; u < PFK; ++u) {
                const int k = k0 + u;
                if (k >= Kt) break;  // wave-uniform
                u32x4 xl_nx[WIN ? KS : 1][WIN ? MBT : 1];
                const uint4 *wk = wl + (ALLK ? k : WPAIR ? (k & 3) : (k & 1)) * SLAB;
                const uint4 *wsrc = wslab(k + 1);
                // rulebook entries for offset k + 3*PFK: requested FIRST in the round, so that they are
                // older than this round's gathers (VMEM returns in order: a young index load in
                // front of the next round's first MFMA would stall it)
                int rawn[MBT];
#pragma unroll
                for (int mb = 0; mb < MBT; ++mb) rawn[mb] = ent_raw(k + 3 * PFK, mb);
                unsigned lnew[WIN ? MBT : 1];  // window addresses of offset k + PFK
                if constexpr (WIN) {
#pragma unroll
                    for (int mb = 0; mb < MBT; ++mb) {
                        const bool ok = live(k + PFK) && (row0 + mb * 16 + l15 < row_end);
                        lnew[mb] = win_off(ok ? rawq[u][mb] : -1, wlo);
                    }
                }
                static_assert(WST <= 2, "weight staging registers");
                uint4 wreg0 = make_uint4(0u, 0u, 0u, 0u), wreg1 = make_uint4(0u, 0u, 0u, 0u);  // (named: see wcur0)
                uint4 wnext0 = make_uint4(0u, 0u, 0u, 0u), wnext1 = make_uint4(0u, 0u, 0u, 0u);
                if (WDEEP && !(FNP_ABLATE & 2)) {
                    const uint4 *w2 = wslab(k + 2);
                    wnext0 = w2[tid < SLAB ? tid : 0];
                    if (NCH > 1) wnext1 = w2[tid + NT];
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (WD4 && !(FNP_ABLATE & 2)) {
                        uint4 &wd = ks == 0 ? wd0 : ks == 1 ? wd1 : ks == 2 ? wd2 : wd3;
                        // chunk ks of W_{k+1} (WPAIR: of W_{k+2}, into the ring slot nobody has read since the last barrier)
                        wl[(WPAIR ? ((k + 2) & 3) : ((k + 1) & 1)) * SLAB + st_pos0 + ks * WDN * NT] = wd;
                        constexpr int AH = WPAIR ? 3 : 2;
                        const uint4 *w2 = wslab(k + AH);
                        wd = w2[tid + ks * WDN * NT];                                         // chunk ks of W_{k+2} (WPAIR: W_{k+3})
                        if constexpr (WDN == 2) {
                            uint4 &we = ks == 0 ? we0 : ks == 1 ? we1 : ks == 2 ? we2 : we3;
                            wl[((k + 1) & 1) * SLAB + st_pos0 + (ks * 2 + 1) * NT] = we;
                            we = w2[tid + (ks * 2 + 1) * NT];
                        }
                    }
                    // (1) previous step's weight chunks -> other LDS buffer; (2) request this step's
                    if (!ALLK && !WDEEP && !WD4 && !(FNP_ABLATE & 2)) {
                        if (ks > 0) {
#pragma unroll
                            for (int j = 0; j < WST; ++j) {
                                const int c = (ks - 1) * WST + j, p = tid + c * NT;
                                if (c < NCH && (SLAB % NT == 0 || p < SLAB))
                                    wl[((k + 1) & 1) * SLAB + st_pos0 + c * NT] = (j == 0 ? wreg0 : wreg1);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < WST; ++j) {
                            const int c = ks * WST + j, p = tid + c * NT;
                            if (c < NCH && (SLAB % NT == 0 || p < SLAB)) (j == 0 ? wreg0 : wreg1) = wsrc[p];
                        }
                    }
                    // (3) matrix blocks of this step on the fragments requested PFK offsets ago
#pragma unroll
                    for (int h = 0; h < NB; h += NBH) {
                        bf16x8 wa[NBH];
#pragma unroll
                        for (int j = 0; j < NBH; ++j) {
                            // (CIN == 16: lanes q >= 2 read chunk 0; their x fragment is zero)
                            const uint4 t = wk[aoff[ks] + (h + j) * 16 * CH];
                            wa[j] = *reinterpret_cast<const bf16x8 *>(&t);
                        }
                        // (3b) window fragments of offset k + XLB into the registers this step frees
                        if constexpr (WIN) {
                            if (h + NBH >= NB) {
#pragma unroll
                                for (int mb = 0; mb < MBT; ++mb) {
                                    const unsigned lo = XLB == PFK ? lnew[mb] : loff[(u + XLB) % PFK][mb];
                                    xl_nx[ks][mb] = win_read(lo ^ (unsigned)(ks << 6));
                                }
                            }
                        }
#pragma unroll
                        for (int mb = 0; mb < MBT; ++mb) {
                            bf16x8 xv = xb[u][ks][mb];
                            if constexpr (WIN) {
                                const u32x4 t = *reinterpret_cast<const u32x4 *>(&xv) | xl[u % XLB][ks][mb];
                                xv = *reinterpret_cast<const bf16x8 *>(&t);
                            }
#pragma unroll
                            for (int j = 0; j < NBH; ++j) {
                                if ((FNP_ABLATE & 4)) {
                                    asm volatile("" ::"v"(wa[j]), "v"(xv));
                                } else {
                                    acc[h + j][mb] = mfma16(wa[j], xv, acc[h + j][mb]);
                                }
                            }
                        }
                    }
                    if constexpr (WIN) {
#pragma unroll
                        for (int mb = 0; mb < MBT; ++mb) xl[u % XLB][ks][mb] = xl_nx[ks][mb];
                    }
                    // (4) the registers are free again: request the fragments of offset k + PFK (the
                    //     rulebook entry was loaded two rounds ago; validity is decided here)
                    if constexpr (GPAIR) {
                        // (FNP_GPAIR: the two 64-byte halves of a 128-byte line requested back to back — behind the odd step — so that
                        //  the second half finds the line in L1 instead of fetching it from L2 again)
                        if (ks & 1) {
#pragma unroll
                            for (int mb = 0; mb < MBT; ++mb) {
                                const bool ok = live(k + PFK) && (row0 + mb * 16 + l15 < row_end);
                                const unsigned ro = row_off(ok ? rawq[u][mb] : -1);
                                xb[u][ks - 1][mb] = gather(ro, ks - 1);
                                xb[u][ks][mb] = gather(ro, ks);
                            }
                        }
                    } else {
#pragma unroll
                    for (int mb = 0; mb < MBT; ++mb) {
                        const bool ok = live(k + PFK) && (row0 + mb * 16 + l15 < row_end);
                        xb[u][ks][mb] = gather(WIN ? row_off_w(ok ? rawq[u][mb] : -1, wlo) : row_off(ok ? rawq[u][mb] : -1), ks);
                    }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the steps in program order
                }
                // rulebook entries run two rounds ahead of their gathers: rotate
#pragma unroll
                for (int mb = 0; mb < MBT; ++mb) {
                    rawq[u][mb] = rawr[u][mb];
                    rawr[u][mb] = rawn[mb];
                    if (WIN) loff[u][mb] = lnew[mb];
                }
                if (!ALLK) {
                    if (WDEEP && !(FNP_ABLATE & 2)) {
                        if (SLAB >= NT || tid < SLAB) wl[((k + 1) & 1) * SLAB + st_pos0] = wcur0;
                        if (NCH > 1) wl[((k + 1) & 1) * SLAB + st_pos0 + NT] = wcur1;
                        wcur0 = wnext0;
                        wcur1 = wnext1;
                    } else if (!WD4 && !(FNP_ABLATE & 2)) {
#pragma unroll
                        for (int j = 0; j < WST; ++j) {
                            const int c = (KS - 1) * WST + j, p = tid + c * NT;
                            if (c < NCH && (SLAB % NT == 0 || p < SLAB))
                                wl[((k + 1) & 1) * SLAB + st_pos0 + c * NT] = (j == 0 ? wreg0 : wreg1);
                        }
                    }
                    FNP_MS(0);
                    if (!WPAIR || (k & 1) || k == Kt - 1) __syncthreads();  // plain loads stay in flight across it; only the LDS writes are waited for
                    FNP_MS(1);
                }
            }
        }
#undef FNP_LDS_POS
#if FNP_SWEEP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if constexpr (NPF > 0) {
#pragma unroll
            for (int j = 0; j < NPF; ++j) asm volatile("" ::"v"(pfv[j]));
        }

        // epilogue: lane holds out[site = row0 + mb*16 + l15][c0 .. c0+3], c0 = nb*16 + q*4
        // (window kernel: measured slower with either strip placement — 194 -> 200 / 214 us — and keeps the narrow form)
        constexpr bool WIDE = Cfg::WIDE && sizeof(TOut) == 2 && !WIN;
        static_assert(!SORTED || WIDE || sizeof(TOut) == 4, "sorted sweep: the wide and the f32 epilogue address rows through perm");
        if constexpr (WIDE) {
            constexpr int EH = Cfg::epi_sites(WIN);   // sites per strip pass
            constexpr int LPR = COUT / 8;        // 16-byte chunks (lanes) per row
            constexpr int SPI = 64 / LPR;        // sites per wave-wide 16-byte access
            constexpr int NRD = EH / SPI;        // accesses per pass
            constexpr int ES = Cfg::ESTRIDE;
            static_assert(LPR <= 16 && EH % SPI == 0 && NRD >= 1, "wide epilogue shape");
            unsigned char *const eb = fnp_smem + Cfg::lds_bytes(NW, MB, WIN) + wave * (EH * ES);
            const int wsite = lane / LPR, wchunk = lane % LPR;
            // the residual rows of ALL the wave's blocks are requested before the first one is used (one memory round trip for
            // the tile instead of one per 16-site block: the sweep's registers are free here)
            u32x4 rs_all[MBT][16 / EH][NRD];
            // SORTED: the rows behind the positions this lane reads the residual of and stores (one dependent load per tile)
            int orow[SORTED ? MBT : 1][SORTED ? 16 / EH : 1][SORTED ? NRD : 1];
            if constexpr (SORTED) {
#pragma unroll
                for (int mb = 0; mb < MBT; ++mb)
#pragma unroll
                    for (int h = 0; h < 16 / EH; ++h)
#pragma unroll
                        for (int i = 0; i < NRD; ++i) {
                            const int r = row0 + mb * 16 + h * EH + i * SPI + wsite;
                            orow[mb][h][i] = srb.perm[r < row_end ? r : row_end - 1];
                        }
            }
            if (residual && !(FNP_ABLATE & 256)) {
#pragma unroll
                for (int mb = 0; mb < MBT; ++mb)
#pragma unroll
                    for (int h = 0; h < 16 / EH; ++h)
#pragma unroll
                        for (int i = 0; i < NRD; ++i) {
                            const int r = row0 + mb * 16 + h * EH + i * SPI + wsite;
                            int ro = r;
                            if constexpr (SORTED) ro = orow[mb][h][i];
                            rs_all[mb][h][i] = u32x4{0u, 0u, 0u, 0u};
                            if (r < row_end) rs_all[mb][h][i] = *reinterpret_cast<const u32x4 *>(residual + (size_t)ro * COUT + wchunk * 8);
                        }
            }
#pragma unroll
            for (int mb = 0; mb < MBT; ++mb) {
#pragma unroll
                for (int h = 0; h < 16 / EH; ++h) {
                    const int rb = row0 + mb * 16 + h * EH;
                    const bool mine = EH == 16 || (l15 / EH) == h;   // this lane's site is in the pass
                    if (residual && !(FNP_ABLATE & 256)) {
#pragma unroll
                        for (int i = 0; i < NRD; ++i)
                            *reinterpret_cast<u32x4 *>(eb + (i * SPI + wsite) * ES + wchunk * 16) = rs_all[mb][h][i];
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int c0 = nb * 16 + q * 4;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[nb][mb][j];
                        if (scale) {
                            const float4 s4 = *reinterpret_cast<const float4 *>(ss_lds + c0);
                            const float4 h4 = *reinterpret_cast<const float4 *>(ss_lds + COUT + c0);
                            v[0] = v[0] * s4.x + h4.x; v[1] = v[1] * s4.y + h4.y; v[2] = v[2] * s4.z + h4.z; v[3] = v[3] * s4.w + h4.w;
                        }
                        if (mine) {
                            bf16x4 *slot = reinterpret_cast<bf16x4 *>(eb + (l15 % EH) * ES + c0 * 2);
                            if (residual && !(FNP_ABLATE & 256)) {
                                const bf16x4 t = *slot;
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = v[j] + (float)t[j];
                            }
                            if (relu) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];
                            }
                            *slot = bf16x4{(TAct)v[0], (TAct)v[1], (TAct)v[2], (TAct)v[3]};
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NRD; ++i) {
                        const int r = rb + i * SPI + wsite;
                        int ro = r;
                        if constexpr (SORTED) ro = orow[mb][h][i];
                        const u32x4 t = *reinterpret_cast<const u32x4 *>(eb + (i * SPI + wsite) * ES + wchunk * 16);
                        if (r < row_end && !(FNP_ABLATE & 128)) {
#if FNP_NT_STORE
                            __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(y + (size_t)ro * COUT + wchunk * 8));
#else
                            *reinterpret_cast<u32x4 *>(y + (size_t)ro * COUT + wchunk * 8) = t;
#endif
                        }
                    }
                }
            }
        } else if constexpr (FNP_SWAP_EPI && (WIN || Cfg::WPAIR) && sizeof(TOut) == 2 && NB % 2 == 0) {
            // window kernel (its LDS is busy with the other resident workgroup's window reads: the strip form
            // measured slower): the 8-byte pieces of channel blocks k and k + 1 are exchanged between the lane
            // rows q, q ^ 1 of a site (v_permlane16_swap: odd rows of the first register <-> even rows of the
            // second), after which a lane holds 16 contiguous bytes — half the residual-load and store
            // instructions, 64 instead of 32 contiguous bytes per site.  Same arithmetic, same single rounding.
            const int poff = (q & 1) * 32 + (q >> 1) * 16;   // byte offset of this lane's 16 bytes inside a 64-byte pair
#pragma unroll
            for (int mb = 0; mb < MBT; ++mb) {
                const int r = row0 + mb * 16 + l15;
                const bool live = r < row_end;
#pragma unroll
                for (int kp = 0; kp < NB / 2; ++kp) {
                    uint2 ra = make_uint2(0u, 0u), rb = make_uint2(0u, 0u);
                    if (residual && !(FNP_ABLATE & 256)) {
                        uint4 rv = make_uint4(0u, 0u, 0u, 0u);
                        if (live) rv = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(residual) + (size_t)r * (COUT * 2) + kp * 64 + poff);
                        auto t0 = __builtin_amdgcn_permlane16_swap(rv.x, rv.z, false, false);
                        auto t1 = __builtin_amdgcn_permlane16_swap(rv.y, rv.w, false, false);
                        ra = make_uint2(t0[0], t1[0]);   // block 2 kp,     channels q*4 .. q*4+3
                        rb = make_uint2(t0[1], t1[1]);   // block 2 kp + 1
                    }
                    uint2 o[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int nb = 2 * kp + h, c0 = nb * 16 + q * 4;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[nb][mb][j];
                        if (scale) {
                            const float4 s4 = *reinterpret_cast<const float4 *>(ss_lds + c0);
                            const float4 h4 = *reinterpret_cast<const float4 *>(ss_lds + COUT + c0);
                            v[0] = v[0] * s4.x + h4.x; v[1] = v[1] * s4.y + h4.y; v[2] = v[2] * s4.z + h4.z; v[3] = v[3] * s4.w + h4.w;
                        }
                        if (residual && !(FNP_ABLATE & 256)) {
                            const uint2 rr = h ? rb : ra;
                            const bf16x4 t = *reinterpret_cast<const bf16x4 *>(&rr);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] + (float)t[j];
                        }
                        if (relu) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];
                        }
                        const bf16x4 ob = {(TAct)v[0], (TAct)v[1], (TAct)v[2], (TAct)v[3]};
                        o[h] = *reinterpret_cast<const uint2 *>(&ob);
                    }
                    auto t0 = __builtin_amdgcn_permlane16_swap(o[0].x, o[1].x, false, false);
                    auto t1 = __builtin_amdgcn_permlane16_swap(o[0].y, o[1].y, false, false);
                    if (live && !(FNP_ABLATE & 128))
                        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(y) + (size_t)r * (COUT * 2) + kp * 64 + poff) =
                            make_uint4(t0[0], t1[0], t0[1], t1[1]);
                }
            }
        } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int c0 = nb * 16 + q * 4;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
            if (scale) {
                const float4 s4 = *reinterpret_cast<const float4 *>(ss_lds + c0);
                const float4 h4 = *reinterpret_cast<const float4 *>(ss_lds + COUT + c0);
                sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
                sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
            }
#pragma unroll
            for (int mb = 0; mb < MBT; ++mb) {
                const int rpos = row0 + mb * 16 + l15;
                if (rpos >= row_end) continue;
                int r = rpos;
                if constexpr (SORTED) r = srb.perm[rpos];   // (class-sorted sweep: the row behind this position)
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = scale ? acc[nb][mb][j] * sc[j] + sh[j] : acc[nb][mb][j];
                if (residual && !(FNP_ABLATE & 256)) {
                    const TOut *rp = residual + (size_t)r * COUT + c0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] + to_f32(rp[j]);
                }
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];
                }
                TOut *yp = y + (size_t)r * COUT + c0;
                if (FNP_ABLATE & 128) { if (v[0] == 12345.678f) yp[0] = from_f32<TOut>(v[1] + v[2] + v[3]); continue; }   // (probe: no stores)
                if constexpr (sizeof(TOut) == 2) {
                    bf16x4 o = {(TAct)v[0], (TAct)v[1], (TAct)v[2], (TAct)v[3]};
                    *reinterpret_cast<bf16x4 *>(yp) = o;
                } else {
                    if (so.hi) {   // (SplitOut: the launcher passed relu = 0; so.relu follows the addend)
                        if (so.addend) {
                            const bf16x4 tv = *reinterpret_cast<const bf16x4 *>(reinterpret_cast<const TAct *>(so.addend) + (size_t)r * COUT + c0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] + (float)tv[j];
                        }
                        if (so.relu) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];
                        }
                        bf16x4 h, l;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            h[j] = (TAct)v[j];
                            l[j] = (TAct)(v[j] - (float)h[j]);
                        }
                        *reinterpret_cast<bf16x4 *>(reinterpret_cast<TAct *>(so.hi) + (size_t)r * COUT + c0) = h;
                        *reinterpret_cast<bf16x4 *>(reinterpret_cast<TAct *>(so.lo) + (size_t)r * COUT + c0) = l;
                    }
                    if (y) *reinterpret_cast<float4 *>(yp) = make_float4(v[0], v[1], v[2], v[3]);   // (null with SplitOut: only the split is wanted)
                }
            }
        }
        }
        FNP_MS(3);
        first_tile = false;
    };
    int full, tper, tail_base, tstride = ROWS_PER_WG, tfirst = row_begin;
    if constexpr (SORTED) {
        // rounds of the XCD group (see XcdRows): this slot's tile of every full round, then its share of the partial round
        const int round_rows = xslots * ROWS_PER_WG;
        full = (row_end - row_begin) / round_rows;
        tstride = round_rows;
        tfirst = row_begin + slot * ROWS_PER_WG;
        const int left0 = row_begin + full * round_rows;
        tper = fnp_tail_blocks(row_end - left0, xslots, NW);
        tail_base = left0 + slot * (NW * tper * 16);
        if (tail_base >= row_end) tper = 0;
    } else {
        const int nblk_wg = (row_end - row_begin + 15) >> 4;
        full = nblk_wg / (NW * MB);
        tper = (nblk_wg - full * NW * MB + NW - 1) / NW;  // blocks per wave in the partial tile (0 = none)
        tail_base = row_begin + full * ROWS_PER_WG;
    }
    if constexpr (SORTED) {
        // the tiles of a round are sorted by class, and a class has its own number of live offsets: slot s takes tile
        // (s + j ROT) mod S of round j, so that every workgroup meets every class in turn (with tile s in every round the
        // slots of the middle class sweep all 27 offsets in every round and set the launch's length)
        const int rot = (xslots * 3 + 4) >> 3;
        for (int t = 0; t < full; ++t) run_tile(std::integral_constant<int, MB>{}, row_begin + t * tstride + ((slot + t * rot) % xslots) * ROWS_PER_WG);
    } else {
        for (int t = 0; t < full; ++t) run_tile(std::integral_constant<int, MB>{}, tfirst + t * tstride);
    }
    if (tper == MB) run_tile(std::integral_constant<int, MB>{}, tail_base);
    if constexpr (MB > 1) { if (tper == 1) run_tile(std::integral_constant<int, 1>{}, tail_base); }
    if constexpr (MB > 2) { if (tper == 2) run_tile(std::integral_constant<int, 2>{}, tail_base); }
    if constexpr (MB > 3) { if (tper == 3) run_tile(std::integral_constant<int, 3>{}, tail_base); }
#ifdef FNP_MFMA_STAMP
    if (CIN == 128 && COUT == 128 && KVOL == 27 && lane == 0)
        for (int ph = 0; ph < 4; ++ph) atomicAdd(&g_mfma_stamps[ph], ms_acc[ph]);
#endif
}

// grid_only: write the workgroup count the launch would use for this capacity and return without launching (the class-sort
// pass sorts the rows of exactly those ranges)
template <int CIN, int COUT, int KVOL, bool WIN, typename TOut, bool FUSED = false, typename TAct = __bf16, bool SORTED = false, int NWO = 0>
int launch_mfma_k(const void *x, int x_bytes, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap,
                  void *y, const float *scale, const float *shift, const void *residual, int relu, int hints, hipStream_t s,
                  const FusedRb *frb_in = nullptr, const SortedRb *srb_in = nullptr, int *grid_only = nullptr, const SplitOut *so_in = nullptr) {
    // 16-site blocks per wave: 4 (64 sites); 3 for 128 output channels (accumulators = COUT/16 * MB * 4
    // registers; 4 spills heavily, 3 spills ~16 registers outside the offset loop and measured 13 %
    // faster than 2 on MI355X: fewer weight-slab sweeps per site); 2 for the 16 -> 16 layers
#ifndef FNP_MB128
#define FNP_MB128 3
#endif
    // (16 -> 32, 8 % of the pairs present: 2 blocks at 4 waves/SIMD measured 9 % faster than 4 blocks at 3)
#ifndef FNP_MB3232
#define FNP_MB3232 2
#endif
#ifndef FNP_MB_SMALL
#define FNP_MB_SMALL 2   // (blocks per wave of the four-wave small-input form)
#endif
#ifndef FNP_MB6464
#define FNP_MB6464 2
#endif
    constexpr int NWX = NwOf<CIN, COUT, NWO>::value;
    constexpr int MB = NWO ? FNP_MB_SMALL : COUT >= 128 ? FNP_MB128 : (CIN == 16 && COUT <= 32) ? 2 : (CIN == 32 && COUT == 32) ? FNP_MB3232 : (CIN == 64 && COUT == 64) ? FNP_MB6464 : 4;
    using Cfg = MfmaCfg<CIN, COUT, KVOL>;
    auto kern = spconv_mfma_kernel<CIN, COUT, MB, KVOL, WIN, TOut, FUSED, TAct, SORTED, NWO>;
    constexpr int lds = Cfg::lds_bytes(NWX, MB, WIN) + Cfg::epi_bytes(NWX, WIN, sizeof(TOut) == 2) +
                        (FUSED ? NWX * 27 * MB * 16 * 4 : 0) + COUT * 8;   // (+ BatchNorm scale / shift)
    FusedRb frb{};
    if (FUSED) {
        if (!frb_in) return FNP_ERR_ARG;
        frb = *frb_in;
    }
    SortedRb srb{nullptr, nullptr};
    if (SORTED && !grid_only) {
        if (!srb_in || !srb_in->perm || !srb_in->blockmask) return FNP_ERR_ARG;
        srb = *srb_in;
    }
    // workgroups a CU holds: by the register budget (launch bounds), and for the fused-rulebook form by its larger LDS
    constexpr int wg_regs = MfmaOcc<CIN, COUT>::WAVES * 4 / NWX;
    constexpr int wg_per_cu = lds * wg_regs > 160 * 1024 ? 160 * 1024 / lds : wg_regs;
    static_assert(wg_per_cu >= 1 && lds * wg_per_cu <= 160 * 1024, "LDS budget of the resident workgroups");
    if (lds > 64 * 1024 && !grid_only) {
        static bool raised = false;  // (idempotent; a race only repeats the call)
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return FNP_ERR_HIP;
            raised = true;
        }
    }
    const int tiles = fnp_divup(cap, NWX * MB * 16);
    // persistent grid: two workgroups per CU are resident (register / LDS budget of the wide
    // layers); the narrow ALLK layers stage all weights once per workgroup, so keep them few too.
    // The kernel splits the rows evenly over whatever grid it gets.
    const int resident = 256 * wg_per_cu;
    // Small inputs (a single scene: the reference's extraction runs batch size 1): with fewer full tiles than resident
    // workgroups most CUs would idle while a few sweep MB blocks per wave — split finer instead, down to one 16-site
    // block per wave (the kernel runs such a range as a partial tile).  The split never changes results.
    const int fine = fnp_divup(cap, NWX * 16);
#ifdef FNP_NO_FINE   // (development switch)
    const int grid = tiles < resident ? tiles : resident;
#else
    const int grid = tiles >= resident ? resident : (fine < resident ? fine : resident);
#endif
    if (grid_only) {
        *grid_only = grid;
        return FNP_OK;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWX * 64), lds, s, (const TAct *)x, x_bytes, (const TAct *)w,
                       nbr, nbr_stride, K, n_out, cap, (TOut *)y, scale, shift, (const TOut *)residual, relu, hints, frb, srb,
                       so_in ? *so_in : SplitOut{nullptr, nullptr, nullptr, 0});
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// The window variant is used when the caller states that neighbour row ids lie close to the output
// row ids (FNP_HINT_ROWS_RANKED: both tensors in rank-grid order).  Measured on MI355X (B = 16):
// 64 -> 64 channels 132 -> 120 us; 32 -> 32 channels 78 -> 115 us (the per-fragment address
// arithmetic and OR of the dual-source operand outweigh the saved gathers at 8 MFMAs per offset),
// so only the 64-channel layers take it.
#ifndef FNP_WIN32
#define FNP_WIN32 0
#endif
#ifndef FNP_WIN64
#define FNP_WIN64 1
#endif
#ifndef FNP_SMALL128
#define FNP_SMALL128 1
#endif
#define FNP_MB128_DEFAULT 3
template <int CIN, int COUT> struct HasWindow { static constexpr bool value = (FNP_WIN64 && CIN == 64 && COUT == 64) || (FNP_WIN32 && CIN == 32 && COUT == 32); };

template <int CIN, int COUT, typename TOut, typename TAct>
int launch_mfma(const void *x, int x_bytes, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out,
                int cap, void *y, const float *scale, const float *shift, const void *residual, int relu, int hints,
                hipStream_t s, const SplitOut *so = nullptr) {
    if (K == 27) {
        if constexpr (HasWindow<CIN, COUT>::value) {
            if (hints & FNP_HINT_ROWS_RANKED)
                return launch_mfma_k<CIN, COUT, 27, true, TOut, false, TAct>(x, x_bytes, w, nbr, nbr_stride, K, n_out, cap, y, scale,
                                                                shift, residual, relu, hints, s, nullptr, nullptr, nullptr, so);
        }
        if constexpr (CIN == 128 && COUT == 128 && sizeof(TOut) == 2) {
            // fewer rows than the persistent grid has 384-row tiles: the four-wave form (see NwOf)
            if (cap < 256 * MfmaWg<CIN, COUT>::NW * FNP_MB128_DEFAULT * 16 && FNP_SMALL128)
                return launch_mfma_k<CIN, COUT, 27, false, TOut, false, TAct, false, 4>(x, x_bytes, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift,
                                                                                        residual, relu, hints, s);
        }
        return launch_mfma_k<CIN, COUT, 27, false, TOut, false, TAct>(x, x_bytes, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift,
                                                         residual, relu, hints, s, nullptr, nullptr, nullptr, so);
    }
    return launch_mfma_k<CIN, COUT, 0, false, TOut, false, TAct>(x, x_bytes, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift, residual,
                                                    relu, hints, s, nullptr, nullptr, nullptr, so);
}

template <typename TIn, typename TOut>
int launch_valu(const void *x, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap, void *y,
                const float *scale, const float *shift, const void *residual, int relu, int Cin, int Cout,
                hipStream_t s) {
    const int grid = fnp_grid_for((long long)cap * Cout, 256, 256 * 16);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(spconv_valu_kernel<TIn, TOut>), dim3(grid), dim3(256), 0, s, (const TIn *)x,
                       (const TIn *)w, nbr, nbr_stride, K, n_out, cap, (TOut *)y, scale, shift, (const TOut *)residual,
                       relu, Cin, Cout);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// ------------------------------------------------------------------------------------------
// Class sort (see SortedRb).  One workgroup per range of the convolution's grid: pass 1 forms the 27-bit neighbour mask of
// every row of the range from the table (kept in `rowmask`) and counts the four z classes — no neighbour in either
// adjacent plane, above only (offsets 18..26), both, below only (offsets 0..8): in that order a tile of one class, or of
// two adjacent ones, still drops a whole plane of offsets; pass 2 gives every row its position — class by class, rows of
// a class in their original (rank) order, so the gathers keep their locality — writes perm[position] = row and ORs the
// row's mask into blockmask[position / 16].  Deterministic: positions depend on the masks only.
// ------------------------------------------------------------------------------------------
constexpr int kSortThreads = 1024, kSortQ = 16;   // a thread places at most kSortQ consecutive rows of a round
__device__ __forceinline__ int fnp_zclass(unsigned m) {
    const bool lo = (m & 0x1ffu) != 0u, hi = (m & (0x1ffu << 18)) != 0u;
    return lo ? (hi ? 2 : 3) : (hi ? 1 : 0);
}
// per-row masks from the table (for a rulebook that was built without them)
__global__ __launch_bounds__(256) void rowmask_kernel(const int *__restrict__ nbr, int nbr_stride, const int *__restrict__ n_out, int cap,
                                                      unsigned *__restrict__ rowmask) {
    const int n = min(*n_out, cap);
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
        unsigned m = 0u;
#pragma unroll
        for (int k = 0; k < 27; ++k) m |= (nbr[(size_t)k * nbr_stride + r] >= 0 ? 1u : 0u) << k;
        rowmask[r] = m;
    }
}
// grid (rounds, 8 XCD groups): one workgroup per round; thread t owns rows [t q, (t + 1) q) of the round (q <= kSortQ), counts
// their classes, a workgroup-wide exclusive scan of the four counts (packed 4 x 16 bits: a round has < 65,536 rows) gives
// each thread its first position per class, and it places its rows in order.  Stable: rows of a class keep their order.
__global__ __launch_bounds__(kSortThreads) void classsort_place_kernel(const unsigned *__restrict__ rowmask, const int *__restrict__ n_out, int cap,
                                                                        int G, int tile_rows, int NW, int *__restrict__ perm,
                                                                        unsigned *__restrict__ blockmask) {
    constexpr int NWV = kSortThreads / 64;
    __shared__ unsigned long long wtot[NWV];
    const int n = min(*n_out, cap), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const XcdRows xr = fnp_xcd_rows(n, G, blockIdx.y);
    if (xr.S == 0 || xr.X0 >= xr.X1) return;   // (whole workgroup)
    const int round_rows = xr.S * tile_rows, full = (xr.X1 - xr.X0) / round_rows;
    for (int i = blockIdx.x; i <= full; i += gridDim.x) {   // (whole workgroups stay in the loop)
        const int b = xr.X0 + i * round_rows, e = min(xr.X1, b + round_rows);
        if (b >= e) break;
        const int q = (e - b + kSortThreads - 1) / kSortThreads, r0 = b + tid * q;
        for (int j = (b >> 4) + tid; j < ((e + 15) >> 4); j += kSortThreads) blockmask[j] = 0u;
        unsigned m[kSortQ];
        unsigned long long cnt = 0ull;
#pragma unroll
        for (int u = 0; u < kSortQ; ++u) {
            m[u] = 0u;
            if (u < q && r0 + u < e) {
                m[u] = rowmask[r0 + u];
                cnt += 1ull << (16 * fnp_zclass(m[u]));
            }
        }
        unsigned long long v = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(v, d);
            if (lane >= d) v += t;
        }
        if (lane == 63) wtot[wave] = v;
        __syncthreads();   // (also orders the zeroing of blockmask before the atomics below)
        unsigned long long woff = 0ull, grand = 0ull;
#pragma unroll
        for (int wv = 0; wv < NWV; ++wv) {
            const unsigned long long t = wtot[wv];
            if (wv < wave) woff += t;
            grand += t;
        }
        const unsigned long long excl = v - cnt + woff;
        int run[4];
        const int t0 = (int)(grand & 0xffffu), t1 = (int)((grand >> 16) & 0xffffu), t2 = (int)((grand >> 32) & 0xffffu);
        run[0] = (int)(excl & 0xffffu);
        run[1] = t0 + (int)((excl >> 16) & 0xffffu);
        run[2] = t0 + t1 + (int)((excl >> 32) & 0xffffu);
        run[3] = t0 + t1 + t2 + (int)((excl >> 48) & 0xffffu);
#pragma unroll
        for (int u = 0; u < kSortQ; ++u) {
            if (u < q && r0 + u < e) {
                const int c = fnp_zclass(m[u]);
                const int pos = b + (c == 0 ? run[0]++ : c == 1 ? run[1]++ : c == 2 ? run[2]++ : run[3]++);
                perm[pos] = r0 + u;
                atomicOr(&blockmask[pos >> 4], m[u]);
            }
        }
        __syncthreads();   // (wtot is rewritten by the next round of this workgroup)
    }
}

template <typename TAct, typename TOut>
int dispatch_16(const void *x, long long n_in, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap,
                  void *y, const float *scale, const float *shift, const void *residual, int relu, int hints, int Cin,
                  int Cout, hipStream_t s, const SplitOut *so = nullptr) {
    const long long xb = n_in * Cin * 2;
    const bool fits = xb > 0 && xb < 0x7fffffffll;   // 32-bit buffer offsets of the MFMA path
#define FNP_CASE(CI, CO)                                                                                       \
    if (fits && Cin == CI && Cout == CO)                                                                       \
        return launch_mfma<CI, CO, TOut, TAct>(x, (int)xb, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift, residual,\
                                         relu, hints, s, so);
    FNP_CASE(16, 16)
    FNP_CASE(16, 32)
    FNP_CASE(32, 32)
    FNP_CASE(32, 64)
    FNP_CASE(64, 64)
    FNP_CASE(64, 128)
    FNP_CASE(128, 128)
    // transposed channel pairs: the data gradients of the three channel-doubling convolutions
    FNP_CASE(32, 16)
    FNP_CASE(64, 32)
    FNP_CASE(128, 64)
#undef FNP_CASE
    if (so) return FNP_ERR_ARG;   // (the split outputs are the matrix kernel's)
    return launch_valu<TAct, TOut>(x, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift, residual, relu, Cin, Cout, s);
}

}  // namespace

#ifdef FNP_MFMA_STAMP
extern "C" int fnp_debug_mfma_stamps(unsigned long long *out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mfma_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return FNP_ERR_HIP;
    unsigned long long zero[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_stamps), zero, sizeof(zero)) != hipSuccess) return FNP_ERR_HIP;
    return FNP_OK;
}
#endif

int fnp_spconv_forward_f32_mfma(const void *feat_in, long long n_in_rows, const void *weight, const int *nbr, int nbr_stride,
                                int K, const int *n_out, int cap_out, void *feat_out, const float *scale, const float *shift,
                                const void *residual, int relu, int wperm, int Cin, int Cout, hipStream_t s);   // spconv_f32.hip

extern "C" int fnp_spconv_forward(const void *feat_in, int in_dtype, int n_in_rows, const void *weight, const int *nbr,
                                  int nbr_stride, int K, const int *n_out, int cap_out, void *feat_out, int out_dtype,
                                  const float *scale, const float *shift, const void *residual, int relu, int hints,
                                  int Cin, int Cout, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!feat_in || !weight || !nbr || !n_out || !feat_out || K <= 0 || Cin <= 0 || Cout <= 0 || cap_out <= 0 ||
        nbr_stride < cap_out || n_in_rows <= 0)
        return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr)) return FNP_ERR_ARG;
    if (in_dtype == FNP_F32 && Cout == 16 && Cin <= 8 && (size_t)K * Cin * 16 * 4 <= 60000) {
        if (out_dtype == FNP_F32)
            return launch_first<float>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                       residual, relu, Cin, s);
        if (out_dtype == FNP_BF16)
            return launch_first<__bf16>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                        residual, relu, Cin, s);
        if (out_dtype == FNP_F16)
            return launch_first<_Float16>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                          residual, relu, Cin, s);
        return FNP_ERR_ARG;
    }
    if (in_dtype == FNP_F32) {
        if (out_dtype == FNP_F32 && !(hints & FNP_HINT_VALU)) {
            // f32 on the matrix pipe (v_mfma_f32_16x16x4_f32: the same fmaf chain, bit for bit); shapes it does not
            // cover fall through to the thread-per-element chain
            const int rc = fnp_spconv_forward_f32_mfma(feat_in, n_in_rows, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out,
                                                       scale, shift, residual, relu, (hints & FNP_HINT_W_PERMUTED) != 0, Cin, Cout, s);
            if (rc != FNP_ERR_ARG) return rc;
        }
        if (hints & FNP_HINT_W_PERMUTED) return FNP_ERR_ARG;   // the other f32 kernels read the plain layout
        if (out_dtype == FNP_F32)
            return launch_valu<float, float>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                             residual, relu, Cin, Cout, s);
        if (out_dtype == FNP_BF16)
            return launch_valu<float, __bf16>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale,
                                              shift, residual, relu, Cin, Cout, s);
        if (out_dtype == FNP_F16)
            return launch_valu<float, _Float16>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale,
                                                shift, residual, relu, Cin, Cout, s);
        return FNP_ERR_ARG;
    }
    if (in_dtype == FNP_BF16) {
        if (out_dtype == FNP_BF16)
            return dispatch_16<__bf16, __bf16>(feat_in, n_in_rows, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                               residual, relu, hints, Cin, Cout, s);
        if (out_dtype == FNP_F32)
            return dispatch_16<__bf16, float>(feat_in, n_in_rows, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                              residual, relu, hints, Cin, Cout, s);
        return FNP_ERR_ARG;
    }
    if (in_dtype == FNP_F16) {   // fp16 features / weights, fp32 accumulate: the reference's AMP mode
        if (out_dtype == FNP_F16)
            return dispatch_16<_Float16, _Float16>(feat_in, n_in_rows, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale,
                                                   shift, residual, relu, hints, Cin, Cout, s);
        if (out_dtype == FNP_F32)
            return dispatch_16<_Float16, float>(feat_in, n_in_rows, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                                residual, relu, hints, Cin, Cout, s);
        return FNP_ERR_ARG;
    }
    return FNP_ERR_ARG;
}

extern "C" int fnp_spconv_forward_split(const void *feat_in, int in_dtype, int n_in_rows, const void *weight, const int *nbr, int nbr_stride,
                                        int K, const int *n_out, int cap_out, float *feat_out, const float *scale, const float *shift,
                                        const float *residual, const void *addend, int relu, int hints, int Cin, int Cout, void *out_hi,
                                        void *out_lo, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!feat_in || !weight || !nbr || !n_out || !out_hi || !out_lo || K <= 0 || Cin <= 0 || Cout <= 0 || cap_out <= 0 ||
        nbr_stride < cap_out || n_in_rows <= 0 || (scale == nullptr) != (shift == nullptr))   // (feat_out may be null: only the split is written)
        return FNP_ERR_ARG;
    if ((((uintptr_t)out_hi | (uintptr_t)out_lo | (uintptr_t)addend) & 7) || ((uintptr_t)feat_out & 15)) return FNP_ERR_ARG;
    const SplitOut so{addend, out_hi, out_lo, relu};
    if (in_dtype == FNP_BF16)
        return dispatch_16<__bf16, float>(feat_in, n_in_rows, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift, residual, 0, hints,
                                          Cin, Cout, s, &so);
    if (in_dtype == FNP_F16)
        return dispatch_16<_Float16, float>(feat_in, n_in_rows, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift, residual, 0, hints,
                                            Cin, Cout, s, &so);
    return FNP_ERR_ARG;
}

// Strided 3x3x3 convolution with its rulebook computed inside the kernel (see FusedRb): for a layer whose rulebook
// has no other user.  bf16 in/out, channel pairs 16->32, 32->64, 64->128; anything else returns FNP_ERR_ARG and the
// caller takes the table path (fnp_rulebook_strided with nbr + fnp_spconv_forward).
extern "C" int fnp_spconv_forward_strided(const void *feat_in, int in_dtype, int n_in_rows, const void *weight,
                                          const fnp_rankgrid *in_grid, const fnp_conv_geom *geom, const int *out_coords,
                                          const int *n_out, int cap_out, void *feat_out, int out_dtype, const float *scale,
                                          const float *shift, int relu, int Cin, int Cout, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!feat_in || !weight || !geom || !out_coords || !n_out || !feat_out || cap_out <= 0 || n_in_rows <= 0 ||
        !fnp_rg_valid(in_grid))
        return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr)) return FNP_ERR_ARG;
    if ((in_dtype != FNP_BF16 && in_dtype != FNP_F16) || out_dtype != in_dtype) return FNP_ERR_ARG;
    for (int d = 0; d < 3; ++d)
        if (geom->ksize[d] != 3 || geom->stride[d] <= 0 || geom->padding[d] < 0) return FNP_ERR_ARG;
    if (in_grid->D != geom->in_shape[0] || in_grid->H != geom->in_shape[1] || in_grid->W != geom->in_shape[2]) return FNP_ERR_ARG;
    const long long xb = (long long)n_in_rows * Cin * 2;
    if (xb <= 0 || xb >= 0x7fffffffll) return FNP_ERR_ARG;
    FusedRb frb;
    frb.gi = fnp_rg_view(in_grid);
    for (int d = 0; d < 3; ++d) {
        frb.s[d] = geom->stride[d];
        frb.p[d] = geom->padding[d];
    }
    frb.out_coords = out_coords;
    frb.ell_rec = nullptr;
    frb.ell_cap = 0;
#define FNP_FCASE(CI, CO)                                                                                                   \
    if (Cin == CI && Cout == CO) {                                                                                          \
        if (in_dtype == FNP_F16)                                                                                            \
            return launch_mfma_k<CI, CO, 27, false, _Float16, true, _Float16>(feat_in, (int)xb, weight, nullptr, cap_out, 27, n_out, \
                                                                              cap_out, feat_out, scale, shift, nullptr, relu, 0, s, &frb); \
        return launch_mfma_k<CI, CO, 27, false, __bf16, true>(feat_in, (int)xb, weight, nullptr, cap_out, 27, n_out, cap_out, feat_out, \
                                                              scale, shift, nullptr, relu, 0, s, &frb);                     \
    }
    FNP_FCASE(16, 32)
    FNP_FCASE(32, 64)
    FNP_FCASE(64, 128)
#undef FNP_FCASE
    return FNP_ERR_ARG;
}

// The 16-channel layers on the COMPACT rulebook, on the matrix pipe (see FusedRb): fnp_spconv_forward_ell's arguments for
// Cin = 16; the values of fnp_spconv_forward on the (27, cap) table of the same rows, bit for bit.
extern "C" int fnp_spconv_forward_ell_mfma(const void *feat_in, int in_dtype, int n_in_rows, const void *weight, const void *records, int cap_rows,
                                           int pool_records, const int *n_out, void *feat_out, int out_dtype, const float *scale, const float *shift,
                                           const void *residual, int relu, int Cin, int Cout, fnp_stream_t stream) {
    if (!feat_in || !weight || !records || !n_out || !feat_out || cap_rows <= 0 || pool_records < 0 || n_in_rows <= 0) return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr) || ((uintptr_t)records & 15)) return FNP_ERR_ARG;
    if ((in_dtype != FNP_BF16 && in_dtype != FNP_F16) || out_dtype != in_dtype || Cin != 16 || (Cout != 16 && Cout != 32)) return FNP_ERR_ARG;
    const long long xb = (long long)n_in_rows * Cin * 2;
    if (xb >= 0x7fffffffll || (long long)cap_rows + pool_records >= (1ll << 26) || n_in_rows >= (1 << 27)) return FNP_ERR_ARG;
    FusedRb frb{};
    frb.ell_rec = (const unsigned *)records;
    frb.ell_cap = cap_rows;
    hipStream_t s = (hipStream_t)stream;
#define FNP_ECASE(CO)                                                                                                                         \
    if (Cout == CO) {                                                                                                                         \
        if (in_dtype == FNP_F16)                                                                                                              \
            return launch_mfma_k<16, CO, 27, false, _Float16, true, _Float16>(feat_in, (int)xb, weight, nullptr, cap_rows, 27, n_out, cap_rows, \
                                                                              feat_out, scale, shift, residual, relu, 0, s, &frb);            \
        return launch_mfma_k<16, CO, 27, false, __bf16, true>(feat_in, (int)xb, weight, nullptr, cap_rows, 27, n_out, cap_rows, feat_out,     \
                                                              scale, shift, residual, relu, 0, s, &frb);                                      \
    }
    FNP_ECASE(16)
    FNP_ECASE(32)
#undef FNP_ECASE
    return FNP_ERR_ARG;
}


// ------------------------------------------------------------------------------------------
// Class-sorted sweep of the 128 -> 128 SubM layers (SortedRb).  fnp_rulebook_classsort restates the processing order of a
// 3x3x3 rulebook once per forward (its four convolutions share it); fnp_spconv_forward_sorted is fnp_spconv_forward for
// (Cin, Cout) = (128, 128), 16-bit features in and out, on that order.  Same values as fnp_spconv_forward.
// ------------------------------------------------------------------------------------------
extern "C" long long fnp_classsort_workspace_bytes(int cap_out) {
    if (cap_out <= 0) return 0;
    return (long long)cap_out * 4;   // the row masks, when the caller has none
}

extern "C" int fnp_rulebook_classsort(const int *nbr, int nbr_stride, int K, const unsigned *rowmask, const int *n_out, int cap_out, int Cin,
                                      int Cout, int *perm, unsigned *blockmask, void *workspace, long long workspace_bytes,
                                      fnp_stream_t stream) {
    if (!n_out || !perm || !blockmask || K != 27 || cap_out <= 0) return FNP_ERR_ARG;
    if (Cin != 128 || Cout != 128) return FNP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!rowmask) {
        if (!nbr || nbr_stride < cap_out || !workspace) return FNP_ERR_ARG;
        if (workspace_bytes < fnp_classsort_workspace_bytes(cap_out)) return FNP_ERR_WORKSPACE;
        hipLaunchKernelGGL(rowmask_kernel, dim3(fnp_grid_for(cap_out, 256)), dim3(256), 0, s, nbr, nbr_stride, n_out, cap_out, (unsigned *)workspace);
        FNP_LAUNCH_CHECK();
        rowmask = (const unsigned *)workspace;
    }
    int grid = 0;
    const int rc = launch_mfma_k<128, 128, 27, false, __bf16, false, __bf16, true>(nullptr, 0, nullptr, nullptr, cap_out, 27, n_out, cap_out, nullptr, nullptr,
                                                                                    nullptr, nullptr, 0, 0, s, nullptr, nullptr, &grid);
    if (rc != FNP_OK) return rc;
    constexpr int NW = MfmaWg<128, 128>::NW, TILE = NW * FNP_MB128 * 16;
    const int slots = (grid >> 3) + ((grid & 7) ? 1 : 0);
    if ((long long)slots * TILE > (long long)kSortThreads * kSortQ) return FNP_ERR_ARG;   // (a round is placed by one workgroup)
    const int rounds = fnp_divup(fnp_divup(cap_out, 8) + 16 * (slots + 1), (long long)((grid >> 3) > 0 ? (grid >> 3) : 1) * TILE) + 1;
    hipLaunchKernelGGL(classsort_place_kernel, dim3(rounds < 1024 ? rounds : 1024, 8), dim3(kSortThreads), 0, s, rowmask, n_out, cap_out, grid, TILE, NW, perm,
                       blockmask);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_spconv_forward_sorted(const void *feat_in, int dtype, int n_in_rows, const void *weight, const int *nbr, int nbr_stride,
                                         const int *perm, const unsigned *blockmask, const int *n_out, int cap_out, void *feat_out,
                                         const float *scale, const float *shift, const void *residual, int relu, int Cin, int Cout,
                                         fnp_stream_t stream) {
    if (!feat_in || !weight || !nbr || !perm || !blockmask || !n_out || !feat_out || cap_out <= 0 || nbr_stride < cap_out || n_in_rows <= 0)
        return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr) || Cin != 128 || Cout != 128) return FNP_ERR_ARG;
    const long long xb = (long long)n_in_rows * Cin * 2;
    if (xb >= 0x7fffffffll) return FNP_ERR_ARG;
    const SortedRb srb{perm, blockmask};
    if (dtype == FNP_BF16)
        return launch_mfma_k<128, 128, 27, false, __bf16, false, __bf16, true>(feat_in, (int)xb, weight, nbr, nbr_stride, 27, n_out, cap_out, feat_out, scale,
                                                                                shift, residual, relu, 0, (hipStream_t)stream, nullptr, &srb);
    if (dtype == FNP_F16)
        return launch_mfma_k<128, 128, 27, false, _Float16, false, _Float16, true>(feat_in, (int)xb, weight, nbr, nbr_stride, 27, n_out, cap_out, feat_out,
                                                                                    scale, shift, residual, relu, 0, (hipStream_t)stream, nullptr, &srb);
    return FNP_ERR_ARG;
}

extern "C" int fnp_spconv_forward_sorted_split(const void *feat_in, int dtype, int n_in_rows, const void *weight, const int *nbr, int nbr_stride,
                                               const int *perm, const unsigned *blockmask, const int *n_out, int cap_out, float *feat_out,
                                               const float *scale, const float *shift, const float *residual, const void *addend, int relu,
                                               int Cin, int Cout, void *out_hi, void *out_lo, fnp_stream_t stream) {
    if (!feat_in || !weight || !nbr || !perm || !blockmask || !n_out || !out_hi || !out_lo || cap_out <= 0 || nbr_stride < cap_out || n_in_rows <= 0)
        return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr) || Cin != 128 || Cout != 128) return FNP_ERR_ARG;
    if ((((uintptr_t)out_hi | (uintptr_t)out_lo | (uintptr_t)addend) & 7) || ((uintptr_t)feat_out & 15)) return FNP_ERR_ARG;
    const long long xb = (long long)n_in_rows * Cin * 2;
    if (xb >= 0x7fffffffll) return FNP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    // perm / blockmask were made for the workgroup ranges of the 16-bit-output sweep (fnp_rulebook_classsort): this launch must cut
    // the rows the same way
    int g16 = 0, g32 = 0;
    launch_mfma_k<128, 128, 27, false, __bf16, false, __bf16, true>(nullptr, 0, nullptr, nullptr, cap_out, 27, n_out, cap_out, nullptr, nullptr, nullptr, nullptr,
                                                                    0, 0, s, nullptr, nullptr, &g16);
    launch_mfma_k<128, 128, 27, false, float, false, __bf16, true>(nullptr, 0, nullptr, nullptr, cap_out, 27, n_out, cap_out, nullptr, nullptr, nullptr, nullptr,
                                                                   0, 0, s, nullptr, nullptr, &g32);
    if (g16 != g32) return FNP_ERR_ARG;
    const SortedRb srb{perm, blockmask};
    const SplitOut so{addend, out_hi, out_lo, relu};
    if (dtype == FNP_BF16)
        return launch_mfma_k<128, 128, 27, false, float, false, __bf16, true>(feat_in, (int)xb, weight, nbr, nbr_stride, 27, n_out, cap_out, feat_out, scale, shift,
                                                                              residual, 0, 0, s, nullptr, &srb, nullptr, &so);
    if (dtype == FNP_F16)
        return launch_mfma_k<128, 128, 27, false, float, false, _Float16, true>(feat_in, (int)xb, weight, nbr, nbr_stride, 27, n_out, cap_out, feat_out, scale,
                                                                                shift, residual, 0, 0, s, nullptr, &srb, nullptr, &so);
    return FNP_ERR_ARG;
}
