// Sparse convolution forward in float32 on the matrix pipe: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate).
//
// The reference's default precision is fp32 and BASELINE.json's tolerance for it is 1e-4:
// this is the kernel behind `FNP_DTYPE: fp32`.  gfx950's f32 MFMA is bit-for-bit a k-ordered fmaf chain
// (MI355X_MICROARCH.md, Matrix cores): with the kernel offsets swept in ascending order and the input channels of an
// offset in ascending order, every output element is the SAME chain the CPU oracle (and spconv_valu_kernel) evaluates,
// so the results are bit-comparable — at 64 FLOP/clk/SIMD instead of one output element per thread.
//
//   D (16 out-channels x 16 sites) += A (16 out-channels x 4 in-channels: weights) * B (4 in-channels x 16 sites)
//   lane (l15 = lane & 15, q = lane >> 4):  a = W_k[co = nb*16 + l15][ci = 4s + q]   b = x[nbr[k][site l15]][ci = 4s + q]
//   and ends up with out[site l15][co = nb*16 + 4q .. 4q+3]: one 16-byte store per lane and channel block.
//
// The kernel is bound by the matrix pipe (1/16 of the bf16 rate), not by its gathers, so it is deliberately simple:
// no LDS, no barriers.  Weights and features are both fetched 16 bytes per lane (4 consecutive channels of the lane's
// row) for a chunk of 16 input channels and put into MFMA order — lane q needs channels 4s + q for the chunk's four
// steps s — by a 4x4 transpose across the four lane rows of a site in registers (two v_permlane32_swap + two
// v_permlane16_swap per four registers).  The fragments of the next chunk are requested before the MFMAs of the
// current one (64 MFMAs = 2,048 cycles per wave at 128 output channels), rulebook entries two offsets ahead.
// A 16-site block none of whose sites has a neighbour at an offset skips that offset's MFMAs (wave-uniform branch;
// 44-68 % of the (block, offset) pairs in the stage-1 layers): fmaf(0, w, acc) == acc, the chains are unchanged.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 4x4 transpose of (lane row q, register r) inside every 16-lane column group: out[r] of lane row q = in[q] of lane row r
__device__ __forceinline__ void transpose4(u32x4 &v) {
    auto s0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);   // rows 2,3 of x <-> rows 0,1 of z
    auto s1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
    auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s1[0], false, false);   // odd rows of the first <-> even rows of the second
    auto t1 = __builtin_amdgcn_permlane16_swap(s0[1], s1[1], false, false);
    v.x = t0[0];
    v.y = t0[1];
    v.z = t1[0];
    v.w = t1[1];
}

// waves per SIMD the register budget is held to: the narrow layers have little matrix work per rulebook entry and are
// bound by the latency of their index -> gather chain, so they run 4 waves per SIMD; the wide ones are bound by the
// matrix pipe and need the registers (2 waves per SIMD)
template <int COUT> struct F32Occ { static constexpr int WAVES = COUT <= 32 ? 4 : 2; };

// WPERM: the weight rows are stored with every group of 16 input channels transposed 4 x 4 (position 4q + r holds
// channel 4r + q; sparse.pack_weight(..., mfma_f32=True), FNP_HINT_W_PERMUTED): a lane's 16-byte load then already
// holds its channels of the chunk's four MFMA steps and the weights need no lane exchange.  Same products, same order.
// PERM (round 3): the rows of a workgroup's range are processed in an order sorted by neighbourhood class (perm[position] = row,
// fnp_rulebook_classsort_f32: no neighbour plane / above only / both / below only, rows of a class in their own order).  The
// kernel is bound by the matrix pipe and already skips the MFMAs of a (16-site block, offset) pair none of whose sites has a
// neighbour — which a block of one class has for a whole plane of offsets (a lidar surface is one or two cells thick after the
// strided layers: 0.99 of the pairs hold a neighbour in row order, 0.73-0.78 after the sort).  Every row still sums its own
// neighbours in ascending offset and channel order: the same fmaf chain, bit for bit.
template <int CIN, int COUT, int MB, bool WPERM, bool PERM = false>
__global__ __launch_bounds__(256, (F32Occ<COUT>::WAVES)) void spconv_mfma_f32_kernel(const float *__restrict__ x, int x_bytes,
                                                                 const float *__restrict__ w,
                                                                 const int *__restrict__ nbr, int nbr_stride, int K,
                                                                 const int *__restrict__ n_out, int cap,
                                                                 float *__restrict__ y, const float *__restrict__ scale,
                                                                 const float *__restrict__ shift,
                                                                 const float *__restrict__ residual, int relu,
                                                                 const int *__restrict__ perm = nullptr) {
    static_assert(CIN % 16 == 0 && COUT % 16 == 0, "channel counts must be multiples of 16");
    constexpr int NB = COUT / 16;    // 16-channel output blocks
    constexpr int NC = CIN / 16;     // 16-channel input chunks (4 MFMA steps each)
    constexpr int NW = 4, ROWS_PER_WAVE = MB * 16, ROWS_PER_WG = NW * ROWS_PER_WAVE;
    const int n = min(*n_out, cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, K * COUT * CIN * 4, 0x00020000);

    // contiguous, balanced, XCD-aware row ranges (same split as spconv_mfma_kernel)
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per = G >> 3, rem = G & 7;
    const int range = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    const long long nblk16 = (n + 15) >> 4;
    const int row_begin = (int)((nblk16 * range) / G) << 4;
    const int row_end = min(n, (int)((nblk16 * (range + 1)) / G) << 4);

    // byte offset of this lane's 16 bytes in weight row co = nb*16 + l15 of offset 0, chunk 0
    const unsigned woff0 = (unsigned)(l15 * CIN + q * 4) * 4u;

    // a range shorter than one workgroup tile (small inputs, finer grid) is cut evenly over the four waves
    const int nblk_wg = (row_end - row_begin + 15) >> 4;
    const bool small = nblk_wg < NW * MB;
    const int bpw = (nblk_wg + NW - 1) / NW;
    const int tile0 = row_begin + wave * (small ? bpw * 16 : ROWS_PER_WAVE);
    const int row_end_w = small ? min(row_end, tile0 + bpw * 16) : row_end;
    for (int tile = tile0; tile < row_end_w; tile += small ? (1 << 30) : ROWS_PER_WG) {
        f32x4 acc[NB][MB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        int prow[MB];   // PERM: the row behind this lane's position of block mb
        if constexpr (PERM) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int r = tile + mb * 16 + l15;
                prow[mb] = perm[r < row_end_w ? r : row_end_w - 1];
            }
        }
        auto entry = [&](int k, int mb) -> int {
            const int r = tile + mb * 16 + l15;
            int rc = r < row_end_w ? r : row_end_w - 1;
            if constexpr (PERM) rc = prow[mb];
            const int kc = k < K ? k : K - 1;
            const int v = nbr[(size_t)kc * nbr_stride + rc];
            return (r < row_end_w && k < K) ? v : -1;
        };
        auto row_off = [&](int id) -> unsigned { return id < 0 ? 0x80000000u : (unsigned)id * (unsigned)(CIN * 4) + (unsigned)q * 16u; };
        auto load_b = [&](unsigned roff, int c) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(xrsrc, roff + (unsigned)c * 64u, 0, 0); };
        auto load_a = [&](int k, int nb, int c, unsigned skip = 0u) -> u32x4 {
            return __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (woff0 + (unsigned)((k * COUT + nb * 16) * CIN + c * 16) * 4u) | skip, 0, 0);
        };

        int idx1[MB], idx2[MB];       // rulebook entries of the offsets k + 1 and k + 2
        unsigned roff[MB];            // feature row offsets of offset k
        bool pres[MB];                // some site of the block has a neighbour at offset k (wave-uniform)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int i0 = entry(0, mb);
            roff[mb] = row_off(i0);
            pres[mb] = __ballot(i0 >= 0) != 0ull;
            idx1[mb] = entry(1, mb);
            idx2[mb] = entry(2, mb);
        }
        u32x4 an[NB], bn[MB];         // fragments of the NEXT chunk, in memory order
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) an[nb] = load_a(0, nb, 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) bn[mb] = load_b(roff[mb], 0);

        for (int k = 0; k < K; ++k) {
            int idx3[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) idx3[mb] = entry(k + 3, mb);
            unsigned roff_nx[MB];
            bool pres_nx[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                roff_nx[mb] = row_off(idx1[mb]);
                pres_nx[mb] = __ballot(idx1[mb] >= 0) != 0ull;
            }
            // an offset none of the wave's rows has a neighbour at (wave-uniform; whole planes of offsets for a class-sorted range)
            // requests its weights beyond the end of the buffer: zeros that never leave the memory pipeline (round 3: the weight
            // stream out of L2 was the second limit of the 128-channel layers)
            bool any = false, any_nx = false;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                any = any || pres[mb];
                any_nx = any_nx || pres_nx[mb];
            }
            const unsigned wskip = any ? 0u : 0x80000000u, wskip_nx = any_nx ? 0u : 0x80000000u;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                u32x4 a[NB], b[MB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) a[nb] = an[nb];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) b[mb] = bn[mb];
                // request the next chunk: (k, c + 1), or (k + 1, 0) behind the last chunk of this offset
                if (c + 1 < NC) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) an[nb] = load_a(k, nb, c + 1, wskip);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) bn[mb] = load_b(roff[mb], c + 1);
                } else {
                    const int kn = k + 1 < K ? k + 1 : k;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) an[nb] = load_a(kn, nb, 0, wskip_nx);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) bn[mb] = load_b(roff_nx[mb], 0);
                }
                if (any) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        if (!WPERM) transpose4(a[nb]);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        if (!pres[mb]) continue;   // wave-uniform
                        transpose4(b[mb]);
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const float bv = __uint_as_float(b[mb][s]);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[nb][s]), bv, acc[nb][mb], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                roff[mb] = roff_nx[mb];
                pres[mb] = pres_nx[mb];
                idx1[mb] = idx2[mb];
                idx2[mb] = idx3[mb];
            }
        }

        // epilogue: lane holds out[site = tile + mb*16 + l15][c0 .. c0+3], c0 = nb*16 + q*4
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int c0 = nb * 16 + q * 4;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
            if (scale) {
                const float4 s4 = *reinterpret_cast<const float4 *>(scale + c0);
                const float4 h4 = *reinterpret_cast<const float4 *>(shift + c0);
                sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
                sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                int r = tile + mb * 16 + l15;
                if (r >= row_end_w) continue;
                if constexpr (PERM) r = prow[mb];
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = scale ? acc[nb][mb][j] * sc[j] + sh[j] : acc[nb][mb][j];
                if (residual) {
                    const float4 rv = *reinterpret_cast<const float4 *>(residual + (size_t)r * COUT + c0);
                    v[0] = v[0] + rv.x; v[1] = v[1] + rv.y; v[2] = v[2] + rv.z; v[3] = v[3] + rv.w;
                }
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];
                }
                *reinterpret_cast<float4 *>(y + (size_t)r * COUT + c0) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int CIN, int COUT>
int launch_f32(const void *x, long long x_bytes, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap,
               void *y, const float *scale, const float *shift, const void *residual, int relu, bool wperm, hipStream_t s,
               const int *perm = nullptr, int *grid_only = nullptr) {
    // sites per wave: 4 blocks up to 64 output channels (a weight fragment then serves 64 sites), 3 for 128
    // (accumulators = COUT/16 * MB * 4 registers; measured on MI355X at 64 scenes: 128 -> 128 4.54 -> 4.30 ms with 3
    // instead of 2, 64 -> 64 2.96 -> 2.68 ms with 4 instead of 2)
#ifndef FNP_F32_MB128
#define FNP_F32_MB128 3
#endif
#ifndef FNP_F32_MB64
#define FNP_F32_MB64 4
#endif
    constexpr int MB = COUT <= 32 ? 4 : COUT == 64 ? FNP_F32_MB64 : FNP_F32_MB128;
    const int tiles = fnp_divup(cap, 4 * MB * 16);
    const int resident = 256 * F32Occ<COUT>::WAVES;      // one 4-wave workgroup per CU and wave slot
    const int fine = fnp_divup(cap, 4 * 16);            // small inputs: down to one 16-site block per wave
    const int grid = tiles >= resident ? resident : (fine < resident ? fine : resident);
    if (grid_only) {
        *grid_only = grid;
        return FNP_OK;
    }
    if (perm) {
        if (wperm)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(spconv_mfma_f32_kernel<CIN, COUT, MB, true, true>), dim3(grid), dim3(256), 0, s,
                               (const float *)x, (int)x_bytes, (const float *)w, nbr, nbr_stride, K, n_out, cap, (float *)y, scale,
                               shift, (const float *)residual, relu, perm);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(spconv_mfma_f32_kernel<CIN, COUT, MB, false, true>), dim3(grid), dim3(256), 0, s,
                               (const float *)x, (int)x_bytes, (const float *)w, nbr, nbr_stride, K, n_out, cap, (float *)y, scale,
                               shift, (const float *)residual, relu, perm);
        FNP_LAUNCH_CHECK();
        return FNP_OK;
    }
    if (wperm)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spconv_mfma_f32_kernel<CIN, COUT, MB, true>), dim3(grid), dim3(256), 0, s,
                           (const float *)x, (int)x_bytes, (const float *)w, nbr, nbr_stride, K, n_out, cap, (float *)y, scale,
                           shift, (const float *)residual, relu);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spconv_mfma_f32_kernel<CIN, COUT, MB, false>), dim3(grid), dim3(256), 0, s,
                           (const float *)x, (int)x_bytes, (const float *)w, nbr, nbr_stride, K, n_out, cap, (float *)y, scale,
                           shift, (const float *)residual, relu);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// Class sort of the rows of every workgroup range of the f32 kernel's grid (one 1024-thread workgroup per range): thread t owns
// q consecutive rows of the range, counts their classes, an exclusive scan of the four counts over the workgroup gives its first
// position per class, and it places its rows in order (stable: rows of a class keep their order).  A range of more than
// 16,384 rows keeps its own order.
constexpr int kF32SortThreads = 1024, kF32SortQ = 16;
__device__ __forceinline__ int f32_zclass(unsigned m) {
    const bool lo = (m & 0x1ffu) != 0u, hi = (m & (0x1ffu << 18)) != 0u;
    return lo ? (hi ? 2 : 3) : (hi ? 1 : 0);
}
__global__ __launch_bounds__(kF32SortThreads) void f32_classsort_kernel(const unsigned *__restrict__ rowmask, const int *__restrict__ n_out, int cap,
                                                                        int *__restrict__ perm) {
    constexpr int NWV = kF32SortThreads / 64;
    __shared__ unsigned long long wtot[NWV];
    const int n = min(*n_out, cap), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = G >> 3, rem = G & 7;
    const int range = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    const long long nblk16 = (n + 15) >> 4;
    const int b = (int)((nblk16 * range) / G) << 4, e = min(n, (int)((nblk16 * (range + 1)) / G) << 4);
    if (b >= e) return;
    if (e - b > kF32SortThreads * kF32SortQ) {   // (uniform) too long for one workgroup: row order
        for (int r = b + tid; r < e; r += kF32SortThreads) perm[r] = r;
        return;
    }
    const int q = (e - b + kF32SortThreads - 1) / kF32SortThreads, r0 = b + tid * q;
    unsigned m[kF32SortQ];
    unsigned long long cnt = 0ull;
#pragma unroll
    for (int u = 0; u < kF32SortQ; ++u) {
        m[u] = 0u;
        if (u < q && r0 + u < e) {
            m[u] = rowmask[r0 + u];
            cnt += 1ull << (16 * f32_zclass(m[u]));
        }
    }
    unsigned long long v = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
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
    for (int u = 0; u < kF32SortQ; ++u) {
        if (u < q && r0 + u < e) {
            const int c = f32_zclass(m[u]);
            perm[b + (c == 0 ? run[0]++ : c == 1 ? run[1]++ : c == 2 ? run[2]++ : run[3]++)] = r0 + u;
        }
    }
}

}  // namespace

template <int CIN, int COUT> static int f32_grid(int cap, int *grid) {
    return launch_f32<CIN, COUT>(nullptr, 0, nullptr, nullptr, cap, 27, nullptr, cap, nullptr, nullptr, nullptr, nullptr, 0, false, nullptr, nullptr, grid);
}

// perm (cap_out) int32 for fnp_spconv_forward_f32_sorted of this (Cin, Cout): valid for the (rowmask, n_out, cap_out) it was made from
extern "C" int fnp_rulebook_classsort_f32(const unsigned *rowmask, const int *n_out, int cap_out, int Cin, int Cout, int *perm, fnp_stream_t stream) {
    if (!rowmask || !n_out || !perm || cap_out <= 0) return FNP_ERR_ARG;
    int grid = 0, rc = FNP_ERR_ARG;
#define FNP_GCASE(CI, CO) if (Cin == CI && Cout == CO) rc = f32_grid<CI, CO>(cap_out, &grid);
    FNP_GCASE(16, 16)
    FNP_GCASE(32, 32)
    FNP_GCASE(64, 64)
    FNP_GCASE(128, 128)
#undef FNP_GCASE
    if (rc != FNP_OK) return rc;
    hipLaunchKernelGGL(f32_classsort_kernel, dim3(grid), dim3(kF32SortThreads), 0, (hipStream_t)stream, rowmask, n_out, cap_out, perm);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

int fnp_spconv_forward_f32_mfma(const void *feat_in, long long n_in_rows, const void *weight, const int *nbr, int nbr_stride,
                                int K, const int *n_out, int cap_out, void *feat_out, const float *scale, const float *shift,
                                const void *residual, int relu, int wperm, int Cin, int Cout, hipStream_t s);

// fnp_spconv_forward for f32 3x3x3 SubM layers (16 / 32 / 64 / 128 channels in = out) on the class-sorted row order: same bits
extern "C" int fnp_spconv_forward_f32_sorted(const void *feat_in, int n_in_rows, const void *weight, const int *nbr, int nbr_stride, const int *perm,
                                             const int *n_out, int cap_out, void *feat_out, const float *scale, const float *shift,
                                             const void *residual, int relu, int wperm, int Cin, int Cout, fnp_stream_t stream) {
    if (!feat_in || !weight || !nbr || !perm || !n_out || !feat_out || cap_out <= 0 || nbr_stride < cap_out || n_in_rows <= 0) return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr) || Cin != Cout) return FNP_ERR_ARG;
    const long long xb = (long long)n_in_rows * Cin * 4;
    if (xb >= 0x7fffffffll) return FNP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
#define FNP_SCASE(C)                                                                                                              \
    if (Cin == C)                                                                                                                 \
        return launch_f32<C, C>(feat_in, xb, weight, nbr, nbr_stride, 27, n_out, cap_out, feat_out, scale, shift, residual, relu, \
                                wperm != 0, s, perm);
    FNP_SCASE(16)
    FNP_SCASE(32)
    FNP_SCASE(64)
    FNP_SCASE(128)
#undef FNP_SCASE
    return FNP_ERR_ARG;
}

// f32 in / f32 out on the matrix pipe; FNP_ERR_ARG when the shape is not built (the caller falls back to the VALU chain)
int fnp_spconv_forward_f32_mfma(const void *feat_in, long long n_in_rows, const void *weight, const int *nbr, int nbr_stride,
                                int K, const int *n_out, int cap_out, void *feat_out, const float *scale, const float *shift,
                                const void *residual, int relu, int wperm, int Cin, int Cout, hipStream_t s) {
    const long long xb = n_in_rows * Cin * 4;
    if (xb <= 0 || xb >= 0x7fffffffll || (long long)K * Cin * Cout * 4 >= 0x7fffffffll) return FNP_ERR_ARG;
#define FNP_CASE(CI, CO)              \
    if (Cin == CI && Cout == CO)      \
        return launch_f32<CI, CO>(feat_in, xb, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift, residual, relu, wperm != 0, s);
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
    return FNP_ERR_ARG;
}
