// The SPARSE-NEIGHBOURHOOD layers of VoxelResBackBone8x on a compact rulebook (pcdet/models/backbones_3d/spconv_backbone.py:
// 193-210: conv_input 5 -> 16, the four 16 -> 16 SubM layers of conv1, the strided 16 -> 32 layer of conv2).
//
// At stage 1 a lidar voxel has 3.6 of its 27 neighbours, an output site of the first strided layer 2.1 of 27 inputs.  The
// (27, cap) int32 table spends 108 bytes per row on that — it is the largest stream of those layers (five convolutions read
// the stage-1 table) — and the MFMA kernel issues a gather and a matrix step per (16-row block, offset) whether the block
// has a neighbour there or not: it is bound by the NUMBER of gather instructions, 87 % of whose lanes are out of range.
//
// Compact rulebook ("ELL-8"): 32 bytes per row, written once by the rulebook kernel:
//     record = 8 x uint32;  slot = (k << 27) | input row   (k = kernel offset 0..26, ascending inside a row)
//                            0xFFFFFFFF = empty;  slot 7 may be (31 << 27) | e = LINK to extension record e
//   row r owns record r; a row with more than 8 neighbours (8 % of the stage-1 rows, 1 % of the 16 -> 32 rows) chains
//   extension records (7 entries + link, the last one up to 8) from a pool behind the cap row records; a workgroup takes
//   the pool records of its 256 rows with ONE atomicAdd (which records a row gets depends on timing; what the convolution
//   reads through them does not).  pool_used counts the records asked for: a caller that finds it above the pool's
//   capacity discards the result (chains were cut) and comes back with a larger pool.
// Convolution: output-stationary on the VALU — 4 (Cout 16) or 8 (Cout 32) lanes per row, 4 output channels per lane.  A wave
// loads the records of its rows (coalesced), then all neighbour rows of a record at once (only rows that exist), and sums
// sum_k W_k^T x in ascending k: 16-bit rows as v_dot2c_f32_bf16 / v_dot2_f32_f16 (two products per lane and cycle, f32
// accumulate), the f32 point features of conv_input as the oracle's fmaf chain (bit-exact).  Weights sit in LDS per
// (offset, lane group), swizzled so that the 16-byte reads of rows at different offsets spread over the banks.
// BatchNorm(eval) scale / shift, residual and ReLU in the epilogue, as in spconv_mfma_kernel.  No atomics in the sum:
// run-to-run identical.
#include "rankgrid.h"
#include <type_traits>

namespace {

constexpr int kThreads = 256;
constexpr unsigned kEmpty = 0xffffffffu, kLinkK = 31u, kIdxMask = (1u << 27) - 1u;

// ------------------------------------------------------------------------------------------ rulebook -> records
// strips: [27][64] per wave (nbr_row's layout).  Compacts the row's valid entries in place (ascending k), returns their count.
__device__ __forceinline__ int ell_compact(int *strip_lane) {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int r = strip_lane[k * 64];
        if (r >= 0) {
            strip_lane[cnt * 64] = (int)(((unsigned)k << 27) | (unsigned)r);   // (cnt <= k: never ahead of the read)
            ++cnt;
        }
    }
    return cnt;
}
__device__ __forceinline__ int ell_records(int cnt) {   // records of a row with cnt entries (7 + link ... , last up to 8)
    int m = 1;
    while (cnt > 8) {
        cnt -= 7;
        ++m;
    }
    return m;
}

// STRIDED: rows = output sites of a strided 3x3x3 convolution (coordinates o * s - p), else SubM (o - 1)
template <bool STRIDED>
__global__ __launch_bounds__(kThreads) void ell_build_kernel(const int *__restrict__ coords, const int *__restrict__ n_rows, int cap, RG g,
                                                             int sz, int sy, int sx, int pz, int py, int px, unsigned *__restrict__ rec,
                                                             int pool_cap, int *__restrict__ pool_used, unsigned char *__restrict__ perm,
                                                             int *__restrict__ nbr) {
    // nbr (optional): the (27, cap) int32 table of the same rows, written in the same pass from the same strips — for the
    // layers of this rulebook that stay on the matrix kernels (the 16 -> 16 layers) while conv_input reads the records
    __shared__ __attribute__((aligned(16))) int strips[kThreads / 64][27 * 64];
    __shared__ int wsum[kThreads / 64];
    __shared__ int pool_base;
    __shared__ int hist[32];
    const int n = min(*n_rows, cap), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *strip_lane = strips[wave] + lane;
    for (int base = blockIdx.x * kThreads; base < n; base += gridDim.x * kThreads) {   // (whole workgroups stay in the loop; NOT XCD-contiguous: +8 / +20 % at 64 scenes, round 5)
        const int o = base + tid;
        int cnt = 0;
        if (o < n) {
            const int4 c = reinterpret_cast<const int4 *>(coords)[o];
            if (STRIDED) nbr_row<3, 3, 3>(g, c.x, c.y * sz - pz, c.z * sy - py, c.w * sx - px, strip_lane, 64);
            else nbr_row<3, 3, 3>(g, c.x, c.y - 1, c.z - 1, c.w - 1, strip_lane, 64);
        }
        if (nbr) nbr_flush<27>(strips[wave], base + (tid & ~63), n, cap, nbr);   // (before the strips are compacted in place)
        if (o < n) cnt = ell_compact(strip_lane);
        const int ne = o < n ? ell_records(cnt) - 1 : 0;   // extension records of this row
        if (tid < 32) hist[tid] = 0;
        // exclusive prefix of ne over the workgroup, one atomicAdd for its total
        int inc = ne;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            const int s = wsum[w];
            if (w < wave) off += s;
            total += s;
        }
        if (tid == 0) pool_base = total ? atomicAdd(pool_used, total) : 0;
        // The convolution gives a group of lanes to a row and sweeps the slots of 16 (or 8) rows in lockstep: rows of one chunk
        // of 256 are handed out in the order of their entry counts (perm[position] = row inside the chunk), so that the rows
        // a wave holds end together.  (Which of two equal rows comes first depends on timing; no result does.)
        const int key = cnt < 31 ? cnt : 31;
        const int within = atomicAdd(&hist[key], 1);
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int q = 0; q < 28; ++q) before += q < key ? hist[q] : 0;
        perm[(size_t)base + before + within] = (unsigned char)tid;
        if (o < n) {
            int e = pool_base + off + inc - ne;   // first extension record of this row (pool numbering)
            unsigned *out = rec + (size_t)o * 8;
            int left = cnt, j0 = 0;
            while (true) {
                const bool last = left <= 8;
                const int take = last ? left : 7;
                unsigned v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = j < take ? (unsigned)strip_lane[(j0 + j) * 64] : kEmpty;
                bool cut = false;
                if (!last) {
                    if (e < pool_cap) v[7] = (kLinkK << 27) | (unsigned)e;
                    else cut = true;   // pool exhausted: the chain ends here (pool_used tells the caller)
                }
                reinterpret_cast<uint4 *>(out)[0] = make_uint4(v[0], v[1], v[2], v[3]);
                reinterpret_cast<uint4 *>(out)[1] = make_uint4(v[4], v[5], v[6], v[7]);
                if (last || cut) break;
                out = rec + ((size_t)cap + e) * 8;
                ++e;
                left -= take;
                j0 += take;
            }
        }
        __syncthreads();   // (strips, wsum and pool_base are rewritten by the next pass)
    }
}

// ------------------------------------------------------------------------------------------ convolution
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c, __bf16) {
    return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2 *>(&a), *reinterpret_cast<const bf16x2 *>(&b), c, false);
}
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c, _Float16) {
    return __builtin_amdgcn_fdot2(*reinterpret_cast<const f16x2 *>(&a), *reinterpret_cast<const f16x2 *>(&b), c, false);
}
template <typename T> __device__ __forceinline__ float ell_to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T ell_from_f32(float v) { return (T)v; }

// weight image in LDS: one block per (offset k, lane group g = 4 output channels), NQ 16-byte quads each:
//   16-bit rows (CIN channels): quad p = input channels 2p, 2p+1 of the group's 4 output channels (one dot2 operand each)
//   f32 rows    (CIN channels): quad ci = the group's 4 weights of input channel ci
// quad p of block b = g * 27 + k sits at (b * NQP + p) * 16 bytes with NQP = NQ + 1 when NQ is even (an odd block stride): the
// lanes of one wave instruction read quad p of blocks at different offsets k, and an odd stride walks k through all sixteen
// 16-byte slots of the 256-byte bank row (an even one would leave four); p stays an immediate offset of the read.
template <int CIN, bool F32IN> struct EllW {
    static constexpr int NQ = F32IN ? CIN : CIN / 2;
    static constexpr int NQP = NQ | 1;
};

template <int CIN, int COUT, typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void spconv_ell_kernel(const TIn *__restrict__ x, int x_bytes, const TIn *__restrict__ w,
                                                              const unsigned *__restrict__ rec, int rec_records,
                                                              const unsigned char *__restrict__ perm,
                                                              const int *__restrict__ n_out, int cap, TOut *__restrict__ y,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              const TOut *__restrict__ residual, int relu) {
    constexpr bool F32IN = std::is_same<TIn, float>::value;
    using W = EllW<CIN, F32IN>;
    constexpr int G = COUT / 4;              // lane groups per row (4 output channels per lane)
    constexpr int RPW = 64 / G;              // rows per wave
    constexpr int RPB = RPW * (kThreads / 64);
    constexpr int NQ = W::NQ, NQP = W::NQP;
    constexpr int ROWB = CIN * (int)sizeof(TIn);   // bytes of a feature row
    constexpr int XW = (ROWB + 3) / 4;             // dwords of a feature row
    static_assert(COUT % 4 == 0 && 64 % G == 0 && (F32IN || CIN % 8 == 0), "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char ell_smem[];
    u32x4 *wl = reinterpret_cast<u32x4 *>(ell_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane % G, rsub = lane / G;
    const int n = min(*n_out, cap);

    // weights (K, COUT, CIN) -> image
    for (int i = tid; i < 27 * G * NQ; i += kThreads) {
        const int b = i / NQ, p = i % NQ, gg = b / 27, k = b % 27;   // blocks ordered [lane group][offset]
        u32x4 v;
        if constexpr (F32IN) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = __float_as_uint(w[((size_t)k * COUT + gg * 4 + c) * CIN + p]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = reinterpret_cast<const unsigned *>(w + ((size_t)k * COUT + gg * 4 + c) * CIN)[p];
        }
        wl[b * NQP + p] = v;
    }
    __syncthreads();
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (scale) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sc[c] = scale[g * 4 + c];
            sh[c] = shift[g * 4 + c];
        }
    }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)rec, 0, rec_records * 32, 0x00020000);

    // A workgroup takes rounds of RPB rows in the count order of their 256-row chunk.  The three dependent fetches of a round —
    // position -> row (perm), row -> record, record -> neighbour rows — are spread over three rounds: while round i computes,
    // the record of round i + 1 and the row of round i + 2 are on their way (a round is a few microseconds of latency and a few
    // hundred cycles of arithmetic: unpipelined the kernel sat at 3 round trips per round).
    const int lim = (n + 255) & ~255, step = gridDim.x * RPB;
    auto row_of = [&](int b) -> int {   // row of this lane in the round that starts at position b (-1: no such round)
        const int pos = b + wave * RPW + rsub;
        return b < lim ? (pos & ~255) + (int)perm[pos] : -1;
    };
    auto rec_off = [&](int row) -> unsigned { return (row >= 0 && row < n) ? (unsigned)row * 32u : 0x80000000u; };
    int base = blockIdx.x * RPB;
    int r_cur = row_of(base), r_nx = row_of(base + step);
    u32x4 e0n = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rec_off(r_cur), 0, 0);
    u32x4 e1n = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rec_off(r_cur) + 16u, 0, 0);
    for (; base < lim; base += step) {   // (whole waves stay in the loop: ballots below)
        const int r = r_cur;
        const bool live = r >= 0 && r < n;
        unsigned roff = rec_off(r);   // byte offset of the row's current record
        u32x4 e0 = e0n, e1 = e1n;
        r_cur = r_nx;
        r_nx = row_of(base + 2 * step);
        e0n = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rec_off(r_cur), 0, 0);
        e1n = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rec_off(r_cur) + 16u, 0, 0);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int hop = 0; hop < 4; ++hop) {   // (uniform: chained records, at most 4 for 27 entries)
            if (hop) {
                e0 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0);
                e1 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff + 16u, 0, 0);
            }
            // (an out-of-range record reads as zeros = entry (k 0, row 0): treat the whole lane as empty instead)
            unsigned e[8] = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
            const bool has = roff != 0x80000000u;
            const bool link = has && e[7] != kEmpty && (e[7] >> 27) == kLinkK;   // (the empty pattern has k = 31 too)
            // every neighbour row of the record is requested before the first product (absent slots present an out-of-range
            // offset and cost nothing), and nothing is allowed to sink behind the slot loop's branches: left to itself the
            // compiler put each slot's load right in front of its use, one exposed round trip per slot
            unsigned xv[8][XW];
            bool ok[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned ev = e[j];
                ok[j] = has && ev != kEmpty && !(j == 7 && link);
                const unsigned off = ok[j] ? (ev & kIdxMask) * (unsigned)ROWB : 0x80000000u;
                if constexpr (ROWB % 16 == 0) {
#pragma unroll
                    for (int q = 0; q < ROWB / 16; ++q) {
                        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off + q * 16u, 0, 0);
                        xv[j][q * 4 + 0] = t[0]; xv[j][q * 4 + 1] = t[1]; xv[j][q * 4 + 2] = t[2]; xv[j][q * 4 + 3] = t[3];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < XW; ++q) xv[j][q] = __builtin_amdgcn_raw_buffer_load_b32(xrsrc, off + q * 4u, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (__ballot(ok[j]) == 0ull) continue;   // (uniform) nobody in the wave has this slot
                const int k = ok[j] ? (int)(e[j] >> 27) : 0;
                const u32x4 *wb = wl + (g * 27 + k) * NQP;
                u32x4 wv[NQ];
#pragma unroll
                for (int p = 0; p < NQ; ++p) wv[p] = wb[p];   // (all reads of the slot in flight together)
#pragma unroll
                for (int p = 0; p < NQ; ++p) {
                    if constexpr (F32IN) {
                        const float xc = __uint_as_float(xv[j][p]);   // (absent: 0 -> fmaf(0, w, acc) == acc)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = fmaf(xc, __uint_as_float(wv[p][c]), acc[c]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = dot2(xv[j][p], wv[p][c], acc[c], TIn{});
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (__ballot(link) == 0ull) break;
            roff = link ? ((unsigned)cap + (e[7] & kIdxMask)) * 32u : 0x80000000u;
        }
        if (live) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = scale ? acc[c] * sc[c] + sh[c] : acc[c];
            if (residual) {
                if constexpr (sizeof(TOut) == 2) {
                    const uint2 t = *reinterpret_cast<const uint2 *>(residual + (size_t)r * COUT + g * 4);
                    const TOut *tp = reinterpret_cast<const TOut *>(&t);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = v[c] + ell_to_f32(tp[c]);
                } else {
                    const float4 t = *reinterpret_cast<const float4 *>(residual + (size_t)r * COUT + g * 4);
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                }
            }
            if (relu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = v[c] < 0.f ? 0.f : v[c];
            }
            if constexpr (sizeof(TOut) == 2) {
                TOut o[4] = {ell_from_f32<TOut>(v[0]), ell_from_f32<TOut>(v[1]), ell_from_f32<TOut>(v[2]), ell_from_f32<TOut>(v[3])};
                *reinterpret_cast<uint2 *>(y + (size_t)r * COUT + g * 4) = *reinterpret_cast<const uint2 *>(o);
            } else {
                *reinterpret_cast<float4 *>(y + (size_t)r * COUT + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int CIN, int COUT, typename TIn, typename TOut>
int launch_ell(const void *x, long long x_bytes, const void *w, const unsigned *rec, long long rec_records, const int *n_out, int cap, void *y,
               const float *scale, const float *shift, const void *residual, int relu, hipStream_t s) {
    using W = EllW<CIN, std::is_same<TIn, float>::value>;
    constexpr int lds = 27 * (COUT / 4) * W::NQP * 16;
    constexpr int RPB = (64 / (COUT / 4)) * (kThreads / 64);
    auto kern = spconv_ell_kernel<CIN, COUT, TIn, TOut>;
    // up to 8 workgroups per CU (64 VGPRs), fewer where the weight image is large
    const int per_cu = lds * 8 <= 160 * 1024 ? 8 : (160 * 1024) / lds;
    const int grid = fnp_grid_for(cap, RPB, 256 * per_cu);
    const unsigned char *perm = reinterpret_cast<const unsigned char *>(rec) + (size_t)rec_records * 32;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, s, (const TIn *)x, (int)x_bytes, (const TIn *)w, rec, (int)rec_records, perm, n_out, cap,
                       (TOut *)y, scale, shift, (const TOut *)residual, relu);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

}  // namespace

extern "C" long long fnp_ell_bytes(int cap_rows, int pool_records) {
    if (cap_rows <= 0 || pool_records < 0) return 0;
    return ((long long)cap_rows + pool_records) * 32 + (((long long)cap_rows + 255) & ~255ll);   // records + the chunk orders
}

// SubM 3x3x3 (geom: in_shape == out_shape) or strided 3x3x3 (out_coords from fnp_rulebook_strided with nbr = NULL) rulebook in
// the compact form.  pool_used (one int32, device) ends as the number of extension records asked for.
extern "C" int fnp_rulebook_ell(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom, const fnp_rankgrid *in_grid,
                                void *records, int pool_records, int *pool_used, int pool_used_is_zero, int *nbr, fnp_stream_t stream) {
    if (!coords || !n_rows || cap <= 0 || !geom || !fnp_rg_valid(in_grid) || !records || pool_records < 0 || !pool_used) return FNP_ERR_ARG;
    if (((uintptr_t)records & 15) || (long long)cap + pool_records >= (1ll << 26)) return FNP_ERR_ARG;   // (32-bit byte offsets into the records)
    bool subm = true;
    for (int d = 0; d < 3; ++d) {
        if (geom->ksize[d] != 3 || geom->stride[d] <= 0 || geom->padding[d] < 0) return FNP_ERR_ARG;
        subm = subm && geom->stride[d] == 1 && geom->padding[d] == 1 && geom->in_shape[d] == geom->out_shape[d];
    }
    if (in_grid->D != geom->in_shape[0] || in_grid->H != geom->in_shape[1] || in_grid->W != geom->in_shape[2]) return FNP_ERR_ARG;
    const RG g = fnp_rg_view(in_grid);
    const dim3 grid(fnp_grid_for(cap, kThreads));
    if (!pool_used_is_zero) {   // (the pool counter starts at zero: a fill kernel, not a memset — common.h; a caller that keeps the
                                // counter and has it reset when it reads it, fnp_gather_counts, saves the launch)
        const int frc = fnp_fill_words(pool_used, 1, 0u, (hipStream_t)stream);
        if (frc) return frc;
    }
    if (subm)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(ell_build_kernel<false>), grid, dim3(kThreads), 0, (hipStream_t)stream, coords, n_rows, cap, g, 1, 1, 1, 1, 1, 1,
                           (unsigned *)records, pool_records, pool_used, (unsigned char *)records + ((size_t)cap + pool_records) * 32, nbr);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(ell_build_kernel<true>), grid, dim3(kThreads), 0, (hipStream_t)stream, coords, n_rows, cap, g, geom->stride[0],
                           geom->stride[1], geom->stride[2], geom->padding[0], geom->padding[1], geom->padding[2], (unsigned *)records,
                           pool_records, pool_used, (unsigned char *)records + ((size_t)cap + pool_records) * 32, nbr);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// fnp_spconv_forward on the compact rulebook: K = 27; (Cin, Cout) = (<= 8 f32 point features, 16), (16, 16) or (16, 32) 16-bit
// features; out_dtype = in_dtype for 16-bit inputs, FNP_F32 / FNP_BF16 / FNP_F16 for the f32 first layer.
extern "C" int fnp_spconv_forward_ell(const void *feat_in, int in_dtype, int n_in_rows, const void *weight, const void *records, int cap_rows,
                                      int pool_records, const int *n_out, void *feat_out, int out_dtype, const float *scale, const float *shift,
                                      const void *residual, int relu, int Cin, int Cout, fnp_stream_t stream) {
    if (!feat_in || !weight || !records || !n_out || !feat_out || cap_rows <= 0 || pool_records < 0 || n_in_rows <= 0) return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr)) return FNP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const unsigned *rec = (const unsigned *)records;
    const long long nrec = (long long)cap_rows + pool_records;
    if (nrec >= (1ll << 26) || n_in_rows >= (1 << 27)) return FNP_ERR_ARG;
    if (in_dtype == FNP_F32) {
        const long long xb = (long long)n_in_rows * Cin * 4;
        if (Cout != 16 || xb >= 0x7fffffffll) return FNP_ERR_ARG;
#define FNP_ELL_FIRST(CI)                                                                                                                   \
    if (Cin == CI) {                                                                                                                        \
        if (out_dtype == FNP_F32) return launch_ell<CI, 16, float, float>(feat_in, xb, weight, rec, nrec, n_out, cap_rows, feat_out, scale, shift, residual, relu, s); \
        if (out_dtype == FNP_BF16) return launch_ell<CI, 16, float, __bf16>(feat_in, xb, weight, rec, nrec, n_out, cap_rows, feat_out, scale, shift, residual, relu, s); \
        if (out_dtype == FNP_F16) return launch_ell<CI, 16, float, _Float16>(feat_in, xb, weight, rec, nrec, n_out, cap_rows, feat_out, scale, shift, residual, relu, s); \
        return FNP_ERR_ARG;                                                                                                                 \
    }
        FNP_ELL_FIRST(4)
        FNP_ELL_FIRST(5)
#undef FNP_ELL_FIRST
        return FNP_ERR_ARG;
    }
    if ((in_dtype != FNP_BF16 && in_dtype != FNP_F16) || out_dtype != in_dtype || Cin != 16) return FNP_ERR_ARG;
    const long long xb = (long long)n_in_rows * Cin * 2;
    if (xb >= 0x7fffffffll) return FNP_ERR_ARG;
    if (Cout == 16) {
        if (in_dtype == FNP_BF16) return launch_ell<16, 16, __bf16, __bf16>(feat_in, xb, weight, rec, nrec, n_out, cap_rows, feat_out, scale, shift, residual, relu, s);
        return launch_ell<16, 16, _Float16, _Float16>(feat_in, xb, weight, rec, nrec, n_out, cap_rows, feat_out, scale, shift, residual, relu, s);
    }
    if (Cout == 32) {
        if (in_dtype == FNP_BF16) return launch_ell<16, 32, __bf16, __bf16>(feat_in, xb, weight, rec, nrec, n_out, cap_rows, feat_out, scale, shift, residual, relu, s);
        return launch_ell<16, 32, _Float16, _Float16>(feat_in, xb, weight, rec, nrec, n_out, cap_rows, feat_out, scale, shift, residual, relu, s);
    }
    return FNP_ERR_ARG;
}
