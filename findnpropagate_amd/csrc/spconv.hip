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
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

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
// MFMA path.
//
// Workgroup = 4 waves; wave w owns MB*16 output sites and all COUT channels, accumulators in
// registers for the whole sweep over the K kernel offsets.  The weight slab W_k (COUT x CIN bf16)
// is shared by the 4 waves through LDS:
//   * ALLK  (K*slab <= 64 KiB: the 16/32-channel layers): every slab is staged once per
//     persistent workgroup and the tile loop runs without barriers;
//   * else  (64/128-channel layers): slabs are double-buffered, W_{k+1} is fetched into registers
//     before the MFMAs of offset k and written to the other LDS buffer after them (one barrier
//     per offset), so the HBM/L2 latency of the weights hides under the matrix work.
// LDS image: row r (one output channel, CIN*2 bytes = CH 16-byte chunks) stores logical chunk c
// at physical chunk c ^ ((r >> SW) & (CH-1)); with that XOR every 16-lane group of the
// ds_read_b128 fragment reads (16 rows x one logical chunk) hits 16 distinct 16-byte slots of the
// 256-byte bank row: conflict-free (checked exhaustively for CH = 16, 8, 4, 2).
// Feature fragments are gathered straight from HBM/L2 (16 contiguous bytes per lane, rows absent
// from the rulebook are exec-masked zeros); a 16-site block with no neighbour at offset k skips
// its MFMAs (wave-uniform branch) — in rank-grid row order neighbour presence is spatially
// coherent, so whole blocks drop out.
// KVOL: kernel volume known at compile time (27) or 0 = runtime K; it also gives the 3x3x3
// layers and conv_out (K = 3) distinct kernel names for per-layer-class profiler statistics.
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int KVOL>
struct MfmaCfg {
    static constexpr int CH = CIN / 8;                    // 16-byte chunks per weight row
    static constexpr int SLAB = COUT * CH;                // chunks per slab
    static constexpr int SW = (CH == 8 || CH == 4) ? 1 : 0;
    static constexpr bool ALLK = KVOL > 0 && (long long)KVOL * SLAB * 16 <= 65536;
    static constexpr int LDS_BYTES = (ALLK ? KVOL : 2) * SLAB * 16;
};

#define FNP_AS1(p) ((const __attribute__((address_space(1))) void *)(p))
#define FNP_AS3(p) ((__attribute__((address_space(3))) void *)(p))

template <int CIN, int COUT, int MB, int KVOL, typename TOut>
__global__ __launch_bounds__(256, 2) void spconv_mfma_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                             const int *__restrict__ nbr, int nbr_stride, int Krt,
                                                             const int *__restrict__ n_out, int cap,
                                                             TOut *__restrict__ y, const float *__restrict__ scale,
                                                             const float *__restrict__ shift,
                                                             const TOut *__restrict__ residual, int relu) {
    using Cfg = MfmaCfg<CIN, COUT, KVOL>;
    constexpr int CH = Cfg::CH, SLAB = Cfg::SLAB, SW = Cfg::SW;
    constexpr bool ALLK = Cfg::ALLK;
    constexpr int KS = (CIN + 31) / 32;   // 32-wide K steps of the MFMA
    constexpr int NB = COUT / 16;         // 16-channel output blocks
    constexpr int NBH = NB < 4 ? NB : 4;  // A fragments held at once
    constexpr int ROWS_PER_WAVE = MB * 16;
    constexpr int ROWS_PER_WG = 4 * ROWS_PER_WAVE;
    static_assert(CIN % 16 == 0 && COUT % 16 == 0, "channel counts must be multiples of 16");
    constexpr bool GLDS = !ALLK && SLAB % 256 == 0;   // wide slabs stream through global_load_lds
    constexpr bool RSTG = !ALLK && !GLDS;             // small slab with runtime K: register staging
    static_assert(ALLK || GLDS || SLAB < 256, "unsupported slab size");
    constexpr int NSRC = GLDS ? SLAB / 256 : 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char fnp_smem[];
    uint4 *wl = reinterpret_cast<uint4 *>(fnp_smem);

    const int K = KVOL > 0 ? KVOL : Krt;
    const int n = min(*n_out, cap);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int tiles = (n + ROWS_PER_WG - 1) / ROWS_PER_WG;
    const bool kvalid0 = (q * 8) < CIN;  // for CIN == 16 only lanes 0..31 carry data in a K step
    if ((int)blockIdx.x >= tiles) return;  // before any barrier: safe early exit

#define FNP_LDS_POS(row, chunk) ((row) * CH + ((chunk) ^ (((row) >> SW) & (CH - 1))))
    if (ALLK) {
        // narrow layers: all K slabs resident in LDS for the lifetime of the workgroup
        for (int p = tid; p < K * SLAB; p += 256) {
            const int kk = p / SLAB, r = p % SLAB;
            wl[kk * SLAB + FNP_LDS_POS(r / CH, r % CH)] = reinterpret_cast<const uint4 *>(w)[p];
        }
        __syncthreads();
    }
    // wide layers: slab k+1 streams HBM/L2 -> LDS directly (global_load_lds, 1 KiB per wave
    // instruction).  The LDS image is linear in lane order, so the swizzle is applied to each
    // lane's SOURCE chunk instead.
    int src_chunk[NSRC];
    if (GLDS) {
#pragma unroll
        for (int j = 0; j < NSRC; ++j) {
            const int p = (j * 4 + wave) * 64 + lane;          // LDS position this lane fills
            const int row = p / CH, phys = p % CH;
            src_chunk[j] = row * CH + (phys ^ ((row >> SW) & (CH - 1)));
        }
    }
#define FNP_STAGE(kk, slot)                                                                                   \
    if (GLDS) {                                                                                                \
        const uint4 *src__ = reinterpret_cast<const uint4 *>(w + (size_t)(kk) * COUT * CIN);                   \
        _Pragma("unroll") for (int j = 0; j < NSRC; ++j)                                                       \
            __builtin_amdgcn_global_load_lds(FNP_AS1(src__ + src_chunk[j]),                                    \
                                             FNP_AS3(wl + (slot) * SLAB + (j * 4 + wave) * 64), 16, 0, 0);     \
    } else if (RSTG && tid < SLAB) {                                                                           \
        wl[(slot) * SLAB + FNP_LDS_POS(tid / CH, tid % CH)] =                                                  \
            reinterpret_cast<const uint4 *>(w + (size_t)(kk) * COUT * CIN)[tid];                               \
    }

    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int row0 = tile * ROWS_PER_WG + wave * ROWS_PER_WAVE;
        f32x4 acc[NB][MB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        int idx_cur[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int r = row0 + mb * 16 + l15;
            idx_cur[mb] = r < n ? nbr[r] : -1;
        }
        if (!ALLK) {
            FNP_STAGE(0, 0)
            __syncthreads();
        }

        for (int k = 0; k < K; ++k) {
            const bool more = k + 1 < K;
            int idx_nxt[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int r = row0 + mb * 16 + l15;
                idx_nxt[mb] = (more && r < n) ? nbr[(size_t)(k + 1) * nbr_stride + r] : -1;
            }
            unsigned has = 0;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) has |= (__ballot(idx_cur[mb] >= 0) != 0ull) ? (1u << mb) : 0u;
            if (!ALLK && more) FNP_STAGE(k + 1, (k + 1) & 1)
            if (has) {
                const uint4 *wk = wl + (ALLK ? k : (k & 1)) * SLAB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bool kvalid = KS > 1 ? true : kvalid0;
                    const int chunk = ks * 4 + q;
                    bf16x8 xb[MB];
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (idx_cur[mb] >= 0 && kvalid)
                            v = *reinterpret_cast<const bf16x8 *>(x + (size_t)idx_cur[mb] * CIN + chunk * 8);
                        xb[mb] = v;
                    }
#pragma unroll
                    for (int h = 0; h < NB; h += NBH) {
                        bf16x8 wa[NBH];
#pragma unroll
                        for (int j = 0; j < NBH; ++j) {
                            const int row = (h + j) * 16 + l15;
                            uint4 t = make_uint4(0u, 0u, 0u, 0u);
                            if (kvalid) t = wk[FNP_LDS_POS(row, chunk)];
                            wa[j] = *reinterpret_cast<bf16x8 *>(&t);
                        }
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) {
                            if (has & (1u << mb)) {
#pragma unroll
                                for (int j = 0; j < NBH; ++j)
                                    acc[h + j][mb] =
                                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], xb[mb], acc[h + j][mb], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (!ALLK) __syncthreads();  // also drains this wave's global_load_lds (vmcnt(0))
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) idx_cur[mb] = idx_nxt[mb];
        }
#undef FNP_STAGE
#undef FNP_LDS_POS

        // epilogue: lane holds out[site = row0 + mb*16 + l15][c0 .. c0+3], c0 = nb*16 + q*4
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
                const int r = row0 + mb * 16 + l15;
                if (r >= n) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = scale ? acc[nb][mb][j] * sc[j] + sh[j] : acc[nb][mb][j];
                if (residual) {
                    const TOut *rp = residual + (size_t)r * COUT + c0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] + to_f32(rp[j]);
                }
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];
                }
                TOut *yp = y + (size_t)r * COUT + c0;
                if constexpr (sizeof(TOut) == 2) {
                    bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4 *>(yp) = o;
                } else {
                    *reinterpret_cast<float4 *>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

template <int CIN, int COUT, int KVOL, typename TOut>
int launch_mfma_k(const void *x, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap,
                  void *y, const float *scale, const float *shift, const void *residual, int relu, hipStream_t s) {
    constexpr int MB = 4;
    using Cfg = MfmaCfg<CIN, COUT, KVOL>;
    auto kern = spconv_mfma_kernel<CIN, COUT, MB, KVOL, TOut>;
    const int tiles = fnp_divup(cap, 4 * MB * 16);
    // persistent grid: two workgroups per CU are resident (register / LDS budget of the wide
    // layers); the narrow ALLK layers stage all weights once per workgroup, so keep them few too
    const int grid = tiles < 512 ? tiles : 512;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, s, (const __bf16 *)x, (const __bf16 *)w, nbr,
                       nbr_stride, K, n_out, cap, (TOut *)y, scale, shift, (const TOut *)residual, relu);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

template <int CIN, int COUT, typename TOut>
int launch_mfma(const void *x, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap, void *y,
                const float *scale, const float *shift, const void *residual, int relu, hipStream_t s) {
    if (K == 27)
        return launch_mfma_k<CIN, COUT, 27, TOut>(x, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift, residual, relu, s);
    return launch_mfma_k<CIN, COUT, 0, TOut>(x, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift, residual, relu, s);
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

template <typename TOut>
int dispatch_bf16(const void *x, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap,
                  void *y, const float *scale, const float *shift, const void *residual, int relu, int Cin, int Cout,
                  hipStream_t s) {
#define FNP_CASE(CI, CO)                                                                                       \
    if (Cin == CI && Cout == CO)                                                                               \
        return launch_mfma<CI, CO, TOut>(x, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift, residual, relu, s);
    FNP_CASE(16, 16)
    FNP_CASE(16, 32)
    FNP_CASE(32, 32)
    FNP_CASE(32, 64)
    FNP_CASE(64, 64)
    FNP_CASE(64, 128)
    FNP_CASE(128, 128)
#undef FNP_CASE
    return launch_valu<__bf16, TOut>(x, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift, residual, relu, Cin, Cout, s);
}

// SparseConvTensor.dense(): thread per (row, channel).
template <typename T>
__global__ __launch_bounds__(256) void dense_kernel(const T *__restrict__ feats, const int *__restrict__ coords,
                                                    const int *__restrict__ n_rows, int cap, int C, int D, int H, int W,
                                                    T *__restrict__ out) {
    const int n = min(*n_rows, cap);
    const long long total = (long long)n * C;
    const long long vol = (long long)D * H * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / C), c = (int)(t % C);
        const int4 cd = reinterpret_cast<const int4 *>(coords)[row];
        const long long sp = ((long long)cd.y * H + cd.z) * W + cd.w;
        out[((long long)cd.x * C + c) * vol + sp] = feats[t];
    }
}

}  // namespace

extern "C" int fnp_spconv_forward(const void *feat_in, int in_dtype, const void *weight, const int *nbr, int nbr_stride,
                                  int K, const int *n_out, int cap_out, void *feat_out, int out_dtype,
                                  const float *scale, const float *shift, const void *residual, int relu, int Cin,
                                  int Cout, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!feat_in || !weight || !nbr || !n_out || !feat_out || K <= 0 || Cin <= 0 || Cout <= 0 || cap_out <= 0 ||
        nbr_stride < cap_out)
        return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr)) return FNP_ERR_ARG;
    if (in_dtype == FNP_F32) {
        if (out_dtype == FNP_F32)
            return launch_valu<float, float>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                             residual, relu, Cin, Cout, s);
        if (out_dtype == FNP_BF16)
            return launch_valu<float, __bf16>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale,
                                              shift, residual, relu, Cin, Cout, s);
        return FNP_ERR_ARG;
    }
    if (in_dtype == FNP_BF16) {
        if (out_dtype == FNP_BF16)
            return dispatch_bf16<__bf16>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                         residual, relu, Cin, Cout, s);
        if (out_dtype == FNP_F32)
            return dispatch_bf16<float>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                        residual, relu, Cin, Cout, s);
        return FNP_ERR_ARG;
    }
    return FNP_ERR_ARG;
}

extern "C" int fnp_sparse_to_dense(const void *feats, int dtype, const int *coords, const int *n_rows, int cap, int C,
                                   int B, int D, int H, int W, void *out, fnp_stream_t stream) {
    if (!feats || !coords || !n_rows || !out || cap <= 0 || C <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0)
        return FNP_ERR_ARG;
    const int grid = fnp_grid_for((long long)cap * C, 256, 256 * 16);
    if (dtype == FNP_F32)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dense_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float *)feats, coords, n_rows, cap, C, D, H, W, (float *)out);
    else if (dtype == FNP_BF16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dense_kernel<__bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16 *)feats, coords, n_rows, cap, C, D, H, W, (__bf16 *)out);
    else
        return FNP_ERR_ARG;
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
